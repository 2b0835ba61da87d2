"""Ad-hoc (not collected): register-resident kernel tuning."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from iris_amd import HipWavefront, Render, Scene
scene = Scene.dummy()
W, H, SPP = 1920, 1080, int(os.environ.get("SPP", "256"))
r = Render(W, H, SPP, scene)
with HipWavefront(0, mode=1, blocks_per_cu=2, slots_per_block=2048) as g:
    g.upload(scene); g.render_film(r); ref = g.render_film(r); st = g.stats()
    print("pool 2x2048: kernel %8.2f ms %8.1f Msamples/s" % (st.kernel_ms, W * H * SPP / st.kernel_ms / 1e3), flush=True)
for rq in (0, 1):
    for rb in (2, 8, 32):
        with HipWavefront(0, mode=2, reg_blocks_per_cu=rb, reg_queue=rq) as g:
            g.upload(scene); g.render_film(r); film = g.render_film(r); st = g.stats()
            print("reg queue=%d blocks/cu %3d: kernel %8.2f ms %8.1f Msamples/s same=%s" % (rq, rb, st.kernel_ms, W * H * SPP / st.kernel_ms / 1e3, np.array_equal(film, ref)), flush=True)

"""Ad-hoc tuning sweep (not collected): pool geometry vs throughput on config B."""
import sys, os, time, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from iris_amd import HipWavefront, Render, Scene
scene = Scene.dummy()
W, H, SPP = 1920, 1080, int(os.environ.get("SPP", "256"))
r = Render(W, H, SPP, scene)
ref = None
for bpc, spb in itertools.product((1, 2, 3, 4, 8), (256, 512, 1024, 2048, 4096)):
    if bpc * spb > 16384: continue
    with HipWavefront(0, blocks_per_cu=bpc, slots_per_block=spb) as g:
        g.upload(scene)
        g.render_film(r)
        film = g.render_film(r)
        st = g.stats()
        if ref is None: ref = film
        print("blocks/cu %2d slots/block %5d pool %6.1f MB: kernel %8.2f ms  %8.1f Msamples/s  same=%s" % (
            bpc, spb, bpc * 256 * spb * 120 / 1e6, st.kernel_ms, W * H * SPP / st.kernel_ms / 1e3, np.array_equal(film, ref)), flush=True)

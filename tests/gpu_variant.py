"""Ad-hoc (not collected): time one library variant: IRIS_HIP_LIB=... python tests/gpu_variant.py bpc spb [spp]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from iris_amd import HipWavefront, Render, Scene
scene = {"s1": Scene.dummy, "s2": Scene.many_spheres, "s3": Scene.glossy_dielectric}[os.environ.get("SCENE", "s1")]()
W, H = 1920, 1080
SPP = int(sys.argv[3]) if len(sys.argv) > 3 else 256
for bpc, spb in [tuple(int(v) for v in a.split("x")) for a in sys.argv[1].split(",")]:
    with HipWavefront(0, blocks_per_cu=bpc, slots_per_block=spb) as g:
        g.upload(scene)
        r = Render(W, H, SPP, scene, max_depth=32 if os.environ.get("SCENE") == "s3" else 15)
        g.render_film(r)
        film = g.render_film(r)
        st = g.stats()
        print("%s  %dx%d: kernel %8.2f ms %8.1f Msamples/s  cksum %.6f" % (os.path.basename(os.environ.get("IRIS_HIP_LIB", "default")), bpc, spb, st.kernel_ms, W * H * SPP / st.kernel_ms / 1e3, float(film.astype(np.float64).sum())), flush=True)

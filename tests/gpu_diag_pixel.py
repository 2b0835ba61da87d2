"""Diagnostic (not collected): find the samples of one pixel whose GPU trace differs from the oracle's."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import oracle_ffi as O
from iris_amd import HipWavefront, Render, Scene

x, y, w, h, spp = (int(v) for v in (sys.argv[1:6] if len(sys.argv) > 5 else (1037, 920, 1920, 1080, 1024)))
O.set_trig_mode(O.TRIG_IRIS)
gpu = HipWavefront(0)
scene = Scene.dummy()
flat = gpu.upload(scene)
osc = O.OracleScene(flat)
r = Render(w, h, spp, scene)
ocam = O.camera_new([0, 0, 0], np.float32(w) / np.float32(h))
p = O.make_params(w, h, spp)
xys = [(x, y, s) for s in range(spp)]
gt = gpu.dbg_trace(r, xys)
nbad = 0
for s, a in enumerate(gt):
    b = osc.trace(ocam, p, x, y, s)
    ga = np.array(list(a.radiance) + list(a.xyz), np.float32).view(np.uint32)
    gb = np.array(list(b.radiance) + list(b.xyz), np.float32).view(np.uint32)
    ta = np.array(list(a.hit_t[:a.n_hits]), np.float32).view(np.uint32)
    tb = np.array(list(b.hit_t[:b.n_hits]), np.float32).view(np.uint32)
    if a.n_hits != b.n_hits or not np.array_equal(ga, gb) or not np.array_equal(ta, tb) or a.draws != b.draws:
        nbad += 1
        print("sample", s, "n_hits", a.n_hits, b.n_hits, "draws", a.draws, b.draws)
        print("  prims", list(a.hit_prim[:a.n_hits]), list(b.hit_prim[:b.n_hits]))
        print("  t gpu", list(a.hit_t[:a.n_hits])); print("  t orc", list(b.hit_t[:b.n_hits]))
        print("  t bits diff", (ta.astype(np.int64) - tb.astype(np.int64)).tolist() if len(ta) == len(tb) else "len")
        print("  rad gpu", list(a.radiance), "\n  rad orc", list(b.radiance))
        print("  rad/xyz bit diff", (ga.astype(np.int64) - gb.astype(np.int64)).tolist())
print("bad samples:", nbad, "of", spp)

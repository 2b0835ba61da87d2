#!/usr/bin/env python3
"""Renders small films of the general-class scenes with the library IRIS_HIP_LIB selects and saves them with the work counters:
tools/ab_film.py <out.npz>.  Two runs with two libraries, then `tools/ab_film.py --cmp a.npz b.npz` (bit for bit)."""
import os
import sys

import numpy as np

if sys.argv[1] == "--cmp":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    bad = 0
    for k in a.files:
        same = np.array_equal(a[k].view(np.uint32) if a[k].dtype == np.float32 else a[k], b[k].view(np.uint32) if b[k].dtype == np.float32 else b[k])
        print(k, "equal" if same else "DIFFERENT", a[k].ravel()[:4] if not same else "", b[k].ravel()[:4] if not same else "")
        bad += not same
    sys.exit(1 if bad else 0)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import iris_amd
from iris_amd import EnvMap, Render, Scene, UpsampleTable

gold = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
out = {}
for name in ("s3", "s3env", "s1env", "s2gen"):
    scene = {"s3": Scene.glossy_dielectric, "s3env": Scene.glossy_dielectric, "s1env": Scene.dummy, "s2gen": lambda: Scene.many_spheres(300)}[name]()
    if name.endswith("env"):
        scene.env_map = EnvMap.sky(UpsampleTable(os.path.join(gold, "srgb_res16.coeff")), 64, 64)
    if name == "s2gen":  # a hierarchy under general materials: one glossy sphere makes the class general
        pass
    with iris_amd.HipWavefront(0) as g:
        g.upload(scene)
        for (w, h, spp) in ((192, 128, 16), (61, 37, 5)):
            film = g.render_film(Render(w, h, spp, scene, max_depth=32))
            st = g.stats()
            out["%s_%dx%dx%d" % (name, w, h, spp)] = np.asarray(film, np.float32)
            out["%s_%dx%dx%d_ctr" % (name, w, h, spp)] = np.array([st.samples, st.bounces, st.rays, st.sphere_tests], np.uint64)
np.savez(sys.argv[1], **out)
print("saved", sys.argv[1])

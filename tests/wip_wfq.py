"""Work-in-progress check of the queue kernels (mode 3) against the register-resident kernel (mode 2): run by name."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import iris_amd
from iris_amd import Render, Scene


def bits(a):
    return np.asarray(a, dtype=np.float32).view(np.uint32)


def stats_tuple(st):
    return (st.samples, st.bounces, st.rays, st.sphere_tests)


def check(scene, w, h, spp, integrator=0, max_depth=15, **opts):
    r = Render(w, h, spp, scene, integrator=integrator, max_depth=max_depth)
    with iris_amd.HipWavefront(0, mode=2) as ref:
        ref.upload(scene)
        want = ref.render_film(r)
        ws = ref.stats()
    with iris_amd.HipWavefront(0, mode=3, **opts) as g:
        g.upload(scene)
        t0 = time.time()
        film = g.render_film(r)
        dt = time.time() - t0
        st = g.stats()
    ok = np.array_equal(bits(film), bits(want)) and stats_tuple(st) == stats_tuple(ws)
    print("%s %dx%dx%d integ %d opts %s: kernel %d launches %d  %.3f s  film %s stats %s %s" % (
        "OK  " if ok else "FAIL", w, h, spp, integrator, opts, st.kernel, st.launches, dt,
        np.array_equal(bits(film), bits(want)), stats_tuple(st), stats_tuple(ws)), flush=True)
    if not ok:
        d = np.argwhere(bits(film) != bits(want))
        print("   differing entries:", len(d), d[:5].tolist(), flush=True)
    return ok


if __name__ == "__main__":
    small = Scene.many_spheres(600)
    ok = check(small, 16, 16, 1)
    ok &= check(small, 96, 64, 6)
    if ok:
        for integ in (() if "integ0" in sys.argv else (1, 2, 3)):
            ok &= check(small, 48, 32, 4, integrator=integ)
        ok &= check(small, 96, 64, 6, wfq_slots=256, wfq_pipes=1)
        ok &= check(small, 96, 64, 6, wfq_slots=1024, wfq_pipes=4, wfq_batch=3)
        ok &= check(small, 70, 33, 5, wfq_pipes=1, wfq_batch=1)
        ok &= check(small, 64, 64, 4, max_depth=1)
        big = Scene.many_spheres(10_000)
        ok &= check(big, 40, 24, 2)
        ok &= check(big, 640, 360, 8)
        ok &= check(big, 640, 360, 8, wfq_slots=65536, wfq_pipes=2)
    print("ALL OK" if ok else "FAILED")
    sys.exit(0 if ok else 1)

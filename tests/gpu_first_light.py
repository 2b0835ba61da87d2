"""Ad-hoc first-light script (not collected by pytest): python tests/gpu_first_light.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import iris_amd, oracle_ffi as O
from iris_amd import HipWavefront, Render, Scene

O.set_trig_mode(O.TRIG_IRIS)
gpu = HipWavefront(0)
scene = Scene.dummy()
flat = gpu.upload(scene)
osc = O.OracleScene(flat)
r = Render(64, 64, 8, scene)
ocam = O.camera_new([0, 0, 0], 1.0)
print("cam equal", np.array_equal(np.array(ocam.clip_to_world, np.float32).reshape(4, 4), r.camera.clip_to_world))
xys = [(x, y, s) for x in (0, 13, 63) for y in (0, 31, 63) for s in (0, 1, 7)]
d = gpu.dbg_sampler(xys, r.seed, 40)
od = np.array([O.sampler_draws(x, y, s, r.seed, 40) for x, y, s in xys])
print("sampler bit-exact:", np.array_equal(d.view(np.uint32), od.view(np.uint32)))
hero, o, dd = gpu.dbg_raygen(r, xys)
p = O.make_params(64, 64, 8)
tr = [osc.trace(ocam, p, *q) for q in xys]
print("hero", np.array_equal(hero, np.array([list(t.hero) for t in tr], np.float32)),
      "dir", np.array_equal(dd, np.array([list(t.ray_d) for t in tr], np.float32)))
gt = gpu.dbg_trace(r, xys)
ok = True
for a, b in zip(gt, tr):
    same = (a.n_hits == b.n_hits and list(a.hit_prim[:a.n_hits]) == list(b.hit_prim[:b.n_hits]) and
            np.array_equal(np.array(a.hit_t[:a.n_hits]), np.array(b.hit_t[:b.n_hits])) and
            list(a.radiance) == list(b.radiance) and list(a.xyz) == list(b.xyz) and a.draws == b.draws)
    if not same:
        ok = False
        print("MISMATCH", a.n_hits, b.n_hits, list(a.hit_prim[:4]), list(b.hit_prim[:4]), list(a.radiance), list(b.radiance), a.draws, b.draws)
print("trace bit-exact:", ok)
t0 = time.time(); film = gpu.render_film(r); t1 = time.time()
ofilm, _, secs, st = osc.render(ocam, p, nthreads=8)
gs = gpu.stats()
print("film bit-exact:", np.array_equal(film.view(np.uint32), ofilm.view(np.uint32)), "maxabs", np.abs(film - ofilm).max(), "mean", film.mean(), ofilm.mean())
print("stats gpu", gs.samples, gs.bounces, gs.rays, gs.sphere_tests, "oracle", st.samples, st.bounces, st.rays, st.sphere_tests, "kernel_ms", gs.kernel_ms)
for W, H, spp in ((512, 512, 64), (1920, 1080, 64)):
    r = Render(W, H, spp, scene)
    gpu.render_film(r)
    t0 = time.time(); film = gpu.render_film(r); t1 = time.time(); gs = gpu.stats()
    print(W, H, spp, "wall %.3fs  kernel %.1f ms  resolve %.1f ms  %.1f Msamples/s (kernel)" % (t1 - t0, gs.kernel_ms, gs.resolve_ms, W * H * spp / gs.kernel_ms / 1e3),
          "bounces/sample %.2f rays/sample %.2f" % (gs.bounces / gs.samples, gs.rays / gs.samples), "nan", int(np.isnan(film).sum()))

"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI, against the
CPU oracle on the same inputs, and against the committed golden fixtures.

Bar: BIT-EXACT -- sampler draws, wavelengths, primary rays, hit indices and t, per-sample radiance and
XYZ, the whole film and the work counters.  The oracle runs in ORC_TRIG_IRIS mode (the sin/cos of
docs/SPEC.md, which the kernels implement); test_film_vs_libm_trig_oracle bounds the difference to the
oracle run with this machine's libm sin/cos (what the reference's f32::sin/cos call)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.fixture
def iris_trig(oracle):
    """The oracle with the specified sin/cos (docs/SPEC.md) -- the mode the kernels match bit for bit."""
    oracle.set_trig_mode(oracle.TRIG_IRIS)
    yield oracle
    oracle.set_trig_mode(oracle.TRIG_LIBM)


@pytest.fixture
def libm_trig(oracle):
    """The oracle with this machine's libm sinf/cosf (what the reference's f32::sin/cos call on Linux)."""
    oracle.set_trig_mode(oracle.TRIG_LIBM)
    yield oracle
    oracle.set_trig_mode(oracle.TRIG_LIBM)


def random_xys(rng, w, h, spp, n):
    xys = rng.randint(0, [w, h, spp], size=(n, 3)).astype(np.uint32)
    xys[0] = (0, 0, 0)
    xys[1] = (w - 1, h - 1, spp - 1)
    return xys


def make(scene_name):
    from iris_amd import Scene, UpsampleTable
    if scene_name == "coloured":  # sigmoid-polynomial spectra from the upsampling table (rows a25, a26)
        return Scene.coloured(UpsampleTable(os.path.join(GOLD, "srgb_res16.coeff")))
    return {"s1": Scene.dummy, "s3": Scene.glossy_dielectric, "furnace": Scene.furnace,
            "s2_small": lambda: Scene.many_spheres(600), "s2_chunked": lambda: Scene.many_spheres(2500)}[scene_name]()


def compare_traces(gt, ot, depth=64):
    for a, b in zip(gt, ot):
        assert list(a.hero) == list(b.hero)
        assert list(a.ray_o) == list(b.ray_o) and list(a.ray_d) == list(b.ray_d)
        assert a.n_hits == b.n_hits
        n = min(a.n_hits, depth)
        assert list(a.hit_prim[:n]) == list(b.hit_prim[:n])
        assert np.array_equal(bits(list(a.hit_t[:n])), bits(list(b.hit_t[:n])))
        assert np.array_equal(bits(list(a.radiance)), bits(list(b.radiance)))
        assert np.array_equal(bits(list(a.xyz)), bits(list(b.xyz)))
        assert a.draws == b.draws


# ---- stage level ---------------------------------------------------------------------------------

def test_sampler_bit_exact(gpu, iris_trig):
    rng = np.random.RandomState(11)
    xys = random_xys(rng, 3840, 2160, 4096, 300)
    xys[2] = (5, 9, 65535)
    xys[3] = (5, 9, 70000)  # index above the 16-bit Sobol range
    got = gpu.dbg_sampler(xys, 123456789, 4 * 66 + 3)  # crosses into the hash fallback at set 64
    want = np.array([iris_trig.sampler_draws(int(x), int(y), int(s), 123456789, 4 * 66 + 3) for x, y, s in xys])
    assert np.array_equal(bits(got), bits(want))
    got2 = gpu.dbg_sampler(xys[:16], 7, 9)
    want2 = np.array([iris_trig.sampler_draws(int(x), int(y), int(s), 7, 9) for x, y, s in xys[:16]])
    assert np.array_equal(bits(got2), bits(want2))


@pytest.mark.parametrize("w,h,spp", [(512, 512, 64), (1920, 1080, 1024), (3840, 2160, 4096), (70, 33, 5)])
def test_raygen_bit_exact(gpu, iris_trig, w, h, spp):
    """Pixel sampling, wavelength sampling and primary rays (tile.rs:134-157) for the BASELINE film sizes."""
    from iris_amd import Render, Scene
    r = Render(w, h, spp, Scene.dummy())
    xys = random_xys(np.random.RandomState(w), w, h, spp, 400)
    hero, o, d = gpu.dbg_raygen(r, xys)
    ocam = iris_trig.camera_new([0, 0, 0], np.float32(w) / np.float32(h))
    osc = iris_trig.OracleScene.dummy()
    p = iris_trig.make_params(w, h, spp)
    tr = [osc.trace(ocam, p, int(x), int(y), int(s)) for x, y, s in xys]
    assert np.array_equal(bits(hero), bits(np.array([list(t.hero) for t in tr])))
    assert np.array_equal(bits(o), bits(np.array([list(t.ray_o) for t in tr])))
    assert np.array_equal(bits(d), bits(np.array([list(t.ray_d) for t in tr])))


@pytest.fixture(scope="module")
def gpu_list():
    """A context that never builds a hierarchy: every scan tests every sphere, like scene.rs:116-166."""
    import iris_amd
    ctx = iris_amd.HipWavefront(device=0, bvh_min_prims=0)
    yield ctx
    ctx.close()


@pytest.mark.parametrize("scene_name,hierarchy", [("s1", True), ("s3", True), ("s2_small", True), ("s2_chunked", True),
                                                  ("s2_small", False), ("s2_chunked", False)])
def test_intersect_bit_exact(gpu, gpu_list, iris_trig, scene_name, hierarchy):
    """Scene::intersection (scene.rs:116-132): hit index (lowest index on ties) and t on identical rays,
    including a list longer than the LDS window (s2_chunked) and rays from inside / grazing spheres; through
    the hierarchy (the default for more than 64 spheres) and through the list scans."""
    gpu = gpu if hierarchy else gpu_list
    scene = make(scene_name)
    flat = gpu.upload(scene)
    osc = iris_trig.OracleScene(flat)
    rng = np.random.RandomState(5)
    n = 2000
    o = rng.uniform(-3, 3, size=(n, 3)).astype(np.float32)
    o[: n // 4] = 0.0
    o[n // 4: n // 2] = flat.spheres[rng.randint(0, len(flat.spheres), n // 4), :3]  # from sphere centres
    d = rng.standard_normal((n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
    prim, t = gpu.dbg_intersect(o, d)
    want = [osc.intersect(o[i], d[i])[:2] for i in range(n)]
    assert prim.tolist() == [w[0] for w in want]
    assert np.array_equal(bits(t), bits(np.array([w[1] for w in want])))
    assert (prim >= 0).sum() > n // 10


def test_intersect_tie_breaks_to_lowest_index(gpu, iris_trig):
    """Two identical spheres: the strict '<' of scene.rs:123 keeps the first."""
    from iris_amd import ConstantSpectrum, LambertianBsdf, Scene, Sphere
    sc = Scene()
    for _ in range(2):
        sc.add_material(Sphere((0.0, 0.0, 5.0), 1.0), LambertianBsdf(ConstantSpectrum(0.5)))
    sc.add_emissive_material(Sphere((0.0, 4.0, 5.0), 1.0), LambertianBsdf(ConstantSpectrum(0.5)), ConstantSpectrum(1.0))
    gpu.upload(sc)
    prim, t = gpu.dbg_intersect([[0, 0, 0]], [[0, 0, 1]])
    assert prim[0] == 0 and t[0] == 4.0


def test_hierarchy_keeps_the_tie_rule_and_the_closest_hit(gpu, iris_trig):
    """Many coincident and overlapping spheres scattered over the leaves of the hierarchy: the traversal visits
    them in any order, yet the lowest index among equal t must win, exactly as the list scan decides."""
    from iris_amd import ConstantSpectrum, LambertianBsdf, Scene, Sphere
    rng = np.random.RandomState(9)
    sc = Scene()
    sc.add_emissive_material(Sphere((0.0, 9.0, 5.0), 1.0), LambertianBsdf(ConstantSpectrum(0.5)), ConstantSpectrum(1.0))
    centres = rng.uniform(-3, 3, size=(40, 3)).astype(np.float32) + np.float32([0, 0, 8])
    for k in range(400):  # every centre appears ten times, interleaved, with one of two radii
        c = centres[k % 40]
        sc.add_material(Sphere((float(c[0]), float(c[1]), float(c[2])), 0.75 if (k // 40) % 2 else 0.5), LambertianBsdf(ConstantSpectrum(0.5)))
    flat = gpu.upload(sc)
    osc = iris_trig.OracleScene(flat)
    n = 4000
    o = np.zeros((n, 3), np.float32)
    o[n // 2:] = centres[rng.randint(0, 40, n - n // 2)]  # from inside the stacks of spheres
    d = (centres[rng.randint(0, 40, n)] + rng.normal(0, 0.3, (n, 3))).astype(np.float32) - o
    d[n // 2:] = rng.standard_normal((n - n // 2, 3))
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    d[:8] = np.float32([[0, 0, 1], [0, 1, 0], [1, 0, 0], [0, 0, -1], [0, -1, 0], [-1, 0, 0], [0, 0.6, 0.8], [0.6, 0, 0.8]])  # axis-parallel rays
    prim, t = gpu.dbg_intersect(o, d)
    want = [osc.intersect(o[i], d[i])[:2] for i in range(n)]
    assert prim.tolist() == [w[0] for w in want]
    assert np.array_equal(bits(t), bits(np.array([w[1] for w in want])))
    assert (prim >= 0).mean() > 0.5


@pytest.mark.parametrize("scene_name,integrator,max_depth", [
    ("s1", 0, 15), ("s1", 1, 15), ("s1", 2, 15), ("s1", 3, 15), ("s3", 0, 32), ("s3", 2, 32), ("s3", 1, 32),
    ("furnace", 0, 15), ("s2_small", 0, 15), ("coloured", 0, 15), ("coloured", 3, 15)])
def test_path_trace_bit_exact(gpu, iris_trig, scene_name, integrator, max_depth):
    """Whole camera samples: hit indices and t per bounce, sampler draw count, 4-wavelength radiance and
    XYZ, for every integrator (integrator/*.rs) and every BSDF (bsdf/*.rs; s3 has all of them)."""
    from iris_amd import Render
    scene = make(scene_name)
    flat = gpu.upload(scene)
    osc = iris_trig.OracleScene(flat)
    w, h, spp = 256, 192, 64
    r = Render(w, h, spp, scene, integrator=integrator, max_depth=max_depth)
    xys = random_xys(np.random.RandomState(integrator + max_depth), w, h, spp, 1500)
    gt = gpu.dbg_trace(r, xys)
    ocam = iris_trig.camera_new([0, 0, 0], np.float32(w) / np.float32(h))
    p = iris_trig.make_params(w, h, spp, integrator=integrator, max_depth=max_depth)
    ot = [osc.trace(ocam, p, int(x), int(y), int(s)) for x, y, s in xys]
    compare_traces(gt, ot)
    assert max(t.n_hits for t in ot) >= 3


def test_traces_match_committed_golden(gpu):
    from iris_amd import Render, Scene
    g = np.load(os.path.join(GOLD, "traces_s1.npz"))
    gpu.upload(Scene.dummy())
    gt = gpu.dbg_trace(Render(512, 512, 64, Scene.dummy()), g["xys"])
    for i, a in enumerate(gt):
        assert np.array_equal(bits(list(a.hero)), bits(g["hero"][i]))
        assert np.array_equal(bits(list(a.ray_d)), bits(g["ray_d"][i]))
        assert a.n_hits == g["n_hits"][i] and a.draws == g["draws"][i]
        n = min(int(a.n_hits), 16)
        assert list(a.hit_prim[:n]) == g["hit_prim"][i][:n].tolist()
        assert np.array_equal(bits(list(a.hit_t[:n])), bits(g["hit_t"][i][:n]))
        assert np.array_equal(bits(list(a.radiance)), bits(g["radiance"][i]))
        assert np.array_equal(bits(list(a.xyz)), bits(g["xyz"][i]))


# ---- whole films -----------------------------------------------------------------------------------

def test_traces_match_the_independent_numpy_restatement(gpu):
    """The GPU against tests/golden/traces_s1_restated.npz: whole HwssNaive camera samples on the default scene computed by
    tests/golden/restate_path.py in numpy float32 from the Rust text -- no C code (neither the oracle nor the kernels)
    produced these numbers.  Bit for bit: wavelengths, primary ray, primitive and t of every bounce, draw count, radiance, XYZ."""
    from iris_amd import Camera, Render, Scene
    g = np.load(os.path.join(GOLD, "traces_s1_restated.npz"))
    scene = Scene.dummy()
    gpu.upload(scene)
    w, h = int(g["width"]), int(g["height"])
    r = Render(w, h, 64, scene, camera=Camera([float(v) for v in g["cam_pos"]], np.float32(w) / np.float32(h)))
    assert np.array_equal(bits(r.camera.clip_to_world), bits(g["clip_to_world"]))
    got = gpu.dbg_trace(r, g["xys"])
    for i, t in enumerate(got):
        n = int(g["n_hits"][i])
        assert t.n_hits == n and t.draws == int(g["draws"][i]), i
        assert list(t.hit_prim[:n]) == list(g["hit_prim"][i][:n])
        assert np.array_equal(bits(list(t.hit_t[:n])), bits(g["hit_t"][i][:n]))
        for name, val in (("hero", t.hero), ("ray_d", t.ray_d), ("radiance", t.radiance), ("xyz", t.xyz)):
            assert np.array_equal(bits(list(val)), bits(g[name][i])), (name, i)


def test_traces_match_the_independent_restatement_of_the_glossy_dielectric_scene(gpu):
    """The GPU against tests/golden/traces_s3_restated.npz: GGX, mirror and dispersive dielectric at depth 32, HwssNaive and
    SwssNaive, computed in numpy float32 from the Rust text (tests/golden/restate_path.py) -- no C code produced these numbers."""
    from iris_amd import Camera, Render, Scene
    g = np.load(os.path.join(GOLD, "traces_s3_restated.npz"))
    scene = Scene.glossy_dielectric()
    gpu.upload(scene)
    w, h = int(g["width"]), int(g["height"])
    for integ in (0, 2):
        sel = np.flatnonzero(g["integrator"] == integ)
        r = Render(w, h, 256, scene, integrator=integ, max_depth=int(g["max_depth"]), camera=Camera([float(v) for v in g["cam_pos"]], np.float32(w) / np.float32(h)))
        got = gpu.dbg_trace(r, g["xys"][sel])
        for i, t in zip(sel, got):
            n = int(g["n_hits"][i])
            assert t.n_hits == n and t.draws == int(g["draws"][i]), i
            assert list(t.hit_prim[:n]) == list(g["hit_prim"][i][:n])
            assert np.array_equal(bits(list(t.hit_t[:n])), bits(g["hit_t"][i][:n]))
            for name, val in (("hero", t.hero), ("ray_d", t.ray_d), ("radiance", t.radiance), ("xyz", t.xyz)):
                assert np.array_equal(bits(list(val)), bits(g[name][i])), (name, i)


@pytest.mark.parametrize("name,scene_name", [("traces_s1_slow_restated.npz", "s1"), ("traces_s3_slow_restated.npz", "s3")])
def test_traces_match_the_independent_restatement_of_the_integrators_without_light_sampling(gpu, name, scene_name):
    """The GPU against tests/golden/traces_s{1,3}_slow_restated.npz: HwssSlow and SwssSlow, 256 camera samples each per scene,
    computed in numpy float32 from hwss_slow.rs / swss_slow.rs (tests/golden/restate_path.py --slow)."""
    from iris_amd import Camera, Render, Scene
    g = np.load(os.path.join(GOLD, name))
    scene = Scene.dummy() if scene_name == "s1" else Scene.glossy_dielectric()
    gpu.upload(scene)
    w, h = int(g["width"]), int(g["height"])
    for integ in (1, 3):
        sel = np.flatnonzero(g["integrator"] == integ)
        r = Render(w, h, 256, scene, integrator=integ, max_depth=int(g["max_depth"]), camera=Camera([float(v) for v in g["cam_pos"]], np.float32(w) / np.float32(h)))
        got = gpu.dbg_trace(r, g["xys"][sel])
        for i, t in zip(sel, got):
            n = int(g["n_hits"][i])
            assert t.n_hits == n and t.draws == int(g["draws"][i]), i
            assert list(t.hit_prim[:n]) == list(g["hit_prim"][i][:n])
            assert np.array_equal(bits(list(t.hit_t[:n])), bits(g["hit_t"][i][:n]))
            for field, val in (("hero", t.hero), ("ray_d", t.ray_d), ("radiance", t.radiance), ("xyz", t.xyz)):
                assert np.array_equal(bits(list(val)), bits(g[field][i])), (field, i)


def test_traces_match_the_independent_restatement_of_the_enclosed_scene(gpu, iris_trig):
    """The GPU against tests/golden/traces_enclosed_restated.npz: two lights, the second one around everything and without a material,
    all four integrators (restate_path.py --enclosed): the uniform-sphere branches of the light code, the pick among lights."""
    from iris_amd import Camera, Render, Scene
    g = np.load(os.path.join(GOLD, "traces_enclosed_restated.npz"))
    scene = Scene.enclosed()
    flat = gpu.upload(scene)
    w, h = int(g["width"]), int(g["height"])
    for integ in (0, 1, 2, 3):
        sel = np.flatnonzero(g["integrator"] == integ)
        r = Render(w, h, 256, scene, integrator=integ, max_depth=int(g["max_depth"]), camera=Camera([float(v) for v in g["cam_pos"]], np.float32(w) / np.float32(h)))
        got = gpu.dbg_trace(r, g["xys"][sel])
        for i, t in zip(sel, got):
            n = int(g["n_hits"][i])
            assert t.n_hits == n and t.draws == int(g["draws"][i]), i
            assert list(t.hit_prim[:n]) == list(g["hit_prim"][i][:n])
            assert np.array_equal(bits(list(t.hit_t[:n])), bits(g["hit_t"][i][:n]))
            for field, val in (("hero", t.hero), ("ray_d", t.ray_d), ("radiance", t.radiance), ("xyz", t.xyz)):
                assert np.array_equal(bits(list(val)), bits(g[field][i])), (field, i)
    # and a whole film of it through the production kernels equals the oracle's
    for integ in (0, 3):
        film = gpu.render_film(Render(96, 64, 8, scene, integrator=integ))
        want, st = oracle_film(iris_trig, flat, 96, 64, 8, integrator=integ)
        gs = gpu.stats()
        assert np.array_equal(bits(film), bits(want)) and film.any()
        assert (gs.samples, gs.bounces, gs.rays, gs.sphere_tests) == (st.samples, st.bounces, st.rays, st.sphere_tests)


def oracle_film(O, flat, w, h, spp, **kw):
    osc = O.OracleScene(flat)
    cam = O.camera_new([0, 0, 0], np.float32(w) / np.float32(h))
    film, _, _, st = osc.render(cam, O.make_params(w, h, spp, **kw), nthreads=8)
    return film, st


@pytest.mark.parametrize("scene_name,w,h,spp,integrator,max_depth", [
    ("s1", 64, 64, 8, 0, 15), ("s1", 70, 33, 5, 0, 15), ("s1", 1, 1, 3, 0, 15), ("s1", 48, 48, 6, 1, 15),
    ("s1", 48, 48, 6, 2, 15), ("s1", 48, 48, 6, 3, 15), ("s3", 64, 48, 16, 0, 32), ("furnace", 16, 16, 32, 0, 15),
    ("s2_small", 48, 32, 4, 0, 15), ("s1", 40, 40, 4, 0, 1), ("s1", 40, 40, 4, 0, 0), ("coloured", 64, 48, 16, 0, 15)])
def test_film_bit_exact(gpu, iris_trig, scene_name, w, h, spp, integrator, max_depth):
    """The film get_pixel_color produces (tile.rs:127-166), samples summed in index order, and the exact
    work counters, for ragged sizes, every integrator and a depth-0 / depth-1 loop."""
    from iris_amd import Render
    scene = make(scene_name)
    flat = gpu.upload(scene)
    r = Render(w, h, spp, scene, integrator=integrator, max_depth=max_depth)
    film = gpu.render_film(r)
    want, st = oracle_film(iris_trig, flat, w, h, spp, integrator=integrator, max_depth=max_depth)
    assert film.shape == (h, w, 3)
    assert np.array_equal(bits(film), bits(want))
    gs = gpu.stats()
    if max_depth > 0:
        assert (gs.samples, gs.bounces, gs.rays) == (st.samples, st.bounces, st.rays)
        if len(flat.spheres) <= 64:  # list scans: the oracle's count; with a hierarchy only the tests it makes
            assert gs.sphere_tests == st.sphere_tests
        else:
            assert 0 < gs.sphere_tests < st.sphere_tests


def test_film_matches_committed_golden(gpu):
    from iris_amd import Render, Scene
    g = np.load(os.path.join(GOLD, "film_s1_32x32x4.npz"))
    gpu.upload(Scene.dummy())
    film = gpu.render_film(Render(32, 32, 4, Scene.dummy()))
    assert np.array_equal(bits(film), bits(g["film"]))
    gs = gpu.stats()
    assert [gs.samples, gs.bounces, gs.rays, gs.sphere_tests] == [int(v) for v in g["stats"]]
    g3 = np.load(os.path.join(GOLD, "film_s3_24x24x8.npz"))
    gpu.upload(Scene.glossy_dielectric())
    film3 = gpu.render_film(Render(24, 24, 8, Scene.glossy_dielectric(), max_depth=32))
    assert np.array_equal(bits(film3), bits(g3["film"]))


def test_sample_ranges_and_schedules_do_not_change_the_film(gpu, iris_trig):
    """The result is a function of (pixel, sample index) only: chunking the sample range, the kernel
    variant (pool/queue kernels, fused or as two kernels; register-resident kernel), the pool geometry
    and the grid size leave every bit unchanged; spp_begin offsets select the same
    streams the oracle selects; an empty range is a black film."""
    import iris_amd
    from iris_amd import Render, Scene
    scene = Scene.dummy()
    flat = gpu.upload(scene)
    r = Render(96, 64, 16, scene)  # 96*64*16 results of 12 bytes: more than the 1 MiB budget of the first variant below
    base = gpu.render_film(r)
    want, _ = oracle_film(iris_trig, flat, 96, 64, 16)
    assert np.array_equal(bits(base), bits(want))
    for opts in ({"result_bytes": 1 << 20}, {"multi_kernel": 1}, {"mode": 1, "blocks_per_cu": 1, "slots_per_block": 256},
                 {"mode": 1, "blocks_per_cu": 8, "slots_per_block": 2048}, {"mode": 1}, {"mode": 2}, {"mode": 1, "specialize": 0}, {"mode": 2, "specialize": 0},
                 {"mode": 2, "reg_blocks_per_cu": 1}, {"mode": 2, "reg_blocks_per_cu": 100},
                 {"work_run": 64}, {"work_run": 192}, {"work_run": 1 << 20}, {"reg_blocks_per_cu": 100, "work_run": 1 << 20}):  # items per claim from the launch's counter; 2^20 > the whole render
        with iris_amd.HipWavefront(0, **opts) as g2:
            g2.upload(scene)
            film = g2.render_film(r)
            assert np.array_equal(bits(film), bits(base)), opts
            if "result_bytes" in opts:
                assert g2.stats().chunks > 1
    part = gpu.render_film(r, spp_begin=5, spp_count=4)
    wantp, _ = oracle_film(iris_trig, flat, 96, 64, 16, spp_begin=5, spp_count=4)
    assert np.array_equal(bits(part), bits(wantp))
    assert not gpu.render_film(r, spp_begin=3, spp_count=0).any()
    # sample indices beyond 2^16 (the Sobol tables hold 16 index bits; higher bits still enter through the index shuffle)
    rbig = Render(96, 64, 70_003, scene)
    big = gpu.render_film(rbig, spp_begin=70_000, spp_count=3)
    wantb, _ = oracle_film(iris_trig, flat, 96, 64, 70_003, spp_begin=70_000, spp_count=3)
    assert np.array_equal(bits(big), bits(wantb)) and big.any()


@pytest.mark.parametrize("integrator", [0, 1, 2, 3])
def test_sorting_hits_by_bsdf_kind_changes_no_bit(iris_trig, integrator):
    """General-class scenes: every wave of the register-resident kernel moves hits between its lanes and per-kind stacks of
    path records in global memory so that a shade iteration runs one BSDF's code (sort_kinds, default 1).  Which lane and
    which iteration shade a bounce must change nothing: films and all four work counters equal the unsorted run's
    (sort_kinds = 0) bit for bit and the oracle's, for S3 (four kinds, depth 32: paths change kind at nearly every bounce),
    S3 under an environment map, and a random soup of 150 spheres of every kind under a hierarchy; with one workgroup per
    CU (long-lived waves: deep stacks) and with runs of 64 items; at a size where every stack fills and drains many times."""
    import iris_amd
    from iris_amd import EnvMap, Render, Scene, UpsampleTable
    from test_gpu_fuzz import random_scene
    table = UpsampleTable(os.path.join(GOLD, "srgb_res16.coeff"))
    scenes = {"s3": Scene.glossy_dielectric(), "s3env": Scene.glossy_dielectric(),
              "soup": random_scene(np.random.RandomState(77), table, 150, False)}
    scenes["s3env"].env_map = EnvMap.sky(table, 64, 64)
    for name, scene in scenes.items():
        w, h, spp = (160, 96, 8) if name != "soup" else (96, 64, 4)
        r = Render(w, h, spp, scene, max_depth=32, integrator=integrator)
        films, stats = {}, {}
        for tag, opts in (("sorted", {}), ("unsorted", {"sort_kinds": 0}), ("one workgroup per CU", {"reg_blocks_per_cu": 1}),
                          ("short runs", {"work_run": 64})):
            with iris_amd.HipWavefront(0, **opts) as g:
                flat = g.upload(scene)
                films[tag] = g.render_film(r)
                st = g.stats()
                stats[tag] = (st.samples, st.bounces, st.rays, st.sphere_tests)
                assert st.kernel == 2, (name, tag)
        want, ctr = oracle_film(iris_trig, flat, w, h, spp, max_depth=32, integrator=integrator)
        for tag in films:
            assert np.array_equal(bits(films[tag]), bits(want)), (name, tag)
            assert stats[tag] == stats["sorted"], (name, tag)
        assert stats["sorted"][:3] == (ctr.samples, ctr.bounces, ctr.rays), name
        if name != "soup":  # (under a hierarchy the kernel makes fewer sphere tests than the oracle's list scan)
            assert stats["sorted"][3] == ctr.sphere_tests, name
    with iris_amd.HipWavefront(0) as g:
        from iris_amd.ffi import IrisError
        for bad in (-1, 2):
            with pytest.raises(IrisError):
                g.set_option("sort_kinds", bad)


def test_film_vs_libm_trig_oracle(gpu, libm_trig):
    """Against the oracle run with libm sin/cos (the reference's f32::sin/cos on Linux): the two films
    differ only where a 1-ulp sin/cos difference propagated (last-bit noise in most lit pixels, a rare
    flipped visibility or roulette decision): relative MSE below 1e-4 and under 2 % of the pixels off by
    more than 0.1 %."""
    from iris_amd import Render, Scene
    scene = Scene.dummy()
    flat = gpu.upload(scene)
    film = gpu.render_film(Render(96, 96, 16, scene))
    want, _ = oracle_film(libm_trig, flat, 96, 96, 16)
    rel_mse = float(np.mean((film - want) ** 2) / np.mean(want ** 2))
    off = np.abs(film - want) > 1e-3 * np.maximum(np.abs(want), 1e-2)
    print("libm-vs-spec trig: rel_mse %.3g, pixels off by >0.1%%: %.4f, pixels differing at all: %.3f"
          % (rel_mse, off.any(axis=2).mean(), np.any(film != want, axis=2).mean()))
    assert rel_mse < 1e-4
    assert off.any(axis=2).mean() < 0.02


def gate_stats(film, parts, ref, se_ref):
    """(rel_mse, fraction of |z| > 3, mean z^2, mean min(z^2, 16), median z^2) of a film (the mean of its `parts`) against a reference"""
    n_parts = len(parts)
    rel_mse = float(np.sum((film - ref) ** 2) / np.sum(ref ** 2))
    se_film = parts.std(axis=0, ddof=1) / np.sqrt(n_parts)
    se = np.sqrt(se_film ** 2 + se_ref ** 2)
    lit = se > 0
    assert lit.mean() > 0.3
    z2 = ((film - ref)[lit] / se[lit]) ** 2
    return rel_mse, float(np.mean(z2 > 9.0)), float(np.mean(z2)), float(np.mean(np.minimum(z2, 16.0))), float(np.median(z2))


def image_gate(gpu, O, scene, w, h, spp, max_depth, n_parts=32, ref_parts=16, null_draws=0):
    """north_star's image gate: the GPU film at `spp` against the CPU oracle (libm sin/cos, what the
    reference calls) at 16x spp over DISJOINT sample indices [spp, 17 spp), so the two are independent estimates.
    (1) relative MSE over all three XYZ channels;
    (2) per pixel and channel z = (gpu - ref) / sqrt(se_gpu^2 + se_ref^2), both standard errors estimated from disjoint
        sample-range parts of the respective render (32 GPU parts, 16 oracle parts of `spp` samples each).
    Returns (rel_mse, fraction of |z| > 3, mean z^2, mean min(z^2, 16), median z^2), and with null_draws > 0 the same five numbers
    for that many UNBIASED films: CPU-oracle renders of further disjoint sample ranges [(17 + k) spp, (18 + k) spp), each in the
    same 32 parts, against the same reference -- draws from the distribution the GPU's statistics have if the GPU is right."""
    from iris_amd import Render
    flat = gpu.upload(scene)
    r = Render(w, h, spp, scene, max_depth=max_depth)
    film = gpu.render_film(r).astype(np.float64)
    osc = O.OracleScene(flat)
    cam = O.camera_new([0, 0, 0], np.float32(w) / np.float32(h))
    rparts = np.stack([osc.render(cam, O.make_params(w, h, spp, spp_begin=spp * (1 + k), spp_count=spp, max_depth=max_depth), nthreads=16)[0].astype(np.float64)
                       for k in range(ref_parts)])
    ref = rparts.mean(axis=0)
    se_ref = rparts.std(axis=0, ddof=1) / np.sqrt(ref_parts)
    per = spp // n_parts
    parts = np.stack([gpu.render_film(r, spp_begin=k * per, spp_count=per).astype(np.float64) * n_parts for k in range(n_parts)])
    stats = gate_stats(film, parts, ref, se_ref)
    if not null_draws:
        return stats
    total = spp * (1 + ref_parts + null_draws)  # spp_total of the null renders: each part is scaled by total / per to a film of its own
    null = []
    for d in range(null_draws):
        begin = spp * (1 + ref_parts + d)
        nparts = np.stack([osc.render(cam, O.make_params(w, h, total, spp_begin=begin + k * per, spp_count=per, max_depth=max_depth), nthreads=16)[0].astype(np.float64)
                           * (total / per) for k in range(n_parts)])
        null.append(gate_stats(nparts.mean(axis=0), nparts, ref, se_ref))
    return stats, np.array(null)


@pytest.mark.parametrize("scene_name,w,h,spp,max_depth", [("s1", 48, 48, 1024, 15), ("s3", 40, 40, 2048, 32), ("s3env", 40, 40, 2048, 32)])
def test_converged_image_tolerance(gpu, libm_trig, scene_name, w, h, spp, max_depth):
    """The image gate on the default scene, on the glossy/dielectric scene at depth 32 and on the same under the HDR sky:
    relative MSE < 1e-3 (all three channels), fewer than 1 % of the z values beyond 3 sigma (0.27 % expected) and mean z^2
    in [0.8, 1.25] -- a bias of one standard error would add 1 to mean z^2.  Two documented departures:
    * S3 without the sky is heavy-tailed (rare bright dispersion/caustic paths dominate ~2 % of the pixel-channels, whose
      2048-sample means are far from normal and whose variance estimates miss the tail), so its z statistics do not follow the
      normal-theory values and fixed bounds would have to be set from a measurement.  Instead their bounds come from the null
      distribution itself, with a STATED FALSE-POSITIVE BUDGET: 200 unbiased films (CPU-oracle renders of 200 further disjoint
      sample ranges, same parts, same reference) give 200 draws of each statistic; the GPU film must not exceed the second
      largest of them in the fraction beyond 3 sigma, in the clipped mean z^2 and in the median z^2 (each a 1 % one-sided test
      under the null: 3 % in all by the union bound), and a film biased by half a standard error fails (checked below on the null
      draws themselves: shifting one of them by 0.5 se puts its clipped mean z^2 above every unshifted draw).
    * Under the sky the Owen-scrambled Sobol points beat independent ones (mean z^2 0.72): the lower bound is 0.5."""
    from iris_amd import EnvMap, Scene, UpsampleTable
    scene = {"s1": Scene.dummy, "s3": Scene.glossy_dielectric, "s3env": Scene.glossy_dielectric}[scene_name]()
    if scene_name == "s3env":
        scene.env_map = EnvMap.sky(UpsampleTable(os.path.join(GOLD, "srgb_res16.coeff")), 64, 64)
    if scene_name == "s3":
        (rel_mse, beyond3, z2, z2_clipped, z2_median), null = image_gate(gpu, libm_trig, scene, w, h, spp, max_depth, null_draws=200)
        bound = np.sort(null, axis=0)[-2]  # the second largest of 200 draws: exceeded by an unbiased film with probability ~1 %
        print("image gate s3: rel_mse %.3g, |z|>3: %.4f (null 99%%: %.4f), mean z^2 %.3f, clipped %.3f (null median %.3f, 99%%: %.3f), median z^2 %.3f (null 99%%: %.3f)"
              % (rel_mse, beyond3, bound[1], z2, z2_clipped, float(np.median(null[:, 3])), bound[3], z2_median, bound[4]))
        assert rel_mse < 1e-3
        assert beyond3 <= bound[1] and z2_clipped <= bound[3] and z2_median <= bound[4]
        # power of the gate: half a standard error of bias adds 0.25 to every z^2 on average, which must leave the null's range
        assert float(np.median(null[:, 3])) + 0.25 > bound[3]
        return
    rel_mse, beyond3, z2, z2_clipped, z2_median = image_gate(gpu, libm_trig, scene, w, h, spp, max_depth)
    print("image gate %s: rel_mse %.3g, |z|>3: %.4f, mean z^2 %.3f (clipped %.3f), median z^2 %.3f" % (scene_name, rel_mse, beyond3, z2, z2_clipped, z2_median))
    assert rel_mse < 1e-3
    assert beyond3 < 0.01 and (0.8 if scene_name == "s1" else 0.5) < z2 < 1.25


# ---- BASELINE sizes: size-independent properties -----------------------------------------------------

def test_config_b_full_size_pixels_bit_exact(gpu, iris_trig):
    """BASELINE config B (1920x1080 at 1024 spp) on the GPU; 96 random pixels of the full-size film are
    recomputed by the oracle at full spp and must match bit for bit; work counters are consistent."""
    from iris_amd import Render, Scene
    scene = Scene.dummy()
    flat = gpu.upload(scene)
    w, h, spp = 1920, 1080, 1024
    film = gpu.render_film(Render(w, h, spp, scene))
    gs = gpu.stats()
    assert gs.samples == w * h * spp
    assert gs.rays >= gs.samples + 2 * 0 and gs.sphere_tests >= 3 * gs.rays
    assert not np.isnan(film).any()
    osc = iris_trig.OracleScene(flat)
    ocam = iris_trig.camera_new([0, 0, 0], np.float32(w) / np.float32(h))
    p = iris_trig.make_params(w, h, spp)
    rng = np.random.RandomState(2025)
    for _ in range(96):
        x, y = int(rng.randint(0, w)), int(rng.randint(0, h))
        assert np.array_equal(bits(film[y, x]), bits(osc.pixel_xyz(ocam, p, x, y))), (x, y)
    # the sky is exactly black, the emitter close to its radiance
    assert film[540, 5].tolist() == [0.0, 0.0, 0.0]
    assert abs(film[60, 960, 1] - 3.0) < 0.1


def test_config_c_one_gpu_share_full_size(gpu, iris_trig):
    """BASELINE config C (3840x2160 at 4096 spp over 8 GPUs): the share of GPU 3 -- sample indices [1536, 2048)
    of every pixel, 4.25e9 samples, more than 2^31 results, so the result buffer is filled in two pieces -- on
    this GPU; 48 random pixels are recomputed by the oracle over the same index range; size-independent
    properties: exact sample count, black sky, and the share of another GPU differs (disjoint streams)."""
    from iris_amd import Render, Scene
    scene = Scene.dummy()
    flat = gpu.upload(scene)
    w, h, spp, share = 3840, 2160, 4096, 512
    r = Render(w, h, spp, scene)
    film = gpu.render_film(r, spp_begin=3 * share, spp_count=share)
    gs = gpu.stats()
    assert gs.samples == w * h * share and gs.chunks >= 2
    assert not np.isnan(film).any() and film[1080, 10].tolist() == [0.0, 0.0, 0.0]
    osc = iris_trig.OracleScene(flat)
    ocam = iris_trig.camera_new([0, 0, 0], np.float32(w) / np.float32(h))
    p = iris_trig.make_params(w, h, spp, spp_begin=3 * share, spp_count=share)
    rng = np.random.RandomState(4096)
    for _ in range(48):
        x, y = int(rng.randint(0, w)), int(rng.randint(h // 3, h))
        assert np.array_equal(bits(film[y, x]), bits(osc.pixel_xyz(ocam, p, x, y))), (x, y)
    # a partial film carries share/spp of the light: the emitter's pixel is close to 3.0 / 8
    assert abs(film[120, 1920, 1] - 3.0 / 8) < 0.02


def test_furnace_converges_on_gpu(gpu):
    """Analytic KAT (scene.rs:36-41, README.md:23): Le / (1 - rho) = 0.5."""
    from iris_amd import Render, Scene
    for integrator in range(4):
        scene = Scene.furnace(0.5, 0.25)
        gpu.upload(scene)
        film = gpu.render_film(Render(32, 32, 512, scene, integrator=integrator, max_depth=40))
        assert abs(film[..., 1].mean() - 0.5) < 0.004, integrator


def test_many_spheres_scene_counters(gpu, gpu_list, iris_trig):
    """Config D geometry (10k spheres).  List scans (streamed through the LDS window in chunks): a small film
    is bit-exact against the oracle and every scan visits all 10 000 spheres.  Hierarchy (the default): the
    same film bit for bit, from a small fraction of the sphere tests, by the pool kernels and by the
    register-resident kernel."""
    import iris_amd
    from iris_amd import Render, Scene
    scene = Scene.many_spheres(10_000)
    flat = gpu_list.upload(scene)
    film = gpu_list.render_film(Render(40, 24, 2, scene))
    want, st = oracle_film(iris_trig, flat, 40, 24, 2)
    assert np.array_equal(bits(film), bits(want))
    gs = gpu_list.stats()
    assert (gs.rays, gs.sphere_tests, gs.bounces) == (st.rays, st.sphere_tests, st.bounces)
    gpu.upload(scene)
    film_h = gpu.render_film(Render(40, 24, 2, scene))
    assert np.array_equal(bits(film_h), bits(want))
    gh = gpu.stats()
    assert (gh.rays, gh.bounces) == (st.rays, st.bounces) and gh.sphere_tests < st.sphere_tests // 50
    assert gh.kernel == 3  # the queue kernels (this scene class under a hierarchy)
    for mode, kernel in ((1, 0), (2, 2)):  # the pool kernels and the register-resident kernel (pooled traversal): same tree, same steps per ray
        with iris_amd.HipWavefront(0, mode=mode) as g2:
            g2.upload(scene)
            assert np.array_equal(bits(g2.render_film(Render(40, 24, 2, scene))), bits(want))
            assert g2.stats().kernel == kernel and g2.stats().sphere_tests == gh.sphere_tests


def test_hierarchy_never_culls_what_the_f32_sphere_test_reports(gpu, gpu_list, iris_trig):
    """Sub-pixel spheres far from the ray origin: the f32 discriminant of sphere.rs:30 errs by up to ~1e-6 |o - c|^2, so a
    sphere of radius r reports hits for rays passing up to ~1e-3 |o - c| outside it -- more than r itself here -- and the
    list scan, which is the reference, sees all of them.  The hierarchy's boxes are therefore grown per ray in proportion
    to the distance (box_entry, iris_device.h); films, counters-independent: hierarchy == list scan == oracle, bit for bit,
    for the camera at the origin, far outside the cloud, and inside it."""
    from iris_amd import Camera, Render, Scene
    from iris_amd.scene import ConstantSpectrum, LambertianBsdf, Sphere
    rng = np.random.RandomState(4242)
    scene = Scene()
    scene.add_emissive_material(Sphere((0.0, 30.0, 60.0), 12.0), LambertianBsdf(ConstantSpectrum(0.5)), ConstantSpectrum(40.0))
    scene.add_material(Sphere((0.0, -1001.5, 2.0), 1000.0), LambertianBsdf(ConstantSpectrum(0.8)))
    for _ in range(400):
        d = rng.uniform(10.0, 400.0)
        ang = rng.uniform(-0.7, 0.7)
        up = rng.uniform(-0.02, 0.5)
        r = float(10.0 ** rng.uniform(-4.0, -2.0))
        scene.add_material(Sphere((d * np.sin(ang), max(d * up - 1.4, -1.45), d * np.cos(ang)), r), LambertianBsdf(ConstantSpectrum(float(rng.uniform(0.2, 0.9)))))
    flat_tree = gpu.upload(scene)
    gpu_list.upload(scene)
    assert len(flat_tree.spheres) >= 65
    for pos in ((0.0, 0.0, 0.0), (0.0, 50.0, -900.0), (3.0, 4.0, 150.0)):
        for w, h, spp in ((160, 120, 4), (31, 17, 16)):
            r = Render(w, h, spp, scene, camera=Camera(pos, np.float32(w) / np.float32(h)))
            tree = gpu.render_film(r)
            assert gpu.stats().bvh_depth > 0
            lst = gpu_list.render_film(r)
            assert gpu_list.stats().bvh_depth == 0
            assert np.array_equal(bits(tree), bits(lst)), (pos, w, h)
    w, h, spp = 64, 48, 4
    r = Render(w, h, spp, scene)
    want, _ = oracle_film(iris_trig, flat_tree, w, h, spp)
    assert np.array_equal(bits(gpu.render_film(r)), bits(want))


@pytest.mark.parametrize("specialize", [1, 0])
def test_every_leaf_size_gives_the_list_scan_film(gpu_list, specialize):
    """Option bvh_leaf (spheres per leaf, 1..7; default 2): the tree's shape changes what is tested, never what is found.  Films
    through the pooled traversal (Lambertian/constant class) and through the general class's own traversal (specialize=0) equal
    the list scan's bit for bit; rays and bounces too (sphere_tests counts the tests made and differs by design)."""
    import iris_amd
    from iris_amd import Render
    scene = make("s2_small")
    gpu_list.upload(scene)
    r = Render(96, 64, 6, scene)
    want = gpu_list.render_film(r)
    ws = gpu_list.stats()
    for leaf in (1, 2, 3, 4, 7):
        for mode in ((0, 2) if specialize else (0,)):  # specialised class: the queue kernels (mode 0) and k_regwave's pooled traversal (mode 2)
            with iris_amd.HipWavefront(device=0, bvh_leaf=leaf, specialize=specialize, mode=mode) as ctx:
                ctx.upload(scene)
                film = ctx.render_film(r)
                st = ctx.stats()
                assert st.bvh_depth > 0
                assert np.array_equal(bits(film), bits(want)), (leaf, mode)
                assert (st.samples, st.bounces, st.rays) == (ws.samples, ws.bounces, ws.rays), (leaf, mode)


def test_list_scan_films_of_long_lists(gpu_list, iris_trig):
    """The window-resident (600 spheres) and streamed (2500) list scans inside whole renders."""
    from iris_amd import Render
    for name in ("s2_small", "s2_chunked"):
        scene = make(name)
        flat = gpu_list.upload(scene)
        film = gpu_list.render_film(Render(32, 24, 3, scene))
        want, st = oracle_film(iris_trig, flat, 32, 24, 3)
        assert np.array_equal(bits(film), bits(want))
        gs = gpu_list.stats()
        assert (gs.samples, gs.bounces, gs.rays, gs.sphere_tests) == (st.samples, st.bounces, st.rays, st.sphere_tests)


# ---- error behaviour ---------------------------------------------------------------------------------

def test_error_codes(gpu):
    import iris_amd
    from iris_amd import ConstantSpectrum, LambertianBsdf, Render, Scene, Sphere
    with iris_amd.HipWavefront(0) as g2:
        with pytest.raises(iris_amd.IrisError, match="no scene"):
            g2.render_film(Render(8, 8, 1, Scene.dummy()))
        nolight = Scene()
        nolight.add_material(Sphere((0.0, 0.0, 3.0), 1.0), LambertianBsdf(ConstantSpectrum(0.5)))
        g2.upload(nolight)
        with pytest.raises(iris_amd.IrisError, match="light"):
            g2.render_film(Render(8, 8, 1, nolight))
        assert not g2.render_film(Render(8, 8, 2, nolight, integrator=iris_amd.HWSS_SLOW)).any()
        with pytest.raises(iris_amd.IrisError):
            g2.render_film(Render(8, 8, 0, nolight))
        with pytest.raises(iris_amd.IrisError):
            g2.set_option("slots_per_block", 100)
        bad = Scene.dummy().flatten()
        bad.prim_material[1] = 7
        with pytest.raises(iris_amd.IrisError, match="prim_material"):
            g2.upload(bad)


# ---- the C++ host driver -----------------------------------------------------------------------------

def read_exr_rgb(path):
    """Parser for the uncompressed scanline EXR the driver writes (3 FLOAT channels B, G, R)."""
    import struct
    b = open(path, "rb").read()
    assert struct.unpack_from("<II", b, 0) == (20000630, 2)
    pos, attrs = 8, {}
    while b[pos] != 0:
        e = b.index(b"\0", pos); name = b[pos:e].decode(); pos = e + 1
        e = b.index(b"\0", pos); typ = b[pos:e].decode(); pos = e + 1
        (size,) = struct.unpack_from("<i", b, pos); pos += 4
        attrs[name] = (typ, b[pos:pos + size]); pos += size
    pos += 1
    x0, y0, x1, y1 = struct.unpack("<4i", attrs["dataWindow"][1])
    w, h = x1 - x0 + 1, y1 - y0 + 1
    assert attrs["compression"][1] == b"\0" and attrs["channels"][1][:2] == b"B\0"
    offs = struct.unpack_from("<%dQ" % h, b, pos)
    img = np.zeros((h, w, 3), np.float32)
    for y in range(h):
        yy, size = struct.unpack_from("<ii", b, offs[y])
        assert yy == y and size == 12 * w
        row = np.frombuffer(b, np.float32, 3 * w, offs[y] + 8).reshape(3, w)
        img[y] = row[::-1].T
    return img


def test_cpp_host_driver_film_equals_python_path(gpu, tmp_path):
    """iris_render (the C++ mirror of main.rs) writes the same linear-RGB film as the Python host path,
    bit for bit, as EXR and as PFM, on one context and with the sample range split over two contexts
    (equal up to the f32 re-association of the partial sums)."""
    import subprocess
    from iris_amd import Render, Scene, build, film_to_rgb
    cli = build.build_cli()
    scene = Scene.dummy()
    gpu.upload(scene)
    want = film_to_rgb(gpu.render_film(Render(80, 48, 6, scene)))
    exr, pfm = str(tmp_path / "o.exr"), str(tmp_path / "o.pfm")
    out = subprocess.run([cli, "--width", "80", "--height", "48", "--spp", "6", "--out", exr], check=True, capture_output=True, text=True).stdout
    assert "Starting render, 80x48@6spp..." in out and "m ray/s" in out
    assert np.array_equal(bits(read_exr_rgb(exr)), bits(want))
    subprocess.run([cli, "--width", "80", "--height", "48", "--spp", "6", "--out", pfm], check=True, capture_output=True)
    raw = open(pfm, "rb").read()
    head = b"PF\n80 48\n-1.0\n"
    assert raw.startswith(head)
    img = np.frombuffer(raw, np.float32, 80 * 48 * 3, len(head)).reshape(48, 80, 3)[::-1]
    assert np.array_equal(bits(img), bits(want))
    # two contexts on the same device stand in for two GPUs: sample ranges [0,3) and [3,6)
    subprocess.run([cli, "--width", "80", "--height", "48", "--spp", "6", "--gpus", "2", "--share-device", "--out", pfm], check=True, capture_output=True)
    img2 = np.frombuffer(open(pfm, "rb").read(), np.float32, 80 * 48 * 3, len(head)).reshape(48, 80, 3)[::-1]
    assert np.allclose(img2, want, rtol=1e-5, atol=1e-6)
    # Scene._env_map through the C++ mirror: a uniform white map (no table needed) and the procedural sky
    from iris_amd import EnvMap
    s_env = Scene.dummy()
    s_env.env_map = EnvMap.constant(1.0)
    gpu.upload(s_env)
    want_env = film_to_rgb(gpu.render_film(Render(80, 48, 6, s_env)))
    subprocess.run([cli, "--width", "80", "--height", "48", "--spp", "6", "--env", "white", "--out", pfm], check=True, capture_output=True)
    img3 = np.frombuffer(open(pfm, "rb").read(), np.float32, 80 * 48 * 3, len(head)).reshape(48, 80, 3)[::-1]
    assert np.array_equal(bits(img3), bits(want_env))
    subprocess.run([cli, "--width", "80", "--height", "48", "--spp", "6", "--env", "sky", "--env-size", "64", "--table",
                    os.path.join(GOLD, "srgb_res16.coeff"), "--scene", "glossy", "--max-depth", "32", "--out", pfm], check=True, capture_output=True)
    img4 = np.frombuffer(open(pfm, "rb").read(), np.float32, 80 * 48 * 3, len(head)).reshape(48, 80, 3)[::-1]
    assert np.isfinite(img4).all() and img4[0, 0].min() > 0.1  # the sky is visible in the corner
    # a Radiance .hdr picture as the map: the C++ chain (iris_hdr_read -> iris_envmap_from_latlong -> texels) and the
    # Python one (EnvMap.from_hdr) are the same library calls, so the films agree bit for bit
    from iris_amd import UpsampleTable
    rng = np.random.RandomState(5)
    px = rng.randint(1, 256, (16, 32, 4)).astype(np.uint8)
    px[..., 3] = rng.randint(124, 134, (16, 32))
    px[px[..., 0] == 1, 0] = 2  # (1, 1, 1, n) would be a repeat marker
    hdr = tmp_path / "map.hdr"
    hdr.write_bytes(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y 16 +X 32\n" + px.tobytes())
    table = UpsampleTable(os.path.join(GOLD, "srgb_res16.coeff"))
    s_hdr = Scene.dummy()
    s_hdr.env_map = EnvMap.from_hdr(hdr, table, 24, 24)
    table.close()
    gpu.upload(s_hdr)
    want_hdr = film_to_rgb(gpu.render_film(Render(80, 48, 6, s_hdr)))
    subprocess.run([cli, "--width", "80", "--height", "48", "--spp", "6", "--env", str(hdr), "--env-size", "24", "--table",
                    os.path.join(GOLD, "srgb_res16.coeff"), "--out", pfm], check=True, capture_output=True)
    img5 = np.frombuffer(open(pfm, "rb").read(), np.float32, 80 * 48 * 3, len(head)).reshape(48, 80, 3)[::-1]
    assert np.array_equal(bits(img5), bits(want_hdr))


def test_multi_gpu_entry_points_on_one_gpu(gpu, iris_trig):
    """iris_render_multi and iris_allreduce_films as far as one GPU can exercise them: one context = the plain
    render bit for bit; two contexts sharing the device = the spp split with the host sum (f32 re-association of
    the per-pixel sum, rtol 2e-6 + the film's own rounding); the RCCL path itself with a single rank (loads
    librccl, creates the communicator, runs ncclAllReduce in place: the film must come back unchanged)."""
    import ctypes as C
    import iris_amd
    from iris_amd import Render, Scene
    hip = C.CDLL("libamdhip64.so")  # the HIP runtime libiris_hip.so already loaded: plain device buffers for the test
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipFree.argtypes = [C.c_void_p]
    scene = Scene.dummy()
    r = Render(64, 40, 12, scene)
    gpu.upload(scene)
    want = gpu.render_film(r)
    assert np.array_equal(bits(iris_amd.render_film_multi([gpu], r)), bits(want))
    with iris_amd.HipWavefront(0) as g2, iris_amd.HipWavefront(0) as g3:
        g2.upload(scene)
        g3.upload(scene)
        got = iris_amd.render_film_multi([gpu, g2, g3], r)
        assert np.allclose(got, want, rtol=2e-6, atol=1e-7)
        # sample ranges [0,4) [4,8) [8,12): the same partial films the single-context path renders
        parts = [gpu.render_film(r, spp_begin=b, spp_count=4) for b in (0, 4, 8)]
        assert np.array_equal(bits(got), bits((parts[0] + parts[1]) + parts[2]))
        d = C.c_void_p()
        assert hip.hipMalloc(C.byref(d), want.nbytes) == 0
        try:
            with pytest.raises(iris_amd.IrisError):  # RCCL needs one rank per GPU
                iris_amd.allreduce_films([gpu, g2], [d.value, d.value], 8)
            assert hip.hipMemcpy(d, want.ctypes.data_as(C.c_void_p), want.nbytes, 1) == 0  # host to device
            iris_amd.allreduce_films([gpu], [d.value], want.size)
            back = np.zeros_like(want)
            assert hip.hipMemcpy(back.ctypes.data_as(C.c_void_p), d, want.nbytes, 2) == 0  # device to host
            assert np.array_equal(bits(back), bits(want))
        finally:
            hip.hipFree(d)
            iris_amd.ffi.load().iris_allreduce_shutdown()


def test_bench_two_ranks_rehearsal(tmp_path):
    """bench.py's multi-rank code path (rendezvous, sample split, film all-reduce, max-over-ranks timing, one JSON
    line from rank 0) with two ranks sharing this GPU and gloo standing in for RCCL (which refuses two ranks on
    one device).  The driver's real multi-GPU runs use one GPU per rank and the nccl backend."""
    import json
    import socket
    import subprocess
    import sys
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, IRIS_BENCH_SHARE_DEVICE="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--width", "320", "--height", "180", "--spp", "8"], env=env, capture_output=True, text=True, timeout=280)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1  # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert d["counters_per_step"]["samples"] == 320 * 180 * 8  # rank 0's share: 8 of the 16 spp
    assert "cpu_baseline" not in d and d["roofline"]["frac"] > 0
    assert d["config"]["ranks"] == 2
    c = d["config_c_strong"]  # the strong-scaling leg: 3840x2160 at 4096 spp in all, 2048 per rank here
    assert c["scaling"] == "strong" and c["ranks"] == 2 and c["n_gpus"] == 2 and c["value"] > 0 and "2048 spp per rank" in c["workload"]

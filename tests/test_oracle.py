"""CPU tests that pin the oracle (oracle/iris_oracle.c): the reference's own unit tests restated, the
committed integer known-answers, the Sobol direction-vector fixture, and analytic end-to-end checks."""
import ctypes as C
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")


def fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def f32(*v):
    return np.array(v, np.float32)


def bits(a):
    return np.asarray(a, np.float32).view(np.uint32)


# ---- the reference's own #[cfg(test)] tests, restated -------------------------------------------

def test_ref_vec4_dot_and_sum(oracle):
    """math/vec4.rs:289-300: test_dot (= 20) and test_sum (= 10 in two orders)."""
    L = oracle.lib()
    assert L.orc_vec4_dot(fp(f32(1, 2, 3, 4)), fp(f32(4, 3, 2, 1))) == 20.0
    assert L.orc_vec4_sum(fp(f32(1, 2, 3, 4))) == 10.0
    assert L.orc_vec4_sum(fp(f32(2, 1, 4, 3))) == 10.0


def test_ref_matrix_multiply(oracle):
    """math/matrix.rs:232-250: identity products, identity inverse, translation of a point."""
    L = oracle.lib()
    ident = np.eye(4, dtype=np.float32).ravel()
    out = np.zeros(16, np.float32)
    L.orc_matrix_mul(fp(ident), fp(ident), fp(out))
    assert np.array_equal(out, ident)
    assert L.orc_matrix_inverse(fp(ident), fp(out)) == 1
    assert np.array_equal(out, ident)
    v = f32(1, 2, 3)
    o3 = np.zeros(3, np.float32)
    L.orc_matrix_mul_vec(fp(ident), fp(v), fp(o3))
    assert np.array_equal(o3, v)
    L.orc_matrix_mul_point(fp(ident), fp(v), fp(o3))
    assert np.array_equal(o3, v)
    trans = np.zeros(16, np.float32)
    L.orc_matrix_translation(fp(f32(2, 0, 1)), fp(trans))
    L.orc_matrix_mul_point(fp(trans), fp(f32(0, 0, 0)), fp(o3))
    assert np.array_equal(o3, f32(2, 0, 1))


def test_ref_upsample_formula(oracle):
    """spectrum/upsample.rs:167-187 needs data/srgb.coeff, which the reference mount does not have
    (.MISSING_LARGE_BLOBS), so its four golden values cannot be used.  What can be pinned without the
    table: the sigmoid of upsample.rs:18-24 at x = 0 and its limits, and the explicit-fma arithmetic
    against a float64 evaluation."""
    L = oracle.lib()
    assert L.orc_upsampled_evaluate_single(fp(f32(0, 0, 0)), 500.0) == 0.5
    assert L.orc_upsampled_evaluate_single(fp(f32(0, 0, 1e6)), 500.0) == 1.0
    assert 0.0 <= L.orc_upsampled_evaluate_single(fp(f32(0, 0, -1e6)), 500.0) < 1e-6
    rng = np.random.RandomState(1)
    for _ in range(200):
        c = (rng.standard_normal(3) * [1e-5, 1e-2, 2.0]).astype(np.float32)
        lam = np.float32(rng.uniform(360, 830))
        x = (float(c[0]) * float(lam) + float(c[1])) * float(lam) + float(c[2])
        ref = 0.5 * x / np.sqrt(x * x + 1.0) + 0.5
        assert abs(L.orc_upsampled_evaluate_single(fp(c), lam) - ref) < 2e-6


# ---- committed integer known-answers (tests/golden/make_golden.py) -------------------------------

@pytest.fixture(scope="module")
def kats():
    return json.load(open(os.path.join(GOLD, "integer_kats.json")))


def test_kat_hash_u32(oracle, kats):
    L = oracle.lib()
    for n, seed, want in kats["hash_u32"]:
        assert L.orc_hash_u32(n, seed) == want


def test_kat_golden_ratio_and_wavelengths(oracle, kats):
    L = oracle.lib()
    for (x, y, s, seed, ubits), wl in zip(kats["golden_ratio"], kats["wavelengths"]):
        u = np.float32(L.orc_golden_ratio(x, y, s, seed))
        assert int(bits([u])[0]) == ubits
        out = np.zeros(4, np.float32)
        hero = np.float32(u * np.float32(470.0) + np.float32(360.0))
        L.orc_wavelength_new(hero, fp(out))
        assert [int(b) for b in bits(out)] == wl[1:]
        assert np.all(out >= 360.0) and np.all(out <= 830.0)


def test_kat_offset_origin(oracle, kats):
    L = oracle.lib()
    for p, n, want in kats["offset_origin"]:
        pa = np.array(p, np.uint32).view(np.float32)
        na = np.array(n, np.uint32).view(np.float32)
        out = np.zeros(3, np.float32)
        L.orc_offset_origin(fp(pa), fp(na), fp(out))
        assert [int(b) for b in bits(out)] == want


# ---- sampler ---------------------------------------------------------------------------------------

def test_sobol_direction_vectors_match_scipy(oracle):
    """The un-scrambled Sobol points of the oracle's table equal scipy's (Joe-Kuo) for all 256
    dimensions at 256 indices below 2^16."""
    g = np.load(os.path.join(GOLD, "sobol_plain_256idx_256dim.npz"))
    L = oracle.lib()
    for i, idx in enumerate(g["index"]):
        got = np.array([L.orc_sobol_plain_u32(int(idx), d) >> 16 for d in range(256)], np.uint16)
        assert np.array_equal(got, g["top16"][i]), idx


def test_owen_scramble_is_a_nested_uniform_scramble(oracle):
    """lk_scramble acts on bit-reversed integers: output bit k may depend only on input bits <= k, i.e.
    after reversing, each bit is flipped as a function of the more significant bits only (Owen)."""
    L = oracle.lib()
    rng = np.random.RandomState(7)
    for _ in range(200):
        n, seed = (int(v) for v in rng.randint(0, 2**32, size=2, dtype=np.uint64))
        k = int(rng.randint(1, 32))
        low = (1 << k) - 1
        other = (n & low) | (int(rng.randint(0, 2**32, dtype=np.uint64)) & ~low & 0xFFFFFFFF)
        assert (L.orc_lk_scramble(n, seed) & low) == (L.orc_lk_scramble(other, seed) & low)


@pytest.mark.parametrize("seed", [0, 1, 0xDEADBEEF])
def test_sobol_sets_are_stratified(oracle, seed):
    """Owen scrambling keeps the net structure: for 2^m consecutive indices every dimension has exactly
    one point per 1/2^m stratum, and dims (0,1) of set 0 form a (0,m,2)-net in base 2."""
    m = 8
    n = 1 << m
    pts = np.array([oracle.sobol_sample_4d(i, 0, seed) for i in range(n)])
    assert pts.min() >= 0.0 and pts.max() < 1.0
    for d in range(4):
        assert len(set((pts[:, d] * n).astype(int))) == n
    for a in range(m + 1):
        cells = set(zip((pts[:, 0] * (1 << a)).astype(int), (pts[:, 1] * (1 << (m - a))).astype(int)))
        assert len(cells) == n
    pts5 = np.array([oracle.sobol_sample_4d(i, 5, seed) for i in range(n)])
    for d in range(4):
        assert len(set((pts5[:, d] * n).astype(int))) == n


def test_sampler_buffer_pops_in_reverse_and_switches_to_hash(oracle):
    """sampler.rs:23-50: a refill yields buf[3] first, then buf[2], buf[1], buf[0]; past the Sobol table
    (set 64, docs/SPEC.md) the four hashes of sampler.rs:36-41 are used."""
    L = oracle.lib()
    x, y, s, seed = 17, 5, 3, 123456789
    d = oracle.sampler_draws(x, y, s, seed, 4 * 66)
    scramble = L.orc_hash_u32(x ^ (y << 16), seed)
    for dim in (0, 1, 63):
        buf = oracle.sobol_sample_4d(s, dim, L.orc_hash_u32(dim, scramble))
        assert np.array_equal(d[4 * dim:4 * dim + 4], buf[::-1])
    for dim in (64, 65):
        buf = np.array([L.orc_hash_u32_to_f32((dim + k) ^ ((s << 16) & 0xFFFFFFFF), scramble) for k in range(4)], np.float32)
        assert np.array_equal(d[4 * dim:4 * dim + 4], buf[::-1])
    assert d.min() >= 0.0 and d.max() <= 1.0


# ---- sin / cos ---------------------------------------------------------------------------------------

def test_iris_trig_is_within_one_ulp_of_libm(oracle):
    L = oracle.lib()
    rng = np.random.RandomState(3)
    xs = np.concatenate([rng.uniform(0, 2 * np.pi, 20000), rng.uniform(-np.pi, 3 * np.pi / 2, 5000), [0.0, np.pi / 4, np.pi / 2, np.pi]]).astype(np.float32)
    diff = 0
    for fn in ("orc_sinf", "orc_cosf"):
        oracle.set_trig_mode(oracle.TRIG_LIBM)
        a = np.array([getattr(L, fn)(float(x)) for x in xs], np.float32)
        oracle.set_trig_mode(oracle.TRIG_IRIS)
        b = np.array([getattr(L, fn)(float(x)) for x in xs], np.float32)
        exact = (np.sin if fn == "orc_sinf" else np.cos)(xs.astype(np.float64))
        ulp = np.spacing(np.abs(exact).astype(np.float32)).astype(np.float64)
        assert np.all(np.abs(b.astype(np.float64) - exact) <= 0.5000001 * ulp + 1e-45), fn  # correctly rounded
        assert np.all(np.abs(a.astype(np.float64) - b.astype(np.float64)) <= ulp), fn
        diff += int((a != b).sum())
    oracle.set_trig_mode(oracle.TRIG_LIBM)
    # the specified sin/cos is correctly rounded on this sample; glibc's sinf/cosf (< 1 ulp, not correctly
    # rounded) differs from it at about 1.4 % of the arguments, always by one ulp
    assert diff <= 2 * len(xs) * 3 // 100


# ---- analytic checks -----------------------------------------------------------------------------

def test_cie_y_integral(oracle):
    """color.rs:6: CIE_Y_INTEGRAL = 116.661843131358 is the integral of the Y table
    (data/processing/cie_xyz.py:29-33 integrates the 1 nm table)."""
    y = np.ctypeslib.as_array(oracle.lib().orc_cie_table(1), (471,)).astype(np.float64)
    assert abs(np.trapz(y, dx=1.0) - 116.661843131358) < 2e-3
    assert abs(y.sum() - 116.661843131358) < 2e-3
    x = np.ctypeslib.as_array(oracle.lib().orc_cie_table(0), (471,))
    assert np.float32(x[0]) == np.float32(0.000000122200)


def test_camera_matches_closed_form(oracle):
    """camera.rs:10-21 with fov 90: clip (cx, cy, 0) maps to a ray direction proportional to
    (cx*aspect, cy, 1) for a camera at the origin."""
    for aspect in (1.0, 16.0 / 9.0):
        cam = oracle.camera_new([0, 0, 0], aspect)
        m = np.array(cam.clip_to_world, np.float32)
        out = np.zeros(3, np.float32)
        oracle.lib().orc_matrix_mul_point(fp(m), fp(f32(0.5, -0.25, 0.0)), fp(out))
        d = out / out[2]
        assert np.allclose(d, [0.5 * aspect, -0.25, 1.0], rtol=1e-5)


@pytest.mark.parametrize("integrator", [0, 1, 2, 3])
def test_furnace(oracle, integrator):
    """scene.rs:36-41 / README.md:23: inside a unit sphere of albedo rho and emission Le every pixel
    converges to Le / (1 - rho); here 0.25 / 0.5 = 0.5 in Y (and X, Z within the CMF integrals)."""
    from iris_amd import Scene
    oracle.set_trig_mode(oracle.TRIG_LIBM)
    sc = oracle.OracleScene(Scene.furnace(0.5, 0.25).flatten())
    cam = oracle.camera_new([0, 0, 0], 1.0)
    p = oracle.make_params(12, 12, 256, integrator=integrator, max_depth=40)
    film, _, _, st = sc.render(cam, p, nthreads=4)
    assert abs(film[..., 1].mean() - 0.5) < 0.01
    assert st.samples == 12 * 12 * 256


def test_default_scene_statistics(oracle):
    """Sanity of the default scene (scene.rs:43-66): the light is seen directly at the top of the
    image with radiance 3 (Y ~ 3), the sky is black, the ground is lit."""
    from iris_amd import Scene
    sc = oracle.OracleScene(Scene.dummy().flatten())
    cam = oracle.camera_new([0, 0, 0], 1.0)
    p = oracle.make_params(32, 32, 32)
    film, rgb, _, st = sc.render(cam, p, nthreads=4, want_rgb=True)
    assert abs(film[2, 16, 1] - 3.0) < 0.35  # looking at the emitter
    assert film[16, 1, 1] == 0.0             # sky: no environment light (scene.rs:112-114)
    assert film[30, 16, 1] > 0.01            # lit ground
    assert np.all(rgb >= 0.0) and not np.isnan(film).any()
    assert st.rays > st.samples and st.sphere_tests >= 3 * st.rays


def test_scene_flatten_matches_oracle_dummy(oracle):
    """The Python Scene.dummy() mirror and the oracle's own orc_scene_dummy() give identical films."""
    from iris_amd import Scene
    cam = oracle.camera_new([0, 0, 0], 1.0)
    p = oracle.make_params(16, 16, 4)
    a, _, _, _ = oracle.OracleScene(Scene.dummy().flatten()).render(cam, p)
    b, _, _, _ = oracle.OracleScene.dummy().render(cam, p)
    assert np.array_equal(bits(a), bits(b))


def test_golden_films_reproduce(oracle):
    """The committed oracle films (tests/golden/make_golden.py) are reproduced by the oracle built here."""
    from iris_amd import Scene
    oracle.set_trig_mode(oracle.TRIG_IRIS)
    try:
        cam = oracle.camera_new([0, 0, 0], 1.0)
        g = np.load(os.path.join(GOLD, "film_s1_32x32x4.npz"))
        film, _, _, st = oracle.OracleScene(Scene.dummy().flatten()).render(cam, oracle.make_params(32, 32, 4), nthreads=2)
        assert np.array_equal(bits(film), bits(g["film"]))
        assert [st.samples, st.bounces, st.rays, st.sphere_tests] == [int(v) for v in g["stats"]]
        g3 = np.load(os.path.join(GOLD, "film_s3_24x24x8.npz"))
        film3, _, _, _ = oracle.OracleScene(Scene.glossy_dielectric().flatten()).render(cam, oracle.make_params(24, 24, 8, max_depth=32), nthreads=2)
        assert np.array_equal(bits(film3), bits(g3["film"]))
    finally:
        oracle.set_trig_mode(oracle.TRIG_LIBM)


def test_sample_ranges_add_up(oracle):
    """tile.rs:130-131,145: rendering [0,k) and [k,n) separately and adding equals the full render up
    to f32 re-association (this is what the multi-GPU sample split relies on)."""
    from iris_amd import Scene
    sc = oracle.OracleScene(Scene.dummy().flatten())
    cam = oracle.camera_new([0, 0, 0], 1.0)
    full, _, _, _ = sc.render(cam, oracle.make_params(16, 16, 8), nthreads=2)
    a, _, _, _ = sc.render(cam, oracle.make_params(16, 16, 8, spp_begin=0, spp_count=3), nthreads=2)
    b, _, _, _ = sc.render(cam, oracle.make_params(16, 16, 8, spp_begin=3, spp_count=5), nthreads=2)
    assert np.allclose(a + b, full, rtol=1e-5, atol=1e-7)
    assert not np.array_equal(a, b)


def test_oracle_matches_the_independent_numpy_restatement(oracle):
    """tests/golden/traces_s1_restated.npz was produced by tests/golden/restate_path.py: a second restatement of a whole
    HwssNaive camera sample on the default scene in numpy float32, written from the Rust text and sharing no code with
    oracle/iris_oracle.c (Sobol points and sin/cos re-implemented from docs/SPEC.md).  Hero wavelengths, primary ray, every
    bounce's primitive and t, the draw count, the radiance and the XYZ value agree bit for bit."""
    g = np.load(os.path.join(GOLD, "traces_s1_restated.npz"))
    from iris_amd import Scene
    oracle.set_trig_mode(oracle.TRIG_IRIS)
    try:
        w, h = int(g["width"]), int(g["height"])
        cam = oracle.camera_new([float(v) for v in g["cam_pos"]], np.float32(w) / np.float32(h))
        assert np.array_equal(np.array(cam.clip_to_world, np.float32).view(np.uint32), g["clip_to_world"].reshape(-1).view(np.uint32))
        sc = oracle.OracleScene(Scene.dummy().flatten())
        p = oracle.make_params(w, h, 64)
        n_bounces = 0
        for i, (x, y, s) in enumerate(g["xys"]):
            t = sc.trace(cam, p, int(x), int(y), int(s))
            n = int(g["n_hits"][i])
            assert t.n_hits == n and t.draws == int(g["draws"][i]), (i, x, y, s)
            assert list(t.hit_prim[:n]) == list(g["hit_prim"][i][:n])
            for name, got in (("hero", t.hero), ("ray_d", t.ray_d), ("radiance", t.radiance), ("xyz", t.xyz)):
                assert np.array_equal(np.array(list(got), np.float32).view(np.uint32), g[name][i].view(np.uint32)), (name, i)
            assert np.array_equal(np.array(list(t.hit_t[:n]), np.float32).view(np.uint32), g["hit_t"][i][:n].view(np.uint32)), i
            n_bounces += n
        assert n_bounces > 150 and float(g["radiance"].max()) > 0.5  # not a set of sky pixels (the light seen directly: 3.0 x balance_heuristic_1 = 0.75)
    finally:
        oracle.set_trig_mode(oracle.TRIG_LIBM)



def test_oracle_matches_the_independent_restatement_of_the_glossy_dielectric_scene(oracle):
    """tests/golden/traces_s3_restated.npz (tests/golden/restate_path.py): whole camera samples on the glossy/dielectric scene at
    depth 32 -- GGX microfacet (ggx.rs, microfacet.rs), mirror (specular.rs), dispersive dielectric (fresnel.rs, math/mod.rs:35-72) --
    192 through HwssNaive and 96 through SwssNaive (swss_naive.rs), restated in numpy float32 from the Rust text without any code of
    the oracle's.  Bit for bit: wavelengths, primary ray, primitive and t of every hit, draw count, radiance, XYZ."""
    g = np.load(os.path.join(GOLD, "traces_s3_restated.npz"))
    from iris_amd import Scene
    oracle.set_trig_mode(oracle.TRIG_IRIS)
    try:
        w, h = int(g["width"]), int(g["height"])
        cam = oracle.camera_new([float(v) for v in g["cam_pos"]], np.float32(w) / np.float32(h))
        flat = Scene.glossy_dielectric().flatten()
        assert [np.float32(b[3]) for b in flat.bsdfs if int(b[0]) == 1] == [g["alpha"]]  # the GGX alpha the traces were made with
        sc = oracle.OracleScene(flat)
        kinds = set()
        for i, (x, y, s) in enumerate(g["xys"]):
            p = oracle.make_params(w, h, 256, integrator=int(g["integrator"][i]), max_depth=int(g["max_depth"]))
            t = sc.trace(cam, p, int(x), int(y), int(s))
            n = int(g["n_hits"][i])
            assert t.n_hits == n and t.draws == int(g["draws"][i]), (i, x, y, s)
            assert list(t.hit_prim[:n]) == list(g["hit_prim"][i][:n])
            kinds.update(g["hit_prim"][i][:n].tolist())
            for name, got in (("hero", t.hero), ("ray_d", t.ray_d), ("radiance", t.radiance), ("xyz", t.xyz)):
                assert np.array_equal(np.array(list(got), np.float32).view(np.uint32), g[name][i].view(np.uint32)), (name, i)
            assert np.array_equal(np.array(list(t.hit_t[:n]), np.float32).view(np.uint32), g["hit_t"][i][:n].view(np.uint32)), i
        assert kinds == {0, 1, 2, 3, 4}  # every primitive, so every BSDF kind, was hit
        assert int(np.sum(g["n_hits"])) > 500 and np.any(g["radiance"][:192, 1:] != g["radiance"][:192, :1])  # dispersion: the wavelengths differ
    finally:
        oracle.set_trig_mode(oracle.TRIG_LIBM)


@pytest.mark.parametrize("name,scene_name", [("traces_s1_slow_restated.npz", "s1"), ("traces_s3_slow_restated.npz", "s3")])
def test_oracle_matches_the_independent_restatement_of_the_integrators_without_light_sampling(oracle, name, scene_name):
    """tests/golden/traces_s{1,3}_slow_restated.npz (tests/golden/restate_path.py --slow): HwssSlow (hwss_slow.rs:28-84) and SwssSlow
    (swss_slow.rs:28-82), 256 camera samples each, on the default scene (depth 15) and on the glossy/dielectric scene (depth 32),
    restated in numpy float32 from the Rust text.  With the two naive integrators of the other files this pins all four integrators
    on every BSDF kind the scenes hold.  Bit for bit: wavelengths, primary ray, primitive and t of every hit, draws, radiance, XYZ."""
    g = np.load(os.path.join(GOLD, name))
    from iris_amd import Scene
    oracle.set_trig_mode(oracle.TRIG_IRIS)
    try:
        w, h = int(g["width"]), int(g["height"])
        cam = oracle.camera_new([float(v) for v in g["cam_pos"]], np.float32(w) / np.float32(h))
        flat = (Scene.dummy() if scene_name == "s1" else Scene.glossy_dielectric()).flatten()
        sc = oracle.OracleScene(flat)
        assert set(g["integrator"].tolist()) == {1, 3}
        for i, (x, y, s) in enumerate(g["xys"]):
            p = oracle.make_params(w, h, 256, integrator=int(g["integrator"][i]), max_depth=int(g["max_depth"]))
            t = sc.trace(cam, p, int(x), int(y), int(s))
            n = int(g["n_hits"][i])
            assert t.n_hits == n and t.draws == int(g["draws"][i]), (i, x, y, s)
            assert list(t.hit_prim[:n]) == list(g["hit_prim"][i][:n])
            for field, got in (("hero", t.hero), ("ray_d", t.ray_d), ("radiance", t.radiance), ("xyz", t.xyz)):
                assert np.array_equal(np.array(list(got), np.float32).view(np.uint32), g[field][i].view(np.uint32)), (field, i)
            assert np.array_equal(np.array(list(t.hit_t[:n]), np.float32).view(np.uint32), g["hit_t"][i][:n].view(np.uint32)), i
        lit = np.any(g["radiance"] != 0, axis=1)
        assert lit[:256].sum() > 40 and lit[256:].sum() > 40  # both integrators found the light often enough to test the emission terms
        assert np.all(g["radiance"][256:, 1:] == 0)           # SwssSlow returns the hero wavelength only (swss_slow.rs:81)
    finally:
        oracle.set_trig_mode(oracle.TRIG_LIBM)


def test_oracle_matches_the_independent_restatement_of_the_enclosed_scene(oracle):
    """tests/golden/traces_enclosed_restated.npz (restate_path.py --enclosed): iris_amd.Scene.enclosed -- two lights, one of them a
    sphere of radius 40 around camera and scene, without a material -- through all four integrators, 128 samples each.  Every
    shading point lies inside the second light: Sphere::sample and Sphere::pdf take their uniform-sphere branches (sphere.rs:83-95,
    :141-153), the light pick has two entries (scene.rs:168-172), paths end on a light without a material (hwss_naive.rs:46-49)."""
    g = np.load(os.path.join(GOLD, "traces_enclosed_restated.npz"))
    from iris_amd import Scene
    oracle.set_trig_mode(oracle.TRIG_IRIS)
    try:
        w, h = int(g["width"]), int(g["height"])
        cam = oracle.camera_new([float(v) for v in g["cam_pos"]], np.float32(w) / np.float32(h))
        flat = Scene.enclosed().flatten()
        assert [np.float32(b[3]) for b in flat.bsdfs if int(b[0]) == 1] == [g["alpha"]]
        sc = oracle.OracleScene(flat)
        assert sorted(set(g["integrator"].tolist())) == [0, 1, 2, 3]
        for i, (x, y, s) in enumerate(g["xys"]):
            p = oracle.make_params(w, h, 256, integrator=int(g["integrator"][i]), max_depth=int(g["max_depth"]))
            t = sc.trace(cam, p, int(x), int(y), int(s))
            n = int(g["n_hits"][i])
            assert t.n_hits == n and t.draws == int(g["draws"][i]), (i, x, y, s)
            assert list(t.hit_prim[:n]) == list(g["hit_prim"][i][:n])
            for field, got in (("hero", t.hero), ("ray_d", t.ray_d), ("radiance", t.radiance), ("xyz", t.xyz)):
                assert np.array_equal(np.array(list(got), np.float32).view(np.uint32), g[field][i].view(np.uint32)), (field, i)
            assert np.array_equal(np.array(list(t.hit_t[:n]), np.float32).view(np.uint32), g["hit_t"][i][:n].view(np.uint32)), i
        ends_on_big_light = sum(int(g["hit_prim"][i][int(g["n_hits"][i]) - 1] == 4) for i in range(len(g["xys"])) if g["n_hits"][i])
        assert ends_on_big_light > 300 and int(np.sum(g["n_hits"])) > 1000
    finally:
        oracle.set_trig_mode(oracle.TRIG_LIBM)

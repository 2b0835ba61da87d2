"""Row f1 of SURVEY.md section 8: HDR environment-map lighting with CDF importance sampling.

The reference only stubs this (Scene._env_map scene.rs:27 is never filled, background_emission returns 0
scene.rs:112-114, the commented miss branch scene.rs:351-353, the clamp note upsample.rs:53-54), so there is
NO reference behaviour to pin: layout, look-up, sampling and the integrator rules are this build's definition
(docs/SPEC.md section 6) -- "parity unpinned" against the reference by construction.  What is tested:
  * CPU: the product's host-side tables equal the oracle's bit for bit; the octahedral mapping round-trips and
    its Jacobian integrates to 4 pi; the sampler's density is the one env_eval reports; the analytic
    white-furnace KAT on the oracle.
  * GPU: traces, films and work counters bit-exact against the oracle for every integrator and kernel
    variant; the furnace KAT on the GPU; env-only scenes (no sphere light).
"""
import ctypes as C
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TABLE = os.path.join(GOLD, "srgb_res16.coeff")


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.fixture(scope="module")
def table():
    import iris_amd
    t = iris_amd.UpsampleTable(TABLE)
    yield t
    t.close()


@pytest.fixture
def iris_trig(oracle):
    oracle.set_trig_mode(oracle.TRIG_IRIS)
    yield oracle
    oracle.set_trig_mode(oracle.TRIG_LIBM)


def env_scene(name, table):
    from iris_amd import EnvMap, Scene
    if name == "s3env":       # BASELINE config E: glossy/dielectric scene, sphere light + HDR sky with a sun
        return Scene.glossy_dielectric_env(table, 64)
    if name == "s1env":       # the default scene under the sky (config A/B as BASELINE.json words them)
        s = Scene.dummy()
        s.env_map = EnvMap.sky(table, 48, 32)
        return s
    if name == "envonly":     # no sphere light at all: the map is the only entry of the light list
        s = Scene()
        from iris_amd import ConstantSpectrum, LambertianBsdf, MicrofacetBsdf, Sphere
        s.add_material(Sphere((0.0, -0.2, 3.0), 1.0), MicrofacetBsdf(ConstantSpectrum(0.9), 0.3, 0.3))
        s.add_material(Sphere((0.0, -101.5, 2.0), 100.0), LambertianBsdf(ConstantSpectrum(0.8)))
        s.env_map = EnvMap.sky(table, 32, 32)
        return s
    if name == "s2env":       # long sphere list (hierarchy) under the sky
        s = Scene.many_spheres(600)
        s.env_map = EnvMap.sky(table, 32, 32)
        return s
    if name == "furnace":
        return Scene.env_furnace(0.5, 1.0)
    raise KeyError(name)


# ---------------------------------------------------------------------------------------------- CPU

def test_host_tables_equal_the_oracle(oracle, table):
    import iris_amd
    rng = np.random.RandomState(5)
    for w, h in ((64, 64), (17, 9), (1, 1), (2, 5)):
        tex = rng.rand(h, w, 4).astype(np.float32)
        imp = rng.rand(h, w).astype(np.float32)
        imp[rng.rand(h, w) < 0.3] = 0.0          # unsampled texels
        if h > 2:
            imp[1, :] = 0.0                      # an all-zero row: uniform conditional, never chosen
        for importance in (imp, None):
            env = iris_amd.EnvMap(tex, importance)
            s = iris_amd.Scene.dummy()
            s.env_map = env
            osc = oracle.OracleScene(s.flatten())
            got, want = env.tables(), osc.env_tables(w, h)
            for a, b in zip(got, want):
                assert np.array_equal(bits(a), bits(b))
            func, cond, marg, integral = got
            assert np.all(np.diff(cond, axis=1) >= 0) and np.all(cond[:, 0] == 0) and np.all(cond[:, -1] == 1)
            assert np.all(np.diff(marg) >= 0) and marg[0] == 0 and marg[-1] == 1
    # all-zero map: nothing to sample, density 0 everywhere
    env = iris_amd.EnvMap(np.zeros((4, 4, 4), np.float32))
    assert env.tables()[3] == 0.0


def test_octahedral_mapping(oracle):
    import iris_amd
    L, O = iris_amd.ffi.load(), oracle.lib()
    fp = C.POINTER(C.c_float)
    rng = np.random.RandomState(11)
    d = rng.normal(size=(2000, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
    d[:6] = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]], np.float32)
    uv, ouv, back, q = np.zeros(2, np.float32), np.zeros(2, np.float32), np.zeros(3, np.float32), np.zeros(3, np.float32)
    jac = C.c_float()
    for v in d:
        v = np.ascontiguousarray(v)
        L.iris_dir_to_oct(v.ctypes.data_as(fp), uv.ctypes.data_as(fp))
        O.orc_dir_to_uv(v.ctypes.data_as(fp), ouv.ctypes.data_as(fp), C.byref(jac))
        assert np.array_equal(bits(uv), bits(ouv))            # product host helper == oracle
        assert np.all(np.abs(uv) <= 1.0)
        L.iris_oct_to_dir(C.c_float(uv[0]), C.c_float(uv[1]), back.ctypes.data_as(fp))
        assert np.allclose(back, v, atol=3e-6)                # round trip
        O.orc_oct_point(C.c_float(uv[0]), C.c_float(uv[1]), q.ctypes.data_as(fp))
        assert abs(np.abs(q).sum() - 1.0) < 1e-6              # on the octahedron
        assert abs(jac.value - np.linalg.norm(q) ** 3) < 1e-5
    # d(omega) = du dv / |q|^3: the texel weights of a fine map add up to the sphere's 4 pi
    n = 512
    u = (np.arange(n) + 0.5) / n * 2 - 1
    uu, vv = np.meshgrid(u, u)
    y = 1 - np.abs(uu) - np.abs(vv)
    x = np.where(y >= 0, uu, (1 - np.abs(vv)) * np.sign(uu))
    z = np.where(y >= 0, vv, (1 - np.abs(uu)) * np.sign(vv))
    total = ((x * x + y * y + z * z) ** -1.5).sum() * (2.0 / n) ** 2
    assert abs(total - 4 * np.pi) < 2e-3


def test_sampling_density_matches_env_eval(oracle, table):
    """Directions drawn by env_sample land in texels in proportion to func, and E[1/pdf] over them is the
    solid angle of the sampled support (4 pi when no texel has zero importance)."""
    import iris_amd
    env = iris_amd.EnvMap.sky(table, 16, 16)
    s = iris_amd.Scene.dummy()
    s.env_map = env
    osc = oracle.OracleScene(s.flatten())
    func, cond, marg, integral = env.tables()
    rng = np.random.RandomState(3)
    n = 40000
    counts = np.zeros((16, 16))
    inv = 0.0
    L = iris_amd.ffi.load()
    fp = C.POINTER(C.c_float)
    uv = np.zeros(2, np.float32)
    for _ in range(n):
        d = osc.env_sample(np.float32(rng.rand()), np.float32(rng.rand()))
        assert abs(np.linalg.norm(d) - 1.0) < 1e-5
        le, pdf = osc.env_eval(d, [450.0, 567.5, 685.0, 802.5])
        assert pdf > 0 and np.all(le >= 0)
        inv += 1.0 / pdf
        L.iris_dir_to_oct(d.ctypes.data_as(fp), uv.ctypes.data_as(fp))
        c, r = min(int((uv[0] * 0.5 + 0.5) * 16), 15), min(int((uv[1] * 0.5 + 0.5) * 16), 15)
        counts[r, c] += 1
    assert abs(inv / n - 4 * np.pi) < 0.15
    expect = func / func.sum() * n
    big = expect > 200
    assert big.sum() >= 4
    assert np.all(np.abs(counts[big] - expect[big]) < 6 * np.sqrt(expect[big]))


def test_white_furnace_on_the_oracle(oracle):
    """A convex Lambertian body of albedo rho in a uniform environment of radiance L reflects rho*L: pixels on
    the sphere converge to rho*L, the others are exactly L's XYZ -- for all four integrators."""
    from iris_amd import Scene
    osc = oracle.OracleScene(Scene.env_furnace(0.5, 1.0).flatten())
    cam = oracle.camera_new([0, 0, 0], np.float32(1.0))
    for integrator in range(4):
        film, _, _, st = osc.render(cam, oracle.make_params(24, 24, 256, integrator=integrator), nthreads=8)
        assert abs(film[0, 0, 1] - 1.0) < 0.03 and abs(film[0, 0, 1] - film[23, 23, 1]) < 0.03
        centre = film[10:14, 10:14, 1].mean()
        assert abs(centre - 0.5) < 0.02, (integrator, centre)


def test_scene_without_lights_needs_an_env_map_only_for_nee(oracle):
    """pick_one_light on an empty list indexes out of bounds in the reference (scene.rs:168-172); with a map
    the list has one entry.  Oracle-side: an env-only scene renders and is not black."""
    from iris_amd import EnvMap, LambertianBsdf, ConstantSpectrum, Scene, Sphere
    s = Scene()
    s.add_material(Sphere((0.0, 0.0, 3.0), 1.0), LambertianBsdf(ConstantSpectrum(0.7)))
    s.env_map = EnvMap.constant(2.0)
    osc = oracle.OracleScene(s.flatten())
    cam = oracle.camera_new([0, 0, 0], np.float32(1.0))
    film, _, _, _ = osc.render(cam, oracle.make_params(8, 8, 64), nthreads=2)
    assert abs(film[4, 4, 1] - 1.4) < 0.1 and abs(film[0, 0, 1] - 2.0) < 0.1


def golden_env_scene():
    from iris_amd import EnvMap, Scene
    g = np.load(os.path.join(GOLD, "film_s3env_24x24x8.npz"))
    s = Scene.glossy_dielectric()
    s.env_map = EnvMap(g["texels"], g["importance"])
    return s, g


def test_committed_env_golden_on_the_oracle_and_host_tables(iris_trig):
    """tests/golden/film_s3env_24x24x8.npz (made by tests/golden/make_golden.py): the oracle reproduces the
    committed film and the product's host-side table builder reproduces the committed tables."""
    s, g = golden_env_scene()
    func, cond, marg, integral = s.env_map.tables()
    assert np.array_equal(bits(func), bits(g["func"])) and np.array_equal(bits(cond), bits(g["cond_cdf"]))
    assert np.array_equal(bits(marg), bits(g["marg_cdf"])) and bits(integral) == bits(g["integral"])
    osc = iris_trig.OracleScene(s.flatten())
    cam = iris_trig.camera_new([0, 0, 0], np.float32(1.0))
    film, _, _, st = osc.render(cam, iris_trig.make_params(24, 24, 8, max_depth=32), nthreads=4)
    assert np.array_equal(bits(film), bits(g["film"]))
    assert [st.samples, st.bounces, st.rays, st.sphere_tests] == [int(v) for v in g["stats"]]


# ---------------------------------------------------------------------------------------------- GPU

@pytest.mark.gpu
def test_env_film_matches_committed_golden(gpu):
    from iris_amd import Render
    s, g = golden_env_scene()
    gpu.upload(s)
    film = gpu.render_film(Render(24, 24, 8, s, max_depth=32))
    assert np.array_equal(bits(film), bits(g["film"]))
    gs = gpu.stats()
    assert [gs.samples, gs.bounces, gs.rays, gs.sphere_tests] == [int(v) for v in g["stats"]]


@pytest.mark.gpu
@pytest.mark.parametrize("scene_name,integrator,max_depth", [
    ("s3env", 0, 32), ("s3env", 1, 32), ("s3env", 2, 32), ("s3env", 3, 32), ("s1env", 0, 15), ("envonly", 0, 15),
    ("envonly", 2, 15), ("s2env", 0, 15), ("furnace", 0, 15)])
def test_env_path_trace_bit_exact(gpu, iris_trig, table, scene_name, integrator, max_depth):
    from iris_amd import Render
    scene = env_scene(scene_name, table)
    flat = gpu.upload(scene)
    osc = iris_trig.OracleScene(flat)
    w, h, spp = 128, 96, 32
    r = Render(w, h, spp, scene, integrator=integrator, max_depth=max_depth)
    rng = np.random.RandomState(7 + integrator)
    xys = rng.randint(0, [w, h, spp], size=(1200, 3)).astype(np.uint32)
    gt = gpu.dbg_trace(r, xys)
    ocam = iris_trig.camera_new([0, 0, 0], np.float32(w) / np.float32(h))
    p = iris_trig.make_params(w, h, spp, integrator=integrator, max_depth=max_depth)
    lit = 0
    for (x, y, s), a in zip(xys, gt):
        b = osc.trace(ocam, p, int(x), int(y), int(s))
        assert a.n_hits == b.n_hits and a.draws == b.draws
        n = min(a.n_hits, 64)
        assert list(a.hit_prim[:n]) == list(b.hit_prim[:n])
        assert np.array_equal(bits(list(a.hit_t[:n])), bits(list(b.hit_t[:n])))
        assert np.array_equal(bits(list(a.radiance)), bits(list(b.radiance))), (x, y, s)
        assert np.array_equal(bits(list(a.xyz)), bits(list(b.xyz)))
        lit += a.radiance[0] > 0
    assert lit > 600


@pytest.mark.gpu
@pytest.mark.parametrize("scene_name,w,h,spp,integrator,max_depth", [
    ("s3env", 64, 48, 16, 0, 32), ("s3env", 48, 32, 8, 1, 32), ("s3env", 48, 32, 8, 2, 32), ("s3env", 48, 32, 8, 3, 32),
    ("s1env", 70, 33, 5, 0, 15), ("envonly", 48, 48, 8, 0, 15), ("s2env", 48, 32, 4, 0, 15), ("furnace", 16, 16, 16, 0, 15)])
def test_env_film_bit_exact_all_kernel_variants(gpu, iris_trig, table, scene_name, w, h, spp, integrator, max_depth):
    """Whole films and the work counters, by the default kernel and by the pool/queue kernels (fused and as two
    kernels): the miss branch lives in three different places there."""
    import iris_amd
    from iris_amd import Render
    scene = env_scene(scene_name, table)
    flat = gpu.upload(scene)
    osc = iris_trig.OracleScene(flat)
    cam = iris_trig.camera_new([0, 0, 0], np.float32(w) / np.float32(h))
    want, _, _, st = osc.render(cam, iris_trig.make_params(w, h, spp, integrator=integrator, max_depth=max_depth), nthreads=8)
    r = Render(w, h, spp, scene, integrator=integrator, max_depth=max_depth)
    film = gpu.render_film(r)
    assert np.array_equal(bits(film), bits(want))
    gs = gpu.stats()
    assert (gs.samples, gs.bounces, gs.rays) == (st.samples, st.bounces, st.rays)
    assert (film[..., 1] > 0.0).mean() > 0.98  # an env-lit scene has (almost) no black pixel
    for opts in ({"mode": 1}, {"multi_kernel": 1}):
        with iris_amd.HipWavefront(0, **opts) as g2:
            g2.upload(scene)
            assert np.array_equal(bits(g2.render_film(r)), bits(want)), opts
            s2 = g2.stats()
            assert (s2.samples, s2.bounces, s2.rays) == (st.samples, st.bounces, st.rays)


@pytest.mark.gpu
def test_env_white_furnace_on_gpu(gpu):
    from iris_amd import Render, Scene
    for integrator in range(4):
        for rho, expect in ((0.5, 0.5), (1.0, 1.0)):
            scene = Scene.env_furnace(rho, 1.0)
            gpu.upload(scene)
            film = gpu.render_film(Render(32, 32, 512, scene, integrator=integrator, max_depth=40))
            assert abs(film[0, 0, 1] - 1.0) < 0.02
            assert abs(film[14:18, 14:18, 1].mean() - expect) < 0.02, (integrator, rho)


@pytest.mark.gpu
def test_env_importance_sampling_beats_bsdf_sampling_alone(gpu, table):
    """The point of the CDF search: under a sky whose sun is 400x brighter than the rest, next-event
    estimation through the map's CDFs (HwssNaive) has far less variance than hitting the sun by BSDF sampling
    (HwssSlow) at equal spp; both converge to the same image."""
    from iris_amd import EnvMap, LambertianBsdf, ConstantSpectrum, Render, Scene, Sphere
    s = Scene()
    s.add_material(Sphere((0.0, -101.0, 3.0), 100.0), LambertianBsdf(ConstantSpectrum(0.6)))
    s.env_map = EnvMap.sky(table, 128, 128)
    gpu.upload(s)
    films = {}
    for integ in (0, 1):
        a = gpu.render_film(Render(32, 16, 256, s, integrator=integ), spp_begin=0, spp_count=128)
        b = gpu.render_film(Render(32, 16, 256, s, integrator=integ), spp_begin=128, spp_count=128)
        ground = (slice(10, 16), slice(None), 1)
        films[integ] = (a[ground] + b[ground], np.mean((2 * a[ground] - 2 * b[ground]) ** 2) / 2)
    mean_nee, var_nee = films[0][0].mean(), films[0][1]
    mean_slow, var_slow = films[1][0].mean(), films[1][1]
    assert abs(mean_nee - mean_slow) < 0.08 * mean_nee
    assert var_nee * 8 < var_slow, (var_nee, var_slow)


@pytest.mark.gpu
def test_env_upload_errors(gpu):
    import iris_amd
    from iris_amd import EnvMap, Scene
    with iris_amd.HipWavefront(0) as g2:
        s = Scene()  # no primitives at all stays an error, env or not
        s.env_map = EnvMap.constant(1.0)
        with pytest.raises(iris_amd.IrisError):
            g2.upload(s)
        # a scene whose env map is removed again renders like the plain scene
        s1 = Scene.dummy()
        s1.env_map = EnvMap.constant(1.0)
        g2.upload(s1)
        lit = g2.render_film(iris_amd.Render(16, 16, 4, s1))
        g2.upload(Scene.dummy())
        plain = g2.render_film(iris_amd.Render(16, 16, 4, Scene.dummy()))
        gpu.upload(Scene.dummy())
        assert np.array_equal(bits(plain), bits(gpu.render_film(iris_amd.Render(16, 16, 4, Scene.dummy()))))
        assert lit[0, 0, 1] > 0.9 and plain[0, 0, 1] == 0.0  # the corner pixel sees the sky


# ---------------------------------------------------------------------------------------------- host helpers (CPU)

def test_envmap_host_helpers(table):
    """iris_envmap_from_latlong, iris_envmap_texels_from_rgb, iris_envmap_tables argument checks (no GPU)."""
    import iris_amd
    from iris_amd import EnvMap
    L = iris_amd.ffi.load()
    fp = C.POINTER(C.c_float)
    # a constant lat-long image stays constant
    src = np.full((16, 32, 3), 0.25, np.float32)
    oct_ = EnvMap.latlong_to_oct(src, 24, 24)
    assert oct_.shape == (24, 24, 3) and np.allclose(oct_, 0.25, atol=1e-6)
    # upper half bright, lower half dark: octahedral texels follow the sign of the texel direction's y
    src = np.zeros((64, 128, 3), np.float32)
    src[:32] = 1.0
    oct_ = EnvMap.latlong_to_oct(src, 32, 32)
    dirs = EnvMap.texel_directions(32, 32)
    up = dirs[..., 1] > 0.1
    down = dirs[..., 1] < -0.1
    assert np.all(oct_[up][:, 0] > 0.9) and np.all(oct_[down][:, 0] < 0.1)
    # one bright lat-long texel towards +x lands in the octahedral texel whose direction is +x
    src = np.zeros((64, 128, 3), np.float32)
    src[32, 96] = 100.0  # theta = pi/2 (+0.5 texel), phi = 3pi/2 -> atan2(dx, -dz) = 3pi/2: dx = -1?  checked below by direction
    oct_ = EnvMap.latlong_to_oct(src, 64, 64)
    r, c = np.unravel_index(np.argmax(oct_[..., 0]), oct_.shape[:2])
    d = EnvMap.texel_directions(64, 64)[r, c]
    phi = 2 * np.pi * (96 + 0.5) / 128
    expect = np.array([np.sin(phi), 0.0, -np.cos(phi)])  # phi = atan2(dx, -dz), theta ~ pi/2
    assert np.dot(d, expect) > 0.99
    # texels_from_rgb == get_spectrum_hdr per texel, with the clamp as a parameter
    rgb = np.array([[0.2, 0.5, 0.1], [3.0, 80.0, 1.0], [0.0, 0.0, 0.0]], np.float32)
    for max_hdr in (50.0, float("inf")):
        tex = np.zeros((3, 4), np.float32)
        imp = np.zeros(3, np.float32)
        assert L.iris_envmap_texels_from_rgb(table.handle, rgb.ctypes.data_as(fp), 3, C.c_float(max_hdr), tex.ctypes.data_as(fp), imp.ctypes.data_as(fp)) == 0
        for i in range(3):
            ref = table.get_spectrum_hdr(rgb[i])
            assert np.array_equal(bits(tex[i, :3]), bits(ref.coefficients))
            assert tex[i, 3] == (min(ref.hdr_coefficient, 50.0) if max_hdr == 50.0 else max(rgb[i].max(), 1.0))
        assert np.allclose(imp, rgb @ np.array([0.2126, 0.7152, 0.0722], np.float32), rtol=1e-6)
    # argument checks
    assert L.iris_envmap_tables(None, None, None, None, None) != 0
    assert L.iris_envmap_from_latlong(None, 1, 1, 1, 1, None) != 0
    assert L.iris_envmap_texels_from_rgb(None, None, 0, C.c_float(1.0), None, None) != 0


# ---- Radiance RGBE pictures (.hdr) ---------------------------------------------------------------------

def _rgbe_header(w, h, magic=b"#?RADIANCE"):
    return magic + b"\n# written by tests/test_envmap.py\nFORMAT=32-bit_rle_rgbe\nEXPOSURE=1.0\n\n" + ("-Y %d +X %d\n" % (h, w)).encode()


def _rle_channel(row):
    """Adaptive run-length form of one channel of one scanline (runs of >= 3 equal bytes, literals otherwise)."""
    out, i, n = bytearray(), 0, len(row)
    while i < n:
        run = 1
        while i + run < n and run < 127 and row[i + run] == row[i]:
            run += 1
        if run >= 3:
            out += bytes([128 + run, row[i]])
            i += run
            continue
        j = i
        while j < n and j - i < 128:
            r = 1
            while j + r < n and r < 3 and row[j + r] == row[j]:
                r += 1
            if r >= 3:
                break
            j += 1
        out += bytes([j - i]) + bytes(row[i:j])
        i = j
    return bytes(out)


def _rgbe_expected(px):
    """Radiance's colr_color: (m + 0.5) * 2^(e - 136), zero when e == 0."""
    e = px[..., 3].astype(np.int32)
    scale = np.where(e == 0, 0.0, np.ldexp(1.0, e - 136)).astype(np.float32)
    return ((px[..., :3].astype(np.float32) + np.float32(0.5)) * scale[..., None]).astype(np.float32) * (e != 0)[..., None]


def test_hdr_reader_all_scanline_forms(tmp_path):
    """iris_hdr_read against the published RGBE definition: plain pixels, the adaptive run-length form (runs and
    literals, every channel), the old repeat-marker form, a zero exponent, both magic lines; damaged and
    unsupported files are refused, not misread."""
    from iris_amd import EnvMap
    from iris_amd.ffi import IrisError
    rng = np.random.RandomState(11)
    w, h = 37, 9
    px = rng.randint(0, 256, (h, w, 4)).astype(np.uint8)
    px[..., 3] = rng.randint(118, 140, (h, w))
    px[2, 5:30] = px[2, 5]          # a long run in every channel
    px[3, :, 3] = 128               # constant exponent: the common case in real pictures
    px[4, 7] = (9, 9, 9, 0)         # zero exponent = black
    px[px[..., 0] == 1, 0] = 2      # (1, 1, 1, n) is the old form's repeat marker: keep it out of the plain data
    want = _rgbe_expected(px)

    flat = tmp_path / "flat.hdr"
    flat.write_bytes(_rgbe_header(w, h) + px.tobytes())
    assert np.array_equal(bits(EnvMap.read_hdr(flat)), bits(want))

    rle = tmp_path / "rle.hdr"
    body = b"".join(bytes([2, 2, w >> 8, w & 255]) + b"".join(_rle_channel(px[y, :, ch]) for ch in range(4)) for y in range(h))
    rle.write_bytes(_rgbe_header(w, h, b"#?RGBE") + body)
    assert len(body) < px.nbytes + 4 * h * 5
    assert np.array_equal(bits(EnvMap.read_hdr(rle)), bits(want))

    old = tmp_path / "old.hdr"   # pixels 1..301 equal: pixel 1, then 300 repeats as two markers, 44 << 0 and 1 << 8
    w2 = 320
    row = np.zeros((w2, 4), np.uint8)
    row[:] = (90, 80, 70, 127)
    row[0] = (10, 20, 30, 130)
    row[1:302] = (50, 60, 70, 129)
    data = bytes(row[0]) + bytes(row[1]) + bytes([1, 1, 1, 44]) + bytes([1, 1, 1, 1]) + b"".join(bytes(row[k]) for k in range(302, w2))
    old.write_bytes(_rgbe_header(w2, 1) + data)
    assert np.array_equal(bits(EnvMap.read_hdr(old)), bits(_rgbe_expected(row[None])))

    for name, blob in (("trunc.hdr", _rgbe_header(w, h) + px.tobytes()[:-5]),
                       ("xyze.hdr", _rgbe_header(w, h).replace(b"rle_rgbe", b"rle_xyze") + px.tobytes()),
                       ("flip.hdr", _rgbe_header(w, h).replace(b"-Y", b"+Y") + px.tobytes()),
                       ("magic.hdr", b"P6\n" + px.tobytes()),
                       ("overrun.hdr", _rgbe_header(w, h) + bytes([2, 2, 0, w]) + bytes([128 + 127, 7]) * 4)):
        p = tmp_path / name
        p.write_bytes(blob)
        with pytest.raises(IrisError):
            EnvMap.read_hdr(p)
    with pytest.raises(IrisError):
        EnvMap.read_hdr(tmp_path / "missing.hdr")


def test_hdr_picture_becomes_an_environment_map(tmp_path, table):
    """An equirectangular .hdr picture through the whole host chain (read -> octahedral resample -> per-texel
    get_spectrum_hdr -> sampling tables): a bright patch near +x in the picture is where the sampling CDF puts
    its mass, and the C++ driver accepts the same file."""
    import subprocess
    from iris_amd import EnvMap
    w, h = 64, 32
    px = np.zeros((h, w, 4), np.uint8)
    px[:] = (128, 128, 128, 126)                      # grey, 0.125
    px[h // 2 - 2:h // 2 + 2, 3 * w // 4 - 2:3 * w // 4 + 2] = (200, 180, 120, 136)   # a patch on the equator, three quarters of the way round
    path = tmp_path / "patch.hdr"
    path.write_bytes(_rgbe_header(w, h) + px.tobytes())
    rgb = EnvMap.read_hdr(path)
    assert rgb.shape == (h, w, 3) and rgb.max() > 150 and abs(rgb[0, 0, 0] - 128.5 / 1024) < 1e-7
    env = EnvMap.from_hdr(path, table, 48, 48)
    func = env.tables()[0]
    dirs = EnvMap.texel_directions(48, 48)
    peak = dirs[np.unravel_index(np.argmax(func), func.shape)]
    # iris_envmap_from_latlong: column 0 = direction -z, turning towards +x; the patch sits on the equator 3/4 of the way round
    phi = 2 * np.pi * 0.75
    want = np.array([np.sin(phi), 0.0, -np.cos(phi)])
    assert np.dot(peak, want) > 0.95, (peak, want)
    assert func.max() > 500 * np.median(func)
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "iris_amd", "host", "iris_render")
    if os.path.exists(exe):  # the CLI stops at the missing GPU, after it has read the picture and built the map
        r = subprocess.run([exe, "--env", str(path), "--env-size", "16", "--table", TABLE, "--dump-scene"], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        assert "env " in r.stdout or "env_map" in r.stdout, r.stdout[-300:]
        r = subprocess.run([exe, "--env", str(tmp_path / "nope.hdr"), "--dump-scene"], capture_output=True, text=True)
        assert r.returncode != 0 and "RGBE" in r.stderr

"""Randomised scenes: the HIP path against the oracle, bit for bit, on scenes nobody designed -- random sphere
soups with every BSDF kind, constant and sigmoid spectra, several lights (some without a material, some
overlapping the camera), optional environment map, every integrator, list scans and the hierarchy.  NaN-producing
geometry is part of the deal (the reference's tangent frame is NaN where the normal is parallel to y,
sphere.rs:38): films are compared as bit patterns, NaNs included."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def random_scene(rng, table, n_spheres, with_env):
    from iris_amd import (ConstantSpectrum, EnvMap, FresnelBsdf, LambertianBsdf, MicrofacetBsdf, NullBsdf, Scene, SpecularBsdf,
                          Sphere)

    def spectrum(hdr=False):
        if rng.rand() < 0.5:
            return ConstantSpectrum(float(np.float32(rng.uniform(0.05, 4.0 if hdr else 0.95))))
        rgb = rng.uniform(0.02, 1.0, 3) * (rng.uniform(1.0, 30.0) if hdr else 1.0)
        return table.get_spectrum_hdr(rgb) if hdr else table.get_spectrum(rgb)

    def bsdf():
        k = rng.randint(0, 9)
        if k < 4:
            return LambertianBsdf(spectrum())
        if k < 6:
            return MicrofacetBsdf(spectrum(), float(rng.uniform(0.05, 0.9)), float(rng.uniform(0.05, 0.9)))
        if k == 6:
            return SpecularBsdf(spectrum())
        if k == 7:
            return FresnelBsdf(spectrum(), spectrum(), float(rng.uniform(1.1, 2.2)), float(rng.uniform(0.0, 0.02)))
        return NullBsdf()

    s = Scene()
    n_lights = 0
    for i in range(n_spheres):
        pos = (float(np.float32(rng.uniform(-4, 4))), float(np.float32(rng.uniform(-2, 4))), float(np.float32(rng.uniform(1.5, 12))))
        geom = Sphere(pos, float(np.float32(rng.uniform(0.2, 1.5))))
        r = rng.rand()
        if r < 0.15 or (i == n_spheres - 1 and n_lights == 0 and not with_env):
            if rng.rand() < 0.3:
                s.add_light(geom, spectrum(hdr=True))           # a light without a material: paths end on it
            else:
                s.add_emissive_material(geom, bsdf(), spectrum(hdr=True))
            n_lights += 1
        else:
            s.add_material(geom, bsdf())
    if rng.rand() < 0.5:
        s.add_material(Sphere((0.0, -101.5, 2.0), 100.0), LambertianBsdf(spectrum()))  # ground
    if rng.rand() < 0.2:
        s.add_material(Sphere((0.0, 0.0, 0.0), 30.0), bsdf())  # the camera inside a big sphere
    if with_env:
        w, h = int(rng.choice([1, 3, 16, 33, 40])), int(rng.choice([1, 2, 16, 31, 64]))
        rgb = rng.uniform(0.0, 1.0, (h, w, 3)).astype(np.float32) * np.float32(rng.uniform(0.2, 3.0))
        rgb[rng.rand(h, w) < 0.2] = 0.0
        if rng.rand() < 0.5:
            rgb[rng.randint(0, h), rng.randint(0, w)] += 200.0
        s.env_map = EnvMap.from_rgb(rgb, table)
    return s


def fuzz_case(seed, table):
    """Scene and render parameters of fuzz case `seed` (the random stream is left where the render seed and the camera come next)."""
    rng = np.random.RandomState(1000 + seed)
    n_spheres = int(rng.choice([1, 2, 5, 12, 40, 90]))  # >= 65: hierarchy
    if os.environ.get("IRIS_FUZZ_BIG") == "1":  # a soak of the hierarchy alone: deeper trees
        n_spheres = int(rng.choice([90, 300, 1000]))
    with_env = bool(seed % 2)
    scene = random_scene(rng, table, n_spheres, with_env)
    integrator = int(rng.randint(0, 4))
    max_depth = int(rng.choice([1, 4, 15, 40]))
    w, h, spp = int(rng.randint(8, 50)), int(rng.randint(8, 40)), int(rng.randint(1, 10))
    scale = int(os.environ.get("IRIS_FUZZ_SCALE", "1"))  # IRIS_FUZZ_SCALE=6: 36 times the samples per scene, for rare events
    return rng, scene, n_spheres, with_env, integrator, max_depth, w * scale, h * scale, spp


@pytest.mark.parametrize("seed", range(int(os.environ.get("IRIS_FUZZ_SEEDS", "48"))))  # IRIS_FUZZ_SEEDS=1000 for a soak
def test_random_scene_film_bit_exact(gpu, oracle, seed):
    import iris_amd
    from iris_amd import Render, UpsampleTable
    table = UpsampleTable(os.path.join(GOLD, "srgb_res16.coeff"))
    rng, scene, n_spheres, with_env, integrator, max_depth, w, h, spp = fuzz_case(seed, table)
    oracle.set_trig_mode(oracle.TRIG_IRIS)
    try:
        flat = gpu.upload(scene)
        render_seed = int(rng.randint(0, 2**31))
        # half of the scenes are seen from somewhere else than the origin (camera.rs:10-21 takes a position)
        pos = [float(np.float32(v)) for v in rng.uniform(-3.0, 3.0, 3)] if rng.rand() < 0.5 else [0.0, 0.0, 0.0]
        r = Render(w, h, spp, scene, integrator=integrator, max_depth=max_depth, seed=render_seed,
                   camera=iris_amd.Camera(tuple(pos), np.float32(w) / np.float32(h)))
        film = gpu.render_film(r)
        gs = gpu.stats()
        osc = oracle.OracleScene(flat)
        cam = oracle.camera_new(pos, np.float32(w) / np.float32(h))
        want, _, _, st = osc.render(cam, oracle.make_params(w, h, spp, integrator=integrator, max_depth=max_depth, seed=r.seed), nthreads=8)
        same = bits(film) == bits(want)
        both_nan = np.isnan(film) & np.isnan(want)  # NaN payloads may differ; being NaN must not
        assert np.all(same | both_nan), (seed, n_spheres, integrator, max_depth, with_env, np.argwhere(~(same | both_nan))[:4])
        assert (gs.samples, gs.bounces, gs.rays) == (st.samples, st.bounces, st.rays)
        if seed % 4 == 0:  # the pool/queue kernels on the same scene
            with iris_amd.HipWavefront(0, mode=1) as g2:
                g2.upload(scene)
                f2 = g2.render_film(r)
                assert np.all((bits(f2) == bits(want)) | (np.isnan(f2) & np.isnan(want)))
    finally:
        oracle.set_trig_mode(oracle.TRIG_LIBM)
        table.close()


def test_sky_mask_never_changes_a_bit(gpu):
    """The sky mask (k_sky_mask) only skips samples that are exactly zero: films and work counters with and
    without it are identical, on scenes chosen to sit on its margins -- spheres a fraction of a pixel wide,
    spheres grazing pixel footprints at full resolution, the camera near, inside and far from spheres, cameras away
    from the origin -- and it does find the sky of the default scene."""
    import iris_amd
    from iris_amd import Camera, ConstantSpectrum, LambertianBsdf, Render, Scene, Sphere
    rng = np.random.RandomState(77)
    cases = []
    s1 = Scene.dummy()
    cases.append((s1, Render(1920, 1080, 1, s1)))
    cases.append((s1, Render(333, 77, 3, s1, camera=Camera((0.3, 0.4, -1.0), np.float32(333) / np.float32(77)))))
    cases.append((s1, Render(64, 64, 4, s1, camera=Camera((0.0, -0.2, 3.0), np.float32(1.0)))))      # inside the middle sphere
    cases.append((s1, Render(64, 64, 4, s1, camera=Camera((0.0, -0.2, 1.9995), np.float32(1.0)))))   # half a millimetre off its surface
    for k in range(6):
        s = Scene()
        s.add_emissive_material(Sphere((0.0, 50.0, 10.0), 5.0), LambertianBsdf(ConstantSpectrum(0.5)), ConstantSpectrum(40.0))
        for _ in range(40):
            z = float(rng.uniform(2.0, 400.0))
            r = float(np.float32(z * rng.uniform(0.0002, 0.004)))  # 0.1 .. 2 pixels wide at 512 pixels across a 90 degree view
            s.add_material(Sphere((float(np.float32(rng.uniform(-0.9, 0.9) * z)), float(np.float32(rng.uniform(-0.9, 0.9) * z)), float(np.float32(z))), r),
                           LambertianBsdf(ConstantSpectrum(0.7)))
        cases.append((s, Render(512, 512, 2, s)))
    with iris_amd.HipWavefront(0, sky_mask=0) as plain:
        for i, (scene, r) in enumerate(cases):
            gpu.upload(scene)
            plain.upload(scene)
            a, sa = gpu.render_film(r), gpu.stats()
            b, sb = plain.render_film(r), plain.stats()
            assert np.array_equal(bits(a), bits(b)), i
            assert (sa.samples, sa.bounces, sa.rays, sa.sphere_tests) == (sb.samples, sb.bounces, sb.rays, sb.sphere_tests), i
            assert sb.sky_samples == 0
            if i == 0:
                assert 0.45 < sa.sky_samples / sa.samples < 0.60
            if i == 2:
                assert sa.sky_samples == 0  # the camera is inside a sphere: nothing can be proved
            if i >= 4:
                assert sa.sky_samples / sa.samples > 0.9  # nearly everything is sky; the specks are still found
                assert (a[..., 1] > 0).sum() > 20

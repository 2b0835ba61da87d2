"""The debug-invariants build (libiris_hip_debug.so, -DIRIS_DEBUG_INVARIANTS=1): where the reference's DEBUG build would have
panicked -- non-finite Vec3 (math/vec3.rs:24-31), negative or NaN SpectralSample / PdfSet (spectrum/sample.rs:64-73,
math/pdf.rs:49-58), a cone-sampled light point off its sphere by more than 2 % (shape/sphere.rs:131-132) -- counted at the same
check points by the kernels and by the oracle.  The fuzz suite accepts "NaN on both sides"; this says how often and of which kind,
and that the designed scenes (S1, S3, config D's geometry) never get there."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.fixture(scope="module")
def dbg():
    import iris_amd
    ctx = iris_amd.HipWavefront(device=0, debug=True)
    yield ctx
    ctx.close()


def oracle_counts(oracle, reset=True):
    out = (C.c_uint64 * 4)()
    oracle.lib().orc_invariant_counts(out, 1 if reset else 0)
    return [int(v) for v in out]


def test_only_the_debug_build_counts(gpu, dbg):
    import iris_amd
    from iris_amd import Render, Scene
    gpu.upload(Scene.dummy())
    gpu.render_film(Render(16, 16, 1, Scene.dummy()))
    with pytest.raises(iris_amd.IrisError):
        gpu.invariant_counts()


@pytest.mark.parametrize("name,w,h,spp,max_depth", [("s1", 160, 120, 16, 15), ("s3", 160, 120, 16, 32), ("s2", 96, 64, 4, 15), ("furnace", 32, 32, 16, 40)])
def test_designed_scenes_never_trip_an_invariant(dbg, oracle, name, w, h, spp, max_depth):
    """The reference's default scene, the glossy/dielectric scene at depth 32, config D's 10 000 spheres (through the hierarchy) and
    the furnace, all four integrators: zero on the GPU, zero in the oracle, and the film is the oracle's bit for bit (the debug
    build computes what the shipped build computes)."""
    from iris_amd import Render, Scene
    scene = {"s1": Scene.dummy, "s3": Scene.glossy_dielectric, "s2": lambda: Scene.many_spheres(10_000), "furnace": lambda: Scene.furnace(0.5, 0.25)}[name]()
    flat = dbg.upload(scene)
    oracle.set_trig_mode(oracle.TRIG_IRIS)
    try:
        osc = oracle.OracleScene(flat)
        cam = oracle.camera_new([0, 0, 0], np.float32(w) / np.float32(h))
        for integrator in range(4):
            film = dbg.render_film(Render(w, h, spp, scene, integrator=integrator, max_depth=max_depth))
            assert dbg.invariant_counts() == [0, 0, 0, 0], (name, integrator)
            if name != "s2" or integrator == 0:  # (the list-scan oracle takes a while on 10 000 spheres)
                oracle_counts(oracle)
                want = osc.render(cam, oracle.make_params(w, h, spp, integrator=integrator, max_depth=max_depth), nthreads=16)[0]
                assert oracle_counts(oracle) == [0, 0, 0, 0]
                assert np.array_equal(bits(film), bits(want))
    finally:
        oracle.set_trig_mode(oracle.TRIG_LIBM)


def test_fuzz_scenes_trip_the_same_invariants_on_both_sides(dbg, oracle):
    """The 48 random scenes of tests/test_gpu_fuzz.py: per scene the four counts of the kernels equal the oracle's; over the suite
    some are non-zero (the reference's tangent is NaN where the normal is parallel to y, sphere.rs:38; Null materials and cameras
    inside spheres do the rest) -- which is what a caller of the debug build learns: where the reference would have stopped."""
    import iris_amd
    from iris_amd import Render, UpsampleTable
    from test_gpu_fuzz import fuzz_case
    table = UpsampleTable(os.path.join(GOLD, "srgb_res16.coeff"))
    total = np.zeros(4, np.int64)
    tripped = 0
    oracle.set_trig_mode(oracle.TRIG_IRIS)
    try:
        for seed in range(48):
            rng, scene, n_spheres, with_env, integrator, max_depth, w, h, spp = fuzz_case(seed, table)
            flat = dbg.upload(scene)
            render_seed = int(rng.randint(0, 2**31))
            pos = [float(np.float32(v)) for v in rng.uniform(-3.0, 3.0, 3)] if rng.rand() < 0.5 else [0.0, 0.0, 0.0]
            r = Render(w, h, spp, scene, integrator=integrator, max_depth=max_depth, seed=render_seed,
                       camera=iris_amd.Camera(tuple(pos), np.float32(w) / np.float32(h)))
            film = dbg.render_film(r)
            got = dbg.invariant_counts()
            osc = oracle.OracleScene(flat)
            cam = oracle.camera_new(pos, np.float32(w) / np.float32(h))
            oracle_counts(oracle)
            want = osc.render(cam, oracle.make_params(w, h, spp, integrator=integrator, max_depth=max_depth, seed=r.seed), nthreads=8)[0]
            assert got == oracle_counts(oracle), (seed, n_spheres, integrator, max_depth, with_env)
            assert np.all((bits(film) == bits(want)) | (np.isnan(film) & np.isnan(want))), seed
            total += got
            tripped += any(got)
    finally:
        oracle.set_trig_mode(oracle.TRIG_LIBM)
        table.close()
    print("invariant counts over the 48 fuzz scenes [Vec3, SpectralSample, PdfSet, light point]: %s; scenes with any: %d" % (total.tolist(), tripped))
    assert total.sum() > 0

"""iris_render_device_noalloc (include/iris_hip.h): the entry point for callers that cannot allocate.

Called through the C ABI on device films (tests/hip_mem.py: plain hipMalloc) and compared BIT FOR BIT with
iris_render_device, the entry point it is a variant of: films and work counters, on two calls in a row, the
second with a smaller input (the static workspace keeps the first call's contents); every scene class the
register-resident kernel serves; the size limit is an error, not a fault.  Collected last (test_zz_*) so that a
failure here cannot hide another test behind pytest -x."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def both(gpu, render, **kw):
    from hip_mem import DeviceFilm
    a, b = DeviceFilm(render.height, render.width, 0x7f), DeviceFilm(render.height, render.width, 0x55)
    gpu.render_film_device(render, a.data_ptr(), **kw)
    sa = gpu.stats()
    gpu.render_film_device_noalloc(render, b.data_ptr(), **kw)
    sb = gpu.stats()
    fa, fb = a.numpy(), b.numpy()
    a.free()
    b.free()
    assert np.array_equal(bits(fa), bits(fb))
    assert (sa.samples, sa.bounces, sa.rays, sa.sphere_tests, sa.sky_samples) == (sb.samples, sb.bounces, sb.rays, sb.sphere_tests, sb.sky_samples)
    assert sb.chunks == sa.chunks and sb.chunks <= 1
    return fb


def test_noalloc_equals_render_device_two_calls_in_a_row(gpu):
    from iris_amd import Render, Scene
    scene = Scene.dummy()
    gpu.upload(scene)
    big = both(gpu, Render(160, 96, 24, scene))      # 160*96*24 results = 4.4 MB of the workspace
    assert big.any() and not np.isnan(big).any()
    small = both(gpu, Render(64, 40, 5, scene))      # smaller film, fewer samples: stale results lie behind and between
    assert small.any()
    part = both(gpu, Render(64, 40, 16, scene), spp_begin=5, spp_count=3)  # a sample range of a larger render
    assert part.any()
    again = both(gpu, Render(160, 96, 24, scene))
    assert np.array_equal(bits(again), bits(big))
    empty = both(gpu, Render(64, 40, 16, scene), spp_begin=2, spp_count=0)
    assert not empty.any()


@pytest.mark.parametrize("scene_name", ["s3", "s3env", "s2_small"])
def test_noalloc_other_scene_classes(scene_name):
    """General class (second hit queue in the workspace), environment-map class, hierarchy (deep stack levels in the
    workspace when the tree is deeper than the LDS levels)."""
    import os
    import iris_amd
    from iris_amd import EnvMap, Render, Scene, UpsampleTable
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    scene = {"s3": Scene.glossy_dielectric, "s3env": Scene.glossy_dielectric, "s2_small": lambda: Scene.many_spheres(600)}[scene_name]()
    if scene_name == "s3env":
        scene.env_map = EnvMap.sky(UpsampleTable(os.path.join(gold, "srgb_res16.coeff")), 64, 64)
    with iris_amd.HipWavefront(0) as g:
        g.upload(scene)
        first = both(g, Render(96, 64, 8, scene, max_depth=32))
        second = both(g, Render(48, 32, 3, scene, max_depth=32))
        assert first.any() and second.any()


@pytest.mark.parametrize("scene_name", ["s1", "s3", "s3env", "s2_small"])
def test_noalloc_leaves_free_device_memory_unchanged(scene_name):
    """The claim itself: on a context that has never rendered (so nothing a render allocates on demand exists yet), the
    call changes the GPU's free memory by nothing, per scene class; iris_render_device on the same fresh state does
    allocate (which shows that the probe sees allocations of this size).  One earlier no-alloc call in ANOTHER context
    loads the kernels first: what the HIP runtime sets up at a kernel's first launch is not the library's."""
    import os
    import iris_amd
    from hip_mem import DeviceFilm, free_bytes
    from iris_amd import EnvMap, Render, Scene, UpsampleTable
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    scene = {"s1": Scene.dummy, "s3": Scene.glossy_dielectric, "s3env": Scene.glossy_dielectric, "s2_small": lambda: Scene.many_spheres(600)}[scene_name]()
    if scene_name == "s3env":
        scene.env_map = EnvMap.sky(UpsampleTable(os.path.join(gold, "srgb_res16.coeff")), 64, 64)
    render = Render(256, 160, 16, scene, max_depth=32)  # 7.9 MB of results: far above the probe's granularity
    film = DeviceFilm(render.height, render.width)
    with iris_amd.HipWavefront(0) as warm:
        warm.upload(scene)
        warm.render_film_device_noalloc(render, film.data_ptr())
    with iris_amd.HipWavefront(0) as g:
        g.upload(scene)
        before = free_bytes()
        g.render_film_device_noalloc(render, film.data_ptr())
        after_first = free_bytes()
        g.render_film_device_noalloc(Render(128, 96, 4, scene, max_depth=32), film.data_ptr())
        after_second = free_bytes()
        assert after_first == before and after_second == before, (before, after_first, after_second)
        g.render_film_device(render, film.data_ptr())
        assert free_bytes() < before  # the allocating entry point does show on this probe
    film.free()


def test_noalloc_limits_are_errors():
    import iris_amd
    from hip_mem import DeviceFilm
    from iris_amd import Render, Scene
    from iris_amd.ffi import IrisError, load
    cap = load().iris_noalloc_workspace_bytes()
    assert cap == 64 << 20
    scene = Scene.dummy()
    with iris_amd.HipWavefront(0) as g:
        g.upload(scene)
        w, h = 512, 512
        film, ref = DeviceFilm(h, w), DeviceFilm(h, w)
        spp_fit = (cap - (1 << 20)) // (12 * w * h)            # leaves room for the fixed parts
        g.render_film_device_noalloc(Render(w, h, spp_fit, scene), film.data_ptr())
        assert g.stats().samples == w * h * spp_fit
        with pytest.raises(IrisError, match="workspace"):
            g.render_film_device_noalloc(Render(w, h, cap // (12 * w * h) + 1, scene), film.data_ptr())
        # the caller's way out: disjoint sample ranges that each fit, summed by the caller
        total = 2 * spp_fit
        r = Render(w, h, total, scene)
        g.render_film_device(r, ref.data_ptr())
        acc = np.zeros((h, w, 3), np.float32)
        for b in range(0, total, spp_fit):
            g.render_film_device_noalloc(r, film.data_ptr(), spp_begin=b, spp_count=spp_fit)
            acc += film.numpy()
        np.testing.assert_allclose(acc, ref.numpy(), rtol=2e-6, atol=1e-7)  # f32 re-association of the per-pixel sum
    with iris_amd.HipWavefront(0, bvh_min_prims=0) as g:  # a long list scanned sphere by sphere runs the pool kernels: not served
        sc = Scene.many_spheres(200)
        g.upload(sc)
        with pytest.raises(IrisError, match="register-resident"):
            g.render_film_device_noalloc(Render(32, 32, 2, sc), film.data_ptr())

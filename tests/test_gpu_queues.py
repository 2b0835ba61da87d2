"""The queue kernels (k_wfq_shade + k_wfq_trace, mode 3: shading and traversal as separate kernels over path-state and ray queues in
HBM; the default for scenes with a hierarchy whose BSDFs are Lambertian over constant spectra) against the register-resident
kernel (mode 2) and the oracle: films bit for bit, and the four work counters -- sphere_tests included, because per ray both
kernels take the same steps through the same tree."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture
def iris_trig(oracle):
    """The oracle with the specified sin/cos (docs/SPEC.md) -- the mode the kernels match bit for bit."""
    oracle.set_trig_mode(oracle.TRIG_IRIS)
    yield oracle
    oracle.set_trig_mode(oracle.TRIG_LIBM)


def bits(a):
    return np.asarray(a, dtype=np.float32).view(np.uint32)


def stats_tuple(st):
    return (st.samples, st.bounces, st.rays, st.sphere_tests)


def both(scene, w, h, spp, integrator=0, max_depth=15, **opts):
    import iris_amd
    from iris_amd import Render
    r = Render(w, h, spp, scene, integrator=integrator, max_depth=max_depth)
    with iris_amd.HipWavefront(0, mode=2) as ref:
        ref.upload(scene)
        want = ref.render_film(r)
        ws = ref.stats()
        assert ws.kernel == 2
    with iris_amd.HipWavefront(0, mode=3, **opts) as g:
        g.upload(scene)
        film = g.render_film(r)
        st = g.stats()
        assert st.kernel == 3
    assert np.array_equal(bits(film), bits(want)), (w, h, spp, integrator, opts)
    assert stats_tuple(st) == stats_tuple(ws), (w, h, spp, integrator, opts)
    return film


@pytest.mark.parametrize("integrator", [0, 1, 2, 3])
def test_queue_kernels_equal_register_kernel_every_integrator(integrator):
    from iris_amd import Scene
    both(Scene.many_spheres(600), 48, 32, 4, integrator=integrator)


def test_queue_kernels_slots_pipelines_batches_do_not_change_a_bit(iris_trig):
    """Slot counts from one tile to more than the render needs, one to four pipelines, batches of one pass: every path is a function
    of its item, whichever slot, pipeline and pass traces it; ragged film sizes; depth 1 and 0."""
    from iris_amd import Render, Scene
    from test_gpu_parity import oracle_film
    scene = Scene.many_spheres(600)
    film = both(scene, 96, 64, 6)
    import iris_amd
    with iris_amd.HipWavefront(0) as g:  # mode 0 chooses the queue kernels for this scene class
        flat = g.upload(scene)
        auto = g.render_film(Render(96, 64, 6, scene))
        assert g.stats().kernel == 3 and np.array_equal(bits(auto), bits(film))
    want, st = oracle_film(iris_trig, flat, 96, 64, 6)
    assert np.array_equal(bits(film), bits(want))
    for opts in ({"wfq_slots": 256, "wfq_pipes": 1}, {"wfq_slots": 1024, "wfq_pipes": 4, "wfq_batch": 3}, {"wfq_pipes": 1, "wfq_batch": 1},
                 {"wfq_item_run": 256}, {"wfq_item_run": 1 << 20}, {"wfq_trace_per_cu": 1, "wfq_shade_per_cu": 1}, {"wfq_trace_per_cu": 8, "wfq_shade_per_cu": 16}):
        both(scene, 96, 64, 6, **opts)
    both(scene, 70, 33, 5, wfq_pipes=1, wfq_batch=1)
    both(scene, 64, 64, 4, max_depth=1)
    both(scene, 1, 1, 3)
    with iris_amd.HipWavefront(0, mode=3) as g:
        g.upload(scene)
        assert not g.render_film(Render(32, 32, 4, scene, max_depth=0)).any()
        assert not g.render_film(Render(32, 32, 4, scene), spp_begin=2, spp_count=0).any()
        part = g.render_film(Render(64, 48, 16, scene), spp_begin=5, spp_count=4)
    wantp, _ = oracle_film(iris_trig, flat, 64, 48, 16, spp_begin=5, spp_count=4)
    assert np.array_equal(bits(part), bits(wantp))


def test_queue_kernels_config_d_geometry_slot_reuse():
    """10 000 spheres: a film larger than the slots, so that every slot carries many paths one after the other."""
    from iris_amd import Scene
    big = Scene.many_spheres(10_000)
    both(big, 40, 24, 2)
    both(big, 640, 360, 8)
    both(big, 640, 360, 8, wfq_slots=65536, wfq_pipes=2)


def test_queue_kernels_refuse_other_scene_classes():
    import iris_amd
    from iris_amd import Render, Scene
    with iris_amd.HipWavefront(0, mode=3) as g:
        g.upload(Scene.dummy())
        with pytest.raises(iris_amd.IrisError, match="mode 3"):
            g.render_film(Render(8, 8, 1, Scene.dummy()))

"""Multi-process CPU tests (gloo, world_size 2 and 3) of the sample-index split + sum all-reduce that the
multi-GPU path uses (iris_amd/distributed.py)."""
import os
import socket
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_sample_range_partitions_every_index_once():
    from iris_amd.distributed import sample_range
    for spp in (0, 1, 5, 64, 100, 1024, 4096):
        for world in (1, 2, 3, 4, 8):
            ranges = [sample_range(spp, r, world) for r in range(world)]
            covered = [i for b, n in ranges for i in range(b, b + n)]
            assert covered == list(range(spp))
            counts = [n for _, n in ranges]
            assert max(counts) - min(counts) <= 1
    with pytest.raises(ValueError):
        sample_range(8, 2, 2)


def test_native_split_equals_the_python_split():
    """iris_render_multi (one process, n GPUs) and bench.py / render_distributed (one process per GPU) must hand out the
    same blocks of sample indices: iris_sample_range (the C ABI's split, host code -- no GPU needed) against
    distributed.sample_range for ragged and degenerate counts."""
    import ctypes as C
    from iris_amd import ffi
    from iris_amd.distributed import sample_range
    L = ffi.load()
    b, n = C.c_uint32(), C.c_uint32()
    for spp in (0, 1, 5, 7, 100, 1023, 4096, 70_003, 2**32 - 1):
        for world in (1, 2, 3, 5, 8, 16):
            for rank in range(world):
                assert L.iris_sample_range(spp, rank, world, C.byref(b), C.byref(n)) == 0
                assert (b.value, n.value) == sample_range(spp, rank, world)
    assert L.iris_sample_range(8, 2, 2, C.byref(b), C.byref(n)) == ffi.IRIS_OK - 1  # IRIS_E_ARG
    assert L.iris_sample_range(8, 0, 0, C.byref(b), C.byref(n)) == -1


@pytest.mark.parametrize("world,spp", [(2, 8), (3, 7)])
def test_gloo_allreduce_of_partial_films(tmp_path, oracle, world, spp):
    import torch.multiprocessing as mp
    sys.path.insert(0, HERE)
    import _dist_worker
    w, h = 24, 16
    port = free_port()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_dist_worker.run, args=(r, world, port, spp, w, h, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    from iris_amd import Scene
    sc = oracle.OracleScene(Scene.dummy().flatten())
    cam = oracle.camera_new([0, 0, 0], np.float32(w) / np.float32(h))
    full, _, _, _ = sc.render(cam, oracle.make_params(w, h, spp), nthreads=2)
    films = [np.load(tmp_path / ("film_rank%d.npy" % r)) for r in range(world)]
    for f in films[1:]:
        assert np.array_equal(f, films[0])  # every rank holds the same reduced film
    # equal to the single-process render up to f32 re-association of the per-pixel sum
    assert np.allclose(films[0], full, rtol=2e-6, atol=1e-7)
    ranges = [tuple(np.load(tmp_path / ("range_rank%d.npy" % r))) for r in range(world)]
    assert sum(n for _, n in ranges) == spp

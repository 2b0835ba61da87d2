"""Worker for tests/test_distributed.py: one rank of a world_size-N gloo job on CPU.  The partial film
comes from the oracle here (there is no GPU in the CPU test run); the code under test is the sample split
and the all-reduce in iris_amd/distributed.py."""
import os
import sys

import numpy as np


def run(rank, world, port, spp_total, w, h, out_dir):
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    sys.path.insert(0, os.path.dirname(here))
    import torch
    import torch.distributed as dist
    import oracle_ffi as O
    from iris_amd import Scene
    from iris_amd.distributed import render_distributed, sample_range

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sc = O.OracleScene(Scene.dummy().flatten())
        cam = O.camera_new([0, 0, 0], np.float32(w) / np.float32(h))

        def partial(begin, count):
            film, _, _, _ = sc.render(cam, O.make_params(w, h, spp_total, spp_begin=begin, spp_count=count), nthreads=1)
            return torch.from_numpy(film)

        film = render_distributed(partial, spp_total)
        np.save(os.path.join(out_dir, "film_rank%d.npy" % rank), film.numpy())
        np.save(os.path.join(out_dir, "range_rank%d.npy" % rank), np.array(sample_range(spp_total, rank, world)))
    finally:
        dist.destroy_process_group()

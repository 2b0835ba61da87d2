"""Multi-GPU rendering: one process per GPU, the sample indices of every pixel split across ranks, one
sum all-reduce of the partial films (RCCL over xGMI when the films are CUDA tensors, gloo on CPU).

Why this shards with a single exchange step: `Sampler::new(x, y, sample_index, SEED)` (sampler.rs:13-21,
tile.rs:145) makes every (pixel, sample index) an independent stream, and get_pixel_color is a plain sum
over sample indices scaled by 1/spp (tile.rs:144-165).  A rank that renders indices [b, b+n) of all
pixels with weight 1/spp_total therefore produces a partial film, and the partial films add up to the
full one (up to f32 re-association of the per-pixel sum).
"""
from typing import Callable, Tuple


def sample_range(spp_total: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous block of sample indices for `rank`: (begin, count).  Blocks differ by at most one
    sample when spp_total is not a multiple of world_size; a rank may get zero samples."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad rank/world_size")
    begin = (spp_total * rank) // world_size
    end = (spp_total * (rank + 1)) // world_size
    return begin, end - begin


def render_distributed(render_partial: Callable[[int, int], "torch.Tensor"], spp_total: int, group=None):
    """render_partial(spp_begin, spp_count) -> (H, W, 3) f32 torch tensor holding this rank's partial film
    (xyz sums already scaled by 1/spp_total).  Returns the full film on every rank."""
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    begin, count = sample_range(spp_total, rank, world)
    film = render_partial(begin, count)
    if world > 1:
        dist.all_reduce(film, op=dist.ReduceOp.SUM, group=group)
    return film


def render_film_multi_gpu(gpu, render, group=None, stream=None):
    """The GPU instance of render_distributed: `gpu` is this rank's HipWavefront, the partial film is
    rendered straight into a CUDA tensor and all-reduced in place by RCCL."""
    import torch

    def partial(begin, count):
        film = torch.empty((render.height, render.width, 3), dtype=torch.float32, device="cuda")
        gpu.render_film_device(render, film.data_ptr(), spp_begin=begin, spp_count=count, stream=stream)
        return film

    return render_distributed(partial, render.spp, group)

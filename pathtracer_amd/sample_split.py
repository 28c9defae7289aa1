"""Sample split of one frame over the GPUs of a node (partitioning A).

Not in the reference (single GPU, PathTracer.cu:25).  Rank r renders its own block of samples of
the SAME frame into its own running mean (strided RNG, so sample n of pixel i is the same stream
whichever GPU renders it); the one exchange step then combines the ranks' means in a FIXED
pairwise order, ((m0+m1)+(m2+m3))+... times 1/world, so the image does not depend on arrival
order.  Traffic per rank: one image out as `world` slices (all_to_all over xGMI, every link used
at once), the tree on its own slice, one all_gather of the combined slices.

`torch.distributed` is plumbing here: backend "nccl" is RCCL on ROCm; "gloo" (CPU tests) has no
all_to_all, so the same routine falls back to all_gather there.
"""
from __future__ import annotations

from typing import Callable


def shard_samples(samples_per_rank: int, rank: int) -> int:
    """RNG sample index of rank `rank`'s first sample (pt_options.sample_offset)."""
    return rank * samples_per_rank


def exchange_and_combine(mean, dist, combine_slices: Callable, out=None, scratch=None, slice_out=None):
    """mean: flat float32 tensor (4 * n_pixels) holding this rank's running mean.

    combine_slices(stacked, shards, n_pixels_in_slice, out_slice): fixed-order tree over
    `shards` shard-major slices (pt_hip_combine_shards on the GPU).
    Returns the combined frame (every rank gets all of it)."""
    import torch

    world = dist.get_world_size()
    rank = dist.get_rank()
    n = mean.numel()
    if n % (4 * world) != 0:
        raise ValueError(f"image of {n // 4} pixels does not split evenly over {world} ranks")
    slice_floats = n // world
    stacked = scratch if scratch is not None else torch.empty(world * slice_floats, dtype=mean.dtype, device=mean.device)
    if dist.get_backend() == "gloo":
        gathered = [torch.empty_like(mean) for _ in range(world)]
        dist.all_gather(gathered, mean)
        for r in range(world):
            stacked[r * slice_floats:(r + 1) * slice_floats] = gathered[r][rank * slice_floats:(rank + 1) * slice_floats]
    else:
        dist.all_to_all_single(stacked, mean)               # slice j of every rank's mean -> rank j, shard-major
    my_slice = slice_out if slice_out is not None else torch.empty(slice_floats, dtype=mean.dtype, device=mean.device)
    combine_slices(stacked, world, slice_floats // 4, my_slice)
    image = out if out is not None else torch.empty(n, dtype=mean.dtype, device=mean.device)
    dist.all_gather_into_tensor(image, my_slice) if dist.get_backend() != "gloo" else _gloo_all_gather(dist, image, my_slice, world)
    return image


def _gloo_all_gather(dist, image, my_slice, world):
    import torch

    parts = [torch.empty_like(my_slice) for _ in range(world)]
    dist.all_gather(parts, my_slice)
    k = my_slice.numel()
    for r in range(world):
        image[r * k:(r + 1) * k] = parts[r]

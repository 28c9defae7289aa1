"""Sample split of one frame over the GPUs of a node (partitioning A) -- the torch.distributed form of
the exchange (one process per GPU; backend "nccl" is RCCL on ROCm).  The C++ form is pt_dist_combine /
pt_combine_instances (include/pt_plugin.h); both follow the same layout and give the same bits.

Not in the reference (single GPU, PathTracer.cu:25).  The frame's samples are dealt to V = 8 VIRTUAL shards whatever
the GPU count (include/pt_plugin.h, PT_SPLIT_SAMPLE): shard v takes the RNG sample indices v * PT_SHARD_SPAN + k and
keeps a running mean of its own, rank g of G renders the shards {v : v % G == g}.  The image is the fixed-order
pairwise tree ((m0+m1)+(m2+m3))+((m4+m5)+(m6+m7)) times 1/8 over the shard means: the same bits on 1, 2, 4 or 8 GPUs
and for any arrival order.

The one exchange: for each of its V/G local shards a rank sends slice r of that shard's mean to rank r
(all_to_all_single: every xGMI link busy at once) and receives its own slice of the shards the others hold -- they land
shard-major in virtual-shard order --, runs the tree on its slice, and one all_gather puts the image on every rank.
`torch.distributed` is plumbing here; gloo (CPU tests) runs the very same calls.
"""
from __future__ import annotations

from typing import Callable

VIRTUAL_SHARDS = 8            # PT_VIRTUAL_SHARDS
SHARD_SPAN = 1 << 20          # PT_SHARD_SPAN


def local_shards(rank: int, world: int, virtual_shards: int = VIRTUAL_SHARDS):
    """Virtual shards rendered by `rank`, in the order the plugin keeps their means."""
    if world < 1 or virtual_shards % world:
        raise ValueError(f"{world} ranks do not divide {virtual_shards} virtual shards")
    return [rank + j * world for j in range(virtual_shards // world)]


def slice_pixels(pixel_count: int, world: int) -> int:
    """Pixels per exchanged slice: the image is padded to world * slice pixels (the plugin's pitch)."""
    return (pixel_count + world - 1) // world


def exchange_and_combine(shard_means, dist, combine_slices: Callable, stacked, image, group=None):
    """shard_means: flat float32 tensor, [V/G local shards][pitch * 4] with pitch = world * slice pixels
    (pt_copy_mean of a PT_SPLIT_SAMPLE instance).  stacked: scratch, V * slice * 4 floats.  image: pitch * 4 floats,
    the result, on every rank.  combine_slices(stacked, shards, n_pixels_in_slice, out_slice): the fixed-order tree over
    `shards` shard-major slices (pt_hip_combine_shards on the GPU)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    pitch_floats = image.numel()
    if pitch_floats % (4 * world) or shard_means.numel() % pitch_floats:
        raise ValueError("image must be world * slice pixels, shard_means a whole number of images")
    n_local = shard_means.numel() // pitch_floats
    slice_floats = pitch_floats // world
    if stacked.numel() < n_local * world * slice_floats:
        raise ValueError("stacked too small")
    for j in range(n_local):
        # slice r of local shard j (virtual shard rank + j*world) -> rank r; from rank r comes my slice of ITS shard
        # r + j*world, which is position j*world + r of `stacked`: shard-major in virtual-shard order
        dist.all_to_all_single(stacked[j * world * slice_floats:(j + 1) * world * slice_floats],
                               shard_means[j * pitch_floats:(j + 1) * pitch_floats], group=group)
    mine = image[rank * slice_floats:(rank + 1) * slice_floats]
    combine_slices(stacked, n_local * world, slice_floats // 4, mine)
    dist.all_gather_into_tensor(image, mine, group=group)
    return image

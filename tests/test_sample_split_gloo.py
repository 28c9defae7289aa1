"""The N>1 path of bench.py on CPU: two gloo ranks run pathtracer_amd.sample_split's exchange with
the oracle standing in for the renderer and for the combine kernel; the result must be, bit for
bit, the oracle's sharded mean (the definition in oracle/pt_oracle.h)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H, SPP_PER_RANK = 32, 16, 3


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import ctypes as C
    from oracle import pt_oracle_py as po
    from pathtracer_amd import sample_split

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # this rank's shard: its own running mean over its own samples, strided RNG
    offset = sample_split.shard_samples(SPP_PER_RANK, rank)
    mean = np.ones((W * H, 4), dtype=np.float32)
    p = po.params(W, H, rng_mode=po.RNG_STRIDED, stride=35, spp=1)
    for s in range(SPP_PER_RANK):
        p.frame_first = offset + s
        _, _, _, frame = po.render(p, want_last_frame=True)
        r = np.float32(1.0) / np.float32(s + 1)
        mean = (frame * r + (mean * np.float32(s)) * r).astype(np.float32)

    def combine(stacked, shards, n_pix, out_slice):
        src = np.ascontiguousarray(stacked.numpy())
        dst = np.empty(n_pix * 4, dtype=np.float32)
        po.lib().pto_combine_shards(src.ctypes.data, shards, n_pix, dst.ctypes.data)
        out_slice.copy_(torch.from_numpy(dst))

    image = sample_split.exchange_and_combine(torch.from_numpy(mean.reshape(-1).copy()), dist, combine)
    np.save(os.path.join(out_dir, f"rank{rank}.npy"), image.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_two_rank_sample_split_equals_sharded_oracle(oracle, tmp_path, world):
    port = 29500 + (os.getpid() % 2000) + world
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    p = oracle.params(W, H, rng_mode=oracle.RNG_STRIDED, stride=35, spp=SPP_PER_RANK * world)
    want, _ = oracle.render_sharded(p, world)
    for r in range(world):
        got = np.load(tmp_path / f"rank{r}.npy").reshape(-1, 4)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), f"rank {r}"


def test_shard_offsets():
    from pathtracer_amd import sample_split
    assert [sample_split.shard_samples(64, r) for r in range(4)] == [0, 64, 128, 192]

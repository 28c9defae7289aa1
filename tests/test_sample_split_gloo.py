"""The N>1 path of bench.py on CPU: gloo ranks run pathtracer_amd.sample_split's exchange (the very calls RCCL runs on
the GPUs) with the oracle standing in for the renderer and for the combine kernel.  The frame is dealt to 8 virtual
shards whatever the world size, so worlds of 1, 2, 4 and 8 ranks must all produce, bit for bit, the oracle's sharded
mean over 8 shards (oracle/pt_oracle.h, pto_render_sharded_span) -- and hence the same image as each other."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H, SPP_PER_SHARD = 33, 17, 2          # 561 pixels: not a multiple of 2, 4 or 8, so the slices are padded


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    from oracle import pt_oracle_py as po
    from pathtracer_amd import sample_split as ss

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n = W * H
    slice_px = ss.slice_pixels(n, world)
    pitch = slice_px * world
    mine = ss.local_shards(rank, world)
    means = np.ones((len(mine), pitch, 4), dtype=np.float32)          # padding pixels stay at the white start value
    p = po.params(W, H, rng_mode=po.RNG_STRIDED, stride=35, spp=1)
    for j, v in enumerate(mine):
        mean = np.ones((n, 4), dtype=np.float32)
        for s in range(SPP_PER_SHARD):
            p.frame_first = v * ss.SHARD_SPAN + s
            _, _, _, frame = po.render(p, want_last_frame=True)
            r = np.float32(1.0) / np.float32(s + 1)
            mean = (frame * r + (mean * np.float32(s)) * r).astype(np.float32)
        means[j, :n] = mean

    def combine(stacked, shards, n_pix, out_slice):
        src = np.ascontiguousarray(stacked[:shards * n_pix * 4].numpy())
        dst = np.empty(n_pix * 4, dtype=np.float32)
        po.lib().pto_combine_shards(src.ctypes.data, shards, n_pix, dst.ctypes.data)
        out_slice.copy_(torch.from_numpy(dst))

    stacked = torch.empty(ss.VIRTUAL_SHARDS * slice_px * 4)
    image = torch.empty(pitch * 4)
    ss.exchange_and_combine(torch.from_numpy(means.reshape(-1).copy()), dist, combine, stacked, image)
    np.save(os.path.join(out_dir, f"rank{rank}.npy"), image.numpy()[:n * 4])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_sample_split_is_the_sharded_oracle_for_every_world_size(oracle, tmp_path, world):
    port = 29500 + (os.getpid() % 2000) + world
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    p = oracle.params(W, H, rng_mode=oracle.RNG_STRIDED, stride=35, spp=SPP_PER_SHARD * 8)
    want, _ = oracle.render_sharded(p, 8, span=1 << 20)
    for r in range(world):
        got = np.load(tmp_path / f"rank{r}.npy").reshape(-1, 4)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), f"world {world} rank {r}"


def test_shard_ownership():
    from pathtracer_amd import sample_split as ss
    assert ss.local_shards(0, 1) == list(range(8))
    assert ss.local_shards(1, 2) == [1, 3, 5, 7]
    assert ss.local_shards(3, 4) == [3, 7]
    assert [ss.local_shards(r, 8) for r in range(8)] == [[r] for r in range(8)]
    assert sorted(v for r in range(4) for v in ss.local_shards(r, 4)) == list(range(8))
    with pytest.raises(ValueError):
        ss.local_shards(0, 3)
    assert ss.slice_pixels(561, 4) == 141 and ss.slice_pixels(2073600, 8) == 259200

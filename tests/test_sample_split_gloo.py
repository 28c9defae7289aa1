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


def _plan_worker(rank, world, port, out_dir):
    """The C-ABI's own exchange plan (pt_hip_comm_exchange_plan: the layout pt_hip_comm_exchange_combine sends and receives by over
    RCCL) executed over gloo with point-to-point sends, beside sample_split.exchange_and_combine's all_to_all on the same data."""
    import ctypes as C
    sys.path.insert(0, ROOT)
    import pathtracer_amd as pt
    from pathtracer_amd import sample_split as ss

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lib = pt.load_library()
    n = 561
    slice_px = ss.slice_pixels(n, world)
    pitch = slice_px * world
    local = ss.VIRTUAL_SHARDS // world
    # every float names its (virtual shard, pixel, channel): a misplaced slice cannot go unnoticed
    means = torch.empty(local, pitch, 4)
    for j, v in enumerate(ss.local_shards(rank, world)):
        means[j] = (v * 1000000 + torch.arange(pitch * 4, dtype=torch.float32).reshape(pitch, 4))
    means = means.reshape(-1).contiguous()

    class Step(C.Structure):
        _fields_ = [("peer", C.c_uint32), ("reserved", C.c_uint32), ("send_offset", C.c_uint64), ("recv_offset", C.c_uint64), ("count", C.c_uint64)]
    steps = (Step * (local * world))()
    assert lib.pt_hip_comm_exchange_plan(rank, world, local, slice_px, C.cast(steps, C.c_void_p)) == 0, lib.pt_hip_last_error()
    stacked = torch.full((ss.VIRTUAL_SHARDS * slice_px * 4,), -1.0)
    ops = []
    for st in steps:
        assert st.count == slice_px * 4
        if st.peer == rank:                       # (RCCL pairs a rank's send with its own receive inside the group; gloo has no pair to itself)
            stacked[st.recv_offset:st.recv_offset + st.count] = means[st.send_offset:st.send_offset + st.count]
            continue
        ops.append(dist.P2POp(dist.isend, means[st.send_offset:st.send_offset + st.count], st.peer))
        ops.append(dist.P2POp(dist.irecv, stacked[st.recv_offset:st.recv_offset + st.count], st.peer))
    for w in dist.batch_isend_irecv(ops):
        w.wait()
    # the torch form of the same exchange
    stacked_ref = torch.full_like(stacked, -2.0)
    image = torch.empty(pitch * 4)
    ss.exchange_and_combine(means, dist, lambda st_, shards, n_pix, out: out.zero_(), stacked_ref, image)
    ok = bool(torch.equal(stacked, stacked_ref))
    # ... and what must have arrived: this rank's slice of every virtual shard, shard-major in virtual-shard order
    want = torch.cat([(v * 1000000 + torch.arange(pitch * 4, dtype=torch.float32))[rank * slice_px * 4:(rank + 1) * slice_px * 4] for v in range(ss.VIRTUAL_SHARDS)])
    ok = ok and bool(torch.equal(stacked, want))
    open(os.path.join(out_dir, f"plan{rank}.txt"), "w").write("ok" if ok else "MISMATCH")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_native_exchange_plan_is_the_torch_exchange(pt, tmp_path, world):
    """the grouped ncclSend / ncclRecv of pt_hip_comm_exchange_combine have never run with more than one rank (RCCL refuses
    two ranks on one device); their index layout is data (pt_hip_comm_exchange_plan), and this runs it over gloo."""
    port = 31500 + (os.getpid() % 2000) + world
    mp.spawn(_plan_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert (tmp_path / f"plan{r}.txt").read_text() == "ok", f"world {world} rank {r}"


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

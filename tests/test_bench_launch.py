"""bench.py's N > 1 path as the driver types it.

CPU: `python bench.py --gpus N` without a launcher starts its ranks itself (launch_ranks): the child must be a
torch.distributed.run launch on 127.0.0.1 with N ranks, started BEFORE the parent has imported torch or touched a GPU, and
the child's return code must come back.  Driven here with a stub script in place of bench.py.

GPU (one card): `bench.py --gpus 1 --force-dist` runs the whole sample-split path with a world of one -- 8 virtual shards
on one rank, the RCCL calls included -- through both exchanges; one JSON line, rccl_ranks == 1, the per-phase fields, and
the image's bits against the oracle's sharded mean over 8 shards."""
import json
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_launch_ranks_starts_the_ranks_before_anything_touches_a_gpu(tmp_path):
    stub = tmp_path / "stub_rank.py"
    stub.write_text(textwrap.dedent("""
        import os, sys
        out = sys.argv[1]
        rank, world = os.environ["RANK"], os.environ["WORLD_SIZE"]
        with open(os.path.join(out, "rank" + rank), "w") as f:
            f.write(" ".join([world, os.environ["LOCAL_RANK"], os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"],
                              os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "unset")] + sys.argv[2:]))
        sys.exit(int(sys.argv[2]) if rank == "1" else 0)
    """))
    driver = textwrap.dedent(f"""
        import sys
        sys.path.insert(0, {ROOT!r})
        import bench
        rc = bench.launch_ranks(2, [{str(tmp_path)!r}, sys.argv[1], "--flag"], script={str(stub)!r})
        # the parent has not imported torch (let alone torch.cuda) nor loaded the plugin: nothing of it ever touches a GPU
        assert "torch" not in sys.modules and "pathtracer_amd" not in sys.modules, sorted(m for m in sys.modules if m.startswith(("torch", "pathtracer")))
        print("RC", rc)
    """)
    env = {k: v for k, v in os.environ.items() if k != "HSA_ENABLE_IPC_MODE_LEGACY"}
    # ranks that succeed: return code 0 comes back
    r = subprocess.run([sys.executable, "-c", driver, "0"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "RC 0" in r.stdout.split("\n"), r.stdout + r.stderr
    seen = sorted(p.name for p in tmp_path.glob("rank*"))
    assert seen == ["rank0", "rank1"]
    ports = set()
    for k in (0, 1):
        world, local, addr, port, ipc, *rest = (tmp_path / f"rank{k}").read_text().split()
        assert (world, local, addr, ipc) == ("2", str(k), "127.0.0.1", "0") and rest == ["0", "--flag"]
        ports.add(port)
    assert len(ports) == 1
    # a rank that fails: the launcher's non-zero code comes back
    r = subprocess.run([sys.executable, "-c", driver, "7"], env=env, capture_output=True, text=True, timeout=300)
    codes = [ln.split()[1] for ln in r.stdout.split("\n") if ln.startswith("RC ")]
    assert r.returncode == 0 and len(codes) == 1 and int(codes[0]) != 0, r.stdout + r.stderr


def test_gpus_must_match_the_world():
    """under an external launcher bench.py is one of the ranks: a --gpus that disagrees with WORLD_SIZE is refused before any GPU work"""
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("exchange", ["torch", "native"])
def test_force_dist_runs_the_whole_sample_split_on_one_gpu(oracle, tmp_path, exchange):
    w, h, steps, warmup = 320, 180, 2, 1
    dump = tmp_path / "image.npy"
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env["MASTER_PORT"] = str(29600 + os.getpid() % 300 + (0 if exchange == "torch" else 1))
    env["MASTER_ADDR"] = "127.0.0.1"
    env["RANK"], env["WORLD_SIZE"], env["LOCAL_RANK"] = "0", "1", "0"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist", "--steps", str(steps), "--warmup", str(warmup),
                        "--exchange", exchange, "--width", str(w), "--height", str(h), "--dump-image", str(dump), "--settle-seconds", "0"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout                                  # ONE JSON line on stdout, nothing else
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["rccl_ranks"] == 1 and out["image_identical_on_all_ranks"] is True
    assert out["exchanges_timed"] == 1 and out["steps"] == steps and out["scaling"] == "weak" and out["unit"] == "Mrays/s"
    assert out["value"] > 0 and out["render_ms_max"] > 0 and out["exchange_ms"] >= 0 and out["render_ms_min"] <= out["render_ms_max"]
    assert abs(out["render_ms_max"] + out["exchange_ms"] - out["ms_per_step"] * steps) < 1e-2 * out["ms_per_step"] * steps + 0.05
    assert ("pt_dist_combine" if exchange == "native" else "torch.distributed") in out["config"]["parallelism"]
    # the image: 8 virtual shards, each (12 probe renders + warmup + steps) x 8 samples, combined by the fixed-order tree
    per_shard = (12 + warmup + steps) * (64 // 8)
    want, _ = oracle.render_sharded(oracle.params(w, h, rng_mode=oracle.RNG_STRIDED, stride=35, spp=per_shard * 8), 8, span=1 << 20)
    got = np.load(dump)
    assert got.shape == want.shape and np.array_equal(got.view(np.uint32), want.view(np.uint32))

#!/usr/bin/env python3
"""gpurun_out/<tag>/ (what tools/profile_round.sh left) -> profiles/: the files a round commits.
Usage: python3 tools/collect_profiles.py <tag> <round>      e.g. r02b r02
  profiles/traffic.json                      the PMC counters bench.py prices its roofline with (hash of the kernel sources inside)
  profiles/<round>_<workload>_<engine>_kernel_stats.csv   rocprofv3 --kernel-trace of `bench.py --bare --workload W --engine E --steps 50`, launches after the marker (tools/steady.py)
  profiles/<round>_<workload>_<engine>_traced_bench.json  the bench line of that traced run (ms per step under the tracer)
  profiles/<round>_bench.json                the bench line of that run (the JSON line only)
  profiles/<round>_pytest_gpu.txt            the tail of the GPU test run"""
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, rnd = sys.argv[1], sys.argv[2]
src = os.path.join(ROOT, "gpurun_out", tag)
dst = os.path.join(ROOT, "profiles")
if os.path.exists(os.path.join(src, "traffic.json")):
    shutil.copy(os.path.join(src, "traffic.json"), os.path.join(dst, "traffic.json"))
# kernel traces: the steady-state summary (tools/steady.py: launches after the bench's marker only) and the bench line of the traced run
for f in sorted(glob.glob(os.path.join(src, "kt_*_steady.csv"))):
    name = os.path.basename(f)[3:-len("_steady.csv")]
    shutil.copy(f, os.path.join(dst, f"{rnd}_{name}_kernel_stats.csv"))
    b = f[:-len("_steady.csv")] + "_bench.json"
    if os.path.exists(b) and os.path.getsize(b):
        d = json.loads(open(b).readline())
        with open(os.path.join(dst, f"{rnd}_{name}_traced_bench.json"), "w") as out:
            json.dump({k: d.get(k) for k in ("value", "unit", "steps", "warmup", "ms_per_step", "config", "roofline")}, out, indent=1)
for name in ("bench.json", "bench_wavefront.json"):
    p = os.path.join(src, name)
    if os.path.exists(p):
        lines = [l for l in open(p) if l.startswith("{")]
        if lines:
            with open(os.path.join(dst, f"{rnd}_{name}"), "w") as f:
                f.write(json.dumps(json.loads(lines[-1]), indent=1) + "\n")
p = os.path.join(src, "pytest_gpu.log")
if os.path.exists(p):
    with open(os.path.join(dst, f"{rnd}_pytest_gpu.txt"), "w") as f:
        f.writelines(open(p).readlines()[-5:])
print(sorted(os.listdir(dst)))

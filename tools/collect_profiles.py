#!/usr/bin/env python3
"""gpurun_out/<tag>/ (what tools/profile_round.sh left) -> profiles/: the files a round commits.
Usage: python3 tools/collect_profiles.py <tag> <round>      e.g. r02b r02
  profiles/traffic.json                      the PMC counters bench.py prices its roofline with (hash of the kernel sources inside)
  profiles/<round>_<workload>_<engine>_kernel_stats.csv   rocprofv3 --kernel-trace --stats of `bench.py --bare --workload W --engine E`
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
for d in sorted(glob.glob(os.path.join(src, "kt_*/"))):
    name = os.path.basename(d.rstrip("/"))[3:]
    for f in glob.glob(d + "*/*kernel_stats.csv"):
        shutil.copy(f, os.path.join(dst, f"{rnd}_{name}_kernel_stats.csv"))
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

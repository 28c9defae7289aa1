#!/usr/bin/env python3
"""rocprofv3 --pmc passes of tools/measure_counters.sh -> the profiles/traffic.json bench.py reads: per (workload, engine)
the counters of ONE bench step (sums over the render kernels of the run / steps run), with the hash of the kernel
sources they were measured on (bench.py marks its roofline "stale" when the sources have changed since).
HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE are in KiB on separate passes, and
gfx950's FETCH_SIZE reports half of the bytes of wide coalesced reads: x2."""
import collections
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

root, steps_run = sys.argv[1], int(sys.argv[2])
RENDER = ("k_fused", "k_bounces", "k_bounce_tail", "k_extend", "k_shade", "k_generate", "k_accumulate", "k_fold_counts", "k_set_count", "k_sample_matrices")
entries = {}
for d in sorted(glob.glob(os.path.join(root, "pmc_*_p*/"))):
    m = re.match(r"pmc_(.+)_(fused|wavefront)_p(\d+)", os.path.basename(d.rstrip("/")))
    if not m:
        continue
    key = f"{m.group(1)}/{m.group(2)}"
    sums = collections.defaultdict(float)
    per_kernel = collections.defaultdict(lambda: collections.defaultdict(float))
    rows = collections.defaultdict(int)
    for f in glob.glob(d + "*/*counter_collection.csv"):
        for row in csv.DictReader(open(f)):
            km = re.search(r"(k_[a-z_0-9]+(<[^>]*>)?)", row["Kernel_Name"])
            if not km or not km.group(1).startswith(RENDER):
                continue
            v = float(row["Counter_Value"])
            sums[row["Counter_Name"]] += v
            per_kernel[km.group(1)][row["Counter_Name"]] += v
            rows[row["Counter_Name"]] += 1
    e = entries.setdefault(key, {})
    for c, v in sums.items():
        e.setdefault("_raw", {})[c] = v / steps_run
    if "SQ_INSTS_VALU" in sums:
        e["valu_insts"] = int(round(sums["SQ_INSTS_VALU"] / steps_run))
        e["launches_per_step"] = round(rows["SQ_INSTS_VALU"] / steps_run, 2)
        e["lanes_per_instruction"] = round(sums["SQ_THREAD_CYCLES_VALU"] / sums["SQ_INSTS_VALU"], 1) if sums.get("SQ_THREAD_CYCLES_VALU") else None
        dom = max(per_kernel.items(), key=lambda kv: kv[1].get("SQ_INSTS_VALU", 0.0))
        e["kernel"] = dom[0]
        e["kernels"] = {k: int(round(v.get("SQ_INSTS_VALU", 0.0) / steps_run)) for k, v in per_kernel.items()}
    if "FETCH_SIZE" in sums:
        e["hbm_fetch_bytes"] = int(round(sums["FETCH_SIZE"] * 1024 * 2 / steps_run))
    if "WRITE_SIZE" in sums:
        e["hbm_write_bytes"] = int(round(sums["WRITE_SIZE"] * 1024 / steps_run))
    if "TA_TA_BUSY_sum" in sums and "GRBM_GUI_ACTIVE" in sums:
        e["ta_busy_per_gui_cycle"] = round(sums["TA_TA_BUSY_sum"] / sums["GRBM_GUI_ACTIVE"], 2)
for e in entries.values():
    if "hbm_fetch_bytes" in e and "hbm_write_bytes" in e:
        e["hbm_bytes"] = e["hbm_fetch_bytes"] + e["hbm_write_bytes"]
json.dump({"_comment": "per bench step (64 spp at 1920x1080, one pt_render): wave64 VALU instructions (SQ_INSTS_VALU), active lanes per instruction "
                       "(SQ_THREAD_CYCLES_VALU / SQ_INSTS_VALU), HBM bytes (FETCH_SIZE x 2 + WRITE_SIZE, KiB -> bytes), from separate rocprofv3 --pmc "
                       "passes over `bench.py --bare --workload W --engine E` (tools/measure_counters.sh); sums over the render kernels of a step",
           "sha": bench.kernel_sources_sha(), "sources": bench.KERNEL_SOURCES, "steps_averaged": steps_run, "entries": entries}, sys.stdout, indent=1)

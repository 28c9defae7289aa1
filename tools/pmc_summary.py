#!/usr/bin/env python3
"""Per-kernel means of rocprofv3 --pmc passes under <dir>/pmc_<engine>_<counter>/ -> JSON on stdout."""
import collections
import csv
import glob
import json
import re
import sys

root = sys.argv[1]
res = {}
for eng in ("fused", "wavefront"):
    for d in sorted(glob.glob(f"{root}/pmc_{eng}_*/")):
        files = glob.glob(d + "*/*counter_collection.csv")
        if not files:
            continue
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(files[0])):
            m = re.search(r"(k_[a-z_0-9]+(<[^>]*>)?)", row["Kernel_Name"])
            if m:
                agg[m.group(1)][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, v in agg.items():
            if any(x in k for x in ("k_fused", "k_shade", "k_extend", "k_generate", "k_accumulate", "k_bounces")):
                for c, vals in v.items():
                    res.setdefault(eng, {}).setdefault(k, {})[c] = {"mean": sum(vals) / len(vals), "max": max(vals), "n": len(vals)}
json.dump(res, sys.stdout, indent=1)

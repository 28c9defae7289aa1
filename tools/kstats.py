#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats kernel_stats.csv: one short line per kernel."""
import csv
import glob
import re
import sys

path = sys.argv[1]
files = glob.glob(path + "/*/*kernel_stats.csv") + glob.glob(path + "/*kernel_stats.csv") if not path.endswith(".csv") else [path]
for r in csv.DictReader(open(files[0])):
    m = re.search(r"(k_[a-z_0-9]+(<[^>]*>)?|__amd\w+)", r["Name"])
    name = m.group(1) if m else r["Name"][:40]
    print("%-34s calls %5s avg_us %10.1f max_us %10.1f total_ms %8.2f %s%%" % (
        name, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MaxNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))

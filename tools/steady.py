#!/usr/bin/env python3
"""Steady-state views of rocprofv3 output.  `bench.py --bare` run with PT_BENCH_MARKER=1 launches one k_fill_f32 of a single
element between its untimed renders (the plugin measures its candidate kernels there) and its timed steps; everything a
profile reports is taken from the dispatches AFTER that marker, so that a kernel's average is the average of the launches the
bench line times and nothing else (the round-2 summaries included the probe launches and cold calls).

  python3 tools/steady.py <rocprofv3 output dir> [out.csv]     per kernel: calls, average / min / max / total ns after the marker"""
import collections
import csv
import glob
import re
import sys

MARKER = "k_fill_f32"


def short_name(kernel_name):
    m = re.search(r"(k_[a-z_0-9]+(<[^>]*>)?|__amd\w+)", kernel_name)
    return m.group(1) if m else kernel_name[:48]


def steady_rows(path_glob):
    """rows of every CSV matching the glob, dispatch order, after the last marker dispatch (all rows if there is no marker)"""
    rows = []
    for f in sorted(glob.glob(path_glob)):
        rows += list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    last = max((int(r["Dispatch_Id"]) for r in rows if short_name(r["Kernel_Name"]) == MARKER), default=-1)
    return [r for r in rows if int(r["Dispatch_Id"]) > last], last >= 0


def trace_summary(out_dir):
    rows, marked = steady_rows(out_dir.rstrip("/") + "/*/*kernel_trace.csv")
    per = collections.OrderedDict()
    for r in rows:
        per.setdefault(short_name(r["Kernel_Name"]), []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    total = sum(sum(v) for v in per.values()) or 1
    out = []
    for name, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
        out.append({"Name": name, "Calls": len(v), "TotalDurationNs": sum(v), "AverageNs": sum(v) / len(v), "MinNs": min(v), "MaxNs": max(v),
                    "Percentage": round(100.0 * sum(v) / total, 2)})
    return out, marked


if __name__ == "__main__":
    summary, marked = trace_summary(sys.argv[1])
    w = csv.DictWriter(open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout, fieldnames=["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "Percentage"])
    w.writeheader()
    for row in summary:
        w.writerow(row)
    if not marked:
        print("no marker dispatch found: the summary covers the whole run", file=sys.stderr)

#!/usr/bin/env python3
"""rocprofv3 --pmc passes of tools/measure_counters.sh -> the profiles/traffic.json bench.py reads.

Per (workload, engine): the counters of ONE timed bench step, taken from the launches AFTER the bench's marker dispatch only
(tools/steady.py: the plugin's probe launches and the cold calls are not in it), per kernel PER LAUNCH (each kernel's sum over
its own number of launches), with each kernel's average duration from the kernel trace of the same command and its own fraction
of the VALU issue peak -- so that bench.py's roofline can be recomputed from profiles/ alone: instructions per launch of the
timed kernel / its average duration (profiles/rNN_<workload>_<engine>_kernel_stats.csv) / 1228.8 G instructions/s.
HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE are in KiB on separate passes, and
gfx950's FETCH_SIZE reports half of the bytes of wide coalesced reads: x2.

Usage: python3 tools/counters_to_json.py <gpurun_out/tag> <timed steps of the pmc runs>"""
import collections
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench   # noqa: E402
import steady  # noqa: E402

root, steps = sys.argv[1], int(sys.argv[2])
XCDS = 8            # MI355X: GRBM_GUI_ACTIVE is the sum of one counter per XCD
RENDER = ("k_fused", "k_bounces", "k_bounce_tail", "k_extend", "k_shade", "k_generate", "k_camera_lists", "k_accumulate", "k_fold_counts", "k_fold_partials",
          "k_set_count", "k_sample_matrices")
entries = {}
for d in sorted(glob.glob(os.path.join(root, "pmc_*_p*/"))):
    m = re.match(r"pmc_(.+)_(fused|wavefront)_p(\d+)", os.path.basename(d.rstrip("/")))
    if not m:
        continue
    key = f"{m.group(1)}/{m.group(2)}"
    rows, marked = steady.steady_rows(d + "*/*counter_collection.csv")
    e = entries.setdefault(key, {"steady_state": marked, "kernels": {}})
    e["steady_state"] = e["steady_state"] and marked
    per = collections.defaultdict(lambda: collections.defaultdict(list))        # kernel -> counter -> per-dispatch values
    clock = []
    for r in rows:
        name = steady.short_name(r["Kernel_Name"])
        if not name.startswith(RENDER):
            continue
        v = float(r["Counter_Value"])
        per[name][r["Counter_Name"]].append(v)
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            ns = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            if ns > 20000:
                clock.append(v / ns)
    for name, counters in per.items():
        k = e["kernels"].setdefault(name, {})
        for c, vals in counters.items():
            k.setdefault("_per_launch", {})[c] = sum(vals) / len(vals)
            k["launches_per_step"] = round(len(vals) / steps, 3)
    if clock:
        # GRBM_GUI_ACTIVE comes summed over the 8 XCDs' own counters (18-19 "cycles" per ns of a dispatch): / 8 = the clock the run held
        e["gui_cycles_per_ns"] = round(sum(clock) / len(clock), 3)
        e["clock_ghz"] = round(sum(clock) / len(clock) / XCDS, 3)
# kernel-trace durations of the same commands (tools/profile_round.sh: kt_<workload>_<engine>)
for key, e in entries.items():
    wl, eng = key.split("/")
    summary, _ = steady.trace_summary(os.path.join(root, f"kt_{wl}_{eng}"))
    dur = {row["Name"]: row["AverageNs"] for row in summary}
    tot = collections.defaultdict(float)
    for name, k in e["kernels"].items():
        pl, n = k.get("_per_launch", {}), k.get("launches_per_step", 0)
        for c, v in pl.items():
            tot[c] += v * n
        if "SQ_INSTS_VALU" in pl:
            k["valu_insts_per_launch"] = int(round(pl["SQ_INSTS_VALU"]))
            k["lanes_per_instruction"] = round(pl["SQ_THREAD_CYCLES_VALU"] / pl["SQ_INSTS_VALU"], 1) if pl.get("SQ_THREAD_CYCLES_VALU") and pl["SQ_INSTS_VALU"] else None
            if name in dur:
                k["avg_us"] = round(dur[name] / 1e3, 2)
                k["valu_frac"] = round(pl["SQ_INSTS_VALU"] / dur[name] / bench.VALU_PEAK_GINST, 4)
    e["_raw_per_step"] = {c: v for c, v in tot.items()}
    if tot.get("SQ_INSTS_VALU"):
        e["valu_insts"] = int(round(tot["SQ_INSTS_VALU"]))
        e["launches_per_step"] = round(sum(k.get("launches_per_step", 0) for k in e["kernels"].values() if "valu_insts_per_launch" in k), 2)
        e["lanes_per_instruction"] = round(tot["SQ_THREAD_CYCLES_VALU"] / tot["SQ_INSTS_VALU"], 1) if tot.get("SQ_THREAD_CYCLES_VALU") else None
        e["kernel"] = max((kv for kv in e["kernels"].items() if "valu_insts_per_launch" in kv[1]),
                          key=lambda kv: kv[1]["valu_insts_per_launch"] * kv[1]["launches_per_step"])[0]
    if "FETCH_SIZE" in tot:
        e["hbm_fetch_bytes"] = int(round(tot["FETCH_SIZE"] * 1024 * 2))
    if "WRITE_SIZE" in tot:
        e["hbm_write_bytes"] = int(round(tot["WRITE_SIZE"] * 1024))
    if "hbm_fetch_bytes" in e and "hbm_write_bytes" in e:
        e["hbm_bytes"] = e["hbm_fetch_bytes"] + e["hbm_write_bytes"]
    if tot.get("TA_TA_BUSY_sum") and tot.get("GRBM_GUI_ACTIVE"):
        # TA_TA_BUSY_sum adds the busy cycles of every texture-address unit; _avr (same pass) is their mean: sum / avr = the units counted
        units = tot["TA_TA_BUSY_sum"] / tot["TA_BUSY_avr"] if tot.get("TA_BUSY_avr") else None
        e["ta_units"] = round(units, 1) if units else None
        e["ta_busy_per_gui_cycle"] = round(tot["TA_TA_BUSY_sum"] / tot["GRBM_GUI_ACTIVE"], 2)
        # busy share of a texture-address unit: its mean busy cycles over the cycles ONE XCD's clock counted (GRBM_GUI_ACTIVE / 8)
        e["ta_busy_pct"] = round(100.0 * tot["TA_BUSY_avr"] * XCDS / tot["GRBM_GUI_ACTIVE"], 1) if tot.get("TA_BUSY_avr") else None
json.dump({"_comment": "per TIMED bench step (64 spp at 1920x1080, one pt_render; launches after the bench's marker dispatch only): wave64 VALU instructions "
                       "(SQ_INSTS_VALU), active lanes per instruction (SQ_THREAD_CYCLES_VALU / SQ_INSTS_VALU), HBM bytes (FETCH_SIZE x 2 + WRITE_SIZE, KiB -> bytes), "
                       "from separate rocprofv3 --pmc passes over `bench.py --bare --workload W --engine E` (tools/measure_counters.sh); per kernel: counts PER LAUNCH, "
                       "average duration in the kernel trace of the same command, valu_frac = instructions per launch / that duration / 1228.8 G/s",
           "sha": bench.kernel_sources_sha(), "sources": bench.KERNEL_SOURCES, "steps_timed": steps, "entries": entries}, sys.stdout, indent=1)

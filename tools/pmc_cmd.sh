#!/bin/bash
# SQ/TCP/TCC counter passes over any python command: bash tools/pmc_cmd.sh <tag> <script.py> [args...]
# -> gpurun_out/<tag>/p<i>/ ; prints per-kernel means.  Separate passes, never combined with tracing.
TAG=$1; shift
export TMPDIR=/tmp
OUT=gpurun_out/$TAG
mkdir -p $OUT
i=0
for c in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_BRANCH" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" \
         "SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_RD SQ_IFETCH" \
         "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum" \
         "FETCH_SIZE" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d $OUT/p$i -- python "$@" > $OUT/p$i.log 2>&1 || echo "pass $i failed"
done
python - "$OUT" <<'PY'
import collections, csv, glob, re, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/p*/*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        m = re.search(r"(k_[a-z_0-9]+(<[^>]*>)?)", row["Kernel_Name"])
        if m:
            agg[m.group(1)][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, v in agg.items():
    tot = sum(v.get("SQ_WAVE_CYCLES", [0]))
    if tot < 1e6:
        continue
    print(k)
    for c, vals in sorted(v.items()):
        print("   %-30s mean %.5g  n %d" % (c, sum(vals) / len(vals), len(vals)))
PY

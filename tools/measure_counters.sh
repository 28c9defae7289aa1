#!/bin/bash
# The counters bench.py's roofline blocks are priced with: for every (workload, engine) bench.py reports, separate
# rocprofv3 --pmc passes (never combined with tracing) over `python3 bench.py --bare --workload W --engine E`:
# wave64 VALU instructions + active lanes, HBM fetch bytes, HBM write bytes.  Then a kernel-trace of the default bench.
# Usage (GPU box, repo root): bash tools/measure_counters.sh <tag>   -> gpurun_out/<tag>/..., gpurun_out/<tag>/traffic.json
TAG=${1:-r02}
export TMPDIR=/tmp
OUT=gpurun_out/$TAG
mkdir -p $OUT
STEPS=8; WARM=1; SETTLE=4      # bench.py renders SETTLE untimed steps first (the plugin measures its candidates there)
for wl in default c5 c4_10k c4_100k; do
  for eng in fused wavefront; do
    i=0
    for c in "SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "FETCH_SIZE" "WRITE_SIZE" "TA_TA_BUSY_sum GRBM_GUI_ACTIVE"; do
      i=$((i+1))
      timeout -s KILL 150 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_${wl}_${eng}_p$i -- python3 bench.py --bare --workload $wl --engine $eng --steps $STEPS --warmup $WARM > $OUT/pmc_${wl}_${eng}_p$i.log 2>&1 || echo "pass $wl $eng $i failed"
      echo "pass $wl $eng $i done"
    done
  done
done
python3 tools/counters_to_json.py $OUT $((STEPS+WARM+SETTLE)) > $OUT/traffic.json && echo counters_done

#!/bin/bash
# The counters bench.py's roofline blocks are priced with: for every (workload, engine) bench.py reports, separate
# rocprofv3 --pmc passes (never combined with tracing) over `python3 bench.py --bare --workload W --engine E`:
# wave64 VALU instructions + active lanes, HBM fetch bytes, HBM write bytes, texture-address busy.  PT_BENCH_MARKER makes the
# bench mark where its timed steps begin (tools/steady.py): only the launches after the mark are counted.
# Usage (GPU box, repo root): bash tools/measure_counters.sh <tag>   -> gpurun_out/<tag>/pmc_*; then, where the kernel traces of
# tools/profile_round.sh are too (per-kernel durations come from them): python3 tools/counters_to_json.py gpurun_out/<tag> 8
TAG=${1:-r03}
export TMPDIR=/tmp
export PT_BENCH_MARKER=1
OUT=gpurun_out/$TAG
mkdir -p $OUT
STEPS=8; WARM=1
for wl in ${WORKLOADS:-default c5 c4_10k c4_100k}; do
  for eng in ${ENGINES:-fused wavefront}; do
    i=0
    for c in "SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "FETCH_SIZE" "WRITE_SIZE" "TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE"; do
      i=$((i+1))
      rm -rf $OUT/pmc_${wl}_${eng}_p$i
      timeout -s KILL 150 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_${wl}_${eng}_p$i -- python3 bench.py --bare --workload $wl --engine $eng --steps $STEPS --warmup $WARM --settle-seconds 0 > $OUT/pmc_${wl}_${eng}_p$i.log 2>&1 || echo "pass $wl $eng $i failed"
      echo "pass $wl $eng $i done"
    done
  done
done
echo counters_done

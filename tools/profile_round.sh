#!/bin/bash
# Collects the rocprofv3 evidence DESIGN.md quotes: kernel-trace stats of the default bench command and of the
# wavefront engine, then HBM-traffic and SQ counters in separate --pmc passes (never combined with tracing).
# Usage (on the GPU box, from the repo root): bash tools/profile_round.sh <tag>
set -e
TAG=${1:-r01}
export TMPDIR=/tmp
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/fused -- python bench.py > $OUT/bench_fused.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/wave -- python bench.py --engine wavefront --cpu-spp 0 > $OUT/bench_wave.log 2>&1
B="python bench.py --steps 5 --warmup 1 --cpu-spp 0"
for eng in fused wavefront; do
  for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_INSTS_LDS" "GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU"; do
    n=$(echo $c | cut -d" " -f1)
    timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_${eng}_$n -- $B --engine $eng > $OUT/pmc_${eng}_$n.log 2>&1
  done
done
echo profile_done

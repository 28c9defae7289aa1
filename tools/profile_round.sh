#!/bin/bash
# One GPU-box call that produces what a round commits under profiles/: GPU tests, rocprofv3 kernel-trace summaries of the
# bench (default workload, both engines; C4-10k and C5 wavefront), the PMC passes bench.py's roofline is priced with, and
# then the bench line itself.
# Usage (GPU box, repo root): bash tools/profile_round.sh <tag>      -> gpurun_out/<tag>/...
# Every GPU step has its own time limit and the script stops at the first step that fails.
TAG=${1:-r02}
export TMPDIR=/tmp
OUT=gpurun_out/$TAG
mkdir -p $OUT
set -e
if [ "${SKIP_TESTS:-0}" != 1 ]; then
  timeout -k 10 600 python3 -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $OUT/pytest_gpu.log 2>&1 || { tail -30 $OUT/pytest_gpu.log; exit 1; }
  tail -2 $OUT/pytest_gpu.log
fi
for spec in "default fused" "default wavefront" "c4_10k wavefront" "c5 wavefront"; do
  set -- $spec
  timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_$1_$2 -- python3 bench.py --bare --workload $1 --engine $2 --steps 10 --warmup 2 > $OUT/kt_$1_$2.log 2>&1 || { echo "kernel trace $1 $2 failed"; tail -5 $OUT/kt_$1_$2.log; exit 1; }
  echo "kernel trace $1 $2 done"
done
bash tools/measure_counters.sh $TAG
# the bench line last, priced with the counters just taken on these very sources (on the box's copy of the tree;
# tools/collect_profiles.py puts the same traffic.json under profiles/ at home)
[ -s $OUT/traffic.json ] && cp $OUT/traffic.json profiles/traffic.json
timeout -k 10 400 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err || { tail -20 $OUT/bench.err; exit 1; }
echo bench_done

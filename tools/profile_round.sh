#!/bin/bash
# What a round commits under profiles/, in three GPU-box calls (a call is limited to 20 minutes, and gpurun_out/ comes home after
# each): GPU tests + rocprofv3 kernel traces of the bench (every workload bench.py reports, both engines; 50 timed steps, summarised
# from the launches after the bench's marker only: tools/steady.py); the PMC passes bench.py's roofline is priced with; the bench line.
#   gpurun -- bash tools/profile_round.sh traces <tag>     -> gpurun_out/<tag>/kt_*            (merged home)
#   gpurun -- bash tools/profile_round.sh counters <tag>   -> gpurun_out/<tag>/pmc_*           (merged home)
#   python3 tools/counters_to_json.py gpurun_out/<tag> 8 > profiles/traffic.json               (at home: needs both)
#   gpurun -- bash tools/profile_round.sh bench <tag>      -> gpurun_out/<tag>/bench.json      (priced with that traffic.json)
#   python3 tools/collect_profiles.py <tag> rNN
# WORKLOADS / ENGINES in the environment narrow the first two.  Every GPU step has its own time limit; the script stops at the first that fails.
PHASE=${1:-traces}
TAG=${2:-r03}
export TMPDIR=/tmp
OUT=gpurun_out/$TAG
mkdir -p $OUT
set -e
WORKLOADS=${WORKLOADS:-"default c5 c4_10k c4_100k"}
ENGINES=${ENGINES:-"fused wavefront"}
case $PHASE in
traces)
  if [ "${SKIP_TESTS:-0}" != 1 ]; then
    timeout -k 10 600 python3 -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $OUT/pytest_gpu.log 2>&1 || { tail -30 $OUT/pytest_gpu.log; exit 1; }
    tail -2 $OUT/pytest_gpu.log
  fi
  export PT_BENCH_MARKER=1
  for wl in $WORKLOADS; do
    for eng in $ENGINES; do
      rm -rf $OUT/kt_${wl}_${eng}          # (a directory left by an earlier run would be summarised together with this one)
      timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_${wl}_${eng} -- python3 bench.py --bare --workload $wl --engine $eng --steps 50 --warmup 2 > $OUT/kt_${wl}_${eng}.log 2>&1 || { echo "kernel trace $wl $eng failed"; tail -5 $OUT/kt_${wl}_${eng}.log; exit 1; }
      python3 tools/steady.py $OUT/kt_${wl}_${eng} $OUT/kt_${wl}_${eng}_steady.csv
      grep "^{" $OUT/kt_${wl}_${eng}.log > $OUT/kt_${wl}_${eng}_bench.json || true
      echo "kernel trace $wl $eng done"
    done
  done ;;
counters)
  WORKLOADS="$WORKLOADS" ENGINES="$ENGINES" bash tools/measure_counters.sh $TAG ;;
bench)
  timeout -k 10 600 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err || { tail -20 $OUT/bench.err; exit 1; }
  echo bench_done ;;
esac

#!/bin/bash
# Same-box A/B of library variants (tools/build_variant.sh) on the BVH benches, both engines:
#   VARIANTS="base name ..." [WORKLOADS="c4_10k c4_100k"] [ENGINES="wavefront fused"] bash tools/variant_bench.sh
for v in $VARIANTS; do
  for w in ${WORKLOADS:-c4_10k}; do
  for e in ${ENGINES:-wavefront fused}; do
    echo -n "$v $w $e: "
    PT_HIP_LIBRARY=$PWD/_ab/lib$v.so timeout -k 10 200 python3 bench.py --bare --workload $w --engine $e --steps 5 --warmup 2 2>/dev/null | grep "^{" | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" || exit 1
  done
  done
done

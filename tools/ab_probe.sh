#!/bin/bash
# Same-box A/B of library variants on the extend stage alone: bash tools/ab_probe.sh "<variant names under _ab/>" [triangles] [samples]
# prints camera-ray and bounced-ray extend throughput per variant (tools/sort_probe.py, first two timing lines)
VARIANTS=$1; shift
for v in $VARIANTS; do
  echo "== $v"
  PT_HIP_LIBRARY=$PWD/_ab/lib$v.so timeout -k 10 200 python tools/sort_probe.py "$@" 2>&1 | grep -E "camera rays|pixel order" || exit 1
done

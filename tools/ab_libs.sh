#!/bin/bash
# Same-box A/B of library variants: bash tools/ab_libs.sh "<variant names under _ab/>" <bench_configs args...>
# prints engine, config and Mrays/s per run.  Variants are built by hand into _ab/lib<name>.so (git-ignored).
VARIANTS=$1; shift
for v in $VARIANTS; do
  PT_HIP_LIBRARY=$PWD/_ab/lib$v.so timeout -k 10 200 python tools/bench_configs.py "$@" 2>&1 | grep "^{" | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('$v', d['engine'][:5], d['config'][:28], d.get('samples_per_pass'), d['Mrays_per_s'], d['oracle_bit_exact_on_sampled_pixels'])
" || exit 1
done

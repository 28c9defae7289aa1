#!/bin/bash
# fused engine on BVH scenes: the persistent form (k_fused_bvh) against the lane-per-pixel form (PT_FUSED_BVH_PLAIN=1), same library
for plain in 1 0; do
  if [ $plain = 1 ]; then export PT_FUSED_BVH_PLAIN=1; echo "== lane per pixel (k_fused<*, 2>)"; else unset PT_FUSED_BVH_PLAIN; echo "== persistent (k_fused_bvh)"; fi
  timeout -k 10 300 python tools/seq_bvh.py 2>/dev/null | grep "fused" || exit 1
done

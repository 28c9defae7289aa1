#!/usr/bin/env python3
"""launch_probe.py <strided 0|1> <spp> [reps]: `reps` fused renders of `spp` samples at 1080p (for rocprofv3 --kernel-trace)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pathtracer_amd as pt
strided, spp = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
tr = pt.PathTracer(1920, 1080, device=0)
if strided:
    tr.set_options(engine=pt.PT_ENGINE_FUSED, rng_mode=pt.PT_RNG_STRIDED, stride=35)
else:
    tr.set_options(engine=pt.PT_ENGINE_FUSED)
for _ in range(3): tr.render(spp)
st0 = tr.stats()
t0 = time.perf_counter()
for _ in range(reps): tr.render(spp)
dt = time.perf_counter() - t0
st = tr.stats()
print(f"strided={strided} spp={spp}: events {(st.total_render_ms-st0.total_render_ms)/reps:.3f} ms/render, wall {dt/reps*1e3:.3f} ms", flush=True)
tr.Destroy()

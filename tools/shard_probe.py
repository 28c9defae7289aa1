#!/usr/bin/env python3
"""Device time of a 64-sample step rendered as 1, 2, 4 or 8 launches (the local shards of a sample split), fused engine."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pathtracer_amd as pt

w, h = 1920, 1080
def run(label, per, reps, **opt):
    tr = pt.PathTracer(w, h, device=0)
    tr.set_options(engine=pt.PT_ENGINE_FUSED, rng_mode=pt.PT_RNG_STRIDED, stride=35, **opt)
    for _ in range(3): tr.render(per)
    st0 = tr.stats()
    t0 = time.perf_counter()
    for _ in range(reps): tr.render(per)
    dt = time.perf_counter() - t0
    st = tr.stats()
    print(f"{label:44s} device {1e3*(st.total_render_ms-st0.total_render_ms)/reps/1e3:8.3f} ms/render  wall {dt/reps*1e3:8.3f} ms  rays/render {(st.rays-st0.rays)//reps}", flush=True)
    tr.Destroy()

run("64 spp, one launch", 64, 20)
run("8 spp, one launch", 8, 20)
run("8 spp, one launch, sample_offset 7<<20", 8, 20, sample_offset=7 << 20)
run("1 spp, one launch", 1, 20)
for v in (2, 4, 8):
    run(f"sample split V={v}: {v} launches of {64//v} spp", 64 // v, 20, split_mode=pt.PT_SPLIT_SAMPLE, virtual_shards=v)

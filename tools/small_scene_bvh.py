#!/usr/bin/env python3
"""The all-materials scene (48 triangles, depth 16: BASELINE.json configs[4]) through the BVH instead of the loop over all triangles:
   python3 tools/small_scene_bvh.py <engine: fused|wavefront> <threshold: triangles above which a BVH is packed> [renders]
prints Mrays/s (device time) of `renders` renders of 64 spp at 1080p after the plugin has settled.  Run it under rocprofv3
(--kernel-trace --stats, or --pmc SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU) for the kernels' durations and instruction counts."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pathtracer_amd as pt   # noqa: E402
import scenes                  # noqa: E402

engine = pt.PT_ENGINE_WAVEFRONT if (len(sys.argv) > 1 and sys.argv[1] == "wavefront") else pt.PT_ENGINE_FUSED
thr = int(sys.argv[2]) if len(sys.argv) > 2 else 16
renders = int(sys.argv[3]) if len(sys.argv) > 3 else 5
lib = pt.load_library()
lib.pt_hip_set_bvh_threshold(thr)
depth = 16
with pt.PathTracer(1920, 1080) as tr:
    tr.set_scene(*pt.make_scene(*scenes.all_materials_scene()))
    tr.set_options(engine=engine, depth=depth, samples_per_pass=16, rng_mode=pt.PT_RNG_STRIDED, stride=2 + 33 * depth)
    for _ in range(12):
        tr.render(64)
    s0 = tr.stats()
    for _ in range(renders):
        tr.render(64)
    s1 = tr.stats()
print(f"48 triangles, threshold {thr}, engine {sys.argv[1] if len(sys.argv) > 1 else 'fused'}: {(s1.rays - s0.rays) / (s1.total_render_ms - s0.total_render_ms) / 1e3:.0f} Mrays/s, "
      f"{(s1.total_render_ms - s0.total_render_ms) / renders:.3f} ms per 64 spp, out of step {s1.fused_out_of_step}")

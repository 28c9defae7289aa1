#!/usr/bin/env python3
"""The reference's frame loop -- ONE sample per call (BeginDrawing / Update / FinalizeDrawing per displayed frame): N calls of
render(1) at 1080p, for `rocprofv3 --kernel-trace --stats` to say what a call is made of.
  one_sample_calls.py [calls=200] [engine: fused | wavefront | auto] [n_tris of the synthetic scene, 0 = default scene]"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pathtracer_amd as pt   # noqa: E402
import scenes                  # noqa: E402

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 200
engine = {"fused": pt.PT_ENGINE_FUSED, "wavefront": pt.PT_ENGINE_WAVEFRONT, "auto": pt.PT_ENGINE_AUTO}[sys.argv[2] if len(sys.argv) > 2 else "fused"]
n_tris = int(sys.argv[3]) if len(sys.argv) > 3 else 0
with pt.PathTracer(1920, 1080) as tr:
    if n_tris:
        tr.set_scene(*pt.make_scene(*scenes.synthetic_scene(n_tris)))
    tr.set_options(engine=engine)
    for _ in range(40):              # the plugin's measurement of its candidates: up to 31 renders of this length
        tr.render(1)
    s0 = tr.stats()
    t0 = time.perf_counter()
    for _ in range(calls):
        tr.render(1)
    s1 = tr.stats()
    wall = time.perf_counter() - t0
dev = s1.total_render_ms - s0.total_render_ms
print(f"{calls} calls of one sample: device {dev / calls * 1e3:.1f} us per call, wall {wall / calls * 1e6:.1f} us per call, "
      f"{(s1.rays - s0.rays) / dev / 1e3:.0f} Mrays/s (device time)")

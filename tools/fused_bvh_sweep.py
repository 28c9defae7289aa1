#!/usr/bin/env python3
"""fused engine, BVH scenes, reference (sequential) streams, 1080p: Mrays/s at 1 and 16 samples per render call"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pathtracer_amd as pt, scenes
tag = sys.argv[1] if len(sys.argv) > 1 else ""
for n in (10000, 100000):
    sc = pt.make_scene(*scenes.synthetic_scene(n))
    row = []
    for (w, h, spp, reps) in ((1920, 1080, 1, 8), (1920, 1080, 16, 2), (320, 180, 16, 4)):
        with pt.PathTracer(w, h) as tr:
            tr.set_scene(*sc)
            tr.set_options(engine=pt.PT_ENGINE_FUSED)
            tr.render(spp)
            s0 = tr.stats()
            for _ in range(reps): tr.render(spp)
            s1 = tr.stats()
        row.append(f"{w}x{h} x{spp}: {(s1.rays - s0.rays) / (s1.total_render_ms - s0.total_render_ms) / 1e3:7.1f}")
    print(f"{tag:22s} {n:6d} triangles  " + "   ".join(row), flush=True)

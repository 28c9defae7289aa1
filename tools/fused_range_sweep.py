#!/usr/bin/env python3
"""k_fused_bvh (the fused engine's persistent form on BVH scenes): pixels per block (PT_FUSED_RANGE; 0 = the launcher's choice)
against scene size and samples per call, reference streams, 1080p; Mrays/s (device time).
0 = the launcher's own choice: interleaved tiles over the resident blocks (fused_bvh_range in pt_fused_bvh.hpp)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pathtracer_amd as pt   # noqa: E402
import scenes                  # noqa: E402

ranges = [int(r) for r in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0, 256, 512, 832, 1664]
for n in (300, 972, 10000, 100000):
    sc = pt.make_scene(*scenes.synthetic_scene(n))
    for spp in (1, 4, 16):
        row = []
        for rng in ranges:
            os.environ.pop("PT_FUSED_RANGE", None)
            if rng:
                os.environ["PT_FUSED_RANGE"] = str(rng)
            with pt.PathTracer(1920, 1080) as tr:
                tr.set_scene(*sc)
                tr.set_options(engine=pt.PT_ENGINE_FUSED)
                for _ in range(3):
                    tr.render(spp)
                s0 = tr.stats()
                for _ in range(3):
                    tr.render(spp)
                s1 = tr.stats()
                row.append(f"{rng:5d}: {(s1.rays - s0.rays) / (s1.total_render_ms - s0.total_render_ms) / 1e3:8.1f}")
        print(f"{n:6d} triangles x{spp:<2d}  " + "   ".join(row), flush=True)

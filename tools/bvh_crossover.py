#!/usr/bin/env python3
"""Where does the BVH overtake brute force?  (kBvhMinTris in pt_device.hpp)  Scenes of 48 .. 972 triangles at 1080p, packed
under a threshold above / below their size, PT_ENGINE_AUTO after its measurement renders; Mrays/s (device time)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import pathtracer_amd as pt   # noqa: E402
import scenes                  # noqa: E402

lib = pt.load_library()
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 16
for n in (48, 108, 192, 300, 432, 588, 972):
    sc = scenes.all_materials_scene() if n == 48 else scenes.synthetic_scene(n)
    depth = 16 if n == 48 else 10
    for strided in (False, True):
        row = []
        for thr in (0xFFFFFFFF, 16):
            lib.pt_hip_set_bvh_threshold(thr)
            with pt.PathTracer(1920, 1080) as tr:
                tr.set_scene(*pt.make_scene(*sc))
                kw = dict(rng_mode=pt.PT_RNG_STRIDED, stride=2 + 33 * depth, samples_per_pass=16) if strided else {}
                tr.set_options(engine=pt.PT_ENGINE_AUTO, depth=depth, **kw)
                for _ in range(12 if spp > 4 else 40):        # the plugin's own measurement: one unmeasured render, then two per candidate (at most five; six each of short renders)
                    tr.render(spp)
                s0 = tr.stats()
                for _ in range(2):
                    tr.render(spp)
                s1 = tr.stats()
                pick = "wavefront" if s1.engine_in_use == pt.PT_ENGINE_WAVEFRONT else ("fused oos" if s1.fused_out_of_step else "fused")
                row.append(f"{(s1.rays - s0.rays) / (s1.total_render_ms - s0.total_render_ms) / 1e3:8.0f} ({pick:9s})")
        print(f"{len(sc[1]):4d} triangles {'strided   ' if strided else 'sequential'} x{spp}   brute force {row[0]}   BVH {row[1]}", flush=True)
lib.pt_hip_set_bvh_threshold(pt.PT_BVH_DEFAULT_MIN_TRIANGLES)

#!/usr/bin/env python3
"""fused engine on brute-force scenes at 1080p, reference streams: samples in step (k_fused) against out of step
(PT_FUSED_OUT_OF_STEP in the environment / PT_FRAME_OUT_OF_STEP), Mrays/s by samples per call; a bit-exactness check of the two"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pathtracer_amd as pt, scenes
cases = [("default", None, 10, 0), ("all materials depth 16", scenes.all_materials_scene(), 16, 0), ("mirror, extended, depth 16", scenes.mirror_scene(), 16, 1),
         ("synthetic 192 (brute force)", scenes.synthetic_scene(192), 10, 0)]
for name, sc, depth, shading in cases:
    imgs = []
    for oos in (0, 1):
        if oos: os.environ["PT_FUSED_OUT_OF_STEP"] = "1"
        else: os.environ.pop("PT_FUSED_OUT_OF_STEP", None)
        row = []
        for spp in (1, 8, 64):
            with pt.PathTracer(1920, 1080) as tr:
                if sc is not None: tr.set_scene(*pt.make_scene(*sc))
                tr.set_options(engine=pt.PT_ENGINE_FUSED, depth=depth, shading=shading)
                tr.render(spp)
                s0 = tr.stats()
                for _ in range(3): tr.render(spp)
                s1 = tr.stats()
                if spp == 8: imgs.append(tr.readback())
            row.append(f"x{spp}: {(s1.rays - s0.rays) / (s1.total_render_ms - s0.total_render_ms) / 1e3:9.1f}")
        print(f"{name:30s} {'out of step' if oos else 'in step    '}  " + "   ".join(row), flush=True)
    print(f"{name:30s} images bit-identical: {np.array_equal(imgs[0].view(np.uint32), imgs[1].view(np.uint32))}", flush=True)
os.environ.pop("PT_FUSED_OUT_OF_STEP", None)

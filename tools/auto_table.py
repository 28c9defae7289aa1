#!/usr/bin/env python3
"""What PT_ENGINE_AUTO picks, and what it costs against the best fixed choice: scene x streams x samples per call at 1080p.
Rates in Mrays/s (device time) after the measurement renders."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pathtracer_amd as pt, scenes

def rate(sc, depth, shading, kw, engine, spp, force_oos=None):
    if force_oos: os.environ["PT_FUSED_OUT_OF_STEP"] = "1"
    try:
        with pt.PathTracer(1920, 1080) as tr:
            if sc is not None: tr.set_scene(*sc)
            tr.set_options(engine=engine, depth=depth, shading=shading, **kw)
            for _ in range(12 if spp > 4 else 40): tr.render(spp)        # (short renders are measured six times per candidate)
            s0 = tr.stats()
            for _ in range(3): tr.render(spp)
            s1 = tr.stats()
            return (s1.rays - s0.rays) / (s1.total_render_ms - s0.total_render_ms) / 1e3, s1.engine_in_use, s1.fused_out_of_step
    finally:
        os.environ.pop("PT_FUSED_OUT_OF_STEP", None)

cases = [("default", None, 10, 0), ("all materials d16", pt.make_scene(*scenes.all_materials_scene()), 16, 0),
         ("mirror extended d16", pt.make_scene(*scenes.mirror_scene()), 16, 1), ("synthetic 10000", pt.make_scene(*scenes.synthetic_scene(10000)), 10, 0)]
for name, sc, depth, shading in cases:
    for strided in (False, True):
        stride = 2 + (2 if shading else 33) * depth
        kw = dict(rng_mode=pt.PT_RNG_STRIDED, stride=stride, samples_per_pass=16) if strided else {}
        for spp in (1, 64):
            a, eng, oos = rate(sc, depth, shading, kw, pt.PT_ENGINE_AUTO, spp)
            w, _, _ = rate(sc, depth, shading, kw, pt.PT_ENGINE_WAVEFRONT, spp)
            f, _, foos = rate(sc, depth, shading, kw, pt.PT_ENGINE_FUSED, spp)
            pick = "wavefront" if eng == pt.PT_ENGINE_WAVEFRONT else ("fused out of step" if oos else "fused")
            print(f"{name:20s} {'strided' if strided else 'sequential':10s} x{spp:<3d} auto {a:9.1f} ({pick:17s})  wavefront {w:9.1f}  fused {f:9.1f} ({'out of step' if foos else 'in step'})  auto/best {a / max(w, f):.2f}", flush=True)

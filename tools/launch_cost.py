#!/usr/bin/env python3
"""Device time of one render call against the samples it takes (default scene, 1080p): what a launch costs before its first
sample -- the per-step price of splitting a step's samples over several launches (sample split: one launch per local shard)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pathtracer_amd as pt   # noqa: E402

for engine, name in ((pt.PT_ENGINE_FUSED, "fused"), (pt.PT_ENGINE_WAVEFRONT, "wavefront")):
    for strided in (False, True):
        row = []
        for spp in (1, 2, 4, 8, 16, 32, 64):
            with pt.PathTracer(1920, 1080) as tr:
                kw = dict(rng_mode=pt.PT_RNG_STRIDED, stride=35) if strided else {}
                tr.set_options(engine=engine, samples_per_pass=min(spp, 16) if strided else 64, **kw)
                for _ in range(3):
                    tr.render(spp)
                s0 = tr.stats()
                for _ in range(10):
                    tr.render(spp)
                s1 = tr.stats()
            row.append(f"{spp}: {(s1.total_render_ms - s0.total_render_ms) / 10:.3f}")
        print(f"{name:9s} {'strided/35' if strided else 'sequential':10s} ms per call by samples  " + "  ".join(row), flush=True)

#!/usr/bin/env python3
"""Render a synthetic scene once (for rocprof): python tools/run_scene.py <n_tris> <engine> <spp> [w h]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pathtracer_amd as pt   # noqa: E402
import scenes                  # noqa: E402

n, engine, spp = int(sys.argv[1]), sys.argv[2], int(sys.argv[3])
w, h = (int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (1920, 1080)
sc = scenes.all_materials_scene() if n == 0 else scenes.synthetic_scene(n)
with pt.PathTracer(w, h) as tr:
    tr.set_scene(*pt.make_scene(*sc))
    tr.set_options(engine=pt.PT_ENGINE_FUSED if engine == "fused" else pt.PT_ENGINE_WAVEFRONT)
    tr.render(spp)
    st = tr.stats()
    print("rays", st.rays, "ms", st.total_render_ms, "Mrays/s", st.rays / st.total_render_ms / 1e3, "per bounce", list(st.rays_per_bounce)[:10])

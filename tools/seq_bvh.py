"""BVH scenes under the reference's own sequential streams: fused / wavefront / PT_ENGINE_AUTO, 8 samples per call, Mrays/s (device time)."""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import pathtracer_amd as pt, scenes
for n in (1000, 10000, 100000):
    sc = pt.make_scene(*scenes.synthetic_scene(n))
    for (w, h) in ((1920, 1080), (320, 180)):
        for engine, name in ((pt.PT_ENGINE_FUSED, "fused"), (pt.PT_ENGINE_WAVEFRONT, "wavefront"), (pt.PT_ENGINE_AUTO, "auto")):
            with pt.PathTracer(w, h) as tr:
                tr.set_scene(*sc)
                tr.set_options(engine=engine)
                for _ in range(8): tr.render(8)          # (PT_ENGINE_AUTO measures its candidates on the first renders of a sample count: one unmeasured, two per candidate)
                s0 = tr.stats(); tr.render(8); s1 = tr.stats()
                print(f"{n} triangles {w}x{h} sequential {name:9s}: {(s1.rays - s0.rays) / (s1.total_render_ms - s0.total_render_ms) / 1e3:8.1f} Mrays/s  in use {s1.engine_in_use}", flush=True)

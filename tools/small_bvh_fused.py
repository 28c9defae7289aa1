"""BVH scenes of 108 .. 2500 triangles under the reference streams, 16 samples per call: the fused engine's persistent form, its
lane-per-pixel form (PT_FUSED_BVH_PLAIN) and the wavefront engine; Mrays/s (device time)."""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import pathtracer_amd as pt, scenes
lib = pt.load_library()
lib.pt_hip_set_bvh_threshold(16)
for n in (108, 192, 300, 432, 588, 972, 2500):
    sc = pt.make_scene(*scenes.synthetic_scene(n))
    row = []
    for engine, name, env in ((pt.PT_ENGINE_FUSED, "fused persistent", {}), (pt.PT_ENGINE_FUSED, "fused plain", {"PT_FUSED_BVH_PLAIN": "1"}), (pt.PT_ENGINE_WAVEFRONT, "wavefront", {})):
        os.environ.pop("PT_FUSED_BVH_PLAIN", None)
        os.environ.update(env)
        with pt.PathTracer(1920, 1080) as tr:
            tr.set_scene(*sc)
            tr.set_options(engine=engine)
            for _ in range(3): tr.render(16)
            s0 = tr.stats(); tr.render(16); s1 = tr.stats()
            row.append(f"{name} {(s1.rays - s0.rays) / (s1.total_render_ms - s0.total_render_ms) / 1e3:8.1f} ({(s1.rays - s0.rays) / (1920 * 1080 * 16):.2f} rays/sample)")
    print(f"{n} triangles sequential x16: " + "   ".join(row), flush=True)

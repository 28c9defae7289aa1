import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import pathtracer_amd as pt, scenes
lib = pt.load_library()
for n in (48, 100, 200, 400, 1000, 3000):
    sc = scenes.all_materials_scene() if n == 48 else scenes.synthetic_scene(n)
    row = []
    for thr in (0xFFFFFFFF, 16):
        lib.pt_hip_set_bvh_threshold(thr)
        for eng in (pt.PT_ENGINE_FUSED, pt.PT_ENGINE_WAVEFRONT):
            with pt.PathTracer(1920, 1080) as tr:
                tr.set_scene(*pt.make_scene(*sc))
                tr.set_options(engine=eng, rng_mode=pt.PT_RNG_STRIDED, stride=332, samples_per_pass=8)
                tr.render(1)
                tr.set_options(engine=eng)
                tr.render(8)
                st = tr.stats()
                row.append(round(st.rays / st.total_render_ms / 1e3, 1))
    print(len(sc[1]), "tris  Mrays/s  brute fused/wave:", row[0], row[1], " bvh fused/wave:", row[2], row[3], flush=True)
lib.pt_hip_set_bvh_threshold(512)

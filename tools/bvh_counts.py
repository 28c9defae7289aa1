#!/usr/bin/env python3
"""Work counters of the BVH traversal (debug build: -DPT_BVH_COUNT, library given by PT_HIP_LIBRARY):
node steps and triangle tests per ray, lane-level and wave-level, camera rays vs all rays."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pathtracer_amd as pt   # noqa: E402
import scenes                  # noqa: E402

# scene: a triangle count (scenes.synthetic_scene) or mixed:<room triangles>:<ball facets asked for> (test_bvh_adversarial.room_with_a_ball)
spec = sys.argv[1] if len(sys.argv) > 1 else "10000"
engine = pt.PT_ENGINE_WAVEFRONT if (len(sys.argv) < 3 or sys.argv[2] == "wavefront") else pt.PT_ENGINE_FUSED
if spec.startswith("mixed:"):
    from test_bvh_adversarial import room_with_a_ball
    V, T, M = room_with_a_ball(int(spec.split(":")[1]), int(spec.split(":")[2]), 8)
else:
    V, T, M = scenes.synthetic_scene(int(spec))
lib = pt.load_library()
out = (C.c_ulonglong * 8)()
for depth in (1, 10):
    with pt.PathTracer(1920, 1080) as tr:
        tr.set_scene(*pt.make_scene(V, T, M))
        tr.set_options(engine=engine, depth=depth, samples_per_pass=4, rng_mode=pt.PT_RNG_STRIDED, stride=2 + 33 * depth)
        lib.pt_hip_debug_bvh_counts(None, 1)
        tr.render(4)
        st = tr.stats()
        lib.pt_hip_debug_bvh_counts(out, 0)
    rays = st.rays
    print(f"triangles {len(T)} depth {depth}: rays {rays}  node steps/ray {out[0] / rays:.1f}  node trips per wave-of-64-rays {out[1] / (rays / 64):.1f} "
          f"(lanes busy {out[0] / max(1, out[1]):.1f})  triangle tests/ray {out[2] / rays:.1f}  triangle trips per 64 rays {out[3] / (rays / 64):.1f} "
          f"(lanes busy {out[2] / max(1, out[3]):.1f})  wave cycles per 64 rays: node walk {out[4] / (rays / 64):.0f}, leaf tests {out[5] / (rays / 64):.0f}, "
          f"write-back + refill {out[6] / (rays / 64):.0f}")

#!/usr/bin/env python3
"""Where the lanes of a wave are during a brute-force triangle test (debug build: bash tools/build_variant.sh tricount -DPT_TRI_COUNT,
library given by PT_HIP_LIBRARY): lanes that enter the determinant / u / v / t part of RayIntersectsTriangle (Functions.cuh:56-89)
and the wave-level trips through each part, all-materials scene (BASELINE.json configs[4]), wavefront engine, strided streams."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pathtracer_amd as pt   # noqa: E402
import scenes                  # noqa: E402

lib = pt.load_library()
out = (C.c_ulonglong * 8)()
depth = 16
for name, eng in (("wavefront", pt.PT_ENGINE_WAVEFRONT), ("fused", pt.PT_ENGINE_FUSED)):
    with pt.PathTracer(1920, 1080) as tr:
        tr.set_scene(*pt.make_scene(*scenes.all_materials_scene()))
        tr.set_options(engine=eng, depth=depth, samples_per_pass=4, rng_mode=pt.PT_RNG_STRIDED, stride=2 + 33 * depth)
        lib.pt_hip_debug_tri_counts(None, 1)
        tr.render(4)
        st = tr.stats()
        lib.pt_hip_debug_tri_counts(out, 0)
    lanes, trips = list(out[0:4]), list(out[4:8])
    # instructions of each part (wave64 VALU, from the ISA of ray_triangle): determinant 15, reciprocal + s + u 16, q + v 17, t + ranking 11
    weight = (15, 16, 17, 11)
    busy = sum(l * w for l, w in zip(lanes, weight)) / max(1, sum(t * w for t, w in zip(trips, weight)))
    print(f"{name}: rays {st.rays}; per part (det, u, v, t): lanes per trip " + " / ".join(f"{l / max(1, t):.1f}" for l, t in zip(lanes, trips))
          + "; trips per first-part trip " + " / ".join(f"{t / max(1, trips[0]):.3f}" for t in trips) + f"; lanes per instruction of the test, weighted: {busy:.1f} of 64")

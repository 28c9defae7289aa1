#!/usr/bin/env python3
"""Generates tests/golden/*.npz with the CPU oracle (oracle/pt_oracle.c).

The reference ships no fixtures, so these vectors pin OUR oracle's output:
they guard the oracle against regressions and give the GPU tests a second, run-independent
anchor.  The oracle itself is pinned by the hand-derived known answers in test_oracle_kat.py.
Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import pt_oracle_py as po   # noqa: E402
import scenes                            # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))

CASES = {
    # name: (scene fn, width, height, spp, depth, rng_mode, stride, camera kwargs[, shading])
    "default_64x48_4spp": ("default", 64, 48, 4, 10, po.RNG_SEQUENTIAL, 35, {}),
    "default_160x120_2spp": ("default", 160, 120, 2, 10, po.RNG_SEQUENTIAL, 35, {}),
    "default_33x17_3spp_d3": ("default", 33, 17, 3, 3, po.RNG_SEQUENTIAL, 35, {}),
    "allmat_96x64_3spp_d16": ("allmat", 96, 64, 3, 16, po.RNG_SEQUENTIAL, 35, {}),
    "allmat_48x32_2spp_d10_strided332": ("allmat", 48, 32, 2, 10, po.RNG_STRIDED, 332, {}),
    "synthetic200_64x48_2spp": ("synthetic200", 64, 48, 2, 10, po.RNG_SEQUENTIAL, 35, {}),
    "default_rotated_64x48_2spp": ("default", 64, 48, 2, 10, po.RNG_SEQUENTIAL, 35,
                                   {"position": (0.05, -0.02, 0.1, 1.0), "rotation": (0.0, 0.08715574, 0.0, 0.9961947), "fov": 70.0}),
    # PT_SHADING_EXTENDED (no reference counterpart, the oracle IS the definition)
    "extended_mirror_96x64_3spp_d16": ("mirror", 96, 64, 3, 16, po.RNG_SEQUENTIAL, 35, {}, po.SHADING_EXTENDED),
    "extended_default_64x48_4spp_strided40": ("default", 64, 48, 4, 10, po.RNG_STRIDED, 40, {}, po.SHADING_EXTENDED),
}


def scene_of(name):
    if name == "default":
        return scenes.default_scene()
    if name == "allmat":
        return scenes.all_materials_scene()
    if name == "mirror":
        return scenes.mirror_scene()
    if name == "synthetic200":
        return scenes.synthetic_scene(200)
    raise KeyError(name)


def run_case(case):
    scene_name, w, h, spp, depth, mode, stride, cam_kw = CASES[case][:8]
    shading = CASES[case][8] if len(CASES[case]) > 8 else po.SHADING_REFERENCE
    V, T, M = scene_of(scene_name)
    cam = po.make_camera(**cam_kw) if cam_kw else po.default_camera()
    p = po.params(w, h, cam=cam, depth=depth, rng_mode=mode, stride=stride, spp=spp, threads=0, shading=shading)
    mean, rng, st, last = po.render(p, scene=po.make_scene(V, T, M), want_last_frame=True)
    return dict(mean=mean, rng=rng if rng is not None else np.zeros(0, np.uint32), last_frame=last,
                rays=np.uint64(st.rays), draws=np.uint64(st.draws),
                rays_per_bounce=np.array(list(st.rays_per_bounce), dtype=np.uint64))


def main():
    for case in CASES:
        d = run_case(case)
        np.savez_compressed(os.path.join(OUT, case + ".npz"), **d)
        print(case, "rays", int(d["rays"]), "draws", int(d["draws"]), "per bounce", [int(x) for x in d["rays_per_bounce"] if x])
    # sample-split definition: 8 samples in 4 shards, strided/35, default scene
    p = po.params(64, 48, rng_mode=po.RNG_STRIDED, stride=35, spp=8)
    mean, st = po.render_sharded(p, 4)
    np.savez_compressed(os.path.join(OUT, "default_64x48_8spp_4shards.npz"), mean=mean, rays=np.uint64(st.rays))
    print("sharded rays", st.rays)


if __name__ == "__main__":
    main()

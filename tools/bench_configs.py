#!/usr/bin/env python3
"""Measures the other BASELINE.json configs (they are parity-test cases, not bench.py lines):
C5 all-materials depth 16 and C4 synthetic scenes, both engines, 1080p; spot-checks a few thousand
pixels of every run against the CPU oracle (bit-exact) and prints one JSON line per run."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import pathtracer_amd as pt           # noqa: E402
import scenes                          # noqa: E402
from oracle import pt_oracle_py as po  # noqa: E402


def run(name, scene, w, h, spp, depth, engine, check_pixels=2000, samples_per_pass=16, shading=0, sequential=False):
    V, T, M = scene
    # strided RNG so that the wavefront engine batches samples; the stride covers the worst case of the shading in use
    # (reference: 2 + 33 draws per bounce; extended: 2 + 2 per bounce).  The fused engine keeps the reference's own
    # sequential streams under reference shading.  Each run is checked against the oracle in its own mode.
    strided = (bool(shading) or engine == "wavefront") and not sequential
    stride = 2 + (2 if shading else 33) * depth
    with pt.PathTracer(w, h) as tr:
        tr.set_scene(*pt.make_scene(V, T, M))
        tr.set_options(engine=pt.PT_ENGINE_WAVEFRONT if engine == "wavefront" else pt.PT_ENGINE_FUSED, depth=depth,
                       samples_per_pass=samples_per_pass, shading=shading,
                       rng_mode=pt.PT_RNG_STRIDED if strided else pt.PT_RNG_SEQUENTIAL, stride=stride)
        for _ in range(12):             # warm-up, and the plugin's measurement of its candidates (decisions survive the restart below)
            tr.render(spp)
        tr.set_options(depth=depth)     # restart accumulation
        t0 = time.perf_counter()
        tr.render(spp)
        dt = time.perf_counter() - t0
        st = tr.stats()
        img = tr.readback()
    ok = None
    if check_pixels:
        ids = np.unique(np.random.default_rng(3).integers(0, w * h, check_pixels)).astype(np.uint32)
        ref, _, _, _ = po.render(po.params(w, h, depth=depth, spp=spp, shading=shading, stride=stride,
                                           rng_mode=po.RNG_STRIDED if strided else po.RNG_SEQUENTIAL), scene=po.make_scene(V, T, M), pixel_ids=ids)
        ok = bool(np.array_equal(img[ids].view(np.uint32), ref.view(np.uint32)))
    print(json.dumps({"config": name, "engine": engine, "samples_per_pass": samples_per_pass, "triangles": len(T), "size": f"{w}x{h}", "spp": spp, "depth": depth, "rng": f"strided/{stride}" if strided else "sequential",
                      "rays": int(st.rays), "rays_per_pixel_sample": round(st.rays / (w * h * spp), 3),
                      "device_ms": round(st.total_render_ms, 3), "wall_ms": round(dt * 1e3, 3),
                      "Mrays_per_s": round(st.rays / (st.total_render_ms * 1e-3) / 1e6, 1),
                      "oracle_bit_exact_on_sampled_pixels": ok}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--which", default="c5,c4small")
    ap.add_argument("--spp", type=int, default=64)
    args = ap.parse_args()
    which = args.which.split(",")
    if "c5" in which:
        for eng in ("fused", "wavefront"):
            run("C5 all materials depth 16", scenes.all_materials_scene(), 1920, 1080, args.spp, 16, eng)
    if "seq" in which:
        # the wavefront engine on the reference's own sequential streams: samples_per_pass = samples taken per launch
        for per in (1, 16, 64):
            run("default scene, sequential streams", scenes.default_scene(), 1920, 1080, args.spp, 10, "wavefront", samples_per_pass=per, sequential=True)
        for per in (1, 64):
            run("C5 all materials depth 16, sequential streams", scenes.all_materials_scene(), 1920, 1080, args.spp, 16, "wavefront", samples_per_pass=per, sequential=True)
        run("C5 all materials depth 16, sequential streams", scenes.all_materials_scene(), 1920, 1080, args.spp, 16, "fused", sequential=True)
    if "seq64" in which:
        run("default scene, sequential streams", scenes.default_scene(), 1920, 1080, args.spp, 10, "wavefront", samples_per_pass=64, sequential=True)
        run("C5 all materials depth 16, sequential streams", scenes.all_materials_scene(), 1920, 1080, args.spp, 16, "wavefront", samples_per_pass=64, sequential=True)
        run("C5 mirror, extended shading, sequential", scenes.mirror_scene(), 1920, 1080, args.spp, 12, "wavefront", samples_per_pass=64, sequential=True, shading=1)
    if "c3" in which:
        # BASELINE configs[2], one rank's share: 3840x2160, samples [3*128, 4*128) of the 1024-sample frame (rank 3 of 8),
        # strided RNG stride 35 = the reference's own stream on this scene
        w, h, spp, rank = 3840, 2160, 128, 3
        for eng in ("fused", "wavefront"):
            with pt.PathTracer(w, h) as tr:
                tr.set_options(engine=pt.PT_ENGINE_FUSED if eng == "fused" else pt.PT_ENGINE_WAVEFRONT, rng_mode=pt.PT_RNG_STRIDED, stride=35,
                               samples_per_pass=8, sample_offset=rank * spp)
                tr.render(1)
                tr.set_options(sample_offset=rank * spp)
                t0 = time.perf_counter()
                tr.render(spp)
                dt = time.perf_counter() - t0
                st = tr.stats()
                img = tr.readback()
            ids = np.unique(np.random.default_rng(5).integers(0, w * h, 1500)).astype(np.uint32)
            # the oracle renders the same 128 samples: RNG index frame_first = rank*spp, mean index from 0
            ref = np.ones((len(ids), 4), dtype=np.float32)
            for s_ in range(spp):
                _, _, _, frame = po.render(po.params(w, h, spp=1, rng_mode=po.RNG_STRIDED, stride=35, frame_first=rank * spp + s_), pixel_ids=ids, want_last_frame=True)
                r = np.float32(1.0) / np.float32(s_ + 1)
                ref = (frame * r + (ref * np.float32(s_)) * r).astype(np.float32)
            ok = bool(np.array_equal(img[ids].view(np.uint32), ref.view(np.uint32)))
            print(json.dumps({"config": "C3 share of rank 3/8: 3840x2160, 128 of 1024 spp", "engine": eng, "rays": int(st.rays), "device_ms": round(st.total_render_ms, 3),
                              "wall_ms": round(dt * 1e3, 3), "Mrays_per_s": round(st.rays / (st.total_render_ms * 1e-3) / 1e6, 1),
                              "oracle_bit_exact_on_sampled_pixels": ok}), flush=True)
    if "c5x" in which:
        for eng in ("fused", "wavefront"):
            run("C5 mirror scene, EXTENDED shading, depth 16", scenes.mirror_scene(), 1920, 1080, args.spp, 16, eng, shading=1)
    if "c4small" in which:
        for n in (1000, 10000):
            sc = scenes.synthetic_scene(n)
            for eng in ("fused", "wavefront"):
                run(f"C4 synthetic {n}", sc, 1920, 1080, 4 if n >= 10000 else 16, 10, eng, check_pixels=300)
    if "c4_10k" in which:
        run("C4 synthetic 10000", scenes.synthetic_scene(10000), 1920, 1080, args.spp, 10, "wavefront", check_pixels=100)
    if "c4big" in which:
        for n in (10000, 30000, 100000):
            sc = scenes.synthetic_scene(n)
            run(f"C4 synthetic {n}", sc, 1920, 1080, args.spp, 10, "wavefront", check_pixels=100)


if __name__ == "__main__":
    main()

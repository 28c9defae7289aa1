"""The fused engine's BVH form (pathtracer_amd/csrc/pt_fused_bvh.hpp: persistent walks, finished rays serviced at a quorum,
pixels handed out from a block counter, walks that start at the leaf a ray leaves) on the CPU, block by block as 256 host
threads (tests/host/fused_bvh_check.cpp, ASan + UBSan), against the oracle: image, RNG streams and ray counts bit for bit."""
import os
import subprocess

import numpy as np
import pytest

import scenes
from test_bvh_host import pack

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHECK = os.path.join(ROOT, "tests", "host", "fused_bvh_check")
needs_binary = pytest.mark.skipif(not os.path.exists(CHECK), reason="tests/host/fused_bvh_check is not built (__graft_entry__.build())")


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def run(pt, oracle, tmp_path, scene, w, h, spp, depth=10, cam=None, rng_mode=0, stride=35, shading=0, frame_first=0, mean_first=0,
        pixel_first=0, pixel_count=None, pixels_per_block="i3", lanes=64):
    V, T, M = scene
    c = oracle.camera_corners(cam or oracle.default_camera(), np.float32(w) / np.float32(h))
    f4 = lambda v: pt.Float4(v.x, v.y, v.z, v.w)
    n = w * h - pixel_first if pixel_count is None else pixel_count
    fr = pt.Frame(width=w, height=h, pixel_first=pixel_first, pixel_count=n, depth=depth, seed=1337, rng_mode=rng_mode, stride=stride,
                  shading=shading, flags=pt.PT_FRAME_SCENE_BVH, origin=f4(c[0]), top_left=f4(c[1]), bottom_left=f4(c[2]), bottom_right=f4(c[3]))
    blob, frame, out = tmp_path / "scene.blob", tmp_path / "frame.bin", tmp_path / "out.bin"
    blob.write_bytes(pack(pt, V, T, M))
    frame.write_bytes(bytes(fr))
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0")
    env.pop("LD_PRELOAD", None)
    r = subprocess.run([CHECK, str(blob), str(frame), str(frame_first), str(mean_first), str(spp), str(pixels_per_block), str(out), str(lanes)],
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    raw = np.fromfile(out, dtype=np.uint8)
    return raw[:16 * n].view(np.float32).reshape(n, 4), raw[16 * n:].view(np.uint32), int(r.stdout.split()[5])


@needs_binary
@pytest.mark.parametrize("n_tris,w,h,spp,per_block,lanes", [(700, 48, 30, 3, 1024, 64), (2500, 40, 24, 2, 256, 64), (700, 37, 19, 4, 300, 64),
                                                            (1500, 33, 21, 2, 512, 37), (700, 48, 30, 3, "i2", 64), (1500, 37, 19, 2, "i5", 64),
                                                            (700, 33, 21, 2, "i40", 64), (2500, 40, 24, 2, "i3", 37)])
def test_reference_streams(pt, oracle, tmp_path, n_tris, w, h, spp, per_block, lanes):
    """sequential streams (a pixel's samples in order in one lane); blocks of 256..1024 contiguous pixels, a range that is no
    multiple of the block, 37-lane waves; iG = what the launcher does: interleaved 64-pixel tiles over G blocks
    (fused_bvh_range) -- a short last tile (48x30 = 22.5 tiles, 37x19 = 10.98), more blocks than tiles (33x21 = 11 tiles over 40)"""
    V, T, M = scenes.synthetic_scene(n_tris)
    ref, ref_rng, st, _ = oracle.render(oracle.params(w, h, spp=spp), scene=oracle.make_scene(V, T, M))
    img, rng, rays = run(pt, oracle, tmp_path, (V, T, M), w, h, spp, pixels_per_block=per_block, lanes=lanes)
    assert np.array_equal(bits(img), bits(ref)) and np.array_equal(rng, ref_rng) and rays == st.rays


@needs_binary
def test_strided_streams_on_a_tile_with_a_moved_camera(pt, oracle, tmp_path):
    V, T, M = scenes.synthetic_scene(1000)
    cam = oracle.make_camera(position=(0.1, -0.05, 0.1, 1.0), rotation=(0.02, 0.04, 0.0, 1.0), fov=80.0)
    w, h, spp, depth, first, count = 40, 30, 3, 6, 123, 700
    stride = 2 + 33 * depth
    p = oracle.params(w, h, cam=cam, depth=depth, spp=spp, rng_mode=oracle.RNG_STRIDED, stride=stride, frame_first=4)
    ref, _, st, _ = oracle.render(p, scene=oracle.make_scene(V, T, M), pixel_first=first, n=count)
    img, _, rays = run(pt, oracle, tmp_path, (V, T, M), w, h, spp, depth=depth, cam=cam, rng_mode=1, stride=stride, frame_first=4, mean_first=4,
                       pixel_first=first, pixel_count=count, pixels_per_block=512)
    assert np.array_equal(bits(img), bits(ref)) and rays == st.rays


@needs_binary
def test_extended_shading_and_a_soup(pt, oracle, tmp_path):
    V, T, M = scenes.synthetic_scene(1500)
    M = [(m[0], m[1], 0.8) for m in M]
    w, h, spp, depth = 36, 22, 2, 8
    ref, ref_rng, st, _ = oracle.render(oracle.params(w, h, depth=depth, spp=spp, shading=oracle.SHADING_EXTENDED), scene=oracle.make_scene(V, T, M))
    img, rng, rays = run(pt, oracle, tmp_path, (V, T, M), w, h, spp, depth=depth, shading=1)
    assert np.array_equal(bits(img), bits(ref)) and np.array_equal(rng, ref_rng) and rays == st.rays
    V, T, M = scenes.triangle_soup(300, 31)       # (300: every box overlaps most others, so each of the 256 lane threads walks most of the tree)
    ref, ref_rng, st, _ = oracle.render(oracle.params(w, h, depth=depth, spp=spp), scene=oracle.make_scene(V, T, M))
    img, rng, rays = run(pt, oracle, tmp_path, (V, T, M), w, h, spp, depth=depth)
    assert np.array_equal(bits(img), bits(ref)) and np.array_equal(rng, ref_rng) and rays == st.rays


@needs_binary
@pytest.mark.parametrize("scene,w,h,spp,depth,shading,strided", [("default", 50, 30, 3, 10, 0, False), ("materials", 45, 27, 4, 16, 0, False),
                                                                 ("mirror", 40, 25, 3, 12, 1, False), ("materials", 37, 23, 3, 8, 0, True),
                                                                 ("synthetic300", 24, 16, 2, 10, 0, False)])
def test_brute_force_scenes_out_of_step(pt, oracle, tmp_path, scene, w, h, spp, depth, shading, strided):
    """fused_samples_out_of_step (k_fused<*, *, true>, PT_FRAME_OUT_OF_STEP): a wave's 64 pixels take their samples out of step;
    image, streams and ray counts are the oracle's.  A ragged last wave (pixel counts that are no multiple of 64)."""
    V, T, M = {"default": scenes.default_scene, "materials": scenes.all_materials_scene, "mirror": scenes.mirror_scene,
               "synthetic300": lambda: scenes.synthetic_scene(300)}[scene]()
    stride = 2 + (2 if shading else 33) * depth
    okw = dict(rng_mode=oracle.RNG_STRIDED, stride=stride, frame_first=2) if strided else {}
    ref, ref_rng, st, _ = oracle.render(oracle.params(w, h, depth=depth, spp=spp, shading=shading, **okw), scene=oracle.make_scene(V, T, M))
    kw = dict(rng_mode=1, stride=stride, frame_first=2, mean_first=2) if strided else {}
    img, rng, rays = run(pt, oracle, tmp_path, (V, T, M), w, h, spp, depth=depth, shading=shading, **kw)
    assert np.array_equal(bits(img), bits(ref)) and rays == st.rays
    if not strided:
        assert np.array_equal(rng, ref_rng)

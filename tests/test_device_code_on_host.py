"""The device code's per-pixel recurrence against the oracle, bit for bit, without a GPU.

tests/host/fused_host_check.cpp compiles pathtracer_amd/csrc/pt_device.hpp with g++ through a stand-in
hip_runtime.h (one lane, ASan + UBSan) and drives it with a loop that mirrors k_fused.  Everything the kernels
compute per pixel is the code under test here -- camera rays (IEEE and verified-fast division forms), the
camera-ray records, brute-force and BVH intersection, both shadings, the running mean, strided stream
addressing -- only the launch geometry, LDS staging and wave-level scheduling are left to the GPU tests.
(v_rcp_f32 / v_sqrt_f32 are exact divisions / square roots in the stand-in; that the short forms built on the
hardware instructions return the same bits is what tests/proofs/exact_rcp_sqrt.hip proves on the GPU.)"""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import scenes
from test_bvh_host import pack

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHECK = os.path.join(ROOT, "tests", "host", "fused_host_check")
BLOCK_CHECK = os.path.join(ROOT, "tests", "host", "bounces_block_check")


def run_host(pt, oracle, tmp_path, scene, w, h, spp, depth=10, cam=None, rng_mode=0, stride=35, shading=0, flags=0,
             frame_first=0, pixel_first=0, pixel_count=None, wavefront=None):
    """wavefront = (samples_per_pass, max_blocks): the one-launch wavefront pass, block by block, instead of the fused loop"""
    V, T, M = scene
    c = oracle.camera_corners(cam or oracle.default_camera(), np.float32(w) / np.float32(h))
    f4 = lambda v: pt.Float4(v.x, v.y, v.z, v.w)
    n = w * h - pixel_first if pixel_count is None else pixel_count
    fr = pt.Frame(width=w, height=h, pixel_first=pixel_first, pixel_count=n, depth=depth, seed=1337, rng_mode=rng_mode, stride=stride,
                  shading=shading, flags=flags, origin=f4(c[0]), top_left=f4(c[1]), bottom_left=f4(c[2]), bottom_right=f4(c[3]))
    blob, frame, out = tmp_path / "scene.blob", tmp_path / "frame.bin", tmp_path / "out.bin"
    blob.write_bytes(pack(pt, V, T, M))
    frame.write_bytes(bytes(fr))
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0")
    env.pop("LD_PRELOAD", None)
    if wavefront:
        cmd = [BLOCK_CHECK, str(blob), str(frame), str(frame_first), str(spp), str(wavefront[0]), str(wavefront[1]), str(out)]
    else:
        cmd = [CHECK, str(blob), str(frame), str(frame_first), str(frame_first), str(spp), str(out)]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    raw = np.fromfile(out, dtype=np.uint8)
    img = raw[:16 * n].view(np.float32).reshape(n, 4)
    rng = raw[16 * n:].view(np.uint32)
    rays = int(r.stdout.split()[5])
    return img, rng, rays


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


needs_binary = pytest.mark.skipif(not os.path.exists(CHECK), reason="tests/host/fused_host_check is not built (__graft_entry__.build())")


@needs_binary
@pytest.mark.parametrize("w,h,spp,flags", [(64, 48, 2, 0), (160, 120, 1, 0), (96, 54, 3, 1)])
def test_default_scene(pt, oracle, tmp_path, w, h, spp, flags):
    """flags = 1 (PT_FRAME_FAST_DIV): the multiply + two residual corrections in place of the two IEEE divisions of
    the ray set-up; on the GPU it is enabled only for sizes pt_hip_verify_fast_div has checked exhaustively."""
    ref, ref_rng, st, _ = oracle.render(oracle.params(w, h, spp=spp))
    img, rng, rays = run_host(pt, oracle, tmp_path, scenes.default_scene(), w, h, spp, flags=flags)
    assert np.array_equal(bits(img), bits(ref)) and np.array_equal(rng, ref_rng) and rays == st.rays


@needs_binary
def test_all_materials_depth_16_and_moved_camera(pt, oracle, tmp_path):
    V, T, M = scenes.all_materials_scene()
    cam = oracle.make_camera(position=(0.1, 0.05, -0.2, 1.0), rotation=(0.05, -0.12, 0.02, 0.99), fov=75.0)
    w, h, spp = 80, 45, 3
    ref, ref_rng, st, _ = oracle.render(oracle.params(w, h, cam=cam, depth=16, spp=spp), scene=oracle.make_scene(V, T, M))
    assert st.rays_per_bounce[3] > 0
    img, rng, rays = run_host(pt, oracle, tmp_path, (V, T, M), w, h, spp, depth=16, cam=cam)
    assert np.array_equal(bits(img), bits(ref)) and np.array_equal(rng, ref_rng) and rays == st.rays


@needs_binary
@pytest.mark.parametrize("n_tris", [192, 300, 700, 2500])
def test_synthetic_scenes_brute_force_and_bvh(pt, oracle, tmp_path, n_tris):
    """300 triangles: the loop over all triangles with the camera-ray records; 700 and 2500: the BVH walk."""
    V, T, M = scenes.synthetic_scene(n_tris)
    w, h, spp = 48, 30, 2
    ref, ref_rng, st, _ = oracle.render(oracle.params(w, h, spp=spp), scene=oracle.make_scene(V, T, M))
    img, rng, rays = run_host(pt, oracle, tmp_path, (V, T, M), w, h, spp)
    assert np.array_equal(bits(img), bits(ref)) and np.array_equal(rng, ref_rng) and rays == st.rays


@needs_binary
def test_strided_streams_and_a_tile(pt, oracle, tmp_path):
    """STRIDED addressing from a later sample on (frame_first = 5); the stride (400) differs from what a sample draws, so
    every sample re-addresses its stream; pixels 1000..2999 only (tile split).  The oracle numbers the running mean from
    frame_first as well (an image of ones, as the plugin's, weighted 5 : 3 here)."""
    w, h, spp = 96, 64, 3
    V, T, M = scenes.all_materials_scene()
    p = oracle.params(w, h, depth=12, spp=spp, rng_mode=oracle.RNG_STRIDED, stride=400, frame_first=5)
    ref, _, st, _ = oracle.render(p, scene=oracle.make_scene(V, T, M), pixel_first=1000, n=2000)
    img, _, rays = run_host(pt, oracle, tmp_path, (V, T, M), w, h, spp, depth=12, rng_mode=1, stride=400, frame_first=5, pixel_first=1000, pixel_count=2000)
    assert np.array_equal(bits(img), bits(ref)) and rays == st.rays


@needs_binary
@pytest.mark.parametrize("scene", ["mirror", "synthetic"])
def test_extended_shading(pt, oracle, tmp_path, scene):
    V, T, M = scenes.mirror_scene() if scene == "mirror" else scenes.synthetic_scene(1500)
    if scene == "synthetic":
        M = [(m[0], m[1], 0.8) for m in M]
    w, h, spp, depth = 64, 36, 2, 12
    p = oracle.params(w, h, depth=depth, spp=spp, rng_mode=oracle.RNG_STRIDED, stride=2 + 2 * depth, shading=oracle.SHADING_EXTENDED)
    ref, _, st, _ = oracle.render(p, scene=oracle.make_scene(V, T, M))
    img, _, rays = run_host(pt, oracle, tmp_path, (V, T, M), w, h, spp, depth=depth, rng_mode=1, stride=2 + 2 * depth, shading=1)
    assert np.array_equal(bits(img), bits(ref)) and rays == st.rays


# ---- the wavefront engine's one-launch pass (pt_wavefront.hpp, bounces_block) as blocks of 256 host threads ----
needs_block_binary = pytest.mark.skipif(not os.path.exists(BLOCK_CHECK), reason="tests/host/bounces_block_check is not built (__graft_entry__.build())")


@needs_block_binary
@pytest.mark.parametrize("w,h,spp,per_pass,max_blocks", [(64, 40, 3, 1, 64), (75, 41, 5, 2, 4), (128, 72, 6, 4, 3)])
def test_wavefront_pass_default_scene(pt, oracle, tmp_path, w, h, spp, per_pass, max_blocks):
    """75 x 41 = 3075 pixels: a ragged last chunk; 4 and 3 blocks: every block owns several chunks; 6 samples in passes
    of 4: a short last pass.  Strided streams (stride 35 = the default scene's own draw count) and, with one sample per
    pass, the reference's sequential streams."""
    mode = dict(rng_mode=oracle.RNG_SEQUENTIAL) if per_pass == 1 else dict(rng_mode=oracle.RNG_STRIDED, stride=35)
    ref, ref_rng, st, _ = oracle.render(oracle.params(w, h, spp=spp, **mode))
    img, rng, rays = run_host(pt, oracle, tmp_path, scenes.default_scene(), w, h, spp, rng_mode=0 if per_pass == 1 else 1, stride=35,
                              wavefront=(per_pass, max_blocks))
    assert np.array_equal(bits(img), bits(ref)) and rays == st.rays
    if per_pass == 1:
        assert np.array_equal(rng, ref_rng)


@needs_block_binary
@pytest.mark.parametrize("scene,w,h,spp,per_launch,max_blocks,depth,shading", [
    ("default", 75, 41, 7, 3, 4, 10, 0),          # every sample ends where it is made: one round per launch
    ("materials", 70, 44, 6, 6, 5, 16, 0),        # samples that bounce on: a pixel's stream waits for its queued sample
    ("materials", 33, 20, 5, 2, 1, 16, 0),        # one block owns every chunk; launches of 2, 2, 1
    ("mirror", 48, 30, 4, 4, 2, 12, 1),           # extended shading: nearly every sample is queued -> a round per sample
    ("soup", 60, 37, 5, 5, 3, 8, 0),
])
def test_wavefront_pass_sequential_streams_in_rounds(pt, oracle, tmp_path, scene, w, h, spp, per_launch, max_blocks, depth, shading):
    """SEQUENTIAL streams (the reference's own: sample s+1 of a pixel starts where sample s stopped), several samples
    per launch: bounces_block<true> makes a pixel's samples in a row while they end at once and runs a round of the
    bounce loop for each that does not.  Image, final stream words and ray count against the oracle."""
    cam = None
    if scene == "default":
        V, T, M = scenes.default_scene()
    elif scene == "materials":
        V, T, M = scenes.all_materials_scene()
    elif scene == "mirror":
        V, T, M = scenes.mirror_scene()
    else:
        V, T, M = scenes.triangle_soup(40, 5)
        cam = oracle.make_camera(position=(0.05, -0.03, -0.1, 1.0), rotation=(0.03, 0.06, -0.02, 0.995), fov=70.0)
    p = oracle.params(w, h, cam=cam, depth=depth, spp=spp, shading=shading)
    ref, ref_rng, st, _ = oracle.render(p, scene=oracle.make_scene(V, T, M))
    img, rng, rays = run_host(pt, oracle, tmp_path, (V, T, M), w, h, spp, depth=depth, cam=cam, shading=shading, wavefront=(per_launch, max_blocks))
    assert np.array_equal(bits(img), bits(ref)) and np.array_equal(rng, ref_rng) and rays == st.rays


@needs_block_binary
def test_wavefront_pass_paths_that_bounce_on(pt, oracle, tmp_path):
    """all materials, depth 16: paths survive several bounces, so the block-private queues, their ping-pong and the
    producer-side resolution of terminal hits are all in play"""
    V, T, M = scenes.all_materials_scene()
    w, h, spp, depth = 70, 44, 4, 16
    p = oracle.params(w, h, depth=depth, spp=spp, rng_mode=oracle.RNG_STRIDED, stride=2 + 33 * depth)
    ref, _, st, _ = oracle.render(p, scene=oracle.make_scene(V, T, M))
    assert st.rays_per_bounce[3] > 0
    img, _, rays = run_host(pt, oracle, tmp_path, (V, T, M), w, h, spp, depth=depth, rng_mode=1, stride=2 + 33 * depth, wavefront=(4, 5))
    assert np.array_equal(bits(img), bits(ref)) and rays == st.rays


@needs_block_binary
def test_wavefront_pass_extended_shading_on_a_tile(pt, oracle, tmp_path):
    V, T, M = scenes.mirror_scene()
    w, h, spp, depth = 64, 36, 4, 12
    p = oracle.params(w, h, depth=depth, spp=spp, rng_mode=oracle.RNG_STRIDED, stride=2 + 2 * depth, shading=oracle.SHADING_EXTENDED)
    ref, _, st, _ = oracle.render(p, scene=oracle.make_scene(V, T, M), pixel_first=300, n=1500)
    img, _, rays = run_host(pt, oracle, tmp_path, (V, T, M), w, h, spp, depth=depth, rng_mode=1, stride=2 + 2 * depth, shading=1,
                            pixel_first=300, pixel_count=1500, wavefront=(2, 2))
    assert np.array_equal(bits(img), bits(ref)) and rays == st.rays


@needs_block_binary
def test_wavefront_pass_at_the_depth_limit(pt, oracle, tmp_path):
    """PT_MAX_DEPTH = 32 bounce iterations on the mirror scene (extended shading keeps most paths alive to the end): the
    per-bounce counter rows, the queue ping-pong and the last-bounce handling at their limit."""
    V, T, M = scenes.mirror_scene()
    w, h, spp, depth = 40, 24, 3, 32
    p = oracle.params(w, h, depth=depth, spp=spp, rng_mode=oracle.RNG_STRIDED, stride=2 + 2 * depth, shading=oracle.SHADING_EXTENDED)
    ref, _, st, _ = oracle.render(p, scene=oracle.make_scene(V, T, M))
    assert st.rays_per_bounce[31] > 0
    img, _, rays = run_host(pt, oracle, tmp_path, (V, T, M), w, h, spp, depth=depth, rng_mode=1, stride=2 + 2 * depth, shading=1, wavefront=(3, 2))
    assert np.array_equal(bits(img), bits(ref)) and rays == st.rays


@needs_block_binary
@pytest.mark.parametrize("w,h", [(1, 1), (257, 1), (3, 200)])
def test_wavefront_pass_degenerate_image_shapes(pt, oracle, tmp_path, w, h):
    spp = 5
    ref, _, st, _ = oracle.render(oracle.params(w, h, spp=spp, rng_mode=oracle.RNG_STRIDED, stride=35))
    img, _, rays = run_host(pt, oracle, tmp_path, scenes.default_scene(), w, h, spp, rng_mode=1, stride=35, wavefront=(2, 8))
    assert np.array_equal(bits(img), bits(ref)) and rays == st.rays


@needs_binary
@pytest.mark.parametrize("n_tris,seed", [(7, 1), (40, 2), (64, 3), (90, 4)])
def test_triangle_soup(pt, oracle, tmp_path, n_tris, seed):
    """Random triangles of arbitrary orientation and size (tests/scenes.py triangle_soup), camera moved and turned: the
    per-pixel camera-ray culling (primary_need_mask -- one lane here, so every pixel culls for itself, the strictest
    form) must never drop a triangle the reference's test would accept.  64 triangles: the mask's last bit; 90: no mask."""
    V, T, M = scenes.triangle_soup(n_tris, seed)
    cam = oracle.make_camera(position=(0.05, -0.03, -0.1, 1.0), rotation=(0.03, 0.06, -0.02, 0.995), fov=70.0)
    w, h, spp = 96, 60, 2
    ref, ref_rng, st, _ = oracle.render(oracle.params(w, h, cam=cam, depth=8, spp=spp), scene=oracle.make_scene(V, T, M))
    img, rng, rays = run_host(pt, oracle, tmp_path, (V, T, M), w, h, spp, depth=8, cam=cam)
    assert np.array_equal(bits(img), bits(ref)) and np.array_equal(rng, ref_rng) and rays == st.rays
    assert st.rays > w * h * spp + 20               # the soup is in view: some camera rays hit something diffuse and bounce


@needs_binary
def test_triangle_soup_many_seeds(pt, oracle, tmp_path):
    """twenty more soups of 5..60 triangles, a different camera each, small images"""
    for seed in range(10, 30):
        V, T, M = scenes.triangle_soup(5 + (seed * 7) % 56, seed)
        cam = oracle.make_camera(position=(0.02 * (seed % 5) - 0.04, 0.01 * (seed % 3), -0.05 * (seed % 4), 1.0),
                                 rotation=(0.01 * (seed % 7) - 0.03, 0.02 * (seed % 5) - 0.04, 0.0, 1.0), fov=60.0 + seed)
        w, h = 40 + seed, 25
        ref, ref_rng, st, _ = oracle.render(oracle.params(w, h, cam=cam, depth=6, spp=1), scene=oracle.make_scene(V, T, M))
        img, rng, rays = run_host(pt, oracle, tmp_path, (V, T, M), w, h, 1, depth=6, cam=cam)
        assert np.array_equal(bits(img), bits(ref)) and np.array_equal(rng, ref_rng) and rays == st.rays, f"seed {seed}"


@needs_block_binary
def test_wavefront_pass_triangle_soup(pt, oracle, tmp_path):
    V, T, M = scenes.triangle_soup(40, 5)
    cam = oracle.make_camera(position=(0.05, -0.03, -0.1, 1.0), rotation=(0.03, 0.06, -0.02, 0.995), fov=70.0)
    w, h, spp, depth = 60, 37, 4, 8
    p = oracle.params(w, h, cam=cam, depth=depth, spp=spp, rng_mode=oracle.RNG_STRIDED, stride=2 + 33 * depth)
    ref, _, st, _ = oracle.render(p, scene=oracle.make_scene(V, T, M))
    img, _, rays = run_host(pt, oracle, tmp_path, (V, T, M), w, h, spp, depth=depth, cam=cam, rng_mode=1, stride=2 + 33 * depth, wavefront=(2, 3))
    assert np.array_equal(bits(img), bits(ref)) and rays == st.rays


def pixel_corner_scene(oracle, w, h, n_tris, seed):
    """Triangles whose vertices lie exactly on the rays through PIXEL CORNERS (at random depths), a third of them with two
    vertices on one pixel row or column: their edges run along pixel boundaries, so for the pixels beside them the
    rejection tests of RayIntersectsTriangle are zero (to rounding) at the patch corners -- the borderline of
    primary_need_mask's margin rule ('argued, covered only by random soups')."""
    rng = np.random.default_rng(seed)
    c = oracle.camera_corners(oracle.default_camera(), np.float32(w) / np.float32(h))
    o, tl, bl, br = (np.array([v.x, v.y, v.z], dtype=np.float64) for v in c)
    V, T = [], []
    for k in range(n_tris):
        ij = rng.integers(0, [w + 1, h + 1], (3, 2))
        if k % 3 == 0: ij[1, 1] = ij[0, 1]              # an edge along a pixel row
        if k % 3 == 1: ij[2, 0] = ij[0, 0]              # an edge along a pixel column
        if len({tuple(x) for x in ij}) < 3:
            ij[2] = (ij[2] + [3, 5]) % [w + 1, h + 1]
        base = len(V)
        for i, j in ij:
            p = tl + (br - bl) * (i / w) + (bl - tl) * (j / h)
            s = rng.uniform(0.6, 2.5)
            q = o + (p - o) * s
            V.append((float(np.float32(q[0])), float(np.float32(q[1])), float(np.float32(q[2])), 1.0))
        T.append((base, base + 1, base + 2, int(rng.integers(0, 3))))
        T.append((base + 2, base + 1, base, int(rng.integers(0, 3))))      # both windings: one of the two faces the camera
    M = [(0, (0.8, 0.7, 0.6, 1.0)), (2, (6.0, 5.0, 4.0, 1.0)), (0, (0.3, 0.5, 0.9, 1.0))]
    return V, T, M


@needs_binary
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_triangles_with_edges_on_pixel_boundaries(pt, oracle, tmp_path, seed):
    w, h, spp = 64, 40, 4
    V, T, M = pixel_corner_scene(oracle, w, h, 30, seed)          # 60 triangles: inside the 64-bit need mask
    ref, ref_rng, st, _ = oracle.render(oracle.params(w, h, depth=6, spp=spp), scene=oracle.make_scene(V, T, M))
    img, rng, rays = run_host(pt, oracle, tmp_path, (V, T, M), w, h, spp, depth=6)
    assert np.array_equal(bits(img), bits(ref)) and np.array_equal(rng, ref_rng) and rays == st.rays
    assert st.rays > w * h * spp * 1.2

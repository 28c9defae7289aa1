"""GPU parity: the HIP plugin, called through its C-ABI, against the CPU oracle -- bit for bit.

Tolerance: none.  Every comparison is on the uint32 view of the float32 images (and on the raw
RNG words); a single differing bit fails."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ENGINES = [("fused", 1), ("wavefront", 0)]


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def assert_bit_equal(img, ref, what):
    diff = (bits(img) != bits(ref)).any(axis=1)
    assert not diff.any(), f"{what}: {int(diff.sum())} of {len(diff)} pixels differ; first at pixel {int(np.argmax(diff))}: {img[np.argmax(diff)]} vs {ref[np.argmax(diff)]}"


@pytest.mark.parametrize("name,engine", ENGINES)
@pytest.mark.parametrize("w,h,spp", [(64, 48, 1), (160, 120, 4), (133, 71, 3), (640, 480, 1)])
def test_default_scene_sequential(pt, oracle, name, engine, w, h, spp):
    ref, ref_rng, ref_stats, _ = oracle.render(oracle.params(w, h, spp=spp))
    with pt.PathTracer(w, h) as tr:
        tr.set_options(engine=engine)
        tr.render(spp)
        img, rng, st = tr.readback(), tr.readback_rng(), tr.stats()
    assert_bit_equal(img, ref, f"{name} {w}x{h}x{spp}")
    assert np.array_equal(rng, ref_rng)
    assert st.rays == ref_stats.rays == 2 * w * h * spp
    assert st.frames == spp


@pytest.mark.parametrize("name,engine,mode", [("fused", 1, "sequential"), ("wavefront", 0, "strided")])
def test_headline_config_every_pixel(pt, oracle, name, engine, mode):
    """BASELINE.json configs[1] at full size -- 1920x1080, 64 spp, depth 10, seed 1337 -- as bench.py runs it (the fused
    engine on the reference's sequential streams, the wavefront engine on strided/35 in passes of 16): all 2 073 600 pixels
    against the oracle (on the box's host threads, ~2 s), bit for bit, plus the ray count 2 x pixels x samples."""
    w, h, spp = 1920, 1080, 64
    strided = mode == "strided"
    ref, _, ref_stats, _ = oracle.render(oracle.params(w, h, spp=spp, rng_mode=oracle.RNG_STRIDED if strided else oracle.RNG_SEQUENTIAL, stride=35))
    with pt.PathTracer(w, h) as tr:
        tr.set_options(engine=engine, rng_mode=pt.PT_RNG_STRIDED if strided else pt.PT_RNG_SEQUENTIAL, stride=35, samples_per_pass=16)
        tr.render(spp)
        img, st = tr.readback(), tr.stats()
    assert_bit_equal(img, ref, f"{name} 1920x1080x64")
    assert st.rays == ref_stats.rays == 2 * w * h * spp


@pytest.mark.parametrize("name,engine", ENGINES)
def test_4k_share_of_one_rank_every_pixel(pt, oracle, name, engine):
    """BASELINE.json configs[2], the share of rank 3 of 8 at full image size: 3840x2160, strided/35, samples 384..399 of the
    1024-sample frame (16 of the rank's 128, to keep the oracle's side at a few seconds): all 8 294 400 pixels, bit for bit."""
    w, h, spp, first = 3840, 2160, 16, 3 * 128
    ones = np.ones((w * h, 4), dtype=np.float32)
    ref = ones
    # the oracle numbers the running mean from frame_first; the plugin's shard starts its own mean at 0: fold frame by frame
    for s_ in range(spp):
        _, _, _, frame = oracle.render(oracle.params(w, h, spp=1, rng_mode=oracle.RNG_STRIDED, stride=35, frame_first=first + s_), want_last_frame=True)
        r = np.float32(1.0) / np.float32(s_ + 1)
        ref = (frame * r + (ref * np.float32(s_)) * r).astype(np.float32)
    with pt.PathTracer(w, h) as tr:
        tr.set_options(engine=engine, rng_mode=pt.PT_RNG_STRIDED, stride=35, samples_per_pass=8, sample_offset=first)
        tr.render(spp)
        img, st = tr.readback(), tr.stats()
    assert_bit_equal(img, ref, f"{name} 3840x2160 samples {first}..{first + spp - 1}")
    assert st.rays == 2 * w * h * spp


@pytest.mark.parametrize("name,engine", ENGINES)
@pytest.mark.parametrize("n_tris,seed", [(7, 1), (40, 2), (64, 3), (90, 4), (33, 21), (58, 22)])
def test_triangle_soup(pt, oracle, name, engine, n_tris, seed):
    """Random triangles of arbitrary orientation and size, camera moved and turned (tests/scenes.py triangle_soup): the
    camera-ray culling (primary_need_mask, per wave of 64 pixels here) must never drop a triangle the reference's test
    accepts; 64 triangles: the mask's last bit, 90: no mask.  Every pixel, both engines, strided batches for the wavefront."""
    V, T, M = scenes.triangle_soup(n_tris, seed)
    cam = dict(position=(0.05, -0.03, -0.1, 1.0), rotation=(0.03, 0.06, -0.02, 0.995), fov=70.0)
    w, h, spp, depth = 320, 200, 4, 8
    stride = 2 + 33 * depth
    ref, _, st_ref, _ = oracle.render(oracle.params(w, h, cam=oracle.make_camera(**cam), depth=depth, spp=spp, rng_mode=oracle.RNG_STRIDED, stride=stride),
                                      scene=oracle.make_scene(V, T, M))
    with pt.PathTracer(w, h, camera=pt.CameraData(cam["position"], cam["rotation"], cam["fov"])) as tr:
        tr.set_scene(*pt.make_scene(V, T, M))
        tr.set_options(engine=engine, depth=depth, rng_mode=pt.PT_RNG_STRIDED, stride=stride, samples_per_pass=2)
        tr.render(spp)
        img, st = tr.readback(), tr.stats()
    assert_bit_equal(img, ref, f"{name} soup {n_tris}/{seed}")
    assert st.rays == st_ref.rays


@pytest.mark.parametrize("name,engine", ENGINES)
@pytest.mark.parametrize("n_tris,seed", [(900, 31), (3000, 32)])
def test_triangle_soup_through_the_bvh(pt, oracle, name, engine, n_tris, seed):
    """the same kind of scene above the BVH threshold: triangles of every orientation, from slivers to a third of the view,
    crossing each other -- boxes with large error inflations next to tight ones; every pixel against the (brute-force) oracle"""
    V, T, M = scenes.triangle_soup(n_tris, seed)
    cam = dict(position=(0.05, -0.03, -0.1, 1.0), rotation=(0.03, 0.06, -0.02, 0.995), fov=70.0)
    w, h, spp, depth = 160, 100, 2, 8
    ref, ref_rng, st_ref, _ = oracle.render(oracle.params(w, h, cam=oracle.make_camera(**cam), depth=depth, spp=spp), scene=oracle.make_scene(V, T, M))
    with pt.PathTracer(w, h, camera=pt.CameraData(cam["position"], cam["rotation"], cam["fov"])) as tr:
        tr.set_scene(*pt.make_scene(V, T, M))
        tr.set_options(engine=engine, depth=depth)
        tr.render(spp)
        img, rng, st = tr.readback(), tr.readback_rng(), tr.stats()
    assert_bit_equal(img, ref, f"{name} BVH soup {n_tris}")
    assert np.array_equal(rng, ref_rng) and st.rays == st_ref.rays


@pytest.mark.parametrize("name,engine", ENGINES)
@pytest.mark.parametrize("w,h", [(1, 1), (1, 7), (257, 1), (3, 1000), (255, 3), (4099, 2)])
def test_degenerate_image_shapes(pt, oracle, name, engine, w, h):
    """one pixel, one column, one row just over a block, sizes around the 256-pixel chunk and a prime width: sequential
    streams, then strided ones in passes of 2 (a short last pass)"""
    for strided in (False, True):
        spp = 5
        ref, _, st_ref, _ = oracle.render(oracle.params(w, h, spp=spp, rng_mode=oracle.RNG_STRIDED if strided else oracle.RNG_SEQUENTIAL, stride=35))
        with pt.PathTracer(w, h) as tr:
            tr.set_options(engine=engine, rng_mode=pt.PT_RNG_STRIDED if strided else pt.PT_RNG_SEQUENTIAL, stride=35, samples_per_pass=2)
            tr.render(spp)
            img, st = tr.readback(), tr.stats()
        assert_bit_equal(img, ref, f"{name} {w}x{h} strided={strided}")
        assert st.rays == st_ref.rays


@pytest.mark.parametrize("name,engine", ENGINES)
@pytest.mark.parametrize("shading", [0, 1])
def test_depth_limit(pt, oracle, name, engine, shading):
    """PT_MAX_DEPTH = 32 bounce iterations: all materials under reference shading (strided/1058 so that the wavefront engine
    batches), the mirror scene under extended shading (paths alive at bounce 31)."""
    V, T, M = scenes.mirror_scene() if shading else scenes.all_materials_scene()
    w, h, spp, depth = 120, 68, 5, 32
    stride = 2 + (2 if shading else 33) * depth
    ref, _, st_ref, _ = oracle.render(oracle.params(w, h, depth=depth, spp=spp, rng_mode=oracle.RNG_STRIDED, stride=stride, shading=shading),
                                      scene=oracle.make_scene(V, T, M))
    if shading:
        assert st_ref.rays_per_bounce[31] > 0
    with pt.PathTracer(w, h) as tr:
        tr.set_scene(*pt.make_scene(V, T, M))
        tr.set_options(engine=engine, depth=depth, rng_mode=pt.PT_RNG_STRIDED, stride=stride, samples_per_pass=3, shading=shading)
        tr.render(spp)
        img, st = tr.readback(), tr.stats()
    assert_bit_equal(img, ref, f"{name} depth 32 shading {shading}")
    assert st.rays == st_ref.rays


@pytest.mark.parametrize("name,engine", ENGINES)
def test_begin_finalize_equals_batched(pt, oracle, name, engine):
    w, h, spp = 96, 64, 5
    ref, _, _, _ = oracle.render(oracle.params(w, h, spp=spp))
    with pt.PathTracer(w, h) as tr:
        tr.set_options(engine=engine)
        for _ in range(spp):
            tr.BeginDrawing()
            tr.Update(pt.CameraData())
            tr.FinalizeDrawing()
        img = tr.readback()
    assert_bit_equal(img, ref, name)


@pytest.mark.parametrize("name,engine", ENGINES)
def test_strided_35_equals_sequential(pt, oracle, name, engine):
    w, h, spp = 128, 72, 6
    ref, _, _, _ = oracle.render(oracle.params(w, h, spp=spp))
    with pt.PathTracer(w, h) as tr:
        tr.set_options(engine=engine, rng_mode=pt.PT_RNG_STRIDED, stride=35, samples_per_pass=4)
        tr.render(spp)
        img = tr.readback()
    assert_bit_equal(img, ref, name)


# ---------------------------------------------------------------- golden fixtures, other scenes, cameras
import os
import subprocess
import sys

import scenes

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import make_golden  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def _pt_scene(pt, name):
    return pt.make_scene(*make_golden.scene_of(name))


@pytest.mark.parametrize("name,engine", ENGINES)
@pytest.mark.parametrize("case", sorted(make_golden.CASES))
def test_gpu_reproduces_golden_vectors(pt, name, engine, case):
    """The committed vectors (tests/golden, made by the oracle) -- no oracle code runs here."""
    scene_name, w, h, spp, depth, mode, stride, cam_kw = make_golden.CASES[case][:8]
    shading = make_golden.CASES[case][8] if len(make_golden.CASES[case]) > 8 else 0
    want = np.load(os.path.join(GOLDEN, case + ".npz"))
    cam = pt.CameraData(cam_kw.get("position", (0, 0, 0, 1)), cam_kw.get("rotation", (0, 0, 0, 1)), cam_kw.get("fov", 90.0))
    with pt.PathTracer(w, h, camera=cam) as tr:
        tr.set_scene(*_pt_scene(pt, scene_name))
        tr.set_options(engine=engine, depth=depth, rng_mode=mode, stride=stride, samples_per_pass=2, shading=shading)
        tr.render(spp)
        img, st = tr.readback(), tr.stats()
        rng = tr.readback_rng()
    assert_bit_equal(img, want["mean"], f"{case} {name}")
    assert st.rays == int(want["rays"])
    if shading == 0:
        assert 2 * st.paths + 33 * st.diffuse_bounces == int(want["draws"])
    elif engine == 1:       # EXTENDED: 2 draws per cosine-sampled bounce, none per mirror bounce (fused counts them apart)
        assert 2 * st.paths + 2 * st.diffuse_bounces == int(want["draws"])
    if mode == 0:
        assert np.array_equal(rng, want["rng"])
    if engine == 0:
        assert list(st.rays_per_bounce) == [int(x) for x in want["rays_per_bounce"]]


@pytest.mark.parametrize("name,engine", ENGINES)
def test_all_materials_depth16(pt, oracle, name, engine):
    """BASELINE.json configs[4] at test size: DIFFUSE + EMISSIVE + REFLECTIVE(absorbing), depth 16."""
    V, T, M = scenes.all_materials_scene()
    w, h, spp = 160, 90, 4
    ref, ref_rng, st_ref, _ = oracle.render(oracle.params(w, h, depth=16, spp=spp), scene=oracle.make_scene(V, T, M))
    assert st_ref.rays_per_bounce[4] > 0            # paths really do bounce
    with pt.PathTracer(w, h) as tr:
        tr.set_scene(*pt.make_scene(V, T, M))
        tr.set_options(engine=engine, depth=16)
        tr.render(spp)
        img, rng, st = tr.readback(), tr.readback_rng(), tr.stats()
    assert_bit_equal(img, ref, name)
    assert np.array_equal(rng, ref_rng) and st.rays == st_ref.rays


@pytest.mark.parametrize("scene,shading,depth,per_launch", [("materials", 0, 16, 24), ("materials", 0, 16, 7), ("mirror", 1, 12, 24)])
def test_wavefront_sequential_streams_in_rounds(pt, oracle, scene, shading, depth, per_launch):
    """The wavefront engine on the reference's own streams with many samples per launch (bounces_block<true>: a pixel's
    samples in a row while they end where they are made, a round of the bounce loop for each that is queued).  24 samples in
    launches of 24 and of 7, 7, 7, 3; every pixel, the final stream words and the per-bounce ray counts against the oracle."""
    V, T, M = scenes.all_materials_scene() if scene == "materials" else scenes.mirror_scene()
    w, h, spp = 320, 180, 24
    ref, ref_rng, st_ref, _ = oracle.render(oracle.params(w, h, depth=depth, spp=spp, shading=shading), scene=oracle.make_scene(V, T, M))
    with pt.PathTracer(w, h) as tr:
        tr.set_scene(*pt.make_scene(V, T, M))
        tr.set_options(engine=0, depth=depth, shading=shading, samples_per_pass=per_launch)
        tr.render(spp)
        img, rng, st = tr.readback(), tr.readback_rng(), tr.stats()
    assert_bit_equal(img, ref, f"{scene} in launches of {per_launch}")
    assert np.array_equal(rng, ref_rng) and st.rays == st_ref.rays
    assert list(st.rays_per_bounce)[:depth] == [int(x) for x in st_ref.rays_per_bounce[:depth]]


@pytest.fixture
def brute_force_only(pt):
    """Scenes packed inside this fixture carry no BVH (reference-style loop over all triangles)."""
    lib = pt.load_library()
    lib.pt_hip_set_bvh_threshold(0xFFFFFFFF)
    yield
    lib.pt_hip_set_bvh_threshold(pt.PT_BVH_DEFAULT_MIN_TRIANGLES)


def _synthetic_case(pt, oracle, name, engine, n_tris):
    V, T, M = scenes.synthetic_scene(n_tris)
    w, h, spp = 64, 40, 2
    ref, ref_rng, st_ref, _ = oracle.render(oracle.params(w, h, spp=spp), scene=oracle.make_scene(V, T, M))
    with pt.PathTracer(w, h) as tr:
        tr.set_scene(*pt.make_scene(V, T, M))
        tr.set_options(engine=engine)
        tr.render(spp)
        img, rng, st = tr.readback(), tr.readback_rng(), tr.stats()
    assert_bit_equal(img, ref, f"{name} {len(T)} triangles")
    assert np.array_equal(rng, ref_rng) and st.rays == st_ref.rays


@pytest.mark.parametrize("name,engine", ENGINES)
@pytest.mark.parametrize("n_tris", [12, 108, 192, 300, 588, 1100, 2500, 20000])
def test_synthetic_scenes(pt, oracle, name, engine, n_tris):
    """BASELINE.json configs[3] at test size.  Above PT_BVH_DEFAULT_MIN_TRIANGLES (96) triangles the extend stage walks a BVH
    whose answer must be the brute-force one bit for bit (the oracle is brute force)."""
    _synthetic_case(pt, oracle, name, engine, n_tris)


@pytest.mark.parametrize("name,engine", ENGINES)
@pytest.mark.parametrize("keep", [96, 97])
def test_scenes_on_either_side_of_the_bvh_threshold(pt, oracle, name, engine, keep):
    """the first 96 / 97 triangles of the 300-triangle box (open: some rays leave): the last scene that is looped over,
    the first that carries a BVH"""
    import ctypes as C
    V, T, M = scenes.synthetic_scene(300)
    T = T[:keep]
    lib = pt.load_library()
    v, t, m = pt.make_scene(V, T, M)
    n = lib.pt_hip_scene_blob_bytes(len(T), len(M))
    blob = (C.c_uint8 * n)()
    assert lib.pt_hip_scene_pack(C.cast(v, C.c_void_p), len(V), C.cast(t, C.c_void_p), len(T), C.cast(m, C.c_void_p), len(M), C.cast(blob, C.c_void_p), n) == 0
    assert lib.pt_hip_scene_has_bvh(C.cast(blob, C.c_void_p), n) == (1 if keep > pt.PT_BVH_DEFAULT_MIN_TRIANGLES else 0)
    w, h, spp = 64, 40, 3
    ref, ref_rng, st_ref, _ = oracle.render(oracle.params(w, h, spp=spp), scene=oracle.make_scene(V, T, M))
    with pt.PathTracer(w, h) as tr:
        tr.set_scene(*pt.make_scene(V, T, M))
        tr.set_options(engine=engine)
        tr.render(spp)
        img, rng, st = tr.readback(), tr.readback_rng(), tr.stats()
    assert_bit_equal(img, ref, f"{name} {keep} triangles")
    assert np.array_equal(rng, ref_rng) and st.rays == st_ref.rays


@pytest.mark.parametrize("name,engine", ENGINES)
@pytest.mark.parametrize("n_tris", [300, 1100, 2500])
def test_synthetic_scenes_brute_force(pt, oracle, name, engine, n_tris, brute_force_only):
    """The same without the BVH: 1100 and 2500 triangles cross the LDS chunk (1024) and the
    shade-in-LDS (1536) thresholds of the brute-force path."""
    _synthetic_case(pt, oracle, name, engine, n_tris)


@pytest.mark.parametrize("name,engine", ENGINES)
@pytest.mark.parametrize("n_tris,spp", [(10000, 2), (1000, 6)])
def test_bvh_equals_brute_force_on_every_pixel(pt, oracle, name, engine, n_tris, spp):
    """10 092 / 972 triangles at 960x540: BVH image == brute-force image on all 518 400 pixels (GPU vs GPU),
    plus 300 pixels against the oracle.  The 972-triangle box has the large triangles for which the worst-case
    error inflation is most of a quad, i.e. where the grazing / non-grazing split of pt_bvh.hpp does the culling."""
    V, T, M = scenes.synthetic_scene(n_tris)
    sc = pt.make_scene(V, T, M)
    w, h = 960, 540
    lib = pt.load_library()
    imgs = []
    for threshold in (pt.PT_BVH_DEFAULT_MIN_TRIANGLES, 0xFFFFFFFF):
        lib.pt_hip_set_bvh_threshold(threshold)
        try:
            with pt.PathTracer(w, h) as tr:
                tr.set_scene(*sc)
                tr.set_options(engine=engine)
                tr.render(spp)
                imgs.append((tr.readback(), tr.readback_rng(), tr.stats().rays))
        finally:
            lib.pt_hip_set_bvh_threshold(pt.PT_BVH_DEFAULT_MIN_TRIANGLES)
    assert_bit_equal(imgs[0][0], imgs[1][0], f"{name}: BVH vs brute force")
    assert np.array_equal(imgs[0][1], imgs[1][1]) and imgs[0][2] == imgs[1][2]
    ids = np.unique(np.random.default_rng(11).integers(0, w * h, 300)).astype(np.uint32)
    ref, _, _, _ = oracle.render(oracle.params(w, h, spp=spp), scene=oracle.make_scene(V, T, M), pixel_ids=ids)
    assert_bit_equal(imgs[0][0][ids], ref, f"{name}: BVH vs oracle")


@pytest.mark.parametrize("name,engine", ENGINES)
def test_bvh_with_moved_camera_and_mixed_triangle_sizes(pt, oracle, name, engine):
    """Big and tiny triangles in one BVH (the all-materials box plus 2000 small ones), camera off-centre."""
    V, T, M = scenes.all_materials_scene()
    V2, T2, M2 = scenes.synthetic_scene(2000, seed=77)
    off, moff = len(V), len(M)
    V = V + [(x * 0.3, y * 0.3, z * 0.3 + 0.4, w_) for (x, y, z, w_) in V2]
    T = T + [(a + off, b + off, c + off, m + moff) for (a, b, c, m) in T2]
    M = M + M2
    cam = dict(position=(0.2, -0.1, 0.05, 1.0), rotation=(0.02, 0.05, 0.0, 0.998), fov=80.0)
    w, h, spp = 96, 60, 2
    ref, _, st_ref, _ = oracle.render(oracle.params(w, h, cam=oracle.make_camera(**cam), depth=12, spp=spp), scene=oracle.make_scene(V, T, M))
    with pt.PathTracer(w, h, camera=pt.CameraData(cam["position"], cam["rotation"], cam["fov"])) as tr:
        tr.set_scene(*pt.make_scene(V, T, M))
        tr.set_options(engine=engine, depth=12)
        tr.render(spp)
        img, st = tr.readback(), tr.stats()
    assert_bit_equal(img, ref, name)
    assert st.rays == st_ref.rays


@pytest.mark.parametrize("name,engine", ENGINES)
def test_empty_scene_and_depth_one(pt, oracle, name, engine):
    w, h = 32, 16
    with pt.PathTracer(w, h) as tr:
        tr.set_scene(*pt.make_scene([], [], []))
        tr.set_options(engine=engine)
        tr.render(2)
        img, st = tr.readback(), tr.stats()
    assert np.array_equal(img, np.tile(np.array([0, 0, 0, 1], dtype=np.float32), (w * h, 1))) and st.rays == w * h * 2
    ref, _, st_ref, _ = oracle.render(oracle.params(w, h, depth=1, spp=3))
    with pt.PathTracer(w, h) as tr:
        tr.set_options(engine=engine, depth=1)
        tr.render(3)
        img, st = tr.readback(), tr.stats()
    assert_bit_equal(img, ref, name)
    assert st.rays == st_ref.rays == w * h * 3


@pytest.mark.parametrize("name,engine", ENGINES)
def test_moved_and_rotated_camera(pt, oracle, name, engine):
    w, h, spp = 96, 54, 2
    q = (0.05, -0.12, 0.02, 0.99)      # not normalised on purpose: inverse() divides by |q|^2
    cam_o = oracle.make_camera(position=(0.1, 0.05, -0.2, 1.0), rotation=q, fov=75.0)
    ref, _, _, _ = oracle.render(oracle.params(w, h, cam=cam_o, spp=spp))
    with pt.PathTracer(w, h, camera=pt.CameraData((0.1, 0.05, -0.2, 1.0), q, 75.0)) as tr:
        tr.set_options(engine=engine)
        tr.render(spp)
        img = tr.readback()
    assert_bit_equal(img, ref, name)


@pytest.mark.parametrize("name,engine", ENGINES)
def test_camera_update_restarts_next_frame(pt, oracle, name, engine):
    w, h = 64, 36
    moved = pt.CameraData(fieldOfView=85.0)
    with pt.PathTracer(w, h) as tr:
        tr.set_options(engine=engine)
        for frame in range(5):
            tr.BeginDrawing()
            tr.Update(moved if frame >= 2 else pt.CameraData())
            tr.FinalizeDrawing()
        img, st = tr.readback(), tr.stats()
    ref, _, _, _ = oracle.render(oracle.params(w, h, cam=oracle.make_camera(fov=85.0), spp=2))
    assert_bit_equal(img, ref, name)
    assert st.frames == 2


@pytest.mark.parametrize("name,engine", ENGINES)
def test_resize(pt, oracle, name, engine):
    with pt.PathTracer(64, 36) as tr:
        tr.set_options(engine=engine)
        tr.render(2)
        tr.Resize(50, 30)
        tr.render(3)
        img = tr.readback()
        assert tr.size()[:2] == (50, 30)
    ref, _, _, _ = oracle.render(oracle.params(50, 30, spp=3))
    assert_bit_equal(img, ref, name)


@pytest.mark.parametrize("name,engine", ENGINES)
def test_tile_split_is_bit_identical(pt, oracle, name, engine):
    """SURVEY 8e partitioning B: each instance renders a pixel range with the sequential RNG."""
    w, h, spp = 80, 45, 3
    ref, ref_rng, _, _ = oracle.render(oracle.params(w, h, spp=spp))
    n = w * h
    cuts = [0, 1000, 1001, 2900, n]
    parts, rngs = [], []
    for a, b in zip(cuts[:-1], cuts[1:]):
        with pt.PathTracer(w, h) as tr:
            tr.set_options(engine=engine, pixel_first=a, pixel_count=b - a)
            tr.render(spp)
            parts.append(tr.readback())
            rngs.append(tr.readback_rng())
    assert_bit_equal(np.concatenate(parts), ref, name)
    assert np.array_equal(np.concatenate(rngs), ref_rng)


@pytest.mark.parametrize("name,engine", ENGINES)
@pytest.mark.parametrize("shards", [2, 8])
def test_sample_split_shards_combine_to_sharded_mean(pt, oracle, name, engine, shards):
    """SURVEY 8e partitioning A on one GPU: `shards` instances stand in for the ranks."""
    from gpu_util import DeviceBuffer, sync
    lib = pt.load_library()
    w, h, per = 64, 32, 3
    n = w * h
    stacked = DeviceBuffer(lib, shards * n * 16)
    host = np.empty((shards, n, 4), dtype=np.float32)
    for v in range(shards):
        with pt.PathTracer(w, h) as tr:
            tr.set_options(engine=engine, rng_mode=pt.PT_RNG_STRIDED, stride=35, sample_offset=v * per, samples_per_pass=2)
            tr.render(per)
            host[v] = tr.readback()
    stacked.upload(host)
    out = DeviceBuffer(lib, n * 16)
    assert lib.pt_hip_combine_shards(stacked.ptr, shards, n, out.ptr, None) == 0
    sync(lib)
    got = out.download(np.float32).reshape(n, 4)
    want, _ = oracle.render_sharded(oracle.params(w, h, rng_mode=oracle.RNG_STRIDED, stride=35, spp=per * shards), shards)
    assert_bit_equal(got, want, name)


def test_hip_plugin_through_the_reference_host_loop(pt, oracle, tmp_path):
    """The HIP plugin loaded by name through IPathTracerManagement.h and driven by the C++ host
    loop (tests/host/host_driver.cpp), vtable calls and all."""
    driver = os.path.join(ROOT, "tests", "host", "host_driver")
    out = tmp_path / "hip.raw"
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "pathtracer_amd") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    r = subprocess.run([driver, "PT_HipWavefront", "640", "480", "3", str(out), "--resize-at", "1", "320", "200"],
                       capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    img = np.fromfile(out, dtype=np.float32).reshape(-1, 4)
    ref, _, _, _ = oracle.render(oracle.params(320, 200, spp=2))
    assert_bit_equal(img, ref, "host loop")


def test_engines_agree_at_full_size_and_sampled_pixels_match_oracle(pt, oracle):
    """BASELINE.json configs[1]: 1920x1080, 64 spp.  Every pixel: fused == wavefront bit for bit.
    20 000 pixels spread over the image: == oracle bit for bit.  Properties over the whole image:
    ray count 2*N*spp, every RNG stream advanced by exactly 35*spp steps."""
    w, h, spp = 1920, 1080, 64
    n = w * h
    imgs = {}
    for name, engine in ENGINES:
        with pt.PathTracer(w, h) as tr:
            if engine == 0:
                tr.set_options(engine=engine, rng_mode=pt.PT_RNG_STRIDED, stride=35, samples_per_pass=16)
            else:
                tr.set_options(engine=engine)
            tr.render(spp)
            imgs[name] = tr.readback()
            st = tr.stats()
            assert st.rays == 2 * n * spp and st.diffuse_bounces == n * spp
            if engine == 1:
                rng = tr.readback_rng()
    assert_bit_equal(imgs["wavefront"], imgs["fused"], "wavefront vs fused")
    ids = np.unique(np.concatenate([np.random.default_rng(7).integers(0, n, 20000), [0, w - 1, n - w, n - 1]])).astype(np.uint32)
    ref, ref_rng, _, _ = oracle.render(oracle.params(w, h, spp=spp), pixel_ids=ids)
    assert_bit_equal(imgs["fused"][ids], ref, "sampled pixels vs oracle")
    assert np.array_equal(rng[ids], ref_rng)
    start = oracle.seed_states(1337, 64, pixel_ids=ids[:64])
    for k in range(64):
        assert rng[ids[k]] == oracle.lib().pto_rng_jump(int(start[k]), 35 * spp)
    img = imgs["fused"]
    assert np.isfinite(img).all()
    assert 0.3 < img[:, 0].mean() < 0.45 and abs(img[:, 1].mean() - 0.2 * img[:, 0].mean()) < 1e-4


# ---------------------------------------------------------------- round-2 regressions
@pytest.mark.parametrize("name,engine", ENGINES)
def test_scene_change_restarts_even_if_update_follows(pt, oracle, name, engine):
    """set_scene -> Update(same camera) -> render: Update overwrites the restart flag like the reference's does
    (PathTracer.cu:99-101), so the scene change must restart on its own -- else the new scene's samples would be folded
    into the old scene's mean with the old streams."""
    w, h, spp = 96, 54, 3
    V, T, M = scenes.all_materials_scene()
    ref, ref_rng, _, _ = oracle.render(oracle.params(w, h, spp=spp), scene=oracle.make_scene(V, T, M))
    with pt.PathTracer(w, h) as tr:
        tr.set_options(engine=engine)
        tr.render(2)                                  # two samples of the default scene
        tr.set_scene(*pt.make_scene(V, T, M))
        tr.Update(pt.CameraData())                    # unchanged camera: clears the reference-style flag
        for _ in range(spp):
            tr.BeginDrawing()
            tr.Update(pt.CameraData())
            tr.FinalizeDrawing()
        img, rng, st = tr.readback(), tr.readback_rng(), tr.stats()
    assert st.frames == spp
    assert_bit_equal(img, ref, name)
    assert np.array_equal(rng, ref_rng)


@pytest.mark.parametrize("name,engine", ENGINES)
@pytest.mark.parametrize("depth", [1, 2])
def test_statistics_at_the_depth_limit(pt, oracle, name, engine, depth):
    """A DIFFUSE shade on the last bounce iteration draws its 33 numbers although no ray follows: both engines count it
    (2 * paths + 33 * diffuse_bounces == draws), sequential and strided."""
    w, h, spp = 64, 36, 3
    V, T, M = scenes.all_materials_scene()
    for strided in (False, True):
        stride = 2 + 33 * depth
        ref, _, st_ref, _ = oracle.render(oracle.params(w, h, depth=depth, spp=spp, rng_mode=1 if strided else 0, stride=stride),
                                          scene=oracle.make_scene(V, T, M))
        with pt.PathTracer(w, h) as tr:
            tr.set_scene(*pt.make_scene(V, T, M))
            tr.set_options(engine=engine, depth=depth, rng_mode=1 if strided else 0, stride=stride, samples_per_pass=2)
            tr.render(spp)
            img, st = tr.readback(), tr.stats()
        assert_bit_equal(img, ref, f"{name} depth {depth}")
        assert st.rays == st_ref.rays and st.paths == w * h * spp
        assert 2 * st.paths + 33 * st.diffuse_bounces == st_ref.draws, (name, depth, strided, st.diffuse_bounces)


def test_blob_keeps_its_bvh_decision_when_the_threshold_moves(pt, oracle):
    """A scene packed while the BVH threshold was lifted (brute force) and rendered after it is back to its default, and the other
    way round: the launchers dispatch on what the blob carries, not on the process-wide knob."""
    V, T, M = scenes.synthetic_scene(1100)
    w, h, spp = 64, 40, 2
    ref, _, st_ref, _ = oracle.render(oracle.params(w, h, spp=spp), scene=oracle.make_scene(V, T, M))
    lib = pt.load_library()
    for packed_under, rendered_under in ((0xFFFFFFFF, pt.PT_BVH_DEFAULT_MIN_TRIANGLES), (pt.PT_BVH_DEFAULT_MIN_TRIANGLES, 0xFFFFFFFF)):
        for name, engine in ENGINES:
            try:
                with pt.PathTracer(w, h) as tr:
                    lib.pt_hip_set_bvh_threshold(packed_under)
                    tr.set_scene(*pt.make_scene(V, T, M))
                    lib.pt_hip_set_bvh_threshold(rendered_under)
                    tr.set_options(engine=engine)
                    tr.render(spp)
                    img, st = tr.readback(), tr.stats()
            finally:
                lib.pt_hip_set_bvh_threshold(pt.PT_BVH_DEFAULT_MIN_TRIANGLES)
            assert_bit_equal(img, ref, f"{name} packed under {packed_under:#x}, rendered under {rendered_under:#x}")
            assert st.rays == st_ref.rays


def test_create_leaves_the_callers_device_alone(pt):
    """Every method binds the instance's device for its own duration only (DeviceScope): the calling thread's current
    device is what it was, before and after."""
    import ctypes as C
    lib = pt.load_library()
    before = C.c_int(-1)
    assert lib.pt_hip_get_device(C.byref(before)) == 0
    with pt.PathTracer(32, 16, device=0) as tr:
        tr.render(1)
        tr.Resize(16, 16)
        tr.render(1)
        assert tr.get_options().device == 0
        after = C.c_int(-1)
        assert lib.pt_hip_get_device(C.byref(after)) == 0 and after.value == before.value


@pytest.mark.parametrize("name,engine", ENGINES)
def test_brute_force_scene_at_the_lds_limit(pt, oracle, name, engine, brute_force_only):
    """1536 triangles with 256 materials, no BVH: triangles + materials + camera-ray records pass the 160 KiB of a CU, so
    the wavefront engine must fall back from its one-launch pass to the staged pipeline instead of failing the launch."""
    V, T, M = scenes.synthetic_scene(1800)          # 1728 triangles (whole quads per face); the first 1536 of them
    T = T[:1536]
    assert len(T) == 1536
    M = [(M[k % len(M)][0], tuple(np.float32(c * (0.9 + 0.0003 * k)) for c in M[k % len(M)][1])) for k in range(256)]
    T = [(a, b, c, (m + 6 * (k % 42)) % 256) for k, (a, b, c, m) in enumerate(T)]
    w, h, spp = 48, 30, 2
    ref, _, st_ref, _ = oracle.render(oracle.params(w, h, spp=spp), scene=oracle.make_scene(V, T, M))
    with pt.PathTracer(w, h) as tr:
        tr.set_scene(*pt.make_scene(V, T, M))
        tr.set_options(engine=engine)
        tr.render(spp)
        img, st = tr.readback(), tr.stats()
    assert_bit_equal(img, ref, name)
    assert st.rays == st_ref.rays


@pytest.mark.gpu
@pytest.mark.parametrize("strided", [False, True], ids=["sequential", "strided"])
def test_fused_engine_on_bvh_scenes_persistent_and_plain_forms_agree(pt, oracle, strided):
    """k_fused_bvh (pt_fused_bvh.hpp: persistent walks, finished rays serviced at a quorum, pixels handed out per block --
    what the fused engine runs on BVH scenes) and k_fused<*, 2> (a lane per pixel to the end; PT_FUSED_BVH_PLAIN in the
    environment) both give the oracle's image, streams and ray count; one sample per call (the reference's frame loop: the
    pixel queue is the only refill) and many."""
    import os
    V, T, M = scenes.synthetic_scene(2500)
    w, h, depth = 333, 187, 10                       # 62 271 pixels: several blocks, a ragged last range
    kw = dict(rng_mode=pt.PT_RNG_STRIDED, stride=2 + 33 * depth) if strided else {}
    okw = dict(rng_mode=oracle.RNG_STRIDED, stride=2 + 33 * depth) if strided else {}
    ref, ref_rng, st_ref, _ = oracle.render(oracle.params(w, h, depth=depth, spp=6, **okw), scene=oracle.make_scene(V, T, M))
    for plain in (False, True):
        if plain:
            os.environ["PT_FUSED_BVH_PLAIN"] = "1"
        try:
            with pt.PathTracer(w, h) as tr:
                tr.set_scene(*pt.make_scene(V, T, M))
                tr.set_options(engine=pt.PT_ENGINE_FUSED, depth=depth, **kw)
                tr.render(1)
                tr.render(5)
                img, rng, st = tr.readback(), tr.readback_rng(), tr.stats()
        finally:
            os.environ.pop("PT_FUSED_BVH_PLAIN", None)
        diff = (img.view(np.uint32) != ref.view(np.uint32)).any(axis=1)
        assert not diff.any(), f"plain={plain}: {int(diff.sum())} pixels differ, first {int(np.argmax(diff))}"
        assert st.rays == st_ref.rays and st.diffuse_bounces * 33 + st.paths * 2 == st_ref.draws
        if not strided:
            assert np.array_equal(rng, ref_rng)


@pytest.mark.gpu
@pytest.mark.parametrize("scene,w,h,spp,depth,shading,strided", [("default", 333, 187, 5, 10, 0, False), ("materials", 200, 120, 6, 16, 0, False),
                                                                 ("mirror", 160, 90, 4, 16, 1, False), ("materials", 128, 72, 6, 12, 0, True),
                                                                 ("synthetic300", 96, 54, 3, 10, 0, False), ("soup", 96, 60, 3, 8, 0, False)])
def test_fused_engine_out_of_step_form(pt, oracle, scene, w, h, spp, depth, shading, strided):
    """k_fused<*, *, true> (PT_FRAME_OUT_OF_STEP; forced for every render here through the environment knob): the pixels of a wave
    take their samples out of step.  Image, streams, ray and draw counts are the oracle's on every scene kind without a BVH."""
    import os
    V, T, M = {"default": scenes.default_scene, "materials": scenes.all_materials_scene, "mirror": scenes.mirror_scene,
               "synthetic300": lambda: scenes.synthetic_scene(300), "soup": lambda: scenes.triangle_soup(60, 3)}[scene]()
    stride = 2 + (2 if shading else 33) * depth
    okw = dict(rng_mode=oracle.RNG_STRIDED, stride=stride) if strided else {}
    kw = dict(rng_mode=pt.PT_RNG_STRIDED, stride=stride) if strided else {}
    ref, ref_rng, st_ref, _ = oracle.render(oracle.params(w, h, depth=depth, spp=spp, shading=shading, **okw), scene=oracle.make_scene(V, T, M))
    os.environ["PT_FUSED_OUT_OF_STEP"] = "1"
    try:
        with pt.PathTracer(w, h) as tr:
            tr.set_scene(*pt.make_scene(V, T, M))
            tr.set_options(engine=pt.PT_ENGINE_FUSED, depth=depth, shading=shading, **kw)
            tr.render(1)
            tr.render(spp - 1)
            img, rng, st = tr.readback(), tr.readback_rng(), tr.stats()
    finally:
        os.environ.pop("PT_FUSED_OUT_OF_STEP", None)
    diff = (img.view(np.uint32) != ref.view(np.uint32)).any(axis=1)
    assert not diff.any(), f"{int(diff.sum())} pixels differ, first {int(np.argmax(diff))}"
    assert st.rays == st_ref.rays
    if not strided:
        assert np.array_equal(rng, ref_rng)


@pytest.mark.gpu
@pytest.mark.parametrize("shading,depth", [(0, 10), (1, 12), (0, 3)])
def test_wavefront_bvh_late_bounces_in_one_persistent_launch(pt, oracle, shading, depth):
    """k_bounce_tail (pt_wavefront.hpp bounce_tail_block): on BVH scenes the rays of bounce 2 and everything after them run in one
    persistent launch; PT_WAVEFRONT_TAIL=off in the environment keeps every bounce staged.  Both give the oracle's image and the
    oracle's rays PER BOUNCE (the tail fills the counter rows the staged shade stages would have); depth 3: the shortest path the
    tail takes, mirrors under extended shading: long ones."""
    import os
    V, T, M = scenes.synthetic_scene(2500)
    if shading:
        M = [(m[0], m[1], 0.8) for m in M]
    w, h, spp = 320, 180, 6
    stride = 2 + (2 if shading else 33) * depth
    ref, _, st_ref, _ = oracle.render(oracle.params(w, h, depth=depth, spp=spp, shading=shading, rng_mode=oracle.RNG_STRIDED, stride=stride),
                                      scene=oracle.make_scene(V, T, M))
    for tail in ("on", "off"):
        if tail == "off":
            os.environ["PT_WAVEFRONT_TAIL"] = "off"
        try:
            with pt.PathTracer(w, h) as tr:
                tr.set_scene(*pt.make_scene(V, T, M))
                tr.set_options(engine=pt.PT_ENGINE_WAVEFRONT, depth=depth, shading=shading, rng_mode=pt.PT_RNG_STRIDED, stride=stride, samples_per_pass=4)
                tr.render(spp)
                img, st = tr.readback(), tr.stats()
        finally:
            os.environ.pop("PT_WAVEFRONT_TAIL", None)
        diff = (img.view(np.uint32) != ref.view(np.uint32)).any(axis=1)
        assert not diff.any(), f"tail {tail}: {int(diff.sum())} pixels differ"
        assert st.rays == st_ref.rays and list(st.rays_per_bounce) == list(st_ref.rays_per_bounce), tail
        if not shading:                                   # (extended: mirrors continue without a draw)
            assert st.diffuse_bounces * 33 + st.paths * 2 == st_ref.draws
    assert st_ref.rays_per_bounce[2] > 0                  # there ARE late bounces on this scene

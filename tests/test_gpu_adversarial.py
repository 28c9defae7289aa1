"""Camera-ray culling at its borderline, on the GPU: triangles whose edges run along pixel boundaries (vertices on the rays
through pixel corners, tests/test_device_code_on_host.py pixel_corner_scene), where the wave ORs the need masks of its 64
pixels (primary_need_mask).  Both engines, every pixel against the oracle, bit for bit."""
import numpy as np
import pytest

from test_device_code_on_host import pixel_corner_scene

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,engine", [("fused", 1), ("wavefront", 0)])
@pytest.mark.parametrize("w,h,seed", [(64, 40, 1), (96, 54, 5), (200, 33, 7)])
def test_edges_on_pixel_boundaries(pt, oracle, name, engine, w, h, seed):
    V, T, M = pixel_corner_scene(oracle, w, h, 31, seed)          # 62 triangles: inside the 64-bit need mask
    spp, depth = 6, 6
    ref, ref_rng, st, _ = oracle.render(oracle.params(w, h, depth=depth, spp=spp), scene=oracle.make_scene(V, T, M))
    with pt.PathTracer(w, h) as tr:
        tr.set_scene(*pt.make_scene(V, T, M))
        tr.set_options(engine=engine, depth=depth, samples_per_pass=3)
        tr.render(spp)
        img, rng, got = tr.readback(), tr.readback_rng(), tr.stats()
    diff = (img.view(np.uint32) != ref.view(np.uint32)).any(axis=1)
    assert not diff.any(), f"{name}: {int(diff.sum())} pixels differ, first {int(np.argmax(diff))}"
    assert np.array_equal(rng, ref_rng) and got.rays == st.rays


@pytest.mark.parametrize("name,engine", [("fused", 1), ("wavefront", 0)])
def test_room_with_a_tessellated_ball(pt, oracle, name, engine):
    """classes for the walls, free triangles (ClassBlock::kappa_free) for the ball's facets: every pixel against the oracle"""
    from test_bvh_adversarial import room_with_a_ball
    V, T, M = room_with_a_ball(300, 1200, 8)
    w, h, spp, depth = 96, 64, 3, 8
    ref, ref_rng, st, _ = oracle.render(oracle.params(w, h, depth=depth, spp=spp), scene=oracle.make_scene(V, T, M))
    with pt.PathTracer(w, h) as tr:
        tr.set_scene(*pt.make_scene(V, T, M))
        tr.set_options(engine=engine, depth=depth)
        tr.render(spp)
        img, rng, got = tr.readback(), tr.readback_rng(), tr.stats()
    diff = (img.view(np.uint32) != ref.view(np.uint32)).any(axis=1)
    assert not diff.any(), f"{name}: {int(diff.sum())} pixels differ, first {int(np.argmax(diff))}"
    assert np.array_equal(rng, ref_rng) and got.rays == st.rays


@pytest.mark.parametrize("name,engine,strided", [("fused", 1, False), ("wavefront", 0, False), ("wavefront strided", 0, True)])
def test_heap_of_large_triangles_fills_the_traversal_stack(pt, oracle, name, engine, strided):
    """1500 triangles that each span the scene: every box overlaps every other, a walk defers three children at every level of the
    four-wide tree and its stack grows past the levels kept in LDS into the private part (BvhStack::spill) -- the host harness shows
    that it does (tests/test_bvh_host.py, heap-1500: deepest > lds_levels), this is the same scene on the GPU: every pixel against the
    oracle's loop over all triangles."""
    import scenes
    V, T, M = scenes.heap_of_large_triangles(1500, 9)
    w, h, spp, depth = 64, 36, 2, 4
    mode = dict(rng_mode=oracle.RNG_STRIDED, stride=2 + 33 * depth) if strided else {}
    ref, ref_rng, st, _ = oracle.render(oracle.params(w, h, depth=depth, spp=spp, **mode), scene=oracle.make_scene(V, T, M))
    with pt.PathTracer(w, h) as tr:
        tr.set_scene(*pt.make_scene(V, T, M))
        if strided:
            tr.set_options(engine=engine, depth=depth, rng_mode=pt.PT_RNG_STRIDED, stride=2 + 33 * depth, samples_per_pass=2)
        else:
            tr.set_options(engine=engine, depth=depth)
        tr.render(spp)
        img, got = tr.readback(), tr.stats()
    diff = (img.view(np.uint32) != ref.view(np.uint32)).any(axis=1)
    assert not diff.any(), f"{name}: {int(diff.sum())} pixels differ, first {int(np.argmax(diff))}"
    assert got.rays == st.rays and st.rays > w * h * spp


@pytest.mark.parametrize("scene_name,depth,strided", [("materials", 16, False), ("materials", 16, True), ("synthetic108", 10, False), ("soup40", 8, True)])
def test_small_trees_walked_from_lds(pt, oracle, scene_name, depth, strided):
    """Scenes of up to 128 triangles that carry a BVH -- the plugin's BVH form of a small scene (17..96 triangles, pt_plugin.h) and scenes just
    above the threshold -- are walked by the fused engine from a copy of the whole tree in LDS (k_fused_bvh<*, true>); PT_LDS_TREE_OFF in
    the environment keeps the walk in global memory.  Both against the oracle, every pixel, and the streams' last words."""
    import os
    import scenes
    V, T, M = {"materials": scenes.all_materials_scene, "synthetic108": lambda: scenes.synthetic_scene(108), "soup40": lambda: scenes.triangle_soup(40, 5)}[scene_name]()
    w, h, spp = 160, 90, 3
    mode = dict(rng_mode=oracle.RNG_STRIDED, stride=2 + 33 * depth) if strided else {}
    ref, ref_rng, st, _ = oracle.render(oracle.params(w, h, depth=depth, spp=spp, **mode), scene=oracle.make_scene(V, T, M))
    lib = pt.load_library()
    lib.pt_hip_set_bvh_threshold(16)                        # every scene here is packed with a BVH
    try:
        for off in (False, True):
            if off:
                os.environ["PT_LDS_TREE_OFF"] = "1"
            try:
                with pt.PathTracer(w, h) as tr:
                    tr.set_scene(*pt.make_scene(V, T, M))
                    kw = dict(rng_mode=pt.PT_RNG_STRIDED, stride=2 + 33 * depth) if strided else {}
                    tr.set_options(engine=pt.PT_ENGINE_FUSED, depth=depth, **kw)
                    tr.render(1)
                    tr.render(spp - 1)
                    img, rng, got = tr.readback(), tr.readback_rng(), tr.stats()
            finally:
                os.environ.pop("PT_LDS_TREE_OFF", None)
            diff = (img.view(np.uint32) != ref.view(np.uint32)).any(axis=1)
            assert not diff.any(), f"{scene_name}, tree in {'global memory' if off else 'LDS'}: {int(diff.sum())} pixels differ, first {int(np.argmax(diff))}"
            assert got.rays == st.rays and (strided or np.array_equal(rng, ref_rng))
    finally:
        lib.pt_hip_set_bvh_threshold(pt.PT_BVH_DEFAULT_MIN_TRIANGLES)


@pytest.mark.parametrize("strided,out_of_step", [(False, False), (True, False), (True, True)])
@pytest.mark.parametrize("w,h,first,count", [(200, 33, 0, None), (96, 54, 1000, 96 * 54 - 1777), (64, 40, 0, None), (30, 70, 7, 30 * 70 - 7)])
def test_need_masks_built_ahead_equal_masks_worked_out_in_the_launch(pt, oracle, w, h, first, count, strided, out_of_step):
    """pt_hip_build_need_masks + pt_frame.need_masks (the plugin's form: one small launch per camera, a lane per 64 pixels) against the
    fused kernel working the masks out itself, a triangle per lane: same image, same streams, same ray count -- on the scene whose edges
    run along pixel boundaries, images whose rows are not multiples of 64 pixels (a wave's pixels span rows) and pixel ranges that
    start and stop inside a row.  The words are not ~0 (something is culled) and every word past the range is 0."""
    import ctypes as C
    from gpu_util import DeviceBuffer, make_frame, sync, upload_scene
    lib = pt.load_library()
    V, T, M = pixel_corner_scene(oracle, w, h, 31, 3)
    scene_dev = upload_scene(pt, lib, pt.make_scene(V, T, M))
    assert not scene_dev.frame_flags & pt.PT_FRAME_SCENE_BVH
    n_pix = w * h - first if count is None else count
    depth, spp = 6, 5
    flags = scene_dev.frame_flags | (pt.PT_FRAME_OUT_OF_STEP if out_of_step else 0)
    fr = make_frame(pt, oracle, w, h, pixel_first=first, pixel_count=n_pix, depth=depth, rng_mode=1 if strided else 0, stride=2 + 33 * depth, flags=flags)
    n_bytes = lib.pt_hip_need_masks_bytes(n_pix)
    assert n_bytes == ((n_pix + 255) // 256) * 4 * 8
    masks = DeviceBuffer(lib, n_bytes).upload(np.full(n_bytes // 8, 0xCDCDCDCDCDCDCDCD, dtype=np.uint64))
    assert lib.pt_hip_build_need_masks(C.byref(fr), scene_dev.ptr, len(T), len(M), masks.ptr, None) == 0, lib.pt_hip_last_error()
    out = []
    for ahead in (False, True):
        fr.need_masks = masks.ptr if ahead else 0
        rng = DeviceBuffer(lib, n_pix * 4)
        assert lib.pt_hip_rng_seed(C.byref(fr), rng.ptr, None) == 0
        mean = DeviceBuffer(lib, n_pix * 16).upload(np.ones(n_pix * 4, dtype=np.float32))
        stats = DeviceBuffer(lib, pt.PT_STATS_WORDS * 8).upload(np.zeros(pt.PT_STATS_WORDS, dtype=np.uint64))
        assert lib.pt_hip_render_fused(C.byref(fr), scene_dev.ptr, len(T), len(M), 0, 0, spp, rng.ptr, mean.ptr, stats.ptr, None) == 0, lib.pt_hip_last_error()
        sync(lib)
        out.append((mean.download(np.uint32).reshape(n_pix, 4), rng.download(np.uint32), int(stats.download(np.uint64)[0])))
    diff = (out[0][0] != out[1][0]).any(axis=1)
    assert not diff.any(), f"{int(diff.sum())} pixels differ, first {int(np.argmax(diff))}"
    assert np.array_equal(out[0][1], out[1][1]) and out[0][2] == out[1][2] and out[0][2] > n_pix * spp
    ref, _, st, _ = oracle.render(oracle.params(w, h, depth=depth, spp=spp, **(dict(rng_mode=oracle.RNG_STRIDED, stride=2 + 33 * depth) if strided else {})),
                                  scene=oracle.make_scene(V, T, M), pixel_first=first, n=n_pix)
    assert np.array_equal(out[1][0], ref.view(np.uint32).reshape(-1, 4)) and out[1][2] == st.rays
    words = masks.download(np.uint64)
    used = (n_pix + 63) // 64
    full = np.uint64((1 << len(T)) - 1)
    assert (words[:used] & ~full == 0).all() and (words[:used] != full).any() and (words[:used] != 0).any()
    assert (words[used:] == 0).all()


def test_need_masks_are_for_scenes_without_a_bvh(pt, oracle):
    import ctypes as C
    import scenes
    from gpu_util import DeviceBuffer, make_frame, upload_scene
    lib = pt.load_library()
    V, T, M = scenes.synthetic_scene(300)
    scene_dev = upload_scene(pt, lib, pt.make_scene(V, T, M))
    fr = make_frame(pt, oracle, 32, 16, flags=scene_dev.frame_flags)
    masks = DeviceBuffer(lib, lib.pt_hip_need_masks_bytes(32 * 16))
    assert lib.pt_hip_build_need_masks(C.byref(fr), scene_dev.ptr, len(T), len(M), masks.ptr, None) == -1
    assert b"BVH" in lib.pt_hip_last_error()


@pytest.mark.parametrize("cap", ["1", "3", "7"])
@pytest.mark.parametrize("strided,out_of_step", [(False, False), (True, False), (False, True)])
def test_fused_pixel_loop_of_several_trips(pt, oracle, monkeypatch, cap, strided, out_of_step):
    """A launch of one or two samples per pixel runs on as many blocks as the machine holds at once and its blocks walk the 256-pixel
    tiles (fused_grid_blocks) -- 1080p on a whole MI355X: four trips.  PT_FUSED_GRID_CAP forces that loop on a small image: blocks
    that take several tiles, the last of them partial, some none at all; with the culling masks built ahead, against the oracle."""
    monkeypatch.setenv("PT_FUSED_GRID_CAP", cap)
    if out_of_step:
        monkeypatch.setenv("PT_FUSED_OUT_OF_STEP", "1")
    w, h, depth = 100, 37, 6                                   # 3700 pixels: 14.45 tiles
    V, T, M = pixel_corner_scene(oracle, w, h, 20, 11)
    mode = dict(rng_mode=oracle.RNG_STRIDED, stride=2 + 33 * depth) if strided else {}
    ref, ref_rng, st, _ = oracle.render(oracle.params(w, h, depth=depth, spp=4, **mode), scene=oracle.make_scene(V, T, M))
    with pt.PathTracer(w, h) as tr:
        tr.set_scene(*pt.make_scene(V, T, M))
        kw = dict(rng_mode=pt.PT_RNG_STRIDED, stride=2 + 33 * depth) if strided else {}
        tr.set_options(engine=pt.PT_ENGINE_FUSED, depth=depth, **kw)
        for n in (1, 2, 1):
            tr.render(n)
        img, rng, got = tr.readback(), tr.readback_rng(), tr.stats()
    diff = (img.view(np.uint32) != ref.view(np.uint32)).any(axis=1)
    assert not diff.any(), f"{int(diff.sum())} pixels differ, first {int(np.argmax(diff))}"
    assert got.rays == st.rays and (strided or np.array_equal(rng, ref_rng))


@pytest.mark.parametrize("name,engine", [("fused", 1), ("wavefront", 0)])
def test_one_sample_calls_at_1080p(pt, oracle, name, engine):
    """the reference's frame loop at its size: calls of one and two samples on the capped grids (four tiles or chunks per block on a
    whole MI355X: fused_grid_blocks, and the one-launch pass's grid in pt_hip_render_wavefront), every pixel against the oracle"""
    w, h = 1920, 1080
    ref, ref_rng, st, _ = oracle.render(oracle.params(w, h, spp=4))
    with pt.PathTracer(w, h) as tr:
        tr.set_options(engine=engine)
        for n in (1, 2, 1):
            tr.render(n)
        img, rng, got = tr.readback(), tr.readback_rng(), tr.stats()
    diff = (img.view(np.uint32) != ref.view(np.uint32)).any(axis=1)
    assert not diff.any(), f"{int(diff.sum())} pixels differ, first {int(np.argmax(diff))}"
    assert np.array_equal(rng, ref_rng) and got.rays == st.rays

"""BASELINE.json configs[2] and configs[3] at their stated sizes -- bit for bit, no tolerance.

configs[3]: synthetic scenes of 10 092 / 30 000 / 99 372 triangles (tests/scenes.py synthetic_scene: the generator
rounds the request to whole quads per face) at 1920x1080, both engines, through the BVH; the oracle is brute force
(the reference's loop over all triangles, kernels.cu:62-68), so it checks a sample of pixels and the size-independent
properties carry the rest: the two engines agree on EVERY pixel, ray counts equal the oracle's on a pixel range, the
image is finite.  configs[2]: the full 128-sample share of rank 3 of 8 at 3840x2160, engines compared on every pixel."""
import functools

import numpy as np
import pytest

import scenes

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def assert_bit_equal(img, ref, what):
    diff = (bits(img) != bits(ref)).any(axis=1)
    assert not diff.any(), f"{what}: {int(diff.sum())} of {len(diff)} pixels differ; first at {int(np.argmax(diff))}: {img[np.argmax(diff)]} vs {ref[np.argmax(diff)]}"


@functools.lru_cache(maxsize=2)
def _scene(n_tris):
    return scenes.synthetic_scene(n_tris)


W, H, DEPTH = 1920, 1080, 10
STRIDE = 2 + 33 * DEPTH          # strided streams: the worst-case draw count of a sample, so the wavefront engine batches


@pytest.mark.parametrize("n_tris", [30000, 100000])
def test_c4_large_scenes_at_1080p(pt, oracle, n_tris):
    """30 000 and 99 372 triangles at 1920x1080, 4 spp, strided/332: fused == wavefront on all 2 073 600 pixels, 2 400
    pixels spread over the image == the brute-force oracle, and the ray count of a 4 096-pixel range == the oracle's."""
    V, T, M = _scene(n_tris)
    assert len(T) > 0.99 * n_tris
    sc_pt, sc_o = pt.make_scene(V, T, M), oracle.make_scene(V, T, M)
    spp, n = 4, W * H
    imgs, rays = {}, {}
    for name, engine in (("fused", pt.PT_ENGINE_FUSED), ("wavefront", pt.PT_ENGINE_WAVEFRONT)):
        with pt.PathTracer(W, H) as tr:
            tr.set_scene(*sc_pt)
            tr.set_options(engine=engine, depth=DEPTH, rng_mode=pt.PT_RNG_STRIDED, stride=STRIDE, samples_per_pass=4)
            tr.render(spp)
            imgs[name], rays[name] = tr.readback(), tr.stats().rays
    assert_bit_equal(imgs["wavefront"], imgs["fused"], f"{len(T)} triangles: wavefront vs fused")
    assert rays["fused"] == rays["wavefront"] and rays["fused"] >= n * spp
    assert np.isfinite(imgs["fused"]).all()
    ids = np.unique(np.concatenate([np.random.default_rng(n_tris).integers(0, n, 2400), [0, W - 1, n - W, n - 1]])).astype(np.uint32)
    assert len(ids) >= 2000
    p = oracle.params(W, H, depth=DEPTH, spp=spp, rng_mode=oracle.RNG_STRIDED, stride=STRIDE)
    ref, _, _, _ = oracle.render(p, scene=sc_o, pixel_ids=ids)
    assert_bit_equal(imgs["fused"][ids], ref, f"{len(T)} triangles: sampled pixels vs oracle")
    # ray count on a pixel range (tile split: the same pixels rendered by an instance of their own)
    first, count = 777 * W + 300, 4096
    _, _, st_ref, _ = oracle.render(p, scene=sc_o, pixel_first=first, n=count)
    with pt.PathTracer(W, H) as tr:
        tr.set_scene(*sc_pt)
        tr.set_options(engine=pt.PT_ENGINE_WAVEFRONT, depth=DEPTH, rng_mode=pt.PT_RNG_STRIDED, stride=STRIDE, samples_per_pass=4,
                       pixel_first=first, pixel_count=count)
        tr.render(spp)
        tile, st = tr.readback(), tr.stats()
    assert st.rays == st_ref.rays
    assert list(st.rays_per_bounce)[:DEPTH] == [int(x) for x in st_ref.rays_per_bounce[:DEPTH]]
    assert_bit_equal(tile, imgs["fused"][first:first + count], "tile == the same pixels of the full image")


def test_c4_10k_at_1080p_256spp(pt, oracle):
    """configs[3] as stated: 10 092 triangles, 1920x1080, 256 spp, depth 10.  Engines agree on every pixel and on the ray
    count; 256 pixels x 256 spp against the oracle; ray count of a 1 024-pixel range == the oracle's; finite image."""
    V, T, M = _scene(10000)
    sc_pt, sc_o = pt.make_scene(V, T, M), oracle.make_scene(V, T, M)
    spp, n = 256, W * H
    imgs, rays = {}, {}
    for name, engine in (("fused", pt.PT_ENGINE_FUSED), ("wavefront", pt.PT_ENGINE_WAVEFRONT)):
        with pt.PathTracer(W, H) as tr:
            tr.set_scene(*sc_pt)
            tr.set_options(engine=engine, depth=DEPTH, rng_mode=pt.PT_RNG_STRIDED, stride=STRIDE, samples_per_pass=16)
            tr.render(spp)
            imgs[name], rays[name] = tr.readback(), tr.stats().rays
    assert_bit_equal(imgs["wavefront"], imgs["fused"], "10k triangles x 256 spp: wavefront vs fused")
    assert rays["fused"] == rays["wavefront"]
    assert np.isfinite(imgs["fused"]).all() and (imgs["fused"][:, 3] > 0).all()
    p = oracle.params(W, H, depth=DEPTH, spp=spp, rng_mode=oracle.RNG_STRIDED, stride=STRIDE)
    ids = np.unique(np.random.default_rng(256).integers(0, n, 256)).astype(np.uint32)
    ref, _, _, _ = oracle.render(p, scene=sc_o, pixel_ids=ids)
    assert_bit_equal(imgs["fused"][ids], ref, "10k triangles x 256 spp: sampled pixels vs oracle")
    first, count = 400 * W + 100, 1024
    _, _, st_ref, _ = oracle.render(p, scene=sc_o, pixel_first=first, n=count)
    with pt.PathTracer(W, H) as tr:
        tr.set_scene(*sc_pt)
        tr.set_options(engine=pt.PT_ENGINE_FUSED, depth=DEPTH, rng_mode=pt.PT_RNG_STRIDED, stride=STRIDE, pixel_first=first, pixel_count=count)
        tr.render(spp)
        st = tr.stats()
    assert st.rays == st_ref.rays


def test_c3_full_share_of_rank_3(pt, oracle):
    """configs[2]: 3840x2160, 1024 spp over 8 GPUs = 128 spp per rank.  Rank 3's whole share (samples 384..511,
    strided/35): fused == wavefront on all 8 294 400 pixels, 2 000 pixels against the oracle, rays = 2 x pixels x 128."""
    w, h, spp, first = 3840, 2160, 128, 3 * 128
    n = w * h
    imgs = {}
    for name, engine in (("fused", pt.PT_ENGINE_FUSED), ("wavefront", pt.PT_ENGINE_WAVEFRONT)):
        with pt.PathTracer(w, h) as tr:
            tr.set_options(engine=engine, rng_mode=pt.PT_RNG_STRIDED, stride=35, samples_per_pass=8, sample_offset=first)
            tr.render(spp)
            imgs[name] = tr.readback()
            assert tr.stats().rays == 2 * n * spp
    assert_bit_equal(imgs["wavefront"], imgs["fused"], "C3 share of rank 3: wavefront vs fused")
    ids = np.unique(np.random.default_rng(3).integers(0, n, 2000)).astype(np.uint32)
    # the shard's own running mean starts at 0 while its RNG samples start at `first`: fold the oracle's frames one by one
    ref = np.ones((len(ids), 4), dtype=np.float32)
    for s_ in range(spp):
        _, _, _, frame = oracle.render(oracle.params(w, h, spp=1, rng_mode=oracle.RNG_STRIDED, stride=35, frame_first=first + s_),
                                       pixel_ids=ids, want_last_frame=True)
        r = np.float32(1.0) / np.float32(s_ + 1)
        ref = (frame * r + (ref * np.float32(s_)) * r).astype(np.float32)
    assert_bit_equal(imgs["fused"][ids], ref, "C3 share of rank 3: sampled pixels vs oracle")


@pytest.mark.parametrize("n_room,n_ball", [(972, 20000), (300, 1200)])
def test_room_with_a_tessellated_ball_at_1080p(pt, oracle, n_room, n_ball):
    """large walls (normal classes) with a finely tessellated ball (free triangles) -- the mix on which one
    inflation per ray collapsed (0.1 Grays/s).  Two trees now (pt_bvh.hpp build); at 1920x1080: fused (reference streams) ==
    the oracle on 2 000 sampled pixels, wavefront (strided) == the oracle on the same pixels in its own mode, the engines agree on
    every pixel when both run strided streams, ray counts of a pixel range == the oracle's."""
    from test_bvh_adversarial import room_with_a_ball
    V, T, M = room_with_a_ball(n_room, n_ball, 8)
    sc_pt, sc_o = pt.make_scene(V, T, M), oracle.make_scene(V, T, M)
    spp, n, depth = 3, W * H, 8
    stride = 2 + 33 * depth
    ids = np.unique(np.concatenate([np.random.default_rng(n_ball).integers(0, n, 2400), [0, W - 1, n - W, n - 1]])).astype(np.uint32)
    imgs, rays = {}, {}
    for name, engine, strided in (("fused", pt.PT_ENGINE_FUSED, False), ("fused-strided", pt.PT_ENGINE_FUSED, True), ("wavefront", pt.PT_ENGINE_WAVEFRONT, True)):
        with pt.PathTracer(W, H) as tr:
            tr.set_scene(*sc_pt)
            tr.set_options(engine=engine, depth=depth, rng_mode=pt.PT_RNG_STRIDED if strided else pt.PT_RNG_SEQUENTIAL, stride=stride, samples_per_pass=3)
            tr.render(spp)
            imgs[name], rays[name] = tr.readback(), tr.stats().rays
    assert_bit_equal(imgs["wavefront"], imgs["fused-strided"], f"room {n_room} + ball: wavefront vs fused, strided streams")
    assert rays["wavefront"] == rays["fused-strided"] and np.isfinite(imgs["fused"]).all()
    for name, mode in (("fused", oracle.RNG_SEQUENTIAL), ("wavefront", oracle.RNG_STRIDED)):
        ref, _, _, _ = oracle.render(oracle.params(W, H, depth=depth, spp=spp, rng_mode=mode, stride=stride), scene=sc_o, pixel_ids=ids)
        assert_bit_equal(imgs[name][ids], ref, f"room {n_room} + ball {len(T) - n_room}: {name} vs oracle on sampled pixels")
    first, count = 500 * W + 700, 2048                           # a range that crosses the ball
    p = oracle.params(W, H, depth=depth, spp=spp, rng_mode=oracle.RNG_STRIDED, stride=stride)
    _, _, st_ref, _ = oracle.render(p, scene=sc_o, pixel_first=first, n=count)
    with pt.PathTracer(W, H) as tr:
        tr.set_scene(*sc_pt)
        tr.set_options(engine=pt.PT_ENGINE_WAVEFRONT, depth=depth, rng_mode=pt.PT_RNG_STRIDED, stride=stride, samples_per_pass=3, pixel_first=first, pixel_count=count)
        tr.render(spp)
        st = tr.stats()
    assert st.rays == st_ref.rays and list(st.rays_per_bounce)[:depth] == [int(x) for x in st_ref.rays_per_bounce[:depth]]


@pytest.mark.parametrize("which", ["all-materials", "mirror-extended"])
def test_c5_at_1080p_64spp(pt, oracle, which):
    """BASELINE.json configs[4] at its stated size -- all material types, depth 16, 1920x1080 x 64 spp (and the
    mirror scene under extended shading).  Every form of both engines agrees on every pixel: fused in step, fused out of step
    (forced), wavefront strided (batches of 16), and -- on the reference's sequential streams -- fused in step, fused out of step
    and the wavefront engine's rounds; 2 000 sampled pixels of each stream mode == the oracle; per-bounce ray counts of a
    4 096-pixel range == the oracle's."""
    extended = which == "mirror-extended"
    V, T, M = scenes.mirror_scene() if extended else scenes.all_materials_scene()
    sc_pt, sc_o = pt.make_scene(V, T, M), oracle.make_scene(V, T, M)
    spp, n, depth = 64, W * H, 16
    shading = 1 if extended else 0
    stride = 2 + (2 if extended else 33) * depth
    ids = np.unique(np.concatenate([np.random.default_rng(64).integers(0, n, 2400), [0, W - 1, n - W, n - 1]])).astype(np.uint32)
    lib = pt.load_library()

    def render(engine, strided, oos=False, per_pass=16, **extra):
        import os
        if oos:
            os.environ["PT_FUSED_OUT_OF_STEP"] = "1"
        try:
            with pt.PathTracer(W, H) as tr:
                tr.set_scene(*sc_pt)
                tr.set_options(engine=engine, depth=depth, shading=shading, rng_mode=pt.PT_RNG_STRIDED if strided else pt.PT_RNG_SEQUENTIAL,
                               stride=stride, samples_per_pass=per_pass, **extra)
                tr.render(spp)
                st = tr.stats()
                return tr.readback(), tr.readback_rng(), st
        finally:
            os.environ.pop("PT_FUSED_OUT_OF_STEP", None)

    for strided, mode in ((True, oracle.RNG_STRIDED), (False, oracle.RNG_SEQUENTIAL)):
        base, base_rng, base_st = render(pt.PT_ENGINE_FUSED, strided)
        for name, args in (("fused out of step", dict(engine=pt.PT_ENGINE_FUSED, oos=True)),
                           ("wavefront", dict(engine=pt.PT_ENGINE_WAVEFRONT, per_pass=16 if strided else 64))):
            img, rng, st = render(strided=strided, **args)
            assert_bit_equal(img, base, f"{which}, {'strided' if strided else 'sequential'} streams: {name} vs fused in step")
            assert st.rays == base_st.rays
            if not strided:
                assert np.array_equal(rng, base_rng)
        ref, _, _, _ = oracle.render(oracle.params(W, H, depth=depth, spp=spp, shading=shading, rng_mode=mode, stride=stride), scene=sc_o, pixel_ids=ids)
        assert_bit_equal(base[ids], ref, f"{which}, {'strided' if strided else 'sequential'} streams: sampled pixels vs oracle")
    first, count = 333 * W + 900, 4096
    p = oracle.params(W, H, depth=depth, spp=spp, shading=shading, rng_mode=oracle.RNG_STRIDED, stride=stride)
    _, _, st_ref, _ = oracle.render(p, scene=sc_o, pixel_first=first, n=count)
    with pt.PathTracer(W, H) as tr:
        tr.set_scene(*sc_pt)
        tr.set_options(engine=pt.PT_ENGINE_WAVEFRONT, depth=depth, shading=shading, rng_mode=pt.PT_RNG_STRIDED, stride=stride, samples_per_pass=16,
                       pixel_first=first, pixel_count=count)
        tr.render(spp)
        st = tr.stats()
    assert st.rays == st_ref.rays and list(st.rays_per_bounce)[:depth] == [int(x) for x in st_ref.rays_per_bounce[:depth]]

"""PT_SHADING_EXTENDED (opt-in): REFLECTIVE mirrors, DIFFUSE samples the cosine-weighted hemisphere,
every bounce finds its own closest hit.  The reference has no such mode (its REFLECTIVE absorbs, kernels.cu:91-93,
and its author flags the sampler as biased, Functions.cuh:92-96), so there is nothing of the reference to pin it to:
PARITY UNPINNED -- the oracle's pto_stage_shade_extended is the definition, checked here for what a mirror and a
cosine sampler must do, and the GPU must reproduce it bit for bit (no tolerance)."""
import ctypes as C

import numpy as np
import pytest

import scenes


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def test_sincos_turn_accuracy_and_quadrants(oracle):
    lib = oracle.lib()
    lib.pto_sincos_turn.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    lib.pto_sincos_turn.restype = None
    worst = 0.0
    for u in np.linspace(0.0, 1.0, 4097, dtype=np.float32):
        c, s = C.c_float(), C.c_float()
        lib.pto_sincos_turn(float(u), C.byref(c), C.byref(s))
        worst = max(worst, abs(c.value - np.cos(2 * np.pi * float(u))), abs(s.value - np.sin(2 * np.pi * float(u))))
    assert worst < 5e-6
    for u, want in ((0.0, (1, 0)), (0.25, (0, 1)), (0.5, (-1, 0)), (0.75, (0, -1)), (1.0, (1, 0))):
        c, s = C.c_float(), C.c_float()
        lib.pto_sincos_turn(u, C.byref(c), C.byref(s))
        assert (c.value, s.value) == want


def test_extended_image_properties(oracle):
    """Energy and path statistics of the extended mode on the mirror scene: paths really bounce (the reference mode dies
    after ~2 rays because of the stale hit distance), draws = 2 per pixel-sample + 2 per cosine bounce, all finite."""
    V, T, M = scenes.mirror_scene()
    w, h, spp = 96, 64, 4
    ref, _, st_ref, _ = oracle.render(oracle.params(w, h, depth=16, spp=spp), scene=oracle.make_scene(V, T, M))
    ext, _, st, _ = oracle.render(oracle.params(w, h, depth=16, spp=spp, shading=oracle.SHADING_EXTENDED), scene=oracle.make_scene(V, T, M))
    assert np.isfinite(ext).all() and (ext[:, :3] >= 0).all()
    assert st.rays > 3 * st_ref.rays and st.rays_per_bounce[15] > 0
    assert (st.draws - 2 * w * h * spp) % 2 == 0 and st.draws < 2 * w * h * spp + 2 * st.rays
    # closed, bright box: cosine sampling with albedo < 1 keeps every pixel bounded by the strongest emitter
    assert ext[:, :3].max() <= 10.0 + 1e-3


def test_cosine_sampler_statistics(oracle):
    """The oracle's shade stage on its own: cosine-sampled directions are unit, in the normal's hemisphere and have
    E[cos] = 2/3; the mirror obeys r = d - 2 (d.n^) n^, draws nothing and scales throughput by reflectivity * colour."""
    lib = oracle.lib()
    P, M4 = oracle.Path, oracle.Material
    lib.pto_stage_shade_extended.argtypes = [C.POINTER(P), C.POINTER(M4)]
    lib.pto_stage_shade_extended.restype = None
    mats = (M4 * 2)()
    mats[0].type = 0; mats[0].color = oracle.F4(0.5, 0.25, 1.0, 1.0)
    mats[1].type = 1; mats[1].color = oracle.F4(0.5, 0.5, 0.5, 1.0); mats[1].reflectivity = 0.5
    n = np.array([0.3, 2.0, -0.4], dtype=np.float32)            # un-normalised on purpose
    nh = n / np.linalg.norm(n)
    cos_sum, cnt = 0.0, 4000
    rng = 2463534242
    for _ in range(cnt):
        p = P()
        p.origin = oracle.F4(0, 1, 0, 1); p.direction = oracle.F4(0.1, -1.0, 0.2, 0); p.normal = oracle.F4(*n, 0)
        p.step = oracle.F4(1, 1, 1, 1); p.t = 1.0; p.materialIdx = 0; p.rng = rng; p.alive = 1
        lib.pto_stage_shade_extended(C.byref(p), mats)
        rng = p.rng
        d = np.array([p.direction.x, p.direction.y, p.direction.z])
        assert abs(np.linalg.norm(d) - 1.0) < 2e-5 and d @ nh > -1e-6
        assert (p.step.x, p.step.y, p.step.z, p.step.w) == (0.5, 0.25, 1.0, 1.0) and p.draws == 2
        cos_sum += d @ nh
    assert abs(cos_sum / cnt - 2.0 / 3.0) < 0.02
    # mirror: r = d - 2 (d.n^) n^ ; throughput = reflectivity * colour ; no draws
    p = P()
    p.origin = oracle.F4(0, 1, 0, 1); p.direction = oracle.F4(0.1, -1.0, 0.2, 0); p.normal = oracle.F4(*n, 0)
    p.step = oracle.F4(1, 1, 1, 1); p.t = 1.0; p.materialIdx = 1; p.rng = 7; p.alive = 1
    lib.pto_stage_shade_extended(C.byref(p), mats)
    d0 = np.array([0.1, -1.0, 0.2])
    want = d0 - 2 * (d0 @ nh) * nh
    assert np.allclose([p.direction.x, p.direction.y, p.direction.z], want, atol=1e-6)
    assert p.rng == 7 and p.draws == 0 and (p.step.x, p.step.w) == (0.25, 0.5) and p.alive == 1


ENGINES = [("fused", 1), ("wavefront", 0)]


@pytest.mark.gpu
@pytest.mark.parametrize("name,engine", ENGINES)
@pytest.mark.parametrize("scene,w,h,spp,depth", [("mirror", 160, 90, 4, 16), ("default", 96, 64, 3, 10), ("synthetic", 96, 54, 2, 10)])
def test_gpu_extended_equals_oracle(pt, oracle, name, engine, scene, w, h, spp, depth):
    V, T, M = {"mirror": scenes.mirror_scene, "default": scenes.default_scene, "synthetic": lambda: scenes.synthetic_scene(1500)}[scene]()
    if scene == "synthetic":
        M = [(m[0], m[1], 0.8) for m in M]
    ref, ref_rng, st_ref, _ = oracle.render(oracle.params(w, h, depth=depth, spp=spp, shading=oracle.SHADING_EXTENDED),
                                            scene=oracle.make_scene(V, T, M))
    with pt.PathTracer(w, h) as tr:
        tr.set_scene(*pt.make_scene(V, T, M))
        tr.set_options(engine=engine, depth=depth, shading=pt.PT_SHADING_EXTENDED)
        tr.render(spp)
        img, rng, st = tr.readback(), tr.readback_rng(), tr.stats()
    diff = (bits(img) != bits(ref)).any(axis=1)
    assert not diff.any(), f"{name} {scene}: {int(diff.sum())} of {len(diff)} pixels differ, first {int(np.argmax(diff))}: {img[np.argmax(diff)]} vs {ref[np.argmax(diff)]}"
    assert np.array_equal(rng, ref_rng) and st.rays == st_ref.rays
    if engine == 0:
        assert list(st.rays_per_bounce) == list(st_ref.rays_per_bounce)


@pytest.mark.gpu
@pytest.mark.parametrize("name,engine", ENGINES)
def test_gpu_extended_strided_batches(pt, oracle, name, engine):
    """Strided RNG (stride >= 2 + 2*depth) with the wavefront engine batching samples per pass; also BVH off/on agree."""
    V, T, M = scenes.mirror_scene()
    w, h, spp, depth = 128, 72, 6, 12
    ref, _, st_ref, _ = oracle.render(oracle.params(w, h, depth=depth, spp=spp, rng_mode=oracle.RNG_STRIDED, stride=26, shading=oracle.SHADING_EXTENDED),
                                      scene=oracle.make_scene(V, T, M))
    with pt.PathTracer(w, h) as tr:
        tr.set_scene(*pt.make_scene(V, T, M))
        tr.set_options(engine=engine, depth=depth, rng_mode=pt.PT_RNG_STRIDED, stride=26, samples_per_pass=4, shading=pt.PT_SHADING_EXTENDED)
        tr.render(spp)
        img, st = tr.readback(), tr.stats()
    assert np.array_equal(bits(img), bits(ref)) and st.rays == st_ref.rays


@pytest.mark.gpu
def test_unknown_shading_is_refused(pt):
    with pt.PathTracer(16, 16) as tr:
        with pytest.raises(pt.PathTracerError, match="unknown shading"):
            tr.set_options(shading=7)

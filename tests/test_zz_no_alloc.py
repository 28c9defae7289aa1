"""pt_hip_render_wavefront_noalloc (include/pt_hip.h): the wavefront render whose queues live in the library's static
device array, for callers that cannot call hipMalloc.  Called through the C-ABI and compared with the existing entry
point, pt_hip_render_wavefront on a caller-allocated workspace: running mean, RNG streams and statistics bit for bit.
Two calls in a row on the same GPU share the one array, the second with a smaller input (other queue geometry: the
first call's contents lie about in the array), and a size beyond the array's is refused.

Collected last (test_zz_*): the round's end runs pytest -x."""
import ctypes as C

import numpy as np
import pytest

import scenes
from gpu_util import DeviceBuffer, make_frame, sync, upload_scene

pytestmark = pytest.mark.gpu


def _render_both(pt, oracle, lib, scene, w, h, n_samples, per_pass, depth, strided, seed):
    """-> [(mean bits, rng, stats words)] of the existing entry point and of the no-alloc one, same inputs"""
    V, T, M = scene
    n = w * h
    scene_dev = upload_scene(pt, lib, pt.make_scene(V, T, M))
    stride = 2 + 33 * depth
    fr = make_frame(pt, oracle, w, h, depth=depth, seed=seed, rng_mode=1 if strided else 0, stride=stride, flags=scene_dev.frame_flags)
    in_flight = min(per_pass, n_samples) if strided else 1
    ws_bytes = lib.pt_hip_wavefront_workspace_bytes(n, in_flight, depth)
    assert ws_bytes <= pt.PT_NOALLOC_SCRATCH_BYTES, "test sizes must fit the static scratch"
    results = []
    for noalloc in (False, True):
        rng = DeviceBuffer(lib, n * 4)
        assert lib.pt_hip_rng_seed(C.byref(fr), rng.ptr, None) == 0, lib.pt_hip_last_error()
        mean = DeviceBuffer(lib, n * 16).upload(np.ones(n * 4, dtype=np.float32))
        stats = DeviceBuffer(lib, pt.PT_STATS_WORDS * 8).upload(np.zeros(pt.PT_STATS_WORDS, dtype=np.uint64))
        if noalloc:
            rc = lib.pt_hip_render_wavefront_noalloc(C.byref(fr), scene_dev.ptr, len(T), len(M), 3, 0, n_samples, per_pass,
                                                     rng.ptr, mean.ptr, stats.ptr, None)
        else:
            # 0xA5 everywhere: the entry points must not rely on a zeroed workspace
            ws = DeviceBuffer(lib, ws_bytes)
            assert lib.pt_hip_memset(ws.ptr, 0xA5, ws_bytes, None) == 0
            rc = lib.pt_hip_render_wavefront(C.byref(fr), scene_dev.ptr, len(T), len(M), 3, 0, n_samples, per_pass,
                                             rng.ptr, mean.ptr, ws.ptr, ws_bytes, stats.ptr, None)
        assert rc == 0, lib.pt_hip_last_error()
        sync(lib)
        results.append((mean.download(np.uint32).reshape(n, 4), rng.download(np.uint32), stats.download(np.uint64)))
    return results


# (scene, w, h, samples, per pass, depth, strided): LDS-resident scenes take the one-launch pass (k_bounces), the
# 2500-triangle scene the staged BVH pipeline; first the larger input, then the smaller one
CASES = [
    ("materials", [(320, 200, 6, 3, 16, True), (96, 64, 5, 2, 16, True)]),
    ("materials-sequential", [(640, 480, 3, 3, 12, False), (133, 71, 4, 4, 12, False)]),
    ("synthetic2500", [(256, 160, 4, 2, 10, True), (64, 40, 3, 3, 10, True)]),
    ("default", [(640, 480, 2, 1, 10, True), (40, 24, 7, 7, 10, True)]),
]


@pytest.mark.parametrize("name,calls", CASES, ids=[c[0] for c in CASES])
def test_noalloc_equals_workspace_entry_point(pt, oracle, name, calls):
    lib = pt.load_library()
    scene = {"materials": scenes.all_materials_scene, "materials-sequential": scenes.all_materials_scene,
             "synthetic2500": lambda: scenes.synthetic_scene(2500), "default": scenes.default_scene}[name]()
    for k, (w, h, n_samples, per_pass, depth, strided) in enumerate(calls):
        (mean_a, rng_a, st_a), (mean_b, rng_b, st_b) = _render_both(pt, oracle, lib, scene, w, h, n_samples, per_pass, depth, strided, 1337 + k)
        bad = (mean_a != mean_b).any(axis=1)
        assert not bad.any(), f"{name} call {k} ({w}x{h}): {int(bad.sum())} of {w * h} pixels differ from pt_hip_render_wavefront"
        assert np.array_equal(rng_a, rng_b), f"{name} call {k}: RNG streams differ"
        assert np.array_equal(st_a, st_b), f"{name} call {k}: statistics differ"
        assert st_b[2] == w * h * n_samples and st_b[0] >= st_b[2]          # paths generated, rays >= paths
        assert np.isfinite(mean_b.view(np.float32)).all()


def test_noalloc_matches_the_oracle(pt, oracle):
    """and the image is the reference's: all-materials scene, sequential streams, against the CPU oracle"""
    lib = pt.load_library()
    V, T, M = scenes.all_materials_scene()
    w, h, spp, depth = 160, 90, 4, 16
    scene_dev = upload_scene(pt, lib, pt.make_scene(V, T, M))
    fr = make_frame(pt, oracle, w, h, depth=depth, flags=scene_dev.frame_flags)
    n = w * h
    rng = DeviceBuffer(lib, n * 4)
    assert lib.pt_hip_rng_seed(C.byref(fr), rng.ptr, None) == 0
    mean = DeviceBuffer(lib, n * 16).upload(np.ones(n * 4, dtype=np.float32))
    rc = lib.pt_hip_render_wavefront_noalloc(C.byref(fr), scene_dev.ptr, len(T), len(M), 0, 0, spp, spp, rng.ptr, mean.ptr, None, None)
    assert rc == 0, lib.pt_hip_last_error()
    sync(lib)
    ref, ref_rng, _, _ = oracle.render(oracle.params(w, h, depth=depth, spp=spp), scene=oracle.make_scene(V, T, M))
    assert np.array_equal(mean.download(np.uint32).reshape(n, 4), ref.view(np.uint32))
    assert np.array_equal(rng.download(np.uint32), ref_rng)


def test_noalloc_rejects_sizes_beyond_the_scratch(pt, oracle):
    lib = pt.load_library()
    V, T, M = scenes.default_scene()
    scene_dev = upload_scene(pt, lib, pt.make_scene(V, T, M))
    w, h = 1920, 1080
    fr = make_frame(pt, oracle, w, h, rng_mode=1, stride=35)
    assert lib.pt_hip_wavefront_workspace_bytes(w * h, 16, 10) > pt.PT_NOALLOC_SCRATCH_BYTES
    rng = DeviceBuffer(lib, w * h * 4)
    mean = DeviceBuffer(lib, w * h * 16)
    rc = lib.pt_hip_render_wavefront_noalloc(C.byref(fr), scene_dev.ptr, len(T), len(M), 0, 0, 16, 16, rng.ptr, mean.ptr, None, None)
    assert rc == -5, rc                                          # PT_ERR_LIMIT
    msg = lib.pt_hip_last_error().decode()
    assert "static scratch" in msg and str(pt.PT_NOALLOC_SCRATCH_BYTES) in msg

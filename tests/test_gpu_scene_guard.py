"""The scene guard: PT_FRAME_SCENE_BVH in pt_frame.flags is the caller's word, the launchers cannot look into
device memory -- so the kernels compare it with the blob's own header and refuse on a mismatch instead of walking nodes
that are not there (out-of-bounds reads) or looping over 10^4 triangles of a blob that carries a BVH (a forgotten flag).
The refusal surfaces as PT_ERR_STATE at the next synchronising entry point, once; the library goes on working."""
import ctypes as C

import numpy as np
import pytest

import scenes
from gpu_util import DeviceBuffer, make_frame, upload_scene

pytestmark = pytest.mark.gpu

PT_ERR_STATE = -4


def _fused(pt, lib, fr, scene_dev, n_tris, n_mats, rng, mean, stats):
    rc = lib.pt_hip_render_fused(C.byref(fr), scene_dev.ptr, n_tris, n_mats, 0, 0, 1, rng.ptr, mean.ptr, stats.ptr, None)
    assert rc == 0, lib.pt_hip_last_error()
    return lib.pt_hip_stream_synchronize(None)


@pytest.mark.parametrize("which", ["flag-set-no-bvh", "flag-clear-bvh"])
def test_flag_and_blob_in_disagreement_are_refused_not_walked(pt, oracle, which):
    lib = pt.load_library()
    w, h, depth = 64, 48, 4
    if which == "flag-set-no-bvh":
        V, T, M = scenes.all_materials_scene()            # 48 triangles: packed without a BVH
    else:
        V, T, M = scenes.synthetic_scene(600)             # above the threshold: packed with one
    scene_dev = upload_scene(pt, lib, pt.make_scene(V, T, M))
    right = scene_dev.frame_flags
    assert (right != 0) == (which == "flag-clear-bvh")
    wrong = right ^ pt.PT_FRAME_SCENE_BVH
    n = w * h
    rng = DeviceBuffer(lib, n * 4)
    mean = DeviceBuffer(lib, n * 16).upload(np.zeros(n * 4, dtype=np.float32))
    stats = DeviceBuffer(lib, pt.PT_STATS_WORDS * 8).upload(np.zeros(pt.PT_STATS_WORDS, dtype=np.uint64))
    fr_ok = make_frame(pt, oracle, w, h, depth=depth, flags=right)
    fr_bad = make_frame(pt, oracle, w, h, depth=depth, flags=wrong)
    assert lib.pt_hip_rng_seed(C.byref(fr_ok), rng.ptr, None) == 0
    assert lib.pt_hip_stream_synchronize(None) == 0, lib.pt_hip_last_error()
    rng0 = rng.download(np.uint32)

    # fused engine with the wrong word: the launch is accepted (the launcher cannot know), the kernel refuses, the sync says so
    assert _fused(pt, lib, fr_bad, scene_dev, len(T), len(M), rng, mean, stats) == PT_ERR_STATE
    assert b"PT_FRAME_SCENE_BVH" in lib.pt_hip_last_error()
    # ... and nothing was touched: streams and mean as they were
    assert np.array_equal(rng.download(np.uint32), rng0)
    assert not mean.download(np.float32).any()
    # reported once: the next synchronisation is clean
    assert lib.pt_hip_stream_synchronize(None) == 0

    # the extend stage alone (bounce 0), wrong word: refused as well
    P = n
    G = lib.pt_hip_queue_blocks(P)
    qb = DeviceBuffer(lib, P * 16)
    counts = DeviceBuffer(lib, (depth + 2) * G * 4)
    hit = DeviceBuffer(lib, P * 8).upload(np.full(P * 2, 0xABABABAB, dtype=np.uint32))
    Q = pt.Queue(a=None, b=qb.ptr, c=None, pid=None)
    assert lib.pt_hip_generate(C.byref(fr_ok), 0, 1, rng.ptr, None, None, 0, Q, None, P, counts.ptr, None) == 0, lib.pt_hip_last_error()
    assert lib.pt_hip_extend(C.byref(fr_bad), 0, scene_dev.ptr, len(T), len(M), Q, counts.ptr, P, hit.ptr, None) == 0, lib.pt_hip_last_error()
    assert lib.pt_hip_stream_synchronize(None) == PT_ERR_STATE
    assert (hit.download(np.uint32) == 0xABABABAB).all()

    # the wavefront engine, wrong word
    ws_bytes = lib.pt_hip_wavefront_workspace_bytes(n, 1, depth)
    ws = DeviceBuffer(lib, ws_bytes)
    rc = lib.pt_hip_render_wavefront(C.byref(fr_bad), scene_dev.ptr, len(T), len(M), 0, 0, 1, 1, rng.ptr, mean.ptr, ws.ptr, ws_bytes, stats.ptr, None)
    assert rc == 0, lib.pt_hip_last_error()
    assert lib.pt_hip_stream_synchronize(None) == PT_ERR_STATE

    # with the right word everything works and equals the oracle (a refused pass leaves its outputs undefined: start afresh)
    rng.upload(rng0)
    mean.upload(np.zeros(n * 4, dtype=np.float32))
    assert _fused(pt, lib, fr_ok, scene_dev, len(T), len(M), rng, mean, stats) == 0, lib.pt_hip_last_error()
    ref, ref_rng, _, _ = oracle.render(oracle.params(w, h, depth=depth, spp=1), scene=oracle.make_scene(V, T, M))
    assert np.array_equal(mean.download(np.float32).view(np.uint32).reshape(n, 4), ref.view(np.uint32))
    assert np.array_equal(rng.download(np.uint32), ref_rng)

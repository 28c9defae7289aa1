"""The C-ABI without a GPU: the library loads, exports every symbol include/*.h declares, the
reference PODs have the reference's layout, host-only entry points work, and argument checks
fail with a clear error before anything touches the device."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pt_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(pt):
    lib = pt.load_library()
    names = declared_functions("pt_hip.h") + declared_functions("pt_plugin.h") + ["Create", "Destroy"]
    assert len(names) > 40
    for name in names:
        assert hasattr(lib, name), f"{name} is declared in include/ but not exported"
        assert name in pt.SYMBOLS, f"{name} has no ctypes prototype in pathtracer_amd.SYMBOLS"


def test_no_undeclared_pt_exports(pt):
    """Everything the binding prototypes is declared in a header (headers are the contract)."""
    declared = set(declared_functions("pt_hip.h") + declared_functions("pt_plugin.h") + ["Create", "Destroy"])
    assert set(pt.SYMBOLS) == declared


def test_missing_library_fails_loudly(pt, tmp_path):
    with pytest.raises(pt.PathTracerError, match="no CPU fallback"):
        pt.load_library(str(tmp_path / "libPT_Nope.so"))


def test_reference_struct_layouts(pt):
    # PT_CudaSimple/Structures.cuh:43-69, include/CameraData.h:14-18
    assert C.sizeof(pt.Vertex) == 16 and C.sizeof(pt.Triangle) == 16
    assert C.sizeof(pt.Material) == 48 and pt.Material.color.offset == 16 and pt.Material.reflectivity.offset == 32
    assert C.sizeof(pt.CameraData) == 36
    cam = pt.CameraData()
    assert list(cam.position) == [0, 0, 0, 1] and list(cam.rotation) == [0, 0, 0, 1] and cam.fieldOfView == 90.0
    assert C.sizeof(pt.Frame) == 32 + 64


def test_scene_pack_matches_reference_arithmetic(pt):
    lib = pt.load_library()
    v, t, m = pt.default_scene()
    n = lib.pt_hip_scene_blob_bytes(4, 2)
    assert n == 32 + 4 * 64 + 2 * 32
    blob = (C.c_uint8 * n)()
    rc = lib.pt_hip_scene_pack(C.cast(v, C.c_void_p), 8, C.cast(t, C.c_void_p), 4, C.cast(m, C.c_void_p), 2, C.cast(blob, C.c_void_p), n)
    assert rc == 0, lib.pt_hip_last_error()
    raw = np.frombuffer(blob, dtype=np.uint8)
    hdr = raw[:32].view(np.uint32)
    assert hdr[0] == 4 and hdr[1] == 2 and hdr[2] == 32 and hdr[3] == 32 + 256
    tri = raw[32:32 + 256].view(np.float32).reshape(4, 16)
    f = np.float32
    # triangle 2 = (4,5,6): v0=(-0.2,0.5,0.5) e0=v1-v0 e1=v2-v0, n = cross(e0,e1)  (Functions.cuh:65-66,88)
    v0 = np.array([f(-0.2), f(0.5), f(0.5)]); v1 = np.array([f(0.2), f(0.5), f(0.5)]); v2 = np.array([f(0.2), f(0.5), f(0.9)])
    e0, e1 = v1 - v0, v2 - v0
    nn = np.array([e0[1] * e1[2] - e0[2] * e1[1], e0[2] * e1[0] - e0[0] * e1[2], e0[0] * e1[1] - e0[1] * e1[0]], dtype=f)
    assert np.array_equal(tri[2, 0:3], v0) and np.array_equal(tri[2, 3:6], e0) and np.array_equal(tri[2, 6:9], e1)
    assert np.array_equal(tri[2, 9:12].view(np.uint32), nn.view(np.uint32))
    assert tri[2, 12:13].view(np.uint32)[0] == 1
    assert np.array_equal(tri[0, 9:12], np.array([0, 0, -1], dtype=f))        # the wall's normal
    mats = raw[32 + 256:].view(np.float32).reshape(2, 8)
    assert mats[1, 0:1].view(np.uint32)[0] == pt.PT_MAT_EMISSIVE and list(mats[1, 4:8]) == [10, 10, 10, 10]


def test_scene_pack_rejects_bad_indices(pt):
    lib = pt.load_library()
    v, t, m = pt.make_scene([(0, 0, 0, 1), (1, 0, 0, 1), (0, 1, 0, 1)], [(0, 1, 3, 0)], [(0, (1, 1, 1, 1))])
    n = lib.pt_hip_scene_blob_bytes(1, 1)
    blob = (C.c_uint8 * n)()
    rc = lib.pt_hip_scene_pack(C.cast(v, C.c_void_p), 3, C.cast(t, C.c_void_p), 1, C.cast(m, C.c_void_p), 1, C.cast(blob, C.c_void_p), n)
    assert rc == -1 and b"vertex index out of range" in lib.pt_hip_last_error()
    v, t, m = pt.make_scene([(0, 0, 0, 1), (1, 0, 0, 1), (0, 1, 0, 1)], [(0, 1, 2, 5)], [(0, (1, 1, 1, 1))])
    rc = lib.pt_hip_scene_pack(C.cast(v, C.c_void_p), 3, C.cast(t, C.c_void_p), 1, C.cast(m, C.c_void_p), 1, C.cast(blob, C.c_void_p), n)
    assert rc == -1 and b"material index out of range" in lib.pt_hip_last_error()
    rc = lib.pt_hip_scene_pack(C.cast(v, C.c_void_p), 3, C.cast(t, C.c_void_p), 1, C.cast(m, C.c_void_p), 1, C.cast(blob, C.c_void_p), n - 1)
    assert rc == -3
    # empty scene packs (every ray misses)
    n0 = lib.pt_hip_scene_blob_bytes(0, 0)
    assert lib.pt_hip_scene_pack(None, 0, None, 0, None, 0, C.cast(blob, C.c_void_p), n0) == 0


def test_argument_checks_precede_device_work(pt):
    lib = pt.load_library()
    assert lib.pt_hip_rng_seed(None, None, None) == -1 and b"frame is null" in lib.pt_hip_last_error()
    fr = pt.Frame(width=0, height=10, depth=10)
    assert lib.pt_hip_rng_seed(C.byref(fr), None, None) == -1
    fr = pt.Frame(width=8, height=8, pixel_first=60, pixel_count=8, depth=10)
    assert lib.pt_hip_rng_seed(C.byref(fr), None, None) == -1 and b"pixel range" in lib.pt_hip_last_error()
    fr = pt.Frame(width=8, height=8, pixel_first=0, pixel_count=64, depth=33)
    assert lib.pt_hip_rng_seed(C.byref(fr), None, None) == -1 and b"depth" in lib.pt_hip_last_error()
    fr = pt.Frame(width=8, height=8, pixel_first=0, pixel_count=64, depth=10, rng_mode=1, stride=0)
    assert lib.pt_hip_rng_seed(C.byref(fr), None, None) == -1 and b"stride" in lib.pt_hip_last_error()
    assert lib.pt_hip_combine_shards(None, 4, 16, None, None) == -1
    assert lib.pt_update(None, None) == -1 and b"null handle" in lib.pt_last_error()
    assert lib.pt_render(None, 1) == -1
    assert lib.pt_readback(None, None, 0) == -1
    assert not lib.pt_device_mean(None)
    # Create validates before touching the device
    cam = pt.CameraData()
    assert not lib.Create(0, 0, 10, C.byref(cam))
    assert not lib.Create(0, 10, 10, None)


def test_workspace_size_accounts_for_every_buffer(pt):
    lib = pt.load_library()
    paths = 1000 * 4
    n = lib.pt_hip_wavefront_workspace_bytes(1000, 4, 10)
    # 2 queues x (3 float4 + pid) + hit + radiance, 256-byte aligned pieces, + counters + matrices
    assert n >= paths * (2 * (48 + 4) + 8 + 16)
    assert n % 256 == 0

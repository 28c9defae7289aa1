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
    assert C.sizeof(pt.Frame) == 56 + 64


def test_scene_pack_matches_reference_arithmetic(pt):
    lib = pt.load_library()
    v, t, m = pt.default_scene()
    n = lib.pt_hip_scene_blob_bytes(4, 2)
    assert n == 64 + 4 * 64 + 2 * 32            # 64-byte header, 64 B per triangle, 32 B per material (no BVH up to PT_BVH_DEFAULT_MIN_TRIANGLES triangles)
    blob = (C.c_uint8 * n)()
    rc = lib.pt_hip_scene_pack(C.cast(v, C.c_void_p), 8, C.cast(t, C.c_void_p), 4, C.cast(m, C.c_void_p), 2, C.cast(blob, C.c_void_p), n)
    assert rc == 0, lib.pt_hip_last_error()
    raw = np.frombuffer(blob, dtype=np.uint8)
    hdr = raw[:64].view(np.uint32)
    assert hdr[0] == 4 and hdr[1] == 2 and hdr[2] == 64 and hdr[3] == 64 + 256 and hdr[6] == 0     # n_nodes = 0
    tri = raw[64:64 + 256].view(np.float32).reshape(4, 16)
    f = np.float32
    # triangle 2 = (4,5,6): v0=(-0.2,0.5,0.5) e0=v1-v0 e1=v2-v0, n = cross(e0,e1)  (Functions.cuh:65-66,88)
    v0 = np.array([f(-0.2), f(0.5), f(0.5)]); v1 = np.array([f(0.2), f(0.5), f(0.5)]); v2 = np.array([f(0.2), f(0.5), f(0.9)])
    e0, e1 = v1 - v0, v2 - v0
    nn = np.array([e0[1] * e1[2] - e0[2] * e1[1], e0[2] * e1[0] - e0[0] * e1[2], e0[0] * e1[1] - e0[1] * e1[0]], dtype=f)
    assert np.array_equal(tri[2, 0:3], v0) and np.array_equal(tri[2, 3:6], e0) and np.array_equal(tri[2, 6:9], e1)
    assert np.array_equal(tri[2, 9:12].view(np.uint32), nn.view(np.uint32))
    assert tri[2, 12:13].view(np.uint32)[0] == 1
    assert np.array_equal(tri[0, 9:12], np.array([0, 0, -1], dtype=f))        # the wall's normal
    mats = raw[64 + 256:].view(np.float32).reshape(2, 8)
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
    fr = pt.Frame(width=8, height=8, pixel_first=0, pixel_count=64, depth=10)
    assert lib.pt_hip_build_need_masks(C.byref(fr), None, 4, 2, None, None) == -1 and b"null pointer" in lib.pt_hip_last_error()
    assert lib.pt_hip_build_need_masks(None, None, 4, 2, None, None) == -1
    # one 64-bit word per 64 pixels, in whole blocks of 256 pixels
    assert [lib.pt_hip_need_masks_bytes(n) for n in (1, 256, 257, 1920 * 1080)] == [32, 32, 64, 8100 * 32]
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


def test_bvh_blob_is_a_valid_hierarchy(pt):
    """Scenes above PT_BVH_DEFAULT_MIN_TRIANGLES (96) triangles carry a BVH (pathtracer_amd/csrc/pt_bvh.hpp): every triangle is in
    exactly one leaf, every child box (on its 16-bit grid) contains its triangles and lies in its parent's."""
    import scenes
    lib = pt.load_library()
    V, T, M = scenes.synthetic_scene(700)
    v, t, m = pt.make_scene(V, T, M)
    n = lib.pt_hip_scene_blob_bytes(len(T), len(M))
    blob = (C.c_uint8 * n)()
    assert lib.pt_hip_scene_pack(C.cast(v, C.c_void_p), len(V), C.cast(t, C.c_void_p), len(T), C.cast(m, C.c_void_p), len(M),
                                 C.cast(blob, C.c_void_p), n) == 0, lib.pt_hip_last_error()
    raw = np.frombuffer(blob, dtype=np.uint8)
    hdr_u = raw[:64].view(np.uint32)
    hdr_f = raw[:64].view(np.float32)
    n_tris, n_nodes, bvh_off, leaf_off = int(hdr_u[0]), int(hdr_u[6]), int(hdr_u[7]), int(hdr_u[8])
    assert n_tris == len(T) and 0 < n_nodes <= n_tris and bvh_off % 64 == 0 and leaf_off == bvh_off + 64 * n_tris
    assert hdr_f[12] > 0 and hdr_f[13] > 0                       # radius, kappa_root
    # QNode4, 64 bytes = four half nodes of {3 words {lo, hi} uint16 grid coordinates, child word}, unused slots hold 0x7FFFFFFD behind an
    # inside-out box; the grid is bvh_grid(center, radius)
    balanced, k = 0, n_tris
    while k > 2:
        k, balanced = (k + 1) // 2, balanced + 1
    # the blob: header, triangles, materials | nodes | leaf records | normal classes
    assert n == bvh_off + (64 + 64) * n_tris + 320
    # ClassBlock: {n_classes, all_classified, pad, pad} + 8 x {axis * cos(h), sin(h)}: every triangle's plane lies in one of the
    # cones (half-angle h around the axis), so |dhat . axis| cos(h) - sin(h) bounds |cos(ray, normal)| from below
    cb = raw[leaf_off + 64 * n_tris:leaf_off + 64 * n_tris + 320]
    n_classes, all_classified = int(cb[:8].view(np.uint32)[0]), int(cb[:8].view(np.uint32)[1])
    cls = cb[16:16 + 16 * 8].view(np.float32).reshape(8, 4).astype(np.float64)
    assert all_classified == 1 and n_classes == 3                # the faces of an axis-aligned box: three plane directions
    Pd = np.array(V, dtype=np.float64)[:, :3]
    Ti = np.array(T, dtype=np.int64)
    nrm = np.cross(Pd[Ti[:, 1]] - Pd[Ti[:, 0]], Pd[Ti[:, 2]] - Pd[Ti[:, 0]])
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    for nv in nrm:
        ok = False
        for k in range(n_classes):
            ax, sin_h = cls[k, :3], cls[k, 3]
            cos_to_axis = abs(ax @ nv) / np.linalg.norm(ax)
            # the normal lies inside the cone the class claims: both the stored sin(h) and the stored cos(h) = |axis| say so
            if np.arcsin(min(1.0, sin_h)) >= np.arccos(min(1.0, cos_to_axis)) and np.linalg.norm(ax) <= cos_to_axis:
                ok = True
        assert ok
    # ... and the bound itself, on random directions: rho_k <= |cos(ray, normal)| for every triangle of class k
    dirs = np.random.default_rng(3).normal(size=(400, 3))
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    for k in range(n_classes):
        members = nrm[np.abs(nrm @ (cls[k, :3] / np.linalg.norm(cls[k, :3]))) > np.cos(0.15)]
        rho_k = np.abs(dirs @ cls[k, :3]) - cls[k, 3]
        assert (rho_k[:, None] <= np.abs(dirs @ members.T) + 1e-12).all()
    nodes_q = raw[bvh_off:bvh_off + 64 * n_nodes].view(np.uint16).reshape(n_nodes, 4, 8)
    nodes_i = raw[bvh_off:bvh_off + 64 * n_nodes].view(np.int32).reshape(n_nodes, 4, 4)
    half = np.float32(1.0001) * hdr_f[12] + np.float32(1e-6)
    glo = (hdr_f[9:12] - half).astype(np.float64)
    cell = float(np.float32(np.float32(2.0) * half) / np.float32(65534.0))
    leaves = raw[leaf_off:leaf_off + 64 * n_tris].view(np.float32).reshape(n_tris, 16)
    orig = leaves[:, 9].view(np.uint32)
    assert sorted(orig.tolist()) == list(range(n_tris))          # a permutation
    P = np.array(V, dtype=np.float32)[:, :3]
    Tm = np.array(T, dtype=np.int64)
    seen = np.zeros(n_tris, dtype=int)
    kappa_root = float(hdr_f[13])

    deepest = [0]
    need = [0]
    EMPTY = 0x7FFFFFFD

    def visit(child, lo, hi, depth=1, held=0):
        """checks containment: a quantised box holds its subtree's vertices EXACTLY (corners rounded outward, no tolerance)
        and lies inside its parent's"""
        if child < 0:
            code = ~child
            first, count = code >> 3, (code & 7) + 1
            for k in range(first, first + count):
                seen[k] += 1

                tri = Tm[orig[k]]
                pts = P[tri[:3]]
                assert (pts.astype(np.float64) >= lo).all() and (pts.astype(np.float64) <= hi).all()
                e0, e1 = pts[1] - pts[0], pts[2] - pts[0]
                assert np.array_equal(leaves[k, 0:3], pts[0]) and np.array_equal(leaves[k, 3:6], e0) and np.array_equal(leaves[k, 6:9], e1)
                assert kappa_root >= 16 * 2.0**-24 / 1e-6 * np.linalg.norm(e0) * np.linalg.norm(e1) * (1 - 1e-5)
            return
        deepest[0] = max(deepest[0], depth)
        slots = [c for c in range(4) if int(nodes_i[child, c, 3]) != EMPTY]
        assert slots == list(range(len(slots))) and len(slots) >= 2          # children first, empty slots last
        for c in range(len(slots), 4):
            q = nodes_q[child, c]
            assert (q[0:6:2] == 65535).all() and (q[1:6:2] == 0).all()      # inside out
        # a walk defers up to (children - 1) of a node's children while it descends into one of them
        held += len(slots) - 1
        need[0] = max(need[0], held)
        for c in slots:
            q = nodes_q[child, c].astype(np.float64)
            clo, chi = glo + q[0:6:2] * cell, glo + q[1:6:2] * cell
            # outward rounding costs at most one cell per face
            assert (clo >= lo).all() and (chi <= hi).all()
            visit(int(nodes_i[child, c, 3]), clo, chi, depth + 1, held)

    import sys
    sys.setrecursionlimit(10000)
    visit(0, np.full(3, -np.inf), np.full(3, np.inf))
    assert (seen == 1).all()
    # the traversal's stack is sized from the triangle count alone (bvh_depth_budget: balanced depth of 2-triangle leaves + 3
    # levels of slack for surface-area splits; a wide node stands for two binary levels and defers up to three children:
    # bvh_stack_total, of which bvh_stack_depth = the budget live in LDS and at most 16 in private memory); the builder must stay inside it
    budget = balanced + 3
    assert 0 < deepest[0] <= (budget + 1) // 2
    assert need[0] <= 3 * ((budget + 1) // 2) and 3 * ((budget + 1) // 2) + 1 - budget <= 16
    # the knob: brute force only
    assert lib.pt_hip_set_bvh_threshold(0xFFFFFFFF) == 0
    try:
        assert lib.pt_hip_scene_blob_bytes(len(T), len(M)) == 64 + 64 * len(T) + 32 * len(M) + (-(64 * len(T) + 32 * len(M)) % 64)
    finally:
        lib.pt_hip_set_bvh_threshold(pt.PT_BVH_DEFAULT_MIN_TRIANGLES)

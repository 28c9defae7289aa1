"""Camera-ray leaf lists (pt_hip_build_primary_lists; pathtracer_amd/csrc/pt_device.hpp bvh_collect_pixel): per pixel, one
conservative walk writes down the leaves ANY camera ray of that pixel could need, and the camera rays then test those
triangles without a node step.  The claim is "same hits, bit for bit" -- checked here three ways: the extend stage of bounce 0
with the lists against the same stage walking the tree (every path, {t, triangle}); the plugin's renders with and without
lists on every pixel, both engines; and (the other GPU suites) against the oracle, since the plugin builds the lists by
itself from the third sample of an accumulation on."""
import ctypes as C
import os

import numpy as np
import pytest

import scenes
from gpu_util import DeviceBuffer, make_frame, sync, upload_scene

pytestmark = pytest.mark.gpu

WALK = 0xFFFFFFFF


def _scene(name):
    if name.startswith("synthetic"):
        return scenes.synthetic_scene(int(name[len("synthetic"):]))
    if name == "soup":
        return scenes.triangle_soup(1500, 11)
    from test_bvh_adversarial import room_with_a_ball, tilted_room
    if name == "room+ball":
        return room_with_a_ball(300, 1200, 8)
    if name == "tilted room":
        return tilted_room(432, 6, 0.05)
    raise ValueError(name)


CAMERAS = {"default": dict(), "moved": dict(position=(0.3, -0.2, 0.1, 1.0), rotation=(0.05, 0.12, -0.03, 1.0), fov=70.0), "wide": dict(fov=120.0)}


@pytest.mark.parametrize("camera", ["default", "moved", "wide"])
@pytest.mark.parametrize("scene_name", ["synthetic2500", "synthetic30000", "room+ball", "tilted room", "soup"])
def test_extend_with_lists_equals_extend_walking(pt, oracle, scene_name, camera):
    lib = pt.load_library()
    V, T, M = _scene(scene_name)
    scene_dev = upload_scene(pt, lib, pt.make_scene(V, T, M))
    assert scene_dev.frame_flags & pt.PT_FRAME_SCENE_BVH
    w, h, first, n_pix, n_samples = 300, 170, 1234, 300 * 170 - 3000, 24
    cam = oracle.make_camera(**CAMERAS[camera])
    fr = make_frame(pt, oracle, w, h, cam=cam, pixel_first=first, pixel_count=n_pix, rng_mode=1, stride=35, flags=scene_dev.frame_flags)
    P = n_pix * n_samples
    G = lib.pt_hip_queue_blocks(P)
    rng = DeviceBuffer(lib, n_pix * 4)
    assert lib.pt_hip_rng_seed(C.byref(fr), rng.ptr, None) == 0
    qb = DeviceBuffer(lib, P * 16)
    Q = pt.Queue(a=None, b=qb.ptr, c=None, pid=None)
    counts = DeviceBuffer(lib, 4 * G * 4)
    smat = DeviceBuffer(lib, n_samples * 32 * 4)
    assert lib.pt_hip_generate(C.byref(fr), 3, n_samples, rng.ptr, smat.ptr, None, 0, Q, None, P, counts.ptr, None) == 0, lib.pt_hip_last_error()
    lists = DeviceBuffer(lib, lib.pt_hip_primary_lists_bytes(n_pix))
    assert lib.pt_hip_primary_lists_bytes(n_pix) == n_pix * 4 * (1 + pt.PT_PRIMARY_LIST_LEAVES)
    assert lib.pt_hip_build_primary_lists(C.byref(fr), scene_dev.ptr, len(T), len(M), lists.ptr, None) == 0, lib.pt_hip_last_error()
    hits = []
    for with_lists in (False, True):
        fr.primary_lists = lists.ptr if with_lists else 0
        hit = DeviceBuffer(lib, P * 8).upload(np.full(P * 2, 0xCDCDCDCD, dtype=np.uint32))
        assert lib.pt_hip_extend(C.byref(fr), 0, scene_dev.ptr, len(T), len(M), Q, counts.ptr, P, hit.ptr, None) == 0, lib.pt_hip_last_error()
        sync(lib)
        hits.append(hit.download(np.uint32).reshape(P, 2))
    diff = (hits[0] != hits[1]).any(axis=1)
    assert not diff.any(), f"{int(diff.sum())} of {P} camera rays differ; first: path {int(np.argmax(diff))} walk {hits[0][np.argmax(diff)]} lists {hits[1][np.argmax(diff)]}"
    assert (hits[0][:, 1] != 0xFFFFFFFF).mean() > 0.3                         # the rays do hit things
    L = lists.download(np.uint32).reshape(1 + pt.PT_PRIMARY_LIST_LEAVES, n_pix)
    cnt = L[0]
    listed = cnt != WALK
    assert (cnt[listed] <= pt.PT_PRIMARY_LIST_LEAVES).all()
    # the lists are worth having: nearly every pixel has one, and they are short
    # (a room of LARGE tilted quads owes its grazing rays inflations of the size of the quads themselves: a third of its pixels overflow)
    if scene_name.startswith("synthetic"):
        assert listed.mean() > 0.97 and cnt[listed].mean() < 4.5, (listed.mean(), cnt[listed].mean())
    # every triangle a camera ray hit is in its pixel's list
    leaf_off = None
    for path in np.random.default_rng(5).integers(0, P, 4000):
        li = int(path) % n_pix
        tri = int(hits[0][path, 1])
        if tri == 0xFFFFFFFF or not listed[li]:
            continue
        if leaf_off is None:
            blob = np.frombuffer(_packed(pt, lib, V, T, M), dtype=np.uint8)
            hdr = blob[:64].view(np.uint32)
            leaf_off = int(hdr[8])
            orig = blob[leaf_off:leaf_off + 64 * len(T)].view(np.uint32).reshape(len(T), 16)[:, 9]
        members = set()
        for code in L[1:1 + cnt[li], li]:
            f0, n = int(code) >> 3, (int(code) & 7) + 1
            members.update(int(x) for x in orig[f0:f0 + n])
        assert tri in members, (int(path), li, tri)


def _packed(pt, lib, V, T, M):
    v, t, m = pt.make_scene(V, T, M)
    n = lib.pt_hip_scene_blob_bytes(len(t), len(m))
    blob = (C.c_uint8 * n)()
    assert lib.pt_hip_scene_pack(C.cast(v, C.c_void_p), len(v), C.cast(t, C.c_void_p), len(t), C.cast(m, C.c_void_p), len(m), C.cast(blob, C.c_void_p), n) == 0
    return bytes(blob)


def test_lists_need_a_bvh_scene(pt, oracle):
    lib = pt.load_library()
    V, T, M = scenes.all_materials_scene()
    scene_dev = upload_scene(pt, lib, pt.make_scene(V, T, M))
    fr = make_frame(pt, oracle, 32, 16, flags=scene_dev.frame_flags)
    lists = DeviceBuffer(lib, lib.pt_hip_primary_lists_bytes(32 * 16))
    assert lib.pt_hip_build_primary_lists(C.byref(fr), scene_dev.ptr, len(T), len(M), lists.ptr, None) == -1
    assert b"PT_FRAME_SCENE_BVH" in lib.pt_hip_last_error()


@pytest.mark.parametrize("engine_name,engine,strided", [("fused", 1, False), ("fused-strided", 1, True), ("wavefront", 0, True), ("wavefront-sequential", 0, False)])
@pytest.mark.parametrize("scene_name", ["synthetic2500", "room+ball"])
def test_renders_with_and_without_lists_are_the_same_image(pt, oracle, scene_name, engine_name, engine, strided):
    """the plugin builds the lists from the third sample of an accumulation on (PT_PRIMARY_LISTS in the environment: never): same
    image, streams and ray counts either way, and the oracle's; a camera change drops the lists and the next ones are the new camera's"""
    V, T, M = _scene(scene_name)
    w, h, depth = 240, 135, 6
    cam2 = pt.CameraData(position=(0.2, 0.1, -0.05, 1.0), rotation=(0.0, 0.1, 0.0, 1.0), fieldOfView=80.0)
    out = {}
    for lists in (True, False):
        if not lists:
            os.environ["PT_PRIMARY_LISTS"] = "off"
        try:
            with pt.PathTracer(w, h) as tr:
                tr.set_scene(*pt.make_scene(V, T, M))
                tr.set_options(engine=engine, depth=depth, rng_mode=pt.PT_RNG_STRIDED if strided else pt.PT_RNG_SEQUENTIAL, stride=2 + 33 * depth, samples_per_pass=4)
                tr.render(1)                       # below the threshold: walks
                tr.render(5)                       # builds the lists, uses them
                tr.render(2)
                a = (tr.readback(), tr.readback_rng(), tr.stats().rays)
                tr.Update(cam2)                    # restart: the lists are dropped ...
                tr.render(4)                       # ... and built for the new camera
                b = (tr.readback(), tr.readback_rng(), tr.stats().rays)
                out[lists] = (a, b)
        finally:
            os.environ.pop("PT_PRIMARY_LISTS", None)
    for k in (0, 1):
        assert np.array_equal(out[True][k][0].view(np.uint32), out[False][k][0].view(np.uint32)), f"{engine_name}: image {k} differs with lists"
        assert np.array_equal(out[True][k][1], out[False][k][1]) and out[True][k][2] == out[False][k][2]
    mode = oracle.RNG_STRIDED if strided else oracle.RNG_SEQUENTIAL
    ref, ref_rng, st, _ = oracle.render(oracle.params(w, h, depth=depth, spp=8, rng_mode=mode, stride=2 + 33 * depth), scene=oracle.make_scene(V, T, M))
    assert np.array_equal(out[True][0][0].view(np.uint32), ref.view(np.uint32)) and out[True][0][2] == st.rays
    ref2, _, st2, _ = oracle.render(oracle.params(w, h, depth=depth, spp=4, rng_mode=mode, stride=2 + 33 * depth,
                                                  cam=oracle.make_camera(position=(0.2, 0.1, -0.05, 1.0), rotation=(0.0, 0.1, 0.0, 1.0), fov=80.0)),
                                    scene=oracle.make_scene(V, T, M))
    assert np.array_equal(out[True][1][0].view(np.uint32), ref2.view(np.uint32))

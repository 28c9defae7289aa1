"""ctypes binding of the CPU ORACLE (oracle/pt_oracle.c).  TEST INFRASTRUCTURE ONLY.

May be imported by tests/, by __graft_entry__.smoke() and by bench.py's cpu_baseline leg --
never by pathtracer_amd.  See oracle/pt_oracle.h for the parity status ("parity unpinned" by
reference fixtures; pinned by the hand-derived known answers of SURVEY.md 8c).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpt_oracle.so")

RNG_SEQUENTIAL, RNG_STRIDED = 0, 1
DIFFUSE, REFLECTIVE, EMISSIVE = 0, 1, 2


class F4(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float), ("w", C.c_float)]

    def tuple(self):
        return (self.x, self.y, self.z, self.w)


class Camera(C.Structure):
    _fields_ = [("position", C.c_float * 4), ("rotation", C.c_float * 4), ("fieldOfView", C.c_float)]


class Vertex(C.Structure):
    _fields_ = [("position", F4)]


class Triangle(C.Structure):
    _fields_ = [("vertexIdx0", C.c_uint32), ("vertexIdx1", C.c_uint32), ("vertexIdx2", C.c_uint32), ("materialIdx", C.c_uint32)]


class Material(C.Structure):
    _fields_ = [("type", C.c_uint32), ("_pad0", C.c_uint32 * 3), ("color", F4), ("reflectivity", C.c_float), ("_pad1", C.c_uint32 * 3)]


class Path(C.Structure):
    _fields_ = [("origin", F4), ("direction", F4), ("step", F4), ("frame", F4), ("normal", F4), ("t", C.c_float),
                ("materialIdx", C.c_uint32), ("rng", C.c_uint32), ("alive", C.c_uint32), ("hit_tri", C.c_int32),
                ("draws", C.c_uint32), ("rays", C.c_uint32), ("_pad", C.c_uint32)]


class RenderParams(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("depth", C.c_uint32), ("seed", C.c_uint32),
                ("rng_mode", C.c_uint32), ("stride", C.c_uint32), ("frame_first", C.c_uint32), ("spp", C.c_uint32),
                ("origin", F4), ("top_left", F4), ("bottom_left", F4), ("bottom_right", F4), ("threads", C.c_int32),
                ("shading", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("draws", C.c_uint64), ("rays_per_bounce", C.c_uint64 * 32)]


def build(force: bool = False) -> None:
    if force or not os.path.exists(LIB_PATH):
        subprocess.check_call(["make", "-C", _HERE, "libpt_oracle.so"], stdout=subprocess.DEVNULL)


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB_PATH)
        L.pto_xorshift32.restype = C.c_uint32
        L.pto_xorshift32.argtypes = [C.c_uint32]
        L.pto_rng_init_naive.restype = C.c_uint32
        L.pto_rng_init_naive.argtypes = [C.c_uint32, C.c_int32]
        L.pto_rng_jump.restype = C.c_uint32
        L.pto_rng_jump.argtypes = [C.c_uint32, C.c_uint64]
        L.pto_rng_uniform.restype = C.c_float
        L.pto_rng_uniform.argtypes = [C.POINTER(C.c_uint32)]
        L.pto_rng_state0.restype = C.c_uint32
        L.pto_rng_state0.argtypes = [C.c_uint32, C.c_uint32]
        L.pto_rng_seed_pixels.restype = None
        L.pto_rng_seed_pixels.argtypes = [C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
        L.pto_camera_default.argtypes = [C.POINTER(Camera)]
        L.pto_camera_corners.argtypes = [C.POINTER(Camera), C.c_float, C.POINTER(F4)]
        L.pto_default_scene.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.pto_ray_triangle.restype = C.c_int
        L.pto_ray_triangle.argtypes = [C.POINTER(F4), C.POINTER(F4), C.POINTER(Triangle), C.c_void_p, C.POINTER(C.c_float),
                                       C.POINTER(C.c_uint32), C.POINTER(F4)]
        L.pto_diffuse_reflection.restype = F4
        L.pto_diffuse_reflection.argtypes = [F4, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.pto_stage_generate.argtypes = [C.POINTER(RenderParams), C.c_uint32, C.c_uint32, C.POINTER(Path)]
        L.pto_stage_extend.argtypes = [C.POINTER(Path), C.c_void_p, C.c_uint32, C.c_void_p]
        L.pto_stage_shade.argtypes = [C.POINTER(Path), C.c_void_p]
        L.pto_accumulate.restype = F4
        L.pto_accumulate.argtypes = [F4, F4, C.c_uint32]
        L.pto_render.restype = C.c_int
        L.pto_render.argtypes = [C.POINTER(RenderParams), C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32,
                                 C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(Stats)]
        L.pto_render_sharded.restype = C.c_int
        L.pto_render_sharded.argtypes = [C.POINTER(RenderParams), C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32,
                                         C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.POINTER(Stats)]
        L.pto_render_sharded_span.restype = C.c_int
        L.pto_render_sharded_span.argtypes = [C.POINTER(RenderParams), C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32,
                                              C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.POINTER(Stats)]
        L.pto_combine_shards.argtypes = [C.c_void_p, C.c_uint32, C.c_size_t, C.c_void_p]
        _lib = L
    return _lib


def default_camera() -> Camera:
    cam = Camera()
    lib().pto_camera_default(C.byref(cam))
    return cam


def make_camera(position=(0, 0, 0, 1), rotation=(0, 0, 0, 1), fov=90.0) -> Camera:
    cam = Camera()
    cam.position[:] = [float(v) for v in position]
    cam.rotation[:] = [float(v) for v in rotation]
    cam.fieldOfView = float(fov)
    return cam


def camera_corners(cam: Camera, aspect: float):
    out = (F4 * 4)()
    lib().pto_camera_corners(C.byref(cam), C.c_float(aspect), out)
    return out


def default_scene():
    v, t, m = (Vertex * 8)(), (Triangle * 4)(), (Material * 2)()
    lib().pto_default_scene(v, t, m)
    return v, t, m


def make_scene(vertices, triangles, materials):
    v = (Vertex * len(vertices))()
    for k, p in enumerate(vertices):
        v[k].position = F4(*[float(c) for c in p])
    t = (Triangle * len(triangles))()
    for k, tri in enumerate(triangles):
        t[k] = Triangle(*[int(c) for c in tri])
    m = (Material * len(materials))()
    for k, mat in enumerate(materials):
        m[k].type = int(mat[0])
        m[k].color = F4(*[float(c) for c in mat[1]])
        m[k].reflectivity = float(mat[2]) if len(mat) > 2 else 0.0
    return v, t, m


SHADING_REFERENCE, SHADING_EXTENDED = 0, 1


def params(width, height, cam=None, depth=10, seed=1337, rng_mode=RNG_SEQUENTIAL, stride=35, frame_first=0, spp=1, threads=0,
           shading=SHADING_REFERENCE):
    cam = cam or default_camera()
    c = camera_corners(cam, np.float32(width) / np.float32(height))
    return RenderParams(width, height, depth, seed, rng_mode, stride, frame_first, spp, c[0], c[1], c[2], c[3], threads, shading)


def seed_states(seed, n, pixel_first=0, pixel_ids=None) -> np.ndarray:
    out = np.empty(n, dtype=np.uint32)
    ids = None
    if pixel_ids is not None:
        ids = np.ascontiguousarray(pixel_ids, dtype=np.uint32)
        assert ids.size == n
    lib().pto_rng_seed_pixels(seed, ids.ctypes.data if ids is not None else None, pixel_first, n, out.ctypes.data)
    return out


def render(p: RenderParams, scene=None, pixel_first=0, n=None, pixel_ids=None, rng_states=None, mean=None, want_last_frame=False):
    """Returns (mean[n,4] float32, rng_states[n] uint32 or None, stats, last_frame or None)."""
    v, t, m = scene or default_scene()
    ids = None
    if pixel_ids is not None:
        ids = np.ascontiguousarray(pixel_ids, dtype=np.uint32)
        n = ids.size
    elif n is None:
        n = p.width * p.height - pixel_first
    if mean is None:
        mean = np.ones((n, 4), dtype=np.float32)       # white start, Surface.cpp:99-115
    else:
        mean = np.ascontiguousarray(mean, dtype=np.float32).copy()
    if p.rng_mode == RNG_SEQUENTIAL:
        rng_states = seed_states(p.seed, n, pixel_first, ids) if rng_states is None else np.ascontiguousarray(rng_states, dtype=np.uint32).copy()
    last = np.empty((n, 4), dtype=np.float32) if want_last_frame else None
    st = Stats()
    rc = lib().pto_render(C.byref(p), C.cast(v, C.c_void_p), len(v), C.cast(t, C.c_void_p), len(t), C.cast(m, C.c_void_p), len(m),
                          ids.ctypes.data if ids is not None else None, pixel_first, n,
                          rng_states.ctypes.data if rng_states is not None else None, mean.ctypes.data,
                          last.ctypes.data if last is not None else None, C.byref(st))
    if rc != 0:
        raise RuntimeError(f"pto_render failed: {rc}")
    return mean, rng_states, st, last


def render_sharded(p: RenderParams, shards: int, scene=None, pixel_first=0, n=None, pixel_ids=None, span=None):
    """span: RNG sample indices between the starts of two shards (None: spp/shards, the shards back to back)"""
    v, t, m = scene or default_scene()
    ids = None
    if pixel_ids is not None:
        ids = np.ascontiguousarray(pixel_ids, dtype=np.uint32)
        n = ids.size
    elif n is None:
        n = p.width * p.height - pixel_first
    mean = np.empty((n, 4), dtype=np.float32)
    st = Stats()
    rc = lib().pto_render_sharded_span(C.byref(p), shards, p.spp // shards if span is None else span,
                                       C.cast(v, C.c_void_p), len(v), C.cast(t, C.c_void_p), len(t),
                                       C.cast(m, C.c_void_p), len(m), ids.ctypes.data if ids is not None else None, pixel_first, n,
                                       mean.ctypes.data, C.byref(st))
    if rc != 0:
        raise RuntimeError(f"pto_render_sharded failed: {rc}")
    return mean, st

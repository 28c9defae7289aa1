"""pathtracer_amd -- Python binding of the MI355X (gfx950) path-tracer plugin.

The product is the shared library ``libPT_HipWavefront.so`` (C++ host + HIP kernels, built by
``pathtracer_amd/csrc/Makefile``).  This module only binds its C-ABI (``include/pt_plugin.h``,
``include/pt_hip.h``) with ctypes, mirroring the reference's plugin interface
(``include/IPathTracer.h:5-14``: Update / BeginDrawing / FinalizeDrawing / Resize, created by
``Create`` and released by ``Destroy``).  There is no CPU fallback: if the library is missing or
no GPU is present the calls fail loudly.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIBRARY_NAME = "libPT_HipWavefront.so"
LIBRARY_PATH = os.path.join(_HERE, LIBRARY_NAME)

PT_OK = 0
PT_RNG_SEQUENTIAL, PT_RNG_STRIDED = 0, 1
PT_ENGINE_WAVEFRONT, PT_ENGINE_FUSED, PT_ENGINE_AUTO = 0, 1, 2
PT_SPLIT_NONE, PT_SPLIT_TILE, PT_SPLIT_SAMPLE = 0, 1, 2
PT_VIRTUAL_SHARDS = 8
PT_SHARD_SPAN = 1 << 20
PT_MAT_DIFFUSE, PT_MAT_REFLECTIVE, PT_MAT_EMISSIVE = 0, 1, 2
PT_SHADING_REFERENCE, PT_SHADING_EXTENDED = 0, 1
PT_FRAME_FAST_DIV = 1
PT_FRAME_SCENE_BVH = 2
PT_FRAME_OUT_OF_STEP = 4
PT_FRAME_DEFER_FOLD = 8
PT_MAX_DEPTH = 32
PT_BVH_DEFAULT_MIN_TRIANGLES = 96
PT_BVH_DUAL_MIN_TRIANGLES = 16
PT_STATS_PUBLIC_WORDS = 64
PT_STATS_WORDS = PT_STATS_PUBLIC_WORDS + 32 * 16
PT_NOALLOC_SCRATCH_BYTES = 64 << 20
PT_PRIMARY_LIST_LEAVES = 8


class PathTracerError(RuntimeError):
    pass


# ---- reference PODs (PT_CudaSimple/Structures.cuh:43-69, include/CameraData.h:14-18) ----
class Float4(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float), ("w", C.c_float)]


class Vertex(C.Structure):
    _fields_ = [("position", Float4)]


class Triangle(C.Structure):
    _fields_ = [("vertexIdx0", C.c_uint32), ("vertexIdx1", C.c_uint32), ("vertexIdx2", C.c_uint32), ("materialIdx", C.c_uint32)]


class Material(C.Structure):
    _fields_ = [("type", C.c_uint32), ("_pad0", C.c_uint32 * 3), ("color", Float4), ("reflectivity", C.c_float), ("_pad1", C.c_uint32 * 3)]


class CameraData(C.Structure):
    _fields_ = [("position", C.c_float * 4), ("rotation", C.c_float * 4), ("fieldOfView", C.c_float)]

    def __init__(self, position=(0.0, 0.0, 0.0, 1.0), rotation=(0.0, 0.0, 0.0, 1.0), fieldOfView=90.0):
        super().__init__()
        self.position[:] = position
        self.rotation[:] = rotation       # x, y, z, w
        self.fieldOfView = fieldOfView


class Frame(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("pixel_first", C.c_uint32), ("pixel_count", C.c_uint32),
                ("depth", C.c_uint32), ("seed", C.c_uint32), ("rng_mode", C.c_uint32), ("stride", C.c_uint32),
                ("shading", C.c_uint32), ("flags", C.c_uint32), ("primary_lists", C.c_uint64), ("need_masks", C.c_uint64),
                ("origin", Float4), ("top_left", Float4), ("bottom_left", Float4), ("bottom_right", Float4)]


class Queue(C.Structure):
    _fields_ = [("a", C.c_void_p), ("b", C.c_void_p), ("c", C.c_void_p), ("pid", C.c_void_p)]


class Options(C.Structure):
    _fields_ = [("engine", C.c_uint32), ("rng_mode", C.c_uint32), ("stride", C.c_uint32), ("depth", C.c_uint32),
                ("seed", C.c_uint32), ("samples_per_begin", C.c_uint32), ("samples_per_pass", C.c_uint32),
                ("device", C.c_int32), ("pixel_first", C.c_uint32), ("pixel_count", C.c_uint32),
                ("sample_offset", C.c_uint32), ("shading", C.c_uint32),
                ("split_mode", C.c_uint32), ("split_rank", C.c_uint32), ("split_ranks", C.c_uint32), ("virtual_shards", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("diffuse_bounces", C.c_uint64), ("paths", C.c_uint64), ("frames", C.c_uint64),
                ("rays_per_bounce", C.c_uint64 * PT_MAX_DEPTH), ("last_render_ms", C.c_double),
                ("total_render_ms", C.c_double), ("renders", C.c_uint64),
                ("engine_in_use", C.c_uint32), ("fused_out_of_step", C.c_uint32), ("auto_rays_per_ms", C.c_double * 2),
                ("fused_out_of_step_rays_per_ms", C.c_double),
                ("small_scene_walks", C.c_uint32), ("small_scene_dual", C.c_uint32), ("walk_rays_per_ms", C.c_double * 2)]


assert C.sizeof(Vertex) == 16 and C.sizeof(Triangle) == 16 and C.sizeof(Material) == 48 and C.sizeof(CameraData) == 36

# Every symbol include/pt_plugin.h and include/pt_hip.h declare: name -> (restype, argtypes)
_VP, _U32, _I32, _SZ = C.c_void_p, C.c_uint32, C.c_int32, C.c_size_t
_FRAMEP, _U32P, _U64P = C.POINTER(Frame), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)
SYMBOLS = {
    # reference exports (PathTracer.cuh:69-76)
    "Create": (_VP, [C.c_uint, C.c_int, C.c_int, C.POINTER(CameraData)]),
    "Destroy": (None, [_VP]),
    # pt_plugin.h
    "pt_last_error": (C.c_char_p, []),
    "pt_set_default_device": (C.c_int, [C.c_int]),
    "pt_update": (C.c_int, [_VP, C.POINTER(CameraData)]),
    "pt_begin_drawing": (C.c_int, [_VP]),
    "pt_finalize_drawing": (C.c_int, [_VP]),
    "pt_resize": (C.c_int, [_VP, C.c_int, C.c_int]),
    "pt_get_options": (C.c_int, [_VP, C.POINTER(Options)]),
    "pt_set_options": (C.c_int, [_VP, C.POINTER(Options)]),
    "pt_set_scene": (C.c_int, [_VP, _VP, _U32, _VP, _U32, _VP, _U32]),
    "pt_scene_file_read": (C.c_int, [C.c_char_p, C.POINTER(_VP), _U32P, C.POINTER(_VP), _U32P, C.POINTER(_VP), _U32P]),
    "pt_scene_file_free": (None, [_VP]),
    "pt_scene_file_write": (C.c_int, [C.c_char_p, _VP, _U32, _VP, _U32, _VP, _U32]),
    "pt_load_scene": (C.c_int, [_VP, C.c_char_p]),
    "pt_save_state": (C.c_int, [_VP, C.c_char_p]),
    "pt_load_state": (C.c_int, [_VP, C.c_char_p]),
    "pt_get_camera": (C.c_int, [_VP, C.POINTER(CameraData)]),
    "pt_render": (C.c_int, [_VP, _U32]),
    "pt_readback": (C.c_int, [_VP, _VP, _SZ]),
    "pt_readback_rng": (C.c_int, [_VP, _VP, _SZ]),
    "pt_copy_mean": (C.c_int, [_VP, _VP, _SZ]),
    "pt_copy_mean_async": (C.c_int, [_VP, _VP, _SZ]),
    "pt_stream": (_VP, [_VP]),
    "pt_device_mean": (_VP, [_VP]),
    "pt_get_stats": (C.c_int, [_VP, C.POINTER(Stats)]),
    "pt_get_size": (C.c_int, [_VP, C.POINTER(C.c_int), C.POINTER(C.c_int), _U32P, _U32P]),
    "pt_combine_instances": (C.c_int, [C.POINTER(_VP), _U32]),
    "pt_dist_unique_id": (C.c_int, [_VP]),
    "pt_dist_init": (C.c_int, [_VP, _VP]),
    "pt_dist_combine": (C.c_int, [_VP]),
    "pt_get_shard_layout": (C.c_int, [_VP, _U32P, _U32P, _U32P]),
    "pt_create_on_device": (_VP, [C.c_int, C.c_int, C.c_int, C.POINTER(CameraData)]),
    # pt_hip.h
    "pt_hip_last_error": (C.c_char_p, []),
    "pt_hip_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "pt_hip_set_device": (C.c_int, [C.c_int]),
    "pt_hip_get_device": (C.c_int, [C.POINTER(C.c_int)]),
    "pt_hip_malloc": (C.c_int, [C.POINTER(_VP), _SZ]),
    "pt_hip_free": (C.c_int, [_VP]),
    "pt_hip_memcpy_h2d": (C.c_int, [_VP, _VP, _SZ, _VP]),
    "pt_hip_memcpy_d2h": (C.c_int, [_VP, _VP, _SZ, _VP]),
    "pt_hip_memcpy_d2d": (C.c_int, [_VP, _VP, _SZ, _VP]),
    "pt_hip_memcpy_peer": (C.c_int, [_VP, C.c_int, _VP, C.c_int, _SZ, _VP]),
    "pt_hip_memset": (C.c_int, [_VP, C.c_int, _SZ, _VP]),
    "pt_hip_fill_f32": (C.c_int, [_VP, C.c_float, _SZ, _VP]),
    "pt_hip_stream_create": (C.c_int, [C.POINTER(_VP)]),
    "pt_hip_stream_destroy": (C.c_int, [_VP]),
    "pt_hip_stream_synchronize": (C.c_int, [_VP]),
    "pt_hip_event_create": (C.c_int, [C.POINTER(_VP)]),
    "pt_hip_event_destroy": (C.c_int, [_VP]),
    "pt_hip_event_record": (C.c_int, [_VP, _VP]),
    "pt_hip_stream_wait_event": (C.c_int, [_VP, _VP]),
    "pt_hip_fold_stats": (C.c_int, [_VP, _VP]),
    "pt_hip_render_fused_shards": (C.c_int, [_FRAMEP, _VP, _U32, _U32, C.POINTER(_U32), _U32, _U32, _U32, _VP, _VP, _SZ, _VP, _VP]),
    "pt_hip_event_elapsed_ms": (C.c_int, [_VP, _VP, C.POINTER(C.c_float)]),
    "pt_hip_set_bvh_threshold": (C.c_int, [_U32]),
    "pt_hip_scene_blob_bytes": (_SZ, [_U32, _U32]),
    "pt_hip_scene_blob_bytes_ex": (_SZ, [_U32, _U32, C.c_int]),
    "pt_hip_scene_has_bvh": (C.c_int, [_VP, _SZ]),
    "pt_hip_scene_pack": (C.c_int, [_VP, _U32, _VP, _U32, _VP, _U32, _VP, _SZ]),
    "pt_hip_scene_pack_ex": (C.c_int, [_VP, _U32, _VP, _U32, _VP, _U32, _VP, _SZ, C.c_int]),
    "pt_hip_verify_fast_div": (C.c_int, [_U32, C.POINTER(C.c_int), _VP]),
    "pt_hip_need_masks_bytes": (_SZ, [_U32]),
    "pt_hip_build_need_masks": (C.c_int, [C.POINTER(Frame), _VP, _U32, _U32, _VP, _VP]),
    "pt_hip_primary_lists_bytes": (_SZ, [_U32]),
    "pt_hip_build_primary_lists": (C.c_int, [C.POINTER(Frame), _VP, _U32, _U32, _VP, _VP]),
    "pt_hip_rng_seed": (C.c_int, [_FRAMEP, _VP, _VP]),
    "pt_hip_queue_blocks": (_U32, [_U32]),
    "pt_hip_generate": (C.c_int, [_FRAMEP, _U32, _U32, _VP, _VP, _VP, _U32, Queue, _VP, _U32, _VP, _VP]),
    "pt_hip_extend": (C.c_int, [_FRAMEP, _U32, _VP, _U32, _U32, Queue, _VP, _U32, _VP, _VP]),
    "pt_hip_shade": (C.c_int, [_FRAMEP, _VP, _U32, _U32, _U32, Queue, _VP, _VP, _U32, Queue, _VP, _VP, _VP, _VP, _VP, _VP]),
    "pt_hip_accumulate": (C.c_int, [_U32, _U32, _U32, _VP, _VP, _VP, _VP]),
    "pt_hip_wavefront_workspace_bytes": (_SZ, [_U32, _U32, _U32]),
    "pt_hip_render_wavefront": (C.c_int, [_FRAMEP, _VP, _U32, _U32, _U32, _U32, _U32, _U32, _VP, _VP, _VP, _SZ, _VP, _VP]),
    "pt_hip_render_wavefront_noalloc": (C.c_int, [_FRAMEP, _VP, _U32, _U32, _U32, _U32, _U32, _U32, _VP, _VP, _VP, _VP]),
    "pt_hip_render_fused": (C.c_int, [_FRAMEP, _VP, _U32, _U32, _U32, _U32, _U32, _VP, _VP, _VP, _VP]),
    "pt_hip_combine_shards": (C.c_int, [_VP, _U32, _SZ, _VP, _VP]),
    "pt_hip_comm_unique_id": (C.c_int, [_VP]),
    "pt_hip_comm_create": (C.c_int, [C.POINTER(_VP), _VP, C.c_int, C.c_int]),
    "pt_hip_comm_destroy": (C.c_int, [_VP]),
    "pt_hip_comm_ranks": (C.c_int, [_VP, C.POINTER(C.c_int)]),
    "pt_hip_comm_exchange_plan": (C.c_int, [C.c_int, C.c_int, _U32, _SZ, _VP]),
    "pt_hip_comm_exchange_combine": (C.c_int, [_VP, C.c_int, C.c_int, _VP, _U32, _SZ, _VP, _VP, _VP]),
}

_lib: Optional[C.CDLL] = None


def load_library(path: Optional[str] = None) -> C.CDLL:
    """dlopen the plugin and type every export.  Raises if the library or a symbol is missing."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or os.environ.get("PT_HIP_LIBRARY", LIBRARY_PATH)
    if not os.path.exists(path):
        raise PathTracerError(
            f"{path} not found: build it with `make -C pathtracer_amd/csrc` (needs hipcc); there is no CPU fallback")
    lib = C.CDLL(path)
    for name, (restype, argtypes) in SYMBOLS.items():
        fn = getattr(lib, name)          # AttributeError names a missing export
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def _check(rc: int, lib: C.CDLL, what: str, hip_layer: bool = False) -> None:
    if rc != PT_OK:
        msg = (lib.pt_hip_last_error() if hip_layer else lib.pt_last_error()) or b""
        raise PathTracerError(f"{what} failed ({rc}): {msg.decode(errors='replace')}")


def default_scene():
    """The reference's literal scene (PathTracer.cu:48-69) as ctypes arrays."""
    V = [(-0.5, 0.5, 1.0, 1.0), (0.5, 0.5, 1.0, 1.0), (0.5, -0.5, 1.0, 1.0), (-0.5, -0.5, 1.0, 1.0),
         (-0.2, 0.5, 0.5, 1.0), (0.2, 0.5, 0.5, 1.0), (0.2, 0.5, 0.9, 1.0), (-0.2, 0.5, 0.9, 1.0)]
    T = [(0, 1, 2, 0), (2, 3, 0, 0), (4, 5, 6, 1), (6, 7, 4, 1)]
    return make_scene(V, T, [(PT_MAT_DIFFUSE, (1.0, 0.2, 0.2, 1.0)), (PT_MAT_EMISSIVE, (10.0, 10.0, 10.0, 10.0))])


def make_scene(vertices, triangles, materials):
    """vertices: (x,y,z,w); triangles: (i0,i1,i2,material); materials: (type,(r,g,b,a)[,reflectivity])."""
    v = (Vertex * len(vertices))()
    for k, p in enumerate(vertices):
        v[k].position = Float4(*[float(c) for c in p])
    t = (Triangle * len(triangles))()
    for k, tri in enumerate(triangles):
        t[k] = Triangle(*[int(c) for c in tri])
    m = (Material * len(materials))()
    for k, mat in enumerate(materials):
        m[k].type = int(mat[0])
        m[k].color = Float4(*[float(c) for c in mat[1]])
        m[k].reflectivity = float(mat[2]) if len(mat) > 2 else 0.0
    return v, t, m


def read_scene_file(path: str):
    """Parse a scene file on the host (no GPU): returns (vertices, triangles, materials) ctypes arrays."""
    lib = load_library()
    pv, pt_, pm = C.c_void_p(), C.c_void_p(), C.c_void_p()
    nv, nt, nm = C.c_uint32(), C.c_uint32(), C.c_uint32()
    _check(lib.pt_scene_file_read(os.fsencode(path), C.byref(pv), C.byref(nv), C.byref(pt_), C.byref(nt), C.byref(pm), C.byref(nm)),
           lib, "pt_scene_file_read")
    try:
        v = (Vertex * nv.value).from_buffer_copy(C.string_at(pv.value, nv.value * 16)) if nv.value else (Vertex * 0)()
        t = (Triangle * nt.value).from_buffer_copy(C.string_at(pt_.value, nt.value * 16)) if nt.value else (Triangle * 0)()
        m = (Material * nm.value).from_buffer_copy(C.string_at(pm.value, nm.value * 48)) if nm.value else (Material * 0)()
    finally:
        for p in (pv, pt_, pm):
            lib.pt_scene_file_free(p)
    return v, t, m


def write_scene_file(path: str, vertices, triangles, materials) -> None:
    lib = load_library()
    _check(lib.pt_scene_file_write(os.fsencode(path), C.cast(vertices, C.c_void_p), len(vertices), C.cast(triangles, C.c_void_p),
                                   len(triangles), C.cast(materials, C.c_void_p), len(materials)), lib, "pt_scene_file_write")


class PathTracer:
    """The plugin behind the reference's IPathTracer surface, plus the headless extras.

    Method names and call order are the reference's (include/IPathTracer.h:7-10,
    PathTracer/Program.cpp:57-58,70): ``BeginDrawing(); Update(cam); FinalizeDrawing()``
    adds one sample per pixel to the running mean.
    """

    def __init__(self, width: int, height: int, camera: Optional[CameraData] = None, device: Optional[int] = None,
                 library: Optional[str] = None):
        self._lib = load_library(library)
        self._camera = camera or CameraData()
        if device is not None:
            self._h = self._lib.pt_create_on_device(int(device), int(width), int(height), C.byref(self._camera))
        else:
            self._h = self._lib.Create(0, int(width), int(height), C.byref(self._camera))     # the reference's export
        if not self._h:
            raise PathTracerError("Create failed: " + (self._lib.pt_last_error() or b"").decode(errors="replace"))

    # -- IPathTracer --
    def Update(self, camera: CameraData) -> None:
        _check(self._lib.pt_update(self._h, C.byref(camera)), self._lib, "Update")

    def BeginDrawing(self) -> None:
        _check(self._lib.pt_begin_drawing(self._h), self._lib, "BeginDrawing")

    def FinalizeDrawing(self) -> None:
        _check(self._lib.pt_finalize_drawing(self._h), self._lib, "FinalizeDrawing")

    def Resize(self, width: int, height: int) -> None:
        _check(self._lib.pt_resize(self._h, int(width), int(height)), self._lib, "Resize")

    def Destroy(self) -> None:
        if getattr(self, "_h", None):
            self._lib.Destroy(self._h)
            self._h = None

    close = Destroy

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.Destroy()

    def __del__(self):
        try:
            self.Destroy()
        except Exception:
            pass

    # -- headless extras (include/pt_plugin.h) --
    def get_options(self) -> Options:
        o = Options()
        _check(self._lib.pt_get_options(self._h, C.byref(o)), self._lib, "pt_get_options")
        return o

    def set_options(self, **kw) -> Options:
        o = self.get_options()
        for k, v in kw.items():
            if not hasattr(o, k):
                raise AttributeError(f"unknown option {k}")
            setattr(o, k, v)
        _check(self._lib.pt_set_options(self._h, C.byref(o)), self._lib, "pt_set_options")
        return o

    def set_scene(self, vertices, triangles, materials) -> None:
        _check(self._lib.pt_set_scene(self._h, C.cast(vertices, C.c_void_p), len(vertices), C.cast(triangles, C.c_void_p),
                                      len(triangles), C.cast(materials, C.c_void_p), len(materials)), self._lib, "pt_set_scene")

    def load_scene(self, path: str) -> None:
        _check(self._lib.pt_load_scene(self._h, os.fsencode(path)), self._lib, "pt_load_scene")

    def save_state(self, path: str) -> None:
        """checkpoint: options, camera, sample count, RNG streams, running mean(s), counters (include/pt_plugin.h)"""
        _check(self._lib.pt_save_state(self._h, os.fsencode(path)), self._lib, "pt_save_state")

    def camera(self) -> CameraData:
        cam = CameraData()
        _check(self._lib.pt_get_camera(self._h, C.byref(cam)), self._lib, "pt_get_camera")
        return cam

    def load_state(self, path: str) -> None:
        """resume a checkpoint on an instance of the same size and scene: the accumulation continues bit for bit"""
        _check(self._lib.pt_load_state(self._h, os.fsencode(path)), self._lib, "pt_load_state")

    def render(self, spp: int) -> None:
        _check(self._lib.pt_render(self._h, int(spp)), self._lib, "pt_render")

    def size(self):
        w, h, first, count = C.c_int(), C.c_int(), C.c_uint32(), C.c_uint32()
        _check(self._lib.pt_get_size(self._h, C.byref(w), C.byref(h), C.byref(first), C.byref(count)), self._lib, "pt_get_size")
        return w.value, h.value, first.value, count.value

    def readback(self) -> np.ndarray:
        """Running mean, float32 [pixel_count, 4]; pixel i = y*width + x, row 0 is the bottom row."""
        _, _, _, count = self.size()
        out = np.empty((count, 4), dtype=np.float32)
        _check(self._lib.pt_readback(self._h, out.ctypes.data_as(C.c_void_p), out.size), self._lib, "pt_readback")
        return out

    def readback_rng(self) -> np.ndarray:
        _, _, _, count = self.size()
        out = np.empty(count, dtype=np.uint32)
        _check(self._lib.pt_readback_rng(self._h, out.ctypes.data_as(C.c_void_p), out.size), self._lib, "pt_readback_rng")
        return out

    def copy_mean(self, device_ptr: int, n_floats: int) -> None:
        """Snapshot of the running mean into caller-owned device memory (e.g. a torch tensor's data_ptr())."""
        _check(self._lib.pt_copy_mean(self._h, C.c_void_p(int(device_ptr)), int(n_floats)), self._lib, "pt_copy_mean")

    def copy_mean_async(self, device_ptr: int, n_floats: int) -> None:
        """The same copy enqueued on the plugin's stream behind the frame in flight; nothing is awaited."""
        _check(self._lib.pt_copy_mean_async(self._h, C.c_void_p(int(device_ptr)), int(n_floats)), self._lib, "pt_copy_mean_async")

    def stream(self) -> int:
        """The plugin's hipStream_t as an integer (torch.cuda.ExternalStream(tracer.stream()))."""
        return int(self._lib.pt_stream(self._h) or 0)

    def device_mean(self) -> int:
        return int(self._lib.pt_device_mean(self._h) or 0)

    # -- sample split (include/pt_plugin.h) --
    def shard_layout(self):
        """(local shards, pitch in pixels between their means, virtual shards)"""
        a, b, c = C.c_uint32(), C.c_uint32(), C.c_uint32()
        _check(self._lib.pt_get_shard_layout(self._h, C.byref(a), C.byref(b), C.byref(c)), self._lib, "pt_get_shard_layout")
        return a.value, b.value, c.value

    def dist_init(self, id128: bytes) -> None:
        buf = C.create_string_buffer(bytes(id128), 128)
        _check(self._lib.pt_dist_init(self._h, C.cast(buf, C.c_void_p)), self._lib, "pt_dist_init")

    def dist_combine(self) -> None:
        _check(self._lib.pt_dist_combine(self._h), self._lib, "pt_dist_combine")

    def stats(self) -> Stats:
        s = Stats()
        _check(self._lib.pt_get_stats(self._h, C.byref(s)), self._lib, "pt_get_stats")
        return s


def dist_unique_id() -> bytes:
    """128 bytes of an ncclUniqueId (rank 0 makes it, the others get it by the host's own means)."""
    lib = load_library()
    buf = C.create_string_buffer(128)
    _check(lib.pt_dist_unique_id(C.cast(buf, C.c_void_p)), lib, "pt_dist_unique_id")
    return buf.raw


def combine_instances(tracers) -> None:
    """pt_combine_instances: all instances of one sample split, in this process; the image lands in tracers[0]."""
    lib = load_library()
    arr = (C.c_void_p * len(tracers))(*[t._h for t in tracers])
    _check(lib.pt_combine_instances(arr, len(tracers)), lib, "pt_combine_instances")


__all__ = ["PathTracer", "dist_unique_id", "combine_instances", "PathTracerError", "CameraData", "Options", "Stats", "Frame", "Queue", "Float4", "Vertex",
           "Triangle", "Material", "load_library", "default_scene", "make_scene", "SYMBOLS", "LIBRARY_PATH"]

"""Device-buffer helpers for the GPU tests: everything goes through the pt_hip_* C-ABI."""
import ctypes as C

import numpy as np


class DeviceBuffer:
    def __init__(self, lib, nbytes):
        self.lib, self.nbytes = lib, int(nbytes)
        p = C.c_void_p()
        rc = lib.pt_hip_malloc(C.byref(p), max(self.nbytes, 4))
        assert rc == 0, lib.pt_hip_last_error()
        self.ptr = p.value

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= self.nbytes
        rc = self.lib.pt_hip_memcpy_h2d(self.ptr, arr.ctypes.data, arr.nbytes, None)
        assert rc == 0, self.lib.pt_hip_last_error()
        return self

    def download(self, dtype, count=None):
        dtype = np.dtype(dtype)
        count = self.nbytes // dtype.itemsize if count is None else count
        out = np.empty(count, dtype=dtype)
        rc = self.lib.pt_hip_memcpy_d2h(out.ctypes.data, self.ptr, out.nbytes, None)
        assert rc == 0, self.lib.pt_hip_last_error()
        return out

    def free(self):
        if self.ptr:
            self.lib.pt_hip_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def upload_scene(pt, lib, scene):
    v, t, m = scene
    n = lib.pt_hip_scene_blob_bytes(len(t), len(m))
    blob = (C.c_uint8 * n)()
    rc = lib.pt_hip_scene_pack(C.cast(v, C.c_void_p), len(v), C.cast(t, C.c_void_p), len(t), C.cast(m, C.c_void_p), len(m),
                               C.cast(blob, C.c_void_p), n)
    assert rc == 0, lib.pt_hip_last_error()
    buf = DeviceBuffer(lib, n).upload(np.frombuffer(blob, dtype=np.uint8))
    has_bvh = lib.pt_hip_scene_has_bvh(C.cast(blob, C.c_void_p), n)
    assert has_bvh in (0, 1), lib.pt_hip_last_error()
    buf.frame_flags = pt.PT_FRAME_SCENE_BVH if has_bvh else 0      # OR into pt_frame.flags of every call that takes this blob
    return buf


def make_frame(pt, oracle, w, h, cam=None, pixel_first=0, pixel_count=None, depth=10, seed=1337, rng_mode=0, stride=35, flags=0):
    cam = cam or oracle.default_camera()
    c = oracle.camera_corners(cam, np.float32(w) / np.float32(h))
    f4 = lambda v: pt.Float4(v.x, v.y, v.z, v.w)
    return pt.Frame(width=w, height=h, pixel_first=pixel_first, pixel_count=w * h - pixel_first if pixel_count is None else pixel_count,
                    depth=depth, seed=seed, rng_mode=rng_mode, stride=stride, flags=flags,
                    origin=f4(c[0]), top_left=f4(c[1]), bottom_left=f4(c[2]), bottom_right=f4(c[3]))


def queue_slots(counts, G):
    """Slot indices of every queued entry, block by block (include/pt_hip.h queue geometry)."""
    out = []
    for b in range(G):
        j = np.arange(int(counts[b]), dtype=np.int64)
        out.append(((j // 256) * G + b) * 256 + (j % 256))
    return np.concatenate(out) if out else np.zeros(0, dtype=np.int64)


def sync(lib):
    assert lib.pt_hip_stream_synchronize(None) == 0, lib.pt_hip_last_error()

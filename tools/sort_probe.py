#!/usr/bin/env python3
"""What would ray reordering buy the BVH extend of bounced rays?  (measurement only)

Stage API on the 10 092-triangle scene at 1080p: generate -> extend(0) -> shade(0) leaves the bounce-1 rays in the queue,
block-private slots in pixel order.  k_extend<false, true> is then timed on that queue as it is and on copies whose
entries are permuted WITHIN each block's own slots (the block-private layout is kept): by direction octant, by octant in
windows of W entries, by octant + Morton code of the origin."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pathtracer_amd as pt            # noqa: E402
import scenes                           # noqa: E402
from oracle import pt_oracle_py as po   # noqa: E402
from gpu_util import DeviceBuffer, upload_scene, make_frame, sync   # noqa: E402

lib = pt.load_library()
n_tris = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
n_samples = int(sys.argv[2]) if len(sys.argv) > 2 else 4
w, h, depth = 1920, 1080, 10
V, T, M = scenes.synthetic_scene(n_tris)
scene_dev = upload_scene(pt, lib, pt.make_scene(V, T, M))
fr = make_frame(pt, po, w, h, depth=depth, rng_mode=1, stride=2 + 33 * depth, flags=scene_dev.frame_flags)
n_pix = w * h
P = n_pix * n_samples
G = lib.pt_hip_queue_blocks(P)
rng = DeviceBuffer(lib, n_pix * 4)
assert lib.pt_hip_rng_seed(C.byref(fr), rng.ptr, None) == 0
q = [dict(a=DeviceBuffer(lib, P * 16), b=DeviceBuffer(lib, P * 16), c=DeviceBuffer(lib, P * 16), pid=DeviceBuffer(lib, P * 4)) for _ in range(2)]
Q = [pt.Queue(a=x["a"].ptr, b=x["b"].ptr, c=x["c"].ptr, pid=x["pid"].ptr) for x in q]
counts = DeviceBuffer(lib, (depth + 2) * G * 4)
hit = DeviceBuffer(lib, P * 8)
radiance = DeviceBuffer(lib, P * 16)
lit = DeviceBuffer(lib, P)
smat = DeviceBuffer(lib, n_samples * 32 * 4)
assert lib.pt_hip_generate(C.byref(fr), 0, n_samples, rng.ptr, smat.ptr, scene_dev.ptr, len(T), Q[0], None, P, counts.ptr, None) == 0, lib.pt_hip_last_error()
c0, c1 = C.c_void_p(counts.ptr), C.c_void_p(counts.ptr + G * 4)


def timed_extend(bounce, queue, cnt, reps=5):
    best = 1e9
    for _ in range(reps):
        sync(lib)
        t0 = time.perf_counter()
        assert lib.pt_hip_extend(C.byref(fr), bounce, scene_dev.ptr, len(T), len(M), queue, cnt, P, hit.ptr, None) == 0, lib.pt_hip_last_error()
        sync(lib)
        best = min(best, time.perf_counter() - t0)
    return best


t_cam = timed_extend(0, Q[0], c0)
print(f"{len(T)} triangles, {P} camera rays: extend {t_cam * 1e3:.2f} ms = {P / t_cam / 1e9:.2f} Grays/s", flush=True)
assert lib.pt_hip_shade(C.byref(fr), scene_dev.ptr, len(T), len(M), 0, Q[0], hit.ptr, c0, P, Q[1], None, c1, radiance.ptr, lit.ptr, None, None) == 0, lib.pt_hip_last_error()
sync(lib)
cnt = counts.download(np.uint32)[G:2 * G].astype(np.int64)
n1 = int(cnt.sum())
A = q[1]["a"].download(np.float32).reshape(P, 4)
B = q[1]["b"].download(np.float32).reshape(P, 4)
ref_hit = None

# slot of every entry, block by block, and the block each belongs to
j = np.concatenate([np.arange(c, dtype=np.int64) for c in cnt])
blk = np.repeat(np.arange(G, dtype=np.int64), cnt)
slots = ((j // 256) * G + blk) * 256 + (j % 256)
o = A[slots, :3]
d = np.stack([A[slots, 3], B[slots, 0], B[slots, 1]], axis=1)
octant = (d[:, 0] < 0).astype(np.int64) | ((d[:, 1] < 0).astype(np.int64) << 1) | ((d[:, 2] < 0).astype(np.int64) << 2)


def morton(o):
    lo, hi = o.min(axis=0), o.max(axis=0)
    g = np.clip(((o - lo) / np.maximum(hi - lo, 1e-9) * 1023).astype(np.int64), 0, 1023)
    m = np.zeros(len(o), dtype=np.int64)
    for bit in range(10):
        for ax in range(3):
            m |= ((g[:, ax] >> bit) & 1) << (3 * bit + ax)
    return m


def major_axis6(d):
    ax = np.abs(d).argmax(axis=1)
    return ax * 2 + (d[np.arange(len(d)), ax] < 0)


def run(name, key):
    """stable sort of each block's entries by key (key None: unchanged)"""
    if key is None:
        order = np.arange(len(slots))
    else:
        order = np.lexsort((j, key, blk))           # by block, then key, then original position
    A2, B2 = A.copy(), B.copy()
    A2[slots] = A[slots[order]]
    B2[slots] = B[slots[order]]
    q[1]["a"].upload(A2)
    q[1]["b"].upload(B2)
    t = timed_extend(1, Q[1], c1)
    H = hit.download(np.uint32).reshape(P, 2)[slots]
    global ref_hit
    if ref_hit is None:
        ref_hit = H.copy()
    else:
        assert np.array_equal(H, ref_hit[order]), "a reordered queue changed a result"
    print(f"  {name:58s} {t * 1e3:7.2f} ms  {n1 / t / 1e9:6.2f} Grays/s", flush=True)


print(f"{n1} bounced rays in {G} blocks ({n1 / G:.0f} per block)")
run("queue as the shade stage leaves it (pixel order)", None)
run("by octant, whole block", octant)
for W in (256, 1024, 4096):
    run(f"by octant within windows of {W}", (j // W) * 8 + octant)
run("by major axis (6 classes), whole block", major_axis6(d))
run("by octant, then Morton code of the origin", octant * (1 << 30) + morton(o))
run("by Morton code of the origin only", morton(o))
run("random order within the block (lower bound)", np.random.default_rng(1).integers(0, 1 << 40, len(slots)))

"""BVH traversal == brute force, on the CPU, with the device code itself.

tests/host/bvh_host_check.cpp compiles pathtracer_amd/csrc/pt_device.hpp with g++ through a stand-in
hip_runtime.h (a wave of one lane) under AddressSanitizer/UBSan and intersects every ray twice: the
reference's loop over all triangles (kernels.cu:62-68) and the BVH walk.  The scene blobs are packed by
the library (pt_hip_scene_pack, host code), the rays are camera rays, rays leaving the walls in random
directions -- a third of them grazing the wall they start on, where the fp32 Moeller-Trumbore test is
least accurate and the culling margins of pt_bvh.hpp matter -- with fresh and stale hit distances.
The wave-level side -- the persistent loop of k_extend<*, BVH> with its ballots, lane refill and the quorum rule of
the node walk -- runs as 64 host threads, one per lane (tests/host/extend_wave_check.cpp)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHECK = os.path.join(ROOT, "tests", "host", "bvh_host_check")
CHECK_SPILL = os.path.join(ROOT, "tests", "host", "bvh_host_check_spill")     # three stack levels in "LDS", the rest in BvhStack::spill
WAVE_CHECK = os.path.join(ROOT, "tests", "host", "extend_wave_check")
FLT_MAX = np.finfo(np.float32).max


def pack(pt, V, T, M):
    lib = pt.load_library()
    v, t, m = pt.make_scene(V, T, M)
    n = lib.pt_hip_scene_blob_bytes(len(T), len(M))
    blob = (C.c_uint8 * n)()
    assert lib.pt_hip_scene_pack(C.cast(v, C.c_void_p), len(V), C.cast(t, C.c_void_p), len(T), C.cast(m, C.c_void_p), len(M),
                                 C.cast(blob, C.c_void_p), n) == 0, lib.pt_hip_last_error()
    return bytes(blob)


def pack_with_bvh(pt, V, T, M):
    """pt_hip_scene_pack_ex(.., with_bvh = 1): a BVH whatever the triangle count (the plugin's second form of a small scene)"""
    lib = pt.load_library()
    v, t, m = pt.make_scene(V, T, M)
    n = lib.pt_hip_scene_blob_bytes_ex(len(T), len(M), 1)
    blob = (C.c_uint8 * n)()
    assert lib.pt_hip_scene_pack_ex(C.cast(v, C.c_void_p), len(V), C.cast(t, C.c_void_p), len(T), C.cast(m, C.c_void_p), len(M),
                                    C.cast(blob, C.c_void_p), n, 1) == 0, lib.pt_hip_last_error()
    assert lib.pt_hip_scene_has_bvh(C.cast(blob, C.c_void_p), n) == 1
    return bytes(blob)


def make_rays(n_rays, seed, camera, lo=(-1.0, -1.0, -0.5), hi=(1.0, 1.0, 1.5)):
    rng = np.random.default_rng(seed)
    lo, hi = np.array(lo), np.array(hi)
    rays = np.zeros((n_rays, 7), dtype=np.float32)
    k = n_rays // 4
    # camera rays, normalised over four lanes like the reference's (|d| < 1)
    d = np.stack([rng.uniform(-0.9, 0.9, k), rng.uniform(-0.5, 0.5, k), np.full(k, 2.0)], 1)
    d /= np.sqrt((d ** 2).sum(1) + 1.0)[:, None]
    rays[:k, 0:3] = camera
    rays[:k, 3:6] = d
    rays[:k, 6] = FLT_MAX
    # rays leaving the walls
    m = n_rays - k
    o = rng.uniform(lo, hi, (m, 3))
    face = rng.integers(0, 6, m)
    for i in range(m):
        a = face[i] // 2
        o[i, a] = lo[a] + 1e-4 if face[i] % 2 == 0 else hi[a] - 1e-4
    dd = rng.normal(size=(m, 3))
    dd /= np.linalg.norm(dd, axis=1)[:, None]
    for i in np.nonzero(rng.random(m) < 0.33)[0]:
        a = face[i] // 2
        dd[i, a] = rng.choice([1e-4, 1e-3, 1e-2, 5e-2]) * (1 if face[i] % 2 == 0 else -1)
        dd[i] /= np.linalg.norm(dd[i])
    dd *= rng.uniform(0.7, 1.0, (m, 1))
    rays[k:, 0:3] = o
    rays[k:, 3:6] = dd
    rays[k:, 6] = np.where(rng.random(m) < 0.5, FLT_MAX, rng.uniform(0.3, 2.5, m)).astype(np.float32)
    return rays


def mixed_scene():
    """big and tiny triangles in one tree (as in test_gpu_parity's mixed-size case)"""
    V, T, M = scenes.all_materials_scene()
    V2, T2, M2 = scenes.synthetic_scene(2000, seed=77)
    off, moff = len(V), len(M)
    V = V + [(x * 0.3, y * 0.3, z * 0.3 + 0.4, w_) for (x, y, z, w_) in V2]
    T = T + [(a + off, b + off, c + off, m + moff) for (a, b, c, m) in T2]
    return V, T, M + M2


def scene_by_name(name):
    if name == "mixed":
        return mixed_scene()
    if name.startswith("heap-"):
        return scenes.heap_of_large_triangles(int(name.split("-")[1]), 9)
    kind, n = name.split("-")
    return scenes.triangle_soup(int(n), 31) if kind == "soup" else scenes.synthetic_scene(int(n))


@pytest.mark.parametrize("name,n_rays,camera", [("synthetic-700", 6000, (0.0, 0.0, 0.0)), ("synthetic-1500", 6000, (0.0, 0.0, 0.0)), ("soup-900", 5000, (0.0, 0.0, 0.0)),
                                                ("synthetic-10000", 1500, (0.2, -0.1, 0.05)), ("mixed", 4000, (0.2, -0.1, 0.05)), ("heap-1500", 400, (0.0, 0.0, 0.0))])
@pytest.mark.parametrize("check", [CHECK, CHECK_SPILL], ids=["lds", "spill"])
def test_bvh_walk_equals_brute_force_on_the_host(pt, tmp_path, name, n_rays, camera, check):
    if not os.path.exists(check):
        pytest.skip("tests/host/bvh_host_check is not built (python -c 'import __graft_entry__ as g; g.build()')")
    V, T, M = scene_by_name(name)
    blob, rays = tmp_path / "scene.blob", tmp_path / "rays.f32"
    blob.write_bytes(pack(pt, V, T, M))
    make_rays(n_rays, 7, camera).tofile(rays)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0")
    env.pop("LD_PRELOAD", None)          # ASan must come first in the library list of its own process
    r = subprocess.run([check, str(blob), str(rays)] + [str(c) for c in camera], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    fields = r.stdout.split()
    assert int(fields[1]) == n_rays and int(fields[3]) > n_rays // 3 and int(fields[5]) == 0, r.stdout
    if name.startswith("heap-") and check == CHECK:
        # the heap of large triangles is there to fill the stack: with the library's own sizing the walk must have gone beyond the LDS levels
        assert int(fields[7]) > int(fields[9]), r.stdout


@pytest.mark.parametrize("n_tris", [1, 2, 3, 4, 5, 9, 17, 48])
def test_trees_of_a_few_triangles(pt, tmp_path, n_tris):
    """pt_hip_scene_pack_ex packs a BVH for ANY triangle count: a single leaf wrapped into a root, wide nodes with two or three
    children and empty slots, the all-materials scene (48 triangles: what the plugin holds in both forms).  Walk == loop, on the host."""
    if not os.path.exists(CHECK):
        pytest.skip("tests/host/bvh_host_check is not built (python -c 'import __graft_entry__ as g; g.build()')")
    V, T, M = scenes.all_materials_scene() if n_tris == 48 else scenes.triangle_soup(n_tris, 40 + n_tris)
    blob, rays = tmp_path / "scene.blob", tmp_path / "rays.f32"
    blob.write_bytes(pack_with_bvh(pt, V, T, M))
    make_rays(3000, 5, (0.0, 0.0, 0.0)).tofile(rays)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0")
    env.pop("LD_PRELOAD", None)
    for check in (CHECK, CHECK_SPILL):
        r = subprocess.run([check, str(blob), str(rays), "0", "0", "0"], capture_output=True, text=True, env=env, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        fields = r.stdout.split()
        assert int(fields[1]) == 3000 and int(fields[5]) == 0, r.stdout


@pytest.mark.parametrize("name,n_rays,lanes,first", [("synthetic-1500", 1400, 64, 0), ("synthetic-10000", 700, 64, 0), ("mixed", 1500, 64, 0),
                                                     ("synthetic-700", 333, 37, 0), ("synthetic-2500", 1200, 64, 1), ("soup-600", 600, 64, 0)])
def test_extend_stage_as_a_wave_of_host_threads(pt, tmp_path, name, n_rays, lanes, first):
    """extend_bvh_block (what k_extend<false, true> runs) with diverging rays: every queue slot gets the brute-force hit.
    37 lanes: a partial wave, a ray count that is no multiple of it.  first = 1: the bounce-0 form, rays from the camera."""
    if not os.path.exists(WAVE_CHECK):
        pytest.skip("tests/host/extend_wave_check is not built (python -c 'import __graft_entry__ as g; g.build()')")
    V, T, M = scene_by_name(name)
    blob, rays = tmp_path / "scene.blob", tmp_path / "rays.f32"
    blob.write_bytes(pack(pt, V, T, M))
    camera = (0.2, -0.1, 0.05)
    make_rays(n_rays, 11, camera).tofile(rays)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0")
    env.pop("LD_PRELOAD", None)
    r = subprocess.run([WAVE_CHECK, str(blob), str(rays)] + [str(c) for c in camera] + [str(lanes), str(first)], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    fields = r.stdout.split()
    assert int(fields[1]) == n_rays and int(fields[3]) > n_rays // 3 and int(fields[5]) == 0, r.stdout

"""Stage-by-stage GPU parity: each wavefront kernel, launched through its own pt_hip_* entry
point, against the oracle's stage functions on the same paths -- bit for bit, every bounce."""
import ctypes as C

import numpy as np
import pytest

import scenes
from gpu_util import DeviceBuffer, make_frame, queue_slots, sync, upload_scene

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("first,count,w,h", [(0, 5000, 100, 50), (777, 3000, 80, 60), (8294400 - 4096, 4096, 3840, 2160), (0, 1, 1, 1)])
def test_rng_seed_matches_naive_semantics(pt, oracle, first, count, w, h):
    lib = pt.load_library()
    fr = make_frame(pt, oracle, w, h, pixel_first=first, pixel_count=count)
    buf = DeviceBuffer(lib, count * 4)
    assert lib.pt_hip_rng_seed(C.byref(fr), buf.ptr, None) == 0, lib.pt_hip_last_error()
    sync(lib)
    got = buf.download(np.uint32)
    want = oracle.seed_states(1337, count, pixel_first=first)
    assert np.array_equal(got, want)
    # and against the literal serial loop of Functions.cuh:31-38 for a few of them
    for k in [0, count // 2, count - 1]:
        i = first + k
        assert got[k] == oracle.lib().pto_rng_init_naive(1337 + i, i // 32)


def _oracle_paths(oracle, p, n_pix, pixel_first, n_samples, frame_first, rng0, strided, scene):
    """Oracle state of every path after generate; list indexed by pid = s*n_pix + li."""
    L = oracle.lib()
    paths = []
    for s in range(n_samples):
        for li in range(n_pix):
            i = pixel_first + li
            rng = L.pto_rng_jump(int(rng0[li]), (frame_first + s) * p.stride) if strided else int(rng0[li])
            path = oracle.Path()
            L.pto_stage_generate(C.byref(p), i, rng, C.byref(path))
            paths.append(path)
    return paths


@pytest.mark.parametrize("fuse", [False, True], ids=["separate-extend", "fused-extend"])
@pytest.mark.parametrize("scene_name,strided,n_samples,depth", [("default", False, 1, 10), ("default", True, 3, 10),
                                                                ("allmat", False, 1, 16), ("allmat", True, 2, 6)])
def test_wavefront_stages_step_by_step(pt, oracle, scene_name, strided, n_samples, depth, fuse):
    lib, L = pt.load_library(), oracle.lib()
    w, h, pixel_first, n_pix = 40, 24, 100, 700
    V, T, M = scenes.default_scene() if scene_name == "default" else scenes.all_materials_scene()
    scene_o = oracle.make_scene(V, T, M)
    scene_dev = upload_scene(pt, lib, pt.make_scene(V, T, M))
    stride = 35 if scene_name == "default" else 2 + 33 * depth
    mode = 1 if strided else 0
    fr = make_frame(pt, oracle, w, h, pixel_first=pixel_first, pixel_count=n_pix, depth=depth, rng_mode=mode, stride=stride)
    p = oracle.params(w, h, depth=depth, rng_mode=mode, stride=stride)
    frame_first = 5 if strided else 0
    P = n_pix * n_samples
    G = lib.pt_hip_queue_blocks(P)

    rng = DeviceBuffer(lib, n_pix * 4)
    assert lib.pt_hip_rng_seed(C.byref(fr), rng.ptr, None) == 0
    sync(lib)
    rng0 = rng.download(np.uint32)
    q = [dict(a=DeviceBuffer(lib, P * 16), b=DeviceBuffer(lib, P * 16), c=DeviceBuffer(lib, P * 16), pid=DeviceBuffer(lib, P * 4)) for _ in range(2)]
    Q = [pt.Queue(a=x["a"].ptr, b=x["b"].ptr, c=x["c"].ptr, pid=x["pid"].ptr) for x in q]
    counts = DeviceBuffer(lib, (depth + 2) * G * 4)
    hits = [DeviceBuffer(lib, P * 8), DeviceBuffer(lib, P * 8)]
    radiance = DeviceBuffer(lib, P * 16).upload(np.full(P * 4, np.nan, dtype=np.float32))
    lit = DeviceBuffer(lib, P).upload(np.full(P, 77, dtype=np.uint8))
    smat = DeviceBuffer(lib, n_samples * 32 * 4)

    # ---- generate (a4)
    rc = lib.pt_hip_generate(C.byref(fr), frame_first, n_samples, rng.ptr, smat.ptr if strided else None, scene_dev.ptr, len(T),
                             Q[0], hits[0].ptr if fuse else None, P, counts.ptr, None)
    assert rc == 0, lib.pt_hip_last_error()
    sync(lib)
    paths = _oracle_paths(oracle, p, n_pix, pixel_first, n_samples, frame_first, rng0, strided, scene_o)
    B = q[0]["b"].download(np.float32).reshape(P, 4)       # bounce-0 layout: {d.x, d.y, d.z, rng}
    cnt = counts.download(np.uint32)[:G]
    assert cnt.sum() == P
    assert sorted(queue_slots(cnt, G).tolist()) == list(range(P))
    want_b = np.array([[x.direction.x, x.direction.y, x.direction.z, 0] for x in paths], dtype=np.float32)
    assert np.array_equal(B[:, :3].view(np.uint32), want_b[:, :3].view(np.uint32))
    assert all(x.origin.tuple() == (fr.origin.x, fr.origin.y, fr.origin.z, fr.origin.w) and x.t == np.finfo(np.float32).max for x in paths)
    assert np.array_equal(B[:, 3].view(np.uint32), np.array([x.rng for x in paths], dtype=np.uint32))

    # ---- bounces: extend (a5, a6) then shade (a7-a9)
    tris_o, verts_o, mats_o = C.cast(scene_o[1], C.c_void_p), C.cast(scene_o[0], C.c_void_p), C.cast(scene_o[2], C.c_void_p)
    alive = {pid for pid in range(P)}
    for bounce in range(depth):
        i_in, i_out = bounce & 1, (bounce & 1) ^ 1
        cin = C.c_void_p(counts.ptr + bounce * G * 4)
        cout = C.c_void_p(counts.ptr + (bounce + 1) * G * 4)
        hit = hits[i_in]
        if not fuse:
            rc = lib.pt_hip_extend(C.byref(fr), bounce, scene_dev.ptr, len(T), len(M), Q[i_in], cin, P, hit.ptr, None)
            assert rc == 0, lib.pt_hip_last_error()
        sync(lib)
        cnt = counts.download(np.uint32)[bounce * G:(bounce + 1) * G]
        slots = queue_slots(cnt, G)
        assert len(slots) == len(alive)
        pid_of_slot = slots if bounce == 0 else q[i_in]["pid"].download(np.uint32)[slots]
        assert sorted(int(x) for x in pid_of_slot) == sorted(alive)
        H = hit.download(np.uint32).reshape(P, 2)
        for slot, pid in zip(slots, pid_of_slot):
            path = paths[int(pid)]
            L.pto_stage_extend(C.byref(path), tris_o, len(T), verts_o)
            want_t = np.array([path.t], dtype=np.float32).view(np.uint32)[0]
            want_tri = 0xFFFFFFFF if path.hit_tri < 0 else path.hit_tri
            assert (H[slot, 0], H[slot, 1]) == (want_t, want_tri), (bounce, int(pid))
        rc = lib.pt_hip_shade(C.byref(fr), scene_dev.ptr, len(T), len(M), bounce, Q[i_in], hit.ptr, cin, P, Q[i_out],
                              hits[i_out].ptr if fuse else None, cout, radiance.ptr, lit.ptr, rng.ptr if not strided else None, None)
        assert rc == 0, lib.pt_hip_last_error()
        sync(lib)
        n_diffuse = 0
        for pid in list(alive):
            path = paths[pid]
            L.pto_stage_shade(C.byref(path), mats_o)      # no-op for paths the extend stage dropped
            n_diffuse += int(path.alive)
            if not path.alive or bounce == depth - 1:
                alive.discard(pid)
        cnt_out = counts.download(np.uint32)[(bounce + 1) * G:(bounce + 2) * G]
        assert cnt_out.sum() == n_diffuse
        if bounce < depth - 1:
            slots = queue_slots(cnt_out, G)
            oa = q[i_out]["a"].download(np.float32).reshape(P, 4)[slots]
            ob = q[i_out]["b"].download(np.float32).reshape(P, 4)[slots]
            oc = q[i_out]["c"].download(np.float32).reshape(P, 4)[slots]
            opid = q[i_out]["pid"].download(np.uint32)[slots]
            assert sorted(int(x) for x in opid) == sorted(alive)
            for k, pid in enumerate(opid):
                x = paths[int(pid)]
                wa = np.array([x.origin.x, x.origin.y, x.origin.z, x.direction.x], dtype=np.float32)
                wb = np.array([x.direction.y, x.direction.z, x.t], dtype=np.float32)
                wc = np.array(x.step.tuple(), dtype=np.float32)
                assert np.array_equal(oa[k].view(np.uint32), wa.view(np.uint32)), (bounce, int(pid))
                # with the fused extend b.z already holds the NEXT bounce's accepted t (checked via hit[] above)
                ncmp = 2 if fuse else 3
                assert np.array_equal(ob[k, :ncmp].view(np.uint32), wb[:ncmp].view(np.uint32))
                assert ob[k, 3:4].view(np.uint32)[0] == x.rng
                assert np.array_equal(oc[k].view(np.uint32), wc.view(np.uint32))
        if not alive:
            break
    # every path wrote its radiance exactly once
    Lt = lit.download(np.uint8)
    assert set(np.unique(Lt).tolist()) <= {0, 1}              # every path set its flag exactly once
    R = radiance.download(np.float32).reshape(P, 4)
    R[Lt == 0] = np.array([0, 0, 0, 1], dtype=np.float32)     # unlit samples are (0,0,0,1) by definition
    assert not np.isnan(R).any()
    want_r = np.array([paths[pid].frame.tuple() for pid in range(P)], dtype=np.float32)
    assert np.array_equal(R.view(np.uint32), want_r.view(np.uint32))
    assert np.array_equal(Lt == 1, np.array([paths[pid].frame.tuple() != (0.0, 0.0, 0.0, 1.0) for pid in range(P)]))
    if not strided:
        got_rng = rng.download(np.uint32)
        assert np.array_equal(got_rng, np.array([paths[pid].rng for pid in range(P)], dtype=np.uint32))

    # ---- accumulate (a10)
    mean0 = np.random.default_rng(1).random((n_pix, 4), dtype=np.float32)
    mean = DeviceBuffer(lib, n_pix * 16).upload(mean0)
    mean_first = 7
    assert lib.pt_hip_accumulate(n_pix, mean_first, n_samples, radiance.ptr, lit.ptr, mean.ptr, None) == 0
    sync(lib)
    got = mean.download(np.float32).reshape(n_pix, 4)
    want = mean0.copy()
    for s in range(n_samples):
        r = np.float32(1.0) / np.float32(mean_first + s + 1)
        want = (R[s * n_pix:(s + 1) * n_pix] * r + (want * np.float32(mean_first + s)) * r).astype(np.float32)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("shards", [1, 2, 4, 8, 16])
def test_combine_shards_kernel(pt, oracle, shards):
    lib = pt.load_library()
    n = 1000
    src = np.random.default_rng(shards).random((shards, n, 4), dtype=np.float32) * 10
    d_in = DeviceBuffer(lib, src.nbytes).upload(src)
    d_out = DeviceBuffer(lib, n * 16)
    assert lib.pt_hip_combine_shards(d_in.ptr, shards, n, d_out.ptr, None) == 0, lib.pt_hip_last_error()
    sync(lib)
    got = d_out.download(np.float32).reshape(n, 4)
    want = np.empty((n, 4), dtype=np.float32)
    oracle.lib().pto_combine_shards(src.ctypes.data, shards, n, want.ctypes.data)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert lib.pt_hip_combine_shards(d_in.ptr, 3, n, d_out.ptr, None) == -1

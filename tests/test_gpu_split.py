"""Multi-GPU splits behind the plugin boundary (include/pt_plugin.h; no reference counterpart: single device,
PathTracer.cu:25) and PT_ENGINE_AUTO, on one GPU: the instances of a split share device 0 here, the code path is the one
G GPUs run (per-instance device binding, peer copies, the RCCL exchange with a communicator of one rank).

Bar: bit for bit.  Sample split: the image over 8 virtual shards is the oracle's pto_render_sharded_span and does not
depend on the number of instances.  Tile split: the tiles side by side are the single-instance sequential image."""
import ctypes as C
import os
import subprocess
import threading

import numpy as np
import pytest

import scenes

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ENGINES = [("fused", 1), ("wavefront", 0)]


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def assert_bit_equal(img, ref, what):
    diff = (bits(img) != bits(ref)).any(axis=1)
    assert not diff.any(), f"{what}: {int(diff.sum())} of {len(diff)} pixels differ; first at {int(np.argmax(diff))}: {img[np.argmax(diff)]} vs {ref[np.argmax(diff)]}"


def _sharded_reference(oracle, w, h, per_shard, depth=10, stride=35, scene=None):
    p = oracle.params(w, h, depth=depth, rng_mode=oracle.RNG_STRIDED, stride=stride, spp=per_shard * 8)
    return oracle.render_sharded(p, 8, scene=scene, span=1 << 20)


@pytest.mark.parametrize("name,engine", ENGINES)
def test_sample_split_single_instance(pt, oracle, name, engine):
    """one instance renders all 8 virtual shards and combines them itself on readback"""
    w, h, per = 65, 33, 3
    want, st_ref = _sharded_reference(oracle, w, h, per)
    with pt.PathTracer(w, h) as tr:
        tr.set_options(engine=engine, rng_mode=pt.PT_RNG_STRIDED, stride=35, split_mode=pt.PT_SPLIT_SAMPLE, samples_per_pass=2)
        assert tr.shard_layout() == (8, w * h, 8)
        tr.render(2)
        tr.render(1)
        img, st = tr.readback(), tr.stats()
    assert_bit_equal(img, want, name)
    assert st.frames == per and st.rays == st_ref.rays and st.paths == w * h * per * 8


@pytest.mark.parametrize("instances", [1, 2, 4, 8])
def test_sample_split_image_does_not_depend_on_the_instance_count(pt, oracle, instances):
    """G instances (threads, one each) render their shares of the 8 virtual shards, pt_combine_instances puts the image in
    instance 0: the oracle's sharded mean for every G, on a scene whose paths bounce (all materials, depth 12)"""
    V, T, M = scenes.all_materials_scene()
    w, h, per, depth = 97, 41, 4, 12
    stride = 2 + 33 * depth
    want, st_ref = _sharded_reference(oracle, w, h, per, depth=depth, stride=stride, scene=oracle.make_scene(V, T, M))
    tracers = [pt.PathTracer(w, h, device=0) for _ in range(instances)]
    try:
        for g, tr in enumerate(tracers):
            tr.set_scene(*pt.make_scene(V, T, M))
            tr.set_options(engine=g % 2, depth=depth, rng_mode=pt.PT_RNG_STRIDED, stride=stride, samples_per_pass=3,
                           split_mode=pt.PT_SPLIT_SAMPLE, split_rank=g, split_ranks=instances)
            assert tr.shard_layout()[0] == 8 // instances
        errors = []

        def drive(tr):
            try:
                tr.render(per - 1)
                tr.BeginDrawing()
                tr.FinalizeDrawing()
            except Exception as e:      # noqa: BLE001
                errors.append(e)
        threads = [threading.Thread(target=drive, args=(tr,)) for tr in tracers]
        [t.start() for t in threads]
        [t.join() for t in threads]
        assert not errors, errors
        if instances > 1:
            with pytest.raises(pt.PathTracerError):
                tracers[0].readback()                       # no combine yet
        pt.combine_instances(tracers[::-1])                 # any order; the image lands in the first handle passed
        img = tracers[-1].readback()
        rays = sum(tr.stats().rays for tr in tracers)
    finally:
        for tr in tracers:
            tr.Destroy()
    assert_bit_equal(img, want, f"{instances} instances")
    assert rays == st_ref.rays


def test_dist_combine_with_a_communicator_of_one_rank(pt, oracle):
    """pt_dist_*: librccl is loaded, a communicator made from a unique id, and the exchange (grouped ncclSend/ncclRecv of
    the slices, tree, ncclAllGather) runs through RCCL -- with one rank, which is all one GPU allows"""
    w, h, per = 64, 36, 2
    want, _ = _sharded_reference(oracle, w, h, per)
    with pt.PathTracer(w, h, device=0) as tr:
        tr.set_options(engine=pt.PT_ENGINE_FUSED, rng_mode=pt.PT_RNG_STRIDED, stride=35, split_mode=pt.PT_SPLIT_SAMPLE, split_rank=0, split_ranks=1)
        tr.dist_init(pt.dist_unique_id())
        tr.render(per)
        tr.dist_combine()
        img = tr.readback()
    assert_bit_equal(img, want, "dist_combine, world of one")


@pytest.mark.parametrize("name,engine", ENGINES)
def test_tile_split_option(pt, oracle, name, engine):
    w, h, spp, ranks = 80, 45, 3, 3
    ref, ref_rng, _, _ = oracle.render(oracle.params(w, h, spp=spp))
    parts, rngs, at = [], [], 0
    for g in range(ranks):
        with pt.PathTracer(w, h) as tr:
            tr.set_options(engine=engine, split_mode=pt.PT_SPLIT_TILE, split_rank=g, split_ranks=ranks)
            first, count = tr.size()[2:]
            assert first == at
            at += count
            tr.render(spp)
            parts.append(tr.readback())
            rngs.append(tr.readback_rng())
    assert at == w * h
    assert_bit_equal(np.concatenate(parts), ref, name)
    assert np.array_equal(np.concatenate(rngs), ref_rng)


def test_split_option_errors(pt):
    with pt.PathTracer(32, 16) as tr:
        for bad in (dict(split_mode=pt.PT_SPLIT_SAMPLE),                                           # sequential streams
                    dict(split_mode=pt.PT_SPLIT_SAMPLE, rng_mode=1, split_ranks=3),                 # 3 does not divide 8
                    dict(split_mode=pt.PT_SPLIT_SAMPLE, rng_mode=1, split_ranks=2, split_rank=2),
                    dict(split_mode=pt.PT_SPLIT_SAMPLE, rng_mode=1, virtual_shards=6),
                    dict(split_mode=pt.PT_SPLIT_TILE, split_ranks=4, split_rank=4),
                    dict(split_mode=7)):
            with pytest.raises(pt.PathTracerError):
                tr.set_options(**bad)
        tr.render(1)                                            # the instance still works


@pytest.mark.parametrize("instances,mode", [(2, "sample"), (4, "sample"), (3, "tile")])
def test_cpp_multi_gpu_host(pt, oracle, tmp_path, instances, mode):
    """tests/host/multi_gpu_host.cpp: G instances created with pt_create_on_device, one std::thread each calling through
    the IPathTracer vtable, pt_combine_instances / tile readback -- the native multi-GPU surface, no Python in it"""
    w, h, frames, spf = 120, 68, 2, 3
    out = tmp_path / "multi.raw"
    args = [os.path.join(ROOT, "tests", "host", "multi_gpu_host"), str(instances), str(w), str(h), str(frames), str(spf), str(out)]
    if mode == "tile":
        args.append("tile")
    r = subprocess.run(args, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    img = np.fromfile(out, dtype=np.float32).reshape(-1, 4)
    if mode == "tile":
        want, _, _, _ = oracle.render(oracle.params(w, h, spp=frames * spf))
    else:
        want, _ = _sharded_reference(oracle, w, h, frames * spf)
    assert_bit_equal(img, want, f"C++ host, {instances} instances, {mode} split")
    assert f"instances {instances}" in r.stdout


@pytest.mark.parametrize("scene_name", ["default", "materials", "synthetic2500"])
def test_engine_auto_probes_then_keeps_one_engine(pt, oracle, scene_name):
    """PT_ENGINE_AUTO (the default): the first renders measure one candidate each -- fused in step, fused out of step (scenes
    without a BVH), wavefront, and for a small scene the plugin holds in both forms (more than PT_BVH_DUAL_MIN_TRIANGLES triangles, no
    BVH by the threshold: the 48-triangle all-materials scene) the two engines through the BVH as well -- then the fastest stays; the
    image is the oracle's whichever runs, and a camera restart keeps the decision"""
    V, T, M = {"default": scenes.default_scene, "materials": scenes.all_materials_scene, "synthetic2500": lambda: scenes.synthetic_scene(2500)}[scene_name]()
    w, h = 200, 120
    has_bvh = len(T) > pt.PT_BVH_DEFAULT_MIN_TRIANGLES
    dual = not has_bvh and len(T) > pt.PT_BVH_DUAL_MIN_TRIANGLES
    # one unmeasured render (a cold device would count against whoever goes first), then two renders per candidate (the better one
    # counts) -- six where a render takes less than a quarter of a millisecond, as these do: BVH scenes have 2 candidates, small
    # scenes held in both forms 5, the rest 3
    n_cands = 2 if has_bvh else 5 if dual else 3
    with pt.PathTracer(w, h) as tr:
        assert tr.get_options().engine == pt.PT_ENGINE_AUTO
        tr.set_scene(*pt.make_scene(V, T, M))
        assert tr.stats().engine_in_use == pt.PT_ENGINE_AUTO and tr.stats().small_scene_dual == (1 if dual else 0)
        n_probe = 0
        while tr.stats().engine_in_use == pt.PT_ENGINE_AUTO:                # still measuring
            assert n_probe < 1 + 6 * n_cands
            tr.render(1)
            n_probe += 1
        assert n_probe >= 1 + 2 * n_cands
        ref, ref_rng, st_ref, _ = oracle.render(oracle.params(w, h, spp=n_probe + 1), scene=oracle.make_scene(V, T, M))
        st = tr.stats()
        assert st.engine_in_use in (pt.PT_ENGINE_FUSED, pt.PT_ENGINE_WAVEFRONT)
        rates = {"wavefront": st.auto_rays_per_ms[0], "fused": st.auto_rays_per_ms[1]}
        if not has_bvh:
            rates["fused out of step"] = st.fused_out_of_step_rays_per_ms
        else:
            assert st.fused_out_of_step_rays_per_ms == 0 and st.fused_out_of_step == 0
        if dual:
            rates["wavefront, BVH"], rates["fused, BVH"] = st.walk_rays_per_ms[0], st.walk_rays_per_ms[1]
        else:
            assert st.walk_rays_per_ms[0] == 0 and st.walk_rays_per_ms[1] == 0 and st.small_scene_walks == 0
        assert all(v > 0 for v in rates.values()), rates
        best = max(rates, key=rates.get)
        assert st.engine_in_use == (pt.PT_ENGINE_WAVEFRONT if best.startswith("wavefront") else pt.PT_ENGINE_FUSED)
        assert st.fused_out_of_step == (1 if best == "fused out of step" else 0)
        assert st.small_scene_walks == (1 if best.endswith("BVH") else 0)
        tr.render(1)
        img, rng, st = tr.readback(), tr.readback_rng(), tr.stats()
        assert_bit_equal(img, ref, f"auto on {scene_name}")
        assert np.array_equal(rng, ref_rng) and st.rays == st_ref.rays
        chosen = st.engine_in_use
        tr.Update(pt.CameraData(fieldOfView=80.0))            # camera restart: same decision, fresh accumulation
        tr.render(2)
        assert tr.stats().engine_in_use == chosen and tr.stats().frames == 2
        img2 = tr.readback()
    ref2, _, _, _ = oracle.render(oracle.params(w, h, cam=oracle.make_camera(fov=80.0), spp=2), scene=oracle.make_scene(V, T, M))
    assert_bit_equal(img2, ref2, f"auto on {scene_name} after a camera restart")


def test_candidates_are_measured_again_when_the_samples_per_call_change(pt, oracle, monkeypatch):
    """the fastest kernels at one sample per call need not be the fastest at 64: a fourfold change of the samples per render
    starts the measurement over (the image stays the oracle's throughout).  PT_PROBE_RENDERS pins the renders per candidate: renders as
    short as this test's would get six each (test_engine_auto_probes_then_keeps_one_engine runs that way)."""
    monkeypatch.setenv("PT_PROBE_RENDERS", "2")
    V, T, M = scenes.all_materials_scene()
    w, h = 160, 90
    with pt.PathTracer(w, h) as tr:
        tr.set_scene(*pt.make_scene(V, T, M))
        tr.set_options(engine=pt.PT_ENGINE_FUSED, depth=12)
        for _ in range(7):                                 # one unmeasured render, then two per form
            tr.render(1)
        first = tr.stats()
        # every form measured at 1 sample per call: in step, out of step, and -- 48 triangles: the plugin holds the scene in both forms -- through the BVH
        assert first.auto_rays_per_ms[1] > 0 and first.fused_out_of_step_rays_per_ms > 0 and first.walk_rays_per_ms[1] > 0 and first.small_scene_dual == 1
        tr.render(2)                                       # not a fourfold change: nothing is measured again
        assert tr.stats().auto_rays_per_ms[1] == first.auto_rays_per_ms[1]
        tr.render(16); tr.render(16)                       # measured again: in step first ...
        mid = tr.stats()
        assert mid.auto_rays_per_ms[1] != first.auto_rays_per_ms[1] and mid.fused_out_of_step_rays_per_ms == 0
        tr.render(16); tr.render(16)                       # ... then out of step ...
        assert tr.stats().fused_out_of_step_rays_per_ms > 0 and tr.stats().walk_rays_per_ms[1] == 0
        tr.render(16); tr.render(16)                       # ... then through the BVH
        last = tr.stats()
        assert last.walk_rays_per_ms[1] > 0
        best = max(last.auto_rays_per_ms[1], last.fused_out_of_step_rays_per_ms, last.walk_rays_per_ms[1])
        assert last.fused_out_of_step == (1 if last.fused_out_of_step_rays_per_ms == best else 0)
        assert last.small_scene_walks == (1 if last.walk_rays_per_ms[1] == best else 0)
        # back to one sample per call: that class has been decided before and is NOT measured again (a host that alternates between
        # 1 sample while the camera moves and 16 when it rests pays for each class once) -- the remembered rates and choice return
        tr.render(1)
        again = tr.stats()
        assert again.auto_rays_per_ms[1] == first.auto_rays_per_ms[1] and again.fused_out_of_step_rays_per_ms == first.fused_out_of_step_rays_per_ms
        assert again.fused_out_of_step == first.fused_out_of_step
        tr.render(20)                                      # and the 16..63 class likewise
        assert tr.stats().auto_rays_per_ms[1] == last.auto_rays_per_ms[1] and tr.stats().fused_out_of_step == last.fused_out_of_step
        # a Resize to another size and back: the first size's decision is still there (no new measurement), the accumulation restarts
        tr.Resize(w + 8, h)
        tr.render(1)
        tr.Resize(w, h)
        tr.render(1)
        back = tr.stats()
        assert back.frames == 1 and back.auto_rays_per_ms[1] == first.auto_rays_per_ms[1] and back.fused_out_of_step == first.fused_out_of_step
        tr.set_options(engine=pt.PT_ENGINE_FUSED, depth=12)          # restart for the image check below
        for spp_ in (1, 1, 1, 1, 2, 16, 16, 16):
            tr.render(spp_)
        img = tr.readback()
    ref, _, _, _ = oracle.render(oracle.params(w, h, depth=12, spp=4 + 2 + 48), scene=oracle.make_scene(V, T, M))
    assert_bit_equal(img, ref, "all materials, fused engine across two measurements")


@pytest.mark.parametrize("scene_name,shards,flags", [("default", 8, 0), ("materials", 4, 0), ("materials", 3, "oos"), ("synthetic700", 4, 0), ("default", 16, 0)])
def test_several_shards_in_one_launch_equal_a_launch_per_shard(pt, oracle, scene_name, shards, flags):
    """pt_hip_render_fused_shards through the C-ABI against one pt_hip_render_fused per shard: every shard's running mean and the
    counters bit for bit, on two calls in a row (the second continues the means).  Brute-force scenes take the one-launch kernel,
    the BVH scene and the out-of-step form the entry point's own loop over the shards."""
    import ctypes as C
    from gpu_util import DeviceBuffer, make_frame, sync, upload_scene
    lib = pt.load_library()
    V, T, M = {"default": scenes.default_scene, "materials": scenes.all_materials_scene, "synthetic700": lambda: scenes.synthetic_scene(700)}[scene_name]()
    w, h, depth = 97, 41, 12
    n, pitch = w * h, w * h + 37
    scene_dev = upload_scene(pt, lib, pt.make_scene(V, T, M))
    fr = make_frame(pt, oracle, w, h, depth=depth, rng_mode=1, stride=2 + 33 * depth, flags=scene_dev.frame_flags | (pt.PT_FRAME_OUT_OF_STEP if flags == "oos" else 0))
    firsts = (C.c_uint32 * shards)(*[5 + 1000 * k for k in range(shards)])
    out = []
    for one_call in (False, True):
        rng = DeviceBuffer(lib, n * 4)
        assert lib.pt_hip_rng_seed(C.byref(fr), rng.ptr, None) == 0, lib.pt_hip_last_error()
        mean = DeviceBuffer(lib, shards * pitch * 16).upload(np.ones(shards * pitch * 4, dtype=np.float32))
        stats = DeviceBuffer(lib, pt.PT_STATS_WORDS * 8).upload(np.zeros(pt.PT_STATS_WORDS, dtype=np.uint64))
        for mean_first, samples in ((0, 3), (3, 2)):
            for k in range(shards):
                firsts[k] = 5 + 1000 * k + mean_first
            if one_call:
                assert lib.pt_hip_render_fused_shards(C.byref(fr), scene_dev.ptr, len(T), len(M), firsts, shards, mean_first, samples,
                                                      rng.ptr, mean.ptr, pitch, stats.ptr, None) == 0, lib.pt_hip_last_error()
            else:
                for k in range(shards):
                    assert lib.pt_hip_render_fused(C.byref(fr), scene_dev.ptr, len(T), len(M), firsts[k], mean_first, samples,
                                                   rng.ptr, mean.ptr + k * pitch * 16, stats.ptr, None) == 0, lib.pt_hip_last_error()
        sync(lib)
        out.append((mean.download(np.uint32).reshape(shards, pitch, 4)[:, :n], stats.download(np.uint64)[:8]))
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
    assert not np.array_equal(out[0][0][0], out[0][0][1])                       # the shards did render different samples
    # limits
    assert lib.pt_hip_render_fused_shards(C.byref(fr), scene_dev.ptr, len(T), len(M), firsts, 17, 0, 1, rng.ptr, mean.ptr, pitch, None, None) == -5      # PT_ERR_LIMIT
    assert lib.pt_hip_render_fused_shards(C.byref(fr), scene_dev.ptr, len(T), len(M), firsts, 2, 0, 1, rng.ptr, mean.ptr, n - 1, None, None) == -1      # PT_ERR_INVALID

#!/usr/bin/env python3
"""bench.py -- rays/s of the path-tracing hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (config.workload): BASELINE.json configs[1] -- the reference's default scene
(PT_CudaSimple/PathTracer.cu:48-69), default camera, 1920x1080, depth 10, seed 1337.  One step =
one pass of the hot path over one batch: 64 more samples per pixel folded into the running mean
(the reference's 64 BeginDrawing/FinalizeDrawing pairs), i.e. 2*N*64 = 265.4 M rays per GPU.
A "ray" is one extend-stage work item; the count comes from device counters.

N > 1 is xGMI sample split with weak scaling: every rank renders 64 *different* samples per
step of the same frame (strided RNG, stride 35 = reference-exact on this scene), then the ranks'
running means are combined in a fixed pairwise order (all_to_all of image slices over RCCL ->
tree combine kernel -> all_gather), so the image is bit-identical whatever N is.

The JSON line also carries `roofline` (algorithmic bytes of the dominant kernel / its device
time measured with HIP events on the plugin's stream, against the 8 TB/s HBM peak) and
`cpu_baseline` (the CPU oracle -- a port of the reference path -- timed on this box's cores).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np

WIDTH, HEIGHT, SPP_PER_STEP, DEPTH = 1920, 1080, 64, 10
HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# VALU issue peak: 256 CUs x 4 SIMDs, one wave64 instruction per 2 cycles per SIMD at 2.4 GHz (same guide: v_fma_f32
# 2 cycles on the SIMD-32; tools/valu_microbench.hip measures 2.4-2.6 at the clock the chip holds under load)
VALU_PEAK_GINST = 256 * 4 * 2.4 / 2.0
# SURVEY 8d algorithmic traffic of the wavefront formulation per pixel-sample on the default
# scene: generate 52 + extend 36 + shade-diffuse 104 + extend 36 + shade-emissive 44*p +
# accumulate 48 = 276 + 44 p bytes for 2 rays; p = share of secondary rays that reach the light
# (measured by the oracle at 1080p: 0.0473).
P_LIGHT = 0.0473
BYTES_PER_RAY = (276.0 + 44.0 * P_LIGHT) / 2.0


def cpu_baseline(spp):
    """The oracle (kind 'port': our C restatement of the reference path) on the host cores: `spp` samples over all of
    them, and ('also report 1-thread') a twentieth of that on one thread."""
    from oracle import pt_oracle_py as po
    import ctypes as C
    po.lib()
    cores = int(os.environ.get("OMP_NUM_THREADS", "0")) or len(os.sched_getaffinity(0))
    n = WIDTH * HEIGHT
    v, t, m = po.default_scene()

    def timed(threads, samples):
        p = po.params(WIDTH, HEIGHT, depth=DEPTH, spp=samples, threads=threads)
        rng = po.seed_states(1337, n)                   # RNG seeding excluded, as on the GPU side
        mean = np.ones((n, 4), dtype=np.float32)
        st = po.Stats()
        t0 = time.perf_counter()
        rc = po.lib().pto_render(C.byref(p), C.cast(v, C.c_void_p), 8, C.cast(t, C.c_void_p), 4, C.cast(m, C.c_void_p), 2,
                                 None, 0, n, rng.ctypes.data, mean.ctypes.data, None, C.byref(st))
        dt = time.perf_counter() - t0
        assert rc == 0
        return st.rays, dt

    rays, dt = timed(cores, spp)
    one_spp = max(1, spp // 20)
    rays1, dt1 = timed(1, one_spp)
    return {"value": round(rays / dt / 1e6, 2), "unit": "Mrays/s", "cores": cores, "kind": "port",
            "sample": f"default scene {WIDTH}x{HEIGHT}, {spp} spp, depth {DEPTH}: {rays} rays in {dt:.2f} s wall",
            "one_thread": {"value": round(rays1 / dt1 / 1e6, 2), "unit": "Mrays/s",
                           "sample": f"{one_spp} spp: {rays1} rays in {dt1:.2f} s"}}


def parity_sample(tracer, w, h, spp_total, n_pixels=2048):
    """BASELINE.json's third metric, 'RMSE vs reference': a bounded sample of pixels of the image the timed steps
    produced, against the oracle rendering the same pixels and samples (sequential RNG = the reference's stream)."""
    from oracle import pt_oracle_py as po
    img = tracer.readback()
    ids = np.unique(np.random.default_rng(7).integers(0, w * h, n_pixels)).astype(np.uint32)
    ref, _, _, _ = po.render(po.params(w, h, depth=DEPTH, spp=spp_total), pixel_ids=ids)
    got = img[ids]
    equal = int((got.view(np.uint32) == ref.view(np.uint32)).all(axis=1).sum())
    diff = got.astype(np.float64) - ref.astype(np.float64)
    rel = np.abs(diff) / np.maximum(np.abs(ref.astype(np.float64)), 1e-6)
    return {"pixels_checked": int(len(ids)), "pixels_bit_identical": equal, "rmse": float(np.sqrt((diff ** 2).mean())),
            "max_rel_err": float(rel.max()), "spp": int(spp_total), "against": "oracle (CPU restatement of the reference path), same pixels and samples"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--engine", choices=["fused", "wavefront"], default="fused")
    ap.add_argument("--samples-per-pass", type=int, default=16, help="wavefront strided batching")
    ap.add_argument("--cpu-spp", type=int, default=48, help="samples of the CPU baseline leg (0 = skip); 48 spp at 1080p = 199 M rays, about 16 CPU-seconds over the box's 16 threads")
    ap.add_argument("--force-dist", action="store_true", help="run the RCCL exchange even with one rank (self-test of the N>1 path)")
    ap.add_argument("--width", type=int, default=WIDTH)
    ap.add_argument("--height", type=int, default=HEIGHT)
    args = ap.parse_args()

    import torch
    import pathtracer_amd as pt

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        if "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
        # The exchange must run BESIDE the next step's render, whose 8100 workgroups otherwise keep every CU slot: the
        # collectives and the combine kernel go on high-priority streams (measured with one rank: 1413 -> 1340 us per
        # step against 1322 us for the render alone)
        os.environ.setdefault("TORCH_NCCL_HIGH_PRIORITY", "1")
        pg_options = None
        try:
            pg_options = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
        except Exception:
            pg_options = None
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), pg_options=pg_options)

    w, h = args.width, args.height
    n_pix = w * h
    engine = pt.PT_ENGINE_FUSED if args.engine == "fused" else pt.PT_ENGINE_WAVEFRONT
    total_steps = args.warmup + args.steps
    tracer = pt.PathTracer(w, h, device=local_rank)
    if dist:
        # rank r owns samples [r*S, (r+1)*S) of the frame, S = all its steps' samples
        tracer.set_options(engine=engine, rng_mode=pt.PT_RNG_STRIDED, stride=35, depth=DEPTH,
                           samples_per_pass=args.samples_per_pass, sample_offset=rank * total_steps * SPP_PER_STEP)   # = sample_split.shard_samples
    elif args.engine == "wavefront":
        tracer.set_options(engine=engine, rng_mode=pt.PT_RNG_STRIDED, stride=35, depth=DEPTH, samples_per_pass=args.samples_per_pass)
    else:
        tracer.set_options(engine=engine, depth=DEPTH)
    lib = pt.load_library()

    exchange = None
    if dist:
        assert n_pix % world == 0, "image must split evenly over the ranks"
        slice_floats = n_pix * 4 // world
        # the exchange works on a snapshot of the running mean, so the next step's samples render (plugin
        # stream) while RCCL moves the previous step's image (torch/RCCL streams)
        snapshot = torch.empty(n_pix * 4, device="cuda", dtype=torch.float32)
        gathered = torch.empty(world * slice_floats, device="cuda", dtype=torch.float32)   # shard-major slices
        image = torch.empty(n_pix * 4, device="cuda", dtype=torch.float32)
        my_slice = torch.empty(slice_floats, device="cuda", dtype=torch.float32)
        tracer.set_options(samples_per_begin=SPP_PER_STEP)
        plugin_stream = torch.cuda.ExternalStream(tracer.stream())
        exchange_stream = torch.cuda.Stream(priority=-1)
        exchanged = torch.cuda.Event()

        from pathtracer_amd import sample_split

        def combine_slices(stacked, shards, n_pix_slice, out_slice):
            rc = lib.pt_hip_combine_shards(stacked.data_ptr(), shards, n_pix_slice, out_slice.data_ptr(),
                                           exchange_stream.cuda_stream)
            assert rc == 0, lib.pt_hip_last_error()

        def exchange():
            # all_to_all of image slices -> fixed-order tree on this rank's slice -> all_gather; enqueued, not awaited
            with torch.cuda.stream(exchange_stream):
                sample_split.exchange_and_combine(snapshot, dist, combine_slices, out=image, scratch=gathered, slice_out=my_slice)

    def step():
        if not exchange:
            tracer.render(SPP_PER_STEP)                       # blocks until the samples are in the mean
            return
        # one step = 64 more samples into this rank's mean + one exchange of a whole image.  Enqueue order: the
        # render (plugin stream; BeginDrawing returns at once), the exchange of the PREVIOUS step's snapshot beside
        # it (torch/RCCL streams), then -- on the plugin stream, behind both -- the copy of the new mean into the
        # snapshot; FinalizeDrawing is the step's one host wait.
        tracer.BeginDrawing()
        exchange()
        exchanged.record(exchange_stream)
        plugin_stream.wait_event(exchanged)                   # the snapshot is free to overwrite
        tracer.copy_mean_async(snapshot.data_ptr(), n_pix * 4)
        tracer.FinalizeDrawing()

    def sync():
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
            torch.cuda.synchronize()

    if exchange:
        tracer.copy_mean(snapshot.data_ptr(), n_pix * 4)       # the first step exchanges the (empty) start image
    for _ in range(args.warmup):
        step()
    sync()
    st0 = tracer.stats()
    rays0, ms0, renders0 = st0.rays, st0.total_render_ms, st0.renders
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    dt = time.perf_counter() - t0
    st = tracer.stats()
    rays_rank = st.rays - rays0
    # average device time of one render = one k_fused launch (or one wavefront pass sequence):
    # HIP events recorded around it on the plugin's own stream
    last_ms = (st.total_render_ms - ms0) / max(1, st.renders - renders0)

    if dist:
        t = torch.tensor([dt, float(rays_rank)], device="cuda", dtype=torch.float64)
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        dt, rays_total = float(tmax[0]), float(t[1])
    else:
        rays_total = float(rays_rank)

    if rank == 0:
        rays_per_launch = rays_rank / max(1, args.steps)
        achieved = rays_per_launch * BYTES_PER_RAY / (last_ms * 1e-3) / 1e9 if last_ms > 0 else 0.0
        traffic = valu_insts = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                pmc = json.load(open(tpath))
                traffic = pmc.get(f"{args.engine}_{w}x{h}_{SPP_PER_STEP}spp")
                valu_insts = pmc.get(f"{args.engine}_{w}x{h}_{SPP_PER_STEP}spp_valu_insts")
            except Exception:
                traffic = valu_insts = None
        out = {
            "metric": "Mrays/s (primary+bounce)",
            "value": round(rays_total / dt / 1e6, 1),
            "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "spp_per_s": round(SPP_PER_STEP * world * args.steps / dt, 1),
            "config": {"workload": f"reference default scene {w}x{h}, {SPP_PER_STEP} spp per step per GPU, depth {DEPTH}, seed 1337 "
                                   f"(BASELINE.json configs[1])",
                       "engine": args.engine, "rng": "sequential" if (not dist and args.engine == "fused") else "strided/35",
                       "parallelism": "single GPU" if not dist else f"sample split x{world}, fixed-order combine over RCCL (all_to_all + all_gather), overlapped with the next step"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "kernel": "k_fused" if args.engine == "fused" else "wavefront step: 4 passes of 16 spp, one k_bounces launch each (camera rays + bounce loop + fold)",
                         "kernel_ms": round(last_ms, 4), "algorithmic_bytes_per_ray": round(BYTES_PER_RAY, 2),
                         "rays_per_launch": int(rays_per_launch)},
        }
        if valu_insts and last_ms > 0:
            # the fused kernel moves 0.3 B per ray, so the HBM figure above only says that the wavefront formulation's
            # traffic is gone; what bounds it is VALU issue: wave64 instructions per launch (PMC, profiles/) / launch time
            ginst = valu_insts / (last_ms * 1e-3) / 1e9
            out["roofline_valu"] = {"bound": "valu", "achieved": round(ginst, 1), "peak": VALU_PEAK_GINST, "unit": "G wave64-instructions/s",
                                    "frac": round(ginst / VALU_PEAK_GINST, 4), "instructions_per_launch": valu_insts}
        if world == 1 and not dist and args.engine == "fused" and args.cpu_spp > 0:
            out["parity"] = parity_sample(tracer, w, h, total_steps * SPP_PER_STEP)
        if world == 1 and args.cpu_spp > 0:
            out["cpu_baseline"] = cpu_baseline(args.cpu_spp)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    tracer.Destroy()
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

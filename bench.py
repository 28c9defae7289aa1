#!/usr/bin/env python3
"""bench.py -- rays/s of the path-tracing hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

With N > 1 and no WORLD_SIZE in the environment this process starts the N ranks itself (a `torch.distributed.run`
child, before anything here touches a GPU) and passes rank 0's JSON line through; under an external
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...` it is one of the ranks.

Workload (config.workload): BASELINE.json configs[1] -- the reference's default scene (PT_CudaSimple/PathTracer.cu:48-69),
default camera, 1920x1080, depth 10, seed 1337.  One step = one pass of the hot path over one batch: 64 more samples per
pixel folded into the running mean (the reference's 64 BeginDrawing/FinalizeDrawing pairs), 2*N*64 = 265.4 M rays per GPU.
A "ray" is one extend-stage work item; the count comes from device counters.

N > 1: sample split over xGMI, weak scaling (every rank renders 64 samples per step).  The frame's samples are dealt to
8 VIRTUAL shards whatever N is (include/pt_plugin.h PT_SPLIT_SAMPLE; strided RNG, stride 35 = the reference's own stream
on this scene), rank g renders the shards {v : v % N == g} into a running mean each, and ONE exchange inside the timed
region -- after the last step, the shape of BASELINE.json configs[2]: render, then one reduce -- combines them over RCCL:
all_to_all of image slices, the fixed-order tree over the 8 shards on each rank's slice, all_gather
(pathtracer_amd/sample_split.py; --exchange native runs the C++ form, pt_dist_combine).  The image for a given total
sample count is bit-identical on 1, 2, 4 and 8 GPUs (tests/test_sample_split_gloo.py, tests/test_gpu_split.py).
--exchange-every E exchanges after every E-th step instead, beside the next step's render (a progressive display).

The JSON line carries `roofline` -- the bound that binds the dominant kernel: VALU issue, wave64 instructions per launch
(rocprofv3 PMC, profiles/traffic.json, tied to the kernel sources by a hash) over the launch time measured here with HIP
events on the plugin's stream -- `roofline_hbm` (SURVEY 8d's algorithmic bytes over the same time; not binding: the
kernel keeps the path state in registers) and `cpu_baseline` (the CPU oracle, a port of the reference path, timed on this
box's cores).  Outside the timed region it also measures a sustained loop, the wavefront engine, and one figure each for
BASELINE.json's configs[4] (all materials) and configs[3] (10 092 triangles, BVH), each with its own bound.
"""
import argparse
import functools
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))      # tests/scenes.py: the deterministic scenes of BASELINE.json's configs

WIDTH, HEIGHT, SPP_PER_STEP, DEPTH = 1920, 1080, 64, 10
SETTLE_SECONDS = 0.3          # untimed renders before the timed steps, see measure()
PROBE_RENDERS = 12            # the plugin's own measurement: one unmeasured render, then two per candidate (at most five: pt_plugin.cpp)
HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# VALU issue peak: 256 CUs x 4 SIMDs, one wave64 instruction per 2 cycles per SIMD at 2.4 GHz (same guide: v_fma_f32
# 2 cycles on the SIMD-32; tools/valu_microbench.hip measures 2.4-2.6 at the clock the chip holds under load)
VALU_PEAK_GINST = 256 * 4 * 2.4 / 2.0
# SURVEY 8d algorithmic traffic of the wavefront formulation per pixel-sample on the default scene: generate 52 +
# extend 36 + shade-diffuse 104 + extend 36 + shade-emissive 44*p + accumulate 48 = 276 + 44 p bytes for 2 rays;
# p = share of secondary rays that reach the light (measured by the oracle at 1080p: 0.0473).
P_LIGHT = 0.0473
BYTES_PER_RAY = (276.0 + 44.0 * P_LIGHT) / 2.0
KERNEL_SOURCES = ["pathtracer_amd/csrc/pt_kernels.hip", "pathtracer_amd/csrc/pt_device.hpp", "pathtracer_amd/csrc/pt_wavefront.hpp",
                  "pathtracer_amd/csrc/pt_fused_bvh.hpp", "pathtracer_amd/csrc/pt_bvh.hpp"]

# name -> (scene, depth, stride of the strided streams, description)
WORKLOADS = {
    "default": ("default", 10, 35, "reference default scene (BASELINE.json configs[1])"),
    "c5": ("materials", 16, 2 + 33 * 16, "all materials, 48 triangles, depth 16 (BASELINE.json configs[4])"),
    "c4_10k": ("synthetic10000", 10, 2 + 33 * 10, "synthetic 10 092 triangles through the BVH (BASELINE.json configs[3])"),
    "c4_100k": ("synthetic100000", 10, 2 + 33 * 10, "synthetic 99 372 triangles through the BVH (BASELINE.json configs[3], upper end)"),
}


def kernel_sources_sha():
    h = hashlib.sha256()
    for rel in KERNEL_SOURCES:
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def load_counters():
    """profiles/traffic.json: per (workload, engine) the wave64 VALU instructions and HBM bytes of ONE bench step, from
    rocprofv3 --pmc passes (tools/measure_counters.sh), with the hash of the kernel sources they were taken on."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            return json.load(f)
    except Exception:
        return {}


@functools.lru_cache(maxsize=None)
def scene_of(name):
    import scenes
    if name == "default":
        return scenes.default_scene()
    if name == "materials":
        return scenes.all_materials_scene()
    if name.startswith("synthetic"):
        return scenes.synthetic_scene(int(name[len("synthetic"):]))
    raise ValueError(name)


def launch_ranks(n, argv, script=None):
    """--gpus N without a launcher: start the N ranks as a torch.distributed.run child.  Nothing in THIS process has
    touched the GPU (no torch.cuda, no plugin call -- torch is not even imported yet), and it replaces nothing: it waits and
    exits with the child's code.  `script`: what the ranks run (this file; tests/test_bench_launch.py passes a stub)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    # RCCL shares device buffers between the ranks' processes through IPC handles, and this pool's host driver only supports
    # the dmabuf kind: with the legacy mode the first collective fails with `hipIpcGetMemHandle: invalid argument`.  The image
    # exports the variable already; it is set here too so that a cleaned environment (a test runner, a service unit) still works.
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(script or __file__)] + argv
    return subprocess.run(cmd, env=env).returncode


def cpu_baseline(spp):
    """The oracle (kind 'port': our C restatement of the reference path) on the host cores: `spp` samples over all of
    them, and ('also report 1-thread') a twentieth of that on one thread."""
    import ctypes as C
    import numpy as np
    from oracle import pt_oracle_py as po
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


def parity_sample(img, w, h, spp_total, n_pixels=2048):
    """BASELINE.json's third metric, 'RMSE vs reference': a bounded sample of pixels of the image the timed steps
    produced, against the oracle rendering the same pixels and samples (sequential RNG = the reference's stream)."""
    import numpy as np
    from oracle import pt_oracle_py as po
    ids = np.unique(np.random.default_rng(7).integers(0, w * h, n_pixels)).astype(np.uint32)
    ref, _, _, _ = po.render(po.params(w, h, depth=DEPTH, spp=spp_total), pixel_ids=ids)
    got = img[ids]
    equal = int((got.view(np.uint32) == ref.view(np.uint32)).all(axis=1).sum())
    diff = got.astype(np.float64) - ref.astype(np.float64)
    rel = np.abs(diff) / np.maximum(np.abs(ref.astype(np.float64)), 1e-6)
    return {"pixels_checked": int(len(ids)), "pixels_bit_identical": equal, "rmse": float(np.sqrt((diff ** 2).mean())),
            "max_rel_err": float(rel.max()), "spp": int(spp_total), "against": "oracle (CPU restatement of the reference path), same pixels and samples"}


def valu_roofline(counters, key, kernel_ms, sha):
    """The binding bound of these kernels: wave64 VALU instructions per step (PMC, steady-state launches only: tools/steady.py) over
    the device time of the step.  `kernels` prices every kernel of the step by itself: its instructions per launch over its own
    average duration in the kernel trace of the same command (profiles/rNN_*_kernel_stats.csv)."""
    e = (counters.get("entries") or {}).get(key)
    if not e or not e.get("valu_insts") or kernel_ms <= 0:
        return None
    ginst = e["valu_insts"] / (kernel_ms * 1e-3) / 1e9
    kernels = {k: v for k, v in (e.get("kernels") or {}).items() if isinstance(v, dict)}       # (round 2's file kept one number per kernel)
    dom = kernels.get(e.get("kernel")) or {}
    return {"bound": "valu", "achieved": round(ginst, 1), "peak": VALU_PEAK_GINST, "unit": "G wave64-instructions/s",
            "frac": round(ginst / VALU_PEAK_GINST, 4), "traffic": e.get("hbm_bytes"), "kernel": e.get("kernel"),
            "kernel_ms": round(kernel_ms, 4), "instructions_per_step": e["valu_insts"], "launches_per_step": e.get("launches_per_step", 1),
            "instructions_per_launch": dom.get("valu_insts_per_launch"), "kernel_launches_per_step": dom.get("launches_per_step"),
            "kernel_avg_us_in_trace": dom.get("avg_us"), "kernel_frac_in_trace": dom.get("valu_frac"),
            "lanes_per_instruction": e.get("lanes_per_instruction"),
            "kernels": {k: {f: v.get(f) for f in ("launches_per_step", "valu_insts_per_launch", "lanes_per_instruction", "avg_us", "valu_frac")}
                        for k, v in kernels.items() if (v.get("valu_insts_per_launch") or 0) * (v.get("launches_per_step") or 0) >= 0.02 * e["valu_insts"]},
            "ta_busy_pct": e.get("ta_busy_pct"), "gui_cycles_per_ns": e.get("gui_cycles_per_ns"),
            "source": "profiles/traffic.json (rocprofv3 --pmc SQ_INSTS_VALU / FETCH_SIZE x2 / WRITE_SIZE, separate passes; launches after the bench's marker only)",
            "stale": counters.get("sha") != sha}


def measure(pt, workload, engine, w, h, steps, warmup, spp=SPP_PER_STEP, samples_per_pass=16, sequential=False, min_seconds=0.0, device=0, settle_seconds=SETTLE_SECONDS):
    """steps x (one pt_render of `spp` samples); returns wall ms per step, device ms per step, rays per step, the tracer's image"""
    scene_name, depth, stride, _ = WORKLOADS[workload]
    tracer = pt.PathTracer(w, h, device=device)
    if scene_name != "default":
        tracer.set_scene(*pt.make_scene(*scene_of(scene_name)))
    eng = {"fused": pt.PT_ENGINE_FUSED, "wavefront": pt.PT_ENGINE_WAVEFRONT, "auto": pt.PT_ENGINE_AUTO}[engine]
    if sequential:
        tracer.set_options(engine=eng, depth=depth, samples_per_pass=samples_per_pass)
    else:
        tracer.set_options(engine=eng, depth=depth, rng_mode=pt.PT_RNG_STRIDED, stride=stride, samples_per_pass=samples_per_pass)
    # the plugin measures its candidates (fused in step / out of step, wavefront under AUTO) on the first renders after a
    # restart: those come first, untimed and on top of the W warm-up steps, so that the timed steps run the decided kernels
    # ... and the device gets SETTLE_SECONDS of the same renders to reach the clock it holds under this load: the 20 steps the
    # driver times are 20 ms of work, and measured from a cold start they run 3-6 % below what the same step sustains for seconds
    # (`sustained`); settling is untimed work like the W warm-up steps, it skips nothing inside the timed region
    settle = PROBE_RENDERS
    for _ in range(settle + warmup):
        tracer.render(spp)
    t_settle = time.perf_counter()
    while time.perf_counter() - t_settle < settle_seconds:
        tracer.render(spp)
        settle += 1
    if os.environ.get("PT_BENCH_MARKER"):
        # profiling runs (tools/steady.py): one tiny, unmistakable dispatch between the untimed renders and the timed steps
        import ctypes as C
        lib = pt.load_library()
        word = C.c_void_p()
        if lib.pt_hip_malloc(C.byref(word), 4) == 0:
            lib.pt_hip_fill_f32(word, C.c_float(0.0), 1, None)
            lib.pt_hip_stream_synchronize(None)
            lib.pt_hip_free(word)
    st0 = tracer.stats()
    t0 = time.perf_counter()
    done = 0
    while done < steps or time.perf_counter() - t0 < min_seconds:
        tracer.render(spp)
        done += 1
    dt = time.perf_counter() - t0
    st = tracer.stats()
    out = {"steps": done, "untimed_steps": settle + warmup, "ms_per_step": dt / done * 1e3, "fused_out_of_step": int(st.fused_out_of_step),
           "kernel_ms": (st.total_render_ms - st0.total_render_ms) / max(1, st.renders - st0.renders),
           "rays_per_step": (st.rays - st0.rays) / done, "engine_in_use": int(st.engine_in_use),
           "rng": "sequential" if sequential else f"strided/{stride}", "depth": depth}
    return out, tracer


def side_figure(pt, counters, sha, workload, engine, w, h, steps=3, warmup=2):
    """one figure of another BASELINE.json config, with its own bound (outside the timed region of the headline)"""
    m, tracer = measure(pt, workload, engine, w, h, steps, warmup)
    tracer.Destroy()
    fig = {"workload": WORKLOADS[workload][3] + f", {w}x{h}, {SPP_PER_STEP} spp per step", "engine": engine, "rng": m["rng"], "depth": m["depth"],
           "value": round(m["rays_per_step"] / (m["kernel_ms"] * 1e-3) / 1e6, 1), "unit": "Mrays/s (device time)",
           "ms_per_step": round(m["ms_per_step"], 3), "kernel_ms": round(m["kernel_ms"], 3), "rays_per_step": int(m["rays_per_step"]),
           "rays_per_pixel_sample": round(m["rays_per_step"] / (w * h * SPP_PER_STEP), 3), "fused_out_of_step": m["fused_out_of_step"]}
    fig["roofline"] = valu_roofline(counters, f"{workload}/{engine}", m["kernel_ms"], sha)
    return fig


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--engine", choices=["fused", "wavefront", "auto"], default="fused")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="default", help="N = 1 only: which BASELINE.json config the timed steps run (the headline is 'default')")
    ap.add_argument("--samples-per-pass", type=int, default=16, help="wavefront strided batching")
    ap.add_argument("--cpu-spp", type=int, default=48, help="samples of the CPU baseline leg (0 = skip); 48 spp at 1080p = 199 M rays, about 16 CPU-seconds over the box's 16 threads")
    ap.add_argument("--bare", action="store_true", help="timed steps only: no CPU baseline, parity sample or side figures (profiling runs)")
    ap.add_argument("--force-dist", action="store_true", help="run the sample split and its RCCL exchange even with one rank (self-test of the N>1 path)")
    ap.add_argument("--exchange", choices=["torch", "native"], default="torch", help="N>1: torch.distributed collectives (RCCL) or the plugin's own pt_dist_combine (RCCL)")
    ap.add_argument("--exchange-every", type=int, default=0, help="N>1: exchange after every E-th step, beside the next render; 0 = once, after the last step (inside the timed region)")
    ap.add_argument("--dump-image", default=None, help="N>1 path: rank 0 writes the combined image (float32 [pixels, 4], .npy) here after the timed region (tests)")
    ap.add_argument("--settle-seconds", type=float, default=SETTLE_SECONDS, help="untimed renders before the timed steps until the device holds its clock (tests pass 0: they count samples)")
    ap.add_argument("--width", type=int, default=WIDTH)
    ap.add_argument("--height", type=int, default=HEIGHT)
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))

    # stdout carries the ONE JSON line and nothing else: what the library prints there (the reference's console messages,
    # IPathTracerManagement.h / PathTracer.cu) goes to stderr for the length of the run
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    def emit(line):
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(line) + "\n").encode())

    import numpy as np
    import torch
    import pathtracer_amd as pt

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}, or without a launcher")
    torch.cuda.set_device(local_rank)
    w, h = args.width, args.height
    n_pix = w * h
    counters, sha = load_counters(), kernel_sources_sha()
    lib = pt.load_library()

    if world == 1 and not args.force_dist:
        # ------------------------------------------------------------------ one GPU
        sequential = args.engine == "fused" and args.workload == "default"       # the reference's own streams
        m, tracer = measure(pt, args.workload, args.engine, w, h, args.steps, args.warmup, samples_per_pass=args.samples_per_pass,
                            sequential=sequential, device=local_rank, settle_seconds=args.settle_seconds)
        torch.cuda.synchronize()
        value = m["rays_per_step"] / (m["ms_per_step"] * 1e-3) / 1e6
        hbm_achieved = m["rays_per_step"] * BYTES_PER_RAY / (m["kernel_ms"] * 1e-3) / 1e9
        key = f"{args.workload}/{args.engine}"
        roof = valu_roofline(counters, key, m["kernel_ms"], sha)
        traffic = roof["traffic"] if roof else None
        out = {
            "metric": "Mrays/s (primary+bounce)", "value": round(value, 1), "unit": "Mrays/s",
            "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(m["ms_per_step"], 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "spp_per_s": round(SPP_PER_STEP / (m["ms_per_step"] * 1e-3), 1),
            "config": {"workload": f"{WORKLOADS[args.workload][3]}, {w}x{h}, {SPP_PER_STEP} spp per step per GPU, depth {m['depth']}, seed 1337",
                       "engine": args.engine, "rng": m["rng"], "parallelism": "single GPU"},
            "roofline": roof,
        }
        if args.workload == "default":
            out["roofline_hbm"] = {"bound": "hbm", "binding": False, "achieved": round(hbm_achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": round(hbm_achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                                   "measured_frac": round(traffic / (m["kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if traffic else None,
                                   "algorithmic_bytes_per_ray": round(BYTES_PER_RAY, 2), "rays_per_launch": int(m["rays_per_step"]),
                                   "note": "SURVEY 8d's byte model prices the wavefront formulation's queue traffic; these kernels keep the path state in registers, "
                                           "so achieved/peak > 1 only says that traffic is gone (measured_frac = PMC bytes / time / peak)"}
        if not args.bare:
            if args.workload == "default" and sequential:
                out["parity"] = parity_sample(tracer.readback(), w, h, (m["untimed_steps"] + m["steps"]) * SPP_PER_STEP)
            tracer.Destroy()
            try:
                with open(os.path.join(ROOT, "profiles", "tolerance.json")) as f:
                    out["tolerance"] = json.load(f)       # tools/fma_tolerance.py: the oracle built with FMA contraction against the strict one
                out["tolerance"]["measured_in_this_run"] = False          # a committed file (profiles/tolerance.json), read, not measured here
            except Exception:
                pass
            # outside the timed region: a sustained loop of the same step, the other engine, the other configs
            ms, t2 = measure(pt, args.workload, args.engine, w, h, 1, 1, samples_per_pass=args.samples_per_pass, sequential=sequential, min_seconds=2.5, device=local_rank)
            t2.Destroy()
            out["sustained_ms_per_step"] = round(ms["ms_per_step"], 4)
            out["sustained"] = {"steps": ms["steps"], "seconds": round(ms["ms_per_step"] * ms["steps"] / 1e3, 2), "kernel_ms": round(ms["kernel_ms"], 4),
                                "value": round(ms["rays_per_step"] / (ms["ms_per_step"] * 1e-3) / 1e6, 1)}
            figures = []
            other = "wavefront" if args.engine != "wavefront" else "fused"
            figures.append(side_figure(pt, counters, sha, args.workload, other, w, h, steps=10, warmup=3))
            ma, ta = measure(pt, args.workload, "auto", w, h, 3, 3, samples_per_pass=args.samples_per_pass, sequential=sequential, device=local_rank)
            ta.Destroy()
            out["engine_auto_pick"] = {pt.PT_ENGINE_FUSED: "fused", pt.PT_ENGINE_WAVEFRONT: "wavefront"}.get(ma["engine_in_use"], "undecided")
            if args.workload == "default":
                for wl, eng in (("c5", "wavefront"), ("c5", "fused"), ("c4_10k", "wavefront"), ("c4_10k", "fused"), ("c4_100k", "wavefront"), ("c4_100k", "fused")):
                    figures.append(side_figure(pt, counters, sha, wl, eng, w, h))
            out["other_figures"] = figures
            out["cpu_baseline"] = cpu_baseline(args.cpu_spp) if args.cpu_spp > 0 else None
        else:
            tracer.Destroy()
            out["cpu_baseline"] = None
        emit(out)
        return

    # ---------------------------------------------------------------------- sample split over `world` ranks
    import torch.distributed as dist
    from pathtracer_amd import sample_split as ss
    if "MASTER_ADDR" not in os.environ:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
    if ss.VIRTUAL_SHARDS % world:
        raise SystemExit(f"sample split deals the frame to {ss.VIRTUAL_SHARDS} virtual shards: --gpus must divide it (1, 2, 4, 8)")
    # the exchange runs BESIDE a render when --exchange-every is used: collectives and combine on high-priority streams
    os.environ.setdefault("TORCH_NCCL_HIGH_PRIORITY", "1")
    pg_options = None
    try:
        pg_options = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
    except Exception:
        pg_options = None
    dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), pg_options=pg_options)
    ones = torch.ones(1, device="cuda")
    dist.all_reduce(ones)                                    # RCCL is up and saw every rank
    rccl_ranks = int(ones.item())

    n_local = ss.VIRTUAL_SHARDS // world
    per_shard = SPP_PER_STEP // n_local                       # samples each local shard gets per step: 64 per rank and step in all
    assert per_shard * n_local == SPP_PER_STEP
    tracer = pt.PathTracer(w, h, device=local_rank)
    tracer.set_options(engine=pt.PT_ENGINE_FUSED if args.engine != "wavefront" else pt.PT_ENGINE_WAVEFRONT, depth=DEPTH,
                       rng_mode=pt.PT_RNG_STRIDED, stride=35, samples_per_pass=args.samples_per_pass, samples_per_begin=per_shard,
                       split_mode=pt.PT_SPLIT_SAMPLE, split_rank=rank, split_ranks=world)
    local, pitch, vshards = tracer.shard_layout()
    assert (local, vshards) == (n_local, ss.VIRTUAL_SHARDS) and pitch == ss.slice_pixels(n_pix, world) * world
    slice_px = pitch // world
    native = args.exchange == "native"
    if native:
        ids = [pt.dist_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ids, src=0)                # the id travels by the host's own means: here torch's store
        tracer.dist_init(ids[0])
    else:
        snapshot = torch.empty(local * pitch * 4, device="cuda", dtype=torch.float32)     # this rank's shard means
        stacked = torch.empty(vshards * slice_px * 4, device="cuda", dtype=torch.float32)
        image = torch.empty(pitch * 4, device="cuda", dtype=torch.float32)
        plugin_stream = torch.cuda.ExternalStream(tracer.stream())
        exchange_stream = torch.cuda.Stream(priority=-1)
        snapshot_ready, exchanged = torch.cuda.Event(), torch.cuda.Event()

        def combine_slices(stk, shards, n_pix_slice, out_slice):
            rc = lib.pt_hip_combine_shards(stk.data_ptr(), shards, n_pix_slice, out_slice.data_ptr(), torch.cuda.current_stream().cuda_stream)
            assert rc == 0, lib.pt_hip_last_error()

    exchanges = 0

    def exchange():
        """enqueue the one exchange for the shard means as they are now; nothing is awaited (torch) / returns done (native)"""
        nonlocal exchanges
        exchanges += 1
        if native:
            tracer.dist_combine()
            return
        plugin_stream.wait_event(exchanged)                       # the previous exchange has read the snapshot
        tracer.copy_mean_async(snapshot.data_ptr(), snapshot.numel())
        snapshot_ready.record(plugin_stream)
        exchange_stream.wait_event(snapshot_ready)
        with torch.cuda.stream(exchange_stream):
            ss.exchange_and_combine(snapshot, dist, combine_slices, stacked, image)
        exchanged.record(exchange_stream)

    def step(k):
        tracer.BeginDrawing()                                     # this rank's 64 samples of the step, enqueued
        tracer.FinalizeDrawing()
        if args.exchange_every and (k + 1) % args.exchange_every == 0:
            exchange()

    def sync():
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()

    if not native:
        exchanged.record(exchange_stream)
    for k in range(PROBE_RENDERS + args.warmup):                  # the plugin measures its candidates on the first renders (untimed)
        step(k)
    # (measure(): SETTLE_SECONDS of the same steps for the clock the device holds under this load; the same COUNT on every rank --
    # a step may hold a collective)
    t_settle = time.perf_counter()
    if args.settle_seconds > 0:
        step(0)
    torch.cuda.synchronize()
    n_settle = torch.tensor([min(2000, int(args.settle_seconds / max(time.perf_counter() - t_settle, 1e-4)))], dtype=torch.int64, device="cuda")
    dist.all_reduce(n_settle, op=dist.ReduceOp.MAX)
    for _ in range(int(n_settle.item())):
        step(0)
    exchange()                                                    # warm the collectives up too
    sync()
    st0 = tracer.stats()
    exchanges = 0
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k)
    t_rendered = time.perf_counter() - t0                         # FinalizeDrawing has waited for this rank's last step
    if not args.exchange_every or args.steps % args.exchange_every:
        exchange()                                                # the render's one reduce (configs[2]'s shape)
    sync()
    dt = time.perf_counter() - t0
    st = tracer.stats()
    rays_rank = st.rays - st0.rays
    kernel_ms = (st.total_render_ms - st0.total_render_ms) / max(1, st.renders - st0.renders)
    # per-phase figures for whoever reads a SCALE run: this rank's render wall time, and what came after it (the exchange, then
    # the wait for the slowest rank at the barrier) -- the MAX over ranks of each, and the MIN of the render time beside it
    t = torch.tensor([dt, float(rays_rank), t_rendered, dt - t_rendered, -t_rendered], device="cuda", dtype=torch.float64)
    tmax = t.clone()
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    dt, rays_total = float(tmax[0]), float(t[1])
    render_ms_max, exchange_ms, render_ms_min = float(tmax[2]) * 1e3, float(tmax[3]) * 1e3, -float(tmax[4]) * 1e3
    # every rank holds the same image: a checksum of checksums over the ranks
    img = tracer.readback() if native else image[:n_pix * 4].cpu().numpy().reshape(n_pix, 4)
    digest = int(np.bitwise_xor.reduce(img.view(np.uint32).reshape(-1)))
    dg = torch.tensor([digest, -digest], device="cuda", dtype=torch.int64)
    dist.all_reduce(dg, op=dist.ReduceOp.MAX)
    same_image = int(dg[0]) == -int(dg[1])

    if rank == 0:
        roof = valu_roofline(counters, f"default/{'wavefront' if args.engine == 'wavefront' else 'fused'}_strided", kernel_ms, sha) or \
               valu_roofline(counters, f"default/{'wavefront' if args.engine == 'wavefront' else 'fused'}", kernel_ms, sha)
        out = {
            "metric": "Mrays/s (primary+bounce)", "value": round(rays_total / dt / 1e6, 1), "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "spp_per_s": round(SPP_PER_STEP * world * args.steps / dt, 1),
            "config": {"workload": f"{WORKLOADS['default'][3]}, {w}x{h}, {SPP_PER_STEP} spp per step per GPU, depth {DEPTH}, seed 1337",
                       "engine": "wavefront" if args.engine == "wavefront" else "fused", "rng": "strided/35",
                       "parallelism": f"sample split x{world}: {ss.VIRTUAL_SHARDS} virtual shards, {n_local} per rank, {per_shard} spp per shard and step; "
                                      f"{exchanges} exchange(s) in the timed region ({'pt_dist_combine' if native else 'torch.distributed'} over RCCL: all_to_all of slices, fixed-order tree, all_gather)"},
            "rccl_ranks": rccl_ranks, "exchanges_timed": exchanges, "image_identical_on_all_ranks": same_image,
            "kernel_ms_per_step": round(kernel_ms, 4),
            "render_ms_max": round(render_ms_max, 4), "render_ms_min": round(render_ms_min, 4), "exchange_ms": round(exchange_ms, 4),
            "phases": "render_ms_max / render_ms_min: wall time of a rank's timed steps (max / min over ranks); exchange_ms: the rest of the timed "
                      "region on the rank where it was longest (the exchange, then the barrier's wait for the slowest rank)",
            "roofline": roof, "cpu_baseline": None,
        }
        if args.dump_image:
            np.save(args.dump_image, np.ascontiguousarray(img[:n_pix]))
        emit(out)
    tracer.Destroy()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

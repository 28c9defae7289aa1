"""Host-side timing of one sample-split step (BeginDrawing / exchange / snapshot copy / FinalizeDrawing) on one GPU:\n    python tools/dist_probe.py [full|copy|render]   (PROBE_HP=1: exchange on a high-priority stream)"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, torch.distributed as dist
import pathtracer_amd as pt
from pathtracer_amd import sample_split
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29544", RANK="0", WORLD_SIZE="1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
w, h = 1920, 1080
n = w * h
tr = pt.PathTracer(w, h, device=0)
tr.set_options(engine=pt.PT_ENGINE_FUSED, rng_mode=pt.PT_RNG_STRIDED, stride=35, depth=10, samples_per_begin=64)
lib = pt.load_library()
snap = torch.empty(n * 4, device="cuda"); gathered = torch.empty(n * 4, device="cuda"); image = torch.empty(n * 4, device="cuda"); my = torch.empty(n * 4, device="cuda")
ps = torch.cuda.ExternalStream(tr.stream()); ev = torch.cuda.Event()
hp = torch.cuda.Stream(priority=-1) if os.environ.get("PROBE_HP") else torch.cuda.current_stream()
torch.cuda.set_stream(hp)
def comb(stacked, shards, npx, out):
    assert lib.pt_hip_combine_shards(stacked.data_ptr(), shards, npx, out.data_ptr(), torch.cuda.current_stream().cuda_stream) == 0
tr.copy_mean(snap.data_ptr(), n * 4)
mode = sys.argv[1] if len(sys.argv) > 1 else "full"
acc = [0.0] * 6
for it in range(25):
    t = [time.perf_counter()]
    tr.BeginDrawing(); t.append(time.perf_counter())
    if mode == "full":
        sample_split.exchange_and_combine(snap, dist, comb, out=image, scratch=gathered, slice_out=my)
    t.append(time.perf_counter())
    if mode != "render":
        ev.record(torch.cuda.current_stream()); ps.wait_event(ev)
        tr.copy_mean_async(snap.data_ptr(), n * 4)
    t.append(time.perf_counter())
    tr.FinalizeDrawing(); t.append(time.perf_counter())
    if it >= 5:
        for k in range(4): acc[k] += t[k + 1] - t[k]
        acc[4] += t[4] - t[0]
print(mode, "per step us: begin %.1f exchange-enqueue %.1f copy-enqueue %.1f finalize %.1f total %.1f" % tuple(a / 20 * 1e6 for a in acc[:5]), "kernel ms", tr.stats().last_render_ms)
tr.Destroy(); dist.destroy_process_group()

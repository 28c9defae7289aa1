import os, sys
sys.path.insert(0, os.getcwd())
import pathtracer_amd as pt
for ranks, rank in ((1, 0), (2, 0), (4, 0), (8, 0)):
    for engine, name in ((pt.PT_ENGINE_FUSED, "fused"), (pt.PT_ENGINE_WAVEFRONT, "wavefront")):
        with pt.PathTracer(1920, 1080) as tr:
            tr.set_options(engine=engine, rng_mode=pt.PT_RNG_STRIDED, stride=35, split_mode=pt.PT_SPLIT_SAMPLE, split_rank=rank, split_ranks=ranks, samples_per_pass=16)
            per = 64 * ranks // 8
            for _ in range(3): tr.render(per)
            s0 = tr.stats()
            for _ in range(10): tr.render(per)
            s1 = tr.stats()
            print(f"split_ranks {ranks} {name:9s}: {8 // ranks} local shards x {per} spp per step: {(s1.total_render_ms - s0.total_render_ms) / 10:.3f} ms per step, rays/step {(s1.rays - s0.rays) // 10}", flush=True)

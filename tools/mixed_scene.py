"""A room of large quads with a finely tessellated ball in it (tests/test_bvh_adversarial.py room_with_a_ball): classes for the walls,
free triangles (ClassBlock::kappa_free) for the facets; Mrays/s (device time) at 1080p, 16 samples per call."""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import pathtracer_amd as pt
from test_bvh_adversarial import room_with_a_ball
for n_room, n_ball in ((300, 1200), (192, 5000), (972, 20000)):
    V, T, M = room_with_a_ball(n_room, n_ball, 8)
    sc = pt.make_scene(V, T, M)
    for engine, name in ((pt.PT_ENGINE_FUSED, "fused"), (pt.PT_ENGINE_WAVEFRONT, "wavefront")):
        with pt.PathTracer(1920, 1080) as tr:
            tr.set_scene(*sc)
            kw = dict(rng_mode=pt.PT_RNG_STRIDED, stride=332, samples_per_pass=16) if name == "wavefront" else {}
            tr.set_options(engine=engine, **kw)
            for _ in range(3): tr.render(16)
            s0 = tr.stats(); tr.render(16); s1 = tr.stats()
            print(f"room {n_room} + ball {len(T) - n_room} triangles {name:9s}: {(s1.rays - s0.rays) / (s1.total_render_ms - s0.total_render_ms) / 1e3:8.1f} Mrays/s", flush=True)

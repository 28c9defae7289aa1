"""The wavefront engine's persistent late-bounce kernel (bounce_tail_block in pathtracer_amd/csrc/pt_wavefront.hpp, what
k_bounce_tail runs) on the CPU: a block of 4 waves of host threads (tests/host/bounce_tail_check.cpp, ASan + UBSan) takes a
queue of diverging rays at bounce `entry` and carries every path to its end in a lane; the lit flags, radiance, stream words
and rays per bounce must equal the same paths run one at a time by the staged stages' own loop (bvh_intersect + shade_lane)."""
import os
import subprocess

import numpy as np
import pytest

from test_bvh_host import make_rays, pack, scene_by_name

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHECK = os.path.join(ROOT, "tests", "host", "bounce_tail_check")
needs_binary = pytest.mark.skipif(not os.path.exists(CHECK), reason="tests/host/bounce_tail_check is not built (__graft_entry__.build())")


def run(pt, oracle, tmp_path, scene, n_rays, entry, depth, rng_mode=1, shading=0, lanes=64, seed=5, reflectivity=None):
    V, T, M = scene
    if reflectivity is not None:
        M = [(m[0], m[1], reflectivity) for m in M]
    camera = (0.2, -0.1, 0.05)
    c = oracle.camera_corners(oracle.default_camera(), np.float32(1.5))
    f4 = lambda v: pt.Float4(v.x, v.y, v.z, v.w)
    fr = pt.Frame(width=64, height=64, pixel_first=0, pixel_count=n_rays, depth=depth, seed=1337, rng_mode=rng_mode, stride=2 + 33 * depth,
                  shading=shading, flags=pt.PT_FRAME_SCENE_BVH, origin=f4(c[0]), top_left=f4(c[1]), bottom_left=f4(c[2]), bottom_right=f4(c[3]))
    blob, frame, rays = tmp_path / "scene.blob", tmp_path / "frame.bin", tmp_path / "rays.f32"
    blob.write_bytes(pack(pt, V, T, M))
    frame.write_bytes(bytes(fr))
    make_rays(n_rays, seed, camera).tofile(rays)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0")
    env.pop("LD_PRELOAD", None)
    r = subprocess.run([CHECK, str(blob), str(frame), str(rays), str(entry), str(lanes)], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    fields = r.stdout.split()
    assert int(fields[1]) == n_rays and int(fields[7]) == 0, r.stdout
    return int(fields[3]), int(fields[5])          # paths that reached a light, rays of the bounces after `entry`


@needs_binary
@pytest.mark.parametrize("name,n_rays,entry,depth,lanes", [("synthetic-1500", 1500, 2, 10, 64), ("mixed", 1300, 2, 16, 64), ("synthetic-700", 333, 1, 6, 37),
                                                           ("soup-300", 1100, 3, 8, 64)])
def test_late_bounces_in_lanes_equal_the_staged_loop(pt, oracle, tmp_path, name, n_rays, entry, depth, lanes):
    """reference shading, strided streams (no stream words written): some paths reach a light, some bounce on"""
    lit, later = run(pt, oracle, tmp_path, scene_by_name(name), n_rays, entry, depth, lanes=lanes)
    assert later > 0


@needs_binary
def test_sequential_streams_are_written_back(pt, oracle, tmp_path):
    lit, later = run(pt, oracle, tmp_path, scene_by_name("synthetic-1500"), 900, 2, 10, rng_mode=0)
    assert later > 0


@needs_binary
def test_extended_shading_runs_paths_to_the_depth_limit(pt, oracle, tmp_path):
    """every hit that is no light goes on: long paths, the last bounce's shade (no ray after it) included"""
    lit, later = run(pt, oracle, tmp_path, scene_by_name("synthetic-700"), 500, 2, 12, shading=1, reflectivity=0.8)
    assert later > 2 * 500


@needs_binary
def test_entry_at_the_last_bounce(pt, oracle, tmp_path):
    """entry = depth - 1: every path ends at its first shade, the hand-out is all there is"""
    lit, later = run(pt, oracle, tmp_path, scene_by_name("synthetic-700"), 700, 5, 6)

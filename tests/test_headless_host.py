"""pathtracer_amd/pt_headless: the reference host's frame loop without the window
(PathTracer/Program.cpp:51-91), driving a plugin loaded by name through IPathTracerManagement.h.
CPU tests drive the CPU-oracle plugin (no GPU involved); the GPU tests drive the HIP plugin and
compare its PFM output with the oracle bit for bit."""
import os
import re
import subprocess

import numpy as np
import pytest

import scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "pathtracer_amd", "pt_headless")
CPU_PLUGIN = os.path.join(ROOT, "oracle", "libPT_CpuOracle.so")
HIP_PLUGIN = os.path.join(ROOT, "pathtracer_amd", "libPT_HipWavefront.so")


def run_host(*args, env=None):
    return subprocess.run([HOST, *map(str, args)], capture_output=True, text=True, timeout=300, env=env)


def read_pfm(path):
    raw = open(path, "rb").read()
    m = re.match(rb"PF\n(\d+) (\d+)\n(-?[0-9.]+)\n", raw)
    assert m and float(m.group(3)) < 0, "little-endian colour PFM expected"
    w, h = int(m.group(1)), int(m.group(2))
    return np.frombuffer(raw[m.end():], dtype="<f4").reshape(w * h, 3), w, h


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def test_frame_loop_against_the_cpu_plugin(oracle, tmp_path):
    out = tmp_path / "cpu.pfm"
    r = run_host("--plugin", CPU_PLUGIN, "--size", "64x48", "--frames", 3, "--report-every", 1, "--out", out)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("ms / frame") == 4 and "wrote" in r.stdout         # 3 title-bar lines + the summary
    img, w, h = read_pfm(out)
    assert (w, h) == (64, 48)
    ref, _, _, _ = oracle.render(oracle.params(64, 48, spp=3))
    assert np.array_equal(bits(img), bits(ref[:, :3]))                      # row 0 = bottom row on both sides


def test_camera_move_and_resize_follow_the_reference_protocol(oracle, tmp_path):
    out = tmp_path / "move.pfm"
    r = run_host("--plugin", CPU_PLUGIN, "--size", "48x32", "--frames", 6, "--move-at", 1, "--resize-at", 3, "40x20", "--out", out)
    assert r.returncode == 0, r.stdout + r.stderr
    img, w, h = read_pfm(out)
    assert (w, h) == (40, 20)
    # fov 85 from frame 1 on; Resize before frame 3 restarts: frames 3,4,5 = 3 samples
    ref, _, _, _ = oracle.render(oracle.params(40, 20, cam=oracle.make_camera(fov=85.0), spp=3))
    assert np.array_equal(bits(img), bits(ref[:, :3]))


def test_unknown_plugin_and_bad_arguments():
    r = run_host("--plugin", "PT_DoesNotExist", "--frames", 1)
    assert r.returncode == 3 and "Unable to load library." in r.stdout
    assert run_host("--size", "0x10").returncode == 2
    assert run_host("--bogus").returncode == 2


@pytest.mark.gpu
@pytest.mark.parametrize("engine", ["fused", "wavefront"])
def test_hip_plugin_by_bare_name(oracle, tmp_path, engine):
    """Program.cpp:223,229 addresses the plugin by bare name; here 'PT_HipWavefront' on the loader path."""
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "pathtracer_amd") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    out = tmp_path / "gpu.pfm"
    r = run_host("--size", "160x120", "--frames", 4, "--engine", engine, "--out", out, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Mrays/s" in r.stdout and f"{2 * 160 * 120 * 4} rays" in r.stdout
    img, w, h = read_pfm(out)
    ref, _, _, _ = oracle.render(oracle.params(160, 120, spp=4))
    assert np.array_equal(bits(img), bits(ref[:, :3]))


@pytest.mark.gpu
def test_hip_plugin_scene_file_depth_and_batched_frames(pt, oracle, tmp_path):
    V, T, M = scenes.all_materials_scene()
    path = str(tmp_path / "allmat.pts")
    pt.write_scene_file(path, *pt.make_scene(V, T, M))
    out = tmp_path / "allmat.pfm"
    r = run_host("--plugin", HIP_PLUGIN, "--size", "96x64", "--frames", 2, "--spp-per-frame", 3, "--depth", 16, "--scene", path, "--out", out)
    assert r.returncode == 0, r.stdout + r.stderr
    img, _, _ = read_pfm(out)
    ref, _, _, _ = oracle.render(oracle.params(96, 64, depth=16, spp=6), scene=oracle.make_scene(V, T, M))
    assert np.array_equal(bits(img), bits(ref[:, :3]))
    # a scene file that does not parse stops the host with the plugin's message
    bad = tmp_path / "bad.pts"
    bad.write_text("v 1 2\n")
    r = run_host("--plugin", HIP_PLUGIN, "--scene", bad, "--frames", 1)
    assert r.returncode == 5 and "vertex needs x y z" in r.stderr


@pytest.mark.gpu
def test_hip_plugin_checkpoint_and_resume_across_processes(oracle, tmp_path):
    """--checkpoint after 3 frames, --resume in a NEW process for 5 more: the image is the 8-frame one bit for bit
    (pt_save_state / pt_load_state / pt_get_camera; the host's camera follows the file, so its first Update does not restart)."""
    state, out = tmp_path / "state.bin", tmp_path / "resumed.pfm"
    common = ("--plugin", HIP_PLUGIN, "--size", "120x80")
    r = run_host(*common, "--frames", 3, "--fov", 70, "--pos", "0.05,0.0,0.1", "--depth", 7, "--checkpoint", state)
    assert r.returncode == 0 and f"wrote {state}" in r.stdout, r.stdout + r.stderr
    r = run_host(*common, "--frames", 5, "--resume", state, "--out", out)
    assert r.returncode == 0 and "8 samples per pixel accumulated" in r.stdout, r.stdout + r.stderr
    img, _, _ = read_pfm(out)
    cam = oracle.make_camera(position=(0.05, 0.0, 0.1, 1.0), fov=70.0)
    ref, _, _, _ = oracle.render(oracle.params(120, 80, cam=cam, depth=7, spp=8))
    assert np.array_equal(bits(img), bits(ref[:, :3]))
    # the CPU-oracle plugin has no state files: the host says so instead of crashing
    r = run_host("--plugin", CPU_PLUGIN, "--frames", 1, "--checkpoint", tmp_path / "x.bin")
    assert r.returncode == 5 and "not exported" in r.stderr


def test_periodic_snapshots_are_the_running_mean_so_far(oracle, tmp_path):
    """--snapshot-every N: the periodic readback of a progressive display (Program.cpp draws the texture every frame); snapshot k
    is the k-sample image, the final one the full image"""
    out = tmp_path / "prog.pfm"
    r = run_host("--plugin", CPU_PLUGIN, "--size", "48x32", "--frames", 5, "--snapshot-every", 2, "--out", out)
    assert r.returncode == 0, r.stdout + r.stderr
    for k in (2, 4):
        img, w, h = read_pfm(f"{out}.{k}.pfm")
        ref, _, _, _ = oracle.render(oracle.params(48, 32, spp=k))
        assert (w, h) == (48, 32) and np.array_equal(bits(img), bits(ref[:, :3])), k
    img, _, _ = read_pfm(out)
    ref, _, _, _ = oracle.render(oracle.params(48, 32, spp=5))
    assert np.array_equal(bits(img), bits(ref[:, :3]))

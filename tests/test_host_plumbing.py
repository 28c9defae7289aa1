"""BASELINE.json configs[0] (plumbing): the reference host's call sequence, through the
POSIX twin of IPathTracerManagement.h, against the CPU-oracle plugin -- no GPU involved.
tests/host/host_driver.cpp is the headless stand-in for PathTracer/Program.cpp:51-91."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, "tests", "host", "host_driver")
CPU_PLUGIN = os.path.join(ROOT, "oracle", "libPT_CpuOracle.so")


@pytest.fixture(scope="module", autouse=True)
def built():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "host")], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)


def run_driver(library, w, h, frames, out, *extra):
    return subprocess.run([DRIVER, library, str(w), str(h), str(frames), str(out), *map(str, extra)],
                          capture_output=True, text=True, timeout=300)


def test_c1_default_scene_640x480_1spp_through_iplugin(oracle, tmp_path):
    out = tmp_path / "c1.raw"
    r = run_driver(CPU_PLUGIN, 640, 480, 1, out)
    assert r.returncode == 0, r.stdout + r.stderr
    img = np.fromfile(out, dtype=np.float32).reshape(-1, 4)
    ref, _, st, _ = oracle.render(oracle.params(640, 480, spp=1))
    assert st.rays == 2 * 640 * 480
    assert np.array_equal(img.view(np.uint32), ref.view(np.uint32))


def test_camera_change_restarts_on_the_next_begin(oracle, tmp_path):
    """Update() comes AFTER BeginDrawing (Program.cpp:57-58): a change made before frame f is seen by
    Update in frame f and restarts accumulation at frame f+1 (PathTracer.cu:99-108)."""
    out = tmp_path / "move.raw"
    w, h, frames, move_at = 48, 32, 5, 2
    r = run_driver(CPU_PLUGIN, w, h, frames, out, "--move-at", move_at)
    assert r.returncode == 0, r.stdout + r.stderr
    img = np.fromfile(out, dtype=np.float32).reshape(-1, 4)
    cam = oracle.make_camera(fov=85.0)
    ref, _, _, _ = oracle.render(oracle.params(w, h, cam=cam, spp=frames - move_at - 1))
    assert np.array_equal(img.view(np.uint32), ref.view(np.uint32))


def test_resize_reallocates_and_restarts(oracle, tmp_path):
    out = tmp_path / "resize.raw"
    r = run_driver(CPU_PLUGIN, 32, 24, 4, out, "--resize-at", 1, 40, 20)
    assert r.returncode == 0, r.stdout + r.stderr
    img = np.fromfile(out, dtype=np.float32).reshape(-1, 4)
    assert img.shape[0] == 40 * 20
    ref, _, _, _ = oracle.render(oracle.params(40, 20, spp=3))
    assert np.array_equal(img.view(np.uint32), ref.view(np.uint32))


def test_loader_reports_failures_like_the_reference(tmp_path):
    r = run_driver("PT_DoesNotExist", 8, 8, 1, tmp_path / "x.raw")
    assert r.returncode == 3
    assert "Unable to load library." in r.stdout and "library 'PT_DoesNotExist'" in r.stdout
    # a library without Create/Destroy
    r = run_driver(os.path.join(ROOT, "oracle", "libpt_oracle.so"), 8, 8, 1, tmp_path / "x.raw")
    assert r.returncode == 3 and "Unable to get Create and Destroy function addresses." in r.stdout


def test_bare_library_name_is_decorated(oracle, tmp_path):
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "oracle"))
    out = tmp_path / "bare.raw"
    r = subprocess.run([DRIVER, "PT_CpuOracle", "16", "12", "1", str(out)], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr

"""Compiles and runs the exhaustive C proof the device code relies on."""
import os
import subprocess


def test_diffuse_sampler_never_exits_early(tmp_path):
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "proofs", "sampler_never_exits_early.c")
    exe = str(tmp_path / "scan")
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-fopenmp", src, "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout
    assert "early exits 0" in out.stdout and "filter passes 2097153" in out.stdout

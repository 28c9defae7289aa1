"""Exhaustive GPU proofs the device code relies on (tests/proofs/exact_rcp_sqrt.hip): the short reciprocal /
square-root sequences of pathtracer_amd/csrc/pt_device.hpp return, for every one of the 2^32 float32 bit patterns
in their stated ranges, exactly what the compiler's IEEE `1.0f / x` and `sqrtf(x)` return."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "proofs", "_build", "exact_rcp_sqrt")


@pytest.mark.gpu
def test_short_rcp_and_sqrt_are_the_ieee_results_for_every_float():
    assert os.path.exists(EXE), "tests/proofs/_build/exact_rcp_sqrt is built by __graft_entry__.build() (make -C tests/proofs)"
    out = subprocess.run([EXE], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    for name in ("rcp_newton", "sqrt_fixup", "rsqrt_ref_bounded", "rcp_ieee", "sqrt_ieee", "rsqrt_ref"):
        assert f"{name} mismatches 0\n" in out.stdout, out.stdout

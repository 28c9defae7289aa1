"""The floating-point tolerance the north star asks for: the oracle's own source built with FMA
contraction (what nvcc -fmad=true does to the reference) stays within the stated tolerance of the strict build.
GPU vs oracle needs no tolerance (bit-exact everywhere else in this suite)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_committed_tolerance_figures_hold_the_stated_tolerance():
    with open(os.path.join(ROOT, "profiles", "tolerance.json")) as f:
        t = json.load(f)
    assert t["fma_instructions_in_variant"] > 0
    for row in t["workloads"]:
        assert row["rmse"] <= t["stated_tolerance"]["rmse"]
        assert row["pixels_with_a_flipped_decision"] <= t["stated_tolerance"]["pixels_with_a_flipped_decision_share"] * row["pixels"]


def test_fma_build_of_the_oracle_on_a_small_frame():
    import fma_tolerance as ft
    from oracle import pt_oracle_py as po
    po.build()
    ft.FMA_LIB, n_fma = ft.build_fma_oracle()
    assert n_fma > 0
    row = ft.compare("default scene", po.params(160, 90, depth=10, spp=16), po.default_scene())
    assert row["pixels_not_bit_identical"] > 0          # the variant really computes differently ...
    assert row["rmse"] <= 1e-3                          # ... and stays within the stated tolerance
    assert row["rays_strict"] == 2 * 160 * 90 * 16
    assert np.isfinite(row["max_rel_err"])

"""The CPU oracle against the committed golden vectors (tests/golden/*.npz, written by
tests/golden/make_golden.py).  Bit-exact: any change in the oracle's arithmetic shows here."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import make_golden  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("case", sorted(make_golden.CASES))
def test_oracle_reproduces_golden(case):
    want = np.load(os.path.join(GOLDEN, case + ".npz"))
    got = make_golden.run_case(case)
    assert np.array_equal(got["mean"].view(np.uint32), want["mean"].view(np.uint32))
    assert np.array_equal(got["last_frame"].view(np.uint32), want["last_frame"].view(np.uint32))
    assert np.array_equal(got["rng"], want["rng"])
    assert int(got["rays"]) == int(want["rays"]) and int(got["draws"]) == int(want["draws"])
    assert np.array_equal(got["rays_per_bounce"], want["rays_per_bounce"])


def test_sharded_mean_golden(oracle):
    want = np.load(os.path.join(GOLDEN, "default_64x48_8spp_4shards.npz"))
    p = oracle.params(64, 48, rng_mode=oracle.RNG_STRIDED, stride=35, spp=8)
    mean, st = oracle.render_sharded(p, 4)
    assert np.array_equal(mean.view(np.uint32), want["mean"].view(np.uint32)) and st.rays == int(want["rays"])
    # against the plain running mean of the same 8 samples: same up to fp32 rounding of the regrouped sum
    seq, _, _, _ = oracle.render(oracle.params(64, 48, spp=8))
    assert np.allclose(mean, seq, rtol=2e-6, atol=1e-7)


def test_sharded_is_tree_of_shard_means(oracle):
    w, h, spp, shards = 24, 16, 8, 4
    p = oracle.params(w, h, rng_mode=oracle.RNG_STRIDED, stride=35, spp=spp)
    mean, _ = oracle.render_sharded(p, shards)
    parts = []
    for v in range(shards):
        q = oracle.params(w, h, rng_mode=oracle.RNG_STRIDED, stride=35, spp=1)
        m = np.ones((w * h, 4), dtype=np.float32)
        for s in range(spp // shards):
            q.frame_first = v * (spp // shards) + s
            _, _, _, frame = oracle.render(q, want_last_frame=True)
            r = np.float32(1.0) / np.float32(s + 1)
            m = frame * r + (m * np.float32(s)) * r
        parts.append(m.astype(np.float32))
    tree = ((parts[0] + parts[1]) + (parts[2] + parts[3])) * np.float32(0.25)
    assert np.array_equal(tree.view(np.uint32), mean.view(np.uint32))

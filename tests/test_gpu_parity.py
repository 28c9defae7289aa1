"""GPU parity: the HIP plugin, called through its C-ABI, against the CPU oracle -- bit for bit.

Tolerance: none.  Every comparison is on the uint32 view of the float32 images (and on the raw
RNG words); a single differing bit fails."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ENGINES = [("fused", 1), ("wavefront", 0)]


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def assert_bit_equal(img, ref, what):
    diff = (bits(img) != bits(ref)).any(axis=1)
    assert not diff.any(), f"{what}: {int(diff.sum())} of {len(diff)} pixels differ; first at pixel {int(np.argmax(diff))}: {img[np.argmax(diff)]} vs {ref[np.argmax(diff)]}"


@pytest.mark.parametrize("name,engine", ENGINES)
@pytest.mark.parametrize("w,h,spp", [(64, 48, 1), (160, 120, 4), (133, 71, 3), (640, 480, 1)])
def test_default_scene_sequential(pt, oracle, name, engine, w, h, spp):
    ref, ref_rng, ref_stats, _ = oracle.render(oracle.params(w, h, spp=spp))
    with pt.PathTracer(w, h) as tr:
        tr.set_options(engine=engine)
        tr.render(spp)
        img, rng, st = tr.readback(), tr.readback_rng(), tr.stats()
    assert_bit_equal(img, ref, f"{name} {w}x{h}x{spp}")
    assert np.array_equal(rng, ref_rng)
    assert st.rays == ref_stats.rays == 2 * w * h * spp
    assert st.frames == spp


@pytest.mark.parametrize("name,engine", ENGINES)
def test_begin_finalize_equals_batched(pt, oracle, name, engine):
    w, h, spp = 96, 64, 5
    ref, _, _, _ = oracle.render(oracle.params(w, h, spp=spp))
    with pt.PathTracer(w, h) as tr:
        tr.set_options(engine=engine)
        for _ in range(spp):
            tr.BeginDrawing()
            tr.Update(pt.CameraData())
            tr.FinalizeDrawing()
        img = tr.readback()
    assert_bit_equal(img, ref, name)


@pytest.mark.parametrize("name,engine", ENGINES)
def test_strided_35_equals_sequential(pt, oracle, name, engine):
    w, h, spp = 128, 72, 6
    ref, _, _, _ = oracle.render(oracle.params(w, h, spp=spp))
    with pt.PathTracer(w, h) as tr:
        tr.set_options(engine=engine, rng_mode=pt.PT_RNG_STRIDED, stride=35, samples_per_pass=4)
        tr.render(spp)
        img = tr.readback()
    assert_bit_equal(img, ref, name)

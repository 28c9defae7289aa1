"""Camera-ray culling at its borderline, on the GPU: triangles whose edges run along pixel boundaries (vertices on the rays
through pixel corners, tests/test_device_code_on_host.py pixel_corner_scene), where the wave ORs the need masks of its 64
pixels (primary_need_mask).  Both engines, every pixel against the oracle, bit for bit."""
import numpy as np
import pytest

from test_device_code_on_host import pixel_corner_scene

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,engine", [("fused", 1), ("wavefront", 0)])
@pytest.mark.parametrize("w,h,seed", [(64, 40, 1), (96, 54, 5), (200, 33, 7)])
def test_edges_on_pixel_boundaries(pt, oracle, name, engine, w, h, seed):
    V, T, M = pixel_corner_scene(oracle, w, h, 31, seed)          # 62 triangles: inside the 64-bit need mask
    spp, depth = 6, 6
    ref, ref_rng, st, _ = oracle.render(oracle.params(w, h, depth=depth, spp=spp), scene=oracle.make_scene(V, T, M))
    with pt.PathTracer(w, h) as tr:
        tr.set_scene(*pt.make_scene(V, T, M))
        tr.set_options(engine=engine, depth=depth, samples_per_pass=3)
        tr.render(spp)
        img, rng, got = tr.readback(), tr.readback_rng(), tr.stats()
    diff = (img.view(np.uint32) != ref.view(np.uint32)).any(axis=1)
    assert not diff.any(), f"{name}: {int(diff.sum())} pixels differ, first {int(np.argmax(diff))}"
    assert np.array_equal(rng, ref_rng) and got.rays == st.rays


@pytest.mark.parametrize("name,engine", [("fused", 1), ("wavefront", 0)])
def test_room_with_a_tessellated_ball(pt, oracle, name, engine):
    """classes for the walls, free triangles (ClassBlock::kappa_free) for the ball's facets: every pixel against the oracle"""
    from test_bvh_adversarial import room_with_a_ball
    V, T, M = room_with_a_ball(300, 1200, 8)
    w, h, spp, depth = 96, 64, 3, 8
    ref, ref_rng, st, _ = oracle.render(oracle.params(w, h, depth=depth, spp=spp), scene=oracle.make_scene(V, T, M))
    with pt.PathTracer(w, h) as tr:
        tr.set_scene(*pt.make_scene(V, T, M))
        tr.set_options(engine=engine, depth=depth)
        tr.render(spp)
        img, rng, got = tr.readback(), tr.readback_rng(), tr.stats()
    diff = (img.view(np.uint32) != ref.view(np.uint32)).any(axis=1)
    assert not diff.any(), f"{name}: {int(diff.sum())} pixels differ, first {int(np.argmax(diff))}"
    assert np.array_equal(rng, ref_rng) and got.rays == st.rays

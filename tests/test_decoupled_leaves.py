"""The decoupled-leaves form of k_mega3 (csrc/crt_mega3.h: Pool4LdsT; crt_mega3.hip:, inner_arm_dec / leaf_arm_dec), forced with CRT_DEC=1 on
the cases where its differences from the coupled form could show: the tie rule across leaves now rests on one LDS atomic minimum
over (distance, ~triangle) instead of the order of the visits (DeviceBVH.cuh:34-41,144; csrc/crt_trace.h), leaves of many records
are resolved inside one queue entry, rays on the reference-arithmetic path hand their leaves over one by one, a scene that is one
leaf has no inner node at all, and the commit ring shares the smaller pool.  Everything is compared with the oracle, bit for bit,
through the C ABI -- the cases themselves are the ones of test_gpu_parity.py / test_adversarial_traversal.py / test_commit_ring.py."""
import numpy as np
import pytest

import cudaraytracing_amd as crt
import test_adversarial_traversal as A
import test_commit_ring as R
import test_gpu_parity as P
import util

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["1", "0"], ids=["decoupled", "coupled"])
def form(monkeypatch, request):
    """Every case in both forms: CRT_TRAVERSAL_EXACT takes the decoupled one by default, so the coupled one is forced here too."""
    for k in ("CRT_REF16", "CRT_REF32"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("CRT_DEC", request.param)


@pytest.mark.parametrize("thresh", [1, 2, 4])
def test_ties_across_leaves_and_degenerate_triangles(tmp_path, thresh):
    P.test_triangle_soup_with_ties_and_degenerate_triangles(tmp_path, thresh)


@pytest.mark.parametrize("thresh", [1, 3, 20, 200])
def test_leaf_sizes(tmp_path, thresh):
    P.test_other_leaf_sizes(tmp_path, thresh)


def test_one_leaf_of_150_records(tmp_path):
    P.test_leaf_of_300_triangles(tmp_path)


def test_180000_triangles_with_32_bit_stack_entries(tmp_path):
    """60 000 four-wide nodes: beyond the 16-bit layout, three 32-bit levels in LDS and the rest in the spill area."""
    P.test_room_of_180000_triangles(tmp_path)


@pytest.mark.parametrize("specular", [False, True])
def test_reference_arithmetic_rays(tmp_path, specular):
    P.test_rays_with_non_finite_operands_take_the_reference_arithmetic(tmp_path, specular)


@pytest.mark.parametrize("scale", [1e-12, 1e17])
def test_extreme_scales(tmp_path, scale):
    P.test_extreme_coordinate_scales(tmp_path, scale)


@pytest.mark.parametrize("thresh", [1, 2, 4])
def test_grazing_rays(tmp_path, thresh):
    A.test_rays_grazing_triangle_planes(tmp_path, thresh)


@pytest.mark.parametrize("name", ["cornell-box", "veach-mis"])
def test_visibility_queries(name):
    A.test_visibility_queries(name)


def test_commit_ring_on_the_smaller_pool():
    R.test_tiny_rings_reproduce_the_frame("cornell-box", 203, 149, 48, 2)
    R.test_a_shard_of_the_frame_through_the_ring()


@pytest.mark.parametrize("name", ["cornell-box", "veach-mis"])
def test_crop_at_full_spp_and_visit_counts(name):
    """A crop of the BASELINE configuration (C2 / C3) at its full spp; and on a small frame the visits of the counting kernel: the
    decoupled form never tests a leaf the reference does not test (REFERENCE mode's counters are the oracle's), and tests every
    leaf the reference tests for the rays that are not any-hit rays (no emitter: every ray is a closest-hit ray)."""
    t = util.task(name)
    eye, iv, fov = util.camera(name)
    spp = 512 if name == "cornell-box" else 1024
    r = crt.Render(util.host_scene(name), spp, t.P_RR, t.light_sample_n)
    try:
        r.traversal = crt.TRAVERSAL_EXACT
        x0, y0, cw, ch = 352, 264, 24, 16
        orgb, omean, _, st = util.oracle_scene(name).render(eye, iv, fov, 800, 600, spp, t.P_RR, t.light_sample_n, crop=(x0, y0, cw, ch))
        rgb = r.run_view(eye, iv, fov, width=800, height=600)
        assert np.array_equal(util.bits(r.mean_buffer[y0:y0 + ch, x0:x0 + cw]), util.bits(omean))
        assert np.array_equal(rgb[y0:y0 + ch, x0:x0 + cw], orgb)
        r.set_spp(3)
        r.run_view(eye, iv, fov, stats=True, width=96, height=72)
        dec = dict(r.stats)
        r.traversal = crt.TRAVERSAL_REFERENCE
        r.run_view(eye, iv, fov, stats=True, width=96, height=72)
        ref = dict(r.stats)
        assert dec["rays"] == ref["rays"]
        assert 0 < dec["leaf_pops"] <= ref["leaf_pops"] and 0 < dec["tri_tests"] <= ref["tri_tests"]
        # without next-event samples there are no any-hit rays: the same leaves, the same triangle tests
        r.set_light_sample_n(0)
        r.run_view(eye, iv, fov, stats=True, width=96, height=72)
        ref0 = dict(r.stats)
        r.traversal = crt.TRAVERSAL_EXACT
        r.run_view(eye, iv, fov, stats=True, width=96, height=72)
        assert r.stats["rays"] == ref0["rays"] and r.stats["leaf_pops"] == ref0["leaf_pops"] and r.stats["tri_tests"] == ref0["tri_tests"]
        assert r.stats["hits"] == ref0["hits"]
    finally:
        r.free()

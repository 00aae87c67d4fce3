"""A mesh of the reference's real size.  The reference's cornell-box.obj is not in its snapshot (/root/reference/.MISSING_LARGE_BLOBS:1-2;
asset/cornell-box.png shows Suzanne and the Stanford bunny, ~69 000 triangles for the bunny alone); the committed stand-in has 40 972
triangles, whose 24 588 leaf records fit the 16-bit traversal-stack entries of the render kernel's coupled pool.  This variant of the
stand-in (scenes/gen_cornell_box.py --detail 6,5: 102 412 triangles, 61 000 leaf records, 25 000 four-wide nodes) does not: it renders
with the decoupled-leaves pool (csrc/crt_mega3.h: the stack holds inner nodes only, six 16-bit levels), chosen by the library itself.
Oracle crops, EXACT == REFERENCE on a whole frame, and the same frame from every other layout."""
import os
import sys

import numpy as np
import pytest

import cudaraytracing_amd as crt
import oracle_lib as O
import util

sys.path.insert(0, os.path.join(util.ROOT, "scenes"))
import gen_cornell_box  # noqa: E402


@pytest.fixture(scope="module")
def big(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("cornell_detail_6_5"))
    obj, mtl, n = gen_cornell_box.write_variant(d, (6, 5))
    assert n == 102412
    return obj, mtl


def test_the_generator_reproduces_the_committed_scene(tmp_path):
    text, n = gen_cornell_box.generate()
    assert n == 40972
    with open(os.path.join(util.ROOT, "scenes", "cornell-box", "cornell-box.obj")) as f:
        assert f.read() == text
    assert gen_cornell_box.parse_detail("6") == (6, 6) and gen_cornell_box.parse_detail("6,5") == (6, 5)


@pytest.mark.gpu
def test_a_102412_triangle_mesh_on_the_large_mesh_layout(big, monkeypatch):
    obj, mtl = big
    t = util.task("cornell-box")
    eye, iv, fov = util.camera("cornell-box")
    for k in ("CRT_DEC", "CRT_REF16", "CRT_REF32"):
        monkeypatch.delenv(k, raising=False)
    scene = crt.Scene(800, 600)
    scene.add_obj(obj, mtl)
    scene.set_BVH(t.bvh_thresh_n)
    osc = O.OracleScene([(obj, mtl)], t.bvh_thresh_n)
    assert scene.nodes().tobytes() == osc.nodes().tobytes()
    r = crt.Render(scene, 4, t.P_RR, t.light_sample_n)
    try:
        ai = r.accel_info()
        assert ai["n_leaves"] > 32768 and ai["n_nodes4"] <= 32768
        assert ai["layout_caps"] == 2 | 4 | 8  # leaf refs beyond 16 bits, four-wide nodes within: decoupled leaves, 16-bit stack entries, on the tree without rows of refs
        # ---- crops of the 800 x 600 frame against the oracle ----
        r.traversal = crt.TRAVERSAL_EXACT
        rgb = r.run_view(eye, iv, fov, width=800, height=600)
        mean = r.mean_buffer.copy()
        for (x0, y0, cw, ch) in ((170, 330, 24, 16), (500, 250, 24, 16), (388, 40, 16, 8)):   # the two meshes, the light
            orgb, omean, _, st = osc.render(eye, iv, fov, 800, 600, 4, t.P_RR, t.light_sample_n, crop=(x0, y0, cw, ch))
            assert np.array_equal(util.bits(mean[y0:y0 + ch, x0:x0 + cw]), util.bits(omean)), (x0, y0)
            assert np.array_equal(rgb[y0:y0 + ch, x0:x0 + cw], orgb)
        rays = r.stats["rays"]
        # ---- the whole frame: EXACT == REFERENCE (the reference's topology, order and arithmetic), and every other layout ----
        r.traversal = crt.TRAVERSAL_REFERENCE
        ref_rgb = r.run_view(eye, iv, fov, width=800, height=600)
        assert np.array_equal(util.bits(r.mean_buffer), util.bits(mean)) and np.array_equal(ref_rgb, rgb) and r.stats["rays"] == rays
        r.traversal = crt.TRAVERSAL_EXACT
        for env in ({"CRT_DEC": "0"}, {"CRT_DEC": "1", "CRT_REF32": "1"}):   # coupled, 32-bit entries; decoupled, 32-bit entries
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            rgb2 = r.run_view(eye, iv, fov, width=800, height=600)
            assert np.array_equal(util.bits(r.mean_buffer), util.bits(mean)) and np.array_equal(rgb2, rgb) and r.stats["rays"] == rays, env
            for k in env:
                monkeypatch.delenv(k)
        # ---- closest hits and visibility of random rays through the same kernels ----
        o, d = util.random_rays("cornell-box", 4096, seed=23)
        otri, ot, _ = osc.intersect(o, d)
        for mode in (crt.TRAVERSAL_EXACT, crt.TRAVERSAL_REFERENCE):
            tri, tt = r.intersect(o, d, traversal=mode)
            assert np.array_equal(tri, otri) and np.array_equal(util.bits(tt), util.bits(ot)), mode
    finally:
        r.free()

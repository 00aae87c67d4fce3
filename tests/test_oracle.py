"""CPU tests of the oracle (test infrastructure): pinned against the reference's vendored Eigen /
Camera.h golden vectors, against libm, against published Philox known answers, against the
structural numbers the reference prints for veach-mis, and against
its own committed known answers (tests/golden/oracle_kat.json)."""
import hashlib
import json
import os

import numpy as np
import pytest

import oracle_lib as O
import util

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_vector_ops_match_reference_eigen_bit_for_bit():
    """Every vector op / scalar chain of the hot path against Eigen 3.4.90 as vendored by the reference."""
    recs = json.load(open(os.path.join(GOLDEN, "eigen_ops.json")))
    ops = {}
    for r in recs:
        got = O.vec_op(r["op"], r["in"])
        assert np.array_equal(got, np.array(r["out"], dtype=np.uint32)), r["op"]
        ops[r["op"]] = ops.get(r["op"], 0) + 1
    assert ops["inverse_view"] == 64 and ops["dot"] >= 64 and len(ops) >= 24


def test_philox_known_answers():
    # Random123 kat_vectors, philox4x32-10
    assert [hex(x) for x in O.philox([0, 0, 0, 0], [0, 0])] == ["0x6627e8d5", "0xe169c58d", "0xbc57ac4c", "0x9b00dbd8"]
    assert [hex(x) for x in O.philox([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2)] == ["0x408f276d", "0x41c83b0e", "0xa20bc7c6", "0x6d5451fd"]
    assert [hex(x) for x in O.philox([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0])] == \
        ["0xd16cfe09", "0x94fdcceb", "0x5001e420", "0x24126ea1"]


def test_rng_addressing_and_uniform_range():
    u, f = O.rng_draw(seed=(7 << 32) | 5, pixel=1234, k=3, depth=2, purpose=1, idx=9)
    ref = O.philox([3, 2 | (1 << 16), 9, 7], [1234, 5])
    assert np.array_equal(u, ref)
    assert np.all(f > 0) and np.all(f <= 1)
    # curand_uniform mapping at the extremes: (0, 1]
    x = np.array([0, 0xFFFFFFFF], dtype=np.uint32)
    m = (x.astype(np.float32) * np.float32(2.3283064365386963e-10) + np.float32(1.1641532182693481e-10)).astype(np.float32)
    assert m[0] > 0 and m[1] == 1.0


def _ulp_err(got, ref):
    ref = ref.astype(np.float64)
    u = np.spacing(np.abs(ref).astype(np.float32)).astype(np.float64)
    return np.max(np.abs(got.astype(np.float64) - ref) / u)


def test_deterministic_math_close_to_libm():
    rng = np.random.default_rng(1)
    x = rng.uniform(-30, 30, 100000).astype(np.float32)
    assert _ulp_err(O.math_fn("sin", x), np.sin(x.astype(np.float64))) < 3
    assert _ulp_err(O.math_fn("cos", x), np.cos(x.astype(np.float64))) < 3
    y = rng.uniform(-1, 1, 100000).astype(np.float32)
    assert _ulp_err(O.math_fn("acos", y), np.arccos(y.astype(np.float64))) < 3
    assert _ulp_err(O.math_fn("atan2", x, y), np.arctan2(x.astype(np.float64), y.astype(np.float64))) < 5
    assert _ulp_err(O.math_fn("exp", x), np.exp(x.astype(np.float64))) < 3
    p = rng.uniform(1e-6, 1e5, 100000).astype(np.float32)
    assert _ulp_err(O.math_fn("log10", p), np.log10(p.astype(np.float64))) < 4
    c = rng.uniform(0, 1, 100000).astype(np.float32)
    assert _ulp_err(O.math_fn("pow", c, np.full_like(c, 0.6)), np.power(c.astype(np.float64), float(np.float32(0.6)))) < 16
    # exact special values
    assert O.math_fn("pow", np.array([0.0, 1.0], np.float32), np.array([0.6, 0.6], np.float32)).tolist() == [0.0, 1.0]
    assert O.math_fn("sin", np.array([0.0], np.float32))[0] == 0.0 and O.math_fn("cos", np.array([0.0], np.float32))[0] == 1.0
    assert np.isnan(O.math_fn("acos", np.array([1.5], np.float32))[0])
    assert O.math_fn("atan2", np.array([0.0, 1.0, -1.0], np.float32), np.array([0.0, 0.0, 0.0], np.float32)).tolist() == \
        [0.0, float(np.float32(np.pi / 2)), -float(np.float32(np.pi / 2))]


def test_tonemap_semantics():
    c = np.array([-1.0, 0.0, 1e-9, 0.25, 1.0, 7.0, np.nan], dtype=np.float32)
    t = O.tonemap(c)
    assert t[0] == 0 and t[1] == 0 and t[4] == 255 and t[5] == 255 and t[6] == 0  # clamp; NaN -> 0
    assert t[3] == int(255 * float(O.math_fn("pow", np.array([0.25], np.float32), np.array([0.6], np.float32))[0]))


def test_veach_structure_matches_what_the_reference_prints():
    """counts and object areas measured from the reference itself."""
    sc = util.oracle_scene("veach-mis")
    assert (sc.num_tris, sc.num_nodes, sc.root, sc.num_lights) == (3092, 4095, 4094, 4)
    assert [sc.light_size(i) for i in range(4)] == [760] * 4
    objs = sc.objects()
    light_areas = [a for is_light, a in objs if is_light]
    np.testing.assert_allclose(light_areas, [0.031062, 0.776560, 3.106239, 12.424994], rtol=0, atol=6e-7)  # printed with %f
    plates = [a for is_light, a in objs if not is_light][:4]
    np.testing.assert_allclose(plates, [66.1605] * 4, rtol=0, atol=1e-3)
    assert [a for is_light, a in objs if not is_light][4:] == [940.89599609375, 940.89599609375]
    nodes = sc.nodes()
    leaves = (nodes["lc"] < 0) & (nodes["rc"] < 0)
    assert leaves.sum() == 2048 and nodes["n"][leaves].max() == 2


def test_cornell_stand_in_structure():
    sc = util.oracle_scene("cornell-box")
    assert (sc.num_tris, sc.num_nodes, sc.num_lights, sc.light_size(0)) == (40972, 49175, 1, 2)
    tris = sc.tris()
    assert tris["has_emit"].sum() == 2 and (tris["mode"] == 0).all()


def test_reference_traversal_statistics_match_survey():
    """Per-ray traversal costs the survey measured on the reference for veach-mis:
    94.3 node pops, 23.9 triangle tests, 7.20 rays per path."""
    sc = util.oracle_scene("veach-mis")
    t = util.task("veach-mis")
    eye, iv, fov = util.camera("veach-mis")
    _, _, _, st = sc.render(eye, iv, fov, 160, 120, 4, t.P_RR, t.light_sample_n)
    assert abs(st["rays"] / st["paths"] - 7.20) < 0.15
    assert abs((st["inner_pops"] + st["leaf_pops"]) / st["rays"] - 94.3) < 2.0
    assert abs(st["tri_tests"] / st["rays"] - 23.9) < 1.0
    assert st["max_bvh_stack"] <= 16


def test_oracle_known_answers():
    kat = json.load(open(os.path.join(GOLDEN, "oracle_kat.json")))
    for name, e in kat.items():
        sc = util.oracle_scene(name)
        t = util.task(name)
        eye, iv, fov = util.camera(name)
        nodes, tris = sc.nodes(), sc.tris()
        assert sha(nodes) == e["nodes_sha256"]
        assert sha(tris["v1"] + tris["v2"] + tris["v3"]) == e["centroid_order_sha256"]
        assert [int(np.float32(a).view(np.uint32)) for _, a in sc.objects()] == e["object_areas_bits"]
        o, d = util.random_rays(name, 4096, seed=11)
        tri, tt, st = sc.intersect(o, d)
        assert sha(tri) == e["intersect"]["tri_sha256"] and sha(tt) == e["intersect"]["t_sha256"]
        for k, v in e["intersect"]["counters"].items():
            assert st[k] == v
        rgb, mean, L, st = sc.render(eye, iv, fov, t.width, t.height, 4, t.P_RR, t.light_sample_n, seed=0,
                                     crop=(368, 268, 64, 48), want_L=True)
        img = e["image_crop_368_268_64x48_spp4"]
        assert sha(rgb) == img["rgb_sha256"] and sha(mean) == img["mean_sha256"] and sha(L) == img["L_sha256"]
        assert {k: st[k] for k in st} == img["counters"]


def test_mean_is_sequential_sum_of_per_path_radiance():
    """temp_color += L / spp in sample order (Render.cuh:348)."""
    sc = util.oracle_scene("cornell-box")
    t = util.task("cornell-box")
    eye, iv, fov = util.camera("cornell-box")
    _, mean, L, _ = sc.render(eye, iv, fov, 800, 600, 5, t.P_RR, t.light_sample_n, crop=(380, 280, 16, 8), want_L=True)
    acc = np.zeros_like(mean)
    for k in range(5):
        acc = acc + L[:, :, k, :] / np.float32(5)
    assert np.array_equal(acc.view(np.uint32), mean.view(np.uint32))


def test_crop_equals_full_frame_region():
    sc = util.oracle_scene("veach-mis")
    t = util.task("veach-mis")
    eye, iv, fov = util.camera("veach-mis")
    _, full, _, _ = sc.render(eye, iv, fov, 64, 48, 2, t.P_RR, t.light_sample_n)
    _, crop, _, _ = sc.render(eye, iv, fov, 64, 48, 2, t.P_RR, t.light_sample_n, crop=(8, 16, 24, 12))
    assert np.array_equal(full[16:28, 8:32].view(np.uint32), crop.view(np.uint32))


def test_sampler_properties():
    rng = np.random.default_rng(5)
    for _ in range(200):
        n = rng.normal(size=3).astype(np.float32)
        n /= np.linalg.norm(n)
        v = O.sample_hemisphere(n, rng.uniform(1e-6, 1), rng.uniform(1e-6, 1))
        assert abs(np.linalg.norm(v) - 1) < 1e-5
        assert np.dot(v, n) >= -1e-6  # z = |1 - 2 x1| keeps samples in the upper hemisphere (Global.h:61)
    out = np.array([0.3, 0.5, 0.8], dtype=np.float32)
    centre = O.sample_lobe(out, 0.2, 0.4, 0.5, 0.5)  # eta = 0: the lobe centre is out / |out|
    np.testing.assert_allclose(centre, out / np.linalg.norm(out), atol=2e-6)


def test_ray_log_and_margin_histogram_diagnostics():
    """The diagnostics behind tools/diff_fast_reference.py and tools/margin_hist.py: the ray log holds every ray of a render (count =
    the `rays` statistic; closest-hit rays carry a NaN limit, next-event samples their t_to_light), replaying its rays through
    orc_intersect gives the logged answers, the hit leaf's box entry never lies far behind the hit, and the margin histogram counts
    the same rays."""
    import ctypes as C
    sc = util.oracle_scene("veach-mis")
    t = util.task("veach-mis")
    eye, iv, fov = util.camera("veach-mis")
    L = O.lib()
    L.orc_ray_log_begin.restype = None
    L.orc_ray_log_end.restype = C.c_uint64
    L.orc_ray_log_end.argtypes = [C.c_void_p, C.c_uint64]
    L.orc_margin_hist.restype = None
    L.orc_margin_hist.argtypes = [C.c_void_p]
    L.orc_margin_hist(None)
    L.orc_ray_log_begin()
    _, _, _, st = sc.render(eye, iv, fov, 64, 48, 3, t.P_RR, t.light_sample_n, crop=(20, 20, 8, 8))
    n = int(L.orc_ray_log_end(None, 0))
    log = np.zeros((n, 10), dtype=np.float32)
    L.orc_ray_log_end(log.ctypes.data_as(C.c_void_p), n)
    hist = np.zeros(26, dtype=np.uint64)
    L.orc_margin_hist(hist.ctypes.data_as(C.c_void_p))
    assert n == st["rays"] == int(hist[25])
    vis = ~np.isnan(log[:, 8])
    assert int(vis.sum()) == st["shadow_rays"]
    # replay (the oracle normalises a direction again: a Ray's own direction is a fixed point of that up to the last bit, so compare ids)
    tri, tt, _ = sc.intersect(log[:, 0:3], log[:, 3:6])
    same = tri == log[:, 7].astype(np.int32)
    assert same.mean() > 0.999
    hit = log[:, 7] >= 0
    reach = np.abs(log[:, 0:3]).max(axis=1) + np.abs(log[:, 6])
    with np.errstate(divide="ignore"):
        steep = np.abs(1.0 / log[:, 3:6]).max(axis=1)
    margin = (log[hit, 9] - log[hit, 6]) / (reach[hit] * steep[hit])
    assert np.all(margin < 1.0e-4)  # (the pruning slack factor of CRT_TRAVERSAL_FAST; a ray beyond it would be a find, not a failure of the oracle)
    assert int(hist[24]) <= int(hit.sum())

"""Adversarial probes of CRT_TRAVERSAL_FAST's pruning rule (csrc/crt_trace.h: prune_bound).

FAST skips a box whose slab entry distance lies beyond prune_bound(best_t) = best_t + (|best_t| * 1e-3 + 1e-3) + CRT_PRUNE_REL x
reach x steep (reach = max |origin coordinate| + |best_t|, steep = max |1 / direction component|).  That is exact as long as the
Moeller-Trumbore distance of every triangle inside the box agrees with the box's own entry distance to within the slack;
Moeller-Trumbore has no bounded relative error for rays that graze a triangle's plane (determinant -> 0), for sliver
triangles, or far from the origin (cancellation in o - v1), so the contract of include/crt.h is "identical to REFERENCE on
every probe of this file and on every scene of the suite", not a theorem.  These tests aim at exactly those cases and
compare FAST with the exhaustive REFERENCE traversal (GPU) and with the oracle, closest-hit ids and distance bits
included."""
import os

import numpy as np
import pytest

import cudaraytracing_amd as crt
import oracle_lib as O
import util
from test_gpu_parity import _write_box_scene, _write_soup_scene

pytestmark = pytest.mark.gpu


def _load(obj, mtl, thresh, w=48, h=36):
    scene = crt.Scene(w, h)
    scene.add_obj(obj, mtl)
    scene.set_BVH(thresh)
    osc = O.OracleScene([(obj, mtl)], thresh)
    assert scene.nodes().tobytes() == osc.nodes().tobytes()
    return scene, osc


def _check_rays(r, osc, o, d):
    otri, ot, _ = osc.intersect(o, d)
    out = {}
    for mode in (crt.TRAVERSAL_REFERENCE, crt.TRAVERSAL_FAST, crt.TRAVERSAL_EXACT):
        tri, t = r.intersect(o, d, traversal=mode)
        out[mode] = (tri, t)
        bad = np.nonzero((tri != otri) | (util.bits(t) != util.bits(ot)))[0]
        assert bad.size == 0, (mode, bad[:8], tri[bad[:8]], otri[bad[:8]], t[bad[:8]], ot[bad[:8]])
    return otri, ot


def _grazing_rays(tris, rng, n_per, angles):
    """Rays that cross triangle planes at the given (tiny) angles, aimed at a point inside the triangle, from both sides,
    from near and far; plus rays lying exactly in the plane (determinant exactly or nearly zero)."""
    o_list, d_list = [], []
    for v in tris:
        e1, e2 = v[1] - v[0], v[2] - v[0]
        n = np.cross(e1, e2)
        ln = np.linalg.norm(n)
        if ln == 0:
            continue
        n /= ln
        for _ in range(n_per):
            a, b = rng.uniform(0.05, 0.9), rng.uniform(0.05, 0.9)
            if a + b > 0.95:
                a, b = 0.3, 0.3
            p = v[0] + a * e1 + b * e2
            tdir = np.cos(rng.uniform(0, 2 * np.pi)) * e1 / np.linalg.norm(e1) + np.sin(rng.uniform(0, 2 * np.pi)) * np.cross(n, e1) / np.linalg.norm(e1)
            tdir /= np.linalg.norm(tdir)
            for th in angles:
                for sgn in (1.0, -1.0):
                    d = np.cos(th) * tdir + sgn * np.sin(th) * n
                    s = rng.choice([0.01, 0.5, 3.0, 40.0])
                    o_list.append(p - s * d)
                    d_list.append(d)
    return np.asarray(o_list, dtype=np.float32), np.asarray(d_list, dtype=np.float32)


def _soup_triangles(obj):
    vs, fs = [], []
    for line in open(obj):
        t = line.split()
        if not t:
            continue
        if t[0] == "v":
            vs.append([float(x) for x in t[1:4]])
        elif t[0] == "f":
            fs.append([int(x.split("/")[0]) - 1 for x in t[1:4]])
    vs = np.asarray(vs, dtype=np.float64)
    return [vs[f] for f in fs]


@pytest.mark.parametrize("thresh", [1, 2, 4])
def test_rays_grazing_triangle_planes(tmp_path, thresh):
    obj, mtl = _write_soup_scene(str(tmp_path), n=300, dup=30, degenerate=10)
    scene, osc = _load(obj, mtl, thresh)
    r = crt.Render(scene, 1, 0.6, 1)
    try:
        rng = np.random.RandomState(7)
        tris = _soup_triangles(obj)
        pick = [tris[i] for i in rng.choice(len(tris), 160, replace=False)]
        o, d = _grazing_rays(pick, rng, 3, [0.0, 1e-7, 1e-6, 1e-5, 1e-4, 1e-3, 1e-2])
        otri, ot = _check_rays(r, osc, o, d)
        assert (otri >= 0).sum() > o.shape[0] // 4  # the probes do hit things
    finally:
        r.free()


def _write_sliver_scene(d, n=500, seed=5):
    """The room with a cloud of sliver triangles in it: aspect ratios 1e3 ... 1e7 (two long edges, one of 1e-3 ... 1e-7),
    needles with nearly collinear vertices, and slivers that share their long edge (equal distances along it)."""
    rng = np.random.RandomState(seed)
    obj, mtl = _write_box_scene(d, n_side=4)
    tris = []
    for i in range(n):
        c = rng.uniform(1.5, 8.5, 3)
        u = rng.normal(size=3); u /= np.linalg.norm(u)
        w = np.cross(u, rng.normal(size=3)); w /= np.linalg.norm(w)
        length = rng.uniform(0.5, 4.0)
        width = 10.0 ** rng.uniform(-7, -3)
        a = c - 0.5 * length * u
        b = c + 0.5 * length * u
        cc = c + rng.uniform(-0.4, 0.4) * length * u + width * w
        tris.append(np.stack([a, b, cc]))
        if i % 5 == 0:  # a second sliver on the same long edge, folded the other way
            tris.append(np.stack([a, b, c - width * w]))
    with open(obj) as f:
        nv = sum(1 for line in f if line.startswith("v "))
    with open(obj, "a") as o:
        o.write("usemtl floor\n")
        for t in tris:
            for v in t.astype(np.float32):
                o.write("v %.9g %.9g %.9g\nvn 0 1 0\nvt 0 0\n" % tuple(v))
            o.write("f %d/%d/%d %d/%d/%d %d/%d/%d\n" % (nv + 1, nv + 1, nv + 1, nv + 2, nv + 2, nv + 2, nv + 3, nv + 3, nv + 3))
            nv += 3
    return obj, mtl, tris


@pytest.mark.parametrize("thresh", [1, 2])
def test_sliver_triangles(tmp_path, thresh):
    obj, mtl, tris = _write_sliver_scene(str(tmp_path))
    scene, osc = _load(obj, mtl, thresh, 64, 48)
    eye = np.array([5.0, 5.0, 0.5], dtype=np.float32)
    iv = crt.get_inverse_view_matrix(eye, [5.0, 4.5, 9.0], [0.0, 1.0, 0.0])
    fov = crt.fov_to_radians(75.0)
    r = crt.Render(scene, 2, 0.6, 2)
    r.seed = 9
    try:
        orgb, omean, _, st = osc.render(eye, iv, fov, 64, 48, 2, 0.6, 2, seed=9)
        for mode in (crt.TRAVERSAL_REFERENCE, crt.TRAVERSAL_FAST, crt.TRAVERSAL_EXACT):
            r.traversal = mode
            rgb = r.run_view(eye, iv, fov)
            assert np.array_equal(util.bits(r.mean_buffer), util.bits(omean)), mode
            assert np.array_equal(rgb, orgb) and r.stats["rays"] == st["rays"]
        rng = np.random.RandomState(3)
        # rays aimed along the slivers (at points of their long edges), across them, and grazing their planes
        o_list, d_list = [], []
        for t in tris[::2]:
            for _ in range(4):
                p = t[0] + rng.uniform(0.02, 0.98) * (t[1] - t[0])
                src = rng.uniform(0.5, 9.5, 3)
                o_list.append(src); d_list.append(p - src)
        o = np.asarray(o_list, dtype=np.float32); d = np.asarray(d_list, dtype=np.float32)
        _check_rays(r, osc, o, d)
        og, dg = _grazing_rays(tris[::7], rng, 2, [0.0, 1e-6, 1e-4, 1e-2])
        _check_rays(r, osc, og, dg)
    finally:
        r.free()


@pytest.mark.parametrize("offset", [1.0e4, 1.0e6, 3.0e7])
def test_scene_far_from_the_origin(tmp_path, offset):
    """Every coordinate translated by `offset` (at 3e7 a float's spacing is 2: vertices collapse onto a coarse lattice,
    o - v1 cancels to a few bits): FAST must still return what REFERENCE returns."""
    obj, mtl = _write_soup_scene(str(tmp_path), n=250, dup=25, degenerate=10)
    lines = open(obj).read().split("\n")
    with open(obj, "w") as f:
        for line in lines:
            if line.startswith("v "):
                x, y, z = (np.float32(float(v) + offset) for v in line.split()[1:4])
                line = "v %.9g %.9g %.9g" % (x, y, z)
            f.write(line + "\n")
    w, h, spp = 48, 36, 2
    scene, osc = _load(obj, mtl, 2, w, h)
    eye = (np.array([5.0, 5.0, 0.5]) + offset).astype(np.float32)
    iv = crt.get_inverse_view_matrix(eye, (np.array([5.0, 4.5, 9.0]) + offset).astype(np.float32), [0.0, 1.0, 0.0])
    fov = crt.fov_to_radians(75.0)
    r = crt.Render(scene, spp, 0.6, 2)
    r.seed = 5
    try:
        orgb, omean, _, st = osc.render(eye, iv, fov, w, h, spp, 0.6, 2, seed=5)
        for mode in (crt.TRAVERSAL_REFERENCE, crt.TRAVERSAL_FAST, crt.TRAVERSAL_EXACT):
            r.traversal = mode
            rgb = r.run_view(eye, iv, fov)
            assert np.array_equal(util.bits(r.mean_buffer), util.bits(omean)), (offset, mode)
            assert np.array_equal(rgb, orgb) and r.stats["rays"] == st["rays"]
        rng = np.random.RandomState(2)
        n = 8192
        o = (rng.uniform(0.5, 9.5, (n, 3)) + offset).astype(np.float32)
        d = rng.normal(size=(n, 3)).astype(np.float32)
        _check_rays(r, osc, o, d)
    finally:
        r.free()


def test_grazing_rays_on_the_shipped_scenes():
    """The same probe on the two benchmark scenes: rays grazing randomly chosen triangles of cornell-box and veach-mis."""
    for name in ("cornell-box", "veach-mis"):
        t = util.task(name)
        osc = util.oracle_scene(name)
        r = crt.Render(util.host_scene(name), 1, t.P_RR, t.light_sample_n)
        try:
            rng = np.random.RandomState(11)
            tr = osc.tris()
            idx = rng.choice(len(tr), 400, replace=False)
            tris = [np.stack([tr["v1"][i], tr["v2"][i], tr["v3"][i]]).astype(np.float64) for i in idx]
            o, d = _grazing_rays(tris, rng, 2, [0.0, 1e-6, 1e-4, 1e-2])
            scale = float(np.abs(tr["v1"]).max())
            o = (o.astype(np.float64)).astype(np.float32)
            _check_rays(r, osc, o, d)
            assert scale > 0
        finally:
            r.free()


# ----------------------------------------------------------------------------------------------
# The fallbacks of the pipeline choice (csrc/crt_render.hip: choose_pipeline): scenes beyond k_mega3's 16-bit leaf offsets,
# 32-bit byte offsets or 8-bit stack depth render with the wavefront pipeline.  The CRT_TEST_* hooks lower each limit so
# that an ordinary scene trips it; the frame must not change (and the many trace launches show which pipeline ran).
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("hook,value", [("CRT_TEST_MAX_LEAF", "1"), ("CRT_TEST_MAX_BYTES", "4096"), ("CRT_TEST_MAX_STACK", "3"),
                                        ("CRT_PIPELINE", "2")])
def test_fallbacks_to_the_wavefront_pipeline(tmp_path, monkeypatch, hook, value):
    obj, mtl = _write_soup_scene(str(tmp_path), n=200, dup=20, degenerate=10)
    w, h, spp = 48, 36, 2
    scene, osc = _load(obj, mtl, 2, w, h)
    eye = np.array([5.0, 5.0, 0.5], dtype=np.float32)
    iv = crt.get_inverse_view_matrix(eye, [5.0, 4.5, 9.0], [0.0, 1.0, 0.0])
    fov = crt.fov_to_radians(75.0)
    r = crt.Render(scene, spp, 0.6, 2)
    r.seed = 5
    try:
        orgb, omean, _, st = osc.render(eye, iv, fov, w, h, spp, 0.6, 2, seed=5)
        r.run_view(eye, iv, fov)
        assert r.stats["kernel_launches"] == 1  # k_mega3: one launch per frame
        monkeypatch.setenv(hook, value)
        for mode in (crt.TRAVERSAL_REFERENCE, crt.TRAVERSAL_FAST, crt.TRAVERSAL_EXACT):
            r.traversal = mode
            rgb = r.run_view(eye, iv, fov)
            assert r.stats["kernel_launches"] > 1, hook  # rounds of k_logic + k_trace
            assert np.array_equal(util.bits(r.mean_buffer), util.bits(omean)), (hook, mode)
            assert np.array_equal(rgb, orgb) and r.stats["rays"] == st["rays"]
        rng = np.random.RandomState(4)
        o = rng.uniform(0.5, 9.5, (4096, 3)).astype(np.float32)
        d = rng.normal(size=(4096, 3)).astype(np.float32)
        otri, ot = _check_rays(r, osc, o, d)  # crt_intersect follows the same choice (k_trace)
        # ... and so do the visibility queries: limits on both sides of the closest hit and the degenerate ones
        lim = np.where(ot < np.float32(3.0e38), ot, np.float32(5.0)) * rng.choice(np.array([0.5, 0.99999, 1.0, 1.00001, 1.5], dtype=np.float32), 4096)
        lim[:40] = np.tile(np.array([0.0, -1.0, np.inf, -np.inf, np.nan, 3.0e38, 1.0e-5, 2.0e-5], dtype=np.float32), 5)
        ob = _oracle_blocked(osc, o, d, lim)
        for mode in (crt.TRAVERSAL_REFERENCE, crt.TRAVERSAL_FAST, crt.TRAVERSAL_EXACT):
            blk, _ = r.blocked(o, d, lim, traversal=mode)
            assert np.array_equal(blk, ob), (hook, mode, np.nonzero(blk != ob)[0][:8])
    finally:
        r.free()


@pytest.mark.parametrize("scale", [1e-12, 1e12, 1e17])
def test_extreme_scales_on_the_wavefront_pipeline(tmp_path, monkeypatch, scale):
    """the wavefront fallback must send rays with a non-finite ORIGIN (overflowed hit positions) down the
    reference arithmetic too, as k_mega3's start_ray does."""
    monkeypatch.setenv("CRT_PIPELINE", "2")
    obj, mtl = _write_soup_scene(str(tmp_path), n=200, dup=20, degenerate=10)
    lines = open(obj).read().split("\n")
    with open(obj, "w") as f:
        for line in lines:
            if line.startswith("v "):
                x, y, z = (np.float32(float(v) * scale) for v in line.split()[1:4])
                line = "v %.9g %.9g %.9g" % (x, y, z)
            f.write(line + "\n")
    w, h, spp = 48, 36, 2
    scene, osc = _load(obj, mtl, 2, w, h)
    eye = (np.array([5.0, 5.0, 0.5]) * scale).astype(np.float32)
    iv = crt.get_inverse_view_matrix(eye, (np.array([5.0, 4.5, 9.0]) * scale).astype(np.float32), [0.0, 1.0, 0.0])
    fov = crt.fov_to_radians(75.0)
    r = crt.Render(scene, spp, 0.6, 2)
    r.seed = 5
    try:
        orgb, omean, _, st = osc.render(eye, iv, fov, w, h, spp, 0.6, 2, seed=5)
        for mode in (crt.TRAVERSAL_REFERENCE, crt.TRAVERSAL_FAST, crt.TRAVERSAL_EXACT):
            r.traversal = mode
            rgb = r.run_view(eye, iv, fov)
            assert np.array_equal(util.bits(r.mean_buffer), util.bits(omean)), (scale, mode)
            assert np.array_equal(rgb, orgb) and r.stats["rays"] == st["rays"]
    finally:
        r.free()


def test_scene_description_must_be_one_tree():
    """crt_scene_create rejects descriptions whose leaves the root cannot reach (the FAST tree is built over all leaves)."""
    import ctypes as C
    from cudaraytracing_amd import _capi as capi
    sc = util.host_scene("veach-mis")
    d = sc.desc()
    nodes = sc.nodes().copy()
    bad = capi.SceneDesc()
    C.memmove(C.byref(bad), C.byref(d), C.sizeof(capi.SceneDesc))
    bad.root = int(nodes[sc.root]["lc"])  # a proper subtree: the other half of the leaves is unreachable
    h = C.c_void_p()
    assert capi.lib().crt_scene_create(C.byref(bad), 0, C.byref(h)) == -1
    assert b"does not reach" in capi.lib().crt_last_error()


def test_the_c3_ray_that_the_old_pruning_bound_lost():
    """Known answer from the full-size C3 frame (veach-mis 800x600 spp 1024, pixel (639, 158), 1 ray in 3.5e9): direction
    component d.y = -7.6e-4, the ray grazes the shared edge of two triangles of a light sphere; Moeller-Trumbore accepts the
    nearer triangle (1985, t = 10.44168) at a point 1.7e-5 ABOVE that triangle's box, so the box is entered at t + 0.022 --
    beyond the old bound t' + 0.011 once the farther triangle (2139, t' = 10.442408) had been found.  Every traversal form must
    return the reference's answer; neighbours of the ray (a few ulp around origin and direction) are probed too."""
    name = "veach-mis"
    t = util.task(name)
    osc = util.oracle_scene(name)
    r = crt.Render(util.host_scene(name), 1, t.P_RR, t.light_sample_n)
    try:
        bits = np.array([3231711228, 1087552060, 3236092846, 1055547749, 3125281698, 1063492088], dtype=np.uint32)
        rng = np.random.RandomState(1)
        n = 4096
        ob = np.tile(bits[:3], (n, 1)).astype(np.int64)
        db = np.tile(bits[3:], (n, 1)).astype(np.int64)
        ob[1:] += rng.randint(-4, 5, (n - 1, 3))
        db[1:] += rng.randint(-64, 65, (n - 1, 3))
        o = ob.astype(np.uint32).view(np.float32)
        d = db.astype(np.uint32).view(np.float32)
        # the oracle normalises directions as Ray's constructor does; the queries below must do the same (no RAW flag): both sides
        # see the same float direction
        otri, ot, _ = osc.intersect(o, d)
        seen = set()
        for mode in (crt.TRAVERSAL_REFERENCE, crt.TRAVERSAL_FAST, crt.TRAVERSAL_FAST | crt.INTERSECT_FORCE_EXACT, crt.TRAVERSAL_EXACT,
                     crt.TRAVERSAL_EXACT | crt.INTERSECT_FORCE_EXACT):
            tri, tt = r.intersect(o, d, traversal=mode)
            bad = np.nonzero((tri != otri) | (util.bits(tt) != util.bits(ot)))[0]
            assert bad.size == 0, (mode, bad[:5], tri[bad[:5]], otri[bad[:5]])
            seen |= set(int(v) for v in np.unique(tri))
        assert 1985 in seen  # the probes do hit the triangle in question
        # and the logged ray itself, direction taken as it is
        tri, tt = r.intersect(o[:1], d[:1], traversal=crt.TRAVERSAL_FAST | crt.INTERSECT_RAW_DIRECTIONS)
        assert tri[0] == 1985 and util.bits(tt)[0] == 1093079327
    finally:
        r.free()


def _oracle_blocked(osc, o, d, lim):
    """blocked() of Render.cuh:19-27 from the oracle's closest hit (t = FLT_MAX on a miss)"""
    _, ot, _ = osc.intersect(o, d)
    with np.errstate(invalid="ignore", over="ignore"):
        return (lim.astype(np.float32) - ot) > np.float32(0.00001)


@pytest.mark.parametrize("name", ["cornell-box", "veach-mis"])
def test_visibility_queries(name):
    """crt_intersect with CRT_INTERSECT_VISIBILITY is blocked() of the reference (Render.cuh:19-27, :272) in every traversal mode:
    rays from surface points towards points on other surfaces, limits around the true distance, and the limits the reference's
    t_to_light = dist.x / dir.x can produce (0, negative, infinite, NaN)."""
    t = util.task(name)
    osc = util.oracle_scene(name)
    r = crt.Render(util.host_scene(name), 1, t.P_RR, t.light_sample_n)
    try:
        rng = np.random.RandomState(11)
        tr = osc.tris()
        n = 20000
        def surface_points(k):
            i = rng.randint(0, tr["v1"].shape[0], k)
            a, b = rng.rand(k, 1).astype(np.float32), rng.rand(k, 1).astype(np.float32)
            flip = (a + b) > 1
            a, b = np.where(flip, 1 - a, a), np.where(flip, 1 - b, b)
            return (tr["v1"][i] + a * (tr["v2"][i] - tr["v1"][i]) + b * (tr["v3"][i] - tr["v1"][i])).astype(np.float32)
        o, p = surface_points(n), surface_points(n)
        d = p - o
        dist = np.linalg.norm(d, axis=1).astype(np.float32)
        lim = dist * rng.choice(np.array([1.0, 1.0, 0.5, 0.999999, 1.000001, 1.00001, 2.0, 1e-3], dtype=np.float32), n)
        lim[:64] = np.tile(np.array([0.0, -1.0, np.inf, -np.inf, np.nan, 3.0e38, 1.0e-5, 2.0e-5], dtype=np.float32), 8)
        ob = _oracle_blocked(osc, o, d, lim)
        assert 0.2 < ob.mean() < 0.95  # both answers are well represented
        for mode in (crt.TRAVERSAL_REFERENCE, crt.TRAVERSAL_FAST, crt.TRAVERSAL_FAST | crt.INTERSECT_FORCE_EXACT, crt.TRAVERSAL_EXACT,
                     crt.TRAVERSAL_EXACT | crt.INTERSECT_FORCE_EXACT):
            blk, tri = r.blocked(o, d, lim, traversal=mode)
            bad = np.nonzero(blk != ob)[0]
            assert bad.size == 0, (mode, bad[:8], lim[bad[:8]])
            assert np.all(tri[~blk] == -1)
            assert np.all(tri[blk & np.isfinite(lim)] >= 0)
    finally:
        r.free()


# Next-event samples of full C5 frames (seeds 4 and 5) whose blocker CRT_TRAVERSAL_FAST loses: the ray lies in the plane of a small
# triangle of a light sphere (cos = 5e-6 / 3e-6), whose Moeller-Trumbore distance lands 0.045 / 0.035 in front of the triangle's own
# leaf box -- 3.5 / 7.7 times the pruning slack.  No slack factor removes such events (tools/margin_hist.py: the number of answers
# lying more than s x reach x steep in front of their box falls only as s^-0.7), so the contract of CRT_TRAVERSAL_FAST is a measured
# rate (docs/experiments.md 4.3: 2 such rays in 3.7e11 on this scene), CRT_TRAVERSAL_REFERENCE is the exact mode, and these two rays
# are kept as known answers: REFERENCE must block them; FAST is reported (xfail) when it does not.
# (origin, Ray direction, t_to_light) as float bits; found by tools/soak_fast_vs_reference.py, isolated by tools/diff_fast_reference.py.
_LOST_VISIBILITY_RAYS = [
    ((3231711232, 1097699506, 3241011755, 1055694332, 3206979767, 1058680934, 1094397382), 675),
    ((1066666608, 1080233611, 3217543700, 3194614196, 1059459322, 1060755361, 1083871202), 1908),
]


@pytest.mark.parametrize("bits,blocker", _LOST_VISIBILITY_RAYS)
def test_the_c5_visibility_rays_that_pruning_loses(bits, blocker):
    name = "veach-mis"
    t = util.task(name)
    osc = util.oracle_scene(name)
    r = crt.Render(util.host_scene(name), 1, t.P_RR, t.light_sample_n)
    try:
        f = np.array(bits, dtype=np.uint32).view(np.float32)
        o, d, lim = f[None, 0:3].copy(), f[None, 3:6].copy(), f[6:7].copy()
        RAW = crt.INTERSECT_RAW_DIRECTIONS
        tri, tt = r.intersect(o, d, traversal=crt.TRAVERSAL_REFERENCE | RAW)
        assert tri[0] == blocker and lim[0] - tt[0] > np.float32(0.00001)
        blk, btri = r.blocked(o, d, lim, traversal=crt.TRAVERSAL_REFERENCE | RAW)
        assert blk[0] and btri[0] == blocker
        # CRT_TRAVERSAL_EXACT -- the same traversal without the pruning rule -- must block it too, on both of its arithmetic paths
        for mode in (crt.TRAVERSAL_EXACT, crt.TRAVERSAL_EXACT | crt.INTERSECT_FORCE_EXACT):
            blk, btri = r.blocked(o, d, lim, traversal=mode | RAW)
            assert blk[0] and btri[0] >= 0, mode
        # as a closest-hit query FAST finds the triangle (nothing nearer sets a bound before its box is reached)
        tri_f, tt_f = r.intersect(o, d, traversal=crt.TRAVERSAL_FAST | RAW)
        assert tri_f[0] == blocker and util.bits(tt_f)[0] == util.bits(tt)[0]
        blk, btri = r.blocked(o, d, lim, traversal=crt.TRAVERSAL_FAST | RAW)
        if not blk[0]:
            pytest.xfail("documented: the blocker's box is entered beyond prune_bound(t_to_light) (DESIGN.md section 4; docs/experiments.md 4.3)")
        assert btri[0] == blocker
    finally:
        r.free()


def test_raw_directions_with_infinite_components():
    """A direction with an infinite component has 1/d = 0 (finite!): (plane - o) * 0 is 0, or NaN once plane - o overflows.  Such rays
    must take the reference-arithmetic path (start_ray's predicate asks for a finite direction as well), so FAST and REFERENCE agree
    on them bit for bit.  Only reachable through CRT_INTERSECT_RAW_DIRECTIONS: a Ray's own direction is normalised."""
    name = "cornell-box"
    t = util.task(name)
    r = crt.Render(util.host_scene(name), 1, t.P_RR, t.light_sample_n)
    try:
        rng = np.random.RandomState(3)
        n = 4096
        o = (rng.rand(n, 3).astype(np.float32) - 0.5) * 600 + np.array([278, 273, 0], dtype=np.float32)
        o[::7] *= np.float32(1e35)  # far out: plane - o is of the order of FLT_MAX / 10
        d = rng.randn(n, 3).astype(np.float32)
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        k = rng.randint(0, 3, n)
        d[np.arange(n), k] = np.where(rng.rand(n) < 0.5, np.inf, -np.inf).astype(np.float32)
        d[::5, (k[::5] + 1) % 3] = np.inf
        RAW = crt.INTERSECT_RAW_DIRECTIONS
        tri_r, t_r = r.intersect(o, d, traversal=crt.TRAVERSAL_REFERENCE | RAW)
        for mode in (crt.TRAVERSAL_FAST, crt.TRAVERSAL_EXACT):
            tri_f, t_f = r.intersect(o, d, traversal=mode | RAW)
            assert np.array_equal(tri_r, tri_f) and np.array_equal(util.bits(t_r), util.bits(t_f)), mode
    finally:
        r.free()


LAYOUTS = {  # environment of each form of k_mega3's pool (csrc/crt_render.hip: use_dec, use_ref16, use_impl)
    "coupled-16": {"CRT_DEC": "0"},
    "coupled-32": {"CRT_DEC": "0", "CRT_REF16": "0"},
    "decoupled-16": {"CRT_DEC": "1"},                       # (on the tree without its rows of refs where the scene offers it: both shipped scenes do)
    "decoupled-16, rows of refs": {"CRT_DEC": "1", "CRT_IMPL": "0"},
    "decoupled-32": {"CRT_DEC": "1", "CRT_REF32": "1"},
    "default": {},  # decoupled-16 for CRT_TRAVERSAL_EXACT on a scene of up to about 160 000 triangles, coupled-16 for the other modes
    "default, leaf records beyond 16 bits": {"CRT_REF16": "0"},  # a scene of 50 000 - 160 000 triangles: coupled-32 for the other modes
}


@pytest.mark.parametrize("name", ["cornell-box", "veach-mis"])
def test_all_stack_layouts(name, monkeypatch):
    """k_mega3's pool has two forms.  Coupled: a ray walks inner nodes and leaves alike, eight 16-bit traversal-stack levels in LDS
    where node AND leaf refs fit 16 bits (both benchmark scenes), four 32-bit levels otherwise.  Decoupled leaves (CRT_TRAVERSAL_EXACT):
    the stack holds inner nodes only -- six 16-bit levels while the four-wide nodes number at most 32 768, else three of 32 bits --
    and the leaf tests are entries of a queue.  In a 16-bit layout a ray on the reference-arithmetic path keeps its whole stack in the
    global spill area (CRT_FLAG_FORCE_EXACT puts every ray there).  Same frames, same ray counts, same closest hits in every form."""
    t = util.task(name)
    eye, iv, fov = util.camera(name)
    osc = util.oracle_scene(name)
    r = crt.Render(util.host_scene(name), 3, t.P_RR, t.light_sample_n)
    try:
        orgb, omean, _, st = osc.render(eye, iv, fov, 96, 72, 3, t.P_RR, t.light_sample_n)
        o, d = util.random_rays(name, 4096, seed=17)
        otri, ot, _ = osc.intersect(o, d)
        for layout, env in LAYOUTS.items():
            for k in ("CRT_DEC", "CRT_REF16", "CRT_REF32", "CRT_IMPL"):
                monkeypatch.delenv(k, raising=False)
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            for mode, flags in ((crt.TRAVERSAL_FAST, 0), (crt.TRAVERSAL_EXACT, 0), (crt.TRAVERSAL_FAST, crt.FLAG_FORCE_EXACT),
                                (crt.TRAVERSAL_EXACT, crt.FLAG_FORCE_EXACT), (crt.TRAVERSAL_REFERENCE, 0)):
                r.traversal, r.extra_flags = mode, flags
                rgb = r.run_view(eye, iv, fov, width=96, height=72)
                assert np.array_equal(rgb, orgb) and np.array_equal(util.bits(r.mean_buffer), util.bits(omean)), (layout, mode, flags)
                assert r.stats["rays"] == st["rays"]
            r.extra_flags = 0
            for mode in (crt.TRAVERSAL_FAST, crt.TRAVERSAL_EXACT, crt.TRAVERSAL_FAST | crt.INTERSECT_FORCE_EXACT,
                         crt.TRAVERSAL_EXACT | crt.INTERSECT_FORCE_EXACT, crt.TRAVERSAL_REFERENCE):
                tri, tt = r.intersect(o, d, traversal=mode)
                assert np.array_equal(tri, otri) and np.array_equal(util.bits(tt), util.bits(ot)), (layout, mode)
    finally:
        r.free()

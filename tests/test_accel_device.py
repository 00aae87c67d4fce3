"""GPU test of the device builder of the FAST traversal's SAH tree (csrc/crt_accel_build.hip against the host builder of
csrc/crt_accel.h).  Any tree over the reference's leaves gives the same hits (crt_accel.h), so frames cannot tell the two
builders apart; the traversal's VISIT COUNTERS can: identical trees visit identical node / leaf / triangle counts on a frame.
On scenes without coinciding leaf centroids the device tree must be the host tree (same node counts, depths, counters); with
duplicates the builders may hand equal leaves to different sides -- frames still equal the oracle's."""
import json
import os

import numpy as np
import pytest

import cudaraytracing_amd as crt
import oracle_lib as O
import util
from test_gpu_parity import _write_box_scene, _write_soup_scene

pytestmark = pytest.mark.gpu
REPORT = os.path.join(util.ROOT, "gpurun_out", "sah_build_report.jsonl")


def _pair(scene, spp, p_rr, lsn, monkeypatch):
    monkeypatch.delenv("CRT_SAH_HOST", raising=False)
    dev = crt.Render(scene, spp, p_rr, lsn)
    dev2 = crt.Render(scene, spp, p_rr, lsn)   # second build: code objects loaded, timing meaningful
    monkeypatch.setenv("CRT_SAH_HOST", "1")
    host = crt.Render(scene, spp, p_rr, lsn)
    monkeypatch.delenv("CRT_SAH_HOST", raising=False)
    dev.free()
    return dev2, host


def _report(name, a, b):
    os.makedirs(os.path.dirname(REPORT), exist_ok=True)
    with open(REPORT, "a") as f:
        f.write(json.dumps({"scene": name, "leaves": a["n_leaves"], "nodes2": a["n_nodes2"], "nodes4": a["n_nodes4"], "depth4": a["depth4"], "index_splits": a["index_splits"],
                            "device_sah_ms": round(a["sah_ms"], 3), "device_level_loop_ms": round(a["sah_device_ms"], 3),
                            "host_sah_ms": round(b["sah_ms"], 3)}) + "\n")


@pytest.mark.parametrize("name", ["cornell-box", "veach-mis"])
def test_device_tree_is_the_host_tree_on_the_benchmark_scenes(name, monkeypatch):
    t = util.task(name)
    eye, iv, fov = util.camera(name)
    dev, host = _pair(util.host_scene(name), 2, t.P_RR, t.light_sample_n, monkeypatch)
    try:
        a, b = dev.accel_info(), host.accel_info()
        _report(name, a, b)
        assert a["sah_on_device"] == 1 and b["sah_on_device"] == 0
        for k in ("n_leaves", "n_nodes2", "n_nodes4", "depth2", "depth4"):
            assert a[k] == b[k], k
        ra = dev.run_view(eye, iv, fov, stats=True, width=160, height=120).copy()
        rb = host.run_view(eye, iv, fov, stats=True, width=160, height=120)
        assert np.array_equal(ra, rb) and np.array_equal(util.bits(dev.mean_buffer), util.bits(host.mean_buffer))
        assert a["index_splits"] == b["index_splits"]
        if a["index_splits"] == 0:  # no range of coinciding leaf centroids: the device tree IS the host tree
            for k in ("rays", "inner_pops", "leaf_pops", "tri_tests", "hits", "stack_max"):
                assert dev.stats[k] == host.stats[k], k
        else:                       # (veach-mis holds coincident triangles: the equal leaves may sit on different sides)
            assert dev.stats["rays"] == host.stats["rays"] and dev.stats["hits"] == host.stats["hits"]
            assert abs(dev.stats["leaf_pops"] - host.stats["leaf_pops"]) < 0.01 * host.stats["leaf_pops"]
    finally:
        dev.free()
        host.free()


def test_room_of_180000_triangles(tmp_path, monkeypatch):
    obj, mtl = _write_box_scene(str(tmp_path), n_side=300)
    scene = crt.Scene(40, 30)
    scene.add_obj(obj, mtl)
    scene.set_BVH(2)
    eye = np.array([5.0, 5.0, 0.5], dtype=np.float32)
    iv = crt.get_inverse_view_matrix(eye, [5.0, 4.0, 9.0], [0.0, 1.0, 0.0])
    fov = crt.fov_to_radians(70.0)
    dev, host = _pair(scene, 1, 0.6, 1, monkeypatch)
    try:
        a, b = dev.accel_info(), host.accel_info()
        _report("room-180000", a, b)
        assert a["sah_on_device"] == 1 and a["n_leaves"] == b["n_leaves"] and a["n_nodes2"] == b["n_nodes2"]
        ra = dev.run_view(eye, iv, fov, stats=True).copy()
        rb = host.run_view(eye, iv, fov, stats=True)
        assert np.array_equal(ra, rb) and np.array_equal(util.bits(dev.mean_buffer), util.bits(host.mean_buffer))
        assert dev.stats["rays"] == host.stats["rays"]
    finally:
        dev.free()
        host.free()


def test_soup_with_duplicates_matches_the_oracle(tmp_path, monkeypatch):
    obj, mtl = _write_soup_scene(str(tmp_path))
    scene = crt.Scene(64, 48)
    scene.add_obj(obj, mtl)
    scene.set_BVH(2)
    osc = O.OracleScene([(obj, mtl)], 2)
    eye = np.array([5.0, 5.0, 0.5], dtype=np.float32)
    iv = crt.get_inverse_view_matrix(eye, [5.0, 4.5, 9.0], [0.0, 1.0, 0.0])
    fov = crt.fov_to_radians(75.0)
    dev, host = _pair(scene, 2, 0.6, 2, monkeypatch)
    try:
        _report("soup", dev.accel_info(), host.accel_info())
        orgb, omean, _, st = osc.render(eye, iv, fov, 64, 48, 2, 0.6, 2, seed=0)
        for r in (dev, host):
            rgb = r.run_view(eye, iv, fov)
            assert np.array_equal(util.bits(r.mean_buffer), util.bits(omean)) and np.array_equal(rgb, orgb) and r.stats["rays"] == st["rays"]
    finally:
        dev.free()
        host.free()


@pytest.mark.parametrize("name", ["cornell-box", "veach-mis"])
def test_collapse_by_dynamic_programming_against_the_greedy_one(name, monkeypatch):
    """The 4-wide collapse minimises the summed area of the wide nodes (dynamic programming, csrc/crt_render.hip); the round-1 rule
    (CRT_COLLAPSE=greedy: open the largest child until there are four) stays selectable.  Same frame either way (any tree over the
    reference's leaves gives the same hits); the optimal collapse has no more nodes and no more inner steps on the frame."""
    t = util.task(name)
    eye, iv, fov = util.camera(name)
    osc = util.oracle_scene(name)
    monkeypatch.delenv("CRT_COLLAPSE", raising=False)
    dp = crt.Render(util.host_scene(name), 2, t.P_RR, t.light_sample_n)
    monkeypatch.setenv("CRT_COLLAPSE", "greedy")
    gr = crt.Render(util.host_scene(name), 2, t.P_RR, t.light_sample_n)
    monkeypatch.delenv("CRT_COLLAPSE", raising=False)
    try:
        a, b = dp.accel_info(), gr.accel_info()
        assert a["n_leaves"] == b["n_leaves"] and a["n_nodes2"] == b["n_nodes2"] and a["n_nodes4"] <= b["n_nodes4"]
        ra = dp.run_view(eye, iv, fov, stats=True, width=160, height=120).copy()
        ma = dp.mean_buffer.copy()
        rb = gr.run_view(eye, iv, fov, stats=True, width=160, height=120)
        assert np.array_equal(ra, rb) and np.array_equal(util.bits(ma), util.bits(gr.mean_buffer))
        orgb, omean, _, st = osc.render(eye, iv, fov, 160, 120, 2, t.P_RR, t.light_sample_n)
        assert np.array_equal(ra, orgb) and np.array_equal(util.bits(ma), util.bits(omean))
        assert dp.stats["rays"] == gr.stats["rays"] == st["rays"]
        assert dp.stats["inner_pops"] <= gr.stats["inner_pops"]
        # every leaf is reachable in both trees: the closest hits of random rays are the oracle's
        o, d = util.random_rays(name, 4096, seed=3)
        otri, ot, _ = osc.intersect(o, d)
        for r in (dp, gr):
            tri, tt = r.intersect(o, d, traversal=crt.TRAVERSAL_EXACT)
            assert np.array_equal(tri, otri) and np.array_equal(util.bits(tt), util.bits(ot))
    finally:
        dp.free(); gr.free()


TREE_HOOKS = {  # environment of crt_scene_create's tree set-up (csrc/crt_render.hip, csrc/crt_accel.h)
    "default": {},
    "children of a node as the collapse leaves them": {"CRT_CHILD_ORDER": "none"},
    "children of a node by box area": {"CRT_CHILD_ORDER": "area"},
    "re-insertion pass: serial form": {"CRT_SAH_OPT_FORM": "serial"},
    "re-insertion pass: one thread": {"CRT_SAH_OPT_THREADS": "1"},
    "re-insertion pass: two passes, margin 0.3": {"CRT_SAH_OPT": "2", "CRT_SAH_OPT_MARGIN": "0.3"},
    "no re-insertion pass": {"CRT_SAH_OPT": "0"},
}


@pytest.mark.parametrize("name", ["cornell-box", "veach-mis"])
def test_tree_set_up_hooks_change_the_tree_and_not_the_frame(name, monkeypatch):
    """Round 6: the children of a four-wide node are ordered by occupancy, and the re-insertion pass runs in blocks on several threads
    (tests/test_sah_opt.py checks the pass on the CPU).  Every hook of the set-up gives another tree over the SAME leaves: the frame, the
    ray count and the closest hits of random rays stay the oracle's in both exact-by-construction modes and in FAST; the batched pass
    gives the same tree on one thread as on many (here: the same number of four-wide nodes and the same depths)."""
    t = util.task(name)
    eye, iv, fov = util.camera(name)
    osc = util.oracle_scene(name)
    orgb, omean, _, st = osc.render(eye, iv, fov, 96, 72, 2, t.P_RR, t.light_sample_n)
    o, d = util.random_rays(name, 2048, seed=11)
    otri, ot, _ = osc.intersect(o, d)
    seen = {}
    for hook, env in TREE_HOOKS.items():
        for k in ("CRT_CHILD_ORDER", "CRT_SAH_OPT_FORM", "CRT_SAH_OPT_THREADS", "CRT_SAH_OPT", "CRT_SAH_OPT_MARGIN"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        r = crt.Render(util.host_scene(name), 2, t.P_RR, t.light_sample_n)
        try:
            for mode in (crt.TRAVERSAL_EXACT, crt.TRAVERSAL_FAST):
                r.traversal = mode
                rgb = r.run_view(eye, iv, fov, stats=True, width=96, height=72)
                assert np.array_equal(rgb, orgb) and np.array_equal(util.bits(r.mean_buffer), util.bits(omean)), (hook, mode)
                assert r.stats["rays"] == st["rays"], (hook, mode)
                if mode == crt.TRAVERSAL_EXACT:
                    a = r.accel_info()
                    seen[hook] = (a["n_nodes4"], a["depth2"], a["depth4"])  # (visit counts of any-hit rays depend on the waves' timing: not compared)
            tri, tt = r.intersect(o, d, traversal=crt.TRAVERSAL_EXACT)
            assert np.array_equal(tri, otri) and np.array_equal(util.bits(tt), util.bits(ot)), hook
        finally:
            r.free()
    assert seen["default"] == seen["re-insertion pass: one thread"], seen

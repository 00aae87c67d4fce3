"""GPU test of the device BVH builder (reference: include/BVH.h:37-84; csrc/crt_bvh_build.hip).

The level-synchronous device build must give node and triangle arrays BYTE-IDENTICAL to the host builder's (which the CPU
tests compare with the oracle's restatement of BVH.h): both shipped scenes, the 180 000-triangle room (whole grid columns with
equal centroid coordinates), the soup with exact duplicates, other leaf sizes, adversarial key patterns, and a frame rendered
from the device-built tree.  Equal sort keys are where std::sort's order is its own: the device replays libstdc++'s introsort."""
import json
import os
import time

import numpy as np
import pytest

import cudaraytracing_amd as crt
import util
from test_gpu_parity import _write_box_scene, _write_soup_scene

pytestmark = pytest.mark.gpu
REPORT = os.path.join(util.ROOT, "gpurun_out", "bvh_build_report.jsonl")


def _both(add, w, h, thresh):
    host, dev = crt.Scene(w, h), crt.Scene(w, h)
    add(host)
    add(dev)
    t0 = time.perf_counter()
    host.set_BVH(thresh)
    host_ms = (time.perf_counter() - t0) * 1e3
    dev.set_BVH(thresh, device=0)   # first call: HIP context / code object load
    dev2 = crt.Scene(w, h)
    add(dev2)
    dev2.set_BVH(thresh, device=0)
    info = dict(dev2.bvh_build_info)
    info["host_builder_ms"] = round(host_ms, 3)
    assert host.nodes().tobytes() == dev2.nodes().tobytes()
    assert host.triangles().tobytes() == dev2.triangles().tobytes()
    assert host.root == dev2.root == len(host.nodes()) - 1
    assert dev.nodes().tobytes() == host.nodes().tobytes()
    return host, dev2, info


def _report(name, info):
    os.makedirs(os.path.dirname(REPORT), exist_ok=True)
    with open(REPORT, "a") as f:
        f.write(json.dumps(dict(scene=name, **{k: (round(v, 3) if isinstance(v, float) else v) for k, v in info.items()})) + "\n")


@pytest.mark.parametrize("name", ["cornell-box", "veach-mis"])
def test_shipped_scenes(name):
    t = util.task(name)

    def add(s):
        for obj, mtl in t.OBJ_paths:
            s.add_obj(obj, mtl)
    host, dev, info = _both(add, t.width, t.height, t.bvh_thresh_n)
    _report(name, info)
    assert info["n_triangles"] == len(host.triangles()) and info["n_nodes"] == len(host.nodes())
    assert info["host_ranges"] == 0 and info["host_triangles"] == 0  # equal centroid coordinates are replayed on the device
    # a frame from the device-built tree is the frame from the host-built tree
    eye, iv, fov = util.camera(name)
    a = crt.Render(host, 2, t.P_RR, t.light_sample_n)
    b = crt.Render(dev, 2, t.P_RR, t.light_sample_n)
    try:
        ra = a.run_view(eye, iv, fov, width=96, height=72)
        rb = b.run_view(eye, iv, fov, width=96, height=72)
        assert np.array_equal(ra, rb) and np.array_equal(util.bits(a.mean_buffer), util.bits(b.mean_buffer))
    finally:
        a.free()
        b.free()


@pytest.mark.parametrize("thresh", [1, 2, 3, 5, 20, 400])
def test_room_with_other_leaf_sizes(tmp_path, thresh):
    obj, mtl = _write_box_scene(str(tmp_path), n_side=12)
    host, dev, info = _both(lambda s: s.add_obj(obj, mtl), 32, 24, thresh)
    _report("room-300 thresh %d" % thresh, info)


def test_room_of_180000_triangles(tmp_path):
    obj, mtl = _write_box_scene(str(tmp_path), n_side=300)
    host, dev, info = _both(lambda s: s.add_obj(obj, mtl), 32, 24, 2)
    _report("room-180000", info)
    assert info["n_triangles"] == 180012 and info["host_triangles"] == 0
    # (grid columns: sorted keys full of ties send std::sort's quicksort phase into its depth limit on some ranges -- those single
    #  sorts are done by the host between two levels, everything else on the device)
    assert info["host_sort_elements"] < 4 * 180012


@pytest.mark.parametrize("thresh", [1, 2, 4])
def test_soup_with_duplicate_triangles(tmp_path, thresh):
    """Exact duplicates have equal centroids in every axis: std::sort's order of them is replayed on the device."""
    obj, mtl = _write_soup_scene(str(tmp_path))
    host, dev, info = _both(lambda s: s.add_obj(obj, mtl), 32, 24, thresh)
    _report("soup thresh %d" % thresh, info)
    assert info["host_ranges"] == 0


def test_negative_zero_coordinates_build_on_the_host(tmp_path):
    """-0.0 among the coordinates: std::min / std::max keep whichever zero they met last, so the tree is built by the host
    builder (reported as one host range over everything) -- still through the same entry point."""
    obj, mtl = _write_box_scene(str(tmp_path), n_side=4)
    s = open(obj).read().replace("v 0 0 0\n", "v -0.0 0 -0.0\n", 1)
    assert "-0.0" in s
    open(obj, "w").write(s)
    host, dev, info = _both(lambda sc: sc.add_obj(obj, mtl), 32, 24, 2)
    assert info["host_triangles"] == info["n_triangles"] and info["host_ranges"] == 1


def _write_random_soup(d, n, seed=3):
    """n random triangles with random (hence pairwise different) centroid coordinates, one of them emissive: no two sort keys are
    equal anywhere, so the device builds the whole tree."""
    rng = np.random.RandomState(seed)
    c = rng.uniform(0.0, 100.0, (n, 3))
    v = (c[:, None, :] + rng.uniform(-0.5, 0.5, (n, 3, 3))).astype(np.float32)
    for _ in range(20):  # (5 000 random float32 keys already collide with probability 1/2: nudge until every axis is tie-free)
        cen = ((v[:, 0, :] + v[:, 1, :]) + v[:, 2, :]) / np.float32(3)  # Triangle.h:26 in float
        dup = np.zeros(n, dtype=bool)
        for k in range(3):
            _, first, counts = np.unique(cen[:, k], return_index=True, return_counts=True)
            keep = np.zeros(n, dtype=bool)
            keep[first] = True
            dup |= ~keep
        if not dup.any():
            break
        v[dup] += rng.uniform(-0.01, 0.01, (int(dup.sum()), 3, 3)).astype(np.float32)
    assert not dup.any()
    obj = os.path.join(d, "soup.obj")
    with open(os.path.join(d, "soup.mtl"), "w") as m:
        m.write("newmtl grey\nKd 0.6 0.6 0.6\nNs 1\nnewmtl light\nKe 20 20 20\nKd 0 0 0\nNs 1\n")
    with open(obj, "w") as o:
        o.write("mtllib soup.mtl\n")
        for t in v:
            for p in t:
                o.write("v %.9g %.9g %.9g\nvn 0 1 0\nvt 0 0\n" % tuple(p))
        o.write("usemtl grey\n")
        for i in range(n - 1):
            o.write("f %d/%d/%d %d/%d/%d %d/%d/%d\n" % ((3 * i + 1,) * 3 + (3 * i + 2,) * 3 + (3 * i + 3,) * 3))
        o.write("usemtl light\n")
        i = n - 1
        o.write("f %d/%d/%d %d/%d/%d %d/%d/%d\n" % ((3 * i + 1,) * 3 + (3 * i + 2,) * 3 + (3 * i + 3,) * 3))
    return obj, d


@pytest.mark.parametrize("n,thresh", [(5000, 2), (50000, 2), (200000, 2), (50000, 5)])
def test_tie_free_geometry_is_built_entirely_on_the_device(tmp_path, n, thresh):
    obj, mtl = _write_random_soup(str(tmp_path), n)
    host, dev, info = _both(lambda s: s.add_obj(obj, mtl), 32, 24, thresh)
    _report("random soup %d thresh %d" % (n, thresh), info)
    assert info["host_ranges"] == 0 and info["host_triangles"] == 0 and info["n_triangles"] == n


@pytest.mark.parametrize("pattern", ["few_values", "sorted", "reversed", "organ_pipe", "all_equal"])
def test_adversarial_key_patterns(tmp_path, pattern):
    """Triangles laid out so that the centroid keys of the top ranges are (a) drawn from a handful of values, (b) already sorted,
    (c) reversed, (d) organ-pipe, (e) all equal along the sort axis: the quicksort replay must follow std::sort through every one
    (median-of-three choices, equal keys stopping both partition pointers), or hand the range to the host when the depth limit
    sends std::sort into heapsort."""
    n = 6000
    rng = np.random.RandomState(5)
    i = np.arange(n)
    if pattern == "few_values":
        x = rng.randint(0, 5, n).astype(np.float64)
    elif pattern == "sorted":
        x = i * 0.01
    elif pattern == "reversed":
        x = (n - i) * 0.01
    elif pattern == "organ_pipe":
        x = np.minimum(i, n - i) * 0.01
    else:
        x = np.zeros(n)
    c = np.stack([x * 10.0, rng.randint(0, 3, n) * 0.5, rng.randint(0, 7, n) * 0.25], axis=1)   # x extent dominates (or y / z when x is flat)
    v = (c[:, None, :] + np.array([[0, 0, 0], [0.03, 0, 0.01], [0, 0.03, 0.02]])[None]).astype(np.float32)
    d = str(tmp_path)
    with open(os.path.join(d, "p.mtl"), "w") as m:
        m.write("newmtl grey\nKd 0.6 0.6 0.6\nNs 1\nnewmtl light\nKe 20 20 20\nKd 0 0 0\nNs 1\n")
    with open(os.path.join(d, "p.obj"), "w") as o:
        o.write("mtllib p.mtl\n")
        for t in v:
            for q in t:
                o.write("v %.9g %.9g %.9g\nvn 0 1 0\nvt 0 0\n" % tuple(q))
        o.write("usemtl grey\n")
        for k in range(n - 1):
            o.write("f %d/%d/%d %d/%d/%d %d/%d/%d\n" % ((3 * k + 1,) * 3 + (3 * k + 2,) * 3 + (3 * k + 3,) * 3))
        o.write("usemtl light\n")
        k = n - 1
        o.write("f %d/%d/%d %d/%d/%d %d/%d/%d\n" % ((3 * k + 1,) * 3 + (3 * k + 2,) * 3 + (3 * k + 3,) * 3))
    obj = os.path.join(d, "p.obj")
    for thresh in (1, 2, 7):
        host, dev, info = _both(lambda s: s.add_obj(obj, d), 32, 24, thresh)
        _report("pattern %s thresh %d" % (pattern, thresh), info)

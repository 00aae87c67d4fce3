"""An anchor for the LOADER that shares no code with it.  The product loader (csrc/crt_host.cpp: Loader::read_OBJ / load_object, after
include/OBJLoader.h:61-203 and Loader.h:40-124) and the oracle's (oracle/crt_oracle.cpp) were written by the same hand from the same
reading of the reference, so their byte-equality (tests/test_host_layer.py) cannot catch a shared misreading.  Here the shipped OBJ /
MTL files are parsed by twenty lines of Python that know nothing but the file formats ("v x y z", "f a/b/c ...", "usemtl", "Kd", "Ke"),
and what the product scene holds is compared with it: the multiset of triangles (float32 vertex triples, exactly), which shape is a light
and which is not (Ke != 0, Material.h:33-40), the material of every triangle, the per-object areas the reference prints (Object.h:25,
SURVEY 3.4), and the geometric normal cross(e1, e2) (Triangle.h:27).  CPU only."""
import os

import numpy as np
import pytest

import cudaraytracing_amd as crt
import util


def _parse(obj_path, mtl_dir):
    mats, cur = {}, None
    mtl = None
    verts, shapes = [], []            # shapes: (material name, [(i, j, k) 0-based vertex indices])
    for line in open(obj_path):
        t = line.split()
        if not t:
            continue
        if t[0] == "mtllib":
            mtl = os.path.join(mtl_dir, t[1])
        elif t[0] == "v":
            verts.append([np.float32(x) for x in t[1:4]])
        elif t[0] == "usemtl":
            shapes.append((t[1], []))
        elif t[0] == "f":
            idx = [int(p.split("/")[0]) - 1 for p in t[1:]]
            assert len(idx) == 3, "the shipped scenes are triangulated"
            shapes[-1][1].append(tuple(idx))
    for line in open(mtl):
        t = line.split()
        if not t:
            continue
        if t[0] == "newmtl":
            cur = t[1]
            mats[cur] = {"Kd": (0.0, 0.0, 0.0), "Ke": (0.0, 0.0, 0.0)}
        elif t[0] in ("Kd", "Ke") and cur:
            mats[cur][t[0]] = tuple(float(x) for x in t[1:4])
    return np.array(verts, dtype=np.float32), [s for s in shapes if s[1]], mats


def _key(v1, v2, v3):
    return np.ascontiguousarray(np.concatenate([v1, v2, v3], axis=-1), dtype=np.float32).view(np.uint32)


@pytest.mark.parametrize("name", ["veach-mis", "cornell-box"])
def test_the_scene_is_the_files_triangles(name):
    t = util.task(name)
    assert len(t.OBJ_paths) == 1
    obj, mtl_dir = t.OBJ_paths[0]
    verts, shapes, mats = _parse(obj, mtl_dir)
    scene = crt.Scene.from_task(t)
    try:
        tris = scene.triangles()           # non-light triangles, BVH order
        ltris = scene.light_triangles()    # light triangles, shape order
        pm = scene.materials()
        # ---- which shapes are lights: Ke != 0 ----
        is_light = [any(c != 0.0 for c in mats[m]["Ke"]) for m, _ in shapes]
        want_n = sum(len(f) for (m, f), l in zip(shapes, is_light) if not l)
        want_l = sum(len(f) for (m, f), l in zip(shapes, is_light) if l)
        assert want_n + want_l == sum(len(f) for _, f in shapes)
        # the BVH holds every triangle a ray can hit -- the lights' too (Scene.h:44-48 adds light objects to the triangle list as well)
        assert len(ltris) == want_l and len(tris) in (want_n, want_n + want_l)
        lights_in_bvh = len(tris) == want_n + want_l
        # ---- the multiset of triangles, vertex for vertex, bit for bit ----
        want = []
        for (m, faces), l in zip(shapes, is_light):
            if l and not lights_in_bvh:
                continue
            f = np.array(faces)
            want.append(_key(verts[f[:, 0]], verts[f[:, 1]], verts[f[:, 2]]))
        want = np.concatenate(want)
        got = _key(tris["v1"], tris["v2"], tris["v3"])

        def rows(a):
            return np.sort(np.ascontiguousarray(a).view([("", a.dtype)] * a.shape[1]).ravel())
        assert np.array_equal(rows(want), rows(got))
        # light triangles keep the file's order (DeviceLights samples them by index, DeviceLights.cuh:33-37)
        lf = np.concatenate([np.array(f) for (m, f), l in zip(shapes, is_light) if l])
        assert np.array_equal(_key(verts[lf[:, 0]], verts[lf[:, 1]], verts[lf[:, 2]]), _key(ltris["v1"], ltris["v2"], ltris["v3"]))
        # ---- material of every triangle: kd / ke of the shape it came from ----
        by_vertices = {}
        for (m, faces), l in zip(shapes, is_light):
            f = np.array(faces)
            for k in _key(verts[f[:, 0]], verts[f[:, 1]], verts[f[:, 2]]):
                by_vertices.setdefault(k.tobytes(), set()).add(m)
        for k, mi in zip(got, tris["material"]):
            names = by_vertices[k.tobytes()]
            assert any(np.allclose(pm[mi]["kd"], mats[n]["Kd"], rtol=0, atol=1e-7) and np.allclose(pm[mi]["ke"], mats[n]["Ke"], rtol=0, atol=1e-6)
                       for n in names), (names, pm[mi])
        # ---- geometric normal and edges as the reference computes them (Triangle.h:25-28) ----
        e1, e2 = tris["v2"] - tris["v1"], tris["v3"] - tris["v1"]
        n = np.cross(e1.astype(np.float64), e2.astype(np.float64))
        ln = np.linalg.norm(n, axis=1)
        ok = ln > 0
        assert np.abs(tris["normal"][ok] - (n[ok] / ln[ok, None])).max() < 2e-6
        # ---- per-object areas (Object.h:20-25: the sum of the triangles' areas), objects in shape order ----
        objs = scene.objects()
        assert len(objs) == len(shapes)
        for (light, area, count), (m, faces), l in zip(objs, shapes, is_light):
            f = np.array(faces)
            a = 0.5 * np.linalg.norm(np.cross((verts[f[:, 1]] - verts[f[:, 0]]).astype(np.float64),
                                              (verts[f[:, 2]] - verts[f[:, 0]]).astype(np.float64)), axis=1)
            assert light == l and count == len(faces)
            assert abs(area - a.sum()) <= 2e-5 * a.sum(), (m, area, a.sum())
    finally:
        scene.free()

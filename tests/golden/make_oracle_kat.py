#!/usr/bin/env python3
"""Writes tests/golden/oracle_kat.json: known answers produced by the CPU oracle itself
(loader/BVH hashes, intersect / per-path / image KATs, traversal counters).  They pin the
oracle against accidental change; they are NOT reference outputs (the reference cannot be
built here, see oracle/crt_oracle.h)."""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import util  # noqa: E402


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


out = {}
for name in ("cornell-box", "veach-mis"):
    sc = util.oracle_scene(name)
    t = util.task(name)
    eye, iv, fov = util.camera(name)
    nodes, tris = sc.nodes(), sc.tris()
    e = {"n_tris": int(sc.num_tris), "n_nodes": int(sc.num_nodes), "root": int(sc.root),
         "lights": [int(sc.light_size(i)) for i in range(sc.num_lights)],
         "object_areas_bits": [int(np.float32(a).view(np.uint32)) for _, a in sc.objects()],
         "nodes_sha256": sha(nodes), "centroid_order_sha256": sha((tris["v1"] + tris["v2"] + tris["v3"]))}
    o, d = util.random_rays(name, 4096, seed=11)
    tri, tt, st = sc.intersect(o, d)
    e["intersect"] = {"tri_sha256": sha(tri), "t_sha256": sha(tt), "hits": int((tri >= 0).sum()),
                      "counters": {k: st[k] for k in ("inner_pops", "leaf_pops", "tri_tests", "hits")}}
    rgb, mean, L, st = sc.render(eye, iv, fov, t.width, t.height, 4, t.P_RR, t.light_sample_n, seed=0,
                                 crop=(368, 268, 64, 48), want_L=True)
    e["image_crop_368_268_64x48_spp4"] = {"rgb_sha256": sha(rgb), "mean_sha256": sha(mean), "L_sha256": sha(L),
                                         "counters": {k: st[k] for k in st},
                                         "mean_sample_bits": [int(x) for x in mean[::16, ::16].reshape(-1).view(np.uint32)[:24]]}
    out[name] = e
with open(os.path.join(HERE, "oracle_kat.json"), "w") as f:
    json.dump(out, f, indent=1, sort_keys=True)
print("wrote oracle_kat.json")

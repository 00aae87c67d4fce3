#!/usr/bin/env python3
"""A mesh beyond 32 768 four-wide nodes: cornell-box --detail L1,L2 rendered by the pool forms that can take it --
coupled with 32-bit stack entries (the default there until round 5), decoupled leaves with 32-bit entries (three LDS levels) -- kernel ms
per form, and each form's frame against CRT_TRAVERSAL_REFERENCE at a low spp.   tools/big_mesh_probe.py [--detail 7,5] [--spp 64]"""
import argparse, json, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scenes"))
import numpy as np
import cudaraytracing_amd as crt
import gen_cornell_box

ap = argparse.ArgumentParser()
ap.add_argument("--detail", default="7,5")
ap.add_argument("--spp", type=int, default=64)
a = ap.parse_args()
t = crt.Task(os.path.join(ROOT, "scenes", "cornell-box", "config.json"), base_dir=ROOT)
iv = crt.get_inverse_view_matrix(t.eye_pos, t.lookat, t.up); fov = crt.fov_to_radians(t.fov_y)
with tempfile.TemporaryDirectory() as td:
    obj, mtl, n_tri = gen_cornell_box.write_variant(td, gen_cornell_box.parse_detail(a.detail))
    sc = crt.Scene(800, 600); sc.add_obj(obj, mtl); sc.set_BVH(t.bvh_thresh_n)
    r = crt.Render(sc, a.spp, t.P_RR, t.light_sample_n)
    ai = r.accel_info()
    print(json.dumps({"triangles": n_tri, "leaves": ai["n_leaves"], "nodes4": ai["n_nodes4"], "layout_caps": ai["layout_caps"]}))
    forms = {"default": {}, "coupled-32": {"CRT_DEC": "0"}, "decoupled-32": {"CRT_DEC": "1"}}
    r.traversal = crt.TRAVERSAL_REFERENCE
    r.set_spp(2)
    ref = r.run_view(t.eye_pos, iv, fov).copy(); refm = r.mean_buffer.copy()
    for name, env in forms.items():
        for k in ("CRT_DEC", "CRT_REF16", "CRT_REF32", "CRT_IMPL"):
            os.environ.pop(k, None)
        os.environ.update(env)
        r.traversal = crt.TRAVERSAL_EXACT
        r.set_spp(2)
        rgb = r.run_view(t.eye_pos, iv, fov)
        same = bool(np.array_equal(rgb, ref) and np.array_equal(r.mean_buffer.view(np.uint32), refm.view(np.uint32)))
        r.set_spp(a.spp)
        ks = []
        for i in range(3):
            r.run_view(t.eye_pos, iv, fov, want_mean=False)
            ks.append(r.stats["kernel_ms"])
        print(json.dumps({"form": name, "spp": a.spp, "kernel_ms": round(min(ks[1:]), 3), "frame_equals_reference_spp2": same, "rays": r.stats["rays"]}))
    r.free()

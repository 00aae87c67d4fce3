#!/usr/bin/env python3
"""Soak test of CRT_TRAVERSAL_FAST (or, --mode exact, of CRT_TRAVERSAL_EXACT) against the exhaustive CRT_TRAVERSAL_REFERENCE: renders the
shards of a frame in both modes and counts the pixel slots whose float mean differs.  One line of JSON per shard."""
import argparse, ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import cudaraytracing_amd as crt
from cudaraytracing_amd import _capi as capi

ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="veach-mis")
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--spp", type=int, default=4096)
ap.add_argument("--ranks", type=int, default=8)
ap.add_argument("--seeds", type=int, nargs="+", default=[0])
ap.add_argument("--out", default=None)
ap.add_argument("--mode", default="fast", choices=["fast", "exact"])
ap.add_argument("--only", type=int, nargs="*", default=None, help="ranks to render (default: all)")
ap.add_argument("--forms", default="default", help="comma list of pool forms the tested mode is rendered with: default, coupled (CRT_DEC=0), decoupled (CRT_DEC=1), refs (decoupled on the tree WITH its rows of refs: CRT_DEC=1 CRT_IMPL=0)")
a = ap.parse_args()
t = crt.Task(os.path.join(ROOT, "scenes", a.scene, "config.json"), base_dir=ROOT)
sc = crt.Scene.from_task(t, a.width, a.height)
r = crt.Render(sc, a.spp, t.P_RR, t.light_sample_n)
iv = crt.get_inverse_view_matrix(t.eye_pos, t.lookat, t.up)
fov = crt.fov_to_radians(t.fov_y)
cam = r._cam(t.eye_pos, iv, fov)
total_rays = 0
total_bad = 0
for seed in a.seeds:
    r.seed = seed
    for rank in (a.only if a.only else range(a.ranks)):
        slots = crt.shard_slots(a.width, a.height, rank, a.ranks)
        res = {}
        forms = [f for f in a.forms.split(",") if f]
        runs = [("fast" if f == forms[0] else "fast_" + f, crt.TRAVERSAL_EXACT if a.mode == "exact" else crt.TRAVERSAL_FAST, f) for f in forms]
        for name, trav, form in runs + [("ref", crt.TRAVERSAL_REFERENCE, "default")]:
            os.environ.pop("CRT_DEC", None)
            os.environ.pop("CRT_IMPL", None)
            if form == "coupled":
                os.environ["CRT_DEC"] = "0"
            elif form == "decoupled":
                os.environ["CRT_DEC"] = "1"
            elif form == "refs":
                os.environ["CRT_DEC"] = "1"; os.environ["CRT_IMPL"] = "0"
            r.traversal = trav
            buf = np.zeros((slots, 3), dtype=np.uint8)
            mean = np.zeros((slots, 3), dtype=np.float32)
            prm = r._params(rank=rank, world=a.ranks, flags=capi.FLAG_TILED_OUTPUT, width=a.width, height=a.height)
            st = capi.Stats()
            t0 = time.perf_counter()
            capi.check(capi.lib().crt_render(r._h, C.byref(cam), C.byref(prm), capi.ptr(buf), capi.ptr(mean), C.byref(st)), "crt_render")
            res[name] = (mean, st.rays, time.perf_counter() - t0)
        bad = sum(int(np.count_nonzero(np.any(res[n][0].view(np.uint32) != res["ref"][0].view(np.uint32), axis=1))) for n in res if n != "ref")
        total_rays += res["fast"][1]
        total_bad += bad
        line = {"mode": a.mode, "scene": a.scene, "size": [a.width, a.height, a.spp], "seed": seed, "rank": rank, "of": a.ranks, "rays": int(res["fast"][1]),
                "rays_equal": all(res[n][1] == res["ref"][1] for n in res), "slots_differ": bad, "forms": forms,
                "form_s": {n: round(res[n][2], 2) for n in res}, "fast_s": round(res["fast"][2], 2), "ref_s": round(res["ref"][2], 2)}
        print(json.dumps(line), flush=True)
        if a.out:
            with open(a.out, "a") as f:
                f.write(json.dumps(line) + "\n")
print(json.dumps({"total_rays": int(total_rays), "slots_differ": total_bad}))

#!/usr/bin/env python3
"""The fixed cost of a launch: rank 0's share of a frame at several sample counts on ONE GPU; kernel time = fixed + per-sample * spp.
usage: tail_fit.py [--world 8]   (env, e.g. CRT_ITEM_ORDER=0 or CRT_ORDER_WINDOW=..., is passed through)"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import cudaraytracing_amd as crt

ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="cornell-box")
ap.add_argument("--width", type=int, default=800)
ap.add_argument("--height", type=int, default=600)
ap.add_argument("--world", type=int, default=8)
ap.add_argument("--spps", default="16,32,64,128,256,512")
a = ap.parse_args()
t = crt.Task(os.path.join(ROOT, "scenes", a.scene, "config.json"), base_dir=ROOT)
sc = crt.Scene.from_task(t, a.width, a.height)
iv = crt.get_inverse_view_matrix(t.eye_pos, t.lookat, t.up)
fov = crt.fov_to_radians(t.fov_y)
dev = torch.device("cuda:0")
slots = crt.shard_slots(a.width, a.height, 0, a.world)
local = torch.empty((slots, 3), dtype=torch.uint8, device=dev)
xs, ys = [], []
for spp in [int(x) for x in a.spps.split(",")]:
    r = crt.Render(sc, spp, t.P_RR, t.light_sample_n)
    best = 1e9
    for rep in range(4):
        st = r.run_view_device(t.eye_pos, iv, fov, local.data_ptr(), None, None, rank=0, world=a.world, tiled=True, want_stats=True,
                               width=a.width, height=a.height)
        torch.cuda.synchronize()
        best = min(best, st["kernel_ms"])
    r.free()
    xs.append(spp); ys.append(best)
    print(json.dumps({"world": a.world, "spp": spp, "kernel_ms": round(best, 3)}), flush=True)
k, c = np.polyfit(np.array(xs[2:], dtype=float), np.array(ys[2:]), 1)
print(json.dumps({"fit_over": xs[2:], "fixed_ms": round(float(c), 3), "ms_per_sample": round(float(k), 5),
                  "env": {k: v for k, v in os.environ.items() if k.startswith("CRT_")}}))

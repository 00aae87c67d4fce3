#!/usr/bin/env python3
"""Batches and rays per phase of the render kernel (counting kernel, CRT_FLAG_STATS): visits per ray, batch fullness, and the frame
time of the plain kernel -- run once per variant (CRT_DEC=0/1 ...)."""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cudaraytracing_amd as crt
ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="cornell-box")
ap.add_argument("--spp", type=int, default=64)
a = ap.parse_args()
t = crt.Task(os.path.join(ROOT, "scenes", a.scene, "config.json"), base_dir=ROOT)
sc = crt.Scene.from_task(t, 800, 600)
r = crt.Render(sc, a.spp, t.P_RR, t.light_sample_n)
iv = crt.get_inverse_view_matrix(t.eye_pos, t.lookat, t.up); fov = crt.fov_to_radians(t.fov_y)
r.run_view(t.eye_pos, iv, fov, stats=True, want_mean=False)
s = r.stats
pc = s["phase_cycles"][4:14]
names = ["inner", "leaf", "LA", "LB", "LC"]
traced = s["rays"] - s.get("rays_untraced", 0)
out = {"scene": a.scene, "env": {k: v for k, v in os.environ.items() if k.startswith("CRT_")}, "rays": s["rays"], "traced": traced,
       "inner_per_ray": round(s["inner_pops"] / traced, 3), "leaf_per_ray": round(s["leaf_pops"] / traced, 3),
       "batches": {n: pc[2 * i] for i, n in enumerate(names)},
       "fill": {n: round(pc[2 * i + 1] / max(1, pc[2 * i]), 2) for i, n in enumerate(names)},
       "stack_deeper_than": {str(k + 1): round(s["phase_cycles"][14 + k] / max(1, s["inner_pops"]), 4) for k in range(6)}, "stack_max": s["stack_max"],
       "stats_kernel_ms": round(s["kernel_ms"], 2)}
ms = []
for i in range(3):
    r.run_view(t.eye_pos, iv, fov, want_mean=False)
    ms.append(round(r.stats["kernel_ms"], 2))
out["kernel_ms"] = ms
print(json.dumps(out))

#!/usr/bin/env python3
"""Visits per ray of the traversal that runs (counting kernel, CRT_FLAG_STATS) and the frame time of the plain kernel: for comparing
tree builds (e.g. CRT_COLLAPSE=dp against the default)."""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cudaraytracing_amd as crt
ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="cornell-box")
ap.add_argument("--spp", type=int, default=512)
a = ap.parse_args()
t = crt.Task(os.path.join(ROOT, "scenes", a.scene, "config.json"), base_dir=ROOT)
sc = crt.Scene.from_task(t, 800, 600)
r = crt.Render(sc, 8, t.P_RR, t.light_sample_n)
r.traversal = {"exact": crt.TRAVERSAL_EXACT, "fast": crt.TRAVERSAL_FAST}[os.environ.get("CRT_PROBE_TRAVERSAL", "exact")]
iv = crt.get_inverse_view_matrix(t.eye_pos, t.lookat, t.up); fov = crt.fov_to_radians(t.fov_y)
r.run_view(t.eye_pos, iv, fov, stats=True, want_mean=False)
s = r.stats
out = {"scene": a.scene, "CRT_COLLAPSE": os.environ.get("CRT_COLLAPSE", ""), "accel": r.accel_info(),
       "inner_per_ray": round(s["inner_pops"] / s["rays"], 3), "leaf_per_ray": round(s["leaf_pops"] / s["rays"], 3),
       "tests_per_ray": round(s["tri_tests"] / s["rays"], 3), "stack_high_water_mean": round(s["stack_sum"] / max(1, s["rays"] - s.get("rays_untraced", 0)), 3),
       "stack_max": s["stack_max"]}
r.set_spp(a.spp)
ms = []
for i in range(3):
    r.run_view(t.eye_pos, iv, fov, want_mean=False)
    ms.append(round(r.stats["kernel_ms"], 2))
out["kernel_ms"] = ms
print(json.dumps(out))

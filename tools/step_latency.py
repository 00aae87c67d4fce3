#!/usr/bin/env python3
"""Per-step latency of a lone path: renders 1x1 pixel, 1 sample (ONE path in ONE wave) for a few seeds and divides the
kernel time by the number of steps the path took (inner + leaf steps + ~3 logic visits per vertex)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cudaraytracing_amd as crt
t = crt.Task(os.path.join(ROOT, "scenes", "cornell-box", "config.json"), base_dir=ROOT)
sc = crt.Scene.from_task(t, 1, 1)
r = crt.Render(sc, 1, t.P_RR, t.light_sample_n)
iv = crt.get_inverse_view_matrix(t.eye_pos, t.lookat, t.up)
fov = crt.fov_to_radians(t.fov_y)
for seed in range(12):
    r.seed = seed
    best = 1e9
    for rep in range(3):
        r.run_view(t.eye_pos, iv, fov, stats=True, want_mean=False, width=1, height=1)
        best = min(best, r.stats["kernel_ms"])
    s = r.stats
    steps = s["inner_pops"] + s["leaf_pops"]
    print(json.dumps({"seed": seed, "rays": s["rays"], "trav_steps": steps, "kernel_us": round(best * 1e3, 1),
                      "us_per_trav_step": round(best * 1e3 / max(steps, 1), 3)}))

#!/usr/bin/env python3
"""What the end of a launch costs: frames of 1, 2, 4 and 64 samples per pixel with the scene's roulette probability and with P_RR = 0
(every path stops at its first vertex).  The 1-sample frame minus the per-sample slope is the fixed cost of a launch: 1.64 ms with
P_RR = 0.6, 0.55 ms with one-vertex paths -- the upper bound of what ordering the work items "short paths last" could save."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cudaraytracing_amd as crt
t = crt.Task(os.path.join(ROOT, "scenes", "cornell-box", "config.json"), base_dir=ROOT)
sc = crt.Scene.from_task(t, 800, 600)
iv = crt.get_inverse_view_matrix(t.eye_pos, t.lookat, t.up); fov = crt.fov_to_radians(t.fov_y)
for prr in (0.6, 0.0):
    for spp in (1, 2, 4, 64):
        r = crt.Render(sc, spp, prr, t.light_sample_n)
        for i in range(3):
            r.run_view(t.eye_pos, iv, fov, want_mean=False)
        print(json.dumps({"p_rr": prr, "spp": spp, "kernel_ms": round(r.stats["kernel_ms"], 3), "rays": r.stats["rays"]}))
        r.free()

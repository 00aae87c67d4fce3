#!/usr/bin/env python3
"""Quick timing probe: renders one frame and prints the per-kernel HIP-event breakdown."""
import argparse, os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cudaraytracing_amd as crt

ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="cornell-box")
ap.add_argument("--width", type=int, default=800)
ap.add_argument("--height", type=int, default=600)
ap.add_argument("--spp", type=int, default=512)
ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--traversal", default="exact", help="exact (the default mode), fast, reference")
a = ap.parse_args()
t = crt.Task(os.path.join(ROOT, "scenes", a.scene, "config.json"), base_dir=ROOT)
sc = crt.Scene.from_task(t, a.width, a.height)
r = crt.Render(sc, a.spp, t.P_RR, t.light_sample_n)
r.traversal = {"fast": crt.TRAVERSAL_FAST, "exact": crt.TRAVERSAL_EXACT, "reference": crt.TRAVERSAL_REFERENCE}[a.traversal]
iv = crt.get_inverse_view_matrix(t.eye_pos, t.lookat, t.up)
fov = crt.fov_to_radians(t.fov_y)
for i in range(a.reps):
    t0 = time.perf_counter()
    r.run_view(t.eye_pos, iv, fov, want_mean=False)
    dt = time.perf_counter() - t0
    s = r.stats
    print(json.dumps({"wall_ms": round(dt * 1e3, 1), "total_ms": round(s["total_ms"], 1), "trace_ms": round(s["kernel_ms"], 3),
                      "logic_ms": round(s["logic_ms"], 1), "launches": s["kernel_launches"], "rays": s["rays"], "untraced": s["rays_untraced"], "shadow": s["shadow_rays"],
                      "Mrays/s": round(s["rays"] / s["total_ms"] / 1e3, 1), **({"stamps": s["phase_cycles"][:12]} if os.environ.get("CRT_PRINT_STAMPS") else {}), "env": {k: v for k, v in os.environ.items() if k.startswith("CRT_")}}))

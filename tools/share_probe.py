#!/usr/bin/env python3
"""Times the per-rank share of a frame on ONE GPU: rank 0 of an N-rank job renders its interleaved tiles.
The ratio t(1) / (N * t(N)) is the strong-scaling efficiency the kernel side allows (no collective included)."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import cudaraytracing_amd as crt

WORKLOADS = {"c2": ("cornell-box", 800, 600, 512), "c3": ("veach-mis", 800, 600, 1024),
             "c4": ("cornell-box", 3840, 2160, 256), "c5": ("veach-mis", 1920, 1080, 4096)}   # BASELINE.json configs
ap = argparse.ArgumentParser()
ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS))
ap.add_argument("--scene", default="cornell-box")
ap.add_argument("--width", type=int, default=800)
ap.add_argument("--height", type=int, default=600)
ap.add_argument("--spp", type=int, default=512)
ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
if a.workload:
    a.scene, a.width, a.height, a.spp = WORKLOADS[a.workload]
t = crt.Task(os.path.join(ROOT, "scenes", a.scene, "config.json"), base_dir=ROOT)
sc = crt.Scene.from_task(t, a.width, a.height)
r = crt.Render(sc, a.spp, t.P_RR, t.light_sample_n)
iv = crt.get_inverse_view_matrix(t.eye_pos, t.lookat, t.up)
fov = crt.fov_to_radians(t.fov_y)
dev = torch.device("cuda:0")
base = None
for world in (1, 2, 4, 8):
    slots = crt.shard_slots(a.width, a.height, 0, world)
    local = torch.empty((slots, 3), dtype=torch.uint8, device=dev)
    best = 1e9
    best_k = 1e9
    for rep in range(a.reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        st = r.run_view_device(t.eye_pos, iv, fov, local.data_ptr(), None, None, rank=0, world=world, tiled=True,
                               want_stats=True, width=a.width, height=a.height)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) * 1e3)
        best_k = min(best_k, st["kernel_ms"])
    if base is None:
        base = best
    print(json.dumps({"workload": a.workload, "scene": a.scene, "width": a.width, "height": a.height, "spp": a.spp, "world": world,
                      "wall_ms": round(best, 2), "kernel_ms": round(best_k, 2), "launches": st["kernel_launches"], "rays": st["rays"],
                      "efficiency": round(base / (world * best), 3)}), flush=True)

#!/usr/bin/env python3
"""Does overlapping consecutive frames (two scene handles, two streams of different priority) hide the end of a launch?
Times K frames of rank 0's share of an N-rank job on ONE GPU, one frame at a time and two in flight."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import cudaraytracing_amd as crt

ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="cornell-box")
ap.add_argument("--spp", type=int, default=512)
ap.add_argument("--frames", type=int, default=12)
a = ap.parse_args()
t = crt.Task(os.path.join(ROOT, "scenes", a.scene, "config.json"), base_dir=ROOT)
sc = crt.Scene.from_task(t, 800, 600)
rs = [crt.Render(sc, a.spp, t.P_RR, t.light_sample_n) for _ in range(2)]
iv = crt.get_inverse_view_matrix(t.eye_pos, t.lookat, t.up)
fov = crt.fov_to_radians(t.fov_y)
dev = torch.device("cuda:0")
streams = [torch.cuda.Stream(device=dev, priority=0), torch.cuda.Stream(device=dev, priority=-1)]
for world in (1, 2, 4, 8):
    slots = crt.shard_slots(800, 600, 0, world)
    bufs = [torch.empty((slots, 3), dtype=torch.uint8, device=dev) for _ in range(2)]
    res = {}
    for depth in (1, 2):
        for warm in range(2):
            for h in range(depth):
                rs[h].run_view_device(t.eye_pos, iv, fov, bufs[h].data_ptr(), None, streams[h].cuda_stream, rank=0, world=world, tiled=True,
                                      want_stats=False, width=800, height=600)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(a.frames):
            h = k % depth
            if k >= depth:
                streams[h].synchronize()
            rs[h].run_view_device(t.eye_pos, iv, fov, bufs[h].data_ptr(), None, streams[h].cuda_stream, rank=0, world=world, tiled=True,
                                  want_stats=False, width=800, height=600)
        torch.cuda.synchronize()
        res[depth] = (time.perf_counter() - t0) * 1e3 / a.frames
    print(json.dumps({"world": world, "ms_per_frame_1_in_flight": round(res[1], 3), "ms_per_frame_2_in_flight": round(res[2], 3),
                      "gain": round(1 - res[2] / res[1], 4)}))

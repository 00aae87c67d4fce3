#!/usr/bin/env python3
"""Times C2 (or another size) with the commit ring at several ring sizes (--rings 0,5,6:32768 = default path, ring of 2^5 samples, ring
of 2^6 with an order window of 32768 items) against one radiance per work item."""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cudaraytracing_amd as crt

ap = argparse.ArgumentParser()
ap.add_argument("--width", type=int, default=800)
ap.add_argument("--height", type=int, default=600)
ap.add_argument("--spp", type=int, default=512)
ap.add_argument("--rings", default="0,5,6,7,8")
ap.add_argument("--scene", default="cornell-box")
a = ap.parse_args()
t = crt.Task(os.path.join(ROOT, "scenes", a.scene, "config.json"), base_dir=ROOT)
sc = crt.Scene.from_task(t, a.width, a.height)
r = crt.Render(sc, a.spp, t.P_RR, t.light_sample_n)
iv = crt.get_inverse_view_matrix(t.eye_pos, t.lookat, t.up)
for ring in a.rings.split(","):
    for k in ("CRT_COMMIT_RING_LOG2", "CRT_ORDER_WINDOW", "CRT_ITEM_ORDER"):
        os.environ.pop(k, None)
    if ring != "0":
        os.environ["CRT_COMMIT_RING_LOG2"] = ring.split(":")[0]
        f = ring.split(":")
        if len(f) > 1:   # second field: order window (0 = no order)
            if f[1] == "0":
                os.environ["CRT_ITEM_ORDER"] = "0"
            else:
                os.environ["CRT_ORDER_WINDOW"] = f[1]
    ms = []
    for rep in range(3):
        r.run_view(t.eye_pos, iv, crt.fov_to_radians(t.fov_y), want_mean=False)
        ms.append(round(r.stats["kernel_ms"], 2))
    print(json.dumps({"ring_log2": ring, "kernel_ms": ms}), flush=True)

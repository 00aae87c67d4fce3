#!/usr/bin/env python3
"""Commit ring against one-radiance-per-work-item on the same renders: the f32 frame (sum of L_k / spp in sample order) and the RGB8
frame must be the same bits.  usage: ring_check.py [--big]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import cudaraytracing_amd as crt

ap = argparse.ArgumentParser()
ap.add_argument("--big", action="store_true", help="also C2 (800x600 spp 512) with the ring the plan chooses")
a = ap.parse_args()


def render(scene, w, h, spp, env):
    for k in ("CRT_COMMIT_RING_LOG2", "CRT_ITEM_ORDER"):
        os.environ.pop(k, None)
    os.environ.update({k: v for k, v in env.items() if k.startswith("CRT_")})
    t = crt.Task(os.path.join(ROOT, "scenes", scene, "config.json"), base_dir=ROOT)
    sc = crt.Scene.from_task(t, w, h)
    r = crt.Render(sc, spp, t.P_RR, t.light_sample_n)
    iv = crt.get_inverse_view_matrix(t.eye_pos, t.lookat, t.up)
    t0 = time.perf_counter()
    if env.get("FLAG"):
        r.extra_flags |= crt.FLAG_BOUNDED_RADIANCE
    rgb = r.run_view(t.eye_pos, iv, crt.fov_to_radians(t.fov_y), want_mean=True)
    return np.asarray(rgb).copy(), np.asarray(r.mean_buffer).copy(), r.stats["kernel_ms"], r.stats["kernel_launches"], r.radiance_storage()


bad = 0
cases = [("cornell-box", 128, 96, 64, "2"), ("cornell-box", 128, 96, 64, "3"), ("cornell-box", 203, 149, 48, "2"), ("veach-mis", 160, 120, 40, "2"),
         ("cornell-box", 64, 64, 200, "4"), ("cornell-box", 128, 96, 64, "2:noorder")]
for scene, w, h, spp, ring in cases:
    env = {"CRT_COMMIT_RING_LOG2": ring.split(":")[0]}
    if ring.endswith("noorder"):
        env["CRT_ITEM_ORDER"] = "0"
    ref = render(scene, w, h, spp, {})
    got = render(scene, w, h, spp, env)
    same = np.array_equal(ref[1].view(np.uint32), got[1].view(np.uint32)) and np.array_equal(ref[0], got[0])
    bad += not same
    print(json.dumps({"scene": scene, "w": w, "h": h, "spp": spp, "ring_log2": ring, "same_bits": bool(same), "ms_full": round(ref[2], 2), "ms_ring": round(got[2], 2),
                      "launches": [ref[3], got[3]], "storage": [ref[4], got[4]], "differing_pixels": int((ref[1].view(np.uint32) != got[1].view(np.uint32)).any(axis=-1).sum())}), flush=True)
if a.big:
    for rep in range(2):
        ref = render("cornell-box", 800, 600, 512, {})
        got = render("cornell-box", 800, 600, 512, {"FLAG": "1"})
        same = np.array_equal(ref[1].view(np.uint32), got[1].view(np.uint32)) and np.array_equal(ref[0], got[0])
        bad += not same
        print(json.dumps({"scene": "C2", "same_bits": bool(same), "ms_full": round(ref[2], 2), "ms_ring": round(got[2], 2), "launches": [ref[3], got[3]], "storage": [ref[4], got[4]]}), flush=True)
sys.exit(1 if bad else 0)

#!/usr/bin/env python3
"""How far in front of its own leaf box does the hit that answers a query lie?  (CPU oracle only; no GPU.)

CRT_TRAVERSAL_FAST skips a box entered beyond t_ref + slack; it loses an answer only if the Moeller-Trumbore distance of a hit lies
more than `slack` in front of the entry of the leaf box the triangle is in.  This tool histograms that margin, in units of reach x steep
(csrc/crt_trace.h: prune_bound), over every ray the oracle traces for random pixel crops, so that the tail beyond the shipped
slack factor can be read off.  Writes one JSON line."""
import argparse, ctypes as C, json, multiprocessing as mp, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np


def work(job):
    scene, width, height, spp, seed, crops = job
    import cudaraytracing_amd as crt   # Task / camera helpers only (host side)
    import oracle_lib as O
    t = crt.Task(os.path.join(ROOT, "scenes", scene, "config.json"), base_dir=ROOT)
    iv = crt.get_inverse_view_matrix(t.eye_pos, t.lookat, t.up)
    fov = crt.fov_to_radians(t.fov_y)
    osc = O.OracleScene(t.OBJ_paths, t.bvh_thresh_n)
    L = O.lib()
    L.orc_margin_hist.restype = None
    L.orc_margin_hist.argtypes = [C.c_void_p]
    L.orc_margin_hist(None)
    for (x, y, w, h) in crops:
        osc.render(t.eye_pos, iv, fov, width, height, spp, t.P_RR, t.light_sample_n, seed=seed, crop=(x, y, w, h))
    out = np.zeros(26, dtype=np.uint64)
    L.orc_margin_hist(out.ctypes.data_as(C.c_void_p))
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="veach-mis")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=64)
    ap.add_argument("--crops", type=int, default=64, help="32x32 crops per worker")
    ap.add_argument("--workers", type=int, default=8)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    jobs = []
    for wkr in range(a.workers):
        crops = [(int(rng.integers(0, a.width - 32)), int(rng.integers(0, a.height - 32)), 32, 32) for _ in range(a.crops)]
        jobs.append((a.scene, a.width, a.height, a.spp, a.seed * 1000 + wkr, crops))
    t0 = time.time()
    with mp.get_context("spawn").Pool(a.workers) as pool:
        hist = sum(pool.map(work, jobs))
    edges = [10.0 ** (b / 2 - 10) for b in range(25)]
    tail = np.cumsum(hist[:24][::-1])[::-1]
    print(json.dumps({"scene": a.scene, "width": a.width, "height": a.height, "spp": a.spp, "rays": int(hist[25]), "answers_with_hit": int(hist[24]),
                      "seconds": round(time.time() - t0, 1),
                      "margin_ge": {f"{edges[b]:.1e}": int(tail[b]) for b in range(24) if tail[b] or b < 16}}))

"""Run in a child process by test_first_frame.py (and by tools/r05/handoff_ab.sh with CRT_LIB_PATH = an experiment build): first frames of
many short-lived Render objects of EQUAL size that alternate between two scenes and between the default path and the commit ring, every new
device allocation pre-filled with 0xFF bytes (CRT_DEBUG_FILL=255: NaN radiance, out-of-range work items), each frame against the oracle.
What it is after: a kernel-to-kernel hand-off that reads what a previous owner of the address left there (docs/experiments.md 6) -- the allocator
hands a freed buffer to the next Render of the same size, so the stale copy and the fresh one share an address.  Prints one JSON line."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
import cudaraytracing_amd as crt  # noqa: E402
import util  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
w, h, spp = 96, 64, 24
scenes = ["cornell-box", "veach-mis"]
ref = {}
for name in scenes:
    t = util.task(name)
    eye, iv, fov = util.camera(name)
    ref[name] = [util.oracle_scene(name).render(eye, iv, fov, w, h, spp, t.P_RR, t.light_sample_n, seed=s)[1] for s in (0, 1)]
bad_frames, bad_pixels, nan_pixels, ring_frames = [], 0, 0, 0
for k in range(n):
    name = scenes[k & 1]
    seed = (k >> 1) & 1
    t = util.task(name)
    eye, iv, fov = util.camera(name)
    r = crt.Render(util.host_scene(name), spp, t.P_RR, t.light_sample_n, device=0)
    try:
        r.seed = seed
        if k % 5 == 4:  # every fifth render goes through the commit ring: its buffers are uncached allocations of other sizes
            r.extra_flags = crt.FLAG_BOUNDED_RADIANCE
            ring_frames += 1
        r.run_view(eye, iv, fov, width=w, height=h)
        bad = (util.bits(r.mean_buffer) != util.bits(ref[name][seed])).any(axis=2)
        if bad.any():
            bad_frames.append(k)
            bad_pixels += int(bad.sum())
            nan_pixels += int(np.isnan(r.mean_buffer).any(axis=2).sum())
    finally:
        r.free()
print(json.dumps({"library": os.environ.get("CRT_LIB_PATH"), "debug_fill": os.environ.get("CRT_DEBUG_FILL"), "first_frames": n, "through_the_commit_ring": ring_frames,
                  "bad_frames": bad_frames, "bad_pixels": bad_pixels, "of_them_nan": nan_pixels}))

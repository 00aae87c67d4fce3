"""Run in a child process by test_leafq_overflow.py with CRT_LIB_PATH = a libcrt.so built with -DLEAFQ_CAP=128: renders crops of both
shipped scenes through the default (decoupled-leaves) kernel and its counting form and compares them with the oracle bit for bit; prints
one JSON line with what differed and how often the queue-overflow paths of the inner step ran (crt_mega3.hip: inner4_step_dec CHECK 1 /
2: lanes taken back out of a first visit, second visits dropped)."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
import cudaraytracing_amd as crt  # noqa: E402
import util  # noqa: E402

out = {"library": os.environ.get("CRT_LIB_PATH"), "cases": []}
for name, w, h, spp in (("cornell-box", 96, 64, 64), ("veach-mis", 96, 64, 48)):
    t = util.task(name)
    eye, iv, fov = util.camera(name)
    _, omean, _, ost = util.oracle_scene(name).render(eye, iv, fov, w, h, spp, t.P_RR, t.light_sample_n)
    for flags, label in ((0, "default"), (crt.FLAG_FORCE_EXACT, "force_exact"), (crt.FLAG_TRACE_ALL, "trace_all")):
        r = crt.Render(util.host_scene(name), spp, t.P_RR, t.light_sample_n, device=0)
        try:
            r.extra_flags = flags
            for stats in (False, True):
                r.run_view(eye, iv, fov, width=w, height=h, stats=stats)
                bad = int((util.bits(r.mean_buffer) != util.bits(omean)).any(axis=2).sum())
                case = {"scene": name, "flags": label, "counting_kernel": stats, "pixels_differ": bad, "rays_equal": bool(r.stats["rays"] == ost["rays"])}
                if stats:
                    pc = r.stats["phase_cycles"]
                    case.update({"lanes_taken_back": int(pc[20]), "second_visits_dropped_lanes": int(pc[21]), "inner_batches": int(pc[4])})
                out["cases"].append(case)
        finally:
            r.free()
print(json.dumps(out))

"""Effect of the insertion-based optimisation of the SAH tree (csrc/crt_accel.h: optimize_sah, CRT_SAH_OPT=<passes>) on the default
render: scene set-up time, visits per ray (counting kernel) and frame time (plain kernel), per scene and number of passes.
usage: python3 tools/sah_opt_probe.py [passes ...]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import cudaraytracing_amd as crt
import util

passes = [int(x) for x in sys.argv[1:]] or [0, 1, 2, 3, 4, 8]
import tempfile
sys.path.insert(0, os.path.join(ROOT, "scenes"))
import gen_cornell_box
td = tempfile.mkdtemp()
big_obj, big_mtl, _ = gen_cornell_box.write_variant(td, (6, 5))
for name, spp in (("cornell-box", 512), ("veach-mis", 256), ("cornell-box-102412", 256)):
    big = name.endswith("102412")
    t = util.task("cornell-box" if big else name)
    eye, iv, fov = util.camera("cornell-box" if big else name)
    for p in passes:
        os.environ["CRT_SAH_OPT"] = str(p)
        if big:
            sc = crt.Scene(800, 600)
            sc.add_obj(big_obj, big_mtl)
            sc.set_BVH(t.bvh_thresh_n)
        else:
            sc = crt.Scene.from_task(t, 800, 600)
        t0 = time.perf_counter()
        r = crt.Render(sc, spp, t.P_RR, t.light_sample_n)
        setup = (time.perf_counter() - t0) * 1e3
        r.traversal = crt.TRAVERSAL_EXACT
        ai = r.accel_info()
        r.set_spp(8)
        r.run_view(eye, iv, fov, stats=True, width=800, height=600)
        st = dict(r.stats)
        r.set_spp(spp)
        ms = []
        for _ in range(4):
            r.run_view(eye, iv, fov, width=800, height=600)
            ms.append(r.stats["kernel_ms"])
        print(json.dumps({"scene": name, "passes": p, "setup_ms": round(setup, 1), "sah_ms": round(ai["sah_ms"], 2), "nodes4": ai["n_nodes4"], "depth": ai.get("depth_fast"),
                          "inner_per_ray": round(st["inner_pops"] / st["rays"], 3), "leaf_per_ray": round(st["leaf_pops"] / st["rays"], 3),
                          "kernel_ms": round(min(ms[1:]), 2)}), flush=True)
        r.free()
        sc.free()

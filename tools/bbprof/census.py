#!/usr/bin/env python3
"""Writes profiles/bbprof_latest.json: the exact dynamic census of the render kernel's vector instructions (basic-block profile on
the production ISA, tools/bbprof) priced with the measured per-opcode issue costs (tools/valu_issue_gen.py) -- what bench.py's
vector-issue roofline uses as "cycles per wave64 VALU instruction" -- STAMPED with the hash of the kernel sources and build flags
(cudaraytracing_amd.build.source_hash); bench.py ignores it when the hash differs from the library that is running.

  census.py <run dir with k_mega3_bb.json, c2.txt [, c3.txt]> --c2-spp 256 [--c3-spp 64] --costs profiles/r03_valu_issue_ops.json
"""
import argparse
import json
import os
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("run")
    ap.add_argument("--c2-spp", type=int, default=256)
    ap.add_argument("--c3-spp", type=int, default=64)
    ap.add_argument("--costs", default=os.path.join(ROOT, "profiles", "r03_valu_issue_ops.json"))
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "bbprof_latest.json"))
    a = ap.parse_args()
    from cudaraytracing_amd import build as B
    out = {"src_hash": B.source_hash(), "build_flags": B.flags_string(), "collected": time.strftime("%Y-%m-%d %H:%M:%S"),
           "method": "tools/bbprof: every basic block of the compiler's assembly of the production kernel counts its executions and active lanes; "
                     "opcode costs from tools/valu_issue_gen.py (4 waves per SIMD, independent streams)",
           "price_list": os.path.relpath(a.costs, ROOT), "workloads": {}}
    for wl, spp_run, spp_full, what in (("c2", a.c2_spp, 512, "cornell-box 800x600"), ("c3", a.c3_spp, 1024, "veach-mis 800x600")):
        counts = os.path.join(a.run, wl + ".txt")
        if not os.path.exists(counts):
            continue
        tmp = os.path.join(a.run, wl + "_report.json")
        txt = subprocess.check_output([sys.executable, os.path.join(HERE, "report.py"), os.path.join(a.run, "k_mega3_bb.json"), counts, "--costs", a.costs,
                                       "--json", tmp, "--top", "40"], text=True)
        open(os.path.join(a.run, wl + "_report.txt"), "w").write(txt)
        r = json.load(open(tmp))
        scale = spp_full / float(spp_run)
        # what the vector instructions are: intersection arithmetic (the box and triangle tests with their reciprocals, minima and
        # maxima), the rest of the two traversal steps (rings, records, stacks, routing), the path logic
        ARITH = ("tri_pair", "slab_quad_pruned", "slab_quad_hits", "slab_pair_hits", "slab_pair", "slab_pair_pruned", "rcp_short", "rcp_short_ok", "rcp_short_ok2", "rcp_short_ok3", "rcp_ieee",
                 "inv3_exact", "fmin3", "fmax3", "finite3", "pk_submul_ll", "pk_submul_hh", "pk_bmul", "pk_bsub")  # (round 6: the packed instructions written with operand selects)
        trav = ("inner", "leaf", "inner_ex", "leaf_ex")
        arith = sum(e["dyn_valu"] for e in r["by_phase_function"] if e["phase"] in trav and e["function"] in ARITH)
        trav_all = sum(e["dyn_valu"] for e in r["by_phase_function"] if e["phase"] in trav) + r["by_phase"].get("other", {}).get("dyn_valu", 0)
        out["workloads"][wl] = {
            "profiled": "%s spp=%d (one launch), scaled x%g to spp=%d" % (what, spp_run, scale, spp_full), "kernel": r["kernel"],
            "valu_instructions_per_launch": r["dyn_valu"] * scale, "lane_utilisation": r["lane_utilisation"],
            "cycles_per_valu": r["cycles_per_valu"]["mid"], "cycles_per_valu_range": [r["cycles_per_valu"]["lo"], r["cycles_per_valu"]["hi"]],
            "unpriced_share": r["cycles_per_valu"]["unpriced_share"], "tiers": r["cycles_per_valu"]["tiers"],
            "arith_share": arith / r["dyn_valu"], "traversal_bookkeeping_share": (trav_all - arith) / r["dyn_valu"],
            "path_logic_share": 1.0 - trav_all / r["dyn_valu"],
            "arith_functions": list(ARITH),
            "instructions_per_launch_by_kind": {k: v * scale for k, v in r["kinds"].items()},
            "by_phase": {ph: {"valu_share": v["dyn_valu"] / r["dyn_valu"], "lanes": v["lanes"]} for ph, v in r["by_phase"].items() if v["dyn_valu"] > 0.0005 * r["dyn_valu"]}}
    json.dump(out, open(a.out, "w"), indent=1)
    print(json.dumps({k: (v if k != "workloads" else {w: {x: y for x, y in d.items() if x in ("valu_instructions_per_launch", "lane_utilisation", "cycles_per_valu")}
                                                         for w, d in v.items()}) for k, v in out.items()}, indent=1))


if __name__ == "__main__":
    main()

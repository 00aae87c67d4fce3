#!/usr/bin/env python3
"""Builds pmc_latest.json (read by bench.py for the roofline) from the summaries tools/collect_profiles.sh writes.

Per workload (c2 = cornell-box 800x600 spp 512, c3 = veach-mis 800x600 spp 1024; one frame = one launch of k_mega3) the raw
counter sums of the dominant kernel, collected with one rocprofv3 --pmc pass per group (FETCH_SIZE and WRITE_SIZE in
separate passes; their KiB unit and the gfx950 read correction are applied by the reader: bytes = 2 x FETCH_SIZE x 1024 +
WRITE_SIZE x 1024, MI355X_MICROARCH.md "HBM").  The file is stamped with the hash of the kernel sources and build flags it was
collected on.  usage: make_pmc_json.py <profiles dir> [valu cycles per instruction]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cudaraytracing_amd import build as B

# the default render path: EXACT traversal (FAST as a fallback for older profiles), no counting, zero-contribution samples answered without traversal, not the query form
DEFAULT_KERNEL = ("k_mega3<2, false, false, false, true, false, true, true>", "k_mega3<2, false, false, false, true, false, true, false>",   # (MODE, STATS, ALL, QUERY, R16, RING, DEC, IMPL)
                  "k_mega3<2, false, false, false, true, false, true>", "k_mega3<2, false, false, false, true, false, false>",   # (MODE, STATS, ALL, QUERY, R16, RING, DEC)
                  "k_mega3<2, false, false, false, false, false, true>", "k_mega3<2, false, false, false, false, false, false>",
                  "k_mega3<2, false, false, false, true, false>", "k_mega3<2, false, false, false, false, false>",   # (MODE, STATS, ALL, QUERY, R16, RING)
                  "k_mega3<2, false, false, false, true>", "k_mega3<2, false, false, false, false>",
                  "k_mega3<0, false, false, false, true>", "k_mega3<0, false, false, false, false>", "k_mega3<0, false, false, false>",
                  "k_mega3<0, false, false>", "k_mega3<0, false>")  # the default render path: CRT_TRAVERSAL_EXACT, 16-bit stack layout if the scene allows it
root = sys.argv[1]
# Issue cost of one wave64 vector instruction on one SIMD of gfx950, by counter class, measured with tools/valu_issue_bench.hip
# (independent streams, 4 and 8 waves per SIMD; profiles/r02_valu_issue.json): v_add_f32 / v_mul_f32 / v_mov_b32 / v_and_b32 2.5
# cycles, v_fma_f32 / v_fmac_f32 3.8-3.9, v_max_f32 / v_max3_f32 / v_pk_add_f32 / v_pk_mul_f32 / v_cndmask_b32 / v_mul_lo_u32 4.2,
# v_cmp + v_cndmask 3.4 each, v_rcp_f32 / v_sqrt_f32 8.2.  INT32 and the unclassified rest (moves, compares, selects, min / max,
# packed) mix 2.5- and 4.2-cycle members: priced at the midpoint, with the all-cheap / all-dear range kept beside it.
COST = {"SQ_INSTS_VALU_ADD_F32": (2.5, 2.5, 2.5), "SQ_INSTS_VALU_MUL_F32": (2.5, 2.5, 2.5), "SQ_INSTS_VALU_FMA_F32": (3.9, 3.9, 3.9),
        "SQ_INSTS_VALU_TRANS_F32": (8.2, 8.2, 8.2), "SQ_INSTS_VALU_INT32": (2.5, 3.35, 4.2), "SQ_INSTS_VALU_CVT": (4.2, 4.2, 4.2),
        "other": (2.5, 3.35, 4.2)}
cyc = float(sys.argv[2]) if len(sys.argv) > 2 else 3.35
res = {"src_hash": B.source_hash(), "build_flags": B.built_flags(), "collected": time.strftime("%Y-%m-%d %H:%M:%S"),
       "command": "tools/collect_profiles.sh: rocprofv3 --kernel-trace --pmc <one group> -- python3 tools/perf_probe.py --scene S --spp N --reps 1",
       "valu_cycles_per_instr_source": "tools/valu_issue_bench.hip (independent v_add_f32 / v_pk_* streams, 4 waves per SIMD)",
       "workloads": {}}
for wl, desc in (("c2", "cornell-box 800x600 spp=512"), ("c3", "veach-mis 800x600 spp=1024")):
    f = os.path.join(root, wl + "_summary.json")
    if not os.path.exists(f):
        continue
    d = json.load(open(f))
    ks = [n for n in DEFAULT_KERNEL if n in d]
    if not ks:
        continue
    v = dict(d[ks[0]])
    v.pop("stats", None)
    w = {"workload": desc + ", 1 MI355X, one frame = 1 launch", "kernel": ks[0].replace(", ", ","), "launches": 1,
         "valu_cycles_per_instr": cyc, "collected": res["collected"]}
    w.update(v)
    if v.get("SQ_INSTS_VALU"):
        w["salu_per_valu"] = round(v.get("SQ_INSTS_SALU", 0.0) / v["SQ_INSTS_VALU"], 4)
        if all(v.get(c) is not None for c in COST if c != "other"):
            rest = v["SQ_INSTS_VALU"] - sum(v[c] for c in COST if c != "other")
            tot = [0.0, 0.0, 0.0]
            for c, cost in COST.items():
                n = rest if c == "other" else v[c]
                for i in range(3):
                    tot[i] += n * cost[i]
            w["valu_cycles_per_instr"] = round(tot[1] / v["SQ_INSTS_VALU"], 3)
            w["valu_cycles_per_instr_range"] = [round(tot[0] / v["SQ_INSTS_VALU"], 3), round(tot[2] / v["SQ_INSTS_VALU"], 3)]
            w["valu_class_costs"] = {c: list(cost) for c, cost in COST.items()}
    if v.get("TCC_HIT_sum") is not None and v.get("TCC_MISS_sum") is not None and (v["TCC_HIT_sum"] + v["TCC_MISS_sum"]) > 0:
        w["tcc_miss_frac"] = round(v["TCC_MISS_sum"] / (v["TCC_HIT_sum"] + v["TCC_MISS_sum"]), 4)
    if v.get("TCC_EA0_RDREQ_DRAM_32B_sum") is not None and v.get("TCC_EA0_WRREQ_WRITE_DRAM_32B_sum") is not None:
        # exact bytes on the L2's fabric side: 32-byte units of reads, writes and atomics bound for the device's own memory
        w["fabric_read_bytes"] = 32.0 * v["TCC_EA0_RDREQ_DRAM_32B_sum"]
        w["fabric_write_bytes"] = 32.0 * (v["TCC_EA0_WRREQ_WRITE_DRAM_32B_sum"] + v.get("TCC_EA0_WRREQ_ATOMIC_DRAM_32B_sum", 0.0))
        if v.get("TCC_EA0_RDREQ_sum"):
            w["fabric_read_bytes_per_request"] = round(w["fabric_read_bytes"] / v["TCC_EA0_RDREQ_sum"], 2)
    if v.get("TA_TA_BUSY_sum") is not None and v.get("GRBM_GUI_ACTIVE"):
        # texture addressers (one per CU, 256) busy / all their cycles; GRBM_GUI_ACTIVE is summed over the 8 XCDs
        w["ta_busy_frac"] = round(v["TA_TA_BUSY_sum"] / (32.0 * v["GRBM_GUI_ACTIVE"]), 4)
    if v.get("TCP_TCC_READ_REQ_sum"):
        w["l1_read_latency_cycles"] = round(v.get("TCP_TCC_READ_REQ_LATENCY_sum", 0.0) / v["TCP_TCC_READ_REQ_sum"], 1)
        if v.get("SQ_INSTS_VMEM"):
            w["l2_read_requests_per_vmem_instr"] = round(v["TCP_TCC_READ_REQ_sum"] / v["SQ_INSTS_VMEM"], 3)
    if all(v.get("TCP_TAGRAM%d_REQ_sum" % i) is not None for i in range(4)) and v.get("SQ_INSTS_VMEM"):
        w["l1_tag_lookups_per_vmem_instr"] = round(sum(v["TCP_TAGRAM%d_REQ_sum" % i] for i in range(4)) / v["SQ_INSTS_VMEM"], 3)
    if v.get("SQ_LDS_BANK_CONFLICT") is not None and v.get("SQ_LDS_IDX_ACTIVE"):
        w["lds_conflict_frac"] = round(v["SQ_LDS_BANK_CONFLICT"] / v["SQ_LDS_IDX_ACTIVE"], 4)
    res["workloads"][wl] = w
# average launch duration of the profiled bench run, for the agreement check against bench.py's HIP-event time
f = os.path.join(root, "stats_summary.json")
if os.path.exists(f):
    d = json.load(open(f))
    for n in DEFAULT_KERNEL:  # (in order of preference: the kernel of the default mode)
        v = d.get(n)
        if v and "stats" in v and "c2" in res["workloads"]:
            res["workloads"]["c2"]["avg_launch_ms"] = round(float(v["stats"]["AverageNs"]) / 1e6, 3)
            res["workloads"]["c2"]["stats_calls"] = int(v["stats"]["Calls"])
            break
print(json.dumps(res, indent=1))

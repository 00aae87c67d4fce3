#!/usr/bin/env python3
"""Summarises rocprofv3 counter_collection / kernel_trace CSVs per kernel (sums over dispatches)."""
import collections, csv, glob, json, re, sys

def kname(n):
    m = re.search(r"(k_\w+(<[^>]*>)?)", n)
    return m.group(1) if m else n[:40]

root = sys.argv[1]
out = {}
for f in glob.glob(root + "/**/*_counter_collection.csv", recursive=True):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    meta = {}
    for r in csv.DictReader(open(f)):
        k = kname(r["Kernel_Name"])
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        meta[k] = {x: r[x] for x in ("VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Workgroup_Size")}
    for k, v in agg.items():
        if k.startswith("k_"):
            d = out.setdefault(k, {"dispatch": meta[k]})
            d.update(v)
for f in glob.glob(root + "/**/*_kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = kname(r["Name"])
        if k.startswith("k_"):
            out.setdefault(k, {})["stats"] = {x: r[x] for x in ("Calls", "TotalDurationNs", "AverageNs", "Percentage")}
for k, d in out.items():
    if "SQ_THREAD_CYCLES_VALU" in d and d.get("SQ_ACTIVE_INST_VALU"):
        d["valu_lane_utilization"] = d["SQ_THREAD_CYCLES_VALU"] / d["SQ_ACTIVE_INST_VALU"] / 64
    if "SQ_WAVE_CYCLES" in d and d.get("SQ_WAVE_CYCLES"):
        for c in ("SQ_WAIT_ANY", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_WAIT_INST_LDS"):
            if c in d:
                d[c + "/WAVE_CYCLES"] = d[c] / d["SQ_WAVE_CYCLES"]
print(json.dumps(out, indent=1))

#!/usr/bin/env python3
"""Builds profiles/hbm_traffic_latest.json (read by bench.py for roofline.traffic) from the summary.json that
tools/collect_profiles.sh writes: FETCH_SIZE and WRITE_SIZE of the dominant kernel, collected in separate --pmc
passes and corrected as MI355X_MICROARCH.md prescribes (KiB units; gfx950 counts the 128-B requests of 16 B/lane
loads as 64 B, so reads are doubled)."""
import json, sys
summary, bench_line, out = sys.argv[1], sys.argv[2], sys.argv[3]
d = json.load(open(summary))
k = [n for n in d if n.startswith("k_mega3<0, false, false>") or n.startswith("k_mega3<0, false>")][0]
v = d[k]
calls = 1  # tools/perf_probe.py --reps 1: one launch per pass
fetch = v["FETCH_SIZE"] * 1024.0 / calls
write = v["WRITE_SIZE"] * 1024.0 / calls
line = json.loads(open(bench_line).read())
r = line["roofline"]
res = {
    "kernel": k.replace(", ", ","),
    "workload": "cornell-box 800x600 spp=512, 1 MI355X, one frame = 1 launch",
    "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE (and, separately, --pmc WRITE_SIZE) --output-format csv -- python3 tools/perf_probe.py --spp 512 --reps 1",
    "fetch_bytes_per_launch_raw": fetch,
    "write_bytes_per_launch": write,
    "bytes_per_launch": 2.0 * fetch + write,
    "correction": "FETCH_SIZE/WRITE_SIZE are KiB; gfx950 FETCH_SIZE counts 128-B requests as 64 B for 16 B/lane loads, so reads are doubled (MI355X_MICROARCH.md, HBM); writes are taken as is",
    "algorithmic_bytes_per_launch": r["rays_per_launch"] * r["bytes_per_ray"],
    "note": "writes: 16 B of radiance per path (3.9 GB) + path-state planes and vertex records evicted from L2; reads: state planes / records that missed L2 + the scene once; nodes and triangles stay cache resident",
}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res))

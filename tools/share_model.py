#!/usr/bin/env python3
"""profiles/share_model.json from the lines of tools/share_probe.py (rank 0's share of a BASELINE configuration at 1 / 2 / 4 / 8 ranks,
timed on ONE GPU): per workload the model share(N) = tail + (t1 - tail) / N -- every launch ends with its waves running their pools
of paths dry, a fixed cost, the rest divides by N -- fitted by least squares to the measured shares.  bench.py --gpus N prints the
model's prediction beside what it measures (per_rank.predicted_share_ms).
usage: share_model.py <share_probe.jsonl> [...] > profiles/share_model.json"""
import json
import sys

rows = {}
for f in sys.argv[1:]:
    for line in open(f):
        if not line.startswith("{"):
            continue
        d = json.loads(line)
        if d.get("workload"):
            rows.setdefault(d["workload"], {})[int(d["world"])] = d
out = {"note": "share(N) = tail + (t1 - tail) / N, least squares over the measured kernel time of rank 0's share at N = 2, 4, 8 on one GPU "
               "(tools/share_probe.py --workload ...); t1 = the whole frame on one GPU", "workloads": {}}
for wl, r in sorted(rows.items()):
    if 1 not in r:
        continue
    t1 = r[1]["kernel_ms"]
    # share - t1 / N = tail (1 - 1 / N)
    num = sum((r[n]["kernel_ms"] - t1 / n) * (1.0 - 1.0 / n) for n in r if n > 1)
    den = sum((1.0 - 1.0 / n) ** 2 for n in r if n > 1)
    tail = max(0.0, num / den) if den else 0.0   # (a frame of several launches on one GPU pays several tails: the fit can come out slightly negative)
    out["workloads"][wl] = {"workload": "%s %dx%d spp=%d" % (r[1]["scene"], r[1]["width"], r[1]["height"], r[1]["spp"]),
                            "one_gpu_kernel_ms": t1, "launches_per_frame_on_one_gpu": r[1].get("launches"), "tail_ms": round(tail, 3),
                            "measured_share_kernel_ms": {str(n): r[n]["kernel_ms"] for n in sorted(r)},
                            "measured_share_wall_ms": {str(n): r[n]["wall_ms"] for n in sorted(r)},
                            "kernel_side_efficiency": {str(n): round(t1 / (n * r[n]["kernel_ms"]), 4) for n in sorted(r)},
                            "note": "share(N) = %.3f + (%.2f - %.3f) / N ms (one-GPU model: tools/share_model.py)" % (tail, t1, tail)}
print(json.dumps(out, indent=1))

#!/usr/bin/env python3
"""Reads the last JSON line of tools/diag_phases.sh output (stdin) and prints the per-phase census of k_mega3."""
import sys, json
line = [l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]
d = json.loads(line); p = d["phase_cycles"]
names = ["INNER", "LEAF", "LA", "LB", "LC"]
it = p[4:9]; ln = p[9:14]; cy = p[14:19]
tot = sum(p[0:4])
print("total cycles %.1fe9, ms %.1f" % (tot / 1e9, d["trace_ms"]))
for i, n in enumerate(names):
    print("%-5s iters %.2fM lanes/iter %.1f cycles/iter %.0f share %.1f%%" % (n, it[i] / 1e6, ln[i] / max(it[i], 1), cy[i] / max(it[i], 1), 100 * cy[i] / tot))
print("other share %.1f%%" % (100 * p[3] / tot))
sec = p[19:24]
ii = max(it[0], 1)
print("inner step sections (cycles per iteration, s_memtime after s_waitcnt 0): ring id %.0f | record %.0f | node data %.0f | boxes+pushes %.0f | pop %.0f | rest (write-back, ring append) %.0f"
      % (sec[0] / ii, sec[1] / ii, sec[2] / ii, sec[3] / ii, sec[4] / ii, (cy[0] - sum(sec)) / ii))

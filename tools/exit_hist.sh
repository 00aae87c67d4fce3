#!/bin/bash
# Diagnostic: histogram of wave lifetimes of k_mega3 (how a launch ends).  usage: tools/exit_hist.sh <bucket_us> <spp>
trap 'unset CRT_EXTRA_CXXFLAGS; python3 cudaraytracing_amd/build.py --force > /dev/null 2>&1' EXIT  # always leave the default build in the tree
b=${1:-100}; spp=${2:-1}
export CRT_EXTRA_CXXFLAGS="-DCRT_EXIT_HIST=$b"; python3 cudaraytracing_amd/build.py --force > /dev/null 2>&1 || { echo "build failed"; exit 1; }
python3 tools/perf_probe.py --spp $spp --reps 2 | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); p = d['phase_cycles'][4:24]
print('kernel ms', d['trace_ms'], 'waves', sum(p))
for i, n in enumerate(p):
    if n: print('%5d..%5d us: %d' % (i * $b, (i + 1) * $b, n))"
unset CRT_EXTRA_CXXFLAGS; python3 cudaraytracing_amd/build.py --force > /dev/null 2>&1
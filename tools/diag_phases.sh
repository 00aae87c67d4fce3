#!/bin/bash
set -o pipefail  # a crashed probe must stop the script (a GPU fault must never be followed by another GPU step)
# Diagnostic: rebuilds libcrt.so with -DCRT_STAMPS on the GPU box and prints the per-phase cycle /
# iteration / lane counters of k_mega3 (phase_cycles of crt_stats).  usage: tools/diag_phases.sh [spp] [scene]
spp=${1:-64}; scene=${2:-cornell-box}
trap 'unset CRT_EXTRA_CXXFLAGS; python3 cudaraytracing_amd/build.py --force > /dev/null 2>&1' EXIT  # always leave the default build in the tree
export CRT_EXTRA_CXXFLAGS=-DCRT_STAMPS; python3 cudaraytracing_amd/build.py --force > /dev/null 2>&1 || { echo "diag build failed"; exit 1; }
python3 tools/perf_probe.py --spp $spp --reps 2 --scene $scene
unset CRT_EXTRA_CXXFLAGS; python3 cudaraytracing_amd/build.py --force > /dev/null 2>&1
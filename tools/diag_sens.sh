#!/bin/bash
# Diagnostic: sensitivity of k_mega3 to extra divergent loads / extra VALU per inner step (-DCRT_STAMPS build).
trap 'unset CRT_EXTRA_CXXFLAGS; python3 cudaraytracing_amd/build.py --force > /dev/null 2>&1' EXIT  # always leave the default build in the tree
spp=${1:-64}
export CRT_EXTRA_CXXFLAGS=-DCRT_STAMPS; python3 cudaraytracing_amd/build.py --force > /dev/null 2>&1 || { echo "diag build failed"; exit 1; }
for cfg in "0 0" "4 0" "8 0" "0 50" "0 100" "0 200"; do
  set -- $cfg
  echo "== extra loads $1, extra valu $2"
  CRT_DBG_LOADS=$1 CRT_DBG_VALU=$2 python3 tools/perf_probe.py --spp $spp --reps 2 | tail -1
done
unset CRT_EXTRA_CXXFLAGS; python3 cudaraytracing_amd/build.py --force > /dev/null 2>&1
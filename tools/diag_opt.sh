#!/bin/bash
set -o pipefail
trap 'unset CRT_EXTRA_CXXFLAGS; python3 cudaraytracing_amd/build.py --force > /dev/null 2>&1' EXIT
export CRT_EXTRA_CXXFLAGS=-DCRT_STAMPS; python3 cudaraytracing_amd/build.py --force > /dev/null 2>&1 || { echo "diag build failed"; exit 1; }
for o in 0 1 2; do
  echo "== CRT_SAH_OPT=$o"
  CRT_SAH_OPT=$o python3 tools/perf_probe.py --spp 512 --reps 2 --scene cornell-box | tail -1 | python3 tools/diag_summary.py || exit 2
done

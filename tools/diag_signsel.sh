#!/bin/bash
set -o pipefail  # a crashed probe must stop the script (a GPU fault must never be followed by another GPU step)
trap 'unset CRT_EXTRA_CXXFLAGS; python3 cudaraytracing_amd/build.py --force > /dev/null 2>&1' EXIT
for f in "-DCRT_STAMPS -DCRT_NODE_SIGNSEL=0" "-DCRT_STAMPS -DCRT_NODE_SIGNSEL=1"; do
export CRT_EXTRA_CXXFLAGS="$f"; python3 cudaraytracing_amd/build.py --force > /dev/null 2>&1 || { echo "diag build failed"; exit 1; }
echo "== $f"
python3 tools/perf_probe.py --spp 512 --reps 2 --scene cornell-box | tail -1 | python3 tools/diag_summary.py
done

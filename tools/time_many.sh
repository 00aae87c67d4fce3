#!/bin/bash
set -o pipefail
# usage: tools/time_many.sh "<flags>" ...   parity (default-pipeline tests) then C2 / veach timing of each variant; restores the default build
trap 'unset CRT_EXTRA_CXXFLAGS; python3 cudaraytracing_amd/build.py --force > /dev/null 2>&1' EXIT
for flags in "$@"; do
  export CRT_EXTRA_CXXFLAGS="$flags"; python3 cudaraytracing_amd/build.py --force > /dev/null 2>&1 || { echo "build failed: $flags"; continue; }
  echo "== $flags"
  timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -1 || { echo "parity failed: stopping"; exit 2; }
  timeout -k 10 120 python3 tools/perf_probe.py --spp 512 --reps 3 | tail -1 | cut -c1-90 || exit 3
  timeout -k 10 120 python3 tools/perf_probe.py --scene veach-mis --spp 256 --reps 2 | tail -1 | cut -c1-90 || exit 3
done

#!/bin/bash
set -o pipefail  # a crashed probe must stop the script (a GPU fault must never be followed by another GPU step)
# usage: tools/time_variant.sh "<flags>" ...   timing only (C2 and veach spp 256) of builds with extra compile flags; restores the default build
trap 'unset CRT_EXTRA_CXXFLAGS; python3 cudaraytracing_amd/build.py --force > /dev/null 2>&1' EXIT  # always leave the default build in the tree
for flags in "$@"; do
  export CRT_EXTRA_CXXFLAGS="$flags"; python3 cudaraytracing_amd/build.py --force > /dev/null 2>&1 || { echo "build failed: $flags"; continue; }
  echo "== $flags"
  timeout -k 10 120 python3 tools/perf_probe.py --spp 512 --reps 3 | tail -1 | cut -c1-90 || exit 1
  timeout -k 10 120 python3 tools/perf_probe.py --scene veach-mis --spp 256 --reps 2 | tail -1 | cut -c1-90 || exit 1
done
unset CRT_EXTRA_CXXFLAGS; python3 cudaraytracing_amd/build.py --force > /dev/null 2>&1
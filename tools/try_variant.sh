#!/bin/bash
set -o pipefail  # a crashed probe must stop the script (a GPU fault must never be followed by another GPU step)
# Builds libcrt.so with extra compile flags on the GPU box, runs the parity tests of the default pipeline and a timing probe.
# usage: tools/try_variant.sh "<flags>" [spp]
trap 'unset CRT_EXTRA_CXXFLAGS; python3 cudaraytracing_amd/build.py --force > /dev/null 2>&1' EXIT  # always leave the default build in the tree
flags="$1"; spp=${2:-512}
export CRT_EXTRA_CXXFLAGS="$flags"; python3 cudaraytracing_amd/build.py --force > /dev/null 2>&1 || { echo "build failed: $flags"; exit 1; }
echo "== $flags"
timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -1
timeout -k 10 120 python3 tools/perf_probe.py --spp $spp --reps 2 | tail -1 | cut -c1-120

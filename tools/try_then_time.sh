#!/bin/bash
set -o pipefail  # a crashed probe must stop the script (a GPU fault must never be followed by another GPU step)
# usage: tools/try_then_time.sh "<flags>"   builds with the flags, runs the parity tests of the default pipeline FIRST, and only if
# they pass times C2 and veach-mis against the default build; always leaves the default build in the tree
trap 'unset CRT_EXTRA_CXXFLAGS; python3 cudaraytracing_amd/build.py --force > /dev/null 2>&1' EXIT
flags="$1"
export CRT_EXTRA_CXXFLAGS="$flags"; python3 cudaraytracing_amd/build.py --force > /dev/null 2>&1 || { echo "build failed: $flags"; exit 1; }
echo "== $flags: parity"
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_adversarial_traversal.py -m gpu -x -q 2>&1 | tail -2 || { echo "parity failed: stopping"; exit 2; }
for rep in 1 2; do
  export CRT_EXTRA_CXXFLAGS="$flags"; python3 cudaraytracing_amd/build.py --force > /dev/null 2>&1 || exit 1
  echo "== $flags"
  timeout -k 10 120 python3 tools/perf_probe.py --spp 512 --reps 3 | tail -1 | cut -c1-90 || exit 3
  timeout -k 10 120 python3 tools/perf_probe.py --scene veach-mis --spp 256 --reps 2 | tail -1 | cut -c1-90 || exit 3
  unset CRT_EXTRA_CXXFLAGS; python3 cudaraytracing_amd/build.py --force > /dev/null 2>&1 || exit 1
  echo "== default"
  timeout -k 10 120 python3 tools/perf_probe.py --spp 512 --reps 3 | tail -1 | cut -c1-90 || exit 3
  timeout -k 10 120 python3 tools/perf_probe.py --scene veach-mis --spp 256 --reps 2 | tail -1 | cut -c1-90 || exit 3
done

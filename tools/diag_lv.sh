#!/bin/bash
set -o pipefail
# what the stack levels beyond LDS cost: stamped sections of the inner step with 3 (default) and 6 LDS levels (smaller pool)
trap 'unset CRT_EXTRA_CXXFLAGS; python3 cudaraytracing_amd/build.py --force > /dev/null 2>&1' EXIT
for flags in "-DCRT_STAMPS" "-DCRT_STAMPS -DPOOL_LV=6 -DPOOL3_P=124"; do
  export CRT_EXTRA_CXXFLAGS="$flags"; python3 cudaraytracing_amd/build.py --force > /dev/null 2>&1 || { echo "build failed: $flags"; exit 1; }
  echo "== $flags"
  python3 tools/perf_probe.py --spp 512 --reps 2 --scene cornell-box | tail -1 | python3 tools/diag_summary.py || exit 2
done

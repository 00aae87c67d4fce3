#!/bin/bash
# usage: tools/try_variants_pl.sh "P:LV ..."   builds each (pool size, LDS stack levels) variant of k_mega3 and times C2 / veach
trap 'unset CRT_EXTRA_CXXFLAGS; python3 cudaraytracing_amd/build.py --force > /dev/null 2>&1' EXIT  # always leave the default build in the tree
for v in $1; do
  P=${v%%:*}; LV=${v##*:}
  export CRT_EXTRA_CXXFLAGS="-DPOOL3_P=$P -DPOOL_LV=$LV"; python3 cudaraytracing_amd/build.py --force > /dev/null 2>&1 || { echo "build failed: $v"; continue; }
  echo "== P=$P LV=$LV"
  timeout -k 10 120 python3 tools/perf_probe.py --spp 512 --reps 2 | tail -1 | cut -c1-90 || exit 1
  timeout -k 10 120 python3 tools/perf_probe.py --scene veach-mis --spp 256 --reps 2 | tail -1 | cut -c1-90 || exit 1
done

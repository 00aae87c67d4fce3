#!/bin/bash
# A/B timing of prebuilt variants of libcrt.so on the GPU box.  usage: tools/ab.sh <tag> <lib>...   (libs built with tools/ab_build.sh)
# For each: the smoke render against the oracle (bits), then C2 (spp 512, 3 frames) and veach-mis spp 256 (3 frames), EXACT mode.
set -o pipefail
tag=$1; shift
mkdir -p gpurun_out/$tag
for lib in "$@"; do
  name=$(basename $lib .so)
  export CRT_LIB_PATH=$PWD/$lib
  timeout -k 10 200 python3 __graft_entry__.py --smoke > gpurun_out/$tag/$name.smoke.log 2>&1 || { echo "$name: smoke FAILED"; tail -3 gpurun_out/$tag/$name.smoke.log; exit 2; }
  timeout -k 10 200 python3 tools/perf_probe.py --spp 512 --reps 4 > gpurun_out/$tag/$name.c2.log 2>&1 || { echo "$name: c2 failed"; exit 3; }
  timeout -k 10 200 python3 tools/perf_probe.py --scene veach-mis --spp 256 --reps 4 > gpurun_out/$tag/$name.c3.log 2>&1 || { echo "$name: c3 failed"; exit 4; }
  python3 - "$name" gpurun_out/$tag/$name.c2.log gpurun_out/$tag/$name.c3.log <<'PY'
import json, sys
def best(f):
    v = [json.loads(l)["trace_ms"] for l in open(f) if l.startswith("{")]
    return min(v[1:]) if len(v) > 1 else v[0]
print("%-28s C2 %.2f ms   veach256 %.2f ms" % (sys.argv[1], best(sys.argv[2]), best(sys.argv[3])))
PY
done

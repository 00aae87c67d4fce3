#!/bin/bash
# A/B of ENVIRONMENT settings on the library in the tree: tools/ab_env.sh <tag> "<env A>" "<env B>" ...  (each: smoke against the oracle, C2 spp 512, veach-mis spp 256; best of 3)
set -o pipefail
tag=$1; shift
mkdir -p gpurun_out/$tag
i=0
for e in "$@"; do
  i=$((i+1)); name=e$i
  ( export $e; timeout -k 10 200 python3 __graft_entry__.py --smoke > gpurun_out/$tag/$name.smoke.log 2>&1 ) || { echo "[$e]: smoke FAILED"; tail -3 gpurun_out/$tag/$name.smoke.log; exit 2; }
  ( export $e; timeout -k 10 200 python3 tools/perf_probe.py --spp 512 --reps 4 > gpurun_out/$tag/$name.c2.log 2>&1 ) || { echo "[$e]: c2 failed"; tail -3 gpurun_out/$tag/$name.c2.log; exit 3; }
  ( export $e; timeout -k 10 200 python3 tools/perf_probe.py --scene veach-mis --spp 256 --reps 4 > gpurun_out/$tag/$name.c3.log 2>&1 ) || { echo "[$e]: c3 failed"; exit 4; }
  python3 - "$e" gpurun_out/$tag/$name.c2.log gpurun_out/$tag/$name.c3.log <<'PY'
import json, sys
def best(f):
    v = [json.loads(l)["trace_ms"] for l in open(f) if l.startswith("{")]
    return min(v[1:]) if len(v) > 1 else v[0]
print("%-28s C2 %.2f ms   veach256 %.2f ms" % (sys.argv[1], best(sys.argv[2]), best(sys.argv[3])))
PY
done

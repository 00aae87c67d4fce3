#!/bin/bash
# Memory-side counters of one C2 frame for prebuilt variants of libcrt.so (tools/ab_build.sh), one rocprofv3 --pmc pass per counter
# group (FETCH_SIZE and WRITE_SIZE apart, never with trace domains other than --kernel-trace).  usage: tools/traffic_ab.sh <tag> <lib>...
# -> gpurun_out/<tag>/<name>.json : per-launch counters of k_mega3 + kernel time
set -o pipefail
tag=$1; shift
export TMPDIR=/tmp
mkdir -p gpurun_out/$tag
groups=("FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU"
        "TCC_EA0_RDREQ_DRAM_32B_sum TCC_EA0_WRREQ_WRITE_DRAM_32B_sum TCC_EA0_WRREQ_ATOMIC_DRAM_32B_sum TCC_EA0_RDREQ_sum")
for lib in "$@"; do
  name=$(basename $lib .so)
  export CRT_LIB_PATH=$PWD/$lib
  out=gpurun_out/$tag/$name
  rm -rf $out; mkdir -p $out
  i=0
  for counters in "${groups[@]}"; do
    i=$((i+1))
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc $counters --output-format csv -d $out/pmc$i -- python3 tools/perf_probe.py --spp 512 --reps 1 > $out/pmc$i.log 2>&1 || { echo "$name: pmc pass $i failed"; tail -3 $out/pmc$i.log; exit 3; }
  done
  timeout -k 10 200 python3 tools/perf_probe.py --spp 512 --reps 3 > $out/time.log 2>&1 || exit 4
  python3 tools/pmc_summary.py $out > $out.json
  python3 - "$name" $out.json $out/time.log <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
k = next(v for n, v in d.items() if n.startswith("k_mega3"))
ms = min(json.loads(l)["trace_ms"] for l in open(sys.argv[3]) if l.startswith("{"))
traffic = (2.0 * k["FETCH_SIZE"] + k["WRITE_SIZE"]) * 1024.0
exact_r = 32.0 * k.get("TCC_EA0_RDREQ_DRAM_32B_sum", 0.0)
exact_w = 32.0 * (k.get("TCC_EA0_WRREQ_WRITE_DRAM_32B_sum", 0.0) + k.get("TCC_EA0_WRREQ_ATOMIC_DRAM_32B_sum", 0.0))
print("%-22s %.2f ms  traffic %.1f GB (fetch %.1f x2 + write %.1f)  exact 32-B units: read %.1f + write %.1f = %.1f GB (%.1f B per read request)  TCC miss %.3f  SQ_WAIT_ANY %.3f  VALU %.3g  lanes %.3f" % (
    sys.argv[1], ms, traffic / 1e9, k["FETCH_SIZE"] * 1024 / 1e9, k["WRITE_SIZE"] * 1024 / 1e9, exact_r / 1e9, exact_w / 1e9, (exact_r + exact_w) / 1e9,
    exact_r / max(1.0, k.get("TCC_EA0_RDREQ_sum", 0.0)), k["TCC_MISS_sum"] / (k["TCC_HIT_sum"] + k["TCC_MISS_sum"]),
    k["SQ_WAIT_ANY"] / k["SQ_WAVE_CYCLES"], k["SQ_INSTS_VALU"], k["SQ_THREAD_CYCLES_VALU"] / k["SQ_ACTIVE_INST_VALU"] / 64))
PY
  rm -rf $out
done

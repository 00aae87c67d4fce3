#!/bin/bash
# PMC counters of one C2 frame without and with the commit ring (CRT_COMMIT_RING_LOG2=$1, default 8): where the ring's time goes.
ring=${1:-8}
out=gpurun_out/ring_pmc; mkdir -p $out; export TMPDIR=/tmp
groups=(
"FETCH_SIZE"
"WRITE_SIZE"
"SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_WAIT_ANY"
"SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_BRANCH SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM"
"TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
"GRBM_GUI_ACTIVE GRBM_TA_BUSY"
)
for wl in base ring; do
  if [ $wl = ring ]; then export CRT_COMMIT_RING_LOG2=$ring; else unset CRT_COMMIT_RING_LOG2; fi
  i=0
  for counters in "${groups[@]}"; do
    i=$((i+1))
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc $counters --output-format csv -d $out/${wl}_pmc$i -- python3 tools/perf_probe.py --scene cornell-box --spp 512 --reps 1 > $out/${wl}_pmc$i.log 2>&1 || echo "pmc pass $wl $i failed"
  done
  mkdir -p $out/$wl && rm -rf $out/$wl/* && mv $out/${wl}_pmc* $out/$wl/ 2>/dev/null
  python3 tools/pmc_summary.py $out/$wl > $out/${wl}_summary.json
done
python3 - <<'PY'
import json
a=json.load(open('gpurun_out/ring_pmc/base_summary.json')); b=json.load(open('gpurun_out/ring_pmc/ring_summary.json'))
ka=[k for k in a if k.startswith('k_mega3')][0]; kb=[k for k in b if k.startswith('k_mega3')][0]
print(ka, '|', kb)
for c in sorted(set(a[ka])|set(b[kb])):
    x=a[ka].get(c); y=b[kb].get(c)
    if isinstance(x,(int,float)) and isinstance(y,(int,float)) and x:
        print("%-28s %14.4g %14.4g  x%.3f" % (c, x, y, y/x))
PY

#!/bin/bash
# PMC counters of one frame under two environment settings, side by side: tools/pmc_ab.sh <tag> "<env A>" "<env B>" [perf_probe args...]
# e.g. tools/pmc_ab.sh dec "CRT_DEC=0" "CRT_DEC=1" --scene cornell-box --spp 256
tag=$1; ea=$2; eb=$3; shift 3
probe=${@:---scene cornell-box --spp 256}
out=gpurun_out/pmc_ab_$tag; mkdir -p $out; export TMPDIR=/tmp
groups=(
"FETCH_SIZE"
"WRITE_SIZE"
"SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_WAIT_ANY"
"SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_BRANCH SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM"
"TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
"SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN"
"GRBM_GUI_ACTIVE GRBM_TA_BUSY"
)
for wl in a b; do
  if [ $wl = a ]; then e=$ea; else e=$eb; fi
  i=0
  for counters in "${groups[@]}"; do
    i=$((i+1))
    ( export $e; timeout -k 10 200 rocprofv3 --kernel-trace --pmc $counters --output-format csv -d $out/${wl}_pmc$i -- python3 tools/perf_probe.py $probe --reps 1 > $out/${wl}_pmc$i.log 2>&1 ) || echo "pmc pass $wl $i failed"
  done
  mkdir -p $out/$wl && rm -rf $out/$wl/* && mv $out/${wl}_pmc* $out/$wl/ 2>/dev/null
  python3 tools/pmc_summary.py $out/$wl > $out/${wl}_summary.json
done
python3 - $out "$ea" "$eb" <<'PY'
import json, sys
out = sys.argv[1]
a=json.load(open(out+'/a_summary.json')); b=json.load(open(out+'/b_summary.json'))
ka=[k for k in a if k.startswith('k_mega3')][0]; kb=[k for k in b if k.startswith('k_mega3')][0]
print("A =", sys.argv[2], ka, '| B =', sys.argv[3], kb)
for c in sorted(set(a[ka])|set(b[kb])):
    x=a[ka].get(c); y=b[kb].get(c)
    if isinstance(x,(int,float)) and isinstance(y,(int,float)) and x:
        print("%-32s %14.5g %14.5g  x%.3f" % (c, x, y, y/x))
PY

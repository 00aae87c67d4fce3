#!/bin/bash
# Memory side of L2 of k_mega3 on one C2 frame (the TA_* / TCP_* / TD_* counters of this rocprofv3 abort the profiled process on this pool
# -- signal 6 after minutes -- and are deliberately NOT collected), one rocprofv3 --pmc
# pass per group (never combined with trace domains).  Output: gpurun_out/diag_mem_<tag>/summary.json
# usage: tools/diag_memory.sh <tag> [perf_probe args]
tag=${1:-x}; shift
probe=${@:---scene cornell-box --spp 512}
out=gpurun_out/diag_mem_$tag
export TMPDIR=/tmp
mkdir -p $out
groups=(
"TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_DRAM_sum TCC_EA0_WRREQ_sum"
"GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM SQ_INSTS_LDS"
)
i=0
for counters in "${groups[@]}"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $counters --output-format csv -d $out/pmc$i -- python3 tools/perf_probe.py $probe --reps 1 > $out/pmc$i.log 2>&1 || echo "pmc pass $i failed"
done
python3 tools/pmc_summary.py $out > $out/summary.json
python3 - <<PY
import json
d=json.load(open("$out/summary.json"))
for k,v in d.items():
    if k.startswith("k_mega3"):
        print(k)
        for c,x in sorted(v.items()):
            if c!="dispatch": print("   %-44s %s"%(c,x))
PY

#!/bin/bash
# Where the vector-memory pipeline of the render kernel spends its time (round 6): TA / TCP / TD busy and stall counters of one frame.
# usage: tools/pmc_mem.sh <tag> [perf_probe args...]   -> gpurun_out/pmc_mem_<tag>/summary.txt
tag=$1; shift
probe=${@:---scene cornell-box --spp 256}
out=gpurun_out/pmc_mem_$tag; mkdir -p $out; export TMPDIR=/tmp
groups=(
"GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_BUSY_avr TA_BUSY_max TA_BUSY_min"
"TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum TA_TOTAL_WAVEFRONTS_sum"
"TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TOTAL_ACCESSES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_READ_sum TCP_TOTAL_WRITE_sum"
"TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum TCP_TCP_LATENCY_sum"
"TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_ADDR_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum"
"TCP_TAGRAM0_REQ_sum TCP_TAGRAM1_REQ_sum TCP_TAGRAM2_REQ_sum TCP_TAGRAM3_REQ_sum"
"TD_TD_BUSY_sum TD_TC_STALL_sum TD_LOAD_WAVEFRONT_sum TD_STORE_WAVEFRONT_sum TD_SPI_STALL_sum"
"SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM"
)
i=0
for counters in "${groups[@]}"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $counters --output-format csv -d $out/pmc$i -- python3 tools/perf_probe.py $probe --reps 1 > $out/pmc$i.log 2>&1 || echo "pmc pass $i failed: $(tail -2 $out/pmc$i.log)"
done
python3 tools/pmc_summary.py $out > $out/summary.json
python3 - $out <<'PY' | tee $out/summary.txt
import json, sys
d = json.load(open(sys.argv[1] + '/summary.json'))
for k, v in d.items():
    if k.startswith('k_mega3'):
        for a, b in sorted(v.items()):
            if isinstance(b, (int, float)): print("%-44s %16.6g" % (a, b))
PY

#!/bin/bash
out=$1; spp=$2
export TMPDIR=/tmp
mkdir -p $out
i=0
while read -r counters; do
  i=$((i+1))
  timeout 120 rocprofv3 --kernel-trace --pmc $counters --output-format csv -d $out/pass$i -- python3 tools/perf_probe.py --spp $spp --reps 1 > $out/log$i.txt 2>&1 || echo "pass $i ($counters) failed/timeout"
done <<'EOC'
SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_WAIT_ANY
SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_BRANCH SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_INSTS_VMEM_WR
GRBM_GUI_ACTIVE GRBM_TA_BUSY
TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum
TCC_HIT_sum TCC_MISS_sum
EOC
python3 tools/pmc_summary.py $out > $out/summary.json

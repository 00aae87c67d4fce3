#!/bin/bash
# Collects the judged profile artefacts on the GPU box into gpurun_out/profiles_<tag>/:
#   kernel stats of the default bench.py command, and HBM traffic (FETCH_SIZE / WRITE_SIZE in
#   separate passes, as MI355X_MICROARCH.md prescribes) + SQ counters of a full-size frame.
tag=${1:-r01}
out=gpurun_out/profiles_$tag
export TMPDIR=/tmp
mkdir -p $out
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $out/bench_stats.log 2>&1 || echo "stats pass failed"
i=0
while read -r counters; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $counters --output-format csv -d $out/pmc$i -- python3 tools/perf_probe.py --spp 512 --reps 1 > $out/pmc$i.log 2>&1 || echo "pmc pass $i failed"
done <<'EOC'
FETCH_SIZE
WRITE_SIZE
SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_WAIT_ANY
SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_BRANCH SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA
TCC_HIT_sum TCC_MISS_sum
GRBM_GUI_ACTIVE GRBM_TA_BUSY
EOC
python3 tools/pmc_summary.py $out > $out/summary.json
grep -h '"metric"' $out/bench_stats.log | tail -1 > $out/bench_line.json
ls $out

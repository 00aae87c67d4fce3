#!/bin/bash
# Collects the judged profile artefacts on the GPU box into gpurun_out/profiles_<tag>/:
#   stats/   rocprofv3 --kernel-trace --stats of bench.py (--no-large-scene: the 102 412-triangle frames of the default command run the
#            same instantiation of k_mega3 as the headline frames and would enter its average; --no-c3 likewise)
#   c2_*/    PMC passes of one full-size C2 frame (cornell-box 800x600 spp 512), one --pmc group per run, FETCH_SIZE and
#            WRITE_SIZE in separate passes as MI355X_MICROARCH.md prescribes (never combined with trace domains)
#   c3_*/    the same groups on one C3 frame (veach-mis 800x600 spp 1024)
#   pmc_latest.json  per-launch counters of both workloads STAMPED with the hash of the kernel sources + build flags
#                    (cudaraytracing_amd.build.source_hash); bench.py drops the numbers when the hash differs
# usage: tools/collect_profiles.sh <tag> [valu_cycles_per_instr]
tag=${1:-r02}
cyc=${2:-2.0}
out=gpurun_out/profiles_$tag
export TMPDIR=/tmp
mkdir -p $out
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-c3 --no-large-scene > $out/bench_stats.log 2>&1 || echo "stats pass failed"
groups=(
"FETCH_SIZE"
"WRITE_SIZE"
"SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_WAIT_ANY"
"SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_BRANCH SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM"
"TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
"SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS"
"GRBM_GUI_ACTIVE GRBM_TA_BUSY"
"SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INST_CYCLES_VALU"
"TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_32B_sum"
"TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_DRAM_sum"
"TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_WRITE_DRAM_32B_sum TCC_EA0_WRREQ_ATOMIC_DRAM_32B_sum"
"TA_TA_BUSY_sum TA_BUSY_avr TA_BUSY_max"
"TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum TCP_TCP_LATENCY_sum"
"TCP_TAGRAM0_REQ_sum TCP_TAGRAM1_REQ_sum TCP_TAGRAM2_REQ_sum TCP_TAGRAM3_REQ_sum"
)
# (round 6: the vector-memory pipe -- texture addresser busy cycles, the L1's requests to L2 and their latency, its tag lookups.  Groups of
# TA / TCP stall counters made rocprofv3 abort on this pool (tools/pmc_mem.sh, four of eight passes): only the three that ran are collected)
# (round 5: the L2's fabric-side request counters by size -- FETCH_SIZE / WRITE_SIZE are derived from them with one assumed size; the
# _32B forms count 32-byte units whatever the request's size, i.e. exact bytes.  "DRAM" = destined for the device's own memory, which
# includes what the Infinity Cache in front of it answers: this rocprofv3 exposes no counter behind that cache, rocprofv3 --list-avail)
for wl in c2 c3; do
  if [ $wl = c2 ]; then probe="--scene cornell-box --spp 512"; else probe="--scene veach-mis --spp 1024"; fi
  i=0
  for counters in "${groups[@]}"; do
    i=$((i+1))
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc $counters --output-format csv -d $out/${wl}_pmc$i -- python3 tools/perf_probe.py $probe --reps 1 > $out/${wl}_pmc$i.log 2>&1 || echo "pmc pass $wl $i failed"
  done
  mkdir -p $out/$wl && rm -rf $out/$wl/* && mv $out/${wl}_pmc* $out/$wl/ 2>/dev/null
  python3 tools/pmc_summary.py $out/$wl > $out/${wl}_summary.json
done
python3 tools/pmc_summary.py $out/stats > $out/stats_summary.json
grep -h '"metric"' $out/bench_stats.log | tail -1 > $out/bench_line.json
python3 tools/make_pmc_json.py $out $cyc > $out/pmc_latest.json
ls $out

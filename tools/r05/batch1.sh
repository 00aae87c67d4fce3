#!/bin/bash
# Round 5, GPU session 1: the hand-off reproducer, the LDS gather microbenchmark, the counters this rocprofv3 knows, the hand-off variants of
# the library against tests/test_first_frame.py, and the first A/B batch of kernel variants (tools/ab.sh).
set -o pipefail
out=gpurun_out/r05_b1; mkdir -p $out
hipcc -O2 --offload-arch=gfx950 tools/handoff_repro.hip -o /tmp/handoff_repro && timeout -k 10 300 /tmp/handoff_repro 40 > $out/handoff_repro.jsonl 2>&1; echo "repro rc=$?"
grep -c stale_words $out/handoff_repro.jsonl; grep -v '"stale_words": 0,' $out/handoff_repro.jsonl | head -20
hipcc -O2 --offload-arch=gfx950 tools/lds_gather_bench.hip -o /tmp/lds_gather_bench && timeout -k 10 120 /tmp/lds_gather_bench > $out/lds_gather.jsonl 2>&1; echo "lds rc=$?"; cat $out/lds_gather.jsonl
export TMPDIR=/tmp
(cd /tmp && timeout -k 10 120 rocprofv3 --list-avail > $OLDPWD/$out/counters_avail.txt 2>&1); grep -c . $out/counters_avail.txt
grep -o "TCC_EA[0-9A-Z_a-z]*\(DRAM\|HBM\|IO\|GMI\)[0-9A-Z_a-z]*" $out/counters_avail.txt | sort -u | head -40
for v in hp hpi hpis hps hpl v0; do
  for rep in 1 2; do
    CRT_LIB_PATH=$PWD/cudaraytracing_amd/lib/ab/$v.so timeout -k 10 300 python3 -m pytest tests/test_first_frame.py -q -x -m gpu -k first_frames > $out/ff_${v}_$rep.log 2>&1
    echo "first_frame $v run $rep: rc=$? $(grep -E 'passed|failed' $out/ff_${v}_$rep.log | tail -1) $(grep -o 'render [0-9]*, frame [0-9]*: [0-9]* pixels differ[^)]*)' $out/ff_${v}_$rep.log | head -1)"
  done
done
tools/ab.sh r05_b1_ab cudaraytracing_amd/lib/ab/v0.so cudaraytracing_amd/lib/ab/v0p.so cudaraytracing_amd/lib/ab/v1.so cudaraytracing_amd/lib/ab/v2.so cudaraytracing_amd/lib/ab/v3.so cudaraytracing_amd/lib/ab/v0.so

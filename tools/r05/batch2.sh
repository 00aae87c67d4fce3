#!/bin/bash
# Round 5, GPU session 2: the whole GPU suite on the new default (mask form of the inner step + stack rings), the second A/B batch, the
# first-frame stress under the hand-off experiment builds, the share probe of C2 (does a stack ring lengthen the end of a launch?), and
# a basic-block profile of the new default.
set -o pipefail
out=gpurun_out/r05_b2; mkdir -p $out
timeout -k 10 1000 python3 -m pytest tests -m gpu -q -x > $out/tests.log 2>&1; rc=$?
tail -4 $out/tests.log
[ $rc -eq 0 ] || { echo "pytest rc=$rc"; grep -E "^(FAILED|ERROR)" $out/tests.log | head; }
for v in hp hpi hps hpl b0; do
  CRT_DEBUG_FILL=255 CRT_LIB_PATH=$PWD/cudaraytracing_amd/lib/ab/$v.so timeout -k 10 300 python3 tests/first_frame_stress_driver.py 200 > $out/stress_$v.json 2> $out/stress_$v.err || echo "stress $v failed"
  echo "stress $v: $(tail -1 $out/stress_$v.json | cut -c1-300)"
done
tools/ab.sh r05_b2_ab cudaraytracing_amd/lib/ab/b0.so cudaraytracing_amd/lib/ab/b1.so cudaraytracing_amd/lib/ab/b2.so cudaraytracing_amd/lib/ab/b3.so cudaraytracing_amd/lib/ab/b4.so cudaraytracing_amd/lib/ab/b5.so cudaraytracing_amd/lib/ab/b6.so cudaraytracing_amd/lib/ab/b7.so cudaraytracing_amd/lib/ab/b8.so cudaraytracing_amd/lib/ab/b0.so
timeout -k 10 300 python3 tools/share_probe.py --workload c2 > $out/share_c2.jsonl 2> $out/share_c2.err; cat $out/share_c2.jsonl | cut -c1-400
tools/bbprof/run.sh r05_b2_bb 256 64 && python3 tools/bbprof/census.py gpurun_out/r05_b2_bb --c2-spp 256 --c3-spp 64 --out $out/bbprof.json > $out/census.log 2>&1; tail -3 $out/census.log
python3 tools/bbprof/report.py gpurun_out/r05_b2_bb/k_mega3_bb.json gpurun_out/r05_b2_bb/c2.txt > $out/bbprof_c2_report.txt 2>&1; head -12 $out/bbprof_c2_report.txt

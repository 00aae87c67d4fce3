#!/bin/bash
# Round 5, GPU session 3: the whole GPU suite (new: leaf-queue overflow variant, 200 first frames, four self-started ranks at C4's size, the
# AUTO-gather record), the first-frame stress under the hand-off experiment builds, the one-record leaf step A/B.
set -o pipefail
out=gpurun_out/r05_b3; mkdir -p $out
timeout -k 10 1100 python3 -m pytest tests -m gpu -q > $out/tests.log 2>&1; rc=$?
tail -4 $out/tests.log
[ $rc -eq 0 ] || { echo "pytest rc=$rc"; grep -E "^(FAILED|ERROR)" $out/tests.log | head; }
for v in hp hpi hps hpl c0; do
  CRT_DEBUG_FILL=255 CRT_LIB_PATH=$PWD/cudaraytracing_amd/lib/ab/$v.so timeout -k 10 300 python3 tests/first_frame_stress_driver.py 200 > $out/stress_$v.json 2> $out/stress_$v.err || echo "stress $v failed"
  echo "stress $v: $(tail -1 $out/stress_$v.json | cut -c1-300)"
done
tools/ab.sh r05_b3_ab cudaraytracing_amd/lib/ab/c0.so cudaraytracing_amd/lib/ab/c1.so cudaraytracing_amd/lib/ab/c0.so cudaraytracing_amd/lib/ab/c1.so

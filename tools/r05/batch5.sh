#!/bin/bash
# Round 5, GPU session 5: the 6- and 8-wide trees of the decoupled kernels (experiment builds) -- parity first (the smoke frame, the
# decoupled-leaves, adversarial-traversal and parity tests under each library), then the A/B.
set -o pipefail
out=gpurun_out/r05_b5; mkdir -p $out
for v in w6 w8; do
  CRT_LIB_PATH=$PWD/cudaraytracing_amd/lib/ab/$v.so timeout -k 10 900 python3 -m pytest tests/test_decoupled_leaves.py tests/test_adversarial_traversal.py tests/test_gpu_parity.py tests/test_large_mesh.py -m gpu -q -x > $out/tests_$v.log 2>&1
  echo "$v tests rc=$? $(tail -1 $out/tests_$v.log)"
done
tools/ab.sh r05_b5_ab cudaraytracing_amd/lib/ab/e0.so cudaraytracing_amd/lib/ab/w6.so cudaraytracing_amd/lib/ab/w8.so cudaraytracing_amd/lib/ab/e0.so cudaraytracing_amd/lib/ab/w6.so cudaraytracing_amd/lib/ab/w8.so
for v in e0 w6 w8; do
  CRT_LIB_PATH=$PWD/cudaraytracing_amd/lib/ab/$v.so timeout -k 10 200 python3 tools/batch_probe.py > $out/batch_$v.json 2>&1; echo "$v: $(tail -1 $out/batch_$v.json | cut -c1-600)"
done

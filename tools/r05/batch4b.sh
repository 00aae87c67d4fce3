#!/bin/bash
# Round 5, GPU session 4b: the rest of session 4 (the vn-plane build had not compiled): A/B + fabric traffic of the memory-side experiments.
set -o pipefail
out=gpurun_out/r05_b4; mkdir -p $out
tools/ab.sh r05_b4_ab cudaraytracing_amd/lib/ab/d0.so cudaraytracing_amd/lib/ab/d2.so cudaraytracing_amd/lib/ab/d3.so cudaraytracing_amd/lib/ab/d4.so cudaraytracing_amd/lib/ab/d5.so cudaraytracing_amd/lib/ab/d6.so cudaraytracing_amd/lib/ab/d0.so
tools/traffic_ab.sh r05_b4_traffic cudaraytracing_amd/lib/ab/d0.so cudaraytracing_amd/lib/ab/d2.so cudaraytracing_amd/lib/ab/d4.so cudaraytracing_amd/lib/ab/d5.so | tee $out/traffic_b.txt

#!/bin/bash
# Round 5, GPU session 6 (host-side): does the speculative tree optimisation scale with threads on the GPU box's host?
set -o pipefail
out=gpurun_out/r05_b6; mkdir -p $out
g++ -O2 -std=c++17 -pthread -Iinclude -Icudaraytracing_amd/csrc tools/sah_opt_bench.cpp -Lcudaraytracing_amd/lib -lcrt -Wl,-rpath,$PWD/cudaraytracing_amd/lib -o /tmp/sah_opt_bench || exit 1
nproc
for n in 1 2 4 8 16; do echo "threads $n: $(CRT_SAH_OPT_THREADS=$n /tmp/sah_opt_bench scenes/cornell-box/config.json 5 | tr '\n' ' ' | cut -c1-700)"; done | tee $out/sah_threads.txt
python3 tools/sah_probe.py 2>&1 | tail -5 | tee $out/sah_probe.txt

#!/bin/bash
# Round 6 soak on the final sources: CRT_TRAVERSAL_EXACT through the default pool (decoupled leaves on the tree without rows of refs, wave-mask
# steps, pipelined second visit), through the decoupled pool on the tree WITH rows of refs and through the coupled pool, against the exhaustive
# CRT_TRAVERSAL_REFERENCE on full-size C5 shares (2 seeds x 8 ranks) and C4 shares (4 seeds x 8 ranks).
set -o pipefail
mkdir -p gpurun_out/r06_soak
timeout -k 10 1100 python3 tools/soak_fast_vs_reference.py --mode exact --forms default,refs,coupled --scene veach-mis --width 1920 --height 1080 --spp 4096 --ranks 8 --seeds 0 1 --out gpurun_out/r06_soak/c5.jsonl > gpurun_out/r06_soak/c5.log 2>&1 || { echo c5 failed; tail -5 gpurun_out/r06_soak/c5.log; exit 2; }
tail -2 gpurun_out/r06_soak/c5.log | cut -c1-400
timeout -k 10 600 python3 tools/soak_fast_vs_reference.py --mode exact --forms default,refs,coupled --scene cornell-box --width 3840 --height 2160 --spp 256 --ranks 8 --seeds 0 1 2 3 --out gpurun_out/r06_soak/c4.jsonl > gpurun_out/r06_soak/c4.log 2>&1 || { echo c4 failed; tail -5 gpurun_out/r06_soak/c4.log; exit 3; }
tail -2 gpurun_out/r06_soak/c4.log | cut -c1-400

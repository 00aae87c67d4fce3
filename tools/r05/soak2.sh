#!/bin/bash
# Round 5, extended soak of the final sources: four more seeds of the full-size C5 shares (EXACT through both pool forms against REFERENCE).
mkdir -p gpurun_out/r05_soak
timeout -k 10 1100 python3 tools/soak_fast_vs_reference.py --mode exact --forms default,coupled --scene veach-mis --width 1920 --height 1080 --spp 4096 --ranks 8 --seeds 2 3 4 5 --out gpurun_out/r05_soak/c5_seeds2to5.jsonl > gpurun_out/r05_soak/c5b.log 2>&1 || { echo c5 failed; tail -5 gpurun_out/r05_soak/c5b.log; exit 2; }
tail -1 gpurun_out/r05_soak/c5b.log

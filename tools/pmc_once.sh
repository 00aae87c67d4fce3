#!/bin/bash
# One rocprofv3 --pmc pass of one frame: tools/pmc_once.sh <tag> "<counters>" [perf_probe args...]; prints the k_mega3 rows
tag=$1; counters=$2; shift 2
probe=${@:---scene cornell-box --spp 256}
out=gpurun_out/pmc_once_$tag; mkdir -p $out; export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --pmc $counters --output-format csv -d $out -- python3 tools/perf_probe.py $probe --reps 1 > $out/log.txt 2>&1 || { echo "pass failed"; tail -5 $out/log.txt; exit 1; }
python3 tools/pmc_summary.py $out | python3 -c "
import json, sys
d = json.load(sys.stdin)
for k, v in d.items():
    if k.startswith('k_mega3'):
        print(k, {a: b for a, b in v.items() if isinstance(b, (int, float))})
"

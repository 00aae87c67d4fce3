#!/bin/bash
# One GPU-box session: the whole -m gpu suite, then (only if pytest ended by itself) the issue microbenchmark and the default bench line.
# usage: tools/gpu_round.sh <tag> [pytest args...]
tag=${1:-r02}; shift
mkdir -p gpurun_out
timeout -k 10 1000 python3 -m pytest tests -m gpu -q "$@" > gpurun_out/${tag}_tests.log 2>&1
rc=$?
tail -5 gpurun_out/${tag}_tests.log
[ $rc -eq 0 ] || { echo "pytest rc=$rc: stopping"; exit $rc; }
hipcc -O2 --offload-arch=gfx950 tools/valu_issue_bench.hip -o /tmp/valu_issue_bench && timeout -k 10 120 /tmp/valu_issue_bench > gpurun_out/${tag}_valu_issue.json || { echo "valu bench failed"; exit 3; }
timeout -k 10 400 python3 bench.py > gpurun_out/${tag}_bench.log 2>&1 || { echo "bench failed"; tail -20 gpurun_out/${tag}_bench.log; exit 4; }
tail -1 gpurun_out/${tag}_bench.log

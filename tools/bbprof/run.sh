#!/bin/bash
# On the GPU box: the smoke render through the instrumented kernel (must equal the oracle bit for bit), then the block counts of a
# C2 / C3 slice.  usage: tools/bbprof/run.sh <tag> [spp]
set -o pipefail
tag=${1:-bb}; spp=${2:-8}
co=tools/bbprof/out/k_mega3_bb.co
# (always rebuilt: a code object left from an earlier state of the sources travels to the box with the snapshot and would be profiled in
# place of the kernel in the tree -- it renders the same frame, so nothing else notices; BBPROF_KEEP=1 reuses it)
if [ "${BBPROF_KEEP:-0}" != 1 ] || [ ! -f $co ]; then rm -rf tools/bbprof/out; tools/bbprof/build_co.sh > /dev/null || exit 1; fi
mkdir -p gpurun_out/$tag; cp tools/bbprof/out/k_mega3_bb.json gpurun_out/$tag/
export CRT_BBPROF_CO=$co
CRT_BBPROF_OUT=gpurun_out/$tag/smoke.txt timeout -k 10 300 python3 __graft_entry__.py --smoke > gpurun_out/$tag/smoke.log 2>&1 || { echo "instrumented smoke failed"; tail -5 gpurun_out/$tag/smoke.log; exit 2; }
tail -1 gpurun_out/$tag/smoke.log
CRT_BBPROF_OUT=gpurun_out/$tag/c2.txt timeout -k 10 300 python3 tools/perf_probe.py --spp $spp --reps 1 > gpurun_out/$tag/c2.log 2>&1 || { echo "c2 failed"; tail -5 gpurun_out/$tag/c2.log; exit 3; }
tail -1 gpurun_out/$tag/c2.log
CRT_BBPROF_OUT=gpurun_out/$tag/c3.txt timeout -k 10 300 python3 tools/perf_probe.py --scene veach-mis --spp ${3:-$spp} --reps 1 > gpurun_out/$tag/c3.log 2>&1 || { echo "c3 failed"; tail -5 gpurun_out/$tag/c3.log; exit 4; }
tail -1 gpurun_out/$tag/c3.log

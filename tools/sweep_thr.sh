#!/bin/bash
# Sweeps the k_mega3 phase thresholds (env CRT_THR_*), prints ms per frame at the given spp.
spp=${1:-128}; shift
for cfg in "64 48 48 48 48" "64 48 32 32 32" "64 48 24 24 24" "64 48 16 16 16" "64 56 32 32 32" "64 40 32 32 32" "56 48 32 32 32" "64 64 32 32 32" "64 32 16 16 16" "48 32 24 24 24"; do
  set -- $cfg
  r=$(CRT_THR_INNER=$1 CRT_THR_LEAF=$2 CRT_THR_LA=$3 CRT_THR_LB=$4 CRT_THR_LC=$5 python3 tools/perf_probe.py --spp $spp --reps 2 | tail -1 | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['trace_ms'])")
  echo "inner $1 leaf $2 LA $3 LB $4 LC $5 : $r ms"
done

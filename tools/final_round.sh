#!/bin/bash
# One GPU-box session that produces every judged artefact of a round from the sources in the tree:
#   tests -> basic-block profile + census -> PMC passes -> share probes of C2 .. C5 -> bench line (which then finds profiles stamped with
#   its own source hash).  Stops at the first step that fails -- a failing test suite included: nothing is copied into profiles/*_latest.json
#   from code whose tests are red.
# usage: tools/final_round.sh <tag>       outputs under gpurun_out/<tag>/ and gpurun_out/profiles_<tag>/
set -o pipefail
tag=${1:-r04}
mkdir -p gpurun_out/$tag
timeout -k 10 900 python3 -m pytest tests -m gpu -q > gpurun_out/$tag/tests.log 2>&1; rc=$?
tail -3 gpurun_out/$tag/tests.log
[ $rc -eq 0 ] || { echo "pytest rc=$rc: stopping (no profile of this code is kept)"; exit $rc; }
tools/bbprof/run.sh ${tag}_bb 256 64 || exit 3
python3 tools/bbprof/census.py gpurun_out/${tag}_bb --c2-spp 256 --c3-spp 64 --out gpurun_out/$tag/bbprof_latest.json > gpurun_out/$tag/census.log 2>&1 || { tail -5 gpurun_out/$tag/census.log; exit 4; }
cp gpurun_out/$tag/bbprof_latest.json profiles/bbprof_latest.json
tools/collect_profiles.sh $tag > gpurun_out/$tag/collect.log 2>&1 || { tail -5 gpurun_out/$tag/collect.log; exit 5; }
tail -2 gpurun_out/$tag/collect.log
cp gpurun_out/profiles_$tag/pmc_latest.json profiles/pmc_latest.json
: > gpurun_out/$tag/share_probe.jsonl
for wl in c2 c3 c4 c5; do
  timeout -k 10 400 python3 tools/share_probe.py --workload $wl >> gpurun_out/$tag/share_probe.jsonl 2>gpurun_out/$tag/share_probe_$wl.err || { echo "share probe $wl failed"; tail -5 gpurun_out/$tag/share_probe_$wl.err; exit 6; }
done
cat gpurun_out/$tag/share_probe.jsonl
python3 tools/share_model.py gpurun_out/$tag/share_probe.jsonl > gpurun_out/$tag/share_model.json && cp gpurun_out/$tag/share_model.json profiles/share_model.json
timeout -k 10 600 python3 bench.py > gpurun_out/$tag/bench.log 2>&1 || { echo "bench failed"; tail -20 gpurun_out/$tag/bench.log; exit 7; }
tail -1 gpurun_out/$tag/bench.log | cut -c1-400
CRT_BENCH_ONE_DEVICE=1 timeout -k 10 300 python3 bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline --no-c3 > gpurun_out/$tag/bench_2ranks_one_device.log 2>&1 || { echo "2-rank rehearsal failed"; tail -20 gpurun_out/$tag/bench_2ranks_one_device.log; exit 8; }
tail -1 gpurun_out/$tag/bench_2ranks_one_device.log | python3 -c "import sys, json; d = json.loads(sys.stdin.read()); print('2 ranks on one device:', d['ms_per_step'], d.get('per_rank'))"

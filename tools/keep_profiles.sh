#!/bin/bash
# Copies what tools/final_round.sh <tag> left under gpurun_out/ (merged back from the GPU box) into the tracked profiles/<tag>/ and the
# hash-stamped profiles/*_latest.json that bench.py reads.  usage: tools/keep_profiles.sh <tag>
set -e
tag=$1; src=gpurun_out; dst=profiles/$tag
mkdir -p $dst
cp $src/${tag}_bb/c2_report.txt $dst/bbprof_c2_report.txt
cp $src/${tag}_bb/c3_report.txt $dst/bbprof_c3_report.txt
cp $src/${tag}_bb/c2.txt $dst/bbprof_c2_spp256_counts.txt
cp $src/${tag}_bb/c3.txt $dst/bbprof_c3_spp64_counts.txt
cp $src/$tag/bbprof_latest.json profiles/bbprof_latest.json
cp $src/profiles_$tag/pmc_latest.json profiles/pmc_latest.json
cp $src/$tag/share_model.json profiles/share_model.json
cp $src/$tag/share_probe.jsonl $dst/share_probe.jsonl
cp $src/profiles_$tag/c2_summary.json $dst/c2_pmc_summary.json
cp $src/profiles_$tag/c3_summary.json $dst/c3_pmc_summary.json
cp $src/profiles_$tag/stats_summary.json $dst/bench_stats_summary.json
cp $src/profiles_$tag/bench_line.json $dst/bench_line_profiled.json
f=$(ls $src/profiles_$tag/stats/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $dst/bench_kernel_stats.csv
grep -h '"metric"' $src/$tag/bench.log | tail -1 > $dst/bench_line.json
grep -h '"metric"' $src/$tag/bench_2ranks_one_device.log | tail -1 > $dst/bench_line_2ranks_one_device.json
tail -3 $src/$tag/tests.log > $dst/gpu_tests_tail.txt
ls -la $dst

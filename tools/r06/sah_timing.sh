# set-up timing on the GPU box's host: sections of the device SAH build, the optimisation pass by thread count, and the frame times of the
# batched pass's tree against the serial pass's (round 6)
export TMPDIR=/tmp
out=gpurun_out/sah_timing; mkdir -p $out
CRT_SAH_TIMING=1 python3 tools/sah_probe.py > $out/sections.txt 2>&1
g++ -O2 -std=c++17 -pthread -Iinclude -Icudaraytracing_amd/csrc tools/sah_opt_bench.cpp -Lcudaraytracing_amd/lib -lcrt -Wl,-rpath,$PWD/cudaraytracing_amd/lib -o /tmp/sah_opt_bench || exit 1
for t in 1 2 4 8 16; do echo "threads $t: $(CRT_SAH_OPT_THREADS=$t /tmp/sah_opt_bench scenes/cornell-box/config.json 7 | tr '\n' ' ')"; done > $out/opt_threads.txt
echo "veach-mis default threads: $(/tmp/sah_opt_bench scenes/veach-mis/config.json 7 | tr '\n' ' ')" >> $out/opt_threads.txt
bash tools/ab_env.sh sah_timing/ab "CRT_SAH_OPT_FORM=batched" "CRT_SAH_OPT_FORM=serial" "CRT_SAH_OPT_FORM=batched" "CRT_SAH_OPT_FORM=serial" > $out/ab.txt 2>&1
cat $out/sections.txt $out/opt_threads.txt $out/ab.txt

#!/bin/bash
# Round 5, GPU session 4: what the first HIP calls cost with and without libcrt.so in the process; the first-frame stress under the
# hand-off experiment builds WITHOUT the allocation fill (a memset sweeps the caches); A/B of the memory-side experiments (vertex records
# streamed past L2, the vn plane dropped, smaller pools) with their fabric traffic.
set -o pipefail
out=gpurun_out/r05_b4; mkdir -p $out
hipcc -O2 --offload-arch=gfx950 tools/init_probe.cpp -o /tmp/init_probe_plain 2>/dev/null
hipcc -O2 --offload-arch=gfx950 -DWITH_LIBCRT tools/init_probe.cpp -Iinclude -Lcudaraytracing_amd/lib -lcrt -Wl,-rpath,$PWD/cudaraytracing_amd/lib -o /tmp/init_probe_crt 2>/dev/null
for i in 1 2 3; do /tmp/init_probe_plain; /tmp/init_probe_crt; done | tee $out/init_probe.jsonl
for v in hp hps hpl; do
  CRT_LIB_PATH=$PWD/cudaraytracing_amd/lib/ab/$v.so timeout -k 10 300 python3 tests/first_frame_stress_driver.py 300 > $out/stress_nofill_$v.json 2> $out/stress_nofill_$v.err || echo "stress $v failed"
  echo "stress (no fill) $v: $(tail -1 $out/stress_nofill_$v.json | cut -c1-300)"
done
tools/ab.sh r05_b4_ab cudaraytracing_amd/lib/ab/d0.so cudaraytracing_amd/lib/ab/d1.so cudaraytracing_amd/lib/ab/d2.so cudaraytracing_amd/lib/ab/d3.so cudaraytracing_amd/lib/ab/d4.so cudaraytracing_amd/lib/ab/d5.so cudaraytracing_amd/lib/ab/d6.so cudaraytracing_amd/lib/ab/d0.so
tools/traffic_ab.sh r05_b4_traffic cudaraytracing_amd/lib/ab/d0.so cudaraytracing_amd/lib/ab/d1.so cudaraytracing_amd/lib/ab/d2.so cudaraytracing_amd/lib/ab/d3.so cudaraytracing_amd/lib/ab/d4.so | tee $out/traffic.txt

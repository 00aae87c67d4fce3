#!/bin/bash
# Round 5, GPU session 9: does the size of libcrt.so's code objects show in the runtime's start?  tools/init_probe.cpp against the library
# as it is (fifty instantiations of k_mega3, 4.5 MB) and against a build that holds ONE instantiation (-DCRT_ASM_ONLY_DEFAULT, 2.4 MB).
out=gpurun_out/r05_b9; mkdir -p $out /tmp/small_lib
cp cudaraytracing_amd/lib/ab/small.so /tmp/small_lib/libcrt.so
hipcc -O2 --offload-arch=gfx950 tools/init_probe.cpp -o /tmp/init_probe_plain 2>/dev/null
hipcc -O2 --offload-arch=gfx950 -DWITH_LIBCRT tools/init_probe.cpp -Iinclude -Lcudaraytracing_amd/lib -lcrt -Wl,-rpath,$PWD/cudaraytracing_amd/lib -o /tmp/init_probe_crt 2>/dev/null
hipcc -O2 --offload-arch=gfx950 -DWITH_LIBCRT tools/init_probe.cpp -Iinclude -L/tmp/small_lib -lcrt -Wl,-rpath,/tmp/small_lib -o /tmp/init_probe_small 2>/dev/null
for i in 1 2 3 4; do /tmp/init_probe_plain; /tmp/init_probe_crt; /tmp/init_probe_small | sed 's/"libcrt_linked": true/"libcrt_linked": "one instantiation"/'; done | tee $out/init_probe_small.jsonl

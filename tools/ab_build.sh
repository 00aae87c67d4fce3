#!/bin/bash
# Builds a variant of libcrt.so next to the default one: tools/ab_build.sh <name> [extra compiler flags...] -> cudaraytracing_amd/lib/ab/<name>.so
# Every translation unit that sees the render kernel's headers (crt_mega3.h / crt_path.h: crt_mega3, crt_wavefront, crt_frame, crt_render)
# is recompiled with the extra flags into a private object directory, so that a -D which changes a shared layout (MParams3, LParams,
# pool sizes, LEAF_REC_MAX ...) cannot produce a library whose units disagree; the host-only units (loader, tree builders,
# multi-device entry) are the default build's.  AB_UNITS="crt_mega3.hip" restricts the recompilation for flags known to be local.
set -e
root=$(cd "$(dirname "$0")/.." && pwd); name=$1; shift
mkdir -p $root/cudaraytracing_amd/lib/ab
python3 -c "import sys; sys.path.insert(0, '$root'); from cudaraytracing_amd import build as b; b.build_lib()"
flags=$(python3 -c "import sys; sys.path.insert(0, '$root'); from cudaraytracing_amd import build as b; print(' '.join(b.COMMON + b.DEVICE))")
dev=$(python3 -c "import sys; sys.path.insert(0, '$root'); from cudaraytracing_amd import build as b; print(' '.join(b.DEVICE))")
obj=$root/cudaraytracing_amd/lib/obj
priv=$root/cudaraytracing_amd/lib/obj_ab/$name
rm -rf $priv; mkdir -p $priv
units=${AB_UNITS:-"crt_mega3.hip crt_wavefront.hip crt_frame.hip crt_render.hip"}
pids=()
for u in $units; do
  /opt/rocm/bin/hipcc $flags "$@" -c $root/cudaraytracing_amd/csrc/$u -o $priv/${u%.*}.o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
others=""
for o in $obj/*.o; do
  b=$(basename $o)
  [ -f $priv/$b ] || others="$others $o"
done
/opt/rocm/bin/hipcc $dev -shared -fPIC $priv/*.o $others -ldl -lpthread -o $root/cudaraytracing_amd/lib/ab/$name.so
rm -rf $priv
echo cudaraytracing_amd/lib/ab/$name.so

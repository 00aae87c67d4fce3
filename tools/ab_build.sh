#!/bin/bash
# Builds a variant of libcrt.so next to the default one: tools/ab_build.sh <name> [extra compiler flags...] -> cudaraytracing_amd/lib/ab/<name>.so
# Only the render kernel (crt_mega3.hip) is recompiled with the extra flags; the other objects are the default build's (lib/obj/).
set -e
root=$(cd "$(dirname "$0")/.." && pwd); name=$1; shift
mkdir -p $root/cudaraytracing_amd/lib/ab
python3 -c "import sys; sys.path.insert(0, '$root'); from cudaraytracing_amd import build as b; b.build_lib()"
flags=$(python3 -c "import sys; sys.path.insert(0, '$root'); from cudaraytracing_amd import build as b; print(' '.join(b.COMMON + b.DEVICE))")
dev=$(python3 -c "import sys; sys.path.insert(0, '$root'); from cudaraytracing_amd import build as b; print(' '.join(b.DEVICE))")
obj=$root/cudaraytracing_amd/lib/obj
/opt/rocm/bin/hipcc $flags "$@" -c $root/cudaraytracing_amd/csrc/crt_mega3.hip -o $obj/ab_$name.o
others=$(ls $obj/*.o | grep -v "/ab_" | grep -v "/crt_mega3.o")
/opt/rocm/bin/hipcc $dev -shared -fPIC $obj/ab_$name.o $others -ldl -lpthread -o $root/cudaraytracing_amd/lib/ab/$name.so
rm -f $obj/ab_$name.o
echo cudaraytracing_amd/lib/ab/$name.so

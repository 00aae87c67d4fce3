#!/bin/bash
# Builds a variant of libcrt.so next to the default one: tools/ab_build.sh <name> [extra compiler flags...] -> cudaraytracing_amd/lib/ab/<name>.so
set -e
root=$(cd "$(dirname "$0")/.." && pwd); name=$1; shift
mkdir -p $root/cudaraytracing_amd/lib/ab
flags=$(python3 -c "import sys; sys.path.insert(0, '$root'); from cudaraytracing_amd import build as b; print(' '.join(b.COMMON + b.DEVICE))")
src=$(python3 -c "import sys; sys.path.insert(0, '$root'); from cudaraytracing_amd import build as b; import os; print(' '.join(os.path.join(b.CSRC, s) for s in b.LIB_SOURCES))")
/opt/rocm/bin/hipcc $flags "$@" -shared $src -ldl -lpthread -o $root/cudaraytracing_amd/lib/ab/$name.so
echo cudaraytracing_amd/lib/ab/$name.so

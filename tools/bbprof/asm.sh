#!/bin/bash
# The compiler's annotated assembly of crt_mega3.hip (the render kernel) with the production flags (+ CRT_EXTRA_CXXFLAGS): $1 = output .s
root=$(cd "$(dirname "$0")/../.." && pwd)
flags=$(python3 -c "import sys; sys.path.insert(0, '$root'); from cudaraytracing_amd import build as b; print(' '.join(b.COMMON + b.DEVICE))")
rm -f "$1"; /opt/rocm/bin/hipcc $flags -gline-tables-only -S --cuda-device-only -o "$1" "$root/cudaraytracing_amd/csrc/crt_mega3.hip" 2> /dev/null || { echo "asm.sh: compile failed" >&2; exit 1; }

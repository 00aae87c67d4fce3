#!/bin/bash
# Builds the instrumented copy of the default render kernel (tools/bbprof/instrument.py) from the production sources with the
# production flags.  usage: tools/bbprof/build_co.sh [out dir]   ->  <out>/k_mega3_bb.co, <out>/k_mega3_bb.json
set -e -o pipefail
here=$(cd "$(dirname "$0")" && pwd); root=$(cd "$here/../.." && pwd)
out=${1:-$root/tools/bbprof/out}; mkdir -p "$out"
sym=${CRT_BBPROF_SYM:-_ZN4crtk7k_mega3ILi2ELb0ELb0ELb0ELb1ELb0ELb1ELb1EEEvNS_8MParams3E}  # the default render kernel (decoupled leaves); ...ELb1ELb0ELb0E... = the coupled form (run with CRT_DEC=0)
flags=$(python3 -c "import sys; sys.path.insert(0, '$root'); from cudaraytracing_amd import build as b; print(' '.join(b.COMMON + b.DEVICE))")
llvm=/opt/rocm/lib/llvm/bin
/opt/rocm/bin/hipcc $flags -gline-tables-only -S --cuda-device-only -o "$out/crt_mega3.s" "$root/cudaraytracing_amd/csrc/crt_mega3.hip" 2> /dev/null
python3 "$here/instrument.py" "$out/crt_mega3.s" "$sym" "$out/k_mega3_bb.s" "$out/k_mega3_bb.json"
$llvm/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c "$out/k_mega3_bb.s" -o "$out/k_mega3_bb.o"
$llvm/ld.lld -shared "$out/k_mega3_bb.o" -o "$out/k_mega3_bb.co"
rm -f "$out/crt_mega3.s" "$out/k_mega3_bb.o"
echo "$out/k_mega3_bb.co"

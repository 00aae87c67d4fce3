#!/bin/bash
# Assembly of the DEFAULT instantiation of k_mega3 alone (seconds instead of a minute), then its static census: tools/diet/asm_default.sh out.s [flags...]
root=$(cd "$(dirname "$0")/../.." && pwd); out=$1; shift
flags=$(python3 -c "import sys; sys.path.insert(0, '$root'); from cudaraytracing_amd import build as b; print(' '.join(b.COMMON + b.DEVICE))")
/opt/rocm/bin/hipcc $flags -DCRT_ASM_ONLY_DEFAULT "$@" -gline-tables-only -S --cuda-device-only -o "$out" "$root/cudaraytracing_amd/csrc/crt_mega3.hip" 2> "$out.err" || { echo "compile failed"; tail -20 "$out.err"; exit 1; }
awk '/\.size.*k_mega3ILi2ELb0ELb0ELb0ELb1ELb0ELb1ELb1E/{f=1} f&&/; (codeLenInByte|TotalNumSgprs|NumVgprs|ScratchSize|LDSByteSize|Occupancy)/{printf "%s ", $0} f&&/; Occupancy/{print ""; exit}' "$out"
python3 "$root/tools/diet/loopcount.py" "$out" | tail -1

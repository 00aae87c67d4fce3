#!/usr/bin/env python3
"""Static look at the compiler's code for one kernel (no GPU): per phase of k_mega3's main loop (attributed through the inlined-at
chains, as report.py does) the vector instructions by kind -- arithmetic against register traffic (v_mov / v_readlane /
v_writelane / v_readfirstlane: copies at joins, SGPR spills, uniform values parked in vector registers) -- and the register
numbers.  usage: static.py <crt_mega3.s> [kernel symbol]"""
import collections
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import report  # noqa: E402

SYM = "_ZN4crtk7k_mega3ILi2ELb0ELb0ELb0ELb1ELb0ELb1EEEvNS_8MParams3E"


def main():
    src = sys.argv[1]
    sym = sys.argv[2] if len(sys.argv) > 2 else SYM
    lines = open(src).read().split("\n")
    begin = next(i for i, l in enumerate(lines) if l.startswith(sym + ":"))
    kd = next(i for i in range(begin, len(lines)) if lines[i].strip().startswith(".amdhsa_kernel " + sym))
    rng = report.phase_ranges()

    def phase_of(line):
        for n, a, b in rng:
            if a <= line <= b:
                return n
        return "other"

    outer = 0
    depth = 0
    per = collections.defaultdict(collections.Counter)
    per_depth = collections.defaultdict(collections.Counter)
    for l in lines[begin + 1:kd]:
        m = re.search(r";\s+in Loop: Header=\S+ Depth=(\d+)", l) or re.search(r"Loop Header: Depth=(\d+)", l)
        if m and (l.startswith(".LBB") or l.startswith("; %bb.")):
            depth = int(m.group(1))
        elif l.startswith(".LBB") or l.startswith("; %bb."):
            depth = 0
        m = re.match(r"^\s*\.loc\s+(\d+)\s+(\d+)", l)
        if m:
            chain = re.findall(r"([\w.+-]+):(\d+):\d+", l.split(";", 1)[1]) if ";" in l else []
            outer = int(chain[-1][1]) if chain else int(m.group(2))
            continue
        s = l.strip()
        if not s or s.startswith((";", ".", "//")) or re.match(r"^[.\w$]+:", s):
            continue
        mn = s.split()[0]
        ph = phase_of(outer)
        if mn.startswith("v_"):
            kind = "copy" if mn.startswith(("v_mov_b32", "v_mov_b64", "v_accvgpr")) else "lane" if mn.startswith(("v_readlane", "v_writelane", "v_readfirstlane")) else "valu"
        elif mn.startswith("s_") and not mn.startswith(("s_waitcnt", "s_nop", "s_cbranch", "s_branch", "s_load", "s_endpgm")):
            kind = "salu"
        elif mn.startswith(("global_", "flat_", "buffer_", "scratch_")):
            kind = "flat" if mn.startswith("flat_") else "vmem"
        elif mn.startswith("ds_"):
            kind = "lds"
        else:
            kind = "ctl"
        per[ph][kind] += 1
        per_depth[depth][kind] += 1
    kinds = ["valu", "copy", "lane", "salu", "vmem", "flat", "lds", "ctl"]
    print("%-9s " % "phase" + " ".join("%6s" % k for k in kinds))
    for ph in ["sched", "inner", "leaf", "inner_ex", "leaf_ex", "LA", "LB", "LC", "other", "prologue", "epilogue"]:
        print("%-9s " % ph + " ".join("%6d" % per[ph][k] for k in kinds))
    print("by loop depth:")
    for d in sorted(per_depth):
        print("%-9s " % ("depth %d" % d) + " ".join("%6d" % per_depth[d][k] for k in kinds))
    for key in ("sgpr_spill_count", "vgpr_spill_count", "vgpr_count", "sgpr_count"):
        pass
    txt = "\n".join(lines[kd:kd + 60])
    for k in (".amdhsa_next_free_vgpr", ".amdhsa_next_free_sgpr", ".amdhsa_group_segment_fixed_size"):
        m = re.search(re.escape(k) + r"\s+(\d+)", txt)
        print(k, m.group(1) if m else "?")
    md = "\n".join(lines)
    m = re.search(r"\.name:\s+" + re.escape(sym) + r"\s*\n(?:.*\n){0,12}", md)
    i = md.find(".name:           " + sym)
    if i >= 0:
        seg = md[i:i + 600]
        for k in ("sgpr_spill_count", "vgpr_spill_count"):
            mm = re.search(k + r":\s+(\d+)", seg)
            print(k, mm.group(1) if mm else "?")


if __name__ == "__main__":
    main()

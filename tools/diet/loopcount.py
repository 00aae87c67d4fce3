#!/usr/bin/env python3
"""Static instruction census of the render kernel's traversal loop (the innermost loop of the default instantiation of k_mega3), from the
compiler's assembly: tools/bbprof/asm.sh out.s ; tools/diet/loopcount.py out.s [symbol].  Prints, per basic block of that loop, the
instructions by kind (vector / scalar / branch / wait / LDS / memory) and the source lines they come from, and the totals -- a quick
local reading of what an edit did to the steps before an A/B on the GPU (the dynamic truth is tools/bbprof)."""
import re, sys, collections
SYM = "_ZN4crtk7k_mega3ILi2ELb0ELb0ELb0ELb1ELb0ELb1ELb1EEEvNS_8MParams3E"

def kind(op):
    if op.startswith("v_"): return "valu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vmem"
    if op.startswith("s_load") or op.startswith("s_buffer_load"): return "smem"
    if op.startswith(("s_cbranch", "s_branch", "s_setpc", "s_swappc")): return "branch"
    if op.startswith(("s_waitcnt", "s_nop", "s_sleep", "s_barrier")): return "wait"
    if op.startswith("s_"): return "salu"
    return "other"

def main():
    args = [a for a in sys.argv[1:] if not a.startswith("-")]
    path = args[0]; sym = args[1] if len(args) > 1 else SYM
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith(sym + ":"))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    blocks = []  # (label, loopinfo, [(op, loc)])
    cur = ["entry", "", []]
    loc = ""
    for l in lines[start + 1:end]:
        m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", l)
        if m: loc = "%s:%s" % (m.group(1), m.group(2)); continue
        m = re.match(r"^(\.LBB\d+_\d+):\s*(;.*)?$", l)
        if m:
            blocks.append(cur); cur = [m.group(1), (m.group(2) or ""), []]; continue
        st = l.strip()
        if not st or st.startswith((";", ".")): continue
        cur[2].append((st.split()[0], loc))
    blocks.append(cur)
    # innermost loop: the deepest "Depth=" seen
    depth = lambda b: int(re.search(r"Depth=(\d+)", b[1]).group(1)) if "Depth=" in b[1] else 0
    dmin = 3  # scheduler loop 1 > traversal loop 2 > the alternating steps 3 (and the leaf records' loop 4)
    tot = collections.Counter()
    for b in blocks:
        if depth(b) < dmin: continue
        c = collections.Counter(kind(op) for op, _ in b[2])
        tot.update(c)
        srcs = collections.Counter(loc for _, loc in b[2])
        if "-v" in sys.argv:
            print("%-12s %s  %s" % (b[0], dict(c), " ".join("%s(%d)" % kv for kv in srcs.most_common(4))))
    print("total", dict(tot), "all", sum(tot.values()))

main()

#!/usr/bin/env python3
"""Basic-block profile of one gfx950 kernel, exact: every basic block of the compiler's own assembly gets a prologue that adds
(1 << 32 | popcount(exec)) to a 64-bit counter of its own, so a run yields, per block, how often it ran and with how many lanes.

  instrument.py <in.s> <kernel symbol> <out.s> <blocks.json> [--kernarg-off N]

<in.s> is `hipcc -S --cuda-device-only -gline-tables-only` of the production sources with the production flags: the code that is
profiled IS the production code, instruction for instruction; only the prologues are added.  They use registers the compiler left
alone -- s[100:101] (the kernel must not number SGPRs beyond s99) and six VGPRs above the kernel's own (which lowers the occupancy:
counts do not depend on it) -- and touch neither SCC nor VCC; a no-return atomic only makes the compiler's s_waitcnt counts more
conservative.  The counter buffer's address arrives in the two kernel-argument dwords at --kernarg-off (MParams3::dbg_loads /
dbg_valu, unused by the production build); counter i lives at byte i * 128 and the buffer must not cross a 4 GiB boundary
(the host hook sees to that).  blocks.json describes every block: label, opcode histogram, source lines.

A counted unit is a SEGMENT: a basic block cut again behind every instruction that rewrites exec (the `s_or_b64 exec, exec, sN`
that opens a join block, an `s_and_saveexec_b64` whose branch the compiler left out, `s_mov_b64 exec, ...`, `v_cmpx_*`), so that
every vector instruction is counted with the exec mask it runs under.  (Until round 5 the prologue stood before a block's first
instruction: a join block was sampled with the mask of the arm that fell into it -- 13 % of the dynamic vector instructions sat in
such blocks and 40 % of the census's lost lanes were theirs.)  A segment that would begin with a branch is not opened: the branch
stays with the segment before it (no vector instruction is affected).  Segments of one block are labelled <label>, <label>+1, ...

The other kernels of the translation unit are dropped from <out.s>: the code object holds the one instrumented kernel.
"""
import collections
import json
import re
import sys

STRIDE = 128


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    opts = dict(a[2:].split("=", 1) for a in sys.argv[1:] if a.startswith("--") and "=" in a)
    src, sym, dst, meta = args[:4]
    karg_off = int(opts.get("kernarg-off", "748"), 0)
    lines = open(src).read().split("\n")

    # ---- cut the translation unit down to: header directives, the kernel's section, its descriptor and metadata ----
    begin = next(i for i, l in enumerate(lines) if l.startswith(sym + ":"))
    # the section header precedes the label by a few lines
    sec = begin
    while not lines[sec].lstrip().startswith(".section"):
        sec -= 1
    # layout of a kernel: label, code, `.section .rodata`, descriptor (.amdhsa_kernel ... .end_amdhsa_kernel), back to the text
    # section, .Lfunc_endN, .size
    kd = next(i for i in range(begin, len(lines)) if lines[i].strip().startswith(".amdhsa_kernel " + sym))
    end = kd
    while not lines[end].lstrip().startswith(".section"):
        end -= 1
    tail = kd
    while ".end_amdhsa_kernel" not in lines[tail]:
        tail += 1
    fend = next(i for i in range(tail, len(lines)) if lines[i].startswith(".Lfunc_end"))
    body = lines[begin + 1:end]
    files = {}
    for l in lines:
        m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
        if m:
            files[int(m.group(1))] = m.group(3) or m.group(2)

    # ---- basic blocks ----
    blocks = []   # each: dict(label, first (index into body), instrs [(mnemonic, text)], locs Counter)
    cur = None
    cur_loc = None
    pending_new = True
    out = []

    def classify(mn):
        if mn.startswith("v_"):
            return "valu"
        if mn.startswith(("global_", "flat_", "buffer_", "scratch_")):
            return "vmem"
        if mn.startswith("ds_"):
            return "lds"
        if mn.startswith(("s_load", "s_store", "s_memtime", "s_memrealtime", "s_dcache", "s_buffer_load", "s_atc")):
            return "smem"
        if mn.startswith(("s_waitcnt", "s_nop", "s_sleep", "s_barrier", "s_endpgm", "s_sethalt", "s_setprio", "s_trap", "s_inst_prefetch", "s_code_end")):
            return "ctl"
        if mn.startswith(("s_cbranch", "s_branch", "s_setpc", "s_swappc", "s_call")):
            return "branch"
        if mn.startswith("s_"):
            return "salu"
        return "other"

    def writes_exec(s):
        """Does this instruction rewrite the exec mask?  (destination operand exec / exec_lo / exec_hi, the *_saveexec_* forms, v_cmpx_*)"""
        mn = s.split()[0]
        if "saveexec" in mn or mn.startswith("v_cmpx"):
            return True
        ops = s[len(mn):].split(";", 1)[0].split(",")
        return bool(ops) and re.match(r"^\s*exec(_lo|_hi)?\s*$", ops[0]) is not None and not mn.startswith(("s_cmp", "s_bitcmp", "v_cmp", "s_cbranch", "s_waitcnt"))

    label_of_next = None
    split_pending = False    # the instruction before this one rewrote exec: what follows is a segment of its own (unless it is a branch)
    seg_base, seg_k = None, 0
    for idx, l in enumerate(body):
        s = l.strip()
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            pending_new = True
            label_of_next = m.group(1)
            out.append(l)
            continue
        m = re.match(r"^\s*\.loc\s+(\d+)\s+(\d+)", l)
        if m:
            # innermost source line, and -- from the inlined-at chain the compiler prints as a comment -- the line of the kernel's own
            # body that this code was inlined into (the outermost frame): that one names the phase
            chain = re.findall(r"([\w.+-]+):(\d+):\d+", l.split(";", 1)[1]) if ";" in l else []
            inner = "%s:%d" % (files.get(int(m.group(1)), "?").split("/")[-1], int(m.group(2)))
            outer = int(chain[-1][1]) if chain else int(m.group(2))
            cur_loc = (inner, outer)
            out.append(l)
            continue
        if not s or s.startswith((";", ".", "//")) or re.match(r"^[.\w$]+:", s):
            out.append(l)
            continue
        mn = s.split()[0]
        if split_pending and not pending_new and classify(mn) != "branch" and mn != "s_endpgm":
            seg_k += 1
            cur = {"id": len(blocks), "label": "%s+%d" % (seg_base, seg_k), "instrs": [], "locs": collections.Counter(), "after_exec_write": True}
            blocks.append(cur)
            out.append("@@PROLOGUE %d@@" % cur["id"])
        split_pending = False
        if pending_new:
            cur = {"id": len(blocks), "label": label_of_next or ("fall%d" % len(blocks)), "instrs": [], "locs": collections.Counter()}
            blocks.append(cur)
            seg_base, seg_k = cur["label"], 0
            label_of_next = None
            pending_new = False
            out.append("@@PROLOGUE %d@@" % cur["id"])
        cur["instrs"].append(mn)
        cur.setdefault("seq", []).append([mn, cur_loc[0] if cur_loc else "?", cur_loc[1] if cur_loc else 0])
        if cur_loc and classify(mn) == "valu":
            cur["locs"][cur_loc[0]] += 1
        out.append(l)
        if classify(mn) == "branch" or mn == "s_endpgm":
            pending_new = True
        elif writes_exec(s):
            split_pending = True

    # ---- registers ----
    desc = lines[end:tail + 1]
    nv = next(int(re.search(r"(\d+)", l).group(1)) for l in desc if ".amdhsa_next_free_vgpr" in l)
    ns = next(int(re.search(r"(\d+)", l).group(1)) for l in desc if ".amdhsa_next_free_sgpr" in l)
    if ns > 100:
        sys.exit("kernel numbers SGPRs up to s%d: s[100:101] are not free" % (ns - 1))
    vb = (nv + 1) // 2 * 2  # (register tuples are 64-bit aligned) v[vb:vb+1] = (popcount, 1), v[vb+2:vb+3] = address (hi = base_hi), v[vb+4] = base_lo
    new_nv = (vb + 5 + 7) // 8 * 8

    def prologue(i):
        return "\n".join([
            "\ts_mov_b64 s[100:101], exec",
            "\ts_mov_b64 exec, 1",
            "\tv_bcnt_u32_b32 v%d, s100, 0" % vb,
            "\tv_bcnt_u32_b32 v%d, s101, v%d" % (vb, vb),
            "\tv_add_u32_e32 v%d, 0x%x, v%d" % (vb + 2, i * STRIDE, vb + 4),
            "\tglobal_atomic_add_x2 v[%d:%d], v[%d:%d], off" % (vb + 2, vb + 3, vb, vb + 1),
            "\ts_mov_b64 exec, s[100:101]",
            "\ts_nop 4",
        ])

    entry = "\n".join([
        "\ts_load_dword s100, s[0:1], 0x%x" % karg_off,
        "\ts_load_dword s101, s[0:1], 0x%x" % (karg_off + 4),
        "\ts_waitcnt lgkmcnt(0)",
        "\tv_mov_b32_e32 v%d, s100" % (vb + 4),
        "\tv_mov_b32_e32 v%d, s101" % (vb + 3),
        "\tv_mov_b32_e32 v%d, 1" % (vb + 1),
        "\ts_nop 4",
    ])
    text = []
    first = True
    for l in out:
        m = re.match(r"^@@PROLOGUE (\d+)@@$", l)
        if m:
            if first:
                text.append(entry)
                first = False
            text.append(prologue(int(m.group(1))))
        else:
            text.append(l)

    new_desc = []
    for l in desc:
        if ".amdhsa_next_free_vgpr" in l:
            l = re.sub(r"\d+", str(new_nv), l, count=1)
        elif ".amdhsa_accum_offset" in l:
            l = re.sub(r"\d+", str(new_nv), l, count=1)
        elif ".amdhsa_next_free_sgpr" in l:
            l = re.sub(r"\d+", "102", l, count=1)
        new_desc.append(l)

    # header: everything before the first .section .text / first function (target directives, .file table)
    head_end = next(i for i, l in enumerate(lines) if l.lstrip().startswith(".section\t.text") or l.lstrip().startswith(".protected") or l.lstrip().startswith(".globl"))
    head = [l for l in lines[:head_end] if not l.lstrip().startswith(".file") or True]
    # the .file table is spread through the unit: keep every .file line (needed by the .loc directives we keep)
    file_lines = [l for l in lines if re.match(r"^\s*\.file\s+\d+", l)]
    # amdgpu metadata: keep only this kernel's entry
    md_begin = next(i for i, l in enumerate(lines) if l.strip() == ".amdgpu_metadata")
    md_end = next(i for i, l in enumerate(lines) if l.strip() == ".end_amdgpu_metadata")
    md = lines[md_begin:md_end + 1]
    # split the kernels list into entries ("  - .agpr_count:" starts one)
    k0 = next(i for i, l in enumerate(md) if l.strip() == "amdhsa.kernels:")
    k1 = next(i for i in range(k0 + 1, len(md)) if re.match(r"^amdhsa\.", md[i]) or md[i].strip() == "...")
    entries, e = [], None
    for l in md[k0 + 1:k1]:
        if l.startswith("  - "):
            e = [l]
            entries.append(e)
        else:
            e.append(l)
    mine = [e for e in entries if any(re.search(r"\.name:\s+" + re.escape(sym) + r"\s*$", x) for x in e)]
    if len(mine) != 1:
        sys.exit("kernel metadata entry not found")
    ent = []
    for l in mine[0]:
        if re.match(r"^\s+\.vgpr_count:", l):
            l = re.sub(r"\d+", str(new_nv), l, count=1)
        elif re.match(r"^\s+\.sgpr_count:", l):
            l = re.sub(r"\d+", "108", l, count=1)
        ent.append(l)
    new_md = md[:k0 + 1] + ent + md[k1:]

    nodebug = lambda ls: [l for l in ls if not re.match(r"^\s*\.(cfi_|loc\s|file\s)", l)]  # (no debug sections in the output)
    with open(dst, "w") as f:
        f.write("\n".join(nodebug(head)) + "\n")
        f.write("\n".join(nodebug(lines[sec:begin + 1])) + "\n")
        f.write("\n".join(nodebug("\n".join(text).split("\n"))) + "\n")
        f.write("\n".join(new_desc) + "\n")
        f.write("\n".join(nodebug(lines[tail + 1:fend + 2])) + "\n")
        f.write("\n".join(new_md) + "\n")
    for b in blocks:
        h = collections.Counter(b["instrs"])
        b["hist"] = dict(h)
        b["n"] = {k: sum(v for mn, v in h.items() if classify(mn) == k) for k in ("valu", "salu", "vmem", "lds", "smem", "branch", "ctl")}
        b["locs"] = dict(b["locs"].most_common(12))
        del b["instrs"]
    json.dump({"kernel": sym, "stride": STRIDE, "n_blocks": len(blocks), "vgpr": [nv, new_nv], "blocks": blocks}, open(meta, "w"))
    print("blocks %d, VALU %d, vgprs %d -> %d" % (len(blocks), sum(b["n"]["valu"] for b in blocks), nv, new_nv))


if __name__ == "__main__":
    main()

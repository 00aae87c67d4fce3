#!/usr/bin/env python3
"""Reads the block description (instrument.py) and the counters of a run (CRT_BBPROF_OUT) and prints where the vector
instructions of the kernel go and where their lanes are masked off.

  report.py <k_mega3_bb.json> <counts.txt> [--json out.json] [--costs profiles/r02_valu_issue.json] [--top N]

Per basic block b: n_valu(b) static vector instructions, N(b) executions, A(b) active lanes summed over the executions.
  dynamic vector instructions      I = sum n_valu * N
  lane utilisation                 U = sum n_valu * A / (64 * I)          (what SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU) measures)
  lost lane-instructions of b      n_valu * (64 N - A)
Blocks are attributed to a phase by the line of the kernel body their code was inlined into (the outermost frame of the
compiler's inlined-at chain) and to a function by the innermost frame.
"""
import collections
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SRC = os.path.join(ROOT, "cudaraytracing_amd", "csrc")


def phase_ranges():
    """Line ranges of the phase arms of k_mega3's main loop, found by their markers in the source."""
    lines = open(os.path.join(SRC, "crt_mega3.hip")).read().split("\n")
    k0 = next(i for i, l in enumerate(lines) if "void k_mega3(const MParams3 M3)" in l) + 1
    marks = [("sched", r"^\s*for \(;;\) \{\s*$"), ("inner", r"auto inner_arm = "), ("leaf", r"auto leaf_arm = "), ("inner", r"auto inner_arm_dec = "), ("leaf", r"auto leaf_arm_dec = "),
             ("sched2", r"^\s*const bool plain = MODE == 1 \|\| n_exact == 0;"),
             ("LA", r"if \(act == PH3_LA\) \{"), ("LB", r"else if \(act == PH3_LB\) \{"), ("LC", r"if \(act != PH3_LA && act != PH3_LB\) \{"), ("end", r"^#undef PUSH3")]
    at, cur = [], k0
    for name, pat in marks:
        while not re.search(pat, lines[cur]):
            cur += 1
        at.append((name, cur + 1))
        cur += 1
    rng = [("prologue", k0, at[0][1] - 1)]
    for (n, a), (_, b) in zip(at[:-1], at[1:]):
        rng.append(("sched" if n == "sched2" else n, a, b - 1))
    rng.append(("epilogue", at[-1][1], at[-1][1] + 60))
    # the traversal steps are lambdas instantiated twice (with / without rays on the reference-arithmetic path): the outermost frame
    # of their code is the line of the call
    calls = []
    for i, l in enumerate(lines):
        for name, pat in (("inner", "inner_arm(std::false_type{})"), ("inner_ex", "inner_arm(std::true_type{})"), ("leaf", "leaf_arm(std::false_type{})"),
                          ("leaf_ex", "leaf_arm(std::true_type{})"), ("inner", "inner_arm_dec(std::false_type{})"), ("inner_ex", "inner_arm_dec(std::true_type{})"),
                          ("leaf", "leaf_arm_dec()")):
            if pat in l:
                calls.append((name, i + 1, i + 1))
    return calls + rng


def function_map():
    """file -> sorted list of (first line, name) of the function definitions in the kernel sources."""
    out = {}
    for fn in ("crt_mega3.hip", "crt_mega3.h", "crt_path.h", "crt_device.h", "crt_detmath.h", "crt_trace.h"):
        defs = []
        for i, l in enumerate(open(os.path.join(SRC, fn)).read().split("\n")):
            if not re.match(r"^(__device__|__global__|static|inline|__host__|CRT_HD)\b", l) or l.rstrip().endswith(";"):
                continue
            l2 = re.sub(r"__launch_bounds__\(\d+\)|__attribute__\(\(.*?\)\)\)|__attribute__\(\(\w+\)\)", "", l)
            m = re.search(r"\b([A-Za-z_]\w*)\s*\(", l2)
            if m:
                defs.append((i + 1, m.group(1)))
        out[fn] = defs
    return out


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    opts = {}
    it = iter(sys.argv[1:])
    for a in it:
        if a.startswith("--"):
            opts[a[2:]] = next(it, None)
    meta = json.load(open(args[0]))
    cnt = {}
    for l in open(args[1]):
        i, n, a = l.split()
        cnt[int(i)] = (int(n), int(a))
    top = int(opts.get("top", 25))
    rng = phase_ranges()
    fmap = function_map()

    def phase_of(line):
        for n, a, b in rng:
            if a <= line <= b:
                return n
        return "other"

    def func_of(loc):
        fn, ln = loc.rsplit(":", 1)
        ln = int(ln)
        best = "?"
        for a, name in fmap.get(fn, []):
            if a <= ln:
                best = name
            else:
                break
        return best if fn in fmap else fn

    costs = None
    if opts.get("costs"):
        costs = json.load(open(opts["costs"]))

    tot_i = tot_a = 0
    by_phase = collections.defaultdict(lambda: [0, 0, 0])       # dyn instr, lane sum, static
    by_pf = collections.defaultdict(lambda: [0, 0, 0])
    by_op = collections.Counter()
    by_op_phase = collections.defaultdict(collections.Counter)
    kinds = collections.Counter()
    blocks = []
    for b in meta["blocks"]:
        n, a = cnt.get(b["id"], (0, 0))
        lanes = a / n if n else 0.0
        pv = collections.Counter()
        for mn, inner, outer in b.get("seq", []):
            if mn.startswith("v_"):
                ph = phase_of(outer)
                pv[ph] += 1
                f = func_of(inner)
                by_phase[ph][0] += n; by_phase[ph][1] += a; by_phase[ph][2] += 1
                by_pf[(ph, f)][0] += n; by_pf[(ph, f)][1] += a; by_pf[(ph, f)][2] += 1
                by_op[mn] += n
                by_op_phase[ph][mn] += n
        for k, v in b["n"].items():
            kinds[k] += v * n
        nv = b["n"]["valu"]
        tot_i += nv * n; tot_a += nv * a
        blocks.append({"id": b["id"], "label": b["label"], "phase": pv.most_common(1)[0][0] if pv else "-", "n_valu": nv, "exec": n, "lanes": round(lanes, 2),
                       "dyn_valu": nv * n, "lost": nv * (64 * n - a), "where": list(b["locs"].items())[:4]})
    scale = float(opts.get("scale") or 1.0)
    print("dynamic VALU wave-instructions %.4g (x scale %.4g = %.4g), lane utilisation %.4f" % (tot_i, scale, tot_i * scale, tot_a / (64.0 * tot_i)))
    print("per kind (dynamic wave-instructions): " + ", ".join("%s %.4g" % (k, v) for k, v in kinds.most_common()))
    print()
    print("%-9s %12s %7s %7s %8s" % ("phase", "dyn VALU", "share", "lanes", "lost sh."))
    lost_tot = 64.0 * tot_i - tot_a
    for ph, (i, a, s) in sorted(by_phase.items(), key=lambda x: -x[1][0]):
        print("%-9s %12.4g %6.1f%% %7.2f %7.1f%%" % (ph, i, 100.0 * i / tot_i, a / i if i else 0, 100.0 * (64.0 * i - a) / lost_tot))
    print()
    print("%-9s %-22s %6s %12s %7s %7s %8s" % ("phase", "function", "static", "dyn VALU", "share", "lanes", "lost sh."))
    for (ph, f), (i, a, s) in sorted(by_pf.items(), key=lambda x: -x[1][0])[:top + 15]:
        print("%-9s %-22s %6d %12.4g %6.1f%% %7.2f %7.1f%%" % (ph, f, s, i, 100.0 * i / tot_i, a / i if i else 0, 100.0 * (64.0 * i - a) / lost_tot))
    print()
    print("blocks by lost lane-instructions:")
    print("%5s %-12s %-8s %6s %10s %6s %7s  %s" % ("id", "label", "phase", "n_valu", "exec", "lanes", "lost%", "where"))
    for b in sorted(blocks, key=lambda x: -x["lost"])[:top]:
        print("%5d %-12s %-8s %6d %10d %6.1f %6.1f%%  %s" % (b["id"], b["label"], b["phase"], b["n_valu"], b["exec"], b["lanes"], 100.0 * b["lost"] / lost_tot,
                                                          " ".join("%s(%d)" % (k, v) for k, v in b["where"])))
    print()
    print("opcodes (dynamic):")
    for mn, n in by_op.most_common(40):
        print("  %-28s %12.4g %5.1f%%" % (mn, n, 100.0 * n / tot_i))
    res = {"kernel": meta["kernel"], "dyn_valu": tot_i, "lane_utilisation": tot_a / (64.0 * tot_i), "kinds": dict(kinds),
           "by_phase": {ph: {"dyn_valu": i, "lanes": a / i if i else 0, "static": s} for ph, (i, a, s) in by_phase.items()},
           "by_phase_function": [{"phase": ph, "function": f, "static": s, "dyn_valu": i, "lanes": a / i if i else 0} for (ph, f), (i, a, s) in sorted(by_pf.items(), key=lambda x: -x[1][0])],
           "opcodes": dict(by_op), "opcodes_by_phase": {ph: dict(c) for ph, c in by_op_phase.items()},
           "blocks": sorted(blocks, key=lambda x: -x["lost"])}
    if costs:
        table = costs.get("cycles_per_instr") or costs

        def price(mn):
            """Issue cost of an opcode from the measured table (tools/valu_issue_gen.py): the opcode itself, else its family."""
            enc = "e64" if mn.endswith("_e64") else "e32"
            key = re.sub(r"_e32$|_e64$|_dpp$|_sdwa$", "", mn)
            if key == "v_cndmask_b32":   # (the e32 form reads vcc; the microbenchmark's number for it is an artefact of its constant vcc: the e64 number stands for both)
                return table.get("v_cndmask_b32_e64")
            if key in table:
                return table[key]
            fam = [(r"^v_cmp_class", "v_cmp_class_f32"), (r"^v_cmpx?_\w+_f32$", "v_cmp_f32_" + enc), (r"^v_cmpx?_\w+_[iu](32|16)$", "v_cmp_u32_" + enc),
                   (r"^v_(min|max)_[iu]32$", "v_min_u32"), (r"^v_(min|max)_f32$", "v_max_f32"), (r"^v_med3_", "v_max3_f32"), (r"^v_(min3|max3)_[iu]32$", "v_max3_f32"),
                   (r"^v_subrev_f32$", "v_sub_f32"), (r"^v_subrev_u32$", "v_sub_u32"), (r"^v_ashrrev_i32$", "v_lshrrev_b32"), (r"^v_mul_u32_u24$", "v_mul_i32_i24"),
                   (r"^v_cvt_", "v_cvt_f32_u32"), (r"^v_alignbit_b32$", "v_perm_b32"), (r"^v_add3_u32$", "v_and_or_b32"), (r"^v_or3_b32$", "v_and_or_b32"),
                   (r"^v_xad_u32$", "v_and_or_b32"), (r"^v_lshl_or_b32$", "v_and_or_b32"), (r"^v_add_lshl_u32$", "v_lshl_add_u32"), (r"^v_bfi_b32$", "v_and_or_b32"),
                   (r"^v_mad_u32_u16$|^v_mad_i32_i16$", "v_mad_u32_u24"), (r"^v_mul_hi_i32$", "v_mul_hi_u32"), (r"^v_rsq_f32$|^v_exp_f32$|^v_log_f32$", "v_rcp_f32"),
                   (r"^v_floor_f32$|^v_fract_f32$|^v_rndne_f32$|^v_trunc_f32$|^v_ldexp_f32$|^v_frexp", "v_cvt_f32_u32"), (r"^v_accvgpr", "v_mov_b32"),
                   (r"^v_pk_", "v_pk_mul_f32"), (r"^v_ashrrev_i64$|^v_lshrrev_b64$", "v_lshlrev_b64"), (r"^v_add_co_u32$|^v_addc_co_u32$|^v_sub_co_u32$|^v_subb_co_u32$", "v_mad_u64_u32"),
                   (r"^v_bcnt_u32_b32$|^v_ffbh|^v_ffbl|^v_bfrev", "v_bfe_u32"), (r"^v_swap_b32$", "v_mov_b64")]
            for pat, rep in fam:
                if re.match(pat, key):
                    return table.get(rep)
            return None
        c_lo = c_hi = c_mid = 0.0
        unknown = collections.Counter()
        by_tier = collections.Counter()
        for mn, n in by_op.items():
            v = price(mn)
            if isinstance(v, (int, float)):
                c_lo += n * v; c_hi += n * v; c_mid += n * v
                by_tier["fast (< 3.5 cycles)" if v < 3.5 else "slow (>= 3.5 cycles)" if v < 7 else "transcendental"] += n
            else:
                unknown[re.sub(r"_e32$|_e64$", "", mn)] += n
                c_lo += n * 2.5; c_hi += n * 4.9; c_mid += n * 3.7
        res["cycles_per_valu"] = {"mid": c_mid / tot_i, "lo": c_lo / tot_i, "hi": c_hi / tot_i, "unpriced_share": sum(unknown.values()) / tot_i,
                                  "unpriced": dict(unknown.most_common(30)), "tiers": {k: v / tot_i for k, v in by_tier.items()},
                                  "price_list": opts["costs"]}
        print()
        print("cycles per VALU instruction: %.3f (%.3f .. %.3f), %.2f %% of the instructions unpriced %s" %
              (c_mid / tot_i, c_lo / tot_i, c_hi / tot_i, 100.0 * sum(unknown.values()) / tot_i, dict(unknown.most_common(8))))
        print("tiers: " + ", ".join("%s %.1f %%" % (k, 100.0 * v / tot_i) for k, v in by_tier.most_common()))
    if opts.get("json"):
        json.dump(res, open(opts["json"], "w"), indent=1)


if __name__ == "__main__":
    main()

"""The basic-block profiler (tools/bbprof): the instrumented copy of the default render kernel is the production code plus counting
prologues, so (1) its frame must equal the oracle's bit for bit -- the same smoke render __graft_entry__.smoke() runs -- and (2) its
counters must add up: the blocks of the scheduler loop run once per iteration, no block reports more than 64 lanes per execution."""
import json
import os
import subprocess
import sys

import pytest

import util


@pytest.mark.gpu
def test_instrumented_kernel_renders_the_same_frame_and_its_counters_add_up(tmp_path):
    out = str(tmp_path)
    r = subprocess.run([os.path.join(util.ROOT, "tools", "bbprof", "build_co.sh"), out], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    co, meta = os.path.join(out, "k_mega3_bb.co"), os.path.join(out, "k_mega3_bb.json")
    assert os.path.exists(co) and os.path.exists(meta)
    counts = os.path.join(out, "counts.txt")
    env = dict(os.environ, CRT_BBPROF_CO=co, CRT_BBPROF_OUT=counts)
    env.pop("CRT_LIB_PATH", None)
    r = subprocess.run([sys.executable, os.path.join(util.ROOT, "__graft_entry__.py"), "--smoke"], capture_output=True, text=True, env=env, cwd=util.ROOT)
    assert r.returncode == 0 and "f32 bit-identical=True rgb8 identical=True" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    blocks = json.load(open(meta))["blocks"]
    rows = {}
    for line in open(counts):
        i, n, a = (int(v) for v in line.split())
        rows[i] = (n, a)
    assert len(rows) > 100                                     # the smoke frame reaches most of the kernel
    assert all(a <= 64 * n for n, a in rows.values())          # lanes per execution
    dyn = sum(b["n"]["valu"] * rows.get(b["id"], (0, 0))[0] for b in blocks)
    lanes = sum(b["n"]["valu"] * rows.get(b["id"], (0, 0))[1] for b in blocks)
    assert dyn > 10 ** 6 and 0.2 < lanes / (64.0 * dyn) < 1.0  # a plausible lane utilisation
    rep = subprocess.run([sys.executable, os.path.join(util.ROOT, "tools", "bbprof", "report.py"), meta, counts, "--costs",
                          os.path.join(util.ROOT, "profiles", "r03_valu_issue_ops.json")], capture_output=True, text=True)
    assert rep.returncode == 0 and "cycles per VALU instruction" in rep.stdout, rep.stderr[-2000:]


# ---- CPU: where the counting prologues go ----
_SYNTH = r"""
	.amdgcn_target "amdgcn-amd-amdhsa--gfx950"
	.file	1 "synth.hip"
	.section	.text.synth,"axG",@progbits,synth,comdat
	.protected	synth
	.globl	synth
	.p2align	8
	.type	synth,@function
synth:
	.loc	1 10 0
	v_mov_b32_e32 v1, 0
	v_cmp_gt_u32_e32 vcc, 32, v0
	s_and_saveexec_b64 s[4:5], vcc
	s_cbranch_execz .LBB0_2
	.loc	1 11 0
	v_add_u32_e32 v1, 1, v1
	v_add_u32_e32 v1, 1, v1
.LBB0_2:
	s_or_b64 exec, exec, s[4:5]
	.loc	1 12 0
	v_add_u32_e32 v1, 2, v1
	v_cmp_gt_u32_e32 vcc, 8, v0
	s_and_saveexec_b64 s[6:7], vcc
	v_add_u32_e32 v1, 3, v1
	s_mov_b64 exec, s[6:7]
	v_add_u32_e32 v1, 4, v1
	s_endpgm
	.section	.rodata,"a",@progbits
	.p2align	6, 0x0
	.amdhsa_kernel synth
		.amdhsa_next_free_vgpr 2
		.amdhsa_next_free_sgpr 8
		.amdhsa_accum_offset 4
	.end_amdhsa_kernel
	.section	.text.synth,"axG",@progbits,synth,comdat
.Lfunc_end0:
	.size	synth, .Lfunc_end0-synth
	.amdgpu_metadata
---
amdhsa.kernels:
  - .agpr_count:     0
    .args:
      - .address_space:  global
        .offset:         0
        .size:           8
        .value_kind:     global_buffer
    .group_segment_fixed_size: 0
    .kernarg_segment_align: 8
    .kernarg_segment_size: 8
    .max_flat_workgroup_size: 64
    .name:           synth
    .private_segment_fixed_size: 0
    .sgpr_count:     8
    .sgpr_spill_count: 0
    .symbol:         synth.kd
    .uniform_work_group_size: 1
    .uses_dynamic_stack: false
    .vgpr_count:     2
    .vgpr_spill_count: 0
    .wavefront_size: 64
amdhsa.target:   amdgcn-amd-amdhsa--gfx950
amdhsa.version:
  - 1
  - 2
...
	.end_amdgpu_metadata
"""


def test_a_block_that_rewrites_exec_is_counted_behind_that_instruction(tmp_path):
    """A join block opens with `s_or_b64 exec, exec, sN`; an `s_and_saveexec_b64` without a branch and an `s_mov_b64 exec, ...` sit in the
    middle of blocks.  Every vector instruction must be counted by a prologue that stands BEHIND the last exec write in front of it."""
    src, dst, meta = (str(tmp_path / n) for n in ("in.s", "out.s", "out.json"))
    open(src, "w").write(_SYNTH)
    r = subprocess.run([sys.executable, os.path.join(util.ROOT, "tools", "bbprof", "instrument.py"), src, "synth", dst, meta, "--kernarg-off=0"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    blocks = json.load(open(meta))["blocks"]
    labels = [b["label"] for b in blocks]
    # entry block, the masked arm, the join (its s_or alone), the join's body, behind the saveexec, behind the restore
    assert labels == ["fall0", "fall1", ".LBB0_2", ".LBB0_2+1", ".LBB0_2+2", ".LBB0_2+3"], labels
    assert [b["n"]["valu"] for b in blocks] == [2, 2, 0, 2, 1, 1]
    # (the saveexec + branch at the end of the entry block opens no segment: the branch stays with its block)
    out = [l.strip() for l in open(dst).read().split("\n")]
    is_prologue_head = lambda i: out[i] == "s_mov_b64 s[100:101], exec"
    for write in ("s_or_b64 exec, exec, s[4:5]", "s_and_saveexec_b64 s[6:7], vcc", "s_mov_b64 exec, s[6:7]"):
        i = out.index(write)
        assert is_prologue_head(i + 1), (write, out[i:i + 3])
    i = out.index("s_and_saveexec_b64 s[4:5], vcc")
    assert out[i + 1].startswith("s_cbranch_execz")
    # every vector instruction of the kernel is preceded -- with no exec write in between -- by a prologue's restore of exec
    last = None
    for l in out:
        if l == "s_mov_b64 exec, s[100:101]":
            last = "prologue"
        elif l.startswith(("s_or_b64 exec", "s_and_saveexec", "s_mov_b64 exec, s[6")):
            last = "write"
        elif l.startswith("v_add_u32_e32 v1,"):  # (the kernel's own: the prologues add into registers above it)
            assert last == "prologue", l
    # and the result still assembles
    clang = "/opt/rocm/lib/llvm/bin/clang"
    if os.path.exists(clang):
        a = subprocess.run([clang, "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", dst, "-o", str(tmp_path / "o.o")], capture_output=True, text=True)
        assert a.returncode == 0, a.stderr[-2000:]

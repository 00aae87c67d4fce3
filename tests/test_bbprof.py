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

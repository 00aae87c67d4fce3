"""The leaf queue of the decoupled-leaves kernel when it runs full: an inner step counts a visit's queue entries before it
writes any; if they do not fit, the lanes beyond a quarter of the free entries are taken out of a first visit again (queued as they came)
and a second visit is dropped whole (crt_mega3.hip: inner4_step_dec, CHECK 1 / 2).  With the production queue of 256 entries that is
rare; a library built with -DLEAFQ_CAP=128 (tools/ab_build.sh) makes both paths run in most batches.  The frames of both scenes, with and
without CRT_FLAG_FORCE_EXACT (every ray on the reference-arithmetic arm, whose appends lie outside the counted ones) and with every
sample traced, must stay the oracle's bit for bit, and the counting kernel must report that the paths ran."""
import json
import os
import shutil
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_small_leaf_queue_overflows_and_frames_stay_exact():
    lib = os.path.join(ROOT, "cudaraytracing_amd", "lib", "ab", "leafq128.so")
    # everything the variant is compiled from (a hand-kept list of five files missed crt_wavefront.hip and the headers behind
    # crt_internal.h, so an edit there left an old variant under test): the library's own dependency list
    sys.path.insert(0, ROOT)
    from cudaraytracing_amd import build as B
    srcs = [os.path.join(B.CSRC, d) for d in B.LIB_DEPS] + [os.path.join(ROOT, "tools", "ab_build.sh")]
    if not os.path.exists(lib) or any(os.path.getmtime(s) > os.path.getmtime(lib) for s in srcs):
        if not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
            pytest.skip("no hipcc to build the -DLEAFQ_CAP=128 variant")
        subprocess.run([os.path.join(ROOT, "tools", "ab_build.sh"), "leafq128", "-DLEAFQ_CAP=128"], check=True, cwd=ROOT, stdout=subprocess.DEVNULL, timeout=900)
    env = dict(os.environ, CRT_LIB_PATH=lib)
    for k in ("CRT_DEC", "CRT_REF16", "CRT_REF32"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "leafq_variant_driver.py")], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    res = json.loads(p.stdout.strip().splitlines()[-1])
    assert res["library"] == lib
    for c in res["cases"]:
        assert c["pixels_differ"] == 0 and c["rays_equal"], c
    counted = [c for c in res["cases"] if c["counting_kernel"] and c["flags"] != "force_exact"]
    assert counted and all(c["lanes_taken_back"] > 0 and c["second_visits_dropped_lanes"] > 0 for c in counted), counted

"""The insertion-based optimisation pass over the SAH tree (csrc/crt_accel.h: optimize_sah, round 5): its speculative multi-threaded form must
return the SAME tree whatever the number of threads, and -- on both shipped scenes -- the tree of the serial form of round 4, node for
node.  (Any tree over the reference's leaves renders the same frame, crt_accel.h; what this guards is that a scene's tree, and with it the
frame TIME, does not depend on the machine's core count or on thread timing.)  CPU only: the host layer of libcrt.so, no device."""
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("scene", ["cornell-box", "veach-mis"])
def test_speculative_pass_equals_the_serial_pass(tmp_path, scene):
    from cudaraytracing_amd import build as b
    b.build_lib()
    exe = str(tmp_path / "sah_opt_bench")
    subprocess.run(["g++", "-O2", "-std=c++17", "-pthread", "-I" + os.path.join(ROOT, "include"), "-I" + b.CSRC, os.path.join(ROOT, "tools", "sah_opt_bench.cpp"),
                    "-L" + b.LIBDIR, "-lcrt", "-Wl,-rpath," + b.LIBDIR, "-o", exe], check=True, cwd=ROOT, timeout=600)
    for threads in ("3", "8"):
        p = subprocess.run([exe, os.path.join("scenes", scene, "config.json"), "1"], cwd=ROOT, capture_output=True, text=True, timeout=300,
                           env=dict(os.environ, CRT_SAH_OPT_THREADS=threads))
        assert p.returncode == 0, p.stderr[-2000:]
        first = json.loads(p.stdout.splitlines()[0])
        assert first["one_thread_equals_many"] is True, first
        assert first["arrays_equal_serial"] is True and first["summed_area_serial"] == first["summed_area_speculative"], first

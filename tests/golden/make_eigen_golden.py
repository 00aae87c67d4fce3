#!/usr/bin/env python3
"""Regenerates tests/golden/eigen_ops.json from the reference's vendored Eigen 3.4.90 and
include/Camera.h (authoring container only: needs /root/reference).

    make -C oracle ref_probe && python tests/golden/make_eigen_golden.py

The probe (oracle/ref_probe/eigen_camera_probe.cpp) compiles those reference files where they
lie, with no stand-in headers, and prints input/output float bit patterns for every vector
operation and scalar chain the hot path uses.  Only the resulting data is committed.
"""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
probe = os.path.join(ROOT, "oracle", "_ref", "eigen_camera_probe")
if not os.path.exists(probe):
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref_probe"])
out = subprocess.check_output([probe])
with open(os.path.join(ROOT, "tests", "golden", "eigen_ops.json"), "wb") as f:
    f.write(out)
print("wrote eigen_ops.json (%d bytes)" % len(out))

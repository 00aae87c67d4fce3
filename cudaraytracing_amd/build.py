"""Builds the native parts of cudaraytracing_amd in-tree.

  lib/libcrt.so   HIP kernels (gfx950) + host layer + C ABI   (hipcc)
  lib/crt_cli     headless config.json -> PNG command line     (hipcc, links libcrt.so)

The flags matter for bit parity with the CPU oracle: no FMA contraction, no
fast-math, correctly rounded fp32 divide / sqrt on the device.
"""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libcrt.so")
CLI = os.path.join(LIBDIR, "crt_cli")

HIPCC = os.environ.get("HIPCC") or shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
COMMON = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math"] + os.environ.get("CRT_EXTRA_CXXFLAGS", "").split()
DEVICE = ["--offload-arch=gfx950", "-fhip-fp32-correctly-rounded-divide-sqrt"]

LIB_SOURCES = ["crt_mega3.hip", "crt_wavefront.hip", "crt_frame.hip", "crt_render.hip", "crt_multi.hip", "crt_bvh_build.hip", "crt_accel_build.hip", "crt_host.cpp"]
LIB_DEPS = LIB_SOURCES + ["crt_path.h", "crt_mega3.h", "crt_internal.h", "crt_device.h", "crt_trace.h", "crt_accel.h", "crt_detmath.h", "crt_host.hpp", "crt_png.h", "crt_jpeg.h", "crt_formats.h", "crt_image.h", "crt_bvh_build.h",
                          os.path.join("..", "..", "include", "crt.h")]
FLAGS_FILE = os.path.join(LIBDIR, "libcrt.flags")  # the flag string libcrt.so was built with (a variant build is stale for a default run)


def flags_string():
    return " ".join(COMMON + DEVICE)


def built_flags():
    """Flag string of the libcrt.so in the tree ("" if unknown)."""
    try:
        with open(FLAGS_FILE) as f:
            return f.read().strip()
    except OSError:
        return ""


KERNEL_DEPS = ["crt_mega3.hip", "crt_mega3.h", "crt_path.h", "crt_render.hip", "crt_device.h", "crt_trace.h", "crt_accel.h", "crt_accel_build.hip", "crt_detmath.h"]  # what the render kernel is made of, and the host code that lays out what it walks


def _code_only(text):
    """C / C++ source without comments and with runs of white space collapsed: what the compiler sees, more or less -- so that
    editing a comment does not invalidate the profiles stamped with source_hash()."""
    out, i, n = [], 0, len(text)
    while i < n:
        c = text[i]
        if c == '"' or c == "'":                       # string / character literal: copied as it is
            j = i + 1
            while j < n and text[j] != c:
                j += 2 if text[j] == "\\" else 1
            out.append(text[i:j + 1]); i = j + 1
        elif text.startswith("//", i):
            j = text.find("\n", i)
            i = n if j < 0 else j
        elif text.startswith("/*", i):
            j = text.find("*/", i + 2)
            i = n if j < 0 else j + 2
            out.append(" ")
        else:
            out.append(c); i += 1
    return " ".join("".join(out).split())


def source_hash():
    """SHA-256 over the code (comments and white space aside) of the render kernels and of the host code that lays out what they
    walk, and the flag string: stamps profiles (bench.py drops PMC numbers collected on other code)."""
    import hashlib
    h = hashlib.sha256()
    for d in sorted(KERNEL_DEPS):
        with open(os.path.join(CSRC, d), "r", encoding="utf-8", errors="replace") as f:
            h.update(d.encode() + b"\0" + _code_only(f.read()).encode())
    h.update(flags_string().encode())
    return h.hexdigest()[:16]


def _stale(target, deps, check_flags=False):
    if not os.path.exists(target):
        return True
    if check_flags and built_flags() != flags_string():
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(os.path.join(CSRC, d)) > t for d in deps)


class _BuildLock:
    """The N ranks of a multi-GPU job load the library at the same moment: the staleness check and a rebuild are serialised
    across processes (first one builds into a temporary file and renames it, the others then find a fresh library)."""

    def __enter__(self):
        import fcntl
        os.makedirs(LIBDIR, exist_ok=True)
        self.f = open(os.path.join(LIBDIR, ".build.lock"), "w")
        fcntl.flock(self.f, fcntl.LOCK_EX)
        return self

    def __exit__(self, *a):
        import fcntl
        fcntl.flock(self.f, fcntl.LOCK_UN)
        self.f.close()


def _obj_flags(obj):
    try:
        with open(obj + ".flags") as f:
            return f.read().strip()
    except OSError:
        return ""


def _compile_one(src, obj, verbose):
    cmd = [HIPCC] + COMMON + DEVICE + ["-c", os.path.join(CSRC, src), "-o", obj]
    if verbose:
        print(" ".join(cmd))
    if os.path.exists(obj + ".flags"):
        os.remove(obj + ".flags")  # (an interrupted compile leaves no object that claims these flags)
    subprocess.check_call(cmd)
    with open(obj + ".flags", "w") as f:  # the flags THIS object was compiled with: an interrupted flag-change build cannot link mixed objects
        f.write(flags_string() + "\n")


def build_lib(force=False, verbose=False):
    """One object per translation unit under lib/obj/ (compiled in parallel, only the stale ones), then the link."""
    os.makedirs(LIBDIR, exist_ok=True)
    if not force and not _stale(LIB, LIB_DEPS, check_flags=True):
        return LIB
    with _BuildLock():
        if not force and not _stale(LIB, LIB_DEPS, check_flags=True):  # another process has built it meanwhile
            return LIB
        from concurrent.futures import ThreadPoolExecutor
        objdir = os.path.join(LIBDIR, "obj")
        os.makedirs(objdir, exist_ok=True)
        flags_changed = force or built_flags() != flags_string()
        headers = [d for d in LIB_DEPS if d not in LIB_SOURCES]
        jobs, objs = [], []
        for s in LIB_SOURCES:
            obj = os.path.join(objdir, os.path.splitext(s)[0] + ".o")
            objs.append(obj)
            if flags_changed or _obj_flags(obj) != flags_string() or _stale(obj, [s] + headers):
                jobs.append((s, obj))
        with ThreadPoolExecutor(max_workers=min(6, max(1, len(jobs)))) as ex:
            for f in [ex.submit(_compile_one, s, o, verbose) for s, o in jobs]:
                f.result()
        tmp = LIB + ".tmp.%d" % os.getpid()
        cmd = [HIPCC] + DEVICE + ["-shared", "-fPIC"] + objs + ["-ldl", "-lpthread", "-o", tmp]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        os.replace(tmp, LIB)
        with open(FLAGS_FILE, "w") as f:
            f.write(flags_string() + "\n")
    return LIB


def build_cli(force=False, verbose=False):
    src = os.path.join(CSRC, "crt_cli.cpp")
    if not os.path.exists(src):
        return None
    if not force and not _stale(CLI, ["crt_cli.cpp", "crt_host.hpp"]) and os.path.getmtime(CLI) >= os.path.getmtime(LIB):
        return CLI
    cmd = [HIPCC] + COMMON + [src, "-L" + LIBDIR, "-lcrt", "-Wl,-rpath,$ORIGIN", "-o", CLI]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return CLI


def build_all(force=False, verbose=False):
    build_lib(force, verbose)
    build_cli(force, verbose)


if __name__ == "__main__":
    import sys
    build_all(force="--force" in sys.argv, verbose=True)

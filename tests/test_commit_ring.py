"""GPU tests of the commit ring (CRT_FLAG_BOUNDED_RADIANCE, csrc/crt_path.h: ring_publish / ring_commit): the frame's sum
c += L_k / spp (reference: Render.cuh:348) made in sample order INSIDE the launch with radiance storage for a window of samples.
The ring changes where and when the additions happen, never their order: every frame here must be, bit for bit (f32 sums and RGB8),
the frame of the default path, which keeps one radiance per path and sums after the launch."""
import os

import numpy as np
import pytest

import cudaraytracing_amd as crt
import util

pytestmark = pytest.mark.gpu


def _render(name, w, h, spp, ring_log2=None, flags=0, traversal=None, ranges=None):
    """(rgb, f32 sums, launches, (storage bytes, ring samples))"""
    old = os.environ.pop("CRT_COMMIT_RING_LOG2", None)
    if ring_log2 is not None:
        os.environ["CRT_COMMIT_RING_LOG2"] = str(ring_log2)   # (test hook: a ring of 2^n samples, far smaller than a render would choose)
    try:
        t = util.task(name)
        eye, iv, fov = util.camera(name)
        r = crt.Render(util.host_scene(name), spp, t.P_RR, t.light_sample_n, device=0)
        try:
            r.extra_flags = flags
            if traversal is not None:
                r.traversal = traversal
            launches = 0
            if ranges is None:
                rgb = r.run_view(eye, iv, fov, width=w, height=h).copy()
                launches = r.stats["kernel_launches"]
            else:
                rgb = None
                for (b, n) in ranges:
                    rgb = r.run_view_range(eye, iv, fov, b, n, width=w, height=h)
                rgb = rgb.copy()
            return rgb, r.mean_buffer.copy(), launches, r.radiance_storage()
        finally:
            r.free()
    finally:
        os.environ.pop("CRT_COMMIT_RING_LOG2", None)
        if old is not None:
            os.environ["CRT_COMMIT_RING_LOG2"] = old


def _same(a, b):
    return np.array_equal(a[0], b[0]) and np.array_equal(util.bits(a[1]), util.bits(b[1]))


@pytest.mark.parametrize("name,w,h,spp,ring_log2", [("cornell-box", 128, 96, 64, 2), ("cornell-box", 128, 96, 64, 3), ("cornell-box", 203, 149, 48, 2),
                                                    ("veach-mis", 160, 120, 40, 2), ("cornell-box", 64, 64, 200, 4), ("cornell-box", 37, 21, 70, 1)])
def test_tiny_rings_reproduce_the_frame(name, w, h, spp, ring_log2):
    """Rings of 2 - 16 samples under a pool that could hold every path of the frame at once: nearly every ray slot is handed a work
    item it may not start yet (ST_WAIT), every sample is committed by whichever wave completes it, ragged tiles (203 x 149, 37 x 21)
    leave cursor shards with fewer pixels than slots."""
    ref = _render(name, w, h, spp)
    got = _render(name, w, h, spp, ring_log2=ring_log2)
    assert ref[3][1] == 0 and got[3][1] == 1 << ring_log2 and got[2] == 1
    assert got[3][0] < ref[3][0]
    assert _same(ref, got)


def test_the_flag_picks_a_ring_and_one_launch():
    """CRT_FLAG_BOUNDED_RADIANCE at 256 x 192 spp 160: the ring the render chooses for itself (64 samples here), less storage, same bits;
    with every sample traced (CRT_FLAG_TRACE_ALL) and in the two other traversal modes too."""
    name, w, h, spp = "cornell-box", 256, 192, 160
    ref = _render(name, w, h, spp)
    got = _render(name, w, h, spp, flags=crt.FLAG_BOUNDED_RADIANCE)
    assert got[3][1] in (32, 64, 128) and got[3][0] * 2 <= ref[3][0] and got[2] == 1
    assert _same(ref, got)
    assert _same(ref, _render(name, w, h, spp, flags=crt.FLAG_BOUNDED_RADIANCE | crt.FLAG_TRACE_ALL))
    for mode in (crt.TRAVERSAL_FAST, crt.TRAVERSAL_REFERENCE):
        assert _same(_render(name, w, h, 48, traversal=mode), _render(name, w, h, 48, ring_log2=3, traversal=mode)), mode


def test_progressive_ranges_with_and_without_the_ring():
    """Sample ranges submitted one after the other (crt_render_range): a ring launch continues the accumulator a launch without
    the ring left, and the other way round."""
    name, w, h, spp = "cornell-box", 96, 64, 96
    ref = _render(name, w, h, spp)
    old = os.environ.pop("CRT_COMMIT_RING_LOG2", None)
    try:
        t = util.task(name)
        eye, iv, fov = util.camera(name)
        r = crt.Render(util.host_scene(name), spp, t.P_RR, t.light_sample_n, device=0)
        try:
            os.environ["CRT_COMMIT_RING_LOG2"] = "2"
            r.run_view_range(eye, iv, fov, 0, 40, width=w, height=h)       # ring (4 samples)
            os.environ.pop("CRT_COMMIT_RING_LOG2")
            r.run_view_range(eye, iv, fov, 40, 16, width=w, height=h)      # one radiance per path
            os.environ["CRT_COMMIT_RING_LOG2"] = "3"
            rgb = r.run_view_range(eye, iv, fov, 56, 40, width=w, height=h).copy()   # ring (8 samples), ends the frame
            bad = util.bits(r.mean_buffer) != util.bits(ref[1])
            assert not bad.any(), "%d of %d floats differ, first at %s: %r vs %r" % (
                int(bad.sum()), bad.size, tuple(np.argwhere(bad)[0]), r.mean_buffer[bad][:4].tolist(), ref[1][bad][:4].tolist())
            assert np.array_equal(rgb, ref[0])
        finally:
            r.free()
    finally:
        os.environ.pop("CRT_COMMIT_RING_LOG2", None)
        if old is not None:
            os.environ["CRT_COMMIT_RING_LOG2"] = old


def test_a_shard_of_the_frame_through_the_ring():
    """Rank 1 of 3 (interleaved tiles, compact output): the ring's cursor shards are ranges of THIS rank's pixel slots."""
    import ctypes as C
    from cudaraytracing_amd import _capi as capi
    name, w, h, spp = "cornell-box", 200, 120, 64
    t = util.task(name)
    eye, iv, fov = util.camera(name)
    r = crt.Render(util.host_scene(name), spp, t.P_RR, t.light_sample_n, device=0)
    old = os.environ.pop("CRT_COMMIT_RING_LOG2", None)
    try:
        out = []
        for ring in (None, "2"):
            if ring:
                os.environ["CRT_COMMIT_RING_LOG2"] = ring
            slots = crt.shard_slots(w, h, 1, 3)
            buf = np.zeros((slots, 3), dtype=np.uint8)
            mean = np.zeros((slots, 3), dtype=np.float32)
            prm = r._params(rank=1, world=3, flags=capi.FLAG_TILED_OUTPUT, width=w, height=h)
            cam = r._cam(eye, iv, fov)
            st = capi.Stats()
            capi.check(capi.lib().crt_render(r._h, C.byref(cam), C.byref(prm), capi.ptr(buf), capi.ptr(mean), C.byref(st)), "crt_render")
            out.append((buf.copy(), mean.copy(), r.radiance_storage()))
        assert out[0][2][1] == 0 and out[1][2][1] == 4
        assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(util.bits(out[0][1]), util.bits(out[1][1]))
    finally:
        os.environ.pop("CRT_COMMIT_RING_LOG2", None)
        if old is not None:
            os.environ["CRT_COMMIT_RING_LOG2"] = old
        r.free()


def test_cli_flag_for_the_ring(tmp_path):
    """crt_cli --bounded-radiance (crt::Render::set_flags): the same PNG as without it (320 samples: more than the 256-sample ring a
    160 x 120 frame gets, so the ring is really used)."""
    import subprocess
    from PIL import Image
    from cudaraytracing_amd import build as b
    cli = b.build_cli()
    cfg = util.SCENES["cornell-box"]
    outs = []
    for extra in ([], ["--bounded-radiance"]):
        out = str(tmp_path / ("c%d.png" % len(extra)))
        r = subprocess.run([cli, cfg, "-o", out, "--spp", "320", "--width", "160", "--height", "120", "--seed", "7", "--base-dir", util.ROOT] + extra,
                           capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr
        outs.append(np.asarray(Image.open(out)).copy())
    assert np.array_equal(outs[0], outs[1]) and outs[0].std() > 10


def test_ring_under_the_multi_device_entry():
    """crt_multi_render with three ranks on the one device and CRT_FLAG_BOUNDED_RADIANCE passed through: every rank's scene handle
    renders its shard with its own ring (ranks sharing a device never wait for one another: a commit only needs work that resident
    waves of the same launch already hold); the gathered frame is the single-device frame."""
    name, w, h, spp = "cornell-box", 120, 88, 300
    t = util.task(name)
    eye, iv, fov = util.camera(name)
    ref = _render(name, w, h, spp)
    m = crt.MultiRender(util.host_scene(name), spp, t.P_RR, t.light_sample_n, devices=[0, 0, 0], gather=crt.GATHER_COPY)
    old = os.environ.pop("CRT_COMMIT_RING_LOG2", None)
    os.environ["CRT_COMMIT_RING_LOG2"] = "3"   # (a share of 3 520 pixel slots would get a ring longer than the 300 samples: 8 samples forced)
    try:
        m.extra_flags = crt.FLAG_BOUNDED_RADIANCE
        got = m.run_view(eye, iv, fov, width=w, height=h)
        assert np.array_equal(got, ref[0]) and np.array_equal(util.bits(m.mean_buffer), util.bits(ref[1]))
        assert m.info["paths"] == w * h * spp
    finally:
        os.environ.pop("CRT_COMMIT_RING_LOG2", None)
        if old is not None:
            os.environ["CRT_COMMIT_RING_LOG2"] = old
        m.free()

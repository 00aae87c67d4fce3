"""GPU tests on the two 8-GPU workloads of BASELINE.json at their own resolutions:

  C4  cornell-box 3840x2160 spp=256, pixel tiles across 8 ranks
  C5  veach-mis  1920x1080 spp=4096, 8 ranks

Pixel indices beyond 2^22, more than 2^30 work items per frame (several launches per frame: render_impl's sample chunks),
the 8-way tile shards of those frames and their ragged bottom-right tiles (2160 = 270 x 8, 1080 = 135 x 8, 3840 / 1920
columns) run at production size on the one GPU of the box; the oracle checks crops it can render in seconds, and
size-independent properties (FAST == TRACE_ALL, shard ray counts add up, shards == one device) cover the rest."""
import ctypes as C

import numpy as np
import pytest

import cudaraytracing_amd as crt
from cudaraytracing_amd import _capi as capi
import util

pytestmark = pytest.mark.gpu

C4 = ("cornell-box", 3840, 2160, 256)
C5 = ("veach-mis", 1920, 1080, 4096)


def _crops(w, h, cw, ch):
    """top-left, centre and the bottom-right corner (the last tile of the frame)"""
    return [(0, 0, cw, ch), ((w - cw) // 2 // 8 * 8, (h - ch) // 2 // 8 * 8, cw, ch), (w - cw, h - ch, cw, ch)]


def _oracle_crop(name, w, h, spp, crop, seed=0):
    t = util.task(name)
    eye, iv, fov = util.camera(name)
    return util.oracle_scene(name).render(eye, iv, fov, w, h, spp, t.P_RR, t.light_sample_n, seed=seed, crop=crop)


@pytest.mark.parametrize("name,w,h", [C4[:3], C5[:3]])
def test_eight_shards_of_the_full_resolution_frame(name, w, h):
    """spp 2 at the full resolution: 8 ranks on the one device reassemble to the one-device frame; crops match the oracle."""
    t = util.task(name)
    eye, iv, fov = util.camera(name)
    one = crt.Render(util.host_scene(name), 2, t.P_RR, t.light_sample_n, device=0)
    m = crt.MultiRender(util.host_scene(name), 2, t.P_RR, t.light_sample_n, devices=[0] * 8, gather=crt.GATHER_COPY)
    try:
        rgb = one.run_view(eye, iv, fov, width=w, height=h).copy()
        mean = one.mean_buffer.copy()
        got = m.run_view(eye, iv, fov, width=w, height=h)
        assert np.array_equal(got, rgb)
        assert np.array_equal(util.bits(m.mean_buffer), util.bits(mean))
        assert m.info["rays"] == one.stats["rays"] and m.info["paths"] == w * h * 2
        per_rank = [s["paths"] for s in m.rank_stats]
        assert sum(per_rank) == w * h * 2 and max(per_rank) - min(per_rank) <= 2 * 64  # interleaved tiles: equal shares
        for crop in _crops(w, h, 64, 48):
            x0, y0, cw, ch = crop
            orgb, omean, _, _ = _oracle_crop(name, w, h, 2, crop)
            assert np.array_equal(util.bits(mean[y0:y0 + ch, x0:x0 + cw]), util.bits(omean)), crop
            assert np.array_equal(rgb[y0:y0 + ch, x0:x0 + cw], orgb), crop
    finally:
        one.free()
        m.free()


def test_c4_full_size_on_one_device_in_two_launches():
    """cornell-box 3840x2160 spp=256 = 2.12 G work items: two launches of the megakernel (2^30 items per chunk); the frame
    with every sample traced must be the same frame; ray and path counts of the 8 shards add up to the frame's; crops of the
    frame equal the oracle's (32x24 pixels at spp 256)."""
    name, w, h, spp = C4
    t = util.task(name)
    eye, iv, fov = util.camera(name)
    one = crt.Render(util.host_scene(name), spp, t.P_RR, t.light_sample_n, device=0)
    try:
        rgb = one.run_view(eye, iv, fov, width=w, height=h).copy()
        mean = one.mean_buffer.copy()
        st = dict(one.stats)
        assert st["kernel_launches"] == 2 and st["paths"] == w * h * spp
        one.extra_flags = crt.FLAG_TRACE_ALL
        rgb_all = one.run_view(eye, iv, fov, width=w, height=h)
        assert one.stats["rays"] == st["rays"] and one.stats["rays_untraced"] == 0 and st["rays_untraced"] > 0
        assert np.array_equal(rgb_all, rgb)
        assert np.array_equal(util.bits(one.mean_buffer), util.bits(mean))
        one.extra_flags = 0
        # ... in ONE launch with the commit ring (CRT_FLAG_BOUNDED_RADIANCE): 32 samples of radiance instead of 2^30 paths'
        full_bytes = one.radiance_storage()[0]
        one.extra_flags = crt.FLAG_BOUNDED_RADIANCE
        rgb_ring = one.run_view(eye, iv, fov, width=w, height=h)
        assert one.stats["kernel_launches"] == 1 and one.stats["rays"] == st["rays"]
        ring_bytes, ring_samples = one.radiance_storage()
        assert ring_samples == 32 and ring_bytes * 4 <= full_bytes   # (32 samples x 8.3 M pixels against 2^30 paths)
        assert np.array_equal(rgb_ring, rgb) and np.array_equal(util.bits(one.mean_buffer), util.bits(mean))
        one.extra_flags = 0
        # ... and the exhaustive REFERENCE traversal of the same 7.9 G rays, and CRT_TRAVERSAL_FAST (the default mode plus distance pruning,
        # whose rule is not a theorem: the full-size C3 frame once found a ray it lost, csrc/crt_trace.h)
        for mode in (crt.TRAVERSAL_REFERENCE, crt.TRAVERSAL_FAST):
            one.traversal = mode
            rgb_ref = one.run_view(eye, iv, fov, width=w, height=h)
            one.traversal = crt.TRAVERSAL_EXACT
            assert one.stats["rays"] == st["rays"]
            diff = np.argwhere(np.any(util.bits(one.mean_buffer) != util.bits(mean), axis=2))
            assert diff.size == 0 and np.array_equal(rgb_ref, rgb), (mode, diff[:8])
        for crop in _crops(w, h, 32, 24)[1:]:
            x0, y0, cw, ch = crop
            orgb, omean, _, _ = _oracle_crop(name, w, h, spp, crop)
            assert np.array_equal(util.bits(mean[y0:y0 + ch, x0:x0 + cw]), util.bits(omean)), crop
            assert np.array_equal(rgb[y0:y0 + ch, x0:x0 + cw], orgb), crop
        # the 8-way shards at full size: rank 0 and rank 7 (which owns the last tile), compact tile order
        rays = 0
        for rank in range(8):
            slots = crt.shard_slots(w, h, rank, 8)
            buf = np.zeros((slots, 3), dtype=np.uint8)
            prm = one._params(rank=rank, world=8, flags=capi.FLAG_TILED_OUTPUT, width=w, height=h)
            cam = one._cam(eye, iv, fov)
            stt = capi.Stats()
            capi.check(capi.lib().crt_render(one._h, C.byref(cam), C.byref(prm), capi.ptr(buf), None, C.byref(stt)), "crt_render")
            rays += stt.rays
            assert stt.kernel_launches == 1
            tx = (w + 7) // 8
            for k in (0, slots // 64 // 2, slots // 64 - 1):  # first, middle and last tile of the rank
                tile = k * 8 + rank
                ty_, tx_ = divmod(tile, tx)
                assert np.array_equal(buf[k * 64:(k + 1) * 64].reshape(8, 8, 3), rgb[ty_ * 8:ty_ * 8 + 8, tx_ * 8:tx_ * 8 + 8]), (rank, k)
        assert rays == st["rays"]
    finally:
        one.free()


def test_c5_shares_of_rank_0_and_rank_7_at_full_size():
    """veach-mis 1920x1080 spp=4096: one rank's share is 1.06 G work items (one launch, just under the 2^30 of a chunk; 17 GB of
    per-path radiance).  Rank 0's share with every sample traced is the same share; a tile of rank 0 and the frame's last tile
    (rank 7) equal the oracle at spp 4096."""
    name, w, h, spp = C5
    t = util.task(name)
    eye, iv, fov = util.camera(name)
    one = crt.Render(util.host_scene(name), spp, t.P_RR, t.light_sample_n, device=0)
    tx, ty = (w + 7) // 8, (h + 7) // 8
    try:
        def share(rank, flags=0):
            slots = crt.shard_slots(w, h, rank, 8)
            buf = np.zeros((slots, 3), dtype=np.uint8)
            mean = np.zeros((slots, 3), dtype=np.float32)
            prm = one._params(rank=rank, world=8, flags=capi.FLAG_TILED_OUTPUT | flags, width=w, height=h)
            cam = one._cam(eye, iv, fov)
            stt = capi.Stats()
            capi.check(capi.lib().crt_render(one._h, C.byref(cam), C.byref(prm), capi.ptr(buf), capi.ptr(mean), C.byref(stt)), "crt_render")
            return buf, mean, stt.as_dict()

        b0, m0, s0 = share(0)
        assert s0["kernel_launches"] == 1 and s0["paths"] == (tx * ty // 8) * 64 * spp
        b0a, m0a, s0a = share(0, capi.FLAG_TRACE_ALL)
        assert s0a["rays"] == s0["rays"] and s0a["rays_untraced"] == 0 and s0["rays_untraced"] > 0
        assert np.array_equal(b0a, b0) and np.array_equal(util.bits(m0a), util.bits(m0))
        for mode in (crt.TRAVERSAL_REFERENCE, crt.TRAVERSAL_FAST):   # the exhaustive / the pruned traversal of the same 7.6 G rays
            one.traversal = mode
            b0r, m0r, s0r = share(0)
            one.traversal = crt.TRAVERSAL_EXACT
            diff = np.argwhere(np.any(util.bits(m0r) != util.bits(m0), axis=1))
            assert s0r["rays"] == s0["rays"] and diff.size == 0 and np.array_equal(b0r, b0), (mode, diff[:8])
        b7, m7, s7 = share(7)
        # rank 0: a tile in the middle of the plates; rank 7: the last tile of the frame
        for rank, buf, mean, tile in ((0, b0, m0, (ty // 2) * tx + tx // 2 - ((ty // 2) * tx + tx // 2) % 8), (7, b7, m7, tx * ty - 1)):
            assert tile % 8 == rank
            k = tile // 8
            ty_, tx_ = divmod(tile, tx)
            orgb, omean, _, _ = _oracle_crop(name, w, h, spp, (tx_ * 8, ty_ * 8, 8, 8))
            assert np.array_equal(util.bits(mean[k * 64:(k + 1) * 64].reshape(8, 8, 3)), util.bits(omean)), (rank, tile)
            assert np.array_equal(buf[k * 64:(k + 1) * 64].reshape(8, 8, 3), orgb), (rank, tile)
    finally:
        one.free()


def test_c3_full_size():
    """BASELINE configuration C3 at full size (veach-mis 800x600 spp 1024, 3.5 G rays, 42 % of them answered without traversal):
    the default mode (CRT_TRAVERSAL_EXACT), the same with every sample traced, FAST and the exhaustive REFERENCE traversal give the same frame bit for bit
    and the same ray counts; three 16x12 crops (a plate, a light, the last tile rows) equal the oracle at spp 1024."""
    name, w, h, spp = "veach-mis", 800, 600, 1024
    t = util.task(name)
    eye, iv, fov = util.camera(name)
    r = crt.Render(util.host_scene(name), spp, t.P_RR, t.light_sample_n, device=0)
    try:
        rgb = r.run_view(eye, iv, fov, width=w, height=h).copy()
        mean, st = r.mean_buffer.copy(), dict(r.stats)
        assert st["paths"] == w * h * spp and st["rays"] > 3 * 10 ** 9 and st["rays_untraced"] > 0.3 * st["rays"]
        for trav, flags in ((crt.TRAVERSAL_EXACT, crt.FLAG_TRACE_ALL), (crt.TRAVERSAL_FAST, 0), (crt.TRAVERSAL_REFERENCE, 0)):
            r.traversal, r.extra_flags = trav, flags
            rgb2 = r.run_view(eye, iv, fov, width=w, height=h)
            assert np.array_equal(rgb2, rgb) and np.array_equal(util.bits(r.mean_buffer), util.bits(mean)), (trav, flags)
            assert r.stats["rays"] == st["rays"] and r.stats["rays_untraced"] == (st["rays_untraced"] if (trav, flags) == (crt.TRAVERSAL_FAST, 0) else 0)
        for crop in ((392, 300, 16, 12), (200, 140, 16, 12), (w - 16, h - 12, 16, 12)):
            x0, y0, cw, ch = crop
            orgb, omean, _, _ = _oracle_crop(name, w, h, spp, crop)
            assert np.array_equal(util.bits(mean[y0:y0 + ch, x0:x0 + cw]), util.bits(omean)), crop
            assert np.array_equal(rgb[y0:y0 + ch, x0:x0 + cw], orgb), crop
    finally:
        r.free()

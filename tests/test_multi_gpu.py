"""GPU tests of the multi-device entry of libcrt.so (crt_multi_*, include/crt.h) and of bench.py's N > 1 paths.

The GPU box has ONE MI355X: several ranks share device 0 (CRT_GATHER_COPY -- RCCL refuses duplicate devices), which runs
everything but the collective itself; RCCL is initialised and its all-gather executed with a one-rank communicator.
Whatever the number of ranks, the frame must be the one crt_render produces on one device (pixels are independent and the
RNG is keyed by the global pixel index: SURVEY 8(e); the reference has no multi-GPU path, src/main.cu:92-105)."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import cudaraytracing_amd as crt
from cudaraytracing_amd import _capi as capi
import util

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def single():
    out = {}
    for name in ("cornell-box", "veach-mis"):
        t = util.task(name)
        out[name] = crt.Render(util.host_scene(name), 2, t.P_RR, t.light_sample_n, device=0)
    yield out
    for r in out.values():
        r.free()


def _multi(name, devices, gather, spp):
    t = util.task(name)
    return crt.MultiRender(util.host_scene(name), spp, t.P_RR, t.light_sample_n, devices=devices, gather=gather)


@pytest.mark.parametrize("name,w,h,ranks", [("cornell-box", 200, 152, 2), ("veach-mis", 100, 75, 3), ("cornell-box", 64, 40, 8)])
def test_ranks_on_one_device_reproduce_the_single_device_frame(single, name, w, h, ranks):
    eye, iv, fov = util.camera(name)
    one = single[name]
    one.set_spp(4)
    one.traversal = crt.TRAVERSAL_EXACT
    rgb = one.run_view(eye, iv, fov, width=w, height=h).copy()
    mean = one.mean_buffer.copy()
    m = _multi(name, [0] * ranks, crt.GATHER_COPY, 4)
    try:
        got = m.run_view(eye, iv, fov, width=w, height=h)
        assert np.array_equal(got, rgb)
        assert np.array_equal(util.bits(m.mean_buffer), util.bits(mean))
        assert m.info["n_ranks"] == ranks and m.info["gather"] == crt.GATHER_COPY and m.info["rccl_ranks"] == 0
        assert m.info["rays"] == one.stats["rays"] == sum(s["rays"] for s in m.rank_stats)
        assert m.info["paths"] == w * h * 4
        # the frame stays on rank 0's device when no host buffer is given
        m.seed = 5
        assert m.run_view(eye, iv, fov, width=w, height=h, to_host=False) is None
        one.seed = 5
        rgb5 = one.run_view(eye, iv, fov, width=w, height=h).copy()
        one.seed = 0
        d_rgb, d_mean, dev = C.c_void_p(), C.c_void_p(), C.c_int(-1)
        capi.check(capi.lib().crt_multi_frame_device(m._mh, C.byref(d_rgb), C.byref(d_mean), C.byref(dev)), "crt_multi_frame_device")
        assert dev.value == 0 and d_rgb.value
        m.seed = 5
        assert np.array_equal(m.run_view(eye, iv, fov, width=w, height=h), rgb5)
    finally:
        m.free()


def test_rccl_all_gather_runs_on_a_one_rank_communicator(single):
    """ncclCommInitAll + ncclAllGather through the run-time binding of librccl.so.1 (one rank: this box has one GPU)."""
    name = "cornell-box"
    eye, iv, fov = util.camera(name)
    one = single[name]
    one.set_spp(2)
    one.traversal = crt.TRAVERSAL_EXACT
    rgb = one.run_view(eye, iv, fov, width=160, height=120).copy()
    m = _multi(name, [0], crt.GATHER_RCCL, 2)
    try:
        got = m.run_view(eye, iv, fov, width=160, height=120)
        assert m.info["gather"] == crt.GATHER_RCCL and m.info["rccl_ranks"] == 1 and m.info["rccl_version"] > 0
        assert m.info["bytes_per_rank"] >= 160 * 120 * 3
        assert np.array_equal(got, rgb)
        assert np.array_equal(util.bits(m.mean_buffer), util.bits(one.mean_buffer))
    finally:
        m.free()


def test_multi_entry_reports_errors():
    lib = capi.lib()
    sc = util.host_scene("cornell-box")
    h = C.c_void_p()
    two = (C.c_int * 2)(0, 0)
    assert lib.crt_multi_create(C.byref(sc.desc()), two, 2, crt.GATHER_RCCL, C.byref(h)) == -1  # RCCL: one rank per device
    assert b"duplicate" in lib.crt_last_error()
    bad = (C.c_int * 1)(99)
    assert lib.crt_multi_create(C.byref(sc.desc()), bad, 1, crt.GATHER_AUTO, C.byref(h)) == -1
    assert lib.crt_multi_create(C.byref(sc.desc()), two, 0, crt.GATHER_AUTO, C.byref(h)) == -1
    assert lib.crt_multi_create(C.byref(sc.desc()), two, 2, 9, C.byref(h)) == -1
    assert lib.crt_multi_render(None, None, None, None, None, None, None) == -1
    m = _multi("cornell-box", [0, 0], crt.GATHER_AUTO, 2)  # AUTO with a repeated device: peer copies
    try:
        eye, iv, fov = util.camera("cornell-box")
        m.set_spp(0)
        with pytest.raises(crt.CrtError) as e:
            m.run_view(eye, iv, fov, width=16, height=16)
        assert "rank 0" in str(e.value) or "rank 1" in str(e.value)
    finally:
        m.free()


def _bench(args, env_extra, timeout=900):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", **env_extra)
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    r = subprocess.run([sys.executable, os.path.join(util.ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout,
                       env=env, cwd=util.ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [x for x in r.stdout.splitlines() if x.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_started_plainly_starts_its_own_ranks(single, tmp_path):
    """`python bench.py --gpus 2` without a launcher: the parent starts two ranks of itself before touching a GPU (here both
    on the one GPU, gathering over gloo); the line must carry the whole frame's rays and the one-device image."""
    from PIL import Image
    png = str(tmp_path / "two.png")
    line = _bench(["--gpus", "2", "--steps", "1", "--warmup", "0", "--spp", "4", "--width", "200", "--height", "152", "--no-cpu-baseline",
                   "--save-png", png], {"CRT_BENCH_ONE_DEVICE": "1"})
    assert line["n_gpus"] == 2 and line["collective"]["launched_by"].startswith("bench.py")
    assert line["collective"]["backend"] == "gloo" and line["collective"]["rccl_ranks"] == 0
    eye, iv, fov = util.camera("cornell-box")
    one = single["cornell-box"]
    one.set_spp(4)
    one.traversal = crt.TRAVERSAL_EXACT
    rgb = one.run_view(eye, iv, fov, width=200, height=152)
    assert line["rays_per_frame"] == one.stats["rays"]
    assert np.array_equal(np.asarray(Image.open(png)), rgb)
    assert line["fast_vs_reference"]["pixels_differ_f32_bits"] == 0 and line["fast_vs_reference"]["rays_equal"]


def test_bench_c4_resolution_on_four_self_started_ranks(single, tmp_path):
    """the N > 1 line at C4's resolution (3840 x 2160, spp 2) with self-started ranks sharing the one GPU and gathering over
    gloo -- FOUR of them: the box allows six processes on its card and the test runner is one (eight ranks in ONE process:
    test_ranks_on_one_device_...[cornell-box-64-40-8] and --engine multi).  The gathered PNG must be the one-device frame, the line must carry
    `per_rank` and say what its collective really spanned."""
    from PIL import Image
    png = str(tmp_path / "c4.png")
    line = _bench(["--gpus", "4", "--workload", "c4", "--spp", "2", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-c3", "--no-large-scene",
                   "--save-png", png], {"CRT_BENCH_ONE_DEVICE": "1"})
    assert line["n_gpus"] == 4 and line["scaling"] == "strong" and line["collective"]["launched_by"].startswith("bench.py")
    ev = line["collective"]["rccl_evidence"]
    assert line["collective"]["backend"] == "gloo" and line["collective"]["rccl_ranks"] == 0 and ev is None
    assert line["collective"]["proof"]["ranks_counted_by_all_reduce"] == 4
    assert line["collective"]["proof"]["distinct_devices"] == 1   # (four ranks on ONE physical device: the key is the device, not the rank -- ADVICE r05)
    assert line["collective"]["proof"]["ranks_with_unknown_device"] == 0
    pr = line["per_rank"]
    assert 0 < pr["kernel_ms_min"] <= pr["kernel_ms_max"] and "non_kernel_ms_per_step" in pr
    eye, iv, fov = util.camera("cornell-box")
    one = single["cornell-box"]
    one.set_spp(2)
    one.traversal = crt.TRAVERSAL_EXACT
    rgb = one.run_view(eye, iv, fov, width=3840, height=2160)
    assert line["rays_per_frame"] == one.stats["rays"]
    assert np.array_equal(np.asarray(Image.open(png)), rgb)


def test_auto_gather_records_why_it_fell_back(single, monkeypatch):
    """CRT_GATHER_AUTO never fails a frame because of RCCL: distinct devices whose communicator cannot be made render by peer copies and
    crt_multi_info.fallback_reason says why (ABI 5).  AUTO takes RCCL only with two or more distinct devices, which this one-GPU box cannot
    offer: with one device, or a repeated one, it is peer copies by rule (no attempt, no reason).  The fallback itself runs here through the
    library's test hooks (crt_multi.hip): CRT_TEST_RCCL_AUTO_ONE_DEVICE=1 makes AUTO attempt a one-rank communicator, CRT_TEST_RCCL_FAIL
    fails its binding / its creation (after the real communicator was made: the clean-up runs) / its rank count.  Every fallback renders the
    one-device frame; an explicit CRT_GATHER_RCCL with the same failure fails loudly and leaves the library usable."""
    eye, iv, fov = util.camera("cornell-box")
    one = single["cornell-box"]
    one.set_spp(2)
    one.traversal = crt.TRAVERSAL_EXACT
    want_rgb = one.run_view(eye, iv, fov, width=64, height=40).copy()
    for k in ("CRT_TEST_RCCL_AUTO_ONE_DEVICE", "CRT_TEST_RCCL_FAIL"):
        monkeypatch.delenv(k, raising=False)
    for devices in ([0], [0, 0]):
        m = _multi("cornell-box", devices, crt.GATHER_AUTO, 2)
        try:
            rgb = m.run_view(eye, iv, fov, width=64, height=40)
            assert m.info["fallback_reason"] == "" and m.info["gather"] == crt.GATHER_COPY and m.info["rccl_ranks"] == 0
            assert np.array_equal(rgb, want_rgb)
        finally:
            m.free()
    monkeypatch.setenv("CRT_TEST_RCCL_AUTO_ONE_DEVICE", "1")
    m = _multi("cornell-box", [0], crt.GATHER_AUTO, 2)          # no failure: a one-rank RCCL communicator
    try:
        rgb = m.run_view(eye, iv, fov, width=64, height=40)
        assert m.info["fallback_reason"] == "" and m.info["gather"] == crt.GATHER_RCCL and m.info["rccl_ranks"] == 1
        assert np.array_equal(rgb, want_rgb)
    finally:
        m.free()
    for how, text in (("load", "cannot load librccl"), ("init", "ncclCommInitAll"), ("count", "ncclCommCount reports 2 ranks for 1 devices")):
        monkeypatch.setenv("CRT_TEST_RCCL_FAIL", how)
        m = _multi("cornell-box", [0], crt.GATHER_AUTO, 2)
        try:
            rgb = m.run_view(eye, iv, fov, width=64, height=40)
            assert text in m.info["fallback_reason"], (how, m.info["fallback_reason"])
            assert m.info["gather"] == crt.GATHER_COPY and m.info["rccl_ranks"] == 0
            assert np.array_equal(rgb, want_rgb), how
        finally:
            m.free()
        with pytest.raises(Exception):                            # asked for by name, RCCL's failure is the caller's
            _multi("cornell-box", [0], crt.GATHER_RCCL, 2)
    monkeypatch.delenv("CRT_TEST_RCCL_FAIL")
    m = _multi("cornell-box", [0], crt.GATHER_RCCL, 2)           # ... and the library is as usable as before
    try:
        assert np.array_equal(m.run_view(eye, iv, fov, width=64, height=40), want_rgb) and m.info["rccl_ranks"] == 1
    finally:
        m.free()


def test_bench_multi_engine_one_process(single):
    """`--engine multi`: one process, crt_multi_render; two ranks on the one GPU (peer copies), one rank over RCCL."""
    line = _bench(["--gpus", "2", "--engine", "multi", "--steps", "2", "--warmup", "1", "--spp", "4", "--width", "200", "--height", "152",
                   "--no-cpu-baseline"], {"CRT_BENCH_ONE_DEVICE": "1"})
    assert line["n_gpus"] == 2 and line["multi_info"]["n_ranks"] == 2 and line["collective"]["backend"] == "copy"
    eye, iv, fov = util.camera("cornell-box")
    one = single["cornell-box"]
    one.set_spp(4)
    one.traversal = crt.TRAVERSAL_EXACT
    one.run_view(eye, iv, fov, width=200, height=152)
    assert line["rays_per_frame"] == one.stats["rays"]


def test_bench_default_line_has_parity_and_bounded_roofline():
    """The single-GPU line on a reduced workload: same-run parity gate against the oracle (all pixels), FAST == REFERENCE
    slice, and every priced bound's fraction <= 1 when PMC counters of this code are present."""
    line = _bench(["--steps", "1", "--warmup", "1", "--spp", "8", "--width", "160", "--height", "120", "--cpu-spp", "2", "--no-c3"], {})
    p = line["parity"]
    assert p["pixels_gt_1e-5"] == 0 and p["pixels_differ_f32_bits"] == 0 and p["rgb8_mismatch"] == 0 and p["rays_equal"]
    assert line["fast_vs_reference"]["pixels_differ_f32_bits"] == 0
    assert line["cpu_baseline"]["cores"] == 1 and line["cpu_baseline"]["kind"] == "port"
    rf = line["roofline"]
    assert rf["frac"] is None or rf["frac"] <= 1.0  # (PMC passes exist for the C2 workload only: null here)
    assert rf["contract_frac"] > 0 and "contract_note" in rf


def test_cli_multi_device(tmp_path):
    from PIL import Image
    from cudaraytracing_amd import build as b
    cli = b.build_cli()
    out = str(tmp_path / "m.png")
    cfg = util.SCENES["veach-mis"]
    r = subprocess.run([cli, cfg, "-o", out, "--spp", "2", "--width", "96", "--height", "72", "--seed", "42", "--base-dir", util.ROOT,
                        "--devices", "0,0,0", "--gather", "copy"], capture_output=True, text=True, timeout=180)
    assert r.returncode == 0, r.stderr
    assert "ranks: 3, gather: copy" in r.stdout
    t = util.task("veach-mis")
    eye, iv, fov = util.camera("veach-mis")
    orgb, _, _, st = util.oracle_scene("veach-mis").render(eye, iv, fov, 96, 72, 2, t.P_RR, t.light_sample_n, seed=42)
    assert np.array_equal(np.asarray(Image.open(out)), orgb)
    assert ("%d rays" % st["rays"]) in r.stdout
    r = subprocess.run([cli, cfg, "-o", out, "--spp", "1", "--width", "32", "--height", "24", "--base-dir", util.ROOT, "--gpus", "1",
                        "--gather", "rccl"], capture_output=True, text=True, timeout=180)
    assert r.returncode == 0, r.stderr
    assert "gather: rccl, rccl ranks: 1" in r.stdout

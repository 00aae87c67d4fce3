"""Hand-offs between kernels through device memory.  Two things the bit-exact frame comparisons of the other tests cannot see, because they
render one frame per Render object or the same frame again:
  * the FIRST frame of a Render created after other renders of the process -- its device buffers are whatever the allocator hands back, and
    the caches may still hold lines of their previous owners.  Round 4 found k_mega3 reading entries of the work-item list (k_order_items)
    as an earlier kernel had left them: 10 - 400 of 589 824 work items never ran, in half of such first frames, once the launches' timing
    had changed (the list is now written and read with agent-scope accesses; docs/experiments.md 6);
  * frames whose content differs from the frame before in the same Render (seed, camera, size): a stale line of the previous frame's
    radiance or path state would show, where a repeated frame hides it.
Every frame against the CPU oracle, bit for bit."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import cudaraytracing_amd as crt
import oracle_lib as O
import test_commit_ring as R
import util

pytestmark = pytest.mark.gpu


def test_first_frames_of_fresh_renders_after_other_work():
    name, w, h, spp = "cornell-box", 96, 64, 96
    t = util.task(name)
    eye, iv, fov = util.camera(name)
    _, omean, _, _ = util.oracle_scene(name).render(eye, iv, fov, w, h, spp, t.P_RR, t.light_sample_n)
    for rep in range(6):
        R.test_the_flag_picks_a_ring_and_one_launch()   # other renders of the same scene: sizes, flags and traversal modes of their own
        r = crt.Render(util.host_scene(name), spp, t.P_RR, t.light_sample_n, device=0)
        try:
            for frame in range(2):
                r.run_view(eye, iv, fov, width=w, height=h)
                bad = (util.bits(r.mean_buffer) != util.bits(omean)).any(axis=2)
                assert not bad.any(), "render %d, frame %d: %d pixels differ from the oracle (of them NaN: %d)" % (
                    rep, frame + 1, int(bad.sum()), int(np.isnan(r.mean_buffer).any(axis=2).sum()))
        finally:
            r.free()


@pytest.mark.parametrize("name", ["cornell-box", "veach-mis"])
def test_frames_that_differ_from_the_frame_before(name):
    t = util.task(name)
    eye, iv, fov = util.camera(name)
    osc = util.oracle_scene(name)
    r = crt.Render(util.host_scene(name), 24, t.P_RR, t.light_sample_n, device=0)
    try:
        for k, (w, h, spp, seed, dx) in enumerate([(96, 64, 24, 0, 0.0), (96, 64, 24, 1, 0.0), (96, 64, 24, 1, 0.3), (128, 96, 16, 2, 0.3),
                                                    (96, 64, 24, 0, 0.0), (64, 48, 40, 5, -0.2), (96, 64, 24, 7, 0.1)]):
            e = np.array(eye, dtype=np.float32).copy()
            e[0] += np.float32(dx)
            r.set_spp(spp)
            r.seed = seed
            rgb = r.run_view(e, iv, fov, width=w, height=h)
            orgb, omean, _, _ = osc.render(e, iv, fov, w, h, spp, t.P_RR, t.light_sample_n, seed=seed)
            assert np.array_equal(util.bits(r.mean_buffer), util.bits(omean)) and np.array_equal(rgb, orgb), (k, w, h, spp, seed, dx)
    finally:
        r.free()


def test_200_first_frames_of_alternating_renders_of_equal_size():
    """200 short-lived Render objects of equal size alternating between the two scenes (every fifth through the commit
    ring), first frames only, every new device allocation pre-filled with 0xFF bytes, against the oracle -- in a child process, because
    CRT_DEBUG_FILL is read when the library makes its first allocation."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, CRT_DEBUG_FILL="255")
    p = subprocess.run([sys.executable, os.path.join(root, "tests", "first_frame_stress_driver.py"), "200"], env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    res = json.loads(p.stdout.strip().splitlines()[-1])
    assert res["first_frames"] == 200 and res["debug_fill"] == "255"
    assert res["bad_frames"] == [], res

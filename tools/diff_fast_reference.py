#!/usr/bin/env python3
"""Finds the pixels on which CRT_TRAVERSAL_FAST and CRT_TRAVERSAL_REFERENCE disagree on a frame and asks the oracle about them."""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import cudaraytracing_amd as crt
import oracle_lib as O

ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="veach-mis")
ap.add_argument("--spp", type=int, default=1024)
ap.add_argument("--width", type=int, default=800)
ap.add_argument("--height", type=int, default=600)
ap.add_argument("--seed", type=int, default=0)
ap.add_argument("--rank", type=int, default=0)
ap.add_argument("--ranks", type=int, default=1, help="> 1: only the tile shard `rank` of `ranks` is rendered")
a = ap.parse_args()
t = crt.Task(os.path.join(ROOT, "scenes", a.scene, "config.json"), base_dir=ROOT)
sc = crt.Scene.from_task(t, a.width, a.height)
r = crt.Render(sc, a.spp, t.P_RR, t.light_sample_n)
r.seed = a.seed
iv = crt.get_inverse_view_matrix(t.eye_pos, t.lookat, t.up)
fov = crt.fov_to_radians(t.fov_y)
print("accel", r.accel_info())
res = {}
import ctypes as C
from cudaraytracing_amd import _capi as capi
from cudaraytracing_amd.distributed import untile_numpy


def shard_frame(trav, flags):
    """the shard's pixels placed into a full frame (other pixels 0)"""
    if a.ranks == 1:
        r.traversal, r.extra_flags = trav, flags
        r.run_view(t.eye_pos, iv, fov)
        return r.mean_buffer.copy(), dict(r.stats)
    r.traversal = trav
    slots = crt.shard_slots(a.width, a.height, a.rank, a.ranks)
    buf = np.zeros((slots, 3), dtype=np.uint8)
    mean = np.zeros((slots, 3), dtype=np.float32)
    prm = r._params(rank=a.rank, world=a.ranks, flags=capi.FLAG_TILED_OUTPUT | flags, width=a.width, height=a.height)
    st = capi.Stats()
    capi.check(capi.lib().crt_render(r._h, C.byref(r._cam(t.eye_pos, iv, fov)), C.byref(prm), capi.ptr(buf), capi.ptr(mean), C.byref(st)), "crt_render")
    g = np.zeros((a.ranks, slots, 3), dtype=np.float32)
    g[a.rank] = mean
    return untile_numpy(g, a.width, a.height), st.as_dict()


for name, trav, flags in (("fast", crt.TRAVERSAL_FAST, 0), ("fast_all", crt.TRAVERSAL_FAST, crt.FLAG_TRACE_ALL), ("ref", crt.TRAVERSAL_REFERENCE, 0)):
    res[name] = shard_frame(trav, flags)
    print(name, "rays", res[name][1]["rays"], "shadow", res[name][1]["shadow_rays"])
bits = lambda x: x.view(np.uint32)
d = np.argwhere(np.any(bits(res["fast"][0]) != bits(res["ref"][0]), axis=2))
print("pixels fast != ref:", len(d), "; fast != fast_all:", int(np.count_nonzero(np.any(bits(res["fast"][0]) != bits(res["fast_all"][0]), axis=2))))
osc = O.OracleScene(t.OBJ_paths, t.bvh_thresh_n)
for (y, x) in d[:12]:
    _, om, L, _ = osc.render(t.eye_pos, iv, fov, a.width, a.height, a.spp, t.P_RR, t.light_sample_n, seed=a.seed, crop=(int(x), int(y), 1, 1), want_L=True)
    f, rf = res["fast"][0][y, x], res["ref"][0][y, x]
    print("pixel", (int(x), int(y)), "oracle", om[0, 0], "fast", f, "ref", rf, "| fast==oracle", bool(np.array_equal(bits(f), bits(om[0, 0]))),
          "ref==oracle", bool(np.array_equal(bits(rf), bits(om[0, 0]))))

# ---- which ray?  every ray the oracle traces for a differing pixel, replayed through crt_intersect in both modes: closest-hit
# rays as closest-hit queries, next-event samples as visibility queries with their own t_to_light ----
EPS = np.float32(0.00001)  # Global.h:11
for (y, x) in [(int(v[0]), int(v[1])) for v in d[:4]]:
    L = O.lib()
    L.orc_ray_log_begin.restype = None
    L.orc_ray_log_end.restype = C.c_uint64
    L.orc_ray_log_end.argtypes = [C.c_void_p, C.c_uint64]
    L.orc_ray_log_begin()
    osc.render(t.eye_pos, iv, fov, a.width, a.height, a.spp, t.P_RR, t.light_sample_n, seed=a.seed, crop=(x, y, 1, 1))
    n = int(L.orc_ray_log_end(None, 0))
    log = np.zeros((n, 10), dtype=np.float32)
    L.orc_ray_log_end(log.ctypes.data_as(C.c_void_p), n)
    o, dd = np.ascontiguousarray(log[:, 0:3]), np.ascontiguousarray(log[:, 3:6])
    ot, otri, lim, entry = np.ascontiguousarray(log[:, 6]), log[:, 7].astype(np.int32), np.ascontiguousarray(log[:, 8]), log[:, 9]
    vis = ~np.isnan(lim)
    RAW = crt.INTERSECT_RAW_DIRECTIONS  # the logged directions are a Ray's own: not normalised again
    bad, ans = {}, {}
    for name, trav in (("fast", crt.TRAVERSAL_FAST), ("ref", crt.TRAVERSAL_REFERENCE)):
        tri_c, t_c = r.intersect(o[~vis], dd[~vis], traversal=trav | RAW)
        wrong = np.zeros(n, dtype=bool)
        wrong[np.nonzero(~vis)[0]] = (tri_c != otri[~vis]) | (bits(t_c) != bits(ot[~vis]))
        if vis.any():
            with np.errstate(invalid="ignore", over="ignore"):
                o_blocked = (lim[vis] - ot[vis]) > EPS
            blk, btri = r.blocked(o[vis], dd[vis], lim[vis], traversal=trav | RAW)
            wrong[np.nonzero(vis)[0]] = blk != o_blocked
            ans[name] = (np.nonzero(vis)[0], blk, btri, o_blocked)
        bad[name] = wrong
    print("pixel", (x, y), "rays:", n, "(visibility:", int(vis.sum()), ") | reference mode != oracle:", int(bad["ref"].sum()), "| fast != oracle:", int(bad["fast"].sum()))
    tr_ = osc.tris()
    for i in np.nonzero(bad["fast"])[0][:8]:
        inv = np.float32(1) / dd[i]
        print("  ray bits", " ".join(str(int(v)) for v in np.concatenate([o[i], dd[i], lim[i:i + 1]]).view(np.uint32)))
        print("  ray", int(i), "visibility" if vis[i] else "closest", "o", o[i], "d", dd[i], "inv", inv, "limit", lim[i], "oracle (t, tri)", ot[i], otri[i], "box entry of its leaf", entry[i])
        if vis[i]:
            for name in ans:
                k = int(np.searchsorted(ans[name][0], i))
                print("     ", name, "blocked", bool(ans[name][1][k]), "by", int(ans[name][2][k]), "| oracle blocked", bool(ans[name][3][k]), "limit - t =", float(lim[i] - ot[i]))
        ti = int(otri[i])
        if ti >= 0:
            v1, v2, v3 = tr_["v1"][ti], tr_["v2"][ti], tr_["v3"][ti]
            nrm = np.cross((v2 - v1).astype(np.float64), (v3 - v1).astype(np.float64))
            det = float(np.dot(dd[i].astype(np.float64), nrm))
            print("     tri", ti, "v1", v1, "v2", v2, "v3", v3, "| d.N =", det, "|N| =", float(np.linalg.norm(nrm)), "cos =", det / float(np.linalg.norm(nrm)))
            ref_t = lim[i] if vis[i] else ot[i]
            reach = float(np.abs(o[i]).max() + abs(ref_t)); steep = float(np.abs(inv).max())
            print("     bound needs: entry - t_ref =", float(entry[i]) - float(ref_t), "| reach*steep =", reach * steep, "| ratio =", (float(entry[i]) - float(ref_t)) / (reach * steep))

"""Two closed-form anchors for the oracle's INTEGRATOR (oracle/crt_oracle.cpp: cast_ray, after include/Render.cuh:199-326) that do not
go through its code: the expectation of the reference's estimator, written down from the formulas of the reference and integrated by
numerical quadrature in float64 numpy, against the oracle's mean radiance at high spp (within 4 standard errors of that mean, the
per-sample radiances being the oracle's own `want_L` output).

1. P_RR = 0: `u > P_RR` (Render.cuh:216-221, u in (0, 1]) always holds, every path stops at its first vertex, and a pixel is direct light
   only.  The estimator of Render.cuh:259-283 with DeviceLights.cuh:33-37 / DeviceTriangle.cuh:67-74: a triangle of the light uniformly
   by COUNT (not area), the point alpha = u1, beta = u2 (1 - alpha), gamma = 1 - alpha - beta (not uniform on the triangle),
   inv_pdf = area of the whole light object -- a biased estimator whose expectation is
       E = sum_lights (1 / n_tri) sum_tri  INT INT  ke (.) kd / pi * cos(theta) cos(theta') * area_obj / r^2 * V  du1 du2 .
   The light here is a trapezoid of two triangles of different areas, so count-against-area and the non-uniform point both matter.
2. One bounce, P_RR = 0.5: floor -> underside of a shelf whose normal points up, away from the floor, lit from above.  A bounce that
   leaves the shelf can only reach the emitter (which contributes nothing below depth 0, Render.cuh:249-255) or escape, so the
   recursion ends there and
       E = E_direct(x0) + INT_hemisphere L_direct(x1(w)) (.) kd_floor / pi * cos(theta) dw
   -- the indirect term L (.) f_r * cos * 2 pi / P_RR of Render.cuh:288-293 under the uniform hemisphere sampler (Global.h:57-66) and the
   roulette, whose 1 / P_RR and P_RR cancel.

This does not pin the oracle to the reference (nothing can: SURVEY 8c); it makes a gross error of the restatement -- a missing cosine,
pi, 1 / lsn, the wrong pdf -- impossible to miss."""
import os

import numpy as np

import oracle_lib as O

PI = np.pi


def _write(d, name, quads, mtl):
    """quads: (material, 4 corners); two triangles (a, b, c), (a, c, e) each, vertex order as given (the normal is cross(e1, e2))."""
    v, f = [], []
    for m, (a, b, c, e) in quads:
        i = len(v)
        v.extend([a, b, c, e])
        f.append((m, i + 1, i + 2, i + 3))
        f.append((m, i + 1, i + 3, i + 4))
    with open(os.path.join(d, name + ".mtl"), "w") as o:
        o.write(mtl)
    with open(os.path.join(d, name + ".obj"), "w") as o:
        o.write("mtllib %s.mtl\n" % name)
        for p in v:
            o.write("v %.9g %.9g %.9g\nvn 0 1 0\nvt 0 0\n" % tuple(p))
        cur = None
        for m, a, b, c in f:
            if m != cur:
                o.write("usemtl %s\n" % m)
                cur = m
            o.write("f %d/%d/%d %d/%d/%d %d/%d/%d\n" % (a, a, a, b, b, b, c, c, c))
    return os.path.join(d, name + ".obj"), d


def _tris_of(quad):
    a, b, c, e = (np.asarray(p, dtype=np.float64) for p in quad)
    return [(a, b, c), (a, c, e)]


def _normal(t):
    n = np.cross(t[1] - t[0], t[2] - t[0])
    return n / np.linalg.norm(n)


def _area(t):
    return 0.5 * np.linalg.norm(np.cross(t[1] - t[0], t[2] - t[0]))


def _camera_points(eye, inv_view, fov, W, H, i, j, ja, jb):
    """Camera rays of pixel (i, j) for jitters (ja, jb) in (0, 1] (Render.cuh:344-347) and where they meet the plane y = 0."""
    M = np.asarray(inv_view, dtype=np.float64).reshape(3, 3).T  # column-major 9 floats
    scale, ar = np.tan(fov / 2.0), W / float(H)
    x = (2.0 * (i + ja) / W - 1.0) * scale * ar
    y = (1.0 - 2.0 * (j + jb) / H) * scale
    cd = np.stack([-x, y, np.ones_like(x)], axis=-1)
    cd /= np.linalg.norm(cd, axis=-1, keepdims=True)
    d = cd @ M.T
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    t = -eye[1] / d[..., 1]
    assert (t > 0).all()
    return eye + t[..., None] * d


def _direct(x, n, light_tris, ke, kd, blocker=None, nu=48, unbiased=False):
    """E[L_dir] at the points x (..., 3) with normal n for ONE light object: midpoint rule over (u1, u2).  unbiased=True: the integral
    over the light's AREA instead (the map (u1, u2) -> point has the area element 2 A_tri (1 - u1) du1 du2), which is what an unbiased
    estimator would converge to -- not what the reference computes."""
    u = (np.arange(nu) + 0.5) / nu
    u1, u2 = np.meshgrid(u, u, indexing="ij")
    al, be = u1, u2 * (1.0 - u1)
    ga = 1.0 - al - be
    area_obj = sum(_area(t) for t in light_tris)
    out = np.zeros(x.shape[:-1] + (3,))
    for t in light_tris:
        lp = al[..., None] * t[0] + be[..., None] * t[1] + ga[..., None] * t[2]          # (nu, nu, 3)
        dist = lp[None, ...] - x.reshape(-1, 1, 1, 3)                                      # (P, nu, nu, 3)
        r2 = (dist ** 2).sum(-1)
        dirn = dist / np.sqrt(r2)[..., None]
        c1 = np.clip((dirn * n).sum(-1), 0.0, None)
        c2 = np.clip(-(dirn * _normal(t)).sum(-1), 0.0, None)
        vis = np.ones_like(c1)
        if blocker is not None:  # an axis-aligned rectangle in the plane y = h: (h, x0, x1, z0, z1)
            h, bx0, bx1, bz0, bz1 = blocker
            xs = x.reshape(-1, 1, 1, 3)
            s = (h - xs[..., 1]) / dist[..., 1]
            px, pz = xs[..., 0] + s * dist[..., 0], xs[..., 2] + s * dist[..., 2]
            hit = (s > 1e-9) & (s < 1.0 - 1e-9) & (px > bx0) & (px < bx1) & (pz > bz0) & (pz < bz1)
            vis = np.where(hit, 0.0, 1.0)
        g = c1 * c2 * vis / r2
        if unbiased:
            g = g * (2.0 * _area(t) * (1.0 - u1) * len(light_tris) / area_obj)
        w = g.mean(axis=(1, 2)) * area_obj / len(light_tris)                              # (P,)
        out += (w[:, None] * (ke * kd / PI)).reshape(out.shape)
    return out


def _oracle_pixels(osc, eye, iv, fov, W, H, spp, p_rr, lsn, seeds):
    """Mean radiance per pixel and its standard error, from the oracle's per-sample radiances."""
    Ls = []
    for s in seeds:
        _, _, L, _ = osc.render(eye, iv, fov, W, H, spp, p_rr, lsn, seed=s, want_L=True)
        Ls.append(L.astype(np.float64))
    L = np.concatenate(Ls, axis=2)                       # (H, W, samples, 3)
    return L.mean(axis=2), L.std(axis=2, ddof=1) / np.sqrt(L.shape[2])


MTL = "newmtl floor\nKd 0.5 0.6 0.7\nNs 1\nnewmtl shelf\nKd 0.8 0.7 0.3\nNs 1\nnewmtl light\nKe 20 15 10\nKd 0 0 0\nNs 1\n"
KD_FLOOR, KD_SHELF, KE = np.array([0.5, 0.6, 0.7]), np.array([0.8, 0.7, 0.3]), np.array([20.0, 15.0, 10.0])
FLOOR = [(-50, 0, -50), (-50, 0, 50), (50, 0, 50), (50, 0, -50)]         # cross(e1, e2) = +y
UP = np.array([0.0, 1.0, 0.0])


def test_direct_light_is_the_expectation_of_the_reference_estimator(tmp_path):
    # a trapezoid facing down (cross(e1, e2) = -y): its two triangles have areas 3 and 1.5
    light = [(-1, 4, -1), (2, 4, -1), (0.5, 4, 1), (-1, 4, 1)]
    obj, mtl = _write(str(tmp_path), "direct", [("floor", FLOOR), ("light", light)], MTL)
    osc = O.OracleScene([(obj, mtl)], 2)
    lt = _tris_of(light)
    assert osc.num_lights == 1 and abs(_area(lt[0]) - 3.0) < 1e-12 and abs(_area(lt[1]) - 1.5) < 1e-12
    assert (_normal(lt[0]) == -UP).all() and (_normal(_tris_of(FLOOR)[0]) == UP).all()
    eye = np.array([0.7, 3.0, -2.5], dtype=np.float32)
    iv = O.inverse_view(eye, [0.3, 0.0, 0.4], [0.0, 1.0, 0.0])
    fov, W, H, lsn = np.float32(np.deg2rad(50.0)), 4, 3, 2
    mean, se = _oracle_pixels(osc, eye, iv, fov, W, H, 4096, 0.0, lsn, seeds=(1, 2))
    nj = 4
    j = (np.arange(nj) + 0.5) / nj
    ja, jb = np.meshgrid(j, j, indexing="ij")
    for py in range(H):
        for px in range(W):
            x0 = _camera_points(eye.astype(np.float64), iv, float(fov), W, H, px, py, ja, jb)
            want = _direct(x0, UP, lt, KE, KD_FLOOR).mean(axis=(0, 1))
            assert (want > 0.05).all()
            err = np.abs(mean[py, px] - want)
            assert (err <= 4.0 * se[py, px] + 2e-3 * want).all(), (px, py, mean[py, px], want, se[py, px])
    # and the comparison has teeth: the integral over the light's area (what an unbiased estimator converges to) is far outside
    x0 = _camera_points(eye.astype(np.float64), iv, float(fov), W, H, 1, 1, ja, jb)
    by_area = _direct(x0, UP, lt, KE, KD_FLOOR, unbiased=True).mean(axis=(0, 1))
    assert (np.abs(by_area - mean[1, 1]) > 20.0 * se[1, 1]).all()


def test_one_bounce_is_the_expectation_of_the_reference_estimator(tmp_path):
    h = 1.5
    shelf = [(-1.5, h, -1.5), (-1.5, h, 1.5), (1.5, h, 1.5), (1.5, h, -1.5)]   # normal +y: away from the floor
    light = [(-1, 4, -1), (1, 4, -1), (1, 4, 1), (-1, 4, 1)]                    # normal -y
    obj, mtl = _write(str(tmp_path), "bounce", [("floor", FLOOR), ("shelf", shelf), ("light", light)], MTL)
    osc = O.OracleScene([(obj, mtl)], 2)
    lt = _tris_of(light)
    assert (_normal(_tris_of(shelf)[0]) == UP).all() and (_normal(lt[0]) == -UP).all()
    eye = np.array([0.4, 0.9, -3.0], dtype=np.float32)   # below the shelf, looking at the floor under its edge
    iv = O.inverse_view(eye, [0.2, 0.0, -0.9], [0.0, 1.0, 0.0])
    fov, W, H, lsn, p_rr = np.float32(np.deg2rad(40.0)), 2, 2, 1, 0.5
    mean, se = _oracle_pixels(osc, eye, iv, fov, W, H, 16384, p_rr, lsn, seeds=(1, 2, 3, 4))
    blocker = (h, -1.5, 1.5, -1.5, 1.5)
    # L_direct on the shelf's top is smooth (nothing between shelf and light): a 65 x 65 table, interpolated
    from scipy.interpolate import RegularGridInterpolator
    gx = np.linspace(-1.5, 1.5, 65)
    GX, GZ = np.meshgrid(gx, gx, indexing="ij")
    table = _direct(np.stack([GX, np.full_like(GX, h), GZ], axis=-1), UP, lt, KE, KD_SHELF, nu=40)
    Ld_shelf = [RegularGridInterpolator((gx, gx), table[..., c], bounds_error=False, fill_value=None) for c in range(3)]
    nj, nh = 6, 400
    j = (np.arange(nj) + 0.5) / nj
    ja, jb = np.meshgrid(j, j, indexing="ij")
    hs = (np.arange(nh) + 0.5) / nh
    s1, s2 = np.meshgrid(hs, hs, indexing="ij")
    z = np.abs(1.0 - 2.0 * s1)                                        # Global.h:57-66: uniform over the hemisphere, pdf 1 / 2 pi
    r = np.sqrt(1.0 - z * z)
    w = np.stack([r * np.cos(2 * PI * s2), z, r * np.sin(2 * PI * s2)], axis=-1)   # around +y
    for py in range(H):
        for px in range(W):
            x0 = _camera_points(eye.astype(np.float64), iv, float(fov), W, H, px, py, ja, jb)   # (nj, nj, 3)
            direct = _direct(x0, UP, lt, KE, KD_FLOOR, blocker=blocker, nu=128).mean(axis=(0, 1))
            indirect = np.zeros(3)
            for q in x0.reshape(-1, 3):
                s = (h - q[1]) / w[..., 1]
                p1 = q + s[..., None] * w
                on = (np.abs(p1[..., 0]) < 1.5) & (np.abs(p1[..., 2]) < 1.5)
                xz = p1[on][:, [0, 2]]
                Ld = np.stack([f(xz) for f in Ld_shelf], axis=-1)
                # L (.) f_r * cos * 2 pi / P_RR, taken with probability P_RR, averaged over the uniform hemisphere
                indirect += (Ld * (KD_FLOOR / PI) * z[on][:, None] * 2.0 * PI).sum(axis=0) / z.size
            indirect /= nj * nj
            want = direct + indirect
            assert (indirect > 0.02).all()                            # the bounce is what this pixel is made of
            err = np.abs(mean[py, px] - want)
            assert (err <= 4.0 * se[py, px] + 3e-3 * want).all(), (px, py, mean[py, px], want, se[py, px], direct)


def test_specular_probe_is_the_expectation_of_the_reference_estimator(tmp_path):
    """3. The SPECULAR emitter probe (Render.cuh:294-314): the one term of the integrator that had no check outside the
    author's own reading.  A plate with Ns = 100 (Loader.h:107: Ns > 1 is SPECULAR) under a rectangular emitter and a huge BLACK ceiling
    (kd = 0: whatever comes back from it is multiplied by f_r = 0, so the indirect term of the plate is exactly zero -- but a bounce that
    meets it makes the plate a vertex that is not the path's last, which is when the probe is evaluated, :289, :294).  For a pixel on
    the plate
        E = E_direct(x0) + P_RR * p(x0) * E_eta[ hit(eta) * (0.5 log10(ns) + 1) * ke (.) kd * max(dir_y, 0) * 2 pi / 8 ] ,
    p = the share of the uniform hemisphere (Global.h:57-66) in which the bounce meets the emitter or the ceiling, eta = (eta1, eta2) uniform
    on (-1, 1]^2, dir(eta) the lobe sample of Global.h:68-94 around the mirror direction of the camera ray -- polar angle from +z, azimuth
    in the xy plane, half-widths (exp(25 / ns) - 1) / (e - 1) * (30, 120) degrees --, hit = "the probe's first surface is the emitter".
    The emitter cuts the lobe's footprint, so the lobe's frame, its widths, the shininess factor, the cosine and the 2 pi / 8 all show."""
    ns = 100.0
    mtl = "newmtl plate\nKd 0.6 0.5 0.4\nNs 100\nnewmtl black\nKd 0 0 0\nNs 1\nnewmtl light\nKe 20 15 10\nKd 0 0 0\nNs 1\n"
    kd_plate = np.array([0.6, 0.5, 0.4])
    plate = [(-2, 0, -2), (-2, 0, 2), (2, 0, 2), (2, 0, -2)]                       # normal +y
    hl, hc = 3.0, 10.0
    lx0, lx1, lz0, lz1 = -0.6, 1.5, 5.2, 7.0
    light = [(lx0, hl, lz0), (lx1, hl, lz0), (lx1, hl, lz1), (lx0, hl, lz1)]       # normal -y
    big = 600.0
    ceiling = [(-big, hc, -big), (big, hc, -big), (big, hc, big), (-big, hc, big)]  # normal -y
    obj, mdir = _write(str(tmp_path), "probe", [("plate", plate), ("black", ceiling), ("light", light)], mtl)
    osc = O.OracleScene([(obj, mdir)], 2)
    lt = _tris_of(light)
    assert osc.num_lights == 1 and (_normal(lt[0]) == -UP).all() and (_normal(_tris_of(plate)[0]) == UP).all()
    eye = np.array([0.8, 2.0, -4.0], dtype=np.float32)
    iv = O.inverse_view(eye, [0.3, 0.0, 0.2], [0.0, 1.0, 0.0])
    fov, W, H, lsn, p_rr = np.float32(np.deg2rad(8.0)), 2, 2, 1, 0.5
    mean, se = _oracle_pixels(osc, eye, iv, fov, W, H, 16384, p_rr, lsn, seeds=(1, 2, 3, 4))
    delta = (np.exp(25.0 / ns) - 1.0) / (np.e - 1.0)
    d_theta, d_phi = delta * 30.0 * PI / 180.0, delta * 120.0 * PI / 180.0
    shininess = 0.5 * np.log10(ns) + 1.0
    nj, ne, nh = 4, 500, 600
    j = (np.arange(nj) + 0.5) / nj
    ja, jb = np.meshgrid(j, j, indexing="ij")
    e = -1.0 + 2.0 * (np.arange(ne) + 0.5) / ne
    e1, e2 = np.meshgrid(e, e, indexing="ij")
    hs = (np.arange(nh) + 0.5) / nh
    s1, s2 = np.meshgrid(hs, hs, indexing="ij")
    z = np.abs(1.0 - 2.0 * s1)
    rr = np.sqrt(1.0 - z * z)
    w = np.stack([rr * np.cos(2 * PI * s2), z, rr * np.sin(2 * PI * s2)], axis=-1)   # uniform hemisphere around +y (to_world of N = +y)

    def meets(q, d, h, x0, x1, z0, z1):
        s = (h - q[1]) / np.where(d[..., 1] > 1e-12, d[..., 1], np.inf)
        px, pz = q[0] + s * d[..., 0], q[2] + s * d[..., 2]
        return (d[..., 1] > 1e-12) & (px > x0) & (px < x1) & (pz > z0) & (pz < z1)

    eye64 = eye.astype(np.float64)
    for py in range(H):
        for px in range(W):
            x0 = _camera_points(eye64, iv, float(fov), W, H, px, py, ja, jb)          # (nj, nj, 3) on the plate
            assert (np.abs(x0[..., 0]) < 2).all() and (np.abs(x0[..., 2]) < 2).all()
            direct = _direct(x0, UP, lt, KE, kd_plate, nu=96).mean(axis=(0, 1))
            probe = np.zeros(3)
            cut = []
            for q in x0.reshape(-1, 3):
                # the bounce leaves the plate as a vertex that is not the last one iff it meets the emitter or the ceiling
                p_nf = (meets(q, w, hl, lx0, lx1, lz0, lz1) | meets(q, w, hc, -big, big, -big, big)).mean()
                din = (q - eye64) / np.linalg.norm(q - eye64)                          # from_dir of vertex 0
                out = din - 2.0 * din.dot(UP) * UP
                th0 = np.arccos(out[2] / np.linalg.norm(out))
                ph0 = (PI / 2 if out[1] > 0 else -PI / 2) if abs(out[0]) < 1e-5 else np.arctan2(out[1], out[0])
                th, ph = th0 + e1 * d_theta, ph0 + e2 * d_phi
                d = np.stack([np.sin(th) * np.cos(ph), np.sin(th) * np.sin(ph), np.cos(th)], axis=-1)
                hit = meets(q, d, hl, lx0, lx1, lz0, lz1)                              # (nothing else lies below the emitter's plane)
                g = (hit * np.clip(d[..., 1], 0.0, None)).mean()
                cut.append(hit.mean())
                probe += p_rr * p_nf * g * shininess * KE * kd_plate * (2.0 * PI / 8.0)
            probe /= nj * nj
            want = direct + probe
            assert (probe > 5.0 * direct).all() and 0.1 < float(np.mean(cut)) < 0.9   # the probe is what the pixel is made of, and the emitter cuts the lobe
            err = np.abs(mean[py, px] - want)
            assert (err <= 4.0 * se[py, px] + 3e-3 * want).all(), (px, py, mean[py, px], want, se[py, px], direct, probe)

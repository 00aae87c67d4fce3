// cudaraytracing_amd/csrc/crt_trace.h -- BVH traversal and samplers (device).
//
// Reference semantics (DeviceBVH.cuh:128-170, :31-43, DeviceTriangle.cuh:39-65):
// every node whose box passes hit_AABB is visited (no pruning; the root box is
// never tested), right child first; inside a leaf the first triangle among
// equal t wins (strict <), and an earlier-visited leaf wins equal t.  Leaves are
// therefore visited in DESCENDING order of their first-triangle index, so the
// winner among equal-t candidates is "largest leaf start, then smallest
// triangle index".  Encoding that rule explicitly makes the result independent
// of visit order, which is what lets the FAST mode (near-first order, pruning of
// boxes that start beyond the current best by a conservative margin, any-hit
// exit for shadow rays) return exactly what the exhaustive REFERENCE mode does.
#ifndef CRT_TRACE_H
#define CRT_TRACE_H

#include "crt_device.h"

#include <cfloat>

namespace crtdev {

struct TravCounters {
    uint32_t inner, leaf, tests, hits;
};

struct Hit {
    float t;
    int32_t tri;      // BVH-order triangle index, -1 = miss
    int32_t leaf_it;  // first triangle of the leaf that produced it
};

// reference: DeviceBVH.cuh:87-126.  nx/ny/nz = dir component < 0 (the swap).
__device__ __forceinline__ bool slab_test(float4 lo, float4 hi, const RayT& r, bool nx, bool ny, bool nz, float& t_enter)
{
    float tx0 = ((nx ? hi.x : lo.x) - r.o.x) * r.inv.x;
    float tx1 = ((nx ? lo.x : hi.x) - r.o.x) * r.inv.x;
    float ty0 = ((ny ? hi.y : lo.y) - r.o.y) * r.inv.y;
    float ty1 = ((ny ? lo.y : hi.y) - r.o.y) * r.inv.y;
    float tz0 = ((nz ? hi.z : lo.z) - r.o.z) * r.inv.z;
    float tz1 = ((nz ? lo.z : hi.z) - r.o.z) * r.inv.z;
    t_enter = maxf_ref(maxf_ref(tx0, ty0), tz0);
    float t_exit = minf_ref(minf_ref(tx1, ty1), tz1);
    return t_enter <= t_exit + CRT_EPSILON && t_exit >= 0;
}

// Pruning bound: a box may be skipped only if its slab entry distance lies beyond this.  Two slacks on top of the distance t of
// the best hit (or of the light, for visibility rays):
//   * 0.1 % + 1e-3: the rounding difference between the slab arithmetic and the Moeller-Trumbore t of a well-conditioned pair;
//   * CRT_PRUNE_REL x reach x steep, reach = max |origin coordinate| + |t| (no coordinate of a point of the ray up to t is larger),
//     steep = max |1 / direction component|: the hit Moeller-Trumbore accepts can lie OUTSIDE the triangle's box by a displacement
//     that scales with the coordinates in play, and along an axis the ray barely moves in that displacement is a distance
//     difference of displacement / |d_axis|.  Found by full-size FAST-against-REFERENCE frames: one ray in 3.5e9 of C3 (d.y =
//     -7.6e-4, grazing the shared edge of two triangles of a light sphere; the nearer triangle's hit lay 1.7e-5 above its box, i.e.
//     the box is entered at t + 0.022) and four more in 1.2e11 rays of C5 with a per-axis form of the term.
// NOT a theorem, and no factor makes it one: for a ray lying in a triangle's plane (det -> 0) t = (s2 . e2) / det is a ratio of
// rounding noise, and the reference accepts it when the equally noisy barycentrics land in (0,1).  With the shipped factor the soak
// (tools/soak_fast_vs_reference.py) finds 2 lost visibility rays in 3.66e11 rays of veach-mis (cos = 5e-6 / 3e-6 against a small
// sphere triangle; t lands 0.045 / 0.035 in front of the leaf box, 3.5 / 7.7 x the slack) and 0 in 3.4e11 of cornell-box; the
// margin histogram of tools/margin_hist.py shows a tail ~ s^-0.7 on veach-mis (factor 1e-3: a fifth of the events for +8.5 % frame
// time) and no tail at all on cornell-box.  DESIGN.md section 4; docs/experiments.md 4.3 has the numbers; include/crt.h states the contract; the lost
// rays are known answers in tests/test_adversarial_traversal.py.  Rays with a zero direction component get an infinite bound.
#ifndef CRT_PRUNE_REL
#define CRT_PRUNE_REL 1.0e-4f
#endif
__device__ __forceinline__ float prune_bound(float t, const F3 o, const F3 inv)
{
    const float reach = fmaxf(fmaxf(absf(o.x), absf(o.y)), absf(o.z)) + absf(t);
    const float steep = fmaxf(fmaxf(absf(inv.x), absf(inv.y)), absf(inv.z));
    return t + (absf(t) * 1.0e-3f + 1.0e-3f) + CRT_PRUNE_REL * reach * steep;
}

// Moeller-Trumbore exactly as DeviceTriangle.cuh:39-56 + inside() :58-65 + the t > EPSILON
// filter of DeviceBVHNode::hit (DeviceBVH.cuh:37).  Returns true for an accepted hit.
__device__ __forceinline__ bool tri_test(const DevScene& sc, int i, const RayT& r, float& t_out)
{
    const float4* g = sc.tri_geo + (size_t)i * 3;
    float4 A = g[0], B = g[1], C = g[2];
    F3 v1 = f3(A.x, A.y, A.z), e1 = f3(A.w, B.x, B.y), e2 = f3(B.z, B.w, C.x);
    F3 s = sub3(r.o, v1);
    F3 s1 = cross3(r.d, e2);
    F3 s2 = cross3(s, e1);
    float reciprocal = 1 / dot3(s1, e1);
    float beta = dot3(s1, s) * reciprocal;
    float gamma = dot3(s2, r.d) * reciprocal;
    float t = dot3(s2, e2) * reciprocal;
    float alpha = 1 - beta - gamma;
    bool inside = 0 < alpha && alpha < 1 && 0 < beta && beta < 1 && 0 < gamma && gamma < 1;
    t_out = t;
    return inside && t > CRT_EPSILON;
}

// ------------------------------------------------------------- samplers ----
// reference: include/Global.h:35-50
__device__ __forceinline__ F3 to_world(F3 a, F3 N)
{
    F3 C;
    if (absf(N.x) > absf(N.y)) {
        float invLen = 1.0f / sqrt_f(N.x * N.x + N.z * N.z);
        C = f3(N.z * invLen, 0.0f, -N.x * invLen);
    } else {
        float invLen = 1.0f / sqrt_f(N.y * N.y + N.z * N.z);
        C = f3(0.0f, N.z * invLen, -N.y * invLen);
    }
    F3 B = cross3(C, N);
    return add3(add3(scalel3(a.x, B), scalel3(a.y, C)), scalel3(a.z, N));
}
// reference: include/Global.h:57-66
__device__ __forceinline__ F3 sample_hemisphere(F3 N, float x_1, float x_2)
{
    float z = absf(1.0f - 2.0f * x_1);
    float r = sqrt_f(1.0f - z * z);
    float phi = (float)(2 * 3.14159265358979323846 * (double)x_2);
    float sn, cs;
    det_sincosf(phi, &sn, &cs);
    return to_world(f3(r * cs, r * sn, z), N);
}
// reference: include/Global.h:68-94
__device__ __forceinline__ F3 sample_lobe(F3 out, float delta_theta, float delta_phi, float u1, float u2)
{
    float eta_1 = 2 * u1 - 1;
    float eta_2 = 2 * u2 - 1;
    float r = norm3(out);
    float theta_0 = det_acosf(out.z / r);
    float phi_0;
    if ((double)absf(out.x) < 1e-5)
        phi_0 = out.y > 0.0f ? (float)1.57079632679489661923 : -(float)1.57079632679489661923;
    else
        phi_0 = det_atan2f(out.y, out.x);
    float theta = theta_0 + eta_1 * delta_theta;
    float phi = phi_0 + eta_2 * delta_phi;
    float st, ct, sp, cp;
    det_sincosf(theta, &st, &ct);
    det_sincosf(phi, &sp, &cp);
    return f3(st * cp, st * sp, ct);
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

} // namespace crtdev
#endif

// cudaraytracing_amd/csrc/crt_kernels.hip -- device layer of libcrt.so.
//
// Hand-written HIP for gfx950 (MI355X).  Replaces the reference's
// view_render_kernel / cast_ray_v2 / DeviceBVH::intersect
// (include/Render.cuh:199-354, include/DeviceBVH.cuh:87-170) with:
//
//   k_paths       one work item = one (pixel, sample) path.  Work items are
//                 ordered tile-major so the 64 lanes of a wave start on one 8x8
//                 pixel tile.  Every random draw is addressed explicitly
//                 (Philox counter = sample/depth/purpose/index, crt_detmath.h),
//                 which lets next-event estimation run in the forward pass while
//                 the radiance recursion is still evaluated deepest-vertex-first
//                 in the reference's float order (Render.cuh:238-326).  The
//                 reference's 7.7 KB/pixel global bounce stack becomes a 20 B
//                 per-vertex record; its 1 KB/pixel BVH stack lives in LDS.
//   k_accumulate  per pixel, sums L_k / spp in sample order (Render.cuh:348),
//                 tone-maps (Render.cuh:350) and writes RGB8 + float mean.
//
// Results are bit-identical to the CPU oracle (oracle/crt_oracle.cpp).
// Build: -ffp-contract=off, correctly rounded fp32 divide/sqrt (see build.py).
#include "../../include/crt.h"
#include "crt_device.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cfloat>
#include <cstring>
#include <string>
#include <vector>

using namespace crtdev;

extern "C" void crt_set_last_error_(const char* msg);

namespace {

enum { C_RAYS = 0, C_SHADOW, C_PROBE, C_INNER, C_LEAF, C_TESTS, C_HITS, C_PATHS, C_COUNT };

struct KParams {
    DevScene sc;
    float eye[3];
    float inv_view[9];
    float scale, ar;
    uint32_t width, height, spp;
    float p_rr;
    int32_t lsn;
    uint64_t seed;
    uint32_t rank, world, tiles_x, n_tiles;
    uint32_t nslots;        // pixel slots of this shard (local tiles * 64)
    uint32_t sample_begin;  // first sample index of this chunk
    uint64_t n_items;       // nslots * samples in this chunk
    float* L;               // 3 planes of `plane` floats
    uint64_t plane;
    unsigned long long* counters;
    int32_t stack_cap;
};

struct Counters {
    uint32_t rays, shadow, probe, inner, leaf, tests, hits;
};

// slot -> pixel.  Returns false for padding slots (ragged image edge / tile beyond the image).
__device__ __forceinline__ bool slot_to_pixel(const KParams& P, uint32_t slot, uint32_t& i, uint32_t& j)
{
    uint32_t tile = (slot >> 6) * P.world + P.rank;
    uint32_t pix = slot & 63u;
    if (tile >= P.n_tiles) return false;
    uint32_t ty = tile / P.tiles_x, tx = tile - ty * P.tiles_x;
    i = tx * CRT_TILE + (pix & 7u);
    j = ty * CRT_TILE + (pix >> 3);
    return i < P.width && j < P.height;
}

// ---------------------------------------------------------------------------
// BVH traversal.
//
// Reference semantics (DeviceBVH.cuh:128-170, :31-43, DeviceTriangle.cuh:39-65):
// every node whose box passes hit_AABB is visited (no pruning, the root box is
// never tested), right child first; a leaf keeps its first triangle among equal
// t (strict <), and an earlier-visited leaf wins equal t.  Leaves are therefore
// visited in DESCENDING order of their first-triangle index, so the winner among
// equal-t candidates is "largest leaf start, then smallest triangle index".
// Encoding that rule explicitly makes the result independent of visit order,
// which is what allows the FAST mode (near-first order, pruning of boxes that
// start beyond the current best by a conservative margin, any-hit exit for
// shadow rays) to return exactly what the exhaustive REFERENCE mode returns.
// ---------------------------------------------------------------------------
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

// conservative pruning bound: a box may be skipped only if it starts beyond this
__device__ __forceinline__ float prune_bound(float t) { return t + (absf(t) * 1.0e-3f + 1.0e-3f); }

// MODE: 0 = FAST, 1 = REFERENCE.  ANY: shadow query "exists valid hit with t_limit - t > EPSILON"
// (equivalent to the reference's closest-hit test in blocked(), Render.cuh:19-27, because float
// subtraction is monotone); returns hit.tri >= 0 iff blocked.
template <int MODE, bool ANY, bool STATS>
__device__ __forceinline__ Hit trace(const DevScene& sc, const RayT& r, float t_limit, int* stack, float* tstack, int lane_stride,
                                     Counters& cnt)
{
    Hit best;
    best.t = FLT_MAX; best.tri = -1; best.leaf_it = -1;
    const bool nx = r.d.x < 0, ny = r.d.y < 0, nz = r.d.z < 0;
    int sp = 0;
    int ref = sc.root_ref;
    float ref_t = -FLT_MAX;
    bool have = true;
    // shadow-ray pruning bound is fixed; closest-hit bound shrinks with best.t
    float bound = ANY ? prune_bound(t_limit) : FLT_MAX;
    if (ANY && MODE == 0) {
        // NaN or -inf limit can never be "blocked"; +inf is blocked by any hit (bound = inf)
        if (!(t_limit == t_limit) || t_limit == -pinf()) return best;
    }
    while (true) {
        if (!have) {
            if (sp == 0) break;
            sp--;
            ref = stack[sp * lane_stride];
            if (MODE == 0) {
                ref_t = tstack[sp * lane_stride];
                if (ref_t > bound) continue; // pruned after a closer hit was found
            }
        }
        have = false;
        if (ref >= 0) {
            if (STATS) cnt.inner++;
            const float4* n = sc.nodes + (size_t)ref * 4;
            float4 a = n[0], b = n[1], c = n[2], d = n[3];
            float tl, tr;
            bool hl = slab_test(a, b, r, nx, ny, nz, tl);
            bool hr = slab_test(c, d, r, nx, ny, nz, tr);
            int lref = __float_as_int(a.w), rref = __float_as_int(b.w);
            if (MODE == 1) {
                // push lc then rc: rc is visited first (DeviceBVH.cuh:154-166)
                if (hl && hr) { stack[sp * lane_stride] = lref; sp++; ref = rref; have = true; }
                else if (hl) { ref = lref; have = true; }
                else if (hr) { ref = rref; have = true; }
            } else {
                hl = hl && !(tl > bound);
                hr = hr && !(tr > bound);
                if (hl && hr) {
                    bool left_first = tl <= tr;
                    int far_ref = left_first ? rref : lref;
                    float far_t = left_first ? tr : tl;
                    stack[sp * lane_stride] = far_ref;
                    tstack[sp * lane_stride] = far_t;
                    sp++;
                    ref = left_first ? lref : rref;
                    have = true;
                } else if (hl) { ref = lref; have = true; }
                else if (hr) { ref = rref; have = true; }
            }
        } else {
            if (STATS) cnt.leaf++;
            uint32_t code = (uint32_t)~ref;
            int it = (int)(code >> 4);
            int n = (int)(code & 15u);
            if (n == 0) n = sc.leaf_count[it];
            for (int i = it; i < it + n; i++) {
                if (STATS) cnt.tests++;
                const float4* g = sc.tri_geo + (size_t)i * 3;
                float4 A = g[0], B = g[1], C = g[2];
                F3 v1 = f3(A.x, A.y, A.z), e1 = f3(A.w, B.x, B.y), e2 = f3(B.z, B.w, C.x);
                // Moeller-Trumbore exactly as DeviceTriangle.cuh:39-56
                F3 s = sub3(r.o, v1);
                F3 s1 = cross3(r.d, e2);
                F3 s2 = cross3(s, e1);
                float reciprocal = 1 / dot3(s1, e1);
                float beta = dot3(s1, s) * reciprocal;
                float gamma = dot3(s2, r.d) * reciprocal;
                float t = dot3(s2, e2) * reciprocal;
                float alpha = 1 - beta - gamma;
                bool inside = 0 < alpha && alpha < 1 && 0 < beta && beta < 1 && 0 < gamma && gamma < 1; // :58-65
                if (inside && t > CRT_EPSILON) { // DeviceBVH.cuh:37
                    if (ANY) {
                        if (t_limit - t > CRT_EPSILON) { best.t = t; best.tri = i; best.leaf_it = it; return best; }
                    } else if (t < best.t || (t == best.t && it > best.leaf_it)) {
                        best.t = t; best.tri = i; best.leaf_it = it;
                        if (MODE == 0) bound = prune_bound(t);
                    }
                }
            }
        }
    }
    if (STATS && best.tri >= 0) cnt.hits++;
    return best;
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

// ---------------------------------------------------------------- paths ----
struct Vertex {          // what the next step needs to know about the previous vertex
    F3 pos, n, from_dir, Ldir;
    int32_t mat;
};

template <int MODE, bool STATS>
__global__ __launch_bounds__(256) void k_paths(const KParams P)
{
    extern __shared__ int s_lds[];
    const int tid = threadIdx.x;
    int* stack = s_lds + tid;
    float* tstack = reinterpret_cast<float*>(s_lds + 256 * P.stack_cap) + tid;
    const DevScene& sc = P.sc;

    Counters cnt;
    cnt.rays = cnt.shadow = cnt.probe = cnt.inner = cnt.leaf = cnt.tests = cnt.hits = 0;

    const uint64_t item = (uint64_t)blockIdx.x * 256u + (uint64_t)tid;
    uint32_t pi = 0, pj = 0;
    bool live = item < P.n_items;
    uint32_t slot = 0, s_in_chunk = 0;
    if (live) {
        s_in_chunk = (uint32_t)(item / P.nslots);
        slot = (uint32_t)(item - (uint64_t)s_in_chunk * P.nslots);
        live = slot_to_pixel(P, slot, pi, pj);
    }
    F3 L = f3(0.0f, 0.0f, 0.0f);
    if (live) {
        const uint32_t k = P.sample_begin + s_in_chunk;
        const uint32_t pixel_index = pj * P.width + pi; // Render.cuh:336
        const float inv_pdf_sphere = (float)(2.0f * 3.14159265358979323846); // Global.h:96-99
        // ---- camera ray: Render.cuh:344-347 ----
        U4 rj = rng_draw(P.seed, pixel_index, k, 0, RNG_JITTER, 0);
        float x = (2 * ((int)pi + rng_uniform(rj.x)) / P.width - 1) * P.scale * P.ar;
        float y = (1 - 2 * ((int)pj + rng_uniform(rj.y)) / P.height) * P.scale;
        F3 cd = unit3(f3(-x, y, 1));
        F3 wd = f3(P.inv_view[0] * cd.x + (P.inv_view[3] * cd.y + P.inv_view[6] * cd.z),
                   P.inv_view[1] * cd.x + (P.inv_view[4] * cd.y + P.inv_view[7] * cd.z),
                   P.inv_view[2] * cd.x + (P.inv_view[5] * cd.y + P.inv_view[8] * cd.z));
        RayT ray = make_ray(f3(P.eye[0], P.eye[1], P.eye[2]), wd);

        // per-vertex records for the backward recursion
        float h_r[CRT_BOUNCE_STACK_SIZE], h_g[CRT_BOUNCE_STACK_SIZE], h_b[CRT_BOUNCE_STACK_SIZE], h_cos[CRT_BOUNCE_STACK_SIZE];
        int32_t h_mat[CRT_BOUNCE_STACK_SIZE];
        int deepest = -1;           // deepest vertex that "happend"
        bool deepest_emissive = false;
        F3 deepest_ke = f3(0.0f, 0.0f, 0.0f);
        Vertex prev;
        prev.mat = 0;
        prev.pos = prev.n = prev.from_dir = prev.Ldir = f3(0.0f, 0.0f, 0.0f);

        // ---- forward walk: Render.cuh:205-230, with the per-vertex work of the backward loop
        //      (NEE :262-286, probe :294-314) hoisted to where its inputs become known ----
        for (int depth = 0;; depth++) {
            cnt.rays++;
            Hit hit = trace<MODE, false, STATS>(sc, ray, 0.0f, stack, tstack, 256, cnt);
            if (hit.tri < 0) {
                // vertex `depth` did not happen: the previous vertex is the final one (direct light only)
                if (depth > 0) {
                    h_r[depth - 1] = prev.Ldir.x; h_g[depth - 1] = prev.Ldir.y; h_b[depth - 1] = prev.Ldir.z;
                    h_cos[depth - 1] = 0.0f; h_mat[depth - 1] = prev.mat;
                    deepest = depth - 1;
                }
                break;
            }
            F3 pos = add3(ray.o, scalel3(hit.t, ray.d)); // DeviceTriangle.cuh:50
            float4 gC = sc.tri_geo[(size_t)hit.tri * 3 + 2];
            F3 nrm = f3(gC.y, gC.z, gC.w);
            int32_t mat = sc.tri_mat[hit.tri];
            float4 m0 = sc.mats[mat * 3 + 0], m1 = sc.mats[mat * 3 + 1];
            uint32_t mflags = (uint32_t)__float_as_int(m1.w);

            if (depth > 0) {
                // previous vertex is not the deepest: it gets the indirect term and (SPECULAR) the probe
                float cos_prev = dot3(unit3(sub3(pos, prev.pos)), prev.n); // Render.cuh:291
                cos_prev = cos_prev > 0.0f ? cos_prev : 0.0f;
                float4 pm0 = sc.mats[prev.mat * 3 + 0], pm1 = sc.mats[prev.mat * 3 + 1];
                if ((uint32_t)__float_as_int(pm1.w) & 2u) { // SPECULAR: Render.cuh:294-314
                    float ns = pm0.w;
                    float delta_coeff = (float)((double)(det_expf(25 / ns) - 1) / (2.71828182845904523536 - 1));
                    F3 in = unit3(prev.from_dir);
                    F3 out = sub3(in, scale3(prev.n, 2.f * dot3(in, prev.n)));
                    float d_theta = (float)((double)(delta_coeff * 30) * 3.14159265358979323846 / 180);
                    float d_phi = (float)((double)(delta_coeff * 120) * 3.14159265358979323846 / 180);
                    U4 rp = rng_draw(P.seed, pixel_index, k, (uint32_t)(depth - 1), RNG_PROBE, 0);
                    F3 refd = unit3(sample_lobe(out, d_theta, d_phi, rng_uniform(rp.x), rng_uniform(rp.y)));
                    RayT probe = make_ray(prev.pos, refd);
                    cnt.rays++; cnt.probe++;
                    Hit ph = trace<MODE, false, STATS>(sc, probe, 0.0f, stack, tstack, 256, cnt);
                    if (ph.tri >= 0) {
                        int32_t pmat = sc.tri_mat[ph.tri];
                        float4 q1 = sc.mats[pmat * 3 + 1], q2 = sc.mats[pmat * 3 + 2];
                        if ((uint32_t)__float_as_int(q1.w) & 1u) { // probe hit an emitter (:304)
                            float log_shininess = det_log10f(ns);
                            float shininess_coeff = (float)((double)log_shininess * 0.5 + 1);
                            float ip = inv_pdf_sphere / 8.f;
                            F3 hp = add3(probe.o, scalel3(ph.t, probe.d));
                            float ct = dot3(unit3(sub3(hp, prev.pos)), prev.n);
                            ct = ct > 0.0f ? ct : 0.0f;
                            // shininess * (ke (.) kd) * cos * inv_pdf  (:311, eager)
                            F3 kekd = mul3(f3(q2.x, q2.y, q2.z), f3(pm1.x, pm1.y, pm1.z));
                            F3 temp = scale3(scale3(scalel3(shininess_coeff, kekd), ct), ip);
                            prev.Ldir = add3(prev.Ldir, temp);
                        }
                    }
                }
                h_r[depth - 1] = prev.Ldir.x; h_g[depth - 1] = prev.Ldir.y; h_b[depth - 1] = prev.Ldir.z;
                h_cos[depth - 1] = cos_prev; h_mat[depth - 1] = prev.mat;
            }
            deepest = depth;
            if (mflags & 1u) { // hit an emitter: path ends (:210); contributes only as camera vertex (:249-255)
                deepest_emissive = true;
                float4 m2 = sc.mats[mat * 3 + 2];
                deepest_ke = f3(m2.x, m2.y, m2.z);
                break;
            }
            // ---- next-event estimation at this vertex: Render.cuh:258-286 ----
            F3 f_r = f3(m0.x, m0.y, m0.z);
            F3 Ldir = f3(0.0f, 0.0f, 0.0f);
            for (int li = 0; li < sc.n_lights; li++) {
                uint2 lg = sc.lights[li];
                for (int sj = 0; sj < P.lsn; sj++) {
                    U4 rl = rng_draw(P.seed, pixel_index, k, (uint32_t)depth, RNG_NEE, (uint32_t)(li * P.lsn + sj));
                    uint32_t ti = rl.x % lg.y; // DeviceLights.cuh:35
                    const float4* lt = sc.ltri + (size_t)(lg.x + ti) * 4;
                    float4 l0 = lt[0], l1 = lt[1], l2 = lt[2], l3 = lt[3];
                    float alpha = rng_uniform(rl.y);                 // DeviceTriangle.cuh:69-71
                    float beta = rng_uniform(rl.z) * (1 - alpha);
                    float gamma = 1 - alpha - beta;
                    F3 lv1 = f3(l0.x, l0.y, l0.z), lv2 = f3(l0.w, l1.x, l1.y), lv3 = f3(l1.z, l1.w, l2.x);
                    F3 lpos = add3(add3(scalel3(alpha, lv1), scalel3(beta, lv2)), scalel3(gamma, lv3));
                    F3 dist = sub3(lpos, pos);
                    F3 dir = unit3(dist);
                    RayT back = make_ray(pos, dir);
                    float t_to_light = dist.x / dir.x; // Render.cuh:272
                    cnt.rays++; cnt.shadow++;
                    bool blocked;
                    if (MODE == 1) {
                        Hit sh = trace<1, false, STATS>(sc, back, 0.0f, stack, tstack, 256, cnt);
                        blocked = t_to_light - sh.t > CRT_EPSILON; // Render.cuh:22
                    } else {
                        Hit sh = trace<0, true, false>(sc, back, t_to_light, stack, tstack, 256, cnt);
                        blocked = sh.tri >= 0;
                    }
                    if (!blocked) {
                        float tl = norm3(dist);
                        float t2 = tl * tl;
                        float cos_theta = dot3(dir, nrm);
                        float cos_theta_2 = -dot3(dir, f3(l2.y, l2.z, l2.w));
                        cos_theta = cos_theta > 0.0f ? cos_theta : 0.0f;
                        cos_theta_2 = cos_theta_2 > 0.0f ? cos_theta_2 : 0.0f;
                        // ((((Le*fr)*cos)*cos2)*inv_pdf)/t2)/lsn  (:283)
                        F3 c = mul3(f3(l3.x, l3.y, l3.z), f_r);
                        c = scale3(c, cos_theta);
                        c = scale3(c, cos_theta_2);
                        c = scale3(c, l3.w);
                        c = div3(c, t2);
                        c = div3(c, (float)P.lsn);
                        Ldir = add3(Ldir, c);
                    }
                }
            }
            // ---- continue or stop: Render.cuh:210-228 ----
            bool stop = depth == CRT_BOUNCE_STACK_SIZE - 1; // bounce stack full
            U4 rb;
            if (!stop) {
                rb = rng_draw(P.seed, pixel_index, k, (uint32_t)depth, RNG_BOUNCE, 0);
                stop = rng_uniform(rb.x) > P.p_rr;
            }
            if (stop) {
                h_r[depth] = Ldir.x; h_g[depth] = Ldir.y; h_b[depth] = Ldir.z; h_cos[depth] = 0.0f; h_mat[depth] = mat;
                break;
            }
            F3 ndir = unit3(sample_hemisphere(nrm, rng_uniform(rb.y), rng_uniform(rb.z)));
            prev.pos = pos; prev.n = nrm; prev.from_dir = ray.d; prev.Ldir = Ldir; prev.mat = mat;
            ray = make_ray(pos, ndir);
        }

        // ---- backward recursion, deepest vertex first: Render.cuh:238-326 ----
        if (deepest >= 0) {
            int v = deepest;
            if (deepest_emissive) {
                L = v == 0 ? add3(f3(0.0f, 0.0f, 0.0f), deepest_ke) : f3(0.0f, 0.0f, 0.0f); // :249-255, :323
            } else {
                L = add3(f3(0.0f, 0.0f, 0.0f), f3(h_r[v], h_g[v], h_b[v])); // final hit: direct light only (:316-319)
            }
            for (v = deepest - 1; v >= 0; v--) {
                float4 fm = sc.mats[h_mat[v] * 3 + 0];
                F3 ind = mul3(L, f3(fm.x, fm.y, fm.z)); // L (.) f_r * cos * inv_pdf / P_RR  (:293)
                ind = scale3(ind, h_cos[v]);
                ind = scale3(ind, inv_pdf_sphere);
                ind = div3(ind, P.p_rr);
                L = add3(ind, f3(h_r[v], h_g[v], h_b[v])); // :323
            }
        }
        P.L[item] = L.x;
        P.L[P.plane + item] = L.y;
        P.L[2 * P.plane + item] = L.z;
    }

    // ---- counters: one atomic per wave ----
    uint32_t r = wave_sum(cnt.rays), s = wave_sum(cnt.shadow), p = wave_sum(cnt.probe), lv = wave_sum(live ? 1u : 0u);
    if ((tid & 63) == 0) {
        atomicAdd(&P.counters[C_RAYS], (unsigned long long)r);
        atomicAdd(&P.counters[C_SHADOW], (unsigned long long)s);
        atomicAdd(&P.counters[C_PROBE], (unsigned long long)p);
        atomicAdd(&P.counters[C_PATHS], (unsigned long long)lv);
    }
    if (STATS) {
        uint32_t a = wave_sum(cnt.inner), b = wave_sum(cnt.leaf), c = wave_sum(cnt.tests), d = wave_sum(cnt.hits);
        if ((tid & 63) == 0) {
            atomicAdd(&P.counters[C_INNER], (unsigned long long)a);
            atomicAdd(&P.counters[C_LEAF], (unsigned long long)b);
            atomicAdd(&P.counters[C_TESTS], (unsigned long long)c);
            atomicAdd(&P.counters[C_HITS], (unsigned long long)d);
        }
    }
}

// ----------------------------------------------------------- accumulate ----
struct AParams {
    uint32_t width, height, spp;
    uint32_t rank, world, tiles_x, n_tiles;
    uint32_t nslots;
    uint32_t chunk_samples;
    uint32_t first_chunk, last_chunk, tiled_output;
    const float* L;
    uint64_t plane;
    float* accum;      // 3 planes of nslots (running sum across chunks)
    uint8_t* out_rgb;
    float* out_mean;   // may be null
};

__device__ __forceinline__ uint8_t to_u8(float v)
{
    if (!(v == v)) return 0;
    if (v <= 0.0f) return 0;
    if (v >= 255.0f) return 255;
    return (uint8_t)v; // truncation (Render.cuh:350)
}
// reference: Global.h:121-124 then Render.cuh:350
__device__ __forceinline__ uint8_t tonemap(float c)
{
    float cl = maxf_ref(0.0f, minf_ref(1.0f, c));
    return to_u8(255 * det_powf(cl, 0.6f));
}

__global__ __launch_bounds__(256) void k_accumulate(const AParams A)
{
    uint32_t slot = blockIdx.x * 256u + threadIdx.x;
    if (slot >= A.nslots) return;
    uint32_t tile = (slot >> 6) * A.world + A.rank, pix = slot & 63u;
    uint32_t i = 0, j = 0;
    bool valid = tile < A.n_tiles;
    if (valid) {
        uint32_t ty = tile / A.tiles_x, tx = tile - ty * A.tiles_x;
        i = tx * CRT_TILE + (pix & 7u);
        j = ty * CRT_TILE + (pix >> 3);
        valid = i < A.width && j < A.height;
    }
    F3 c = f3(0.0f, 0.0f, 0.0f);
    if (valid) {
        if (!A.first_chunk) c = f3(A.accum[slot], A.accum[A.nslots + slot], A.accum[2ull * A.nslots + slot]);
        const float fspp = (float)A.spp;
        for (uint32_t s = 0; s < A.chunk_samples; s++) { // temp_color += L / spp, in sample order (Render.cuh:348)
            uint64_t it = (uint64_t)s * A.nslots + slot;
            c.x = c.x + A.L[it] / fspp;
            c.y = c.y + A.L[A.plane + it] / fspp;
            c.z = c.z + A.L[2 * A.plane + it] / fspp;
        }
        if (!A.last_chunk) {
            A.accum[slot] = c.x; A.accum[A.nslots + slot] = c.y; A.accum[2ull * A.nslots + slot] = c.z;
            return;
        }
    } else if (!A.tiled_output || !A.last_chunk) {
        return;
    }
    uint64_t o = A.tiled_output ? (uint64_t)slot : (uint64_t)j * A.width + i;
    A.out_rgb[o * 3 + 0] = valid ? tonemap(c.x) : 0;
    A.out_rgb[o * 3 + 1] = valid ? tonemap(c.y) : 0;
    A.out_rgb[o * 3 + 2] = valid ? tonemap(c.z) : 0;
    if (A.out_mean) { A.out_mean[o * 3 + 0] = c.x; A.out_mean[o * 3 + 1] = c.y; A.out_mean[o * 3 + 2] = c.z; }
}

// ------------------------------------------------------------ test kernels --
template <int MODE>
__global__ __launch_bounds__(256) void k_intersect(DevScene sc, uint32_t n, const float* o, const float* d, int32_t* out_tri,
                                                   float* out_t, int stack_cap)
{
    extern __shared__ int s_lds[];
    int* stack = s_lds + threadIdx.x;
    float* tstack = reinterpret_cast<float*>(s_lds + 256 * stack_cap) + threadIdx.x;
    uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    RayT r = make_ray(f3(o[3 * i], o[3 * i + 1], o[3 * i + 2]), f3(d[3 * i], d[3 * i + 1], d[3 * i + 2]));
    Counters cnt;
    Hit h = trace<MODE, false, false>(sc, r, 0.0f, stack, tstack, 256, cnt);
    out_tri[i] = h.tri;
    out_t[i] = h.t;
}

__global__ void k_math(int fn, uint32_t n, const float* a, const float* b, float* out)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float x = a[i], y = b ? b[i] : 0.0f, r;
    switch (fn) {
    case 0: r = det_sinf(x); break;
    case 1: r = det_cosf(x); break;
    case 2: r = det_tanf(x); break;
    case 3: r = det_acosf(x); break;
    case 4: r = det_atan2f(x, y); break;
    case 5: r = det_expf(x); break;
    case 6: r = det_log10f(x); break;
    case 7: r = det_powf(x, y); break;
    case 8: r = rng_uniform(__float_as_uint(x)); break;
    case 9: { float s, c; det_sincosf(x, &s, &c); r = s; break; }
    case 10: { float s, c; det_sincosf(x, &s, &c); r = c; break; }
    default: r = qnan();
    }
    out[i] = r;
}
__global__ void k_philox(uint32_t n, const uint32_t* ctr, const uint32_t* key, uint32_t* out)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    U4 c;
    c.x = ctr[4 * i]; c.y = ctr[4 * i + 1]; c.z = ctr[4 * i + 2]; c.w = ctr[4 * i + 3];
    U4 r = philox4x32_10(c, key[2 * i], key[2 * i + 1]);
    out[4 * i] = r.x; out[4 * i + 1] = r.y; out[4 * i + 2] = r.z; out[4 * i + 3] = r.w;
}

// ------------------------------------------------------------------ host ----
struct HipFail {
    hipError_t e;
    const char* what;
};
#define HIP_CHECK(call)                                          \
    do {                                                         \
        hipError_t e_ = (call);                                  \
        if (e_ != hipSuccess) throw HipFail{e_, #call};          \
    } while (0)

template <typename T> struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    void alloc(size_t count)
    {
        release();
        if (count == 0) count = 1;
        HIP_CHECK(hipMalloc((void**)&p, count * sizeof(T)));
        n = count;
    }
    void upload(const std::vector<T>& v)
    {
        alloc(v.size());
        if (!v.empty()) HIP_CHECK(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    }
    void release()
    {
        if (p) { (void)hipFree(p); p = nullptr; n = 0; }
    }
    ~DevBuf() { release(); }
};

} // namespace

struct crt_scene {
    int device = 0;
    DevBuf<float4> nodes, tri_geo, mats, ltri;
    DevBuf<int32_t> tri_mat, leaf_count;
    DevBuf<uint2> lights;
    DevBuf<float> L, accum;
    DevBuf<unsigned long long> counters;
    DevScene dev{};
    int stack_cap = 0;
    uint32_t n_tris = 0;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
};

namespace {

int fail(int status, const std::string& msg)
{
    crt_set_last_error_(msg.c_str());
    return status;
}
int fail_hip(const HipFail& f)
{
    return fail(CRT_ERR_HIP, std::string(f.what) + ": " + hipGetErrorString(f.e));
}

float as_float(int32_t v) { float f; std::memcpy(&f, &v, 4); return f; }

// Converts the reference-layout BVH (post-order, boxes in the nodes themselves) into the
// device layout of crt_device.h.  Returns the tree depth (root = 1).
int convert_bvh(const crt_scene_desc& d, std::vector<float4>& nodes, std::vector<int32_t>& leaf_count, int32_t& root_ref)
{
    auto is_leaf = [&](int32_t i) { return d.nodes[i].lc < 0 && d.nodes[i].rc < 0; };
    auto leaf_ref = [&](int32_t i) -> int32_t {
        const crt_bvh_node& n = d.nodes[i];
        uint32_t cnt = n.n <= 15 ? n.n : 0;
        if (n.n > 15) leaf_count[n.it] = (int32_t)n.n;
        return (int32_t)~(((uint32_t)n.it << 4) | cnt);
    };
    leaf_count.assign(d.n_tris ? d.n_tris : 1, 0);
    nodes.clear();
    if (is_leaf(d.root)) { root_ref = leaf_ref(d.root); return 1; }
    // breadth-first numbering of inner nodes
    std::vector<int32_t> order, index(d.n_nodes, -1), depth_of;
    order.push_back(d.root);
    depth_of.push_back(1);
    index[d.root] = 0;
    int max_depth = 1;
    for (size_t q = 0; q < order.size(); q++) {
        const crt_bvh_node& n = d.nodes[order[q]];
        int32_t ch[2] = {n.lc, n.rc};
        for (int c = 0; c < 2; c++) {
            max_depth = std::max(max_depth, depth_of[q] + 1);
            if (!is_leaf(ch[c])) {
                index[ch[c]] = (int32_t)order.size();
                order.push_back(ch[c]);
                depth_of.push_back(depth_of[q] + 1);
            }
        }
    }
    nodes.resize(order.size() * 4);
    for (size_t q = 0; q < order.size(); q++) {
        const crt_bvh_node& n = d.nodes[order[q]];
        const crt_bvh_node& l = d.nodes[n.lc];
        const crt_bvh_node& r = d.nodes[n.rc];
        int32_t lref = is_leaf(n.lc) ? leaf_ref(n.lc) : index[n.lc];
        int32_t rref = is_leaf(n.rc) ? leaf_ref(n.rc) : index[n.rc];
        nodes[q * 4 + 0] = make_float4(l.aa[0], l.aa[1], l.aa[2], as_float(lref));
        nodes[q * 4 + 1] = make_float4(l.bb[0], l.bb[1], l.bb[2], as_float(rref));
        nodes[q * 4 + 2] = make_float4(r.aa[0], r.aa[1], r.aa[2], 0.0f);
        nodes[q * 4 + 3] = make_float4(r.bb[0], r.bb[1], r.bb[2], 0.0f);
    }
    root_ref = 0;
    return max_depth;
}

int validate_desc(const crt_scene_desc* d)
{
    if (!d || !d->nodes || !d->tris || !d->materials || d->n_nodes == 0 || d->n_tris == 0 || d->n_materials == 0)
        return fail(CRT_ERR_INVALID_ARG, "crt_scene_create: empty scene description");
    if (d->root < 0 || (uint32_t)d->root >= d->n_nodes) return fail(CRT_ERR_INVALID_ARG, "crt_scene_create: root index out of range");
    if (d->n_tris >= (1u << 27)) return fail(CRT_ERR_UNSUPPORTED, "crt_scene_create: more than 2^27 triangles");
    if (d->n_lights && (!d->lights || !d->light_tris)) return fail(CRT_ERR_INVALID_ARG, "crt_scene_create: lights without triangles");
    for (uint32_t i = 0; i < d->n_nodes; i++) {
        const crt_bvh_node& n = d->nodes[i];
        bool leaf = n.lc < 0 && n.rc < 0;
        if (leaf) {
            if (n.it < 0 || n.n == 0 || (uint64_t)n.it + n.n > d->n_tris) return fail(CRT_ERR_INVALID_ARG, "crt_scene_create: leaf range outside the triangle array");
        } else {
            // inner nodes of the reference builder always have two children (BVH.h:79-81) that precede them (post-order)
            if (n.lc < 0 || n.rc < 0 || (uint32_t)n.lc >= i || (uint32_t)n.rc >= i) return fail(CRT_ERR_INVALID_ARG, "crt_scene_create: inner node children must precede it (post-order)");
        }
    }
    for (uint32_t i = 0; i < d->n_tris; i++)
        if (d->tris[i].material < 0 || (uint32_t)d->tris[i].material >= d->n_materials) return fail(CRT_ERR_INVALID_ARG, "crt_scene_create: triangle material index out of range");
    for (uint32_t i = 0; i < d->n_light_tris; i++)
        if (d->light_tris[i].material < 0 || (uint32_t)d->light_tris[i].material >= d->n_materials) return fail(CRT_ERR_INVALID_ARG, "crt_scene_create: light triangle material index out of range");
    for (uint32_t i = 0; i < d->n_lights; i++)
        if (d->lights[i].count == 0 || (uint64_t)d->lights[i].first_tri + d->lights[i].count > d->n_light_tris) return fail(CRT_ERR_INVALID_ARG, "crt_scene_create: light range outside the light triangle array");
    return CRT_OK;
}

struct Shard {
    uint32_t tiles_x, tiles_y, n_tiles, local_tiles, nslots;
};
Shard make_shard(uint32_t w, uint32_t h, uint32_t rank, uint32_t world)
{
    Shard s;
    s.tiles_x = (w + CRT_TILE - 1) / CRT_TILE;
    s.tiles_y = (h + CRT_TILE - 1) / CRT_TILE;
    s.n_tiles = s.tiles_x * s.tiles_y;
    s.local_tiles = (s.n_tiles + world - 1) / world; // padded so every rank writes the same number of slots
    (void)rank;
    s.nslots = s.local_tiles * 64u;
    return s;
}

template <int MODE, bool STATS> void launch_paths(const KParams& P, size_t lds, hipStream_t st)
{
    uint64_t blocks = (P.n_items + 255) / 256;
    hipLaunchKernelGGL((k_paths<MODE, STATS>), dim3((unsigned)blocks), dim3(256), lds, st, P);
}

const uint64_t kMaxChunkItems = 1ull << 28; // 268 M paths per launch -> 3.2 GB of per-path radiance

int render_impl(crt_scene* sc, const crt_camera* cam, const crt_params* prm, void* d_rgb, void* d_mean, hipStream_t st, crt_stats* stats,
                bool sync_for_stats)
{
    if (!sc || !cam || !prm || !d_rgb) return fail(CRT_ERR_INVALID_ARG, "crt_render: null argument");
    if (prm->width == 0 || prm->height == 0 || prm->spp == 0) return fail(CRT_ERR_INVALID_ARG, "crt_render: width, height and spp must be positive");
    if (prm->world == 0 || prm->rank >= prm->world) return fail(CRT_ERR_INVALID_ARG, "crt_render: need rank < world");
    if (prm->light_sample_n < 0) return fail(CRT_ERR_INVALID_ARG, "crt_render: light_sample_n must be >= 0");
    if ((uint64_t)prm->width * prm->height > 0xffffffffull) return fail(CRT_ERR_UNSUPPORTED, "crt_render: more than 2^32 pixels");
    if (prm->traversal != CRT_TRAVERSAL_FAST && prm->traversal != CRT_TRAVERSAL_REFERENCE) return fail(CRT_ERR_INVALID_ARG, "crt_render: unknown traversal mode");
    const bool want_stats = (prm->flags & CRT_FLAG_STATS) != 0;
    if (want_stats && prm->traversal != CRT_TRAVERSAL_REFERENCE) return fail(CRT_ERR_INVALID_ARG, "crt_render: CRT_FLAG_STATS needs CRT_TRAVERSAL_REFERENCE");
    const bool tiled = (prm->flags & CRT_FLAG_TILED_OUTPUT) != 0;
    if (prm->world > 1 && !tiled) return fail(CRT_ERR_INVALID_ARG, "crt_render: world > 1 needs CRT_FLAG_TILED_OUTPUT");
    try {
        HIP_CHECK(hipSetDevice(sc->device));
        Shard sh = make_shard(prm->width, prm->height, prm->rank, prm->world);
        uint32_t chunk = (uint32_t)std::min<uint64_t>(prm->spp, std::max<uint64_t>(1, kMaxChunkItems / sh.nslots));
        uint64_t cap = (uint64_t)chunk * sh.nslots;
        if (sc->L.n < cap * 3) sc->L.alloc(cap * 3);
        if (sc->accum.n < (size_t)sh.nslots * 3) sc->accum.alloc((size_t)sh.nslots * 3);
        HIP_CHECK(hipMemsetAsync(sc->counters.p, 0, C_COUNT * sizeof(unsigned long long), st));

        KParams P;
        std::memset(&P, 0, sizeof(P));
        P.sc = sc->dev;
        std::memcpy(P.eye, cam->eye, sizeof(P.eye));
        std::memcpy(P.inv_view, cam->inv_view, sizeof(P.inv_view));
        P.scale = det_tanf(cam->fov_y / 2);                       // Render.cuh:338
        P.ar = (float)prm->width / (float)prm->height;            // Render.cuh:339
        P.width = prm->width; P.height = prm->height; P.spp = prm->spp;
        P.p_rr = prm->p_rr; P.lsn = prm->light_sample_n; P.seed = prm->seed;
        P.rank = prm->rank; P.world = prm->world; P.tiles_x = sh.tiles_x; P.n_tiles = sh.n_tiles;
        P.nslots = sh.nslots;
        P.L = sc->L.p; P.plane = cap;
        P.counters = sc->counters.p;
        P.stack_cap = sc->stack_cap;
        size_t lds = (size_t)sc->stack_cap * 256 * 4 * 2;

        AParams A;
        std::memset(&A, 0, sizeof(A));
        A.width = prm->width; A.height = prm->height; A.spp = prm->spp;
        A.rank = prm->rank; A.world = prm->world; A.tiles_x = sh.tiles_x; A.n_tiles = sh.n_tiles;
        A.nslots = sh.nslots; A.tiled_output = tiled ? 1 : 0;
        A.L = sc->L.p; A.plane = cap; A.accum = sc->accum.p;
        A.out_rgb = (uint8_t*)d_rgb; A.out_mean = (float*)d_mean;

        HIP_CHECK(hipEventRecord(sc->ev[0], st));
        float kernel_ms = 0.0f;
        uint32_t launches = 0;
        std::vector<std::pair<hipEvent_t, hipEvent_t>> kev;
        for (uint32_t s0 = 0; s0 < prm->spp; s0 += chunk) {
            uint32_t ns = std::min(chunk, prm->spp - s0);
            P.sample_begin = s0;
            P.n_items = (uint64_t)ns * sh.nslots;
            if (P.n_items / 256 + 1 > 0x7fffffffull) throw HipFail{hipErrorInvalidValue, "grid too large"};
            // the first chunk's path kernel is bracketed by events 1/2 (reported as kernel_ms);
            // further chunks are identical launches
            if (s0 == 0) HIP_CHECK(hipEventRecord(sc->ev[1], st));
            if (prm->traversal == CRT_TRAVERSAL_REFERENCE) {
                if (want_stats) launch_paths<1, true>(P, lds, st); else launch_paths<1, false>(P, lds, st);
            } else {
                launch_paths<0, false>(P, lds, st);
            }
            HIP_CHECK(hipGetLastError());
            if (s0 == 0) HIP_CHECK(hipEventRecord(sc->ev[2], st));
            launches++;
            A.chunk_samples = ns;
            A.first_chunk = s0 == 0; A.last_chunk = s0 + ns >= prm->spp;
            hipLaunchKernelGGL(k_accumulate, dim3((sh.nslots + 255) / 256), dim3(256), 0, st, A);
            HIP_CHECK(hipGetLastError());
        }
        HIP_CHECK(hipEventRecord(sc->ev[3], st));
        if (stats && sync_for_stats) {
            HIP_CHECK(hipStreamSynchronize(st));
            unsigned long long c[C_COUNT];
            HIP_CHECK(hipMemcpy(c, sc->counters.p, sizeof(c), hipMemcpyDeviceToHost));
            std::memset(stats, 0, sizeof(*stats));
            stats->paths = c[C_PATHS]; stats->rays = c[C_RAYS]; stats->shadow_rays = c[C_SHADOW]; stats->probe_rays = c[C_PROBE];
            stats->inner_pops = c[C_INNER]; stats->leaf_pops = c[C_LEAF]; stats->tri_tests = c[C_TESTS]; stats->hits = c[C_HITS];
            HIP_CHECK(hipEventElapsedTime(&kernel_ms, sc->ev[1], sc->ev[2]));
            float total = 0.0f;
            HIP_CHECK(hipEventElapsedTime(&total, sc->ev[0], sc->ev[3]));
            // with several identical chunks, scale the first chunk's kernel time by the work share
            stats->kernel_ms = launches > 1 ? kernel_ms * ((float)prm->spp / (float)chunk) : kernel_ms;
            stats->total_ms = total;
            stats->kernel_launches = launches;
        }
        return CRT_OK;
    } catch (const HipFail& f) {
        return fail_hip(f);
    }
}

} // namespace

extern "C" {

int crt_device_count(int* count)
{
    if (!count) return fail(CRT_ERR_INVALID_ARG, "crt_device_count: null argument");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *count = 0; return fail(CRT_ERR_NO_DEVICE, std::string("hipGetDeviceCount: ") + hipGetErrorString(e)); }
    *count = n;
    return CRT_OK;
}

int crt_shard_slots(uint32_t width, uint32_t height, uint32_t rank, uint32_t world, uint64_t* slots)
{
    if (!slots || width == 0 || height == 0 || world == 0 || rank >= world) return fail(CRT_ERR_INVALID_ARG, "crt_shard_slots: bad arguments");
    *slots = make_shard(width, height, rank, world).nslots;
    return CRT_OK;
}

int crt_scene_create(const crt_scene_desc* d, int device, crt_scene** out)
{
    if (!out) return fail(CRT_ERR_INVALID_ARG, "crt_scene_create: null output");
    *out = nullptr;
    int rc = validate_desc(d);
    if (rc != CRT_OK) return rc;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(CRT_ERR_NO_DEVICE, "crt_scene_create: no HIP device available");
    if (device < 0 || device >= n) return fail(CRT_ERR_INVALID_ARG, "crt_scene_create: device index out of range");
    crt_scene* sc = nullptr;
    try {
        sc = new crt_scene();
        sc->device = device;
        HIP_CHECK(hipSetDevice(device));
        std::vector<float4> nodes, geo(d->n_tris * 3ull), mats(d->n_materials * 3ull), ltri(d->n_light_tris * 4ull);
        std::vector<int32_t> leaf_count, tri_mat(d->n_tris);
        int32_t root_ref = 0;
        int depth = convert_bvh(*d, nodes, leaf_count, root_ref);
        for (uint32_t i = 0; i < d->n_tris; i++) {
            const crt_triangle& t = d->tris[i];
            // e1 = v2 - v1, e2 = v3 - v1 as DeviceTriangle's constructor computes them (DeviceTriangle.cuh:27-28)
            float e1[3] = {t.v2[0] - t.v1[0], t.v2[1] - t.v1[1], t.v2[2] - t.v1[2]};
            float e2[3] = {t.v3[0] - t.v1[0], t.v3[1] - t.v1[1], t.v3[2] - t.v1[2]};
            geo[i * 3ull + 0] = make_float4(t.v1[0], t.v1[1], t.v1[2], e1[0]);
            geo[i * 3ull + 1] = make_float4(e1[1], e1[2], e2[0], e2[1]);
            geo[i * 3ull + 2] = make_float4(e2[2], t.normal[0], t.normal[1], t.normal[2]);
            tri_mat[i] = t.material;
        }
        for (uint32_t i = 0; i < d->n_materials; i++) {
            const crt_material& m = d->materials[i];
            const float pi_f = (float)3.14159265358979323846;
            int32_t flags = (m.has_emit ? 1 : 0) | (m.mode == 1 ? 2 : 0);
            mats[i * 3ull + 0] = make_float4(m.kd[0] / pi_f, m.kd[1] / pi_f, m.kd[2] / pi_f, m.ns); // f_r = kd / float(M_PI) (Render.cuh:259)
            mats[i * 3ull + 1] = make_float4(m.kd[0], m.kd[1], m.kd[2], as_float(flags));
            mats[i * 3ull + 2] = make_float4(m.ke[0], m.ke[1], m.ke[2], 0.0f);
        }
        for (uint32_t i = 0; i < d->n_light_tris; i++) {
            const crt_triangle& t = d->light_tris[i];
            const crt_material& m = d->materials[t.material];
            ltri[i * 4ull + 0] = make_float4(t.v1[0], t.v1[1], t.v1[2], t.v2[0]);
            ltri[i * 4ull + 1] = make_float4(t.v2[1], t.v2[2], t.v3[0], t.v3[1]);
            ltri[i * 4ull + 2] = make_float4(t.v3[2], t.normal[0], t.normal[1], t.normal[2]);
            ltri[i * 4ull + 3] = make_float4(m.ke[0], m.ke[1], m.ke[2], t.area_of_obj);
        }
        std::vector<uint2> lights(d->n_lights);
        for (uint32_t i = 0; i < d->n_lights; i++) lights[i] = make_uint2(d->lights[i].first_tri, d->lights[i].count);
        sc->nodes.upload(nodes); sc->tri_geo.upload(geo); sc->tri_mat.upload(tri_mat); sc->mats.upload(mats);
        sc->ltri.upload(ltri); sc->lights.upload(lights); sc->leaf_count.upload(leaf_count);
        sc->counters.alloc(C_COUNT);
        for (int i = 0; i < 4; i++) HIP_CHECK(hipEventCreate(&sc->ev[i]));
        sc->dev.nodes = sc->nodes.p; sc->dev.tri_geo = sc->tri_geo.p; sc->dev.tri_mat = sc->tri_mat.p; sc->dev.mats = sc->mats.p;
        sc->dev.ltri = sc->ltri.p; sc->dev.lights = sc->lights.p; sc->dev.leaf_count = sc->leaf_count.p;
        sc->dev.root_ref = root_ref; sc->dev.n_lights = (int32_t)d->n_lights;
        sc->n_tris = d->n_tris;
        // Both traversal modes hold at most one pending sibling per tree level.
        sc->stack_cap = depth + 2;
        if ((size_t)sc->stack_cap * 256 * 8 > 64 * 1024) { delete sc; return fail(CRT_ERR_UNSUPPORTED, "crt_scene_create: BVH deeper than the LDS traversal stack allows"); }
        *out = sc;
        return CRT_OK;
    } catch (const HipFail& f) {
        delete sc;
        return fail_hip(f);
    } catch (const std::bad_alloc&) {
        delete sc;
        return fail(CRT_ERR_OOM, "crt_scene_create: out of host memory");
    }
}

int crt_scene_destroy(crt_scene* sc)
{
    if (!sc) return CRT_OK;
    (void)hipSetDevice(sc->device);
    for (int i = 0; i < 4; i++)
        if (sc->ev[i]) (void)hipEventDestroy(sc->ev[i]);
    delete sc;
    return CRT_OK;
}

int crt_render_device(crt_scene* sc, const crt_camera* cam, const crt_params* prm, void* d_rgb, void* d_mean, void* stream, crt_stats* stats)
{
    return render_impl(sc, cam, prm, d_rgb, d_mean, (hipStream_t)stream, stats, true);
}

int crt_render(crt_scene* sc, const crt_camera* cam, const crt_params* prm, uint8_t* out_rgb, float* out_mean, crt_stats* stats)
{
    if (!sc || !prm || !out_rgb) return fail(CRT_ERR_INVALID_ARG, "crt_render: null argument");
    if (prm->world == 0 || prm->rank >= prm->world || prm->width == 0 || prm->height == 0) return fail(CRT_ERR_INVALID_ARG, "crt_render: bad shard or size");
    try {
        HIP_CHECK(hipSetDevice(sc->device));
        const bool tiled = (prm->flags & CRT_FLAG_TILED_OUTPUT) != 0;
        uint64_t npix = tiled ? make_shard(prm->width, prm->height, prm->rank, prm->world).nslots : (uint64_t)prm->width * prm->height;
        DevBuf<uint8_t> d_rgb;
        DevBuf<float> d_mean;
        d_rgb.alloc(npix * 3);
        if (out_mean) d_mean.alloc(npix * 3);
        int rc = render_impl(sc, cam, prm, d_rgb.p, out_mean ? d_mean.p : nullptr, nullptr, stats, true);
        if (rc != CRT_OK) return rc;
        HIP_CHECK(hipDeviceSynchronize()); // Render.cuh:440
        HIP_CHECK(hipMemcpy(out_rgb, d_rgb.p, npix * 3, hipMemcpyDeviceToHost)); // Render.cuh:464
        if (out_mean) HIP_CHECK(hipMemcpy(out_mean, d_mean.p, npix * 3 * sizeof(float), hipMemcpyDeviceToHost));
        return CRT_OK;
    } catch (const HipFail& f) {
        return fail_hip(f);
    }
}

int crt_intersect(crt_scene* sc, uint32_t n, const float* origins, const float* dirs, uint32_t traversal, int32_t* out_tri, float* out_t)
{
    if (!sc || !origins || !dirs || !out_tri || !out_t) return fail(CRT_ERR_INVALID_ARG, "crt_intersect: null argument");
    if (n == 0) return CRT_OK;
    try {
        HIP_CHECK(hipSetDevice(sc->device));
        DevBuf<float> o, d, t;
        DevBuf<int32_t> tri;
        o.alloc(n * 3ull); d.alloc(n * 3ull); t.alloc(n); tri.alloc(n);
        HIP_CHECK(hipMemcpy(o.p, origins, n * 12ull, hipMemcpyHostToDevice));
        HIP_CHECK(hipMemcpy(d.p, dirs, n * 12ull, hipMemcpyHostToDevice));
        size_t lds = (size_t)sc->stack_cap * 256 * 8;
        if (traversal == CRT_TRAVERSAL_REFERENCE)
            hipLaunchKernelGGL(k_intersect<1>, dim3((n + 255) / 256), dim3(256), lds, 0, sc->dev, n, o.p, d.p, tri.p, t.p, sc->stack_cap);
        else
            hipLaunchKernelGGL(k_intersect<0>, dim3((n + 255) / 256), dim3(256), lds, 0, sc->dev, n, o.p, d.p, tri.p, t.p, sc->stack_cap);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipDeviceSynchronize());
        HIP_CHECK(hipMemcpy(out_tri, tri.p, n * 4ull, hipMemcpyDeviceToHost));
        HIP_CHECK(hipMemcpy(out_t, t.p, n * 4ull, hipMemcpyDeviceToHost));
        return CRT_OK;
    } catch (const HipFail& f) {
        return fail_hip(f);
    }
}

int crt_device_math(int device, const char* fn, uint32_t n, const float* a, const float* b, float* out)
{
    if (!fn || !a || !out) return fail(CRT_ERR_INVALID_ARG, "crt_device_math: null argument");
    static const char* names[] = {"sin", "cos", "tan", "acos", "atan2", "exp", "log10", "pow", "uniform", "sincos_s", "sincos_c"};
    int id = -1;
    for (int i = 0; i < 11; i++)
        if (std::strcmp(fn, names[i]) == 0) id = i;
    if (id < 0) return fail(CRT_ERR_INVALID_ARG, std::string("crt_device_math: unknown function ") + fn);
    if (n == 0) return CRT_OK;
    try {
        HIP_CHECK(hipSetDevice(device));
        DevBuf<float> da, db, dout;
        da.alloc(n); dout.alloc(n);
        HIP_CHECK(hipMemcpy(da.p, a, n * 4ull, hipMemcpyHostToDevice));
        if (b) { db.alloc(n); HIP_CHECK(hipMemcpy(db.p, b, n * 4ull, hipMemcpyHostToDevice)); }
        hipLaunchKernelGGL(k_math, dim3((n + 255) / 256), dim3(256), 0, 0, id, n, da.p, b ? db.p : nullptr, dout.p);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipDeviceSynchronize());
        HIP_CHECK(hipMemcpy(out, dout.p, n * 4ull, hipMemcpyDeviceToHost));
        return CRT_OK;
    } catch (const HipFail& f) {
        return fail_hip(f);
    }
}

int crt_device_philox(int device, uint32_t n, const uint32_t* ctr4, const uint32_t* key2, uint32_t* out4)
{
    if (!ctr4 || !key2 || !out4) return fail(CRT_ERR_INVALID_ARG, "crt_device_philox: null argument");
    if (n == 0) return CRT_OK;
    try {
        HIP_CHECK(hipSetDevice(device));
        DevBuf<uint32_t> c, k, o;
        c.alloc(n * 4ull); k.alloc(n * 2ull); o.alloc(n * 4ull);
        HIP_CHECK(hipMemcpy(c.p, ctr4, n * 16ull, hipMemcpyHostToDevice));
        HIP_CHECK(hipMemcpy(k.p, key2, n * 8ull, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_philox, dim3((n + 255) / 256), dim3(256), 0, 0, n, c.p, k.p, o.p);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipDeviceSynchronize());
        HIP_CHECK(hipMemcpy(out4, o.p, n * 16ull, hipMemcpyDeviceToHost));
        return CRT_OK;
    } catch (const HipFail& f) {
        return fail_hip(f);
    }
}

} // extern "C"

// cudaraytracing_amd/csrc/crt_mega3.hip -- k_mega3, the render kernel: one persistent launch per chunk of samples, path logic and traversal in the
// same waves, every wave the owner of a pool of rays in LDS (layouts and limits: crt_mega3.h; shared path logic: crt_path.h).
// Replaces view_render_kernel / cast_ray_v2 / DeviceBVH::intersect (include/Render.cuh:199-354, include/DeviceBVH.cuh:87-170).
#include "crt_internal.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace crtk {

#ifndef CRT_X_SLIGHT
#define CRT_X_SLIGHT 1 /* 0: the light table through vector loads only (A/B) */
#endif
struct NewRay {
    F3 o, d;
    float tl;
    uint32_t kind, flags;
};

__device__ __forceinline__ float fmin3(float a, float b, float c) { return __builtin_fminf(__builtin_fminf(a, b), c); }
__device__ __forceinline__ float fmax3(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }
__device__ __forceinline__ v2f v2(float a, float b) { v2f r; r.x = a; r.y = b; return r; }
__device__ __forceinline__ v2f v2s(float a) { v2f r; r.x = a; r.y = a; return r; }

// max of two wave-uniform integers on the scalar unit
__device__ __forceinline__ int smax(const int a, const int b)
{
    int r;
    asm("s_max_i32 %0, %1, %2" : "=s"(r) : "s"(a), "s"(b) : "scc");
    return r;
}

// Ring index in [0, 2 * QCAP) -> [0, QCAP).
// (CRT_RING_MODE 1: rings of 256 entries, the wrap is a mask; 2: the rings are stacks -- a batch is the NEWEST ids, no head, no tail, no wrap)
template <int QCAP>
__device__ __forceinline__ uint32_t ring_wrap(const uint32_t x)
{
#if CRT_RING_MODE == 1
    static_assert(QCAP == 256, "CRT_RING_MODE 1: rings of 256 entries");
    return x & 255u;
#elif CRT_RING_MODE == 2
    return x;
#else
    return min(x, x - (uint32_t)QCAP);
#endif
}
#if CRT_RING_MODE == 2
// (the count doubles as the place of the next id, i.e. as the addend of v_mbcnt, a vector operand: handed over as a scalar COPY, or the
// compiler moves the count itself into a vector register, where the scheduler's scalar maxima cannot reach it)
__device__ __forceinline__ uint32_t scalar_copy(int x) { asm volatile("" : "+s"(x)); return (uint32_t)x; }
#define RQ_PUSH_BASE(p_) (STATS ? (uint32_t)qn[p_] : scalar_copy(qn[p_])) /* where the next id goes (the counting kernels keep more scalars: theirs may live in vector registers) */
#define RQ_PUSH_ADV(p_, n_) { qn[p_] += (n_); }
#define RQ_POP_BASE(p_, take_) ((uint32_t)(qn[p_] - (take_)))
#define RQ_POP_ADV(p_, take_) { qn[p_] -= (take_); }
#else
#define RQ_PUSH_BASE(p_) ((uint32_t)qt[p_])
#if CRT_RING_MODE == 1
#define RQ_PUSH_ADV(p_, n_) { qn[p_] += (n_); qt[p_] = (qt[p_] + (n_)) & 255; }
#define RQ_POP_ADV(p_, take_) { qh[p_] = (qh[p_] + (take_)) & 255; qn[p_] -= (take_); }
#else
#define RQ_PUSH_ADV(p_, n_) { qn[p_] += (n_); qt[p_] += (n_); if (qt[p_] >= QCAP) qt[p_] -= QCAP; }
#define RQ_POP_ADV(p_, take_) { qh[p_] += (take_); if (qh[p_] >= QCAP) qh[p_] -= QCAP; qn[p_] -= (take_); }
#endif
#define RQ_POP_BASE(p_, take_) ((uint32_t)qh[p_])
#endif

// Where a ray goes once its traversal is over: a next-event sample to LA (LB after the last one of its vertex), a probe
// or a closest-hit ray that found a surface to LA, a closest-hit ray that found nothing to LC.
template <bool QUERY = false>
__device__ __forceinline__ uint32_t route_done(uint32_t rec_flags)
{
    if (QUERY) return PH3_LC;
    return ((rec_flags >> RR_ROUTE_SHIFT) & 3u) + (uint32_t)PH3_LA;
}
// the route bits of a new ray's record (NewRay flags -> record flags)
__device__ __forceinline__ uint32_t route_bits(uint32_t nr_flags)
{
    const uint32_t r = (nr_flags & RF_SHADOW) ? ((nr_flags & RF_LAST) ? 1u : 0u) : ((nr_flags & RF_PROBE) ? 0u : 2u);
    return r << RR_ROUTE_SHIFT;
}

// 1 / d per component (Ray.cuh:14), bit for bit the IEEE quotient: the short reciprocal where it is proven equal (rcp_ieee),
// the division itself for the other lanes behind a wave-uniform branch.
__device__ __forceinline__ F3 inv3_exact(const F3 d)
{
    F3 inv = f3(rcp_short(d.x), rcp_short(d.y), rcp_short(d.z));
    asm volatile("" : "+v"(inv.x), "+v"(inv.y), "+v"(inv.z));
    const bool ok = rcp_short_ok3(d.x, d.y, d.z);
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(!ok) != 0, 0)) {
        if (!ok) inv = f3(1 / d.x, 1 / d.y, 1 / d.z);
    }
    return inv;
}

// Writes the new ray into the pool record `id` and returns its first phase.
template <int MODE, bool QUERY = false, class LDS = Pool3Lds, bool IMPL = false>
__device__ __forceinline__ uint32_t start_ray(const DevScene& sc, LDS& S, uint32_t id, const NewRay& nr, PathCounters& cnt, const bool force_exact,
                                              bool& enters_exact)
{
    cnt.rays++;
    cnt.shadow += (nr.flags & RF_SHADOW) ? 1u : 0u;
    cnt.probe += (nr.flags & RF_PROBE) ? 1u : 0u;
    const F3 inv = inv3_exact(nr.d); // 1 / d (Ray.cuh:14)
    uint32_t flags = (nr.flags & ~(RF_SKIP | RF_SHADOW | RF_LAST | RF_PROBE)) | route_bits(nr.flags);
    // rays with a zero / denormal direction component can put NaNs into the slab test; they walk the reference
    // topology, whose box tests are the reference's own (crt_accel.h)
    // The 4-wide step (slab_quad_pruned) needs every plane distance (plane - o) * (1/d) of the tree to be FINITE: then no operand of
    // its v_max3 / v_min3 is a NaN, "+inf" can only mean "missed", and "no bound" can be any value >= FLT_MAX.  |plane - o| <=
    // coord_max + max |o|, so a product of that with max |1/d| at or below 2^126 cannot overflow (two roundings of 2^-24 on the way);
    // the comparison is false for a NaN anywhere and for an infinite origin, 1/d or scene coordinate.  A finite d keeps 1/d away from 0.
    const float max_o = __builtin_elementwise_maximum(__builtin_elementwise_maximum(absf(nr.o.x), absf(nr.o.y)), absf(nr.o.z));
    const float max_inv = __builtin_elementwise_maximum(__builtin_elementwise_maximum(absf(inv.x), absf(inv.y)), absf(inv.z));
    const bool finite = ((sc.coord_max + max_o) * max_inv <= 0x1p126f) & finite3(nr.d.x, nr.d.y, nr.d.z);
    if (MODE == 1 || !finite || force_exact) flags |= RF_EXACT;
    // (MODE 0 / 2: a ray that is not RF_EXACT walks the 4-wide tree)
    const int ref = (MODE != 1 && finite && !force_exact) ? (IMPL ? sc.root4i : sc.root4) : sc.root3_exact; // (IMPL: the tree without its rows of refs, inner4_step_dec)
    bool answered = false;
    float T = FLT_MAX;
    if (MODE != 1 && nr.kind == RAY_SHADOW) { // REFERENCE mode resolves shadow rays with the full closest-hit query, as blocked() does
        flags |= RF_ANYHIT;
        T = nr.tl;
        // a NaN or -inf limit can never be "blocked"; +inf is blocked by any hit
        answered = !(nr.tl == nr.tl) || nr.tl == -pinf() || (nr.flags & RF_SKIP) != 0;
    }
    if constexpr (LDS::DEC) {
        static_assert(!LDS::DEC || MODE == 2, "decoupled leaves: CRT_TRAVERSAL_EXACT");
        // (a scene that is one leaf has no inner node to start at: its rays take the reference-arithmetic arm, which hands leaf refs
        // to the queue one by one)
        if (ref < 0) flags |= RF_EXACT;
        S.A[id] = make_float4(nr.o.x, nr.o.y, nr.o.z, (flags & RF_ANYHIT) ? T : pinf());
        S.B[id] = make_float4(nr.d.x, nr.d.y, nr.d.z, __int_as_float(ref));
        S.best[id] = (unsigned long long)0x7f7fffffu << 32; // (FLT_MAX, no triangle)
        S.D[id] = flags;
        enters_exact = false;
        if (answered) return route_done<QUERY>(flags);
        enters_exact = (flags & RF_EXACT) != 0;
        return PH3_INNER;
    } else {
    S.A[id] = make_float4(nr.o.x, nr.o.y, nr.o.z, T);
    S.B[id] = make_float4(nr.d.x, nr.d.y, nr.d.z, __int_as_float(-1));
    S.node[id] = ref;
    S.D[id] = flags;
    enters_exact = false;
    if (answered) return route_done<QUERY>(flags);
    enters_exact = MODE != 1 && (flags & RF_EXACT) != 0; // (counted by the caller: the traversal steps of a pool without such rays skip their handling)
    return ref >= 0 ? PH3_INNER : PH3_LEAF;
    }
}

// DEC: the answer of a finished ray as the logic phases read it from the non-DEC record (A.w = distance, B.w = triangle)
template <class LDS>
__device__ __forceinline__ void ray_result(LDS& S, const uint32_t id, float4& qa, float4& qb)
{
    qa = S.A[id]; qb = S.B[id];
    if constexpr (LDS::DEC) {
        const unsigned long long b = S.best[id];
        qa.w = __uint_as_float((uint32_t)(b >> 32));
        qb.w = __uint_as_float(~(uint32_t)b);
    }
}
// DEC: where a complete ray goes (route_done); a closest-hit ray that has found a surface goes to LA instead of LC
template <bool QUERY>
__device__ __forceinline__ uint32_t route_complete(const uint32_t rec_flags, const bool has_hit)
{
    if (QUERY) return PH3_LC;
    const uint32_t r = (rec_flags >> RR_ROUTE_SHIFT) & 3u;
    return (has_hit ? (r & 1u) : r) + (uint32_t)PH3_LA;
}

// Visibility of a next-event sample (Render.cuh:19-27, :272): tl - hit.t > EPSILON with hit.t = FLT_MAX when nothing was
// hit.  An any-hit ray only ever records hits that satisfy the comparison, so its answer is "recorded a hit", plus the
// reference's own quirk that an infinite limit minus FLT_MAX is still "blocked".
template <int MODE>
__device__ __forceinline__ bool shadow_blocked(float tl, float T, int tri)
{
    if (MODE != 1) return tri >= 0 || tl - FLT_MAX > CRT_EPSILON;
    return tl - T > CRT_EPSILON;
}

// Backward recursion over k_mega3's vertex records, deepest first (Render.cuh:238-326; crt_path.h: finish_path is the wavefront pipeline's).
// Round 6 layout: vertex j of a path that went ON from it has  rec_a[j] = (L_dir.xyz, bits(triangle-row word: material | flags)), written
// by LB when the roulette lets the path continue, and  rec_b[j].w = cos to vertex j + 1, written when that vertex is found (rec_b[j].xyz,
// the direction that arrived at j, is written for SPECULAR vertices only: nothing else reads it).  The deepest vertex has no record: its
// L_dir arrives in the la plane (`have_ld`) when the path stopped there, is rec_a's when the ray that left it found nothing, and is not
// needed when it is an emitter.  Against one 16-byte record store more per vertex and one per path that stops (round 5).
__device__ __forceinline__ F3 finish_path_m3(const LParams& P, const Tables<false>& tb, const uint32_t slot, const int deepest, const bool emissive, const F3 ke,
                                            const bool have_ld, const F3 ld)
{
    const Pool& pl = P.pool;
    F3 L = f3(0.0f, 0.0f, 0.0f);
    if (deepest < 0) return L;
    const float inv_pdf_sphere = (float)(2.0f * 3.14159265358979323846); // Global.h:96-99
    if (emissive) {
        L = deepest == 0 ? add3(f3(0.0f, 0.0f, 0.0f), ke) : f3(0.0f, 0.0f, 0.0f); // :249-255, :323
    } else if (have_ld) {
        L = add3(f3(0.0f, 0.0f, 0.0f), ld); // final hit: direct light only (:316-319)
    } else {
        const float4 a = gld_rec(&pl.rec_a[(size_t)deepest * pl.n + slot]);
        L = add3(f3(0.0f, 0.0f, 0.0f), f3(a.x, a.y, a.z));
    }
    // (the loads of CRT_FINISH_PF vertices are fetched together, as in finish_path)
    for (int v = deepest - 1; v >= 0; v -= CRT_FINISH_PF) {
        float4 a[CRT_FINISH_PF], fm[CRT_FINISH_PF];
        float cs[CRT_FINISH_PF];
#pragma unroll
        for (int j = 0; j < CRT_FINISH_PF; j++) {
            const int vj = v - j > 0 ? v - j : 0;
            a[j] = gld_rec(&pl.rec_a[(size_t)vj * pl.n + slot]);
            cs[j] = __uint_as_float(gld((const uint32_t*)&pl.rec_b[(size_t)vj * pl.n + slot].w));
        }
#pragma unroll
        for (int j = 0; j < CRT_FINISH_PF; j++) fm[j] = mat_row(tb, TNM_MAT(__float_as_uint(a[j].w)), 0);
#pragma unroll
        for (int j = 0; j < CRT_FINISH_PF; j++) {
            if (v - j >= 0) {
                F3 ind = mul3(L, f3(fm[j].x, fm[j].y, fm[j].z)); // L (.) f_r * cos * inv_pdf / P_RR  (:293)
                ind = scale3(ind, cs[j]);
                ind = scale3(ind, inv_pdf_sphere);
                ind = div3(ind, P.p_rr);
                L = add3(ind, f3(a[j].x, a[j].y, a[j].z)); // :323
            }
        }
    }
    return L;
}

// The same with the limit known only as "is +inf" (TRI_CC below): tl - FLT_MAX > EPSILON holds for tl = +inf alone (a finite tl gives <= 0, NaN fails)
__device__ __forceinline__ bool shadow_blocked_bit(const bool tl_inf, const int tri) { return tri >= 0 || tl_inf; }
#define ST_TL_INF (1u << 12) /* state word of the la plane, bits 12 .. 15 are free: the in-flight next-event ray's limit is +inf */

// LA: consumes the result of a next-event sample that is not the last one of its vertex, of a closest-hit ray
// that found a surface, or of a probe ray; enters the vertex if it is new; sets up the next next-event sample.
// Returns PH3_NONE when a ray was emitted into nr, else the phase the path has to visit instead.
template <int MODE, bool RING = false>
__device__ __forceinline__ uint32_t logic_A(const LParams& P, const Tables<false>& tb, const uint32_t g, const float4 qa, const float4 qb, NewRay& nr,
                                            PathCounters& cnt, const bool trace_all)
{
    const DevScene& sc = P.sc;
    const Pool& pl = P.pool;
    // The phase is a chain of dependent loads (path planes -> triangle / material / light tables -> light triangle), and a wave
    // that waits issues nothing: everything whose address is known is fetched up front, needed by this lane's stage or not.
    //   round 1: the path planes and the triangle record of the hit (the new vertex, if this ray found one)
    // TRI_CC (round 6, every mode but REFERENCE): the vertex's triangle rides in cc.w -- cc is written with every next-event sample and read
    // by every visit anyway -- instead of a store of its own into the id plane at every vertex; the limit of the sample's ray, which was
    // there, is needed as "is +inf" only (shadow_blocked) and is a bit of the state word.  REFERENCE compares the limit with the nearest
    // hit's distance and keeps the round-5 planes.  (docs/experiments.md 6.12: the store of the vertex position that went with it was the
    // gain, C2 75.8 -> 73.6 ms; this one is level in time and takes 6.7 % off the bytes written.)
    constexpr bool TRI_CC = CRT_X_NOVN && MODE != 1;
    const float4 la = gld(&pl.la[g]);
    const uint4 idv = load_path_id<RING, !TRI_CC>(P, g);
    const float4 cc = gld(&pl.cc[g]); // pending next-event contribution; .w = bits(triangle of the vertex the samples belong to) (REFERENCE: distance to the light sample)
    const uint32_t st = __float_as_uint(la.w);
    const uint32_t stage = (st >> 8) & 15u;
    // (a path's first visit -- its camera ray found vertex 0 -- has no vertex in the planes yet: cc.w is what the slot's last path left, or
    // never written; the speculative row is then row 0 and is not used)
    const uint32_t vtri_old = TRI_CC ? ((st & 0xfffu) == ((uint32_t)ST_HIT << 8) ? 0u : __float_as_uint(cc.w)) : idv.w;
#if CRT_X_NOVN
    const float4 vn = gld(&sc.tri_nm[vtri_old]); // (normal, material) of the vertex in the planes: from its triangle
#else
    const float4 vn = gld(&pl.vn[g]);
#endif
    const float res_t = qa.w;
    const int res_tri = __float_as_int(qb.w);
    const float4 gq_hit = gld(&sc.tri_nm[res_tri >= 0 ? res_tri : 0]);
    //   round 2: material rows of the vertex the samples belong to after this visit (the new one for ST_HIT), row 1 of the
    //   vertex the ray left (specular flag, ST_HIT), and the light of the sample that is set up below
    //   (the material word of a slot's very first vertex comes from row 0, see vtri_old: the index is clamped into the table all the same)
    // (round 6: "is an emitter" / "is SPECULAR" ride in the two top bits of the triangle row's material word -- rows 1 of two materials were
    // fetched for those two bits alone, and what bounds this kernel is the NUMBER of vector-memory instructions)
    // (the material word of the samples' vertex in the id plane's free second word, so that its BSDF row is fetched WITH the triangle row instead
    // of after it, was measured in round 6: C2 +0.8 % -- one more store per vertex, and the round it saves is not the phase's last)
    const uint32_t mat_old = min(TNM_MAT(__float_as_uint(vn.w)), P.n_mats - 1u);
    const uint32_t mat_cur = stage == ST_HIT ? TNM_MAT(__float_as_uint(gq_hit.w)) : mat_old;
    uint32_t tnm_cur = stage == ST_HIT ? __float_as_uint(gq_hit.w) : __float_as_uint(vn.w);
    float4 m0_cur = mat_row(tb, mat_cur, 0);
    const uint32_t n_nee = (uint32_t)(sc.n_lights * P.lsn);
    const uint32_t q_next = stage == ST_SHADOW ? (st >> 16) + 1 : 0u;
    uint4 lg_next = make_uint4(0u, 1u, 0u, 0u);
    // A scene with ONE light has its table entry fetched through the scalar cache: one vector load per visit fewer (round 6: C2 72.11 -> 71.75 ms).
    // (The eight rows of a light of <= 2 triangles -- the quad of a Cornell box -- fetched the same way and picked per lane: C2 level, and
    // veach-mis, which does not take that path, +1.2 % from the second copy of the set-up code; not kept.)
    if (n_nee > 0) {
        if (CRT_X_SLIGHT && sc.n_lights == 1) {
            const crt_u4v_ l0_ = *(const __attribute__((address_space(4))) crt_u4v_*)tb.lights;
            lg_next = make_uint4(l0_.x, l0_.y, l0_.z, l0_.w);
        } else lg_next = gld(&tb.lights[fast_div(q_next < n_nee ? q_next : 0u, P.lsn_div.m, P.lsn_div.sh)]);
    }
    Lane s;
    s.depth = st & 255u; s.q = st >> 16; s.stage = stage;
    s.Ld = f3(la.x, la.y, la.z);
    s.nrm = f3(vn.x, vn.y, vn.z); s.mat = TNM_MAT(__float_as_uint(vn.w));
    s.pixel_index = idv.x; s.k = idv.y; s.item = idv.z;
    s.ro = f3(qa.x, qa.y, qa.z); s.tl = 0.0f;
    s.rd = f3(qb.x, qb.y, qb.z);
    s.pos = s.ro; s.vtri = vtri_old; s.c = f3(0.0f, 0.0f, 0.0f); s.kind = RAY_NONE;
    bool do_enter = false;
    if (stage == ST_SHADOW) {
        // visibility of next-event sample q (Render.cuh:19-27, :272-284); shadow rays start at the vertex: s.pos == s.ro
        const bool blocked = TRI_CC ? shadow_blocked_bit((st & ST_TL_INF) != 0u, res_tri) : shadow_blocked<MODE>(cc.w, res_t, res_tri);
        if (!blocked) s.Ld = add3(s.Ld, f3(cc.x, cc.y, cc.z));
        s.q++;
    } else if (stage == ST_HIT) {
        // the camera / bounce ray found vertex `depth` (Render.cuh:207-213)
        const F3 pos = add3(s.ro, scalel3(res_t, s.rd)); // DeviceTriangle.cuh:50
        do_enter = true;
        if (s.depth > 0) {
            // the previous vertex (vn: its triangle's row) is not the deepest one: cosine of its indirect term (Render.cuh:291)
            const size_t pr = (size_t)(s.depth - 1) * pl.n + g;
            const F3 pn = s.nrm;
            float cos_prev = dot3(unit3(sub3(pos, s.ro)), pn); // prev.pos == origin of this ray
            cos_prev = cos_prev > 0.0f ? cos_prev : 0.0f;
            gst_rec(&pl.rec_b[pr].w, cos_prev); // (finish_path_m3: the cosine lives in rec_b.w, rec_a.w is the vertex's material)
            if (TNM_SPECULAR(__float_as_uint(vn.w))) { // SPECULAR: emitter probe, Render.cuh:294-303
                const float ns = mat_row(tb, s.mat, 0).w;
                const float4 pb = gld_rec(&pl.rec_b[pr]); // direction that arrived at the previous vertex
                const float delta_coeff = (float)((double)(det_expf(25 / ns) - 1) / (2.71828182845904523536 - 1));
                const F3 in = unit3(f3(pb.x, pb.y, pb.z));
                const F3 out = sub3(in, scale3(pn, 2.f * dot3(in, pn)));
                const float d_theta = (float)((double)(delta_coeff * 30) * 3.14159265358979323846 / 180);
                const float d_phi = (float)((double)(delta_coeff * 120) * 3.14159265358979323846 / 180);
                const U4 rp = rng_draw(P.seed, s.pixel_index, s.k, s.depth - 1, RNG_PROBE, 0);
                const F3 refd = unit3(sample_lobe(out, d_theta, d_phi, rng_uniform(rp.x), rng_uniform(rp.y)));
                // the probe leaves from prev.pos (= this ray's origin); the bounce direction waits in rec_b[depth]
                gst_rec(&pl.rec_b[(size_t)s.depth * pl.n + g], make_float4(s.rd.x, s.rd.y, s.rd.z, 0.0f));
                gst(&pl.vx[g], make_float4(pos.x, pos.y, pos.z, __int_as_float(res_tri)));
                gst(&pl.la[g], make_float4(s.Ld.x, s.Ld.y, s.Ld.z, __uint_as_float(s.depth | ((uint32_t)ST_PROBE << 8) | (s.q << 16))));
                nr.o = s.ro; nr.d = unit3(refd); /* Ray.cuh:13 */ nr.tl = 0.0f; nr.kind = RAY_CLOSEST; nr.flags = RF_PROBE;
                return PH3_NONE;
            }
        }
        s.pos = pos; s.vtri = (uint32_t)res_tri;
    } else { // ST_PROBE: the probe ray of vertex depth-1 (Render.cuh:304-313); vn still describes that vertex
        const float4 vx = gld(&pl.vx[g]);
        s.pos = f3(vx.x, vx.y, vx.z); s.vtri = __float_as_uint(vx.w);
        if (res_tri >= 0) {
            const int hmat = gld(&sc.tri_mat[res_tri]);
            const float4 h1 = mat_row(tb, hmat, 1);
            if (__float_as_uint(h1.w) & 1u) {
                const float4 h2 = mat_row(tb, hmat, 2);
                const size_t pr = (size_t)(s.depth - 1) * pl.n + g;
                const F3 pn = s.nrm;
                const float4 pm0 = mat_row(tb, s.mat, 0), pm1 = mat_row(tb, s.mat, 1);
                const float log_shininess = det_log10f(pm0.w);
                const float shininess_coeff = (float)((double)log_shininess * 0.5 + 1);
                const float ip = (float)(2.0f * 3.14159265358979323846) / 8.f;
                const F3 hp = add3(s.ro, scalel3(res_t, s.rd));
                float ct = dot3(unit3(sub3(hp, s.ro)), pn); // probe origin == prev.pos
                ct = ct > 0.0f ? ct : 0.0f;
                // shininess * (ke (.) kd) * cos * inv_pdf  (:311, eager)
                const F3 kekd = mul3(f3(h2.x, h2.y, h2.z), f3(pm1.x, pm1.y, pm1.z));
                const F3 temp = scale3(scale3(scalel3(shininess_coeff, kekd), ct), ip);
                float4 a = gld_rec(&pl.rec_a[pr]);
                a.x = a.x + temp.x; a.y = a.y + temp.y; a.z = a.z + temp.z;
                gst_rec(&pl.rec_a[pr], a);
            }
        }
        const float4 pb = gld_rec(&pl.rec_b[(size_t)s.depth * pl.n + g]); // the bounce direction that found the current vertex
        s.rd = f3(pb.x, pb.y, pb.z);
        do_enter = true;
    }
    if (do_enter) { // a new vertex (pos, vtri) at `depth`, reached along s.rd
        float4 gq = gq_hit;
        if (stage == ST_PROBE) { // the vertex was found by the ray before the probe: its triangle waits in the vx plane
            gq = gld(&sc.tri_nm[s.vtri]);
            m0_cur = mat_row(tb, TNM_MAT(__float_as_uint(gq.w)), 0); tnm_cur = __float_as_uint(gq.w);
        }
        s.nrm = f3(gq.x, gq.y, gq.z);
        s.mat = TNM_MAT(__float_as_uint(gq.w));
        // (the direction that arrived: read by the probe of a SPECULAR vertex when the next vertex is found, by nothing else)
        if (TNM_SPECULAR(tnm_cur)) gst_rec(&pl.rec_b[(size_t)s.depth * pl.n + g], make_float4(s.rd.x, s.rd.y, s.rd.z, 0.0f));
        // (round 6: the vertex position is not written to the vx plane any more -- LB takes it from the slot's ray record, which is a
        // next-event ray of this vertex or, for a vertex without one, is given the position below.  The plane lives on for the probe rays.)
#if CRT_X_NOVN
        if (!TRI_CC) store_path_tri(P, g, s.vtri);
#else
        gst(&pl.vn[g], make_float4(s.nrm.x, s.nrm.y, s.nrm.z, __uint_as_float(s.mat)));
#endif
        if (TNM_EMITTER(tnm_cur)) { // emitter: the path ends here (Render.cuh:210); TRI_CC: LC finds the emitter's triangle in la.x
            gst(&pl.la[g], make_float4(TRI_CC ? __uint_as_float(s.vtri) : 0.0f, 0.0f, 0.0f, __uint_as_float(s.depth | ((uint32_t)ST_FIN << 8) | (1u << 16))));
            return PH3_LC;
        }
        s.Ld = f3(0.0f, 0.0f, 0.0f);
        s.q = 0;
        if (n_nee == 0) {
            gst(&pl.la[g], make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(s.depth | ((uint32_t)ST_NEED << 8))));
            if (TRI_CC) gst(&pl.cc[g].w, __uint_as_float(s.vtri)); // (a vertex without a sample: its triangle for LB and for the next vertex's visit)
            nr.o = s.pos; // (no ray: the caller puts the position into the slot's ray record for LB)
            return PH3_LB;
        }
    }
    // next-event samples of the current vertex, from q on.  The reference traces every shadow ray and then adds
    // Le (.) f_r * cos * cos' * ... to L_dir if it is unblocked (Render.cuh:272-284).  When that contribution is exactly
    // zero (the surface or the light faces away: the cosines are clamped to 0; a black BSDF) the addition is the identity
    // whatever the ray finds -- L_dir is never -0 -- so the FAST traversal answers the sample without tracing it.  It still
    // counts as a ray of the reference (`rays`, `shadow_rays`); `rays_untraced` says how many there were.  A NaN contribution
    // fails the comparison and is traced.  (CRT_TRAVERSAL_REFERENCE traces everything: its counters are the reference's visit set.)
    // The next sample of such a lane is set up right here while enough lanes of the batch need it (setup_shadow is the most
    // expensive section of the phase and the others wait); the last few stragglers are instead handed to start_ray as
    // "answered" (RF_SKIP) and go back to the ring of their consumer, which adds the zero contribution.
#ifndef LA_LOOP_MIN
#define LA_LOOP_MIN 16
#endif
    const float4 m0 = m0_cur;
    bool skip;
    for (bool first = true;; first = false) {
        if (first) setup_shadow_lg(P, s, f3(m0.x, m0.y, m0.z), lg_next); // (s.q == q_next: the light entry is already here)
        else setup_shadow(P, tb, s, f3(m0.x, m0.y, m0.z));
        skip = MODE != 1 && !trace_all && (s.c.x == 0.0f && s.c.y == 0.0f && s.c.z == 0.0f);
        if (!skip) break;
        cnt.untraced++;
        if (__popcll(__builtin_amdgcn_ballot_w64(true)) < LA_LOOP_MIN) break; // (the lanes still in the loop are the ones that skip)
        cnt.rays++; cnt.shadow++;
        s.q++;
        if (s.q == n_nee) { // that was the last sample of the vertex: on to the roulette
            gst(&pl.la[g], make_float4(s.Ld.x, s.Ld.y, s.Ld.z, __uint_as_float(s.depth | ((uint32_t)ST_NEED << 8))));
            if (TRI_CC) gst(&pl.cc[g].w, __uint_as_float(s.vtri)); // (every sample of the vertex may have been answered here: none has written cc)
            nr.o = s.pos;
            return PH3_LB;
        }
    }
    gst(&pl.la[g], make_float4(s.Ld.x, s.Ld.y, s.Ld.z, __uint_as_float(s.depth | ((uint32_t)ST_SHADOW << 8) | (s.q << 16) | ((TRI_CC && s.tl == pinf()) ? ST_TL_INF : 0u))));
    gst(&pl.cc[g], make_float4(s.c.x, s.c.y, s.c.z, TRI_CC ? __uint_as_float(s.vtri) : s.tl));
    nr.o = s.ro; nr.d = s.rd; nr.tl = s.tl; nr.kind = RAY_SHADOW;
    nr.flags = RF_SHADOW | (s.q + 1 == n_nee ? RF_LAST : 0u) | (skip ? RF_SKIP : 0u);
    return PH3_NONE;
}

// LB: direct light of vertex `depth` is complete -> vertex record, Russian roulette, bounce (Render.cuh:210-228).
template <int MODE, bool RING = false>
__device__ __forceinline__ uint32_t logic_B(const LParams& P, const uint32_t g, const float4 qa, const float4 qb, NewRay& nr)
{
    const Pool& pl = P.pool;
    constexpr bool TRI_CC = CRT_X_NOVN && MODE != 1; // (see logic_A)
    const float4 la = gld(&pl.la[g]);
    const uint4 idv = load_path_id<RING, !TRI_CC>(P, g);
    const float4 cc = gld(&pl.cc[g]); // (with the other planes, not after the stage is known: one round trip less, see logic_A)
    const uint32_t st = __float_as_uint(la.w);
    const uint32_t stage = (st >> 8) & 15u;
    uint32_t depth = st & 255u;
    F3 Ld = f3(la.x, la.y, la.z);
    if (stage == ST_SHADOW) { // the last next-event sample (Render.cuh:272-284)
        const bool blocked = TRI_CC ? shadow_blocked_bit((st & ST_TL_INF) != 0u, __float_as_int(qb.w)) : shadow_blocked<MODE>(cc.w, qa.w, __float_as_int(qb.w));
        if (!blocked) Ld = add3(Ld, f3(cc.x, cc.y, cc.z));
    }
    bool stop = depth == CRT_BOUNCE_STACK_SIZE - 1; // bounce stack full
    U4 rb;
    rb.x = rb.y = rb.z = rb.w = 0;
    if (!stop) {
        rb = rng_draw(P.seed, idv.x, idv.y, depth, RNG_BOUNCE, 0);
        stop = rng_uniform(rb.x) > P.p_rr;
    }
    if (stop) { // the deepest vertex: its L_dir goes to LC in the la plane, it has no record (finish_path_m3)
        gst(&pl.la[g], make_float4(Ld.x, Ld.y, Ld.z, __uint_as_float(depth | ((uint32_t)ST_FIN << 8))));
        return PH3_LC;
    }
    // the vertex: the origin of the slot's last ray -- a next-event ray starts at its vertex (setup_shadow_lg) -- or what LA's caller put there
#if CRT_X_NOVN
    const float4 vn = gld(&P.sc.tri_nm[TRI_CC ? __float_as_uint(cc.w) : idv.w]);
#else
    const float4 vn = gld(&pl.vn[g]);
#endif
    const float4 vx = qa;
    gst_rec(&pl.rec_a[(size_t)depth * pl.n + g], make_float4(Ld.x, Ld.y, Ld.z, vn.w)); // the vertex's record: L_dir and its material (with the row's flag bits)
    const F3 ndir = unit3(sample_hemisphere(f3(vn.x, vn.y, vn.z), rng_uniform(rb.y), rng_uniform(rb.z)));
    // (leaving this store out -- the plane then still holds a state only LB consumes, which LA and LC can read as "the ray for vertex depth + 1
    // is in flight" -- was measured in round 6: C2 +0.4 %, veach-mis +0.3 %: LA's load of the line then misses the L2 the store had left it in)
    depth++;
    gst(&pl.la[g], make_float4(Ld.x, Ld.y, Ld.z, __uint_as_float(depth | ((uint32_t)ST_HIT << 8))));
    nr.o = f3(vx.x, vx.y, vx.z); nr.d = unit3(ndir); /* Ray.cuh:13 */ nr.tl = 0.0f; nr.kind = RAY_CLOSEST; nr.flags = 0;
    return PH3_NONE;
}

// LC: the path is complete (miss, emitter, roulette, stack full) -> backward recursion (Render.cuh:238-326), next
// work item and its camera ray (Render.cuh:344-347).  Returns LC_DEAD when the work items are exhausted (the ray slot dies), LC_RAY
// with the camera ray of a new path, or -- commit ring only -- LC_WAIT: the slot holds a work item it may not start yet and comes
// back to this phase.  fin_key: see ring_publish.
enum { LC_DEAD = 0, LC_RAY = 1, LC_WAIT = 2 };
template <int MODE, bool RING>
__device__ __forceinline__ int logic_C(const LParams& P, const Tables<false>& tb, const uint32_t g, PathCounters& cnt, NewRay& nr, uint32_t& fin_key)
{
    constexpr bool TRI_CC = CRT_X_NOVN && MODE != 1; // (see logic_A: an emitter's triangle arrives in la.x)
    const Pool& pl = P.pool;
    const float4 la = gld(&pl.la[g]);
    const uint4 idv = load_path_id<RING, !TRI_CC>(P, g); // (with la, not after the stage is known: one round trip less, see logic_A)
    const uint32_t st = __float_as_uint(la.w);
    const uint32_t stage = (st >> 8) & 15u;
    const uint32_t depth = st & 255u;
    constexpr bool ring = RING;
    const bool waiting = ring && stage == ST_WAIT;
    // The next work item is asked for NOW -- one atomic on the wave's home cursor for all its lanes -- and looked at after the
    // backward recursion: the cursor's round trip hides behind the recursion's own loads (the earlier attempt read the answer with
    // a readfirstlane at once, which waits).  A home shard that has run dry (the end of a launch) falls back to grab_item below.
    const uint32_t home_ = RING ? blockIdx.x & (P.ring_shards - 1u) : blockIdx.x & (ITEM_SHARDS - 1);
    const uint32_t lo_ = home_ * P.items_per_shard, hi_ = min(lo_ + P.items_per_shard, P.n_items);
    const unsigned long long gmask_ = __ballot(!waiting);
    const int lane_ = threadIdx.x & 63;
    const uint32_t grank_ = (uint32_t)__popcll(gmask_ & ((1ull << lane_) - 1ull));
    unsigned int pre_base_ = 0;
    const bool pre_ok_ = lo_ < P.n_items && gmask_ != 0ull;
    if (pre_ok_ && lane_ == __ffsll((long long)gmask_) - 1) pre_base_ = __hip_atomic_fetch_add((CRT_GAS unsigned int*)(P.item_next + home_ * ITEM_STRIDE), (unsigned int)__popcll(gmask_), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t home_word_ = 0;
    if (RING) home_word_ = ring_load(P.ring_state + home_ * ITEM_STRIDE);
    if (stage != ST_NEW && !waiting) {
        int deepest = (int)depth;
        bool emissive = false;
        F3 ke = f3(0.0f, 0.0f, 0.0f);
        if (stage == ST_HIT) deepest = (int)depth - 1; // the ray that looked for vertex `depth` missed (Render.cuh:210)
        else if ((st >> 16) & 1u) {
            emissive = true;
#if CRT_X_NOVN
            const float4 m2 = mat_row(tb, TNM_MAT(__float_as_uint(gld(&P.sc.tri_nm[TRI_CC ? __float_as_uint(la.x) : idv.w]).w)), 2);
#else
            const float4 m2 = mat_row(tb, __float_as_uint(gld(&pl.vn[g]).w), 2);
#endif
            ke = f3(m2.x, m2.y, m2.z);
        }
        const F3 L = finish_path_m3(P, tb, g, deepest, emissive, ke, stage == ST_FIN && !emissive, f3(la.x, la.y, la.z));
        if (ring) {
            const uint32_t sh = fast_div(idv.z, P.items_per_shard_div.m, P.items_per_shard_div.sh), c = idv.z - sh * P.items_per_shard;
            const uint32_t s = fast_div(c, P.spsh_div.m, P.spsh_div.sh), rs = s & P.ring_mask;
            float4* Lr = P.L + (size_t)rs * P.ring_stride + (size_t)sh * P.spsh + (c - s * P.spsh);
            ring_store16(Lr, L.x, L.y, L.z);
            fin_key = (sh << 16) | rs;
        } else {
            // written once, read once by k_accumulate after the launch: a streaming store keeps it from displacing the path state
            // and the scene in L2
            // (ONE 16-byte store -- three 4-byte ones until round 5: the instruction count is what costs)
            typedef float f4v_ __attribute__((ext_vector_type(4)));
            f4v_ Lv_; Lv_.x = L.x; Lv_.y = L.y; Lv_.z = L.z; Lv_.w = 0.0f;
            __builtin_nontemporal_store(Lv_, (CRT_GAS f4v_*)&P.L[idv.z]);
        }
    }
    bool first_ = pre_ok_ && !waiting;
    for (;;) {
        uint32_t item = ITEM_NONE;
        if (waiting) item = idv.z; // the item this slot was handed earlier
        else {
            if (first_) { // the answer of the atomic issued above (the leader is the first active lane)
                const unsigned long long idx_ = (unsigned long long)lo_ + (unsigned int)__builtin_amdgcn_readfirstlane((int)pre_base_) + grank_;
                if (idx_ < hi_) item = (uint32_t)idx_;
                first_ = false;
            }
            if (item == ITEM_NONE) item = RING ? grab_item_ring(P.item_next, P.items_per_shard, P.ring_shards, home_)
                                               : grab_item(nullptr, P.item_next, P.items_per_shard, P.n_items, blockIdx.x & (ITEM_SHARDS - 1));
            if (item == ITEM_NONE) return LC_DEAD;
            if (P.item_list) { // the tail of every cursor shard is handed out "paths that stop at their first vertex last" (k_order_items)
                const uint32_t sh_ = fast_div(item, P.items_per_shard_div.m, P.items_per_shard_div.sh);
                const uint32_t slo_ = sh_ * P.items_per_shard, shi_ = min(slo_ + P.items_per_shard, P.n_items);
                const uint32_t wlo_ = shi_ - min(P.order_window, shi_ - slo_);
                // (agent-scope load, as k_order_items' stores: with plain accesses the FIRST frame of a render created after other renders of
                // the process came out with 10 - 400 work items of the 589 824 of a 96 x 64 x 96 frame never run -- their list entries read as
                // what an earlier kernel had left at the address -- in half of the runs once the launches' timing had changed; docs/experiments.md 6)
#if defined(CRT_HANDOFF_PLAIN) || defined(CRT_HANDOFF_PLAIN_LOAD) /* experiment builds only (tools/handoff_ab.sh): the accesses as they were */
                if (item >= wlo_) item = ((CRT_GAS const unsigned int*)P.item_list)[sh_ * P.order_window + (item - wlo_)];
#else
                if (item >= wlo_) item = __hip_atomic_load((CRT_GAS const unsigned int*)&P.item_list[sh_ * P.order_window + (item - wlo_)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
            }
        }
        bool valid; uint32_t pi, pj, pixel_index, k;
        decode_item<RING>(P, item, pixel_index, k, valid, pi, pj);
        if (!valid) continue; // padding slot of a ragged tile: take another item
        if (ring && !ring_gate_open(P, item, home_, home_word_)) { // its sample's slot of the ring is not free yet: hold the item
            if (!waiting) {
                store_path_id<!TRI_CC>(P, g, item);
                gst(&pl.la[g], make_float4(0.0f, 0.0f, 0.0f, __uint_as_float((uint32_t)ST_WAIT << 8)));
            }
            return LC_WAIT;
        }
        cnt.paths++;
        if (!waiting) store_path_id<!TRI_CC>(P, g, item);
        const U4 rj = rng_draw(P.seed, pixel_index, k, 0, RNG_JITTER, 0);
        const float x = (2 * ((int)pi + rng_uniform(rj.x)) / P.width - 1) * P.scale * P.ar;
        const float y = (1 - 2 * ((int)pj + rng_uniform(rj.y)) / P.height) * P.scale;
        const F3 cd = unit3(f3(-x, y, 1));
        const F3 wd = f3(P.inv_view[0] * cd.x + (P.inv_view[3] * cd.y + P.inv_view[6] * cd.z),
                         P.inv_view[1] * cd.x + (P.inv_view[4] * cd.y + P.inv_view[7] * cd.z),
                         P.inv_view[2] * cd.x + (P.inv_view[5] * cd.y + P.inv_view[8] * cd.z));
        gst(&pl.la[g], make_float4(0.0f, 0.0f, 0.0f, __uint_as_float((uint32_t)ST_HIT << 8)));
        nr.o = f3(P.eye[0], P.eye[1], P.eye[2]); nr.d = unit3(wd); /* Ray.cuh:13 */ nr.tl = 0.0f; nr.kind = RAY_CLOSEST; nr.flags = 0;
        return LC_RAY;
    }
}

// crt_intersect's form of LC: the work items are query rays (origin, normalised direction); a finished ray's record holds
// the answer (T = distance or FLT_MAX, best triangle or -1), which goes to L[ray].  The rays walk exactly the traversal phases
// of the render (4-wide tree, packed pair tests, tie rule, pruning) -- DeviceBVH::intersect (DeviceBVH.cuh:128-170) per ray.
__device__ __forceinline__ bool query_C(const LParams& P, const uint32_t g, const float4 qa, const float4 qb, NewRay& nr)
{
    const Pool& pl = P.pool;
    const float4 la = gld(&pl.la[g]);
    const uint4 idv = load_path_id(P, g);
    if (((__float_as_uint(la.w) >> 8) & 15u) != ST_NEW) gst(&P.L[idv.z], make_float4(qa.w, qb.w, 0.0f, 0.0f));
    const uint32_t item = grab_item(nullptr, P.item_next, P.items_per_shard, P.n_items, blockIdx.x & (ITEM_SHARDS - 1));
    if (item == ITEM_NONE) return false;
    store_path_id(P, g, item);
    gst(&pl.la[g], make_float4(0.0f, 0.0f, 0.0f, __uint_as_float((uint32_t)ST_HIT << 8)));
    const float4 o = gld(&P.q_o[item]), d = gld(&P.q_d[item]);
    nr.o = f3(o.x, o.y, o.z); nr.d = f3(d.x, d.y, d.z); nr.tl = o.w; nr.kind = __float_as_uint(d.w); nr.flags = RF_QUERY;
    return true;
}

// Both child boxes of an inner node at once (hit_AABB, DeviceBVH.cuh:87-126); lane .x = left child, .y = right child.
// Node layout: crt_device.h (nodes3).  exact = reference arithmetic (sign-selected planes, x<y?x:y minima) for rays with
// non-finite operands; otherwise minima / maxima of the two plane distances, which are the same numbers.
__device__ __forceinline__ void slab_pair(const float4 n0, const float4 n1, const float4 n2, const F3 o, const F3 inv, const F3 d, const bool exact,
                                          bool& hl, bool& hr, float& tl, float& tr)
{
    const v2f tx0 = (v2(n0.x, n0.y) - v2s(o.x)) * v2s(inv.x), ty0 = (v2(n0.z, n0.w) - v2s(o.y)) * v2s(inv.y), tz0 = (v2(n1.x, n1.y) - v2s(o.z)) * v2s(inv.z);
    const v2f tx1 = (v2(n1.z, n1.w) - v2s(o.x)) * v2s(inv.x), ty1 = (v2(n2.x, n2.y) - v2s(o.y)) * v2s(inv.y), tz1 = (v2(n2.z, n2.w) - v2s(o.z)) * v2s(inv.z);
    float el, er, xl, xr;
    if (!exact) {
        el = fmax3(__builtin_fminf(tx0.x, tx1.x), __builtin_fminf(ty0.x, ty1.x), __builtin_fminf(tz0.x, tz1.x));
        er = fmax3(__builtin_fminf(tx0.y, tx1.y), __builtin_fminf(ty0.y, ty1.y), __builtin_fminf(tz0.y, tz1.y));
        xl = fmin3(__builtin_fmaxf(tx0.x, tx1.x), __builtin_fmaxf(ty0.x, ty1.x), __builtin_fmaxf(tz0.x, tz1.x));
        xr = fmin3(__builtin_fmaxf(tx0.y, tx1.y), __builtin_fmaxf(ty0.y, ty1.y), __builtin_fmaxf(tz0.y, tz1.y));
    } else {
        const bool nx = d.x < 0, ny = d.y < 0, nz = d.z < 0; // the swap of DeviceBVH.cuh:101-119
        el = maxf_ref(maxf_ref(nx ? tx1.x : tx0.x, ny ? ty1.x : ty0.x), nz ? tz1.x : tz0.x);
        er = maxf_ref(maxf_ref(nx ? tx1.y : tx0.y, ny ? ty1.y : ty0.y), nz ? tz1.y : tz0.y);
        xl = minf_ref(minf_ref(nx ? tx0.x : tx1.x, ny ? ty0.x : ty1.x), nz ? tz0.x : tz1.x);
        xr = minf_ref(minf_ref(nx ? tx0.y : tx1.y, ny ? ty0.y : ty1.y), nz ? tz0.y : tz1.y);
    }
    hl = (el <= xl + CRT_EPSILON) & (xl >= 0);
    hr = (er <= xr + CRT_EPSILON) & (xr >= 0);
    tl = el; tr = er;
}

// slab_pair for rays with finite operands, with the pruning test folded in: returns each child's entry distance, or +inf when
// the box is missed (hit_AABB: t_enter <= t_exit + EPSILON && t_exit >= 0) or entered beyond `bound`
// (t_enter <= min(t_exit + EPSILON, bound) is the conjunction of the two upper limits; a NaN box -- an empty slot -- fails).
__device__ __forceinline__ void slab_pair_pruned(const float4 n0, const float4 n1, const float4 n2, const F3 o, const F3 inv, const float bound,
                                                 float& tl, float& tr)
{
    const v2f tx0 = (v2(n0.x, n0.y) - v2s(o.x)) * v2s(inv.x), ty0 = (v2(n0.z, n0.w) - v2s(o.y)) * v2s(inv.y), tz0 = (v2(n1.x, n1.y) - v2s(o.z)) * v2s(inv.z);
    const v2f tx1 = (v2(n1.z, n1.w) - v2s(o.x)) * v2s(inv.x), ty1 = (v2(n2.x, n2.y) - v2s(o.y)) * v2s(inv.y), tz1 = (v2(n2.z, n2.w) - v2s(o.z)) * v2s(inv.z);
    const float el = fmax3(__builtin_fminf(tx0.x, tx1.x), __builtin_fminf(ty0.x, ty1.x), __builtin_fminf(tz0.x, tz1.x));
    const float er = fmax3(__builtin_fminf(tx0.y, tx1.y), __builtin_fminf(ty0.y, ty1.y), __builtin_fminf(tz0.y, tz1.y));
    const float xl = fmin3(__builtin_fmaxf(tx0.x, tx1.x), __builtin_fmaxf(ty0.x, ty1.x), __builtin_fmaxf(tz0.x, tz1.x));
    const float xr = fmin3(__builtin_fmaxf(tx0.y, tx1.y), __builtin_fmaxf(ty0.y, ty1.y), __builtin_fmaxf(tz0.y, tz1.y));
    const float inf = pinf();
    tl = ((el <= __builtin_fminf(xl + CRT_EPSILON, bound)) & (xl >= 0)) ? el : inf;
    tr = ((er <= __builtin_fminf(xr + CRT_EPSILON, bound)) & (xr >= 0)) ? er : inf;
}

// Plane-major nodes (CRT_NODE_SIGNSEL): the four children's near planes and far planes of each axis arrive as one float4 each,
// picked per ray by the sign of its direction -- hit_AABB's own swap (DeviceBVH.cuh:101-119) done by the load address instead
// of by comparisons: t_enter = max of the three near distances, t_exit = min of the three far ones (no operand is a NaN for a
// ray with finite origin and 1/d and a finite box: x>y?x:y and v_max3 / v_min3 are the same numbers).  An empty slot is the
// inverted box (+inf, -inf): t_enter = +inf, t_exit = -inf for either sign.
template <bool PRUNE = true>
__device__ __forceinline__ void slab_quad_pruned(const float4 nx, const float4 fx, const float4 ny, const float4 fy, const float4 nz, const float4 fz,
                                                 const F3 o, const F3 inv, const float bound, float& t0, float& t1, float& t2, float& t3)
{
    const v2f nxa = (v2(nx.x, nx.y) - v2s(o.x)) * v2s(inv.x), nxb = (v2(nx.z, nx.w) - v2s(o.x)) * v2s(inv.x);
    const v2f nya = (v2(ny.x, ny.y) - v2s(o.y)) * v2s(inv.y), nyb = (v2(ny.z, ny.w) - v2s(o.y)) * v2s(inv.y);
    const v2f nza = (v2(nz.x, nz.y) - v2s(o.z)) * v2s(inv.z), nzb = (v2(nz.z, nz.w) - v2s(o.z)) * v2s(inv.z);
    const v2f fxa = (v2(fx.x, fx.y) - v2s(o.x)) * v2s(inv.x), fxb = (v2(fx.z, fx.w) - v2s(o.x)) * v2s(inv.x);
    const v2f fya = (v2(fy.x, fy.y) - v2s(o.y)) * v2s(inv.y), fyb = (v2(fy.z, fy.w) - v2s(o.y)) * v2s(inv.y);
    const v2f fza = (v2(fz.x, fz.y) - v2s(o.z)) * v2s(inv.z), fzb = (v2(fz.z, fz.w) - v2s(o.z)) * v2s(inv.z);
    const float e0 = fmax3(nxa.x, nya.x, nza.x), e1 = fmax3(nxa.y, nya.y, nza.y), e2 = fmax3(nxb.x, nyb.x, nzb.x), e3 = fmax3(nxb.y, nyb.y, nzb.y);
    const float x0 = fmin3(fxa.x, fya.x, fza.x), x1 = fmin3(fxa.y, fya.y, fza.y), x2 = fmin3(fxb.x, fyb.x, fzb.x), x3 = fmin3(fxb.y, fyb.y, fzb.y);
    const float inf = pinf();
    if (PRUNE) {
        t0 = ((e0 <= __builtin_fminf(x0 + CRT_EPSILON, bound)) & (x0 >= 0)) ? e0 : inf;
        t1 = ((e1 <= __builtin_fminf(x1 + CRT_EPSILON, bound)) & (x1 >= 0)) ? e1 : inf;
        t2 = ((e2 <= __builtin_fminf(x2 + CRT_EPSILON, bound)) & (x2 >= 0)) ? e2 : inf;
        t3 = ((e3 <= __builtin_fminf(x3 + CRT_EPSILON, bound)) & (x3 >= 0)) ? e3 : inf;
    } else { // no bound (CRT_TRAVERSAL_EXACT): hit_AABB's own test
        t0 = ((e0 <= x0 + CRT_EPSILON) & (x0 >= 0)) ? e0 : inf;
        t1 = ((e1 <= x1 + CRT_EPSILON) & (x1 >= 0)) ? e1 : inf;
        t2 = ((e2 <= x2 + CRT_EPSILON) & (x2 >= 0)) ? e2 : inf;
        t3 = ((e3 <= x3 + CRT_EPSILON) & (x3 >= 0)) ? e3 : inf;
    }
}

// (plane pair - o_axis) * inv_axis for two children at once, with the ray's scalar BROADCAST by the instruction's own operand selects
// (op_sel / op_sel_hi pick the low or the high half of a 64-bit register pair for both result halves): the reference's two roundings per
// plane (DeviceBVH.cuh:95-100), and no move that builds a (x, x) pair -- left to itself the compiler builds such pairs for half of these
// instructions (nine v_mov per visit in round 5, and the duplicates cost registers: the second visit's 1 / d went to scratch memory).
// HALF: 0 = the scalar is the pair's low half, 1 = its high half.
#define CRT_PK_SUBMUL_IMPL(OH, IH)                                                                                          \
    v2f d_;                                                                                                                 \
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0," #OH "] op_sel_hi:[1," #OH "] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d_) : "v"(p), "v"(o)); \
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0," #IH "] op_sel_hi:[1," #IH "]" : "=v"(d_) : "v"(d_), "v"(i));                    \
    return d_;
__device__ __forceinline__ v2f pk_submul_ll(const v2f p, const v2f o, const v2f i) { CRT_PK_SUBMUL_IMPL(0, 0) }
__device__ __forceinline__ v2f pk_submul_hh(const v2f p, const v2f o, const v2f i) { CRT_PK_SUBMUL_IMPL(1, 1) }
#undef CRT_PK_SUBMUL_IMPL
// (pair.H, pair.H) * x and (pair.H, pair.H) - x: a ray's scalar against two triangles' values, the scalar picked from an aligned pair by the
// instruction (tri_pair).  The products and differences are the plain IEEE ones of the scalar form.
template <int H>
__device__ __forceinline__ v2f pk_bmul(const v2f pair, const v2f x)
{
    v2f d_;
    if (H == 0) asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(d_) : "v"(pair), "v"(x));
    else asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(d_) : "v"(pair), "v"(x));
    return d_;
}
template <int H>
__device__ __forceinline__ v2f pk_bsub(const v2f pair, const v2f x)
{
    v2f d_;
    if (H == 0) asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d_) : "v"(pair), "v"(x));
    else asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d_) : "v"(pair), "v"(x));
    return d_;
}
// the ray as the 4-wide step keeps it: (o.x, o.y), (o.z, -) and (1/d.x, 1/d.y), (1/d.z, -) in aligned register pairs
struct RayPk { v2f oxy, oz, ixy, iz; };

// The same four boxes with the accept test of hit_AABB (DeviceBVH.cuh:121-125) handed back as predicates beside the entry distances:
// the decoupled step needs "hit" as a wave mask (its leaf-queue appends) and as a lane predicate, and the distance only to put the
// nearest inner child first -- a distance forced to +inf and compared with +inf again costs a select and a compare per child.
__device__ __forceinline__ void slab_quad_hits(const float4 nx, const float4 fx, const float4 ny, const float4 fy, const float4 nz, const float4 fz,
                                               const RayPk& R, float& e0, float& e1, float& e2, float& e3,
                                               unsigned long long& h0, unsigned long long& h1, unsigned long long& h2, unsigned long long& h3)
{
    const v2f nxa = pk_submul_ll(v2(nx.x, nx.y), R.oxy, R.ixy), nxb = pk_submul_ll(v2(nx.z, nx.w), R.oxy, R.ixy);
    const v2f nya = pk_submul_hh(v2(ny.x, ny.y), R.oxy, R.ixy), nyb = pk_submul_hh(v2(ny.z, ny.w), R.oxy, R.ixy);
    const v2f nza = pk_submul_ll(v2(nz.x, nz.y), R.oz, R.iz), nzb = pk_submul_ll(v2(nz.z, nz.w), R.oz, R.iz);
    const v2f fxa = pk_submul_ll(v2(fx.x, fx.y), R.oxy, R.ixy), fxb = pk_submul_ll(v2(fx.z, fx.w), R.oxy, R.ixy);
    const v2f fya = pk_submul_hh(v2(fy.x, fy.y), R.oxy, R.ixy), fyb = pk_submul_hh(v2(fy.z, fy.w), R.oxy, R.ixy);
    const v2f fza = pk_submul_ll(v2(fz.x, fz.y), R.oz, R.iz), fzb = pk_submul_ll(v2(fz.z, fz.w), R.oz, R.iz);
    e0 = fmax3(nxa.x, nya.x, nza.x); e1 = fmax3(nxa.y, nya.y, nza.y); e2 = fmax3(nxb.x, nyb.x, nzb.x); e3 = fmax3(nxb.y, nyb.y, nzb.y);
    const float x0 = fmin3(fxa.x, fya.x, fza.x), x1 = fmin3(fxa.y, fya.y, fza.y), x2 = fmin3(fxb.x, fyb.x, fzb.x), x3 = fmin3(fxb.y, fyb.y, fzb.y);
    // (a wave mask per child: the AND of the two compares' own results -- a ballot of their conjunction would cost a select and a compare)
    h0 = __builtin_amdgcn_ballot_w64(e0 <= x0 + CRT_EPSILON) & __builtin_amdgcn_ballot_w64(x0 >= 0);
    h1 = __builtin_amdgcn_ballot_w64(e1 <= x1 + CRT_EPSILON) & __builtin_amdgcn_ballot_w64(x1 >= 0);
    h2 = __builtin_amdgcn_ballot_w64(e2 <= x2 + CRT_EPSILON) & __builtin_amdgcn_ballot_w64(x2 >= 0);
    h3 = __builtin_amdgcn_ballot_w64(e3 <= x3 + CRT_EPSILON) & __builtin_amdgcn_ballot_w64(x3 >= 0);
}

// The two triangles of a leaf record at once: Moeller-Trumbore exactly as DeviceTriangle.cuh:39-56 + inside() :58-65 +
// the t > EPSILON filter of DeviceBVHNode::hit (DeviceBVH.cuh:37); lane .x = first triangle, .y = second.
// (round 6: o and d arrive as the aligned register pairs their LDS records are read into -- (o.x, o.y), (o.z, -), (d.x, d.y), (d.z, -) -- and the
// twelve instructions that take one of their components against both triangles pick it with operand selects: no (x, x) pair is built)
__device__ __forceinline__ void tri_pair(const float4 g0, const float4 g1, const float4 g2, const float4 g3, const float4 g4, const v2f oxy, const v2f oz,
                                         const v2f dxy, const v2f dz, bool& a0, bool& a1, float& t0, float& t1)
{
    const v2f v1x = v2(g0.x, g0.y), v1y = v2(g0.z, g0.w), v1z = v2(g1.x, g1.y);
    const v2f e1x = v2(g1.z, g1.w), e1y = v2(g2.x, g2.y), e1z = v2(g2.z, g2.w);
    const v2f e2x = v2(g3.x, g3.y), e2y = v2(g3.z, g3.w), e2z = v2(g4.x, g4.y);
    const v2f sx = pk_bsub<0>(oxy, v1x), sy = pk_bsub<1>(oxy, v1y), sz = pk_bsub<0>(oz, v1z);
    // s1 = d x e2, s2 = s x e1 (OrthoMethods.h:106-108)
    const v2f s1x = pk_bmul<1>(dxy, e2z) - pk_bmul<0>(dz, e2y), s1y = pk_bmul<0>(dz, e2x) - pk_bmul<0>(dxy, e2z), s1z = pk_bmul<0>(dxy, e2y) - pk_bmul<1>(dxy, e2x);
    const v2f s2x = sy * e1z - sz * e1y, s2y = sz * e1x - sx * e1z, s2z = sx * e1y - sy * e1x;
    const v2f det = s1x * e1x + (s1y * e1y + s1z * e1z);
    v2f rcp; // 1 / det (DeviceTriangle.cuh:47), see rcp_ieee
    {
        v2f r0;
        r0.x = __builtin_amdgcn_rcpf(det.x); r0.y = __builtin_amdgcn_rcpf(det.y);
        rcp = __builtin_elementwise_fma(__builtin_elementwise_fma(-det, r0, v2s(1.0f)), r0, r0);
        asm volatile("" : "+v"(rcp)); // (keeps the short form ahead of the branch instead of in an else-arm)
        const bool ok = rcp_short_ok2(det.x, det.y);
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(!ok) != 0, 0)) {
            if (!ok) { rcp.x = 1 / det.x; rcp.y = 1 / det.y; }
        }
    }
    const v2f beta = (s1x * sx + (s1y * sy + s1z * sz)) * rcp;
    const v2f gamma = (pk_bmul<0>(dxy, s2x) + (pk_bmul<1>(dxy, s2y) + pk_bmul<0>(dz, s2z))) * rcp;
    const v2f t = (s2x * e2x + (s2y * e2y + s2z * e2z)) * rcp;
    const v2f alpha = v2s(1.0f) - beta - gamma;
    // inside(): 0 < alpha, beta, gamma < 1, each comparison false for a NaN.  v_minimum3_f32 / v_maximum3_f32 (IEEE 754-2019
    // minimum / maximum) return NaN if any operand is one, so two comparisons on them are the same six (and -0 fails "0 <" either way).
    const float lo0 = __builtin_elementwise_minimum(__builtin_elementwise_minimum(alpha.x, beta.x), gamma.x);
    const float hi0 = __builtin_elementwise_maximum(__builtin_elementwise_maximum(alpha.x, beta.x), gamma.x);
    const float lo1 = __builtin_elementwise_minimum(__builtin_elementwise_minimum(alpha.y, beta.y), gamma.y);
    const float hi1 = __builtin_elementwise_maximum(__builtin_elementwise_maximum(alpha.y, beta.y), gamma.y);
    a0 = (0 < lo0) & (hi0 < 1) & (t.x > CRT_EPSILON);
    a1 = (0 < lo1) & (hi1 < 1) & (t.y > CRT_EPSILON);
    t0 = t.x; t1 = t.y;
}

// Pops the traversal stack of ray `id`; returns true when it is empty (the ray is finished).  An entry is the node ref alone:
// a node that has fallen behind the pruning bound since it was pushed is weeded out by its own step (keeping the entry
// distance to drop such entries here measured no gain on either scene and costs 4 B of LDS per level).  The LDS levels are
// read unconditionally and the (rare) spilled levels behind a wave-uniform branch: a per-lane choice between the two address
// spaces would compile to a flat load that waits on both memory pipes.
// LDS levels of a ray: all of them, except for a ray on the reference-arithmetic path in the 16-bit layout (none)
template <class LDS>
__device__ __forceinline__ int lds_levels(const bool exact) { return (LDS::R16 && exact) ? 0 : LDS::LV; }
template <class LDS>
__device__ __forceinline__ bool stack_pop(LDS& S, const MParams3& M3, const uint32_t id, const uint32_t g, int& sp, int& ref, const int lv)
{
    if (sp == 0) return true;
    sp--;
    int en = S.stk[sp < lv ? sp : 0][id];
    asm volatile("" : "+v"(en)); // (pins the LDS read: see above)
    if (__builtin_amdgcn_ballot_w64(sp >= lv)) {
        if (sp >= lv) en = M3.spill[(size_t)(sp - lv) * M3.M.spill_stride + g];
    }
    ref = en;
    return false;
}
// The same pop in two halves: the top LDS level is read when the step begins -- nothing a step pushes can land on it (pushes go to
// levels >= sp) -- so that its latency hides behind the node / leaf gather instead of standing alone at the end of the step.
template <class LDS>
__device__ __forceinline__ int stack_top_ahead(LDS& S, const uint32_t id, const int sp, const int lv)
{
    const int top = sp - 1;
    return S.stk[(top >= 0 && top < lv) ? top : 0][id];
}
template <class LDS>
__device__ __forceinline__ bool stack_pop_ahead(LDS& S, const MParams3& M3, const uint32_t id, const uint32_t g, int& sp, int& ref, const int top, const int lv)
{
    if (sp == 0) return true;
    sp--;
    int en = top;
    if (__builtin_amdgcn_ballot_w64(sp >= lv)) {
        if (sp >= lv) en = M3.spill[(size_t)(sp - lv) * M3.M.spill_stride + g];
    }
    ref = en;
    return false;
}
template <class LDS>
__device__ __forceinline__ void stack_push(LDS& S, const MParams3& M3, const uint32_t id, const uint32_t g, int& sp, const int ref, const int lv)
{
    if (sp < lv) S.stk[sp][id] = (typename LDS::stk_t)ref;
    if (__builtin_amdgcn_ballot_w64(sp >= lv)) {
        if (sp >= lv) M3.spill[(size_t)(sp - lv) * M3.M.spill_stride + g] = ref;
    }
    sp++;
}

// One step at a node of the 4-wide tree (rays with finite operands, CRT_TRAVERSAL_FAST / _EXACT): the four child boxes from their
// near and far planes (picked by the sign of the direction, the reference's own swap), the nearest hit child next, the others pushed
// farthest first.  Which children are visited, and in which order, does not change the result (crt_trace.h); the boxes and the test
// are the reference's (hit_AABB, exact for finite operands), so a leaf is entered iff its own box passes -- as in the 2-wide tree.
template <bool STATS, bool SORT = true, class LDS = Pool3Lds, bool DIR = false>
__device__ __forceinline__ bool inner4_step(const DevScene& sc, LDS& S, const MParams3& M, const uint32_t id, const uint32_t g, const F3 o, const F3 inv_or_d,
                                            const float bound, int& ref, int& sp, TravCounters& tc, uint32_t& max_sp
                                            )
{
    const char* nb = (const char*)sc.nodes4; // 32-bit byte offsets: scalar base + vector offset addressing
    const uint32_t noff = (uint32_t)ref * 128u;
    // (DIR: the argument is the direction itself -- 1 / d has d's sign -- and the reciprocals are formed after the loads are on their way)
    const uint32_t ox = noff + ((__float_as_uint(inv_or_d.x) >> 27) & 16u), oy = noff + ((__float_as_uint(inv_or_d.y) >> 27) & 16u),
                   oz = noff + ((__float_as_uint(inv_or_d.z) >> 27) & 16u); // + 16: the ray runs towards -axis, its near plane is hi
    const float4 a0 = *(const float4*)(nb + ox), a1 = *(const float4*)(nb + (ox ^ 16u));
    const float4 a2 = *(const float4*)((nb + oy) + 32), b0 = *(const float4*)((nb + (oy ^ 16u)) + 32);
    const float4 b1 = *(const float4*)((nb + oz) + 64), b2 = *(const float4*)((nb + (oz ^ 16u)) + 64);
    const float4 rf = *(const float4*)((nb + noff) + 96);
    const int top = stack_top_ahead(S, id, sp, LDS::LV);
    if (STATS) tc.inner++;
    const F3 inv = DIR ? inv3_exact(inv_or_d) : inv_or_d;
    float t0, t1, t2, t3; // entry distances; +inf = missed or beyond the pruning bound (sorts last)
    slab_quad_pruned<SORT>(a0, a1, a2, b0, b1, b2, o, inv, bound, t0, t1, t2, t3); // (SORT == pruning mode: CRT_SORT4)
    // (all four entry distances exist before the exchanges and pushes begin: left alone the compiler starts pushing the first pair's
    // loser while the second pair's boxes are still being computed, splits the arithmetic over two blocks and rebuilds the
    // broadcast operand pairs of the packed instructions in the second one -- nine extra moves per step)
    asm volatile("" : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3));
    const float inf = pinf();
    int r0 = __float_as_int(rf.x), r1 = __float_as_int(rf.y), r2 = __float_as_int(rf.z), r3 = __float_as_int(rf.w);
    // ascending by entry distance: (0,1)(2,3)(0,2)(1,3)(1,2)
#define CRT_CE(ta, ra, tb, rb) { const bool sw_ = tb < ta; const float tt_ = sw_ ? tb : ta; tb = sw_ ? ta : tb; ta = tt_; const int rr_ = sw_ ? rb : ra; rb = sw_ ? ra : rb; ra = rr_; }
    // (leaving out the last exchange -- nearest first, farthest last, the middle two as they come -- saves 5 instructions per step
    // and costs more visits than that: C2 +1.5 %, veach-mis -0.3 %)
    if (SORT) {
        CRT_CE(t0, r0, t1, r1) CRT_CE(t2, r2, t3, r3) CRT_CE(t0, r0, t2, r2) CRT_CE(t1, r1, t3, r3) CRT_CE(t1, r1, t2, r2)
    } else { // (CRT_TRAVERSAL_EXACT: only the nearest child to the front, CRT_SORT4)
        CRT_CE(t0, r0, t1, r1) CRT_CE(t2, r2, t3, r3) CRT_CE(t0, r0, t2, r2)
    }
#undef CRT_CE
    // the children to visit are a prefix of the sorted four (the pushes below do not rely on that); all but the nearest go on the
    // stack, farthest first
    const bool c0 = t0 < inf, c1 = t1 < inf, c2 = t2 < inf, c3 = t3 < inf;
    const int l3 = sp, l2 = l3 + (c3 ? 1 : 0), l1 = l2 + (c2 ? 1 : 0);
    constexpr int LV = LDS::LV; // (a ray of this step is not on the reference-arithmetic path: all LDS levels are its own)
    typedef typename LDS::stk_t stk_t;
    if (c3 & (l3 < LV)) S.stk[l3][id] = (stk_t)r3;
    if (c2 & (l2 < LV)) S.stk[l2][id] = (stk_t)r2;
    if (c1 & (l1 < LV)) S.stk[l1][id] = (stk_t)r1;
    const int sp_new = l1 + (c1 ? 1 : 0);
    if (__builtin_amdgcn_ballot_w64((sp_new > l3) & (sp_new > LV))) { // one check per step for the levels beyond LDS (sp_new - 1 is the highest written)
        if (c3 & (l3 >= LV)) M.spill[(size_t)(l3 - LV) * M.M.spill_stride + g] = r3;
        if (c2 & (l2 >= LV)) M.spill[(size_t)(l2 - LV) * M.M.spill_stride + g] = r2;
        if (c1 & (l1 >= LV)) M.spill[(size_t)(l1 - LV) * M.M.spill_stride + g] = r1;
    }
    sp = sp_new;
    if (STATS && (uint32_t)sp > max_sp) max_sp = (uint32_t)sp;
    if (c0) { ref = r0; return false; }
    return stack_pop_ahead(S, M, id, g, sp, ref, top, LV); // (no child was hit: nothing was pushed, the top is the one read above)
}

// The same step with the leaves decoupled (DEC): a leaf child that is hit becomes an entry (ray | leaf record << 8) of the wave's leaf
// queue -- appended right here, child by child, at tail + number of lanes below with an entry of the same child; `added` counts the
// entries of the batch -- and takes no part in the ordering; among the inner children the nearest is next, the others are pushed.
// Row [7] of a node holds the refs in the form this needs: an inner child as in row [6], a leaf child as 0x80000000 | record << 8.
// The ray's count of entries in flight (word D) grows by LDS atomics, one per entry.  CRT_TRAVERSAL_EXACT only (no bound).
// Runs inside the divergent region of the batch's lanes: ballots see those lanes only; lq_t is the queue's tail before the batch.
template <class LDS>
__device__ __forceinline__ void leafq_push(LDS& S, const uint32_t id, const bool hit, const unsigned long long m, const uint32_t entry, const uint32_t lq_t, uint32_t& added)
{
    // m = the ballot of `hit`, formed by the caller from the ballots of its compares (a ballot of their conjunction would cost a
    // select and another compare)
    if (m) {
        const uint32_t slot = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, lq_t + added));
        if (hit) {
            S.leafq[slot & (uint32_t)(LEAFQ_CAP - 1)] = entry;
            __hip_atomic_fetch_add(&S.D[id], 1u << RD_PEND_SHIFT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        added += (uint32_t)__popcll(m);
    }
}
// The same append without a branch, for the four children of a 4-wide step (m is empty for one child in five, and a taken branch costs
// more than the eight instructions it skips): a lane without an entry writes to the spare dword behind the queue, and the ray's count
// of entries in flight is the caller's (one addition for the four children).
template <class LDS>
__device__ __forceinline__ void leafq_push_all(LDS& S, const bool hit, const unsigned long long m, const uint32_t entry, uint32_t& tail)
{
    const uint32_t slot = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, tail));
    S.leafq[hit ? (slot & (uint32_t)(LEAFQ_CAP - 1)) : (uint32_t)LEAFQ_CAP] = entry;
    tail += (uint32_t)__popcll(m);
}
// ---- wave masks (round 6) ----
// A predicate of the traversal steps lives as a WAVE MASK in a scalar register pair from the compare that makes it to the select, store or
// count that uses it: a ballot of a compare is the compare's own result, conjunctions / disjunctions / counts are scalar instructions, and
// `lanes` hands a mask back to the vector unit as it stands.  The compiler's own treatment of a bool that crosses a join or is combined
// before a ballot is a trip through a vector register (v_cndmask 0 / 1, v_cmp_ne: 3 % of the kernel's vector instructions in round 5).
typedef unsigned long long wmask;
__device__ __forceinline__ wmask bal(const bool b) { return __builtin_amdgcn_ballot_w64(b); }
__device__ __forceinline__ bool lanes(const wmask m) { return __builtin_amdgcn_inverse_ballot_w64(m); }
// x + 1 / x - 1 in the lanes of m: the mask rides in as the carry (one instruction; the compiler's form is a select and an addition)
__device__ __forceinline__ int add_mask(const int x, const wmask m)
{
    int r;
    wmask co;
    asm("v_addc_co_u32_e64 %0, %1, 0, %2, %3" : "=v"(r), "=s"(co) : "v"(x), "s"(m));
    return r;
}
__device__ __forceinline__ int sub_mask(const int x, const wmask m)
{
    int r;
    wmask co;
    asm("v_subbrev_co_u32_e64 %0, %1, 0, %2, %3" : "=v"(r), "=s"(co) : "v"(x), "s"(m));
    return r;
}

// A visit of the decoupled inner step, in three parts so that the caller can put the SECOND visit's loads in front of the first visit's
// stores (round 6: what a wave waits for in this step is the node's round trip, ~500 cycles; the first visit's appends and pushes -- some
// eighty instructions that touch LDS only -- now run while the second visit's node is on its way):
//   node_load    the six (seven) rows of the node
//   visit_front  boxes, accept masks, the queue-capacity check, entry counts, the nearest inner child, the new node and depth -- registers only
//   visit_back   the leaf-queue entries, the stack pushes (LDS; spilled levels: global memory)
struct Visit4 {
    float4 a0, a1, a2, b0, b1, b2, rf;     // near / far rows of x, y, z; the row of refs (not IMPL)
    wmask m0, m1, m2, m3;                  // leaf children that are hit
    wmask pd, pg, pb;                      // inner children that are hit and pushed: the loser of (2,3), of the final, of (0,1)
    uint32_t q0, q1, q2, q3, tail;         // queue entries (record << 8) and where the visit's first one goes
    int rd, rg, rb, l3, l2, l1, sp_new;    // the pushed refs, their levels, the depth after the pushes
};
template <bool IMPL>
__device__ __forceinline__ void node_load(const DevScene& sc, const int ref, const wmask EN, const F3 dir, Visit4& V)
{
    // IMPL: the copy of the tree without its rows of refs (crt_render.hip "nodes4i": 96 B per node, SIX loads per visit instead of seven);
    // the children's refs and the leaves' records are implied.  A lane outside EN loads the EMPTY node.
    const char* nb = (const char*)(IMPL ? sc.nodes4i : sc.nodes4);
    const uint32_t noff = lanes(EN) ? (uint32_t)ref * (IMPL ? (uint32_t)(NODE4I_F4 * 16) : 128u) : (IMPL ? sc.empty4i_off : sc.empty4_off);
    // (the direction itself picks the planes -- 1 / d has d's sign -- and the reciprocals are formed after the loads are on their way)
    const uint32_t ox = noff + ((__float_as_uint(dir.x) >> 27) & 16u), oy = noff + ((__float_as_uint(dir.y) >> 27) & 16u),
                   oz = noff + ((__float_as_uint(dir.z) >> 27) & 16u);
    V.a0 = *(const float4*)(nb + ox); V.a1 = *(const float4*)(nb + (ox ^ 16u));
    V.a2 = *(const float4*)((nb + oy) + 32); V.b0 = *(const float4*)((nb + (oy ^ 16u)) + 32);
    V.b1 = *(const float4*)((nb + oz) + 64); V.b2 = *(const float4*)((nb + (oz ^ 16u)) + 64);
    V.rf = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (!IMPL) V.rf = *(const float4*)((nb + noff) + 112);
}
// CHECK: what happens when the queue cannot take the visit's entries (they are counted before anything is written).  0: cannot happen;
// 1: the lanes from `cap_left / 4` on are taken out of the visit (*voided: they keep their state and are queued again); 2: the whole
// visit is dropped (*bailed; nothing has changed).
// EN: the lanes that take the visit.  A lane outside it -- a lane without a ray, an any-hit ray that has its answer, a ray of the
// reference-arithmetic path, a voided lane -- is at the EMPTY node and keeps its node and depth: it hits nothing, appends nothing,
// pushes nothing and does not pop.  Returns the lanes whose walk is over (a subset of EN); n_leaf and any_leaf accumulate.
// `top`: the stack's top level as it is when the visit begins (read after the visit before it has pushed).
template <bool STATS, class LDS, int CHECK = 0, bool IMPL = false>
__device__ __forceinline__ wmask visit_front(const DevScene& sc, const MParams3& M, const uint32_t g, const F3 dir, RayPk& R, Visit4& V, const int top,
                                             int& ref, int& sp, TravCounters& tc, uint32_t& max_sp, const uint32_t lq_t, uint32_t& added, wmask& any_leaf,
                                             int& n_leaf, const wmask EN, const uint32_t cap_left = 0, bool* bailed = nullptr, wmask* voided = nullptr)
{
    if (CHECK != 2) { // (the second visit of a step takes the first one's 1 / d: the same value, and a ballot of a predicate of another block is a trip through a vector register)
        const F3 inv = inv3_exact(dir);
        R.ixy = v2(inv.x, inv.y); R.iz.x = inv.z;
    }
#ifdef CRT_X_EXTRA_VALU /* sensitivity experiment (round 6): N more vector instructions per visit (independent v_add_f32 on a scratch register) */
    {
        float xv_ = dir.x;
        for (int k_ = 0; k_ < CRT_X_EXTRA_VALU; k_++) asm volatile("v_add_f32 %0, %0, %1" : "+v"(xv_) : "v"(dir.y));
        asm volatile("" :: "v"(xv_));
    }
#endif
    float t0, t1, t2, t3;
    int r0, r1, r2, r3;            // the children's refs (an inner child: its node)
    wmask N0, N1, N2, N3;          // the child is a leaf (or an empty slot, which is never hit)
    wmask H0, H1, H2, H3;
    slab_quad_hits(V.a0, V.a1, V.a2, V.b0, V.b1, V.b2, R, t0, t1, t2, t3, H0, H1, H2, H3);
    asm volatile("" : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3));
    if constexpr (IMPL) {
        // 36 bits in the low 12 mantissa bits of child 0's three NEAR planes (the same bits in the lo and the hi plane of an axis): the first
        // mixed child fm (15), the first fringe child ff (15), the numbers of mixed and of fringe children (3 + 3).  A fringe node -- numbered
        // from n_mixed4i on -- has leaves only and no such bits.  Inner children come first: mixed, then fringe; leaf child k is record 4 n + k.
        const uint32_t cx = __float_as_uint(V.a0.x) & 0xfffu, cy = __float_as_uint(V.a2.x) & 0xfffu, cz = __float_as_uint(V.b1.x) & 0xfffu;
        const bool fr = (uint32_t)ref >= sc.n_mixed4i;
        const uint32_t fm = cx | ((cy & 7u) << 12), ff = (cy >> 3) | ((cz & 63u) << 9);
        const uint32_t cm = fr ? 0u : (cz >> 6) & 7u, ci = fr ? 0u : ((cz >> 6) & 7u) + (cz >> 9);
        N0 = bal(ci == 0u); N1 = bal(ci <= 1u); N2 = bal(ci <= 2u); N3 = bal(ci <= 3u);
        const uint32_t ffm = ff - cm;
        r0 = (int)(cm > 0u ? fm : ffm); r1 = (int)(cm > 1u ? fm + 1u : ffm + 1u); r2 = (int)(cm > 2u ? fm + 2u : ffm + 2u); r3 = (int)(cm > 3u ? fm + 3u : ffm + 3u);
        V.q0 = (uint32_t)ref << 10; V.q1 = V.q0 + 0x100u; V.q2 = V.q0 + 0x200u; V.q3 = V.q0 + 0x300u;
    } else {
        r0 = __float_as_int(V.rf.x); r1 = __float_as_int(V.rf.y); r2 = __float_as_int(V.rf.z); r3 = __float_as_int(V.rf.w);
        N0 = bal(r0 < 0); N1 = bal(r1 < 0); N2 = bal(r2 < 0); N3 = bal(r3 < 0);
        V.q0 = (uint32_t)r0 & 0x7fffff00u; V.q1 = (uint32_t)r1 & 0x7fffff00u; V.q2 = (uint32_t)r2 & 0x7fffff00u; V.q3 = (uint32_t)r3 & 0x7fffff00u;
    }
    wmask m0 = H0 & N0, m1 = H1 & N1, m2 = H2 & N2, m3 = H3 & N3;         // leaf children that are hit
    wmask i0 = H0 & ~N0, i1 = H1 & ~N1, i2 = H2 & ~N2, i3 = H3 & ~N3;     // inner children that are hit
    if (CHECK != 0) {
        if ((uint32_t)(__popcll(m0) + __popcll(m1) + __popcll(m2) + __popcll(m3)) > cap_left) {
            if (CHECK == 2) { *bailed = true; return 0ull; }
            // the lanes a quarter of the free entries has room for stay (lane numbers: the batch's lanes are 0 .. take - 1)
            const wmask km = bal((uint32_t)(threadIdx.x & 63) < (cap_left >> 2));
            m0 &= km; m1 &= km; m2 &= km; m3 &= km;
            i0 &= km; i1 &= km; i2 &= km; i3 &= km;
            *voided = ~km;
        }
    }
    const wmask EFF = CHECK == 1 ? EN & ~*voided : EN; // the lanes whose visit counts
    if (STATS && lanes(EFF)) tc.inner++;
    V.m0 = m0; V.m1 = m1; V.m2 = m2; V.m3 = m3;
    V.tail = lq_t + added;
    added += (uint32_t)(__popcll(m0) + __popcll(m1) + __popcll(m2) + __popcll(m3));
    n_leaf = add_mask(add_mask(add_mask(add_mask(n_leaf, m0), m1), m2), m3);
    any_leaf |= (m0 | m1) | (m2 | m3);
    // The nearest inner child that is hit goes to the front, by a tournament (0,1)(2,3)(winners) on (hit, distance): b beats a iff b is hit and
    // (a is not, or b is nearer).  Only the winner's distance is ever compared again, and "is hit" travels as a mask -- front = a | b, back = a & b --
    // so an exchange is a compare, a select of the winner's distance and two selects of the refs (round 5: distances forced to +inf for the
    // children that are not inner hits, five selects per exchange and a compare with +inf per child afterwards: 23 vector instructions, now 11).
    // The order of the visits is the one of round 5: the result cannot depend on it (crt_trace.h), the any-hit rays' visit counts do.
    const wmask S01 = i1 & (bal(t1 < t0) | ~i0), S23 = i3 & (bal(t3 < t2) | ~i2);
    const float tA = lanes(S01) ? t1 : t0, tC = lanes(S23) ? t3 : t2;
    const int rA = lanes(S01) ? r1 : r0, rC = lanes(S23) ? r3 : r2;
    V.rb = lanes(S01) ? r0 : r1; V.rd = lanes(S23) ? r2 : r3;
    const wmask IA = i0 | i1, IC = i2 | i3;
    V.pb = i0 & i1; V.pd = i2 & i3;
    const wmask S02 = IC & (bal(tC < tA) | ~IA);
    const int rF = lanes(S02) ? rC : rA;
    V.rg = lanes(S02) ? rA : rC;
    const wmask IF = IA | IC;
    V.pg = IA & IC;
    // pushed: the loser of (2,3), then the loser of the final, then the loser of (0,1) -- which is popped first
    V.l3 = sp; V.l2 = add_mask(V.l3, V.pd); V.l1 = add_mask(V.l2, V.pg); V.sp_new = add_mask(V.l1, V.pb);
    constexpr int LV = LDS::LV;
    if (STATS && (uint32_t)V.sp_new > max_sp) max_sp = (uint32_t)V.sp_new;
    // the nearest inner child next; without one (nothing was pushed either: the top is the one read when the visit began) the stack's top, or the end
    const wmask NF = EFF & ~IF;
    const wmask POP = NF & bal(V.sp_new > 0);
    const wmask OVER = NF & ~POP;
    const int nref = lanes(IF) ? rF : top;
    ref = lanes(IF | POP) ? nref : ref;
    sp = sub_mask(V.sp_new, POP);
    const wmask DEEP = POP & bal(sp >= LV); // (of the new depth -- the carry instruction's own result: the compiler shares a compare of sp_new with the push block's and sends it through a vector register)
    if (DEEP) {
        if (lanes(DEEP)) ref = M.spill[(size_t)(sp - LV) * M.M.spill_stride + g];
    }
    return OVER;
}
template <class LDS>
__device__ __forceinline__ void visit_back(LDS& S, const MParams3& M, const uint32_t id, const uint32_t g, const Visit4& V)
{
    uint32_t tail = V.tail;
    leafq_push_all(S, lanes(V.m0), V.m0, V.q0 | id, tail);
    leafq_push_all(S, lanes(V.m1), V.m1, V.q1 | id, tail);
    leafq_push_all(S, lanes(V.m2), V.m2, V.q2 | id, tail);
    leafq_push_all(S, lanes(V.m3), V.m3, V.q3 | id, tail);
    constexpr int LV = LDS::LV;
    typedef typename LDS::stk_t stk_t;
    if (lanes(V.pd & bal(V.l3 < LV))) S.stk[V.l3][id] = (stk_t)V.rd;
    if (lanes(V.pg & bal(V.l2 < LV))) S.stk[V.l2][id] = (stk_t)V.rg;
    if (lanes(V.pb & bal(V.l1 < LV))) S.stk[V.l1][id] = (stk_t)V.rb;
    if (bal((V.sp_new > V.l3) & (V.sp_new > LV))) {
        if (lanes(V.pd & bal(V.l3 >= LV))) M.spill[(size_t)(V.l3 - LV) * M.M.spill_stride + g] = V.rd;
        if (lanes(V.pg & bal(V.l2 >= LV))) M.spill[(size_t)(V.l2 - LV) * M.M.spill_stride + g] = V.rg;
        if (lanes(V.pb & bal(V.l1 >= LV))) M.spill[(size_t)(V.l1 - LV) * M.M.spill_stride + g] = V.rb;
    }
}

// One step at a node of a 2-wide tree: the reference topology (CRT_TRAVERSAL_REFERENCE: reference box arithmetic, reference
// visit order, no pruning) or, for the handful of FAST rays with non-finite operands, reference arithmetic on that topology
// with ordering and pruning.  d = direction (the sign selects the near plane, DeviceBVH.cuh:101-119).
template <int MODE, bool STATS, class LDS>
__device__ __forceinline__ bool inner2_step(const DevScene& sc, LDS& S, const MParams3& M, const uint32_t id, const uint32_t g, const F3 o, const F3 inv,
                                            const F3 d, const float bound, int& ref, int& sp, TravCounters& tc, uint32_t& max_sp)
{
    const float4* nd = sc.nodes3 + (size_t)ref * 4;
    const float4 n0 = nd[0], n1 = nd[1], n2 = nd[2];
    const float2 n3 = *(const float2*)(nd + 3);
    if (STATS) tc.inner++;
    bool hl, hr;
    float tl, tr;
    slab_pair(n0, n1, n2, o, inv, d, true, hl, hr, tl, tr);
    const int lref = __float_as_int(n3.x), rref = __float_as_int(n3.y);
    bool left_first;
    if (MODE == 1) {
        left_first = false; // push lc, visit rc first (DeviceBVH.cuh:154-166)
    } else {
        hl = hl && !(tl > bound);
        hr = hr && !(tr > bound);
        left_first = tl <= tr;
    }
    const bool both = hl && hr, any = hl || hr;
    const int near_ref = both ? (left_first ? lref : rref) : (hl ? lref : rref);
    const int lv = lds_levels<LDS>(true); // (the rays of this step are on the reference-arithmetic path)
    if (both) {
        stack_push(S, M, id, g, sp, left_first ? rref : lref, lv);
        if (STATS && (uint32_t)sp > max_sp) max_sp = (uint32_t)sp;
    }
    if (any) { ref = near_ref; return false; }
    return stack_pop(S, M, id, g, sp, ref, lv);
}

// CRT_TRAVERSAL_EXACT visits every child that is hit whatever the order (no bound shrinks): it only brings the nearest to the front
// (three exchanges instead of five: what the any-hit rays gain from a full order is less than the two exchanges cost -- C2 -0.7 %,
// veach-mis -0.6 %); -DCRT_EXACT_FULLSORT restores the full network
#define CRT_SORT4(mode) ((mode) != 2)
// ALL: every next-event sample is traced (CRT_FLAG_TRACE_ALL) -- its own instantiation, so that profiles of the default path
// are not mixed with it
// RING: the commit ring (in-order sum of the samples inside the launch, see ring_publish) -- its own instantiations: the kernels
// without it are, instruction for instruction, what they were before it existed
template <int MODE, bool STATS, bool ALL = false, bool QUERY = false, bool R16 = false, bool RING = false, bool DEC = false, bool IMPL = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(CRT_WAVES, CRT_WAVES))) void k_mega3(const MParams3 M3)
{
    static_assert(!IMPL || (DEC && R16), "the tree without its rows of refs: decoupled leaves, 16-bit stack entries");
    static_assert(!(RING && (QUERY || STATS)), "the commit ring is a render without counters");
    static_assert(!(R16 && MODE == 1), "CRT_TRAVERSAL_REFERENCE walks the 2-wide trees: 32-bit stack entries");
    static_assert(!DEC || MODE == 2, "decoupled leaves: CRT_TRAVERSAL_EXACT");
    typedef typename std::conditional<DEC, Pool4LdsT<R16, RING>, Pool3LdsT<R16>>::type LDS3;
    __shared__ LDS3 S;
    constexpr int QCAP = LDS3::QCAP;
    const MParams& M = M3.M;
    const LParams& P = M.P;
    const DevScene& sc = P.sc; // (one copy of the scene pointers in scalar registers: the logic phases use P.sc too)
    const Pool& pl = P.pool;
    const int lane = threadIdx.x;
    const uint32_t base = blockIdx.x * (uint32_t)LDS3::P; // first global slot of this wave's pool
    Tables<false> tb;
    tb.mats = sc.mats; tb.lights = sc.lights;

    PathCounters cnt;
    cnt = PathCounters{};
    TravCounters tc;
    tc.inner = tc.leaf = tc.tests = tc.hits = 0;
    uint32_t max_sp = 0;
    int n_exact = 0; // rays on the reference-arithmetic path (RF_EXACT) that are in the traversal phases of this pool

    // ring state: wave-uniform scalars
    int qn[PH3_N], qh[PH3_N], qt[PH3_N]; // entries, head, tail (head and tail in [0, QCAP))
#pragma unroll
    for (int p = 0; p < PH3_N; p++) { qn[p] = 0; qh[p] = 0; qt[p] = 0; }
    uint32_t dg_b[5] = {0, 0, 0, 0, 0}, dg_l[5] = {0, 0, 0, 0, 0}; // STATS: batches and rays per phase
    uint32_t dg_sp[6] = {0, 0, 0, 0, 0, 0};                         // STATS, per lane: inner steps that leave the stack deeper than 1 .. 6 entries
    uint32_t dg_ov[2] = {0, 0};                                     // STATS, per lane (DEC): visits taken back because the leaf queue was full (first visit: voided lanes; second visit: bailed, counted by its lanes)
    uint32_t lq_h = 0, lq_t = 0; // DEC: the leaf queue's head and tail, free-running (entries = tail - head, index = counter mod LEAFQ_CAP)
    constexpr bool commit_ring = RING;
    // -DCRT_STAMPS (a diagnostic build, tools/ab_build.sh): the wave's cycles by phase -- s_memtime at the end of every step, the difference to
    // the stamp before it booked on the step that ended (the scheduler's share on "other") -- summed into crt_stats.phase_cycles:
    // [0] LA [1] leaf step [2] inner step [3] scheduler / rest [4] LB [5] LC, and the steps of each kind in [6..11] (same order)
#ifdef CRT_STAMPS
    unsigned long long st_acc[6] = {0, 0, 0, 0, 0, 0};
    uint32_t st_n[6] = {0, 0, 0, 0, 0, 0};
    unsigned long long st_prev = __builtin_amdgcn_s_memtime();
#define STAMP(k_) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_acc[k_] += t_ - st_prev; st_prev = t_; st_n[k_]++; }
#define STAMP_OTHER() { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_acc[3] += t_ - st_prev; st_prev = t_; }
#else
#define STAMP(k_)
#define STAMP_OTHER()
#endif
#ifdef CRT_HANDOFF_INV /* experiment builds only: what a dispatch's agent-scope acquire does, by hand (vector L1 and the non-local lines of L2) */
    asm volatile("buffer_inv sc1" ::: "memory");
#endif
#ifdef CRT_HANDOFF_INV_SYS
    asm volatile("buffer_inv sc0 sc1" ::: "memory");
#endif
    if (commit_ring && lane == 0) S.waitq = 0u;
    // every ray of the pool starts in LC with a path in stage NEW
    {
        // (readfirstlane: the value is wave-uniform, and left to itself the compiler may compute it -- and the LC ring's count that starts
        // from it -- in a vector register, which the scalar maxima of the scheduler cannot take)
        const int n_valid = __builtin_amdgcn_readfirstlane((int)min((uint32_t)LDS3::P, pl.n > base ? pl.n - base : 0u));
        for (int i = lane; i < n_valid; i += 64) {
            S.rq(PH3_LC)[i] = (uint8_t)i;
            pl.la[base + i] = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float((uint32_t)ST_NEW << 8));
        }
        qn[PH3_LC] = n_valid;
        qt[PH3_LC] = n_valid >= QCAP ? n_valid - QCAP : n_valid;
        (void)qh; (void)qt;
    }

// appends the processed rays (lane active = `on`, ray `id`) to the ring of their new phase
#define PUSH3()                                                                                                            \
    _Pragma("unroll") for (int p = 0; p < PH3_N; p++) {                                                                    \
        const bool mine = nph == (uint32_t)p; /* lanes without a ray carry PH3_NONE */                                                                        \
        const unsigned long long m = __ballot(mine);                                                                       \
        if (m) {                                                                                                           \
            /* slot = tail + number of lanes below this one that go the same way: the tail rides in as mbcnt's addend */      \
            const uint32_t slot = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, RQ_PUSH_BASE(p))); \
            if (mine) S.rq(p)[ring_wrap<QCAP>(slot)] = (uint8_t)id;                                                            \
            const int add = (int)__popcll(m);                                                                              \
            RQ_PUSH_ADV(p, add)                                                                                            \
        }                                                                                                                  \
    }
// after a traversal step: a ray goes on to an inner node or a leaf, or it is finished -- only then (one wave-uniform test
// for the three logic rings together) is its route worked out from the flag bits of its record
#define PUSH_TRAV() { if (on) nph = t_done ? route_done<QUERY>(t_flags) : (t_ref >= 0 ? PH3_INNER : PH3_LEAF); PUSH3() }
// takes the (up to) 64 oldest rays of ring p
#define POP3(p)                                                                                                            \
    const int take = min(64, qn[p]);                                                                                       \
    if (STATS) { dg_b[p]++; dg_l[p] += (uint32_t)take; }                                                                   \
    const bool on = lane < take;                                                                                           \
    const uint32_t id = S.rq(p)[ring_wrap<QCAP>(RQ_POP_BASE(p, take) + (uint32_t)lane)];                                       \
    RQ_POP_ADV(p, take)                                                                                                    \
    const uint32_t g = base + id;                                                                                          \
    uint32_t nph = PH3_NONE;

// The logic phases read their parameters from the kernel-argument segment again, through a pointer the compiler cannot see
// through: parameters that are only needed there (camera, tiling, work-item cursors, pool planes ...) would otherwise be
// hoisted into scalar registers for the whole kernel and push the ring cursors of the traversal steps out into VGPR lanes.
#define LOGIC_PARAMS()                                                                                                     \
    const __attribute__((address_space(4))) char* ka_ = (const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr(); \
    asm volatile("" : "+s"(ka_));                                                                                          \
    union { LParams p; uint32_t w[sizeof(LParams) / 4]; } pu_;                                                             \
    {                                                                                                                      \
        const __attribute__((address_space(4))) uint32_t* src_ =                                                           \
            (const __attribute__((address_space(4))) uint32_t*)(ka_ + offsetof(MParams3, M) + offsetof(MParams, P));       \
        _Pragma("unroll") for (unsigned i_ = 0; i_ < sizeof(LParams) / 4; i_++) pu_.w[i_] = src_[i_];                      \
    }                                                                                                                      \
    const LParams& Pl = pu_.p;                                                                                             \
    Tables<false> tl;                                                                                                      \
    tl.mats = Pl.sc.mats; tl.lights = Pl.sc.lights;

    for (;;) {
      int act;
      // The traversal steps -- nine of ten iterations -- are a loop of their own inside the scheduler loop: the register allocator
      // weighs a value by the depth of the loops that use it, and with all five phases at one depth it kept the (cold) inner loops of
      // the logic phases in registers and spilled the ring cursors and node pointers of the traversal steps (19 vector instructions
      // of spill / copy code per iteration: C2 106.6 -> 99.0 ms, veach-mis 99.4 -> 92.4 ms at spp 256).
      for (;;) {
        // ---- choose a phase: the ring with the fullest batch; among equals the logic phases first (they feed the traversal), then
        //      leaves, then inner nodes ----
        {
            // key = batch size * 8 + phase number (the phase numbers are the tie-break order)
            const int kC = min(qn[PH3_LC], 64) * 8 + PH3_LC, kA = min(qn[PH3_LA], 64) * 8 + PH3_LA, kB = min(qn[PH3_LB], 64) * 8 + PH3_LB;
            // (DEC: the leaf queue counts entries, not rays; with 64 or more it is the fullest there can be and wins over the inner ring)
#ifndef LEAFQ_FIRST
#define LEAFQ_FIRST 48 /* DEC: with this many entries the leaf queue goes before everything else: the queue is emptied early and stays far from full
                          (when inner batches were still cut to a quarter of its free entries: C2 93.9 -> 92.1 ms, veach-mis spp 256 90.6 -> 89.4;
                          40 / 56 / 32: 91.9 / 92.9 / 92.0 and 89.6 / 89.5 / 90.8; with full batches 32 / 40 / 56 against 48: within 0.5 %) */
#endif
            // (an inner step starts with fewer than LEAFQ_FIRST entries in the queue -- from that many on the leaf step goes first, here and
            // in the alternating loop -- and must find room for a reference-arithmetic batch, 64 entries appended outside the counted
            // ones, plus four entries, the least a 4-wide visit needs to keep one lane: otherwise it would take every lane back, for ever)
            static_assert(LEAFQ_FIRST >= 1 && LEAFQ_FIRST + 64 + 4 <= LEAFQ_CAP, "leaf queue: LEAFQ_FIRST - 1 entries + a reference-arithmetic batch + one lane's four entries must fit LEAFQ_CAP");
            const int nl_ = DEC ? (int)(lq_t - lq_h) : qn[PH3_LEAF];
            const int kL = ((DEC && nl_ >= LEAFQ_FIRST) ? 64 : min(nl_, 64)) * 8 + PH3_LEAF, kI = min(qn[PH3_INNER], 64) * 8 + PH3_INNER;
            // (s_max_i32 by hand: the compiler folds nested maxima of wave-uniform values into v_max3_i32 -- a vector instruction, plus
            // two moves in and a v_readfirstlane back)
            const int best = STATS ? max(max(max(kC, kA), max(kB, kL)), kI) // (the counting kernels keep more scalars: theirs may live in vector registers)
                                   : smax(smax(smax(kC, kA), smax(kB, kL)), kI);
            act = best < 8 ? PH3_NONE : (best & 7); // (best < 8: every ray of the pool is dead)
        }
        if (act > PH3_LEAF) break;
        // Rays with non-finite operands (RF_EXACT: a handful per frame) walk the 2-wide reference topology with the reference's own
        // box arithmetic and, in the 16-bit layout, keep their stack in the global area.  n_exact counts those in flight in this pool
        // (a wave-uniform scalar): while it is zero -- practically always -- the steps run in the form that has none of that handling.
        auto inner_arm = [&](auto may_exact_) __attribute__((always_inline)) {
            constexpr bool MAY_EXACT = decltype(may_exact_)::value;
            if constexpr (!DEC) {
            // ---- inner-node step: the child boxes, nearest child next, the other hit children pushed ----
            POP3(PH3_INNER)
            bool t_done = false;
            int t_ref = 0;
            uint32_t t_flags = 0;
            if (on) {
                const float4 qa = S.A[id], qbd = S.B[id];
                const uint32_t qd = S.D[id];
                int ref = S.node[id];
                // (EXACT, no ray of the batch on the reference-arithmetic path: the step takes the direction itself and forms 1 / d after its
                // loads are on their way)
                constexpr bool DIR_ = MODE == 2 && !MAY_EXACT;
                const F3 o = f3(qa.x, qa.y, qa.z), inv = DIR_ ? f3(qbd.x, qbd.y, qbd.z) : inv3_exact(f3(qbd.x, qbd.y, qbd.z));
                int sp = (int)(qd & 0xffu);
                // pruning bound: fixed by the light distance for shadow rays, shrinking with the best hit otherwise
                // (MODE 2 = CRT_TRAVERSAL_EXACT: the same traversal without this bound; +inf = no bound -- a box entered at +inf is still
                // a box the reference enters, which matters to the reference-arithmetic step of the rays with non-finite operands)
                const float bound = (MODE == 0 && (qd & (RF_ANYHIT | RF_HASHIT))) ? prune_bound(qa.w, o, inv) : pinf();
                bool done = false;
                if (MODE == 1) {
                    const float4 qb = S.B[id];
                    done = inner2_step<1, STATS>(sc, S, M3, id, g, o, inv, f3(qb.x, qb.y, qb.z), bound, ref, sp, tc, max_sp);
                } else if (!MAY_EXACT) {
                    done = inner4_step<STATS, CRT_SORT4(MODE), LDS3, DIR_>(sc, S, M3, id, g, o, inv, bound, ref, sp, tc, max_sp);
                } else {
                    const bool ex = (qd & RF_EXACT) != 0;
                    if (!ex) done = inner4_step<STATS, CRT_SORT4(MODE)>(sc, S, M3, id, g, o, inv, bound, ref, sp, tc, max_sp);
                    if (__builtin_amdgcn_ballot_w64(ex)) { // reference arithmetic on the reference topology
                        if (ex) {
                            const float4 qb = S.B[id];
                            done = inner2_step<0, STATS>(sc, S, M3, id, g, o, inv, f3(qb.x, qb.y, qb.z), bound, ref, sp, tc, max_sp);
                        }
                    }
                }
                S.node[id] = ref;
                S.D[id] = (qd & ~0xffu) | (uint32_t)sp;
                if (STATS) { for (int k = 0; k < 6; k++) dg_sp[k] += sp > k + 1 ? 1u : 0u; }
                if (STATS && done && (qd & RF_HASHIT)) tc.hits++; // (an any-hit ray that records a hit ends in the leaf step)
                t_done = done; t_ref = ref; t_flags = qd;
            }
            if (MAY_EXACT && MODE != 1) n_exact -= (int)__popcll(__ballot(on && t_done && (t_flags & RF_EXACT) != 0));
            PUSH_TRAV()
            }
        };
        auto leaf_arm = [&](auto may_exact_) __attribute__((always_inline)) {
            constexpr bool MAY_EXACT = decltype(may_exact_)::value;
            if constexpr (!DEC) {
            // ---- leaf step: the record's two triangles in one packed computation ----
            POP3(PH3_LEAF)
            bool t_done = false;
            int t_ref = 0;
            uint32_t t_flags = 0;
            if (on) {
                const float4 qa = S.A[id], qb = S.B[id];
                int ref = S.node[id];
                uint32_t qd = S.D[id];
                const F3 o = f3(qa.x, qa.y, qa.z), d = f3(qb.x, qb.y, qb.z);
                float T = qa.w;
                int tri = __float_as_int(qb.w);
                int sp = (int)(qd & 0xffu);
                const int lv = MAY_EXACT || MODE == 1 ? lds_levels<LDS3>((qd & RF_EXACT) != 0) : (int)LDS3::LV;
                const int top = stack_top_ahead(S, id, sp, lv);
                const bool any_hit = (qd & RF_ANYHIT) != 0;
                int best_leaf = tri - (int)((qd >> 8) & 0xffffu); // first triangle of the leaf that holds the best hit (-1 - 0 if none)
                bool done = false;
                uint32_t rec = (uint32_t)~ref;
                int it0 = 0, left = 1;
                for (int k = 0; left > 0 && !done; k++, rec++) { // one record per pair of triangles: a single pass with bvh_thresh_n <= 2
                    const float4* lg = (const float4*)((const char*)sc.leaf_geo + rec * 80u); // (32-bit byte offset, as for the nodes)
                    const float4 g0 = lg[0], g1 = lg[1], g2 = lg[2], g3 = lg[3], g4 = lg[4];
                    const int it = __float_as_int(g4.z);
                    if (k == 0) { it0 = it; left = __float_as_int(g4.w); }
                    const bool two = left > 1;
                    bool a0, a1;
                    float t0, t1;
                    tri_pair(g0, g1, g2, g3, g4, v2(qa.x, qa.y), v2(qa.z, qa.w), v2(qb.x, qb.y), v2(qb.z, qb.w), a0, a1, t0, t1);
                    a1 = a1 && two;
                    if (STATS) { tc.tests += two ? 2u : 1u; }
                    if (any_hit) {
                        const bool b0 = a0 & (T - t0 > CRT_EPSILON);
                        const bool b1 = a1 & (T - t1 > CRT_EPSILON);
                        done = b0 | b1;
                        tri = b0 ? it : (b1 ? it + 1 : tri);
                    } else {
                        // ascending index, strict <: the first of equal t inside a leaf wins (DeviceBVH.cuh:34-41); across leaves the
                        // larger leaf start wins (reference visit order, see crt_trace.h)
                        const bool w0 = a0 & ((t0 < T) | ((t0 == T) & (it0 > best_leaf)));
                        T = w0 ? t0 : T; tri = w0 ? it : tri; best_leaf = w0 ? it0 : best_leaf;
                        const bool w1 = a1 & ((t1 < T) | ((t1 == T) & (it0 > best_leaf)));
                        T = w1 ? t1 : T; tri = w1 ? it + 1 : tri; best_leaf = w1 ? it0 : best_leaf;
                    }
                    left -= 2;
                }
                if (STATS) tc.leaf++;
                if (!any_hit && tri >= 0) qd = (qd & ~RR_ROUTE_LC_BIT) | RF_HASHIT; // (a surface was found: LC -> LA)
                if (!done) done = stack_pop_ahead(S, M3, id, g, sp, ref, top, lv);
                qd = (qd & 0xff000000u) | ((uint32_t)(tri - best_leaf) << 8 & 0xffff00u) | (uint32_t)sp;
                if (!any_hit) S.A[id].w = T;
                S.B[id].w = __int_as_float(tri);
                S.node[id] = ref;
                S.D[id] = qd;
                if (STATS && done && tri >= 0) tc.hits++;
                t_done = done; t_ref = ref; t_flags = qd;
            }
            if (MAY_EXACT && MODE != 1) n_exact -= (int)__popcll(__ballot(on && t_done && (t_flags & RF_EXACT) != 0));
            PUSH_TRAV()
            }
        };
        // ---- DEC arms (Pool4LdsT): see there ----
// appends the lanes with `cond_` (ray `id`) to ring p_
#define PUSH_ONE(p_, cond_)                                                                                                \
    {                                                                                                                      \
        const bool mine_ = (cond_);                                                                                        \
        const unsigned long long m_ = __builtin_amdgcn_ballot_w64(mine_);                                                  \
        if (m_) {                                                                                                          \
            const uint32_t slot_ = __builtin_amdgcn_mbcnt_hi((uint32_t)(m_ >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m_, RQ_PUSH_BASE(p_))); \
            if (mine_) S.rq(p_)[ring_wrap<QCAP>(slot_)] = (uint8_t)id;                                                     \
            const int add_ = (int)__popcll(m_);                                                                            \
            RQ_PUSH_ADV(p_, add_)                                                                                          \
        }                                                                                                                  \
    }
        auto inner_arm_dec = [&](auto may_exact_) __attribute__((always_inline)) {
            constexpr bool MAY_EXACT = decltype(may_exact_)::value;
            if constexpr (DEC) {
            // ---- inner-node step: the child boxes; leaf children that are hit -> queue entries; nearest inner child next ----
            // All 64 lanes run the step (round 6): a lane without a ray of the batch (ON is the batch) steps at the EMPTY node like every other
            // lane that sits the visit out, so there is no divergent region around the step, every predicate is a wave mask and every
            // wave-uniform count -- queue entries appended, rays re-queued -- stays in a scalar register (until round 5 they left the region
            // in a vector register, through v_readfirstlane).  Stores and LDS atomics are predicated by `lanes(mask)`.
            // (the batch is not cut to the quarter of the free queue entries that could take four entries per ray: the entries of a visit
            // are counted before any is written, and should they not fit -- rarely: a ray adds 0.8 on average -- the lanes beyond that quarter are
            // taken out of the visit again and queued as they came, inner4_step_dec.  Batches 57 -> 62 rays: C2 80.4 -> 79.5 ms, veach 80.4 -> 79.2)
            const int take = min(64, qn[PH3_INNER]);
            if (STATS) { dg_b[PH3_INNER]++; dg_l[PH3_INNER] += (uint32_t)take; }
            const wmask ON = bal(lane < take);
            // (a lane beyond the batch reads an entry of the ring that is not part of it -- a ray id of this pool all the same, or the
            // ring's initial bytes: the id is forced into the pool)
            const uint32_t id_raw = S.rq(PH3_INNER)[ring_wrap<QCAP>(RQ_POP_BASE(PH3_INNER, take) + (uint32_t)lane)];
            RQ_POP_ADV(PH3_INNER, take)
            const uint32_t id = lanes(ON) ? id_raw : 0u;
            const uint32_t g = base + id;
            uint32_t nph = PH3_NONE;      // a ray that is complete (its walk is over and none of its entries is in flight): where it goes
            uint32_t added = 0;           // entries of the batch so far (wave-uniform)
            wmask DONE, EXQ = 0ull;
            {
                const float4 qa = S.A[id], qb = S.B[id];
                const uint32_t qd = S.D[id];
                const uint32_t blo = (uint32_t)__hip_atomic_load(&S.best[id], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                int ref = __float_as_int(qb.w);
                const F3 o = f3(qa.x, qa.y, qa.z), dir = f3(qb.x, qb.y, qb.z);
                int sp = (int)(qd & 0xffu);
                // an any-hit ray that has its answer looks no further (entries of it still in the queue are tested and change nothing)
                const wmask D0 = bal((qd & RF_ANYHIT) != 0) & bal(blo != 0u) & ON;
                const wmask GO = ON & ~D0;
                const wmask EX = MAY_EXACT ? bal((qd & RF_EXACT) != 0) : 0ull;
                DONE = D0;
                int n_leaf = 0;           // entries of this ray from its 4-wide visits
                wmask ANY = 0ull;         // rays that appended an entry in this step
                {
                    wmask VOID = 0ull;
                    RayPk R;
                    R.oxy = v2(qa.x, qa.y); R.oz.x = qa.z;
                    // (free entries, saturating: the scheduler drains the queue from LEAFQ_FIRST entries on and a step adds at most what is
                    // free, so the difference cannot be negative -- but a wrapped unsigned here would switch the overflow check off for good)
                    const uint32_t lq_used = (lq_t - lq_h) + (MAY_EXACT ? 64u : 0u);
                    const uint32_t lq_free = lq_used < (uint32_t)LEAFQ_CAP ? (uint32_t)LEAFQ_CAP - lq_used : 0u;
                    // (MAY_EXACT: the rays of the reference-arithmetic path append one entry each below, outside that count)
                    Visit4 V1, V2;
                    const wmask EN1 = GO & ~EX;
                    node_load<IMPL>(sc, ref, EN1, dir, V1);
#ifdef CRT_X_EXTRA_LOADS /* sensitivity experiment (round 6): N more loads per first visit from the node's own line, issued with the others, consumed at the end of the step */
                    float xl_[CRT_X_EXTRA_LOADS];
                    for (int k_ = 0; k_ < CRT_X_EXTRA_LOADS; k_++) {
                        uint32_t off_ = lanes(EN1) ? (uint32_t)ref * (IMPL ? (uint32_t)(NODE4I_F4 * 16) : 128u) : 0u;
                        asm volatile("" : "+v"(off_));
#if defined(CRT_X_EXTRA_UNIFORM)
                        off_ &= 0u; /* every lane the same address */
#endif
                        xl_[k_] = *(const float*)(((const char*)(IMPL ? sc.nodes4i : sc.nodes4) + off_) + 80);
                    }
#endif
                    const int top1 = stack_top_ahead(S, id, sp, LDS3::LV);
                    DONE |= visit_front<STATS, LDS3, 1, IMPL>(sc, M3, g, dir, R, V1, top1, ref, sp, tc, max_sp, lq_t, added, ANY, n_leaf, EN1, lq_free, nullptr, &VOID);
                    if (STATS && lanes(EN1 & VOID)) dg_ov[0]++;
                    // A SECOND NODE in the same step for the rays that go on, their record still in registers (in the form
                    // this pool allows: with the leaves decoupled a ray that is not finished always has an inner node next).  Taken while at
                    // least VISIT2_MIN lanes go on; its leaf entries are counted before anything is written, and if the queue cannot take
                    // them the visit is dropped.  C2 82.7 -> 81.7 ms, veach-mis spp 256
                    // 81.8 -> 80.3 (40 .. 52: the same; 16 / 32: 82.6 / 82.1 and 80.8 / 80.6; as a loop, or three / four visits: worse).
                    // Round 6: its node is fetched BEFORE the first visit's appends and pushes are written (node_load / visit_back).
#ifndef CRT_VISIT2_MIN
#define CRT_VISIT2_MIN 44
#endif
                    constexpr int VISIT2_MIN = CRT_VISIT2_MIN;
                    const wmask EN2 = ON & ~DONE;
                    const bool second = !MAY_EXACT && (int)__popcll(EN2) >= VISIT2_MIN;
                    if (second) node_load<IMPL>(sc, ref, EN2, dir, V2);
                    visit_back(S, M3, id, g, V1);
                    if (second) {
                        const uint32_t lq_used2 = (lq_t + added) - lq_h;
                        const uint32_t cap_left = lq_used2 < (uint32_t)LEAFQ_CAP ? (uint32_t)LEAFQ_CAP - lq_used2 : 0u;
                        bool bailed = false;
                        const int top2 = stack_top_ahead(S, id, sp, LDS3::LV);
                        DONE |= visit_front<STATS, LDS3, 2, IMPL>(sc, M3, g, dir, R, V2, top2, ref, sp, tc, max_sp, lq_t, added, ANY, n_leaf, EN2, cap_left, &bailed);
                        if (!bailed) visit_back(S, M3, id, g, V2);
                        if (STATS && bailed && lanes(EN2)) dg_ov[1]++;
                    }
#ifdef CRT_X_EXTRA_LOADS
                    for (int k_ = 0; k_ < CRT_X_EXTRA_LOADS; k_++) asm volatile("" :: "v"(xl_[k_]));
#endif
                }
                if (MAY_EXACT) {
                    const wmask GX = GO & EX;
                    if (GX) { // reference arithmetic on the reference topology, one thing per visit: a leaf ref becomes a queue entry, an inner node is stepped
                        const wmask LF = GX & bal(ref < 0);
                        uint32_t lrec = (uint32_t)~ref; // (a leaf ref of nodes3: a record of leaf_geo)
                        if (IMPL) lrec = (uint32_t)sc.rec_map[lanes(LF) ? lrec : 0u]; // ... whose copy in leaf_geo_i the leaf step reads
                        leafq_push(S, id, lanes(LF), LF, (lrec << 8) | id, lq_t, added);
                        bool dn = false;
                        if (lanes(GX)) {
                            if (lanes(LF)) dn = stack_pop(S, M3, id, g, sp, ref, lds_levels<LDS3>(true));
                            else dn = inner2_step<0, STATS>(sc, S, M3, id, g, o, inv3_exact(dir), dir, pinf(), ref, sp, tc, max_sp);
                        }
                        ANY |= LF;
                        DONE |= bal(dn) & GX;
                    }
                    EXQ = DONE & EX; // rays of the reference-arithmetic path whose walk ended
                }
                // the record: node, stack depth, "the walk is over" -- the count of entries in flight in between is touched by atomics only
                // (leafq_push above: those additions are in LDS before this one, same wave, in order)
                if (STATS) { for (int k = 0; k < 6; k++) dg_sp[k] += (lanes(ON) && sp > k + 1) ? 1u : 0u; }
                if (lanes(ON)) {
                    S.B[id].w = __int_as_float(ref);
                    __hip_atomic_fetch_add(&S.D[id], (uint32_t)sp - (qd & 0xffu) + (lanes(DONE) ? RD_FIN : 0u) + ((uint32_t)n_leaf << RD_PEND_SHIFT), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                const wmask COMP = DONE & ~ANY & bal((qd & RD_PEND_MASK) == 0u);
                if (STATS && lanes(COMP) && blo != 0u) tc.hits++;
                if (COMP) nph = lanes(COMP) ? route_complete<QUERY>(qd, blo != 0u) : nph;
            }
            // re-queue the rays that go on
            const wmask MC = ON & ~DONE;
            if (MC) {
                const uint32_t slot = __builtin_amdgcn_mbcnt_hi((uint32_t)(MC >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)MC, RQ_PUSH_BASE(PH3_INNER)));
                if (lanes(MC)) S.rq(PH3_INNER)[ring_wrap<QCAP>(slot)] = (uint8_t)id;
                const int c = (int)__popcll(MC);
                RQ_PUSH_ADV(PH3_INNER, c)
            }
            lq_t += added;
            if (MAY_EXACT) n_exact -= (int)__popcll(EXQ);
            if (bal(nph != PH3_NONE)) {
                PUSH_ONE(PH3_LA, nph == PH3_LA) PUSH_ONE(PH3_LB, nph == PH3_LB) PUSH_ONE(PH3_LC, nph == PH3_LC)
            }
            }
        };
        auto leaf_arm_dec = [&]() __attribute__((always_inline)) {
            if constexpr (DEC) {
            // ---- leaf step: 64 entries of the queue, the record's two triangles in one packed computation ----
            // All 64 lanes run it (round 6, as the inner step): a lane beyond the batch takes entry 0 of record 0 -- valid addresses -- and its
            // result is dropped: the minimum and the count of entries in flight are touched under `lanes(ON)` only.
            const int take = min(64, (int)(lq_t - lq_h));
            if (STATS) { dg_b[PH3_LEAF]++; dg_l[PH3_LEAF] += (uint32_t)take; }
            const wmask ON = bal(lane < take);
            // (fetching these entries and their records from the inner step before this one -- they are known when the queue held 64 before it --
            // was measured in round 6: the values arrive in other registers than the step's own loads use, 20 copies and a full wait: +1.3 %)
            const uint32_t item_raw = S.leafq[(lq_h + (uint32_t)lane) & (uint32_t)(LEAFQ_CAP - 1)];
            lq_h += (uint32_t)take;
            const uint32_t item = lanes(ON) ? item_raw : 0u;
            const uint32_t id = item & 0xffu;
            wmask COMP = 0ull;
            uint32_t t_flags = 0;
            {
                const float4 qa = S.A[id], qb = S.B[id];
                const F3 o = f3(qa.x, qa.y, qa.z), d = f3(qb.x, qb.y, qb.z);
                const float Tl = qa.w;
                // The entry leaves the ray's count of entries in flight NOW (round 6): the returning LDS atomic's round trip hides behind the
                // record's loads instead of standing alone at the end of the step.  Every minimum of this step is in LDS before the step ends,
                // and nothing reads a ray's answer before the step that routes it has ended -- the order of the two atomics inside a step is free.
                if (lanes(ON)) t_flags = __hip_atomic_fetch_sub(&S.D[id], 1u << RD_PEND_SHIFT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                uint32_t rec = item >> 8;
                // the leaf's candidate: the first of its triangles among equal distances (ascending index, strict <: DeviceBVH.cuh:34-41)
                bool have = false;
                float bt = 0.0f;
                int bi = 0;
                int left = 1;
                // (a leaf step without this loop for scenes whose leaves are one record each -- bvh_thresh_n <= 2 -- was measured in round 5: eleven
                // instructions fewer per step, C2 +0.3 %, veach-mis +0.2 %: the second copy of the pair test costs what they save; not kept)
                for (int k = 0; left > 0; k++, rec++) { // one record per pair of triangles: a single pass with bvh_thresh_n <= 2
                    const float4* lg = (const float4*)((const char*)(IMPL ? sc.leaf_geo_i : sc.leaf_geo) + rec * 80u);
                    const float4 g0 = lg[0], g1 = lg[1], g2 = lg[2], g3 = lg[3], g4 = lg[4];
                    const int it = __float_as_int(g4.z);
                    if (k == 0) left = __float_as_int(g4.w);
                    const bool two = left > 1;
                    bool a0, a1;
                    float t0, t1;
                    tri_pair(g0, g1, g2, g3, g4, v2(qa.x, qa.y), v2(qa.z, qa.w), v2(qb.x, qb.y), v2(qb.z, qb.w), a0, a1, t0, t1);
                    if (STATS && lanes(ON)) { tc.tests += two ? 2u : 1u; }
                    // (Tl - t > EPSILON: the visibility test of an any-hit ray, Render.cuh:19-27; always true for Tl = +inf and a finite t,
                    // false for t = +inf, which the reference's t < best.t rejects as well)
                    const bool b0 = a0 & (Tl - t0 > CRT_EPSILON);
                    const bool b1 = a1 & two & (Tl - t1 > CRT_EPSILON);
                    const bool s1 = b1 & (!b0 | (t1 < t0));
                    const float ct = s1 ? t1 : t0;
                    const int ci = s1 ? it + 1 : it;
                    const bool up = (b0 | b1) & (!have | (ct < bt));
                    bt = up ? ct : bt; bi = up ? ci : bi; have = have | up;
                    left -= 2;
                }
                if (STATS && lanes(ON)) tc.leaf++;
                if (lanes(ON)) {
                    // across leaves: the smaller distance, among equal ones the larger leaf start (crt_trace.h) = the larger triangle index, as
                    // the leaves own disjoint ascending ranges -- one 64-bit minimum over (bits(t), ~triangle); t > EPSILON > 0, so its bits order as it does
                    if (have) __hip_atomic_fetch_min(&S.best[id], ((unsigned long long)__float_as_uint(bt) << 32) | (unsigned long long)(uint32_t)~bi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                COMP = bal((t_flags & (RD_PEND_MASK | RD_FIN)) == ((1u << RD_PEND_SHIFT) | RD_FIN)); // the last entry of a ray whose walk is over (t_flags is 0 beyond the batch)
            }
            if (COMP) {
                uint32_t blo = 0;
                if (lanes(COMP)) blo = (uint32_t)__hip_atomic_load(&S.best[id], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (STATS && lanes(COMP) && blo != 0u) tc.hits++;
                const uint32_t nph = lanes(COMP) ? route_complete<QUERY>(t_flags, blo != 0u) : (uint32_t)PH3_NONE;
                PUSH_ONE(PH3_LA, nph == PH3_LA) PUSH_ONE(PH3_LB, nph == PH3_LB) PUSH_ONE(PH3_LC, nph == PH3_LC)
            }
            }
        };
        const bool plain = MODE == 1 || n_exact == 0;
        // The two traversal steps ALTERNATE without going back to the scheduler while the other side holds a batch and no logic ring holds
        // a full one: the scheduler's comparison of the five rings (36 scalar instructions -- a third of them copies of the loop-carried
        // ring cursors -- and the dispatch on its result) stands between every two steps of the wave, and a wave issues one instruction
        // at a time whatever its kind.  Coupled form: C2 92.6 -> 91.1 ms, veach-mis spp 256 88.0 -> 84.8 with CHAIN_MIN 44
        // (56 / 48 / 40 / 32 / 20: 93.0 / 91.3 / 91.2 / 91.7 / 93.2 and 85.9 / 84.9 / 84.9 / 86.2 / 90.2); decoupled form: 92.2 -> 87.7 and
        // 89.5 -> 84.8 (32 / 52: the same).  Not taken: staying on the SAME ring while it holds another full batch (92.9 / 86.7); one chained
        // step only, each arm compiled twice (92.1 / 86.0); the same preference expressed in the scheduler's keys (no gain); a short-cut in
        // front of the scheduler (93.6 / 88.7).
#ifndef CRT_CHAIN_MIN
#define CRT_CHAIN_MIN 44
#endif
        constexpr int CHAIN_MIN = CRT_CHAIN_MIN;
        auto logic_waits = [&]() __attribute__((always_inline)) {
            if constexpr (STATS) return max(max(qn[PH3_LA], qn[PH3_LB]), qn[PH3_LC]) >= 64; // (the counting kernels' cursors may live in vector registers)
            else return smax(smax(qn[PH3_LA], qn[PH3_LB]), qn[PH3_LC]) >= 64;
        };
        if constexpr (DEC) {
            // (the leaf queue goes first while it holds LEAFQ_FIRST entries, as in the scheduler)
            if (plain) {
                bool do_inner = act == PH3_INNER;
                for (;;) {
                    if (do_inner) {
                        STAMP_OTHER()
                        inner_arm_dec(std::false_type{});
                        STAMP(2)
                        if ((int)(lq_t - lq_h) < LEAFQ_FIRST || logic_waits()) break;
                    }
                    STAMP_OTHER()
                    leaf_arm_dec();
                    STAMP(1)
                    if (logic_waits()) break;
                    do_inner = (int)(lq_t - lq_h) < LEAFQ_FIRST;
                    if (do_inner && qn[PH3_INNER] < CHAIN_MIN) break;
                }
            } else if (act == PH3_INNER) {
                inner_arm_dec(std::true_type{});
            } else {
                leaf_arm_dec();
            }
        } else {
        if (plain) {
            bool do_inner = act == PH3_INNER;
            for (;;) {
                if (do_inner) {
                    inner_arm(std::false_type{});
                    if (qn[PH3_LEAF] < CHAIN_MIN || logic_waits()) break;
                }
                leaf_arm(std::false_type{});
                if (qn[PH3_INNER] < CHAIN_MIN || logic_waits()) break;
                do_inner = true;
            }
        } else if (act == PH3_INNER) {
            inner_arm(std::true_type{});
        } else {
            leaf_arm(std::true_type{});
        }
        }
      }
        // (commit ring: a pool with nothing to do but slots that are held back looks at those)
        if (act == PH3_NONE && !(commit_ring && (__builtin_amdgcn_readfirstlane((int)S.waitq) & 0xff) != 0)) break;
        STAMP_OTHER()
        if (act == PH3_LA) {
            POP3(PH3_LA)
            bool new_exact = false;
            if (on) {
                LOGIC_PARAMS()
                NewRay nr;
                float4 ra_, rb_;
                ray_result(S, id, ra_, rb_);
                nph = logic_A<MODE, RING>(Pl, tl, g, ra_, rb_, nr, cnt, ALL);
                if (nph == PH3_LB) { S.A[id].x = nr.o.x; S.A[id].y = nr.o.y; S.A[id].z = nr.o.z; } // (a vertex that goes to the roulette without a ray of its own)
                if (nph == PH3_NONE) nph = start_ray<MODE, false, LDS3, IMPL>(Pl.sc, S, id, nr, cnt, M3.force_exact != 0, new_exact);
            }
            n_exact += (int)__popcll(__ballot(new_exact));
            PUSH3()
            STAMP(0)
        } else if (act == PH3_LB) {
            POP3(PH3_LB)
            bool new_exact = false;
            if (on) {
                LOGIC_PARAMS()
                NewRay nr;
                float4 ra_, rb_;
                ray_result(S, id, ra_, rb_);
                nph = logic_B<MODE, RING>(Pl, g, ra_, rb_, nr);
                if (nph == PH3_NONE) nph = start_ray<MODE, false, LDS3, IMPL>(Pl.sc, S, id, nr, cnt, M3.force_exact != 0, new_exact);
            }
            n_exact += (int)__popcll(__ballot(new_exact));
            PUSH3()
            STAMP(4)
            // (going from here straight to LC when LC holds 16 / 32 paths -- LB's roulette stops feed it -- was measured in round 5: C2 +3.0 % / +0.7 %,
            // veach-mis +1.4 % / +1.0 %: it runs LC with emptier batches than the fullest-first rule; not kept)
        }
        if (act != PH3_LA && act != PH3_LB) {
            // LC, or -- commit ring -- a look at the slots that are held back (PH3_WAIT), through the same code: the held slots have
            // their turn when the pool has nothing else to do and, while there are any, at every second visit of this phase
            // (no loop around the phase: it would count as one more level of nesting in the compiler's register allocation)
            int wn = 0, wh = 0, wt = 0; // ring PH3_WAIT: entries, head, tail
            bool held = false;
            if (commit_ring) {
                const uint32_t wq = (uint32_t)__builtin_amdgcn_readfirstlane((int)S.waitq);
                wn = (int)(wq & 0xffu); wh = (int)((wq >> 8) & 0xffu); wt = (int)((wq >> 16) & 0xffu);
                held = act == PH3_NONE || (wn > 0 && (wq >> 24) != 0u);
            }
            const int src_n = held ? wn : qn[PH3_LC];
            const int take = min(64, src_n);
            const int src_h = held ? wh : (int)RQ_POP_BASE(PH3_LC, take);
            if (STATS) { dg_b[PH3_LC]++; dg_l[PH3_LC] += (uint32_t)take; }
            const bool on = lane < take;
            // (the parking ring of the commit ring is a FIFO whatever CRT_RING_MODE says: its wrap is by compare)
            const uint32_t pop_i = (uint32_t)(src_h + lane);
            const uint32_t id = held ? S.rq(PH3_WAIT)[min(pop_i, pop_i - (uint32_t)QCAP)] : S.rq(PH3_LC)[ring_wrap<QCAP>(pop_i)];
            if (held) {
                int nh = src_h + take;
                if (nh >= QCAP) nh -= QCAP;
                wh = nh; wn -= take;
            } else {
                RQ_POP_ADV(PH3_LC, take)
            }
            const uint32_t g = base + id;
            uint32_t nph = PH3_NONE;
            bool new_exact = false, wait = false;
            uint32_t fin_key = ~0u;
            if (on) {
                LOGIC_PARAMS()
                NewRay nr;
                float4 ra_ = make_float4(0.0f, 0.0f, 0.0f, 0.0f), rb_ = ra_;
                if (QUERY) ray_result(S, id, ra_, rb_);
                const int got = QUERY ? (query_C(Pl, g, ra_, rb_, nr) ? LC_RAY : LC_DEAD) : logic_C<MODE, RING>(Pl, tl, g, cnt, nr, fin_key);
                if (got == LC_RAY) nph = start_ray<MODE, QUERY, LDS3, IMPL>(Pl.sc, S, id, nr, cnt, M3.force_exact != 0, new_exact);
                wait = got == LC_WAIT;
            }
            n_exact += (int)__popcll(__ballot(new_exact));
            PUSH3()
            if (commit_ring) { // count the finished work items, commit what that completes; park the slots that are held back
                LOGIC_PARAMS()
                ring_publish(Pl, fin_key);
                const unsigned long long mw = __ballot(wait);
                if (mw) {
                    const uint32_t slot = __builtin_amdgcn_mbcnt_hi((uint32_t)(mw >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mw, (uint32_t)wt));
                    if (wait) S.rq(PH3_WAIT)[min(slot, slot - (uint32_t)QCAP)] = (uint8_t)id;
                    const int add = (int)__popcll(mw);
                    wn += add;
                    wt += add;
                    if (wt >= QCAP) wt -= QCAP;
                }
                if (held && mw == __ballot(on)) __builtin_amdgcn_s_sleep(64); // (none of them may start yet: no hurry)
                if (lane == 0) S.waitq = (uint32_t)wn | ((uint32_t)wh << 8) | ((uint32_t)wt << 16) | (held ? 0u : 1u << 24);
            }
            STAMP(5)
        }
    }
#undef STAMP
#undef STAMP_OTHER
#undef PUSH3
#undef PUSH_ONE
#undef PUSH_TRAV
#undef POP3
#undef LOGIC_PARAMS

    // ---- counters ----
#ifdef CRT_STAMPS
    if (lane == 0) {
        unsigned long long* cs_ = M.counters + (blockIdx.x & (CNT_SHARDS - 1)) * CNT_STRIDE;
        atomicAdd(&cs_[C_CYC_LOGIC], st_acc[0]); atomicAdd(&cs_[C_CYC_LEAF], st_acc[1]); atomicAdd(&cs_[C_CYC_INNER], st_acc[2]); atomicAdd(&cs_[C_CYC_OTHER], st_acc[3]);
        atomicAdd(&cs_[C_DIAG + 0], st_acc[4]); atomicAdd(&cs_[C_DIAG + 1], st_acc[5]);
        for (int k = 0; k < 6; k++) atomicAdd(&cs_[C_DIAG + 2 + k], (unsigned long long)st_n[k]);
    }
#endif
    uint32_t r = wave_sum(cnt.rays), sh = wave_sum(cnt.shadow), pr = wave_sum(cnt.probe), pa = wave_sum(cnt.paths), un = wave_sum(cnt.untraced);
    unsigned long long* cs = M.counters + (blockIdx.x & (CNT_SHARDS - 1)) * CNT_STRIDE;
    if (lane == 0 && (r | pa)) {
        atomicAdd(&cs[C_RAYS], (unsigned long long)r);
        atomicAdd(&cs[C_SHADOW], (unsigned long long)sh);
        atomicAdd(&cs[C_PROBE], (unsigned long long)pr);
        atomicAdd(&cs[C_PATHS], (unsigned long long)pa);
        atomicAdd(&cs[C_UNTRACED], (unsigned long long)un);
    }
    if (STATS) {
        uint32_t a = wave_sum(tc.inner), b = wave_sum(tc.leaf), c = wave_sum(tc.tests), d = wave_sum(tc.hits);
        uint32_t ms = max_sp;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) ms = max(ms, (uint32_t)__shfl_xor((int)ms, o, 64));
        if (lane == 0) {
            atomicAdd(&cs[C_INNER], (unsigned long long)a);
            atomicAdd(&cs[C_LEAF], (unsigned long long)b);
            atomicAdd(&cs[C_TESTS], (unsigned long long)c);
            atomicAdd(&cs[C_HITS], (unsigned long long)d);
            atomicMax(&cs[C_MAXSP], (unsigned long long)ms);
            for (int p = 0; p < 5; p++) {
                atomicAdd(&cs[C_DIAG + 2 * p], (unsigned long long)dg_b[p]);
                atomicAdd(&cs[C_DIAG + 2 * p + 1], (unsigned long long)dg_l[p]);
            }
        }
        for (int k = 0; k < 6; k++) {
            const uint32_t v = wave_sum(dg_sp[k]);
            if (lane == 0) atomicAdd(&cs[C_DIAG + 10 + k], (unsigned long long)v);
        }
        for (int k = 0; k < 2; k++) {
            const uint32_t v = wave_sum(dg_ov[k]);
            if (lane == 0) atomicAdd(&cs[C_DIAG + 16 + k], (unsigned long long)v);
        }
    }
}

// Orders the LAST order_window work items of every cursor shard (what the waves are handed when a launch ends): first the paths whose
// roulette draw lets them continue past their first vertex, then the ones it stops there (and the padding slots of ragged tiles).  The draws are addressed (crt_detmath.h), so this is known before
// anything is traced; the order of the work items cannot change a result (every path writes its own L[item]).  Why: every launch ends
// with each wave running its pool dry, and the time that takes is the longest path started last -- with one-vertex paths at the
// end of every shard the fixed cost of a launch drops from 2.5 ms to about 1 ms (tools/share_probe.py: a rank's share of C2 at 1 / 2 / 4 /
// 8 ranks 107.3 / 54.9 / 28.4 / 15.9 ms without, 106.8 / 54.1 / 27.5 / 14.6 ms with, this pass included).  One wave orders a span of
// 1 024 items of one shard with two atomics (a cache line per counter).
template <bool RING>
__global__ __launch_bounds__(64) void k_order_items(const LParams P, uint32_t* list, unsigned int* cnt)
{
    const uint32_t spans = (P.order_window + 1023u) / 1024u;
    const uint32_t sh = blockIdx.x / spans, sp = blockIdx.x - sh * spans;
    const uint32_t slo = sh * P.items_per_shard;
    if (slo >= P.n_items) return;
    const uint32_t hi = min(slo + P.items_per_shard, P.n_items);
    const uint32_t lo = hi - min(P.order_window, hi - slo); // the window: the last order_window items of the shard
    const uint32_t b = lo + sp * 1024u;
    if (b >= hi) return;
    uint32_t* out = list + (size_t)sh * P.order_window; // out[k] = the item handed out in place of item lo + k
    const uint32_t lane = threadIdx.x;
    uint32_t goes_on = 0, exists = 0; // bit j: item b + 64 j + lane
#pragma unroll 1
    for (uint32_t j = 0; j < 16; j++) {
        const uint32_t i = b + j * 64u + lane;
        if (i < hi) {
            bool valid; uint32_t pi, pj, pixel_index, k;
            decode_item<RING>(P, i, pixel_index, k, valid, pi, pj);
            exists |= 1u << j;
            // the roulette of the first vertex, as logic_B draws it (Render.cuh:223-227)
            if (valid && !(rng_uniform(rng_draw(P.seed, pixel_index, k, 0, RNG_BOUNCE, 0).x) > P.p_rr)) goes_on |= 1u << j;
        }
    }
    uint32_t n_on = 0, n_all = 0;
#pragma unroll 1
    for (uint32_t j = 0; j < 16; j++) {
        n_on += (uint32_t)__popcll(__ballot((goes_on >> j) & 1u));
        n_all += (uint32_t)__popcll(__ballot((exists >> j) & 1u));
    }
    unsigned int base_on = 0, base_off = 0;
    if (lane == 0) { // (one 128 B line per counter: the atomics of a shard serialise on their line, those of different shards must not)
        base_on = atomicAdd(cnt + (sh * 2u) * 32u, n_on);
        base_off = atomicAdd(cnt + (sh * 2u + 1u) * 32u, n_all - n_on);
    }
    base_on = (unsigned int)__builtin_amdgcn_readfirstlane((int)base_on);
    base_off = (unsigned int)__builtin_amdgcn_readfirstlane((int)base_off);
    const uint32_t wn = hi - lo;
#pragma unroll 1
    for (uint32_t j = 0; j < 16; j++) {
        const bool on_j = (goes_on >> j) & 1u, ex_j = (exists >> j) & 1u;
        const unsigned long long m_on = __ballot(on_j), m_off = __ballot(ex_j && !on_j);
        const unsigned long long below = (1ull << lane) - 1ull;
        const uint32_t i = b + j * 64u + lane;
#if defined(CRT_HANDOFF_PLAIN) || defined(CRT_HANDOFF_PLAIN_STORE)
        if (on_j) out[base_on + (uint32_t)__popcll(m_on & below)] = i;
        else if (ex_j) out[wn - 1u - (base_off + (uint32_t)__popcll(m_off & below))] = i;
#else
        if (on_j) __hip_atomic_store(&out[base_on + (uint32_t)__popcll(m_on & below)], i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (ex_j) __hip_atomic_store(&out[wn - 1u - (base_off + (uint32_t)__popcll(m_off & below))], i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
        base_on += (unsigned int)__popcll(m_on);
        base_off += (unsigned int)__popcll(m_off);
    }
}


// ---- exported to crt_render.hip ----
void launch_order_items(bool ring, uint32_t blocks, hipStream_t st, const LParams& P, uint32_t* list, unsigned int* cnt)
{
    if (ring) hipLaunchKernelGGL(k_order_items<true>, dim3(blocks), dim3(64), 0, st, P, list, cnt);
    else hipLaunchKernelGGL(k_order_items<false>, dim3(blocks), dim3(64), 0, st, P, list, cnt);
}
#ifdef CRT_ASM_ONLY_DEFAULT /* tools/diet/asm_default.sh: the default instantiation alone (a quick assembly listing; never a library) */
#ifndef CRT_ASM_ONLY_ARGS
#define CRT_ASM_ONLY_ARGS 2, false, false, false, true, false, true, true
#endif
Mega3Kernel mega3_kernel(int, bool, bool, bool, bool, bool, bool, bool) { return (Mega3Kernel)k_mega3<CRT_ASM_ONLY_ARGS>; }
#else
template <bool R16, bool DEC, bool IMPL = false> Mega3Kernel mega3_exact_kernel(bool stats, bool all, bool query, bool ring)
{
    if (ring) return all ? (Mega3Kernel)k_mega3<2, false, true, false, R16, true, DEC, IMPL> : (Mega3Kernel)k_mega3<2, false, false, false, R16, true, DEC, IMPL>;
    if (query) return (Mega3Kernel)k_mega3<2, false, false, true, R16, false, DEC, IMPL>;
    if (all) return stats ? (Mega3Kernel)k_mega3<2, true, true, false, R16, false, DEC, IMPL> : (Mega3Kernel)k_mega3<2, false, true, false, R16, false, DEC, IMPL>;
    return stats ? (Mega3Kernel)k_mega3<2, true, false, false, R16, false, DEC, IMPL> : (Mega3Kernel)k_mega3<2, false, false, false, R16, false, DEC, IMPL>;
}
Mega3Kernel mega3_kernel(int mode, bool stats, bool all, bool query, bool r16, bool ring, bool dec, bool impl)
{
    if (mode == 2) {
        if (dec && r16 && impl) return mega3_exact_kernel<true, true, true>(stats, all, query, ring);
        if (dec) return r16 ? mega3_exact_kernel<true, true>(stats, all, query, ring) : mega3_exact_kernel<false, true>(stats, all, query, ring);
        return r16 ? mega3_exact_kernel<true, false>(stats, all, query, ring) : mega3_exact_kernel<false, false>(stats, all, query, ring);
    }
    if (ring) { // (a render without counters)
        if (mode == 1) return (Mega3Kernel)k_mega3<1, false, false, false, false, true>;
        if (all) return r16 ? (Mega3Kernel)k_mega3<0, false, true, false, true, true> : (Mega3Kernel)k_mega3<0, false, true, false, false, true>;
        return r16 ? (Mega3Kernel)k_mega3<0, false, false, false, true, true> : (Mega3Kernel)k_mega3<0, false, false, false, false, true>;
    }
    if (mode == 1) return query ? (Mega3Kernel)k_mega3<1, false, false, true> : stats ? (Mega3Kernel)k_mega3<1, true> : (Mega3Kernel)k_mega3<1, false>;
    if (query) return r16 ? (Mega3Kernel)k_mega3<0, false, false, true, true> : (Mega3Kernel)k_mega3<0, false, false, true, false>;
    if (all) {
        if (stats) return r16 ? (Mega3Kernel)k_mega3<0, true, true, false, true> : (Mega3Kernel)k_mega3<0, true, true, false, false>;
        return r16 ? (Mega3Kernel)k_mega3<0, false, true, false, true> : (Mega3Kernel)k_mega3<0, false, true, false, false>;
    }
    if (stats) return r16 ? (Mega3Kernel)k_mega3<0, true, false, false, true> : (Mega3Kernel)k_mega3<0, true, false, false, false>;
    return r16 ? (Mega3Kernel)k_mega3<0, false, false, false, true> : (Mega3Kernel)k_mega3<0, false, false, false, false>;
}
#endif
// rays per wave / stack levels in LDS of a launch's kernel
uint32_t mega3_pool_p(bool dec, bool ring) { return dec ? (ring ? (uint32_t)Pool4LdsT<true, true>::P : (uint32_t)Pool4LdsT<true, false>::P) : (uint32_t)POOL3_P; }
int mega3_lds_levels(bool dec, bool r16) { return dec ? (r16 ? Pool4LdsT<true, false>::LV : Pool4LdsT<false, false>::LV) : POOL_LV; }
// Diagnostic hook (tools/bbprof): CRT_BBPROF_CO names a code object holding the default instantiation of k_mega3 with a counting
// prologue in every basic block (tools/bbprof/instrument.py applied to the compiler's assembly of THIS file); the launch then goes
// to that copy, the address of its counter buffer travels in MParams3::dbg_loads / dbg_valu, and the summed counters
// (one u64 per block: executions << 32 | active lanes) are written to CRT_BBPROF_OUT after every launch.  Returns false when the
// variable is not set or the kernel is another instantiation: the caller launches as usual.
bool bbprof_launch(Mega3Kernel kern, MParams3 M3, uint32_t blocks, hipStream_t st)
{
    static const char* co = std::getenv("CRT_BBPROF_CO");
    // (the default instantiations: 16-bit stack entries, CRT_TRAVERSAL_EXACT, with the leaves decoupled or not)
#ifdef CRT_ASM_ONLY_DEFAULT
    const char* sym = nullptr;
#else
    const char* sym = kern == (Mega3Kernel)k_mega3<2, false, false, false, true, false, true, true>    ? "_ZN4crtk7k_mega3ILi2ELb0ELb0ELb0ELb1ELb0ELb1ELb1EEEvNS_8MParams3E"
                      : kern == (Mega3Kernel)k_mega3<2, false, false, false, true, false, true, false>  ? "_ZN4crtk7k_mega3ILi2ELb0ELb0ELb0ELb1ELb0ELb1ELb0EEEvNS_8MParams3E"
                      : kern == (Mega3Kernel)k_mega3<2, false, false, false, true, false, false, false> ? "_ZN4crtk7k_mega3ILi2ELb0ELb0ELb0ELb1ELb0ELb0ELb0EEEvNS_8MParams3E"
                                                                                                         : nullptr;
#endif
    if (!co || !*co || !sym) return false;
    enum { N_CNT = 4096, STRIDE = 128 };
    static hipModule_t mod = nullptr;
    static hipFunction_t fn = nullptr;
    static char* buf = nullptr;
    static char* cnt = nullptr;
    static std::vector<unsigned long long> sum(N_CNT, 0ull);
    if (!fn) {
        HIP_CHECK(hipModuleLoad(&mod, co));
        if (hipModuleGetFunction(&fn, mod, sym) != hipSuccess) { fn = nullptr; return false; } // (the code object holds the other form)
        HIP_CHECK(hipMalloc((void**)&buf, 2 * (size_t)N_CNT * STRIDE));
        // the prologues add block offsets to the low address word without a carry: the counters must not straddle a 4 GiB boundary
        cnt = buf;
        const uint64_t lo = (uint64_t)(uintptr_t)buf & 0xffffffffull;
        if (lo + (uint64_t)N_CNT * STRIDE > 0x100000000ull) cnt = buf + (0x100000000ull - lo);
    }
    HIP_CHECK(hipMemsetAsync(cnt, 0, (size_t)N_CNT * STRIDE, st));
    const uint64_t a = (uint64_t)(uintptr_t)cnt;
    M3.dbg_loads = (int32_t)(uint32_t)(a & 0xffffffffull);
    M3.dbg_valu = (int32_t)(uint32_t)(a >> 32);
    size_t sz = sizeof(M3);
    void* cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &M3, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
    HIP_CHECK(hipModuleLaunchKernel(fn, blocks, 1, 1, 64, 1, 1, 0, st, nullptr, cfg));
    HIP_CHECK(hipStreamSynchronize(st));
    std::vector<char> h((size_t)N_CNT * STRIDE);
    HIP_CHECK(hipMemcpy(h.data(), cnt, h.size(), hipMemcpyDeviceToHost));
    for (int i = 0; i < N_CNT; i++) { unsigned long long v; std::memcpy(&v, h.data() + (size_t)i * STRIDE, 8); sum[i] += v; }
    if (const char* out = std::getenv("CRT_BBPROF_OUT")) {
        if (FILE* f = std::fopen(out, "w")) {
            for (int i = 0; i < N_CNT; i++) if (sum[i]) std::fprintf(f, "%d %llu %llu\n", i, sum[i] >> 32, sum[i] & 0xffffffffull);
            std::fclose(f);
        }
    }
    return true;
}


} // namespace crtk

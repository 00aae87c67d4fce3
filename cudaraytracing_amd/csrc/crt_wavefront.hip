// cudaraytracing_amd/csrc/crt_wavefront.hip -- the wavefront pipeline (k_logic + persistent k_trace): the fallback for scenes beyond the megakernel's limits
// (crt_render.hip: choose_pipeline) and the form CRT_PIPELINE=2 selects.  One thread per pool slot and round; the path state streams from HBM.
#include "crt_internal.h"

namespace crtk {

// One pass of the path state machine for one slot, phases in the order every possible chain runs
// through them (result -> enter vertex -> roulette/bounce -> finish -> regenerate -> next-event
// setup), so a wave executes each phase at most once however its lanes are distributed over
// path stages.  `s` arrives with the slot's state and the ray that produced (res_t, res_tri);
// returns true if a new ray was emitted into s.ro / s.rd / s.tl / s.kind (false: the slot is dead).
template <bool LDS_TABLES>
__device__ __forceinline__ bool logic_advance(const LParams& P, const Tables<LDS_TABLES>& tb, const uint32_t slot, Lane& s, const uint32_t stage,
                                              const float res_t, const int res_tri, PathCounters& cnt)
{
    const DevScene& sc = P.sc;
    const Pool& pl = P.pool;
    bool emitted = false;
    bool do_enter = false, do_nee_done = false, do_finish = false, do_new = stage == ST_NEW, do_shadow_setup = false;
    int fin_deepest = -1; bool fin_emissive = false; F3 fin_ke = f3(0.0f, 0.0f, 0.0f);

    // ---- phase 1: consume the result of the slot's last ray ----
    if (stage == ST_SHADOW) {
        // visibility of next-event sample q (Render.cuh:19-27, :272-284)
        bool blocked = s.tl - res_t > CRT_EPSILON;
        if (!blocked) s.Ld = add3(s.Ld, s.c);
        s.q++;
        if (s.q < (uint32_t)(sc.n_lights * P.lsn)) do_shadow_setup = true; else do_nee_done = true;
    } else if (stage == ST_HIT) {
        // the camera / bounce ray that looked for vertex `depth` (Render.cuh:207-213)
        if (res_tri < 0) {
            fin_deepest = (int)s.depth - 1; fin_emissive = false;
            do_finish = true;
        } else {
            F3 pos = add3(s.ro, scalel3(res_t, s.rd)); // DeviceTriangle.cuh:50
            do_enter = true;
            if (s.depth > 0) {
                // the previous vertex (normal / material still in the vn plane) is not the deepest one:
                // cosine of its indirect term (Render.cuh:291)
                const size_t pr = (size_t)(s.depth - 1) * pl.n + slot;
                F3 pn = s.nrm;
                float cos_prev = dot3(unit3(sub3(pos, s.ro)), pn); // prev.pos == origin of this ray
                cos_prev = cos_prev > 0.0f ? cos_prev : 0.0f;
                pl.rec_a[pr].w = cos_prev;
                float4 pm1 = mat_row(tb, s.mat, 1);
                if (__float_as_uint(pm1.w) & 2u) { // SPECULAR: emitter probe, Render.cuh:294-303
                    float ns = mat_row(tb, s.mat, 0).w;
                    float4 pb = pl.rec_b[pr]; // direction that arrived at the previous vertex
                    float delta_coeff = (float)((double)(det_expf(25 / ns) - 1) / (2.71828182845904523536 - 1));
                    F3 in = unit3(f3(pb.x, pb.y, pb.z));
                    F3 out = sub3(in, scale3(pn, 2.f * dot3(in, pn)));
                    float d_theta = (float)((double)(delta_coeff * 30) * 3.14159265358979323846 / 180);
                    float d_phi = (float)((double)(delta_coeff * 120) * 3.14159265358979323846 / 180);
                    U4 rp = rng_draw(P.seed, s.pixel_index, s.k, s.depth - 1, RNG_PROBE, 0);
                    F3 refd = unit3(sample_lobe(out, d_theta, d_phi, rng_uniform(rp.x), rng_uniform(rp.y)));
                    // the probe leaves from prev.pos (= this ray's origin); keep the bounce direction for rec_b
                    pl.rec_b[(size_t)s.depth * pl.n + slot] = make_float4(s.rd.x, s.rd.y, s.rd.z, 0.0f);
                    s.rd = unit3(refd); // Ray.cuh:13
                    s.tl = 0.0f; s.kind = RAY_CLOSEST;
                    s.stage = ST_PROBE;
                    cnt.rays++; cnt.probe++;
                    emitted = true;
                    do_enter = false;
                }
            }
            s.pos = pos; s.vtri = (uint32_t)res_tri;
        }
    } else if (stage == ST_PROBE) {
        // the probe ray of vertex depth-1 (Render.cuh:304-313); vn still describes that vertex
        if (res_tri >= 0) {
            int hmat = sc.tri_mat[res_tri];
            float4 q1 = mat_row(tb, hmat, 1);
            if (__float_as_uint(q1.w) & 1u) {
                float4 q2 = mat_row(tb, hmat, 2);
                const size_t pr = (size_t)(s.depth - 1) * pl.n + slot;
                F3 pn = s.nrm;
                float4 pm0 = mat_row(tb, s.mat, 0), pm1 = mat_row(tb, s.mat, 1);
                float log_shininess = det_log10f(pm0.w);
                float shininess_coeff = (float)((double)log_shininess * 0.5 + 1);
                float ip = (float)(2.0f * 3.14159265358979323846) / 8.f;
                F3 hp = add3(s.ro, scalel3(res_t, s.rd));
                float ct = dot3(unit3(sub3(hp, s.ro)), pn); // probe origin == prev.pos
                ct = ct > 0.0f ? ct : 0.0f;
                // shininess * (ke (.) kd) * cos * inv_pdf  (:311, eager)
                F3 kekd = mul3(f3(q2.x, q2.y, q2.z), f3(pm1.x, pm1.y, pm1.z));
                F3 temp = scale3(scale3(scalel3(shininess_coeff, kekd), ct), ip);
                float4 a = pl.rec_a[pr];
                a.x = a.x + temp.x; a.y = a.y + temp.y; a.z = a.z + temp.z;
                pl.rec_a[pr] = a;
            }
        }
        // the bounce direction that found the current vertex was parked in rec_b[depth]
        float4 pb = pl.rec_b[(size_t)s.depth * pl.n + slot];
        s.rd = f3(pb.x, pb.y, pb.z);
        do_enter = true;
    }

    // ---- phase 2: a new vertex (pos, vtri) at `depth`, reached along s.rd ----
    F3 f_r = f3(0.0f, 0.0f, 0.0f);
    if (do_enter) {
        float4 g = sc.tri_geo[(size_t)s.vtri * 3 + 2];
        s.nrm = f3(g.y, g.z, g.w);
        s.mat = (uint32_t)sc.tri_mat[s.vtri];
        pl.rec_b[(size_t)s.depth * pl.n + slot] = make_float4(s.rd.x, s.rd.y, s.rd.z, __uint_as_float(s.mat));
        float4 m1 = mat_row(tb, s.mat, 1);
        if (__float_as_uint(m1.w) & 1u) { // emitter: the path ends here (Render.cuh:210)
            float4 m2 = mat_row(tb, s.mat, 2);
            fin_deepest = (int)s.depth; fin_emissive = true; fin_ke = f3(m2.x, m2.y, m2.z);
            do_finish = true;
        } else {
            s.Ld = f3(0.0f, 0.0f, 0.0f);
            s.q = 0;
            if (sc.n_lights * P.lsn > 0) do_shadow_setup = true; else do_nee_done = true;
        }
    }
    if (do_shadow_setup) {
        float4 m0 = mat_row(tb, s.mat, 0);
        f_r = f3(m0.x, m0.y, m0.z);
    }

    // ---- phase 3: direct light of vertex `depth` is complete: Russian roulette and bounce (Render.cuh:210-228) ----
    if (do_nee_done) {
        pl.rec_a[(size_t)s.depth * pl.n + slot] = make_float4(s.Ld.x, s.Ld.y, s.Ld.z, 0.0f);
        bool stop = s.depth == CRT_BOUNCE_STACK_SIZE - 1; // bounce stack full
        U4 rb;
        rb.x = rb.y = rb.z = rb.w = 0;
        if (!stop) {
            rb = rng_draw(P.seed, s.pixel_index, s.k, s.depth, RNG_BOUNCE, 0);
            stop = rng_uniform(rb.x) > P.p_rr;
        }
        if (stop) {
            fin_deepest = (int)s.depth; fin_emissive = false;
            do_finish = true;
        } else {
            F3 ndir = unit3(sample_hemisphere(s.nrm, rng_uniform(rb.y), rng_uniform(rb.z)));
            s.ro = s.pos;
            s.rd = unit3(ndir); // Ray.cuh:13
            s.tl = 0.0f; s.kind = RAY_CLOSEST;
            s.depth++;
            s.stage = ST_HIT;
            cnt.rays++;
            emitted = true;
        }
    }

    // ---- phase 4: path complete ----
    if (do_finish) {
        F3 L = finish_path(P, tb, slot, fin_deepest, fin_emissive, fin_ke);
        P.L[s.item] = make_float4(L.x, L.y, L.z, 0.0f);
        do_new = true; // regenerate in place
    }

    // ---- phase 5: take the next work item, camera ray (Render.cuh:344-347) ----
    if (do_new) {
        s.stage = ST_DEAD;
        for (;;) {
            s.item = grab_item(nullptr, P.item_next, P.items_per_shard, P.n_items, (blockIdx.x * 4u + (threadIdx.x >> 6)) & (ITEM_SHARDS - 1));
            if (s.item == ITEM_NONE) break;
            bool valid; uint32_t pi, pj;
            decode_item(P, s.item, s.pixel_index, s.k, valid, pi, pj);
            if (!valid) continue; // padding slot of a ragged tile: take another item
            cnt.paths++;
            pl.id[slot] = make_uint4(s.pixel_index, s.k, s.item, 0u);
            U4 rj = rng_draw(P.seed, s.pixel_index, s.k, 0, RNG_JITTER, 0);
            float x = (2 * ((int)pi + rng_uniform(rj.x)) / P.width - 1) * P.scale * P.ar;
            float y = (1 - 2 * ((int)pj + rng_uniform(rj.y)) / P.height) * P.scale;
            F3 cd = unit3(f3(-x, y, 1));
            F3 wd = f3(P.inv_view[0] * cd.x + (P.inv_view[3] * cd.y + P.inv_view[6] * cd.z),
                       P.inv_view[1] * cd.x + (P.inv_view[4] * cd.y + P.inv_view[7] * cd.z),
                       P.inv_view[2] * cd.x + (P.inv_view[5] * cd.y + P.inv_view[8] * cd.z));
            s.ro = f3(P.eye[0], P.eye[1], P.eye[2]);
            s.rd = unit3(wd); // Ray.cuh:13
            s.tl = 0.0f; s.kind = RAY_CLOSEST;
            s.depth = 0; s.stage = ST_HIT; s.q = 0;
            cnt.rays++;
            emitted = true;
            break;
        }
    }

    // ---- phase 6: next-event sample q of the current vertex ----
    if (do_shadow_setup) {
        setup_shadow(P, tb, s, f_r);
        s.stage = ST_SHADOW;
        cnt.rays++; cnt.shadow++;
        emitted = true;
    }

    return emitted;
}

// Wavefront form: one thread per pool slot and round.  The kernel is latency / bandwidth bound (a
// slot's state streams from HBM): every plane is requested up front, the current / previous
// vertex's normal and material ride along in the pool instead of being re-derived through
// triangle -> material lookups.
template <bool LDS_TABLES>
__global__ __launch_bounds__(256) void k_logic(const LParams P)
{
    const DevScene& sc = P.sc;
    const Pool& pl = P.pool;
    const uint32_t slot = blockIdx.x * 256u + threadIdx.x;
    __shared__ uint32_t s_cnt[5];
    __shared__ float4 s_mats[LDS_TABLES ? LOGIC_TABLE_MAX * 3 : 1];
    __shared__ uint4 s_lights[LDS_TABLES ? LOGIC_TABLE_MAX : 1];
    Tables<LDS_TABLES> tb;
    if (LDS_TABLES) {
        if (threadIdx.x < (uint32_t)P.n_mats * 3u) s_mats[threadIdx.x] = sc.mats[threadIdx.x];
        if (threadIdx.x < (uint32_t)sc.n_lights) s_lights[threadIdx.x] = sc.lights[threadIdx.x];
        tb.mats = s_mats; tb.lights = s_lights;
    } else {
        tb.mats = sc.mats; tb.lights = sc.lights;
    }
    if (threadIdx.x < 5) s_cnt[threadIdx.x] = 0;

    // ---- request the whole slot state at once ----
    const bool in_range = slot < pl.n;
    const uint32_t sl = in_range ? slot : 0;
    float4 la = pl.la[sl];
    float4 cc = pl.cc[sl], vx = pl.vx[sl], ro = pl.ro[sl], rd = pl.rd[sl], vn = pl.vn[sl];
    uint4 idv = pl.id[sl];
    float2 rs = pl.res[sl];
    __syncthreads();

    PathCounters cnt;
    cnt = PathCounters{};
    bool emitted = false;
    uint32_t st = __float_as_uint(la.w);
    const uint32_t stage = in_range ? (st >> 8) & 15u : (uint32_t)ST_DEAD;
    if (stage != ST_DEAD) {
        Lane s;
        s.depth = st & 255u; s.stage = stage; s.q = st >> 16;
        s.Ld = f3(la.x, la.y, la.z);
        s.kind = RAY_NONE;
        s.c = f3(cc.x, cc.y, cc.z);
        s.pos = f3(vx.x, vx.y, vx.z); s.vtri = __float_as_uint(vx.w);
        s.ro = f3(ro.x, ro.y, ro.z); s.tl = ro.w;
        s.rd = f3(rd.x, rd.y, rd.z);
        s.nrm = f3(vn.x, vn.y, vn.z); s.mat = __float_as_uint(vn.w);
        s.pixel_index = idv.x; s.k = idv.y; s.item = idv.z;
        emitted = logic_advance(P, tb, slot, s, stage, rs.x, __float_as_int(rs.y), cnt);

        // ---- write the slot back ----
        st = s.depth | (s.stage << 8) | (s.q << 16);
        pl.la[slot] = make_float4(s.Ld.x, s.Ld.y, s.Ld.z, __uint_as_float(st));
        if (emitted) {
            pl.ro[slot] = make_float4(s.ro.x, s.ro.y, s.ro.z, s.tl);
            pl.rd[slot] = make_float4(s.rd.x, s.rd.y, s.rd.z, __uint_as_float(s.kind));
            if (s.stage == ST_SHADOW) pl.cc[slot] = make_float4(s.c.x, s.c.y, s.c.z, 0.0f);
            if (stage != ST_SHADOW) {
                // the vertex planes only change when a result was a new vertex
                pl.vx[slot] = make_float4(s.pos.x, s.pos.y, s.pos.z, __uint_as_float(s.vtri));
                pl.vn[slot] = make_float4(s.nrm.x, s.nrm.y, s.nrm.z, __uint_as_float(s.mat));
            }
        } else {
            pl.rd[slot].w = __uint_as_float((uint32_t)RAY_NONE);
        }
    }
    // ---- counters: wave sums -> LDS -> one atomic per block and counter, on this block's shard ----
    uint32_t r = wave_sum(cnt.rays), sh = wave_sum(cnt.shadow), pr = wave_sum(cnt.probe), pa = wave_sum(cnt.paths);
    uint32_t al = wave_sum(emitted ? 1u : 0u);
    if ((threadIdx.x & 63) == 0 && (r | pa | al)) {
        atomicAdd(&s_cnt[0], r); atomicAdd(&s_cnt[1], sh); atomicAdd(&s_cnt[2], pr); atomicAdd(&s_cnt[3], pa); atomicAdd(&s_cnt[4], al);
    }
    __syncthreads();
    if (threadIdx.x < 5 && s_cnt[threadIdx.x]) {
        const int idx[5] = {C_RAYS, C_SHADOW, C_PROBE, C_PATHS, C_ALIVE};
        atomicAdd(&P.counters[(blockIdx.x & (CNT_SHARDS - 1)) * CNT_STRIDE + idx[threadIdx.x]], (unsigned long long)s_cnt[threadIdx.x]);
    }
}

__global__ __launch_bounds__(256) void k_pool_init(Pool pl)
{
    uint32_t slot = blockIdx.x * 256u + threadIdx.x;
    if (slot >= pl.n) return;
    pl.la[slot] = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float((uint32_t)ST_NEW << 8));
    if (pl.rd) pl.rd[slot] = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float((uint32_t)RAY_NONE)); // wavefront pipeline only
    if (pl.res) pl.res[slot] = make_float2(FLT_MAX, __int_as_float(-1));
}

// ---------------------------------------------------------------- trace ----
// Persistent traversal kernel.  A wave keeps 64 rays in flight; a lane whose ray is done
// writes its result and goes idle, and idle lanes are refilled together (one atomic on a
// sharded cursor per refill) as soon as REFILL_MIN of them are waiting.  Inside, inner-node
// steps and leaf steps are separate wave-wide phases: a lane that reaches a leaf parks until
// LEAF_MIN lanes hold one (or nobody has inner work left), so both phases run with most lanes
// active instead of serialising the two bodies on every iteration.
#define TR_IDLE 0
#define TR_INNER 1
#define TR_LEAF 2
#define TR_POP 3

struct TravLane {
    RayT r;
    float t_limit, best_t, bound;
    int32_t best_tri, best_leaf, ref, sp;
    uint32_t slot;
    bool any_hit, nx, ny, nz;
};

// Traversal stack: entry = (node ref, t_enter).  The first `cap` levels live in LDS
// ([level][thread] int2, conflict free), deeper ones spill to a per-lane global area.
struct TravStack {
    int2* lds;        // + threadIdx.x
    int2* spill;      // + global lane
    uint32_t spill_stride;
    int cap;
};
__device__ __forceinline__ void trav_push(const TravStack& S, int sp, int ref, float t)
{
    int2 e = make_int2(ref, __float_as_int(t));
    if (sp < S.cap) S.lds[sp * 256] = e;                               // ds_write_b64
    else S.spill[(size_t)(sp - S.cap) * S.spill_stride] = e;
}
// Pops ONE entry.  Returns 0 = stack empty, 1 = popped a node to visit (L.ref), 2 = the popped
// entry lies beyond the pruning bound (the lane pops again on its next turn, so that a wave
// never serialises a chain of dependent LDS reads inside one step).
template <int MODE>
__device__ __forceinline__ int trav_pop(TravLane& L, const TravStack& S)
{
    if (L.sp == 0) return 0;
    L.sp--;
    int2 e;
    if (L.sp < S.cap) e = S.lds[L.sp * 256];                           // ds_read_b64
    else e = S.spill[(size_t)(L.sp - S.cap) * S.spill_stride];
    L.ref = e.x;
    if (MODE == 0 && __int_as_float(e.y) > L.bound) return 2;
    return 1;
}

// One wave-wide traversal step: pop phase, then either the leaf phase (when enough lanes hold a
// leaf, or nobody has inner work) or the inner-node phase.  Returns true in lanes whose ray is
// finished (result in L.best_t / L.best_tri); nothing_to_do = no lane had any traversal work.
// The bodies are written with selects instead of nested branches: every divergent `if` costs
// several scalar instructions of exec-mask bookkeeping, and rocprof shows the scalar unit almost
// as busy as the vector units in this kernel.
template <int MODE, bool STATS>
__device__ __forceinline__ bool trav_step(const DevScene& sc, TravLane& L, int& state, const TravStack& S, const int leaf_min,
                                          TravCounters& cnt, uint32_t& ray_sp, bool& nothing_to_do)
{
    bool finished = false;
    // ---- pop phase: lanes whose subtree is exhausted take the next pending node ----
    if (state == TR_POP) {
        int r = trav_pop<MODE>(L, S);
        finished = r == 0;
        state = r == 1 ? (L.ref >= 0 ? TR_INNER : TR_LEAF) : TR_POP;
    }
    const int n_inner = __popcll(__ballot(state == TR_INNER));
    const int n_leaf = __popcll(__ballot(state == TR_LEAF));
    const int n_pop = __popcll(__ballot(state == TR_POP && !finished));
    nothing_to_do = n_inner == 0 && n_leaf == 0 && n_pop == 0 && __ballot(finished) == 0;
    if (nothing_to_do) return false;
    bool need_pop = false;
    if (n_leaf > 0 && (n_leaf >= leaf_min || n_inner == 0)) {
        // ---- leaf phase ----
        if (state == TR_LEAF) {
            uint32_t code = (uint32_t)~L.ref;
            const int it = (int)(code >> 4);
            int n = (int)(code & 15u);
            if (n == 0) n = sc.leaf_count[it];
            // the first two triangles in straight-line code (the reference's bvh_thresh_n = 2 gives 1-2 per leaf):
            // both fetches are in flight together, nothing branches
            const bool two = n > 1;
            float t0, t1;
            const bool a0 = tri_test(sc, it, L.r, t0);
            const bool a1 = tri_test(sc, two ? it + 1 : it, L.r, t1) && two;
            if (STATS) { cnt.leaf++; cnt.tests += two ? 2u : 1u; }
            bool done = false;
            if (L.any_hit) {
                const bool b0 = a0 && (L.t_limit - t0 > CRT_EPSILON);
                const bool b1 = a1 && (L.t_limit - t1 > CRT_EPSILON);
                done = b0 || b1;
                L.best_t = b0 ? t0 : (b1 ? t1 : L.best_t);
                L.best_tri = b0 ? it : (b1 ? it + 1 : L.best_tri);
            } else {
                // ascending index, strict <: the first of equal t inside a leaf wins (DeviceBVH.cuh:34-41); across leaves the
                // larger leaf start wins (reference visit order, see crt_trace.h)
                const bool w0 = a0 && (t0 < L.best_t || (t0 == L.best_t && it > L.best_leaf));
                L.best_t = w0 ? t0 : L.best_t; L.best_tri = w0 ? it : L.best_tri; L.best_leaf = w0 ? it : L.best_leaf;
                const bool w1 = a1 && (t1 < L.best_t || (t1 == L.best_t && it > L.best_leaf));
                L.best_t = w1 ? t1 : L.best_t; L.best_tri = w1 ? it + 1 : L.best_tri; L.best_leaf = w1 ? it : L.best_leaf;
                if (MODE == 0) L.bound = (w0 || w1) ? prune_bound(L.best_t, L.r.o, L.r.inv) : L.bound;
            }
            for (int i = it + 2; i < it + n && !done; i++) { // only with bvh_thresh_n > 2
                if (STATS) cnt.tests++;
                float t;
                if (tri_test(sc, i, L.r, t)) {
                    if (L.any_hit) {
                        if (L.t_limit - t > CRT_EPSILON) { L.best_t = t; L.best_tri = i; done = true; }
                    } else if (t < L.best_t || (t == L.best_t && it > L.best_leaf)) {
                        L.best_t = t; L.best_tri = i; L.best_leaf = it;
                        if (MODE == 0) L.bound = prune_bound(t, L.r.o, L.r.inv);
                    }
                }
            }
            finished = done;
            need_pop = !done;
        }
    } else if (n_inner > 0) {
        // ---- inner phase ----
        if (state == TR_INNER) {
            if (STATS) cnt.inner++;
            const float4* n = sc.nodes + (size_t)L.ref * 4;
            float4 a = n[0], b = n[1], c = n[2], d = n[3];
            float tl, tr;
            bool hl = slab_test(a, b, L.r, L.nx, L.ny, L.nz, tl);
            bool hr = slab_test(c, d, L.r, L.nx, L.ny, L.nz, tr);
            const int lref = __float_as_int(a.w), rref = __float_as_int(b.w);
            bool left_first;
            if (MODE == 1) {
                left_first = false; // push lc, visit rc first (DeviceBVH.cuh:154-166)
            } else {
                hl = hl && !(tl > L.bound);
                hr = hr && !(tr > L.bound);
                left_first = tl <= tr;
            }
            const bool both = hl && hr, any = hl || hr;
            const int near_ref = both ? (left_first ? lref : rref) : (hl ? lref : rref);
            if (both) {
                trav_push(S, L.sp, left_first ? rref : lref, left_first ? tr : tl);
                L.sp++;
                if (STATS && (uint32_t)L.sp > ray_sp) ray_sp = (uint32_t)L.sp;
            }
            L.ref = any ? near_ref : L.ref;
            state = any ? (near_ref >= 0 ? TR_INNER : TR_LEAF) : state;
            need_pop = !any;
        }
    }
    if (need_pop) {
        int r = trav_pop<MODE>(L, S);
        finished = r == 0;
        state = r == 1 ? (L.ref >= 0 ? TR_INNER : TR_LEAF) : TR_POP;
    }
    return finished;
}

// Starts the traversal of the ray in L.r (origin, direction): returns the lane's new state, TR_IDLE
// if the answer is known without traversal (result already in L.best_t / L.best_tri).
template <int MODE>
__device__ __forceinline__ int trav_begin(const DevScene& sc, TravLane& L, uint32_t kind, float t_limit)
{
    L.r.inv = f3(1 / L.r.d.x, 1 / L.r.d.y, 1 / L.r.d.z); // Ray.cuh:14
    L.nx = L.r.d.x < 0; L.ny = L.r.d.y < 0; L.nz = L.r.d.z < 0;
    L.t_limit = t_limit;
    // REFERENCE mode resolves shadow rays with the full closest-hit query, as blocked() does
    L.any_hit = MODE != 1 && kind == RAY_SHADOW;
    L.best_t = FLT_MAX; L.best_tri = -1; L.best_leaf = -1;
    L.bound = pinf(); // (no bound: a box entered at +inf is still a box the reference enters)
    L.sp = 0;
    // rays with a zero / denormal direction component (inv_dir not finite) can put NaNs into the
    // slab test; they walk the reference topology, whose box tests are the reference's own (crt_accel.h)
    // (the same predicate as k_mega3's start_ray: a non-finite ORIGIN puts NaNs into the min / max form of the slab test too)
    const bool finite_inv = absf(L.r.inv.x) <= FLT_MAX && absf(L.r.inv.y) <= FLT_MAX && absf(L.r.inv.z) <= FLT_MAX;
    const bool finite_o = absf(L.r.o.x) <= FLT_MAX && absf(L.r.o.y) <= FLT_MAX && absf(L.r.o.z) <= FLT_MAX;
    const bool finite_d = absf(L.r.d.x) <= FLT_MAX && absf(L.r.d.y) <= FLT_MAX && absf(L.r.d.z) <= FLT_MAX; // (1/d != 0)
    L.ref = (MODE != 1 && finite_inv && finite_o && finite_d) ? sc.root_fast : sc.root_exact;
    if (L.any_hit) {
        // a NaN or -inf limit can never be "blocked"; +inf is blocked by any hit
        if (!(L.t_limit == L.t_limit) || L.t_limit == -pinf()) return TR_IDLE;
        if (MODE == 0) L.bound = prune_bound(L.t_limit, L.r.o, L.r.inv); // (MODE 2, CRT_TRAVERSAL_EXACT: ordered and any-hit, never pruned)
    }
    return L.ref >= 0 ? TR_INNER : TR_LEAF;
}

template <int MODE, bool STATS>
__global__ __launch_bounds__(256) void k_trace(const TParams T)
{
    extern __shared__ int2 s_lds2[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    TravStack S;
    S.lds = s_lds2 + tid;
    S.spill = T.spill + (size_t)blockIdx.x * 256u + tid;
    S.spill_stride = T.spill_stride;
    S.cap = T.stack_cap;
    const Pool& pl = T.pool;
    const DevScene& sc = T.sc;
    TravCounters cnt;
    cnt.inner = cnt.leaf = cnt.tests = cnt.hits = 0;
    uint32_t max_sp = 0, sum_sp = 0, ray_sp = 0;

    const uint32_t per = T.slots_per_shard;
    uint32_t shard_off = 0; // shards tried so far by this wave (wave-uniform)
    const uint32_t home = (blockIdx.x * 4u + (uint32_t)(tid >> 6)) & (SLOT_SHARDS - 1);
    bool exhausted = false;
    int state = TR_IDLE;
    TravLane L;
    L.slot = 0; L.ref = 0; L.sp = 0; L.best_tri = -1; L.best_leaf = -1; L.best_t = FLT_MAX; L.bound = FLT_MAX; L.t_limit = 0.0f;
    L.any_hit = false; L.nx = L.ny = L.nz = false;

    for (;;) {
        // ---- refill idle lanes ----
        const unsigned long long idle = __ballot(state == TR_IDLE);
        const int n_idle = __popcll(idle);
        if (!exhausted && (n_idle >= T.refill_min)) {
            uint32_t my = 0xffffffffu;
            if (state == TR_IDLE) {
                // all idle lanes are active here; take indices shard by shard
                while (shard_off < SLOT_SHARDS) {
                    const uint32_t sh = (home + shard_off) & (SLOT_SHARDS - 1);
                    const uint32_t lo = sh * per;
                    const uint32_t hi = min(lo + per, pl.n);
                    unsigned int* cur = T.slot_next + sh * SLOT_STRIDE;
                    if (lo >= pl.n || lo + __hip_atomic_load(cur, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= hi) { shard_off++; continue; }
                    const unsigned long long m = __ballot(my == 0xffffffffu);
                    if (m == 0) break;
                    if (my == 0xffffffffu) {
                        const int leader = __ffsll((long long)m) - 1;
                        const uint32_t rank = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                        unsigned int base = 0;
                        if (lane == leader) base = atomicAdd(cur, (unsigned int)__popcll(m));
                        base = (unsigned int)__builtin_amdgcn_readfirstlane((int)base);
                        const unsigned long long idx = (unsigned long long)lo + base + rank;
                        if (idx < hi) my = (uint32_t)idx;
                    }
                    if (__ballot(my == 0xffffffffu) == 0) break; // every idle lane served
                    shard_off++;                                  // this shard ran dry
                }
            }
            // shard_off is advanced by the idle lanes only; make it wave-uniform
            {
                uint32_t so = shard_off;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) so = max(so, (uint32_t)__shfl_xor((int)so, o, 64));
                shard_off = so;
            }
            if (shard_off >= SLOT_SHARDS) exhausted = true;
            if (state == TR_IDLE && my != 0xffffffffu) {
                float4 rd = pl.rd[my];
                uint32_t kind = __float_as_uint(rd.w);
                if (kind != RAY_NONE) {
                    float4 ro = pl.ro[my];
                    L.slot = my;
                    L.r.o = f3(ro.x, ro.y, ro.z);
                    L.r.d = f3(rd.x, rd.y, rd.z);
                    state = trav_begin<MODE>(sc, L, kind, ro.w);
                    if (state == TR_IDLE) pl.res[my] = make_float2(FLT_MAX, __int_as_float(-1));
                }
            }
        }
        bool nothing_to_do = false;
        const bool finished = trav_step<MODE, STATS>(sc, L, state, S, T.leaf_min, cnt, ray_sp, nothing_to_do);
        if (nothing_to_do) {
            if (exhausted) break;
            continue; // every lane is idle: the next iteration refills
        }
        if (finished) {
            if (STATS) {
                if (L.best_tri >= 0) cnt.hits++;
                if (ray_sp > max_sp) max_sp = ray_sp;
                sum_sp += ray_sp;
                ray_sp = 0;
            }
            pl.res[L.slot] = make_float2(L.best_t, __int_as_float(L.best_tri));
            state = TR_IDLE;
        }
    }
    if (STATS) {
        uint32_t a = wave_sum(cnt.inner), b = wave_sum(cnt.leaf), c = wave_sum(cnt.tests), d = wave_sum(cnt.hits);
        if (lane == 0 && (a | b)) {
            unsigned long long* cs = T.counters + (blockIdx.x & (CNT_SHARDS - 1)) * CNT_STRIDE;
            atomicAdd(&cs[C_INNER], (unsigned long long)a);
            atomicAdd(&cs[C_LEAF], (unsigned long long)b);
            atomicAdd(&cs[C_TESTS], (unsigned long long)c);
            atomicAdd(&cs[C_HITS], (unsigned long long)d);
        }
        uint32_t ss = wave_sum(sum_sp);
        uint32_t ms = max_sp;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) ms = max(ms, (uint32_t)__shfl_xor((int)ms, o, 64));
        if (lane == 0) {
            unsigned long long* cs = T.counters + (blockIdx.x & (CNT_SHARDS - 1)) * CNT_STRIDE;
            atomicAdd(&cs[C_SUMSP], (unsigned long long)ss);
            atomicMax(&cs[C_MAXSP], (unsigned long long)ms);
        }
    }
}


// ---- exported to crt_render.hip ----
void launch_pool_init(uint32_t blocks, hipStream_t st, const Pool& pool)
{
    hipLaunchKernelGGL(k_pool_init, dim3(blocks), dim3(256), 0, st, pool);
}
void launch_logic(bool lds_tables, uint32_t blocks, hipStream_t st, const LParams& P)
{
    if (lds_tables) hipLaunchKernelGGL(k_logic<true>, dim3(blocks), dim3(256), 0, st, P);
    else hipLaunchKernelGGL(k_logic<false>, dim3(blocks), dim3(256), 0, st, P);
}
template <int MODE, bool STATS> static int blocks_per_cu_(size_t lds)
{
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_trace<MODE, STATS>, 256, lds) != hipSuccess || nb < 1) nb = 1;
    return nb;
}
// mode_id = (REFERENCE ? 2 : EXACT ? 4 : 0) + (counters ? 1 : 0)
int trace_blocks_per_cu(int mode_id, size_t lds)
{
    return mode_id == 5 ? blocks_per_cu_<2, true>(lds) : mode_id == 4 ? blocks_per_cu_<2, false>(lds) : mode_id == 3 ? blocks_per_cu_<1, true>(lds)
         : mode_id == 2 ? blocks_per_cu_<1, false>(lds) : mode_id == 1 ? blocks_per_cu_<0, true>(lds) : blocks_per_cu_<0, false>(lds);
}
void launch_trace(int mode_id, const TParams& T, uint32_t blocks, size_t lds, hipStream_t st)
{
    if (mode_id == 5) hipLaunchKernelGGL((k_trace<2, true>), dim3(blocks), dim3(256), lds, st, T);
    else if (mode_id == 4) hipLaunchKernelGGL((k_trace<2, false>), dim3(blocks), dim3(256), lds, st, T);
    else if (mode_id == 3) hipLaunchKernelGGL((k_trace<1, true>), dim3(blocks), dim3(256), lds, st, T);
    else if (mode_id == 2) hipLaunchKernelGGL((k_trace<1, false>), dim3(blocks), dim3(256), lds, st, T);
    else if (mode_id == 1) hipLaunchKernelGGL((k_trace<0, true>), dim3(blocks), dim3(256), lds, st, T);
    else hipLaunchKernelGGL((k_trace<0, false>), dim3(blocks), dim3(256), lds, st, T);
}

} // namespace crtk

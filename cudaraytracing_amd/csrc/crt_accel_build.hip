// cudaraytracing_amd/csrc/crt_accel_build.hip -- the binned-SAH tree of crt_accel.h built on the GPU.
//
// crt_accel.h explains why a second tree over the reference's LEAVES is legal (a leaf is entered iff its own box passes, whatever
// hierarchy stands above the leaves).  Its host builder costs 32 ms for the 24 588 leaves of the Cornell stand-in and 153 ms for
// 115 000 -- more than the reference BVH itself.  This file runs the same algorithm level by level on the device: one workgroup per
// range of a level; centroid bounds, the 3 x 32 bins (boxes as order-preserving integers, LDS atomics), the sweep in double
// precision, a stable partition by block scans, the children's boxes as unions of bins.  Every quantity the host builder derives
// from a range is independent of the order of the leaves inside it (bounds, bins, the cheapest (axis, bin) in the same scan order,
// partition MEMBERSHIP, box unions), so the tree is the host builder's tree, node for node -- except where the host splits a range
// "by index" (all centroids coincide: duplicate leaves), where the two builders may hand the equal leaves to different sides.
#include "crt_accel.h"

#include <algorithm>
#include <chrono>
#include <cstring>
#include <string.h>
#include <vector>

#include <hip/hip_runtime.h>

namespace crtaccel {

namespace {

struct AErr {
    hipError_t e;
};
#define AHIP(call)                                  \
    do {                                            \
        hipError_t e_ = (call);                     \
        if (e_ != hipSuccess) throw AErr{e_};       \
    } while (0)

template <typename T> struct ABuf {
    T* p = nullptr;
    void alloc(size_t n) { AHIP(hipMalloc((void**)&p, std::max<size_t>(1, n) * sizeof(T))); }
    ~ABuf() { if (p) (void)hipFree(p); }
};

__device__ inline uint32_t a_ord(float f)
{
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ inline float a_unord(uint32_t k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }

#define SAH_NB 32

struct DSeg {
    int32_t b, e, node, depth;
};
struct DTmp {              // crt_accel.h: Tmp
    float lo[2][3], hi[2][3];
    int32_t child[2];
};

__device__ inline double half_area(const float lo[3], const float hi[3])
{
    const double dx = (double)hi[0] - lo[0], dy = (double)hi[1] - lo[1], dz = (double)hi[2] - lo[2];
    if (dx < 0 || dy < 0 || dz < 0) return 0.0;
    return dx * dy + dy * dz + dz * dx;
}

__device__ inline uint32_t wave_min_u(uint32_t v) { for (int o = 32; o > 0; o >>= 1) v = min(v, (uint32_t)__shfl_xor((int)v, o, 64)); return v; }
__device__ inline uint32_t wave_max_u(uint32_t v) { for (int o = 32; o > 0; o >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, o, 64)); return v; }

__device__ inline uint32_t a_block_rank(bool flag, uint32_t* s_wave, uint32_t& total)
{
    const unsigned long long m = __ballot(flag);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t in_wave = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    __syncthreads();
    if (lane == 0) s_wave[w] = (uint32_t)__popcll(m);
    __syncthreads();
    uint32_t base = 0;
    for (int k = 0; k < w; k++) base += s_wave[k];
    total = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
    return base + in_wave;
}

// one level of the top-down build: crt_accel.h build_sah's loop body for every range of the level
__global__ __launch_bounds__(256) void k_sah_level(const DSeg* segs, const uint32_t* n_segs, DSeg* next, uint32_t* n_next, const uint32_t* order_in, uint32_t* order_out,
                                                   const float4* plo, const float4* phi, DTmp* tmp, uint32_t* n_tmp, uint32_t* max_depth)
{
    __shared__ uint32_t s_clo[3], s_chi[3];
    __shared__ uint32_t s_blo[3][SAH_NB][3], s_bhi[3][SAH_NB][3], s_cnt[3][SAH_NB];
    __shared__ double s_cost[3];
    __shared__ int s_split[3];
    __shared__ int s_axis, s_best_split, s_left;
    __shared__ uint32_t s_wave[4];
    __shared__ uint32_t s_box[2][6];
    const uint32_t ns = *n_segs;
    for (uint32_t si = blockIdx.x; si < ns; si += gridDim.x) {
        const DSeg g = segs[si];
        const int b = g.b, e = g.e, cnt = e - b;
        // ---- centroid bounds and bins ----
        for (int k = threadIdx.x; k < 3; k += 256) { s_clo[k] = 0xffffffffu; s_chi[k] = 0u; }
        for (int k = threadIdx.x; k < 3 * SAH_NB; k += 256) {
            const int a = k / SAH_NB, q = k % SAH_NB;
            s_cnt[a][q] = 0;
            for (int c = 0; c < 3; c++) { s_blo[a][q][c] = 0xffffffffu; s_bhi[a][q][c] = 0u; }
        }
        __syncthreads();
        {   // (thread-local, then wave-level reduction: one LDS atomic per wave and bound instead of one per leaf)
            uint32_t mn[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu}, mx[3] = {0u, 0u, 0u};
            for (int i = b + (int)threadIdx.x; i < e; i += 256) {
                const uint32_t p = order_in[i];
                const float4 lo = plo[p], hi = phi[p];
                const float c[3] = {0.5f * (lo.x + hi.x), 0.5f * (lo.y + hi.y), 0.5f * (lo.z + hi.z)};
                for (int a = 0; a < 3; a++) { mn[a] = min(mn[a], a_ord(c[a])); mx[a] = max(mx[a], a_ord(c[a])); }
            }
            for (int a = 0; a < 3; a++) {
                const uint32_t wmn = wave_min_u(mn[a]), wmx = wave_max_u(mx[a]);
                if ((threadIdx.x & 63) == 0) { atomicMin(&s_clo[a], wmn); atomicMax(&s_chi[a], wmx); }
            }
        }
        __syncthreads();
        float clo[3], chi[3];
        for (int a = 0; a < 3; a++) { clo[a] = a_unord(s_clo[a]); chi[a] = a_unord(s_chi[a]); }
        for (int i = b + (int)threadIdx.x; i < e; i += 256) {
            const uint32_t p = order_in[i];
            const float4 lo = plo[p], hi = phi[p];
            const float c[3] = {0.5f * (lo.x + hi.x), 0.5f * (lo.y + hi.y), 0.5f * (lo.z + hi.z)};
            const float l3[3] = {lo.x, lo.y, lo.z}, h3[3] = {hi.x, hi.y, hi.z};
            for (int a = 0; a < 3; a++) {
                const float ext = chi[a] - clo[a];
                if (!(ext > 0.0f)) continue;
                const float scale = (float)SAH_NB / ext;
                const int k = min(SAH_NB - 1, max(0, (int)((c[a] - clo[a]) * scale)));
                atomicAdd(&s_cnt[a][k], 1u);
                for (int q = 0; q < 3; q++) { atomicMin(&s_blo[a][k][q], a_ord(l3[q])); atomicMax(&s_bhi[a][k][q], a_ord(h3[q])); }
            }
        }
        __syncthreads();
        // ---- the sweep: cheapest (axis, bin), axes and bins in the host builder's scan order, strict < ----
        if (threadIdx.x < 3) {
            const int a = threadIdx.x;
            double best = 1.7976931348623157e308;
            int best_k = -1;
            const float ext = chi[a] - clo[a];
            if (ext > 0.0f) {
                double right_area[SAH_NB];
                int right_cnt[SAH_NB];
                float alo[3] = {3.402823466e38f, 3.402823466e38f, 3.402823466e38f}, ahi[3] = {-3.402823466e38f, -3.402823466e38f, -3.402823466e38f};
                int c = 0;
                for (int k = SAH_NB - 1; k >= 1; k--) {
                    if (s_cnt[a][k]) for (int q = 0; q < 3; q++) { alo[q] = fminf(alo[q], a_unord(s_blo[a][k][q])); ahi[q] = fmaxf(ahi[q], a_unord(s_bhi[a][k][q])); }
                    c += (int)s_cnt[a][k];
                    right_area[k] = half_area(alo, ahi);
                    right_cnt[k] = c;
                }
                for (int q = 0; q < 3; q++) { alo[q] = 3.402823466e38f; ahi[q] = -3.402823466e38f; }
                c = 0;
                for (int k = 0; k < SAH_NB - 1; k++) {
                    if (s_cnt[a][k]) for (int q = 0; q < 3; q++) { alo[q] = fminf(alo[q], a_unord(s_blo[a][k][q])); ahi[q] = fmaxf(ahi[q], a_unord(s_bhi[a][k][q])); }
                    c += (int)s_cnt[a][k];
                    if (c == 0 || right_cnt[k + 1] == 0) continue;
                    const double cost = half_area(alo, ahi) * c + right_area[k + 1] * right_cnt[k + 1];
                    if (cost < best) { best = cost; best_k = k; }
                }
            }
            s_cost[a] = best;
            s_split[a] = best_k;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            int ba = -1, bk = -1;
            double bc = 1.7976931348623157e308;
            for (int a = 0; a < 3; a++)
                if (s_split[a] >= 0 && s_cost[a] < bc) { bc = s_cost[a]; ba = a; bk = s_split[a]; }
            int left = 0;
            if (ba >= 0) for (int k = 0; k <= bk; k++) left += (int)s_cnt[ba][k];
            s_axis = ba; s_best_split = bk;
            s_left = ba >= 0 ? left : cnt / 2; // all centroids coincide: split by index
            if (ba < 0) atomicAdd(max_depth + 1, 1u);
            for (int q = 0; q < 6; q++) { s_box[0][q] = q < 3 ? 0xffffffffu : 0u; s_box[1][q] = q < 3 ? 0xffffffffu : 0u; }
        }
        __syncthreads();
        const int axis = s_axis, split = s_best_split, mid = b + s_left;
        // ---- stable partition into order_out, children's boxes ----
        {
            float scale = 0.0f;
            if (axis >= 0) scale = (float)SAH_NB / (chi[axis] - clo[axis]);
            uint32_t run_l = 0, run_r = 0;
            uint32_t bl[2][6];
            for (int s2 = 0; s2 < 2; s2++) for (int q = 0; q < 6; q++) bl[s2][q] = q < 3 ? 0xffffffffu : 0u;
            for (int base = b; base < e; base += 256) {
                const int i = base + (int)threadIdx.x;
                const bool in = i < e;
                uint32_t p = 0;
                bool left = false;
                float4 lo = make_float4(0, 0, 0, 0), hi = lo;
                if (in) {
                    p = order_in[i];
                    lo = plo[p]; hi = phi[p];
                    if (axis >= 0) {
                        const float cl = axis == 0 ? lo.x : (axis == 1 ? lo.y : lo.z), ch = axis == 0 ? hi.x : (axis == 1 ? hi.y : hi.z);
                        const float c = 0.5f * (cl + ch);
                        const int k = min(SAH_NB - 1, max(0, (int)((c - clo[axis]) * scale)));
                        left = k <= split;
                    } else left = i < mid;
                }
                uint32_t tot_l, tot_r;
                const uint32_t rl = a_block_rank(in && left, s_wave, tot_l);
                const uint32_t rr = a_block_rank(in && !left, s_wave, tot_r);
                if (in) {
                    order_out[left ? b + run_l + rl : mid + run_r + rr] = p;
                    uint32_t* bx = bl[left ? 0 : 1];
                    bx[0] = min(bx[0], a_ord(lo.x)); bx[1] = min(bx[1], a_ord(lo.y)); bx[2] = min(bx[2], a_ord(lo.z));
                    bx[3] = max(bx[3], a_ord(hi.x)); bx[4] = max(bx[4], a_ord(hi.y)); bx[5] = max(bx[5], a_ord(hi.z));
                }
                run_l += tot_l; run_r += tot_r;
            }
            for (int s2 = 0; s2 < 2; s2++)
                for (int q = 0; q < 6; q++) {
                    const uint32_t w = q < 3 ? wave_min_u(bl[s2][q]) : wave_max_u(bl[s2][q]);
                    if ((threadIdx.x & 63) == 0) { if (q < 3) atomicMin(&s_box[s2][q], w); else atomicMax(&s_box[s2][q], w); }
                }
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            DTmp& t = tmp[g.node];
            const int halves[2][2] = {{b, mid}, {mid, e}};
            for (int s = 0; s < 2; s++) {
                for (int q = 0; q < 3; q++) { t.lo[s][q] = a_unord(s_box[s][q]); t.hi[s][q] = a_unord(s_box[s][3 + q]); }
                const int hb = halves[s][0], he = halves[s][1];
                if (he - hb == 1) {
                    t.child[s] = __float_as_int(plo[order_out[hb]].w);
                    atomicMax(max_depth, (uint32_t)(g.depth + 1));
                } else {
                    const uint32_t ci = atomicAdd(n_tmp, 1u);
                    t.child[s] = (int32_t)ci;
                    atomicMax(max_depth, (uint32_t)(g.depth + 2));
                    next[atomicAdd(n_next, 1u)] = DSeg{hb, he, (int32_t)ci, g.depth + 1};
                }
            }
        }
        __syncthreads();
    }
}

} // namespace

// Returns the tree depth (as build_sah), or -1 when the device build could not run (the caller falls back to build_sah).
int build_sah_device(const std::vector<Prim>& prims, std::vector<Node>& nodes, int32_t& root_ref, float* device_ms, uint32_t* index_splits)
{
    if (index_splits) *index_splits = 0;
    nodes.clear();
    const int n = (int)prims.size();
    if (n == 1) { root_ref = prims[0].ref; return 1; }
    const bool timing = std::getenv("CRT_SAH_TIMING") != nullptr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms_since = [&](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(now() - t).count(); };
    auto t_sec = now();
    auto section = [&](const char* what) {
        if (timing) { std::fprintf(stderr, "[sah] %-10s %8.3f ms\n", what, ms_since(t_sec)); }
        t_sec = now();
    };
    try {
        std::vector<float4> lo(n), hi(n);
        for (int i = 0; i < n; i++) {
            float rf;
            std::memcpy(&rf, &prims[i].ref, 4);
            lo[i] = make_float4(prims[i].box.lo[0], prims[i].box.lo[1], prims[i].box.lo[2], rf);
            hi[i] = make_float4(prims[i].box.hi[0], prims[i].box.hi[1], prims[i].box.hi[2], 0.0f);
        }
        section("pack");
        ABuf<float4> d_lo, d_hi;
        ABuf<uint32_t> d_order[2], d_cnt; // d_cnt: [0], [1] ranges of the two level buffers, [2] tmp nodes, [3] depth, [4] splits by index
        ABuf<DSeg> d_segs[2];
        ABuf<DTmp> d_tmp;
        d_lo.alloc(n); d_hi.alloc(n); d_order[0].alloc(n); d_order[1].alloc(n); d_cnt.alloc(5);
        d_segs[0].alloc(n); d_segs[1].alloc(n); d_tmp.alloc(n);
        section("alloc");
        AHIP(hipMemcpy(d_lo.p, lo.data(), (size_t)n * 16, hipMemcpyHostToDevice));
        AHIP(hipMemcpy(d_hi.p, hi.data(), (size_t)n * 16, hipMemcpyHostToDevice));
        {
            std::vector<uint32_t> iota(n);
            for (int i = 0; i < n; i++) iota[i] = (uint32_t)i;
            AHIP(hipMemcpy(d_order[0].p, iota.data(), (size_t)n * 4, hipMemcpyHostToDevice));
            AHIP(hipMemcpy(d_order[1].p, iota.data(), (size_t)n * 4, hipMemcpyHostToDevice));
            const DSeg root{0, n, 0, 1};
            AHIP(hipMemcpy(d_segs[0].p, &root, sizeof(root), hipMemcpyHostToDevice));
            const uint32_t init[5] = {1u /* segments of level 0 */, 0u, 1u /* tmp nodes */, 1u /* depth */, 0u};
            AHIP(hipMemcpy(d_cnt.p, init, sizeof(init), hipMemcpyHostToDevice));
        }
        section("upload");
        hipEvent_t e0, e1;
        AHIP(hipEventCreate(&e0)); AHIP(hipEventCreate(&e1));
        AHIP(hipEventRecord(e0, nullptr));
        int cur = 0;
        bool finished = false;
        for (int level = 0; level < 4096; level++) {
            uint32_t ns = 0;
            AHIP(hipMemcpy(&ns, d_cnt.p + cur, 4, hipMemcpyDeviceToHost));
            if (ns == 0) { finished = true; break; }
            AHIP(hipMemsetAsync(d_cnt.p + (cur ^ 1), 0, 4, nullptr));
            // ranges of a level are disjoint, and a range that is not split any further has no element left in a live range: the
            // order of finished parts is never read again, so order_in / order_out simply alternate
            hipLaunchKernelGGL(k_sah_level, dim3(std::min<uint32_t>(ns, 2048u)), dim3(256), 0, nullptr, d_segs[cur].p, d_cnt.p + cur, d_segs[cur ^ 1].p,
                               d_cnt.p + (cur ^ 1), d_order[cur].p, d_order[cur ^ 1].p, d_lo.p, d_hi.p, d_tmp.p, d_cnt.p + 2, d_cnt.p + 3);
            AHIP(hipGetLastError());
            cur ^= 1;
        }
        AHIP(hipEventRecord(e1, nullptr));
        AHIP(hipDeviceSynchronize());
        section("levels");
        if (device_ms) AHIP(hipEventElapsedTime(device_ms, e0, e1));
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
        uint32_t fin[5];
        AHIP(hipMemcpy(fin, d_cnt.p, sizeof(fin), hipMemcpyDeviceToHost));
        if (index_splits) *index_splits = fin[4];
        if (!finished) return -1; // (did not finish within the level cap)
        const uint32_t n_tmp = fin[2];
        std::vector<DTmp> tmp(n_tmp);
        AHIP(hipMemcpy(tmp.data(), d_tmp.p, (size_t)n_tmp * sizeof(DTmp), hipMemcpyDeviceToHost));
        // breadth-first renumbering (as build_sah)
        std::vector<int> order, index(n_tmp, -1);
        order.push_back(0);
        index[0] = 0;
        for (size_t q = 0; q < order.size(); q++)
            for (int s = 0; s < 2; s++) {
                const int c = tmp[order[q]].child[s];
                if (c >= 0) { index[c] = (int)order.size(); order.push_back(c); }
            }
        nodes.resize(order.size());
        for (size_t q = 0; q < order.size(); q++) {
            const DTmp& t = tmp[order[q]];
            for (int s = 0; s < 2; s++) {
                for (int a = 0; a < 3; a++) { nodes[q].box[s].lo[a] = t.lo[s][a]; nodes[q].box[s].hi[a] = t.hi[s][a]; }
                nodes[q].child[s] = t.child[s] >= 0 ? index[t.child[s]] : t.child[s];
            }
        }
        root_ref = 0;
        section("download");
        return (int)fin[3];
    } catch (const AErr&) {
        (void)hipGetLastError();
        return -1;
    } catch (const std::bad_alloc&) {
        return -1;
    }
}

} // namespace crtaccel

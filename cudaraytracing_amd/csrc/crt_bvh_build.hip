// cudaraytracing_amd/csrc/crt_bvh_build.hip -- the reference's BVH built on the GPU.
//
// reference: include/BVH.h:37-84 -- recursive median split: box of the range, leaf iff n <= thresh_n, else std::sort of the
// range by centroid along the longest axis (ties x >= y >= z), split at mid = (l + r) / 2, nodes appended in post-order.
//
// Two facts make a level-synchronous device build byte-identical to that recursion:
//   * the SHAPE of the tree depends on the triangle count alone (left child floor(n / 2), right child ceil(n / 2) triangles), so
//     every range, its node index in the post-order array and its children's indices are known before a triangle is touched --
//     the host lays the levels out (a few thousand integers) and the device fills them in;
//   * a range whose sort keys are pairwise DIFFERENT has exactly one sorted order, whatever the sorting algorithm: a stable
//     LSD radix sort (rocPRIM) on (range start, order-preserving bits of the float key) sorts every range of a level at once
//     and reproduces std::sort.  A range that does contain two equal keys is where std::sort's result depends on its algorithm
//     (introsort is not stable): such a range is flagged, put back into its pre-sort order and left, with everything below it,
//     to the host builder, which starts from precisely the order the reference recursion would see there (all ancestors were
//     tie-free, hence unique).  Scenes with -0.0 or non-finite coordinates (where std::min / std::max and the comparator depend
//     on scan order) are not started on the device at all.
// Per level: range boxes (wave-aggregated atomic min / max on order-preserving integer keys), node records, 64-bit sort keys,
// one radix sort of the whole index array (elements outside the level's sorting ranges carry their position as key and stay
// put), tie detection, restore of flagged ranges.
#include "crt_bvh_build.h"

#include <algorithm>
#include <chrono>
#include <cstring>
#include <string.h>
#include <string>
#include <vector>

#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>

extern "C" void crt_set_last_error_(const char* msg);

namespace {

struct BErr {
    hipError_t e;
    const char* what;
};
#define BHIP(call)                                          \
    do {                                                    \
        hipError_t e_ = (call);                             \
        if (e_ != hipSuccess) throw BErr{e_, #call};        \
    } while (0)

template <typename T> struct Buf {
    T* p = nullptr;
    void alloc(size_t n) { BHIP(hipMalloc((void**)&p, std::max<size_t>(1, n) * sizeof(T))); }
    ~Buf() { if (p) (void)hipFree(p); }
};

// float -> unsigned with the same order (for finite values and infinities; -0 sorts below +0)
__host__ __device__ inline uint32_t ord(float f)
{
    uint32_t u;
#if defined(__HIP_DEVICE_COMPILE__)
    u = __float_as_uint(f);
#else
    std::memcpy(&u, &f, 4);
#endif
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ inline float unord(uint32_t k)
{
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

struct DevSeg {
    int32_t l, r, node, parent, lc, rc;
};

// index of the level's range that holds position `pos`, or -1
__device__ inline int find_seg(const DevSeg* segs, int n_segs, int pos)
{
    int lo = 0, hi = n_segs - 1, ans = -1;
    while (lo <= hi) {
        const int mid = (lo + hi) >> 1;
        if (segs[mid].l <= pos) { ans = mid; lo = mid + 1; } else hi = mid - 1;
    }
    return (ans >= 0 && pos < segs[ans].r) ? ans : -1;
}

__global__ void k_level_begin(const DevSeg* segs, int n_segs, const uint8_t* parent_dead, uint8_t* dead, uint32_t* box)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_segs) return;
    dead[s] = segs[s].parent >= 0 ? parent_dead[segs[s].parent] : 0;
    for (int c = 0; c < 3; c++) { box[s * 6 + c] = 0xffffffffu; box[s * 6 + 3 + c] = 0u; }
}

// boxes of all live ranges of a level: BVH.h:43-52.  min / max over order-preserving keys are the exact float min / max.
__global__ __launch_bounds__(256) void k_bounds(const DevSeg* segs, int n_segs, const uint8_t* dead, const uint32_t* idx, const float* tmin, const float* tmax,
                                                int n, uint32_t* box)
{
    const int pos = blockIdx.x * 256 + threadIdx.x;
    int s = -1;
    uint32_t v[6] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u};
    if (pos < n) {
        s = find_seg(segs, n_segs, pos);
        if (s >= 0 && dead[s]) s = -1;
        if (s >= 0) {
            const uint32_t t = idx[pos];
            for (int c = 0; c < 3; c++) { v[c] = ord(tmin[t * 3 + c]); v[3 + c] = ord(tmax[t * 3 + c]); }
        }
    }
    // one atomic per wave and component when the whole wave sits in one range (the upper levels), per lane otherwise
    const int s0 = __shfl(s, 0, 64);
    const bool uniform = __all(s == s0);
    if (uniform) {
        if (s0 < 0) return;
        for (int c = 0; c < 6; c++) {
            uint32_t x = v[c];
            for (int o = 32; o > 0; o >>= 1) {
                const uint32_t y = (uint32_t)__shfl_xor((int)x, o, 64);
                x = c < 3 ? min(x, y) : max(x, y);
            }
            v[c] = x;
        }
        if ((threadIdx.x & 63) == 0) {
            for (int c = 0; c < 3; c++) { atomicMin(&box[s0 * 6 + c], v[c]); atomicMax(&box[s0 * 6 + 3 + c], v[3 + c]); }
        }
    } else if (s >= 0) {
        for (int c = 0; c < 3; c++) { atomicMin(&box[s * 6 + c], v[c]); atomicMax(&box[s * 6 + 3 + c], v[3 + c]); }
    }
}

// node records (DeviceBVHNode layout, include/crt.h: crt_bvh_node) and the sort axis of every range: BVH.h:55-76
__global__ void k_nodes(const DevSeg* segs, int n_segs, uint8_t* dead, const uint32_t* box, uint32_t thresh, crt_bvh_node* nodes, int8_t* axis)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_segs) return;
    axis[s] = -1;
    if (dead[s]) return;
    const DevSeg g = segs[s];
    crt_bvh_node nd;
    float lo[3], hi[3];
    for (int c = 0; c < 3; c++) { lo[c] = unord(box[s * 6 + c]); hi[c] = unord(box[s * 6 + 3 + c]); nd.aa[c] = lo[c]; nd.bb[c] = hi[c]; }
    nd.it = g.l;
    nd.n = (uint32_t)(g.r - g.l);
    nd.lc = -1; nd.rc = -1;
    if (nd.n > thresh) {
        nd.lc = g.lc; nd.rc = g.rc;
        const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        int a = -1; // ties x >= y >= z
        if (dx >= dy && dx >= dz) a = 0;
        else if (dy >= dx && dy >= dz) a = 1;
        else if (dz >= dx && dz >= dy) a = 2;
        if (a < 0) { dead[s] = 1; return; } // (NaN extents: the reference does not sort; left to the host)
        axis[s] = (int8_t)a;
    }
    nodes[g.node] = nd;
}

__global__ __launch_bounds__(256) void k_keys(const DevSeg* segs, int n_segs, const uint8_t* dead, const int8_t* axis, const uint32_t* idx, const float* cen, int n,
                                              unsigned long long* keys)
{
    const int pos = blockIdx.x * 256 + threadIdx.x;
    if (pos >= n) return;
    const int s = find_seg(segs, n_segs, pos);
    unsigned long long k = (unsigned long long)(uint32_t)pos << 32; // stays where it is
    if (s >= 0 && !dead[s] && axis[s] >= 0) k = ((unsigned long long)(uint32_t)segs[s].l << 32) | ord(cen[idx[pos] * 3 + axis[s]]);
    keys[pos] = k;
}

// two equal keys in one sorted range: std::sort's order of them is its own business -> the range goes to the host
__global__ __launch_bounds__(256) void k_ties(const DevSeg* segs, int n_segs, uint8_t* dead, const int8_t* axis, const unsigned long long* keys, int n, uint8_t* tie)
{
    const int pos = blockIdx.x * 256 + threadIdx.x;
    if (pos >= n) return;
    const int s = find_seg(segs, n_segs, pos);
    if (s < 0 || dead[s] || axis[s] < 0 || pos == segs[s].l) return;
    const float a = unord((uint32_t)keys[pos]), b = unord((uint32_t)keys[pos - 1]);
    if (!(a > b)) tie[s] = 1; // equal (-0 == +0 included) or unordered
}
__global__ __launch_bounds__(256) void k_restore(const DevSeg* segs, int n_segs, uint8_t* dead, const uint8_t* tie, const uint32_t* idx_in, uint32_t* idx_out, int n)
{
    const int pos = blockIdx.x * 256 + threadIdx.x;
    if (pos >= n) return;
    const int s = find_seg(segs, n_segs, pos);
    if (s >= 0 && tie[s]) {
        idx_out[pos] = idx_in[pos];
        if (pos == segs[s].l) dead[s] = 1;
    }
}

// nodes of the subtree over n triangles (the shape depends on n alone)
struct Shape {
    uint32_t thresh;
    std::vector<int64_t> memo;
    int64_t count(int64_t n)
    {
        if (n <= (int64_t)thresh) return 1;
        if (n < (int64_t)memo.size() && memo[n]) return memo[n];
        const int64_t c = count(n / 2) + count(n - n / 2) + 1;
        if (n < (int64_t)memo.size()) memo[n] = c;
        return c;
    }
};

int bfail(int status, const std::string& msg)
{
    crt_set_last_error_(msg.c_str());
    return status;
}

} // namespace

int crt_bvh_build_device(uint32_t n, const float* tmin, const float* tmax, const float* centroid, uint32_t thresh, int device, uint32_t* out_perm,
                         crt_bvh_node* out_nodes, uint32_t n_nodes_cap, std::vector<crt_bvh_host_range>* host_ranges, crt_bvh_build_info* info)
{
    if (!tmin || !tmax || !centroid || !out_perm || !out_nodes || !host_ranges || n == 0 || thresh == 0) return bfail(CRT_ERR_INVALID_ARG, "crt_bvh_build_device: bad arguments");
    if (n >= (1u << 27)) return bfail(CRT_ERR_UNSUPPORTED, "crt_bvh_build_device: more than 2^27 triangles");
    using clk = std::chrono::steady_clock;
    const auto t_begin = clk::now();
    try {
        // ---- the levels: every range of the recursion, its node and its children's nodes ----
        Shape shape{thresh, std::vector<int64_t>((size_t)n + 1, 0)};
        const int64_t total = shape.count(n);
        if ((uint64_t)total > n_nodes_cap) return bfail(CRT_ERR_INVALID_ARG, "crt_bvh_build_device: node buffer too small");
        std::vector<std::vector<DevSeg>> levels(1);
        levels[0].push_back(DevSeg{0, (int32_t)n, (int32_t)total - 1, -1, -1, -1});
        for (size_t lv = 0;; lv++) {
            std::vector<DevSeg> next;
            for (size_t i = 0; i < levels[lv].size(); i++) {
                DevSeg& g = levels[lv][i];
                const int64_t cnt = g.r - g.l;
                if (cnt <= (int64_t)thresh) continue;
                const int32_t mid = (int32_t)(((int64_t)g.l + g.r) / 2); // BVH.h:78
                const int64_t first = (int64_t)g.node - shape.count(cnt) + 1;
                g.lc = (int32_t)(first + shape.count(mid - g.l) - 1);
                g.rc = g.node - 1;
                next.push_back(DevSeg{g.l, mid, g.lc, (int32_t)i, -1, -1});
                next.push_back(DevSeg{mid, g.r, g.rc, (int32_t)i, -1, -1});
            }
            if (next.empty()) break;
            levels.push_back(std::move(next));
        }
        int key_bits = 32;
        while ((1ull << (key_bits - 32)) < (unsigned long long)n) key_bits++;

        BHIP(hipSetDevice(device));
        Buf<float> d_tmin, d_tmax, d_cen;
        Buf<uint32_t> d_idx[2], d_box;
        Buf<unsigned long long> d_keys[2];
        Buf<crt_bvh_node> d_nodes;
        Buf<DevSeg> d_segs;
        Buf<uint8_t> d_dead, d_tie, d_temp;
        Buf<int8_t> d_axis;
        size_t n_segs_total = 0, max_segs = 0;
        std::vector<size_t> seg_off;
        for (const auto& L : levels) { seg_off.push_back(n_segs_total); n_segs_total += L.size(); max_segs = std::max(max_segs, L.size()); }
        d_tmin.alloc((size_t)n * 3); d_tmax.alloc((size_t)n * 3); d_cen.alloc((size_t)n * 3);
        d_idx[0].alloc(n); d_idx[1].alloc(n); d_keys[0].alloc(n); d_keys[1].alloc(n);
        d_box.alloc(max_segs * 6); d_nodes.alloc((size_t)total); d_segs.alloc(n_segs_total); d_dead.alloc(n_segs_total); d_tie.alloc(max_segs);
        d_axis.alloc(max_segs);
        {
            std::vector<DevSeg> flat;
            flat.reserve(n_segs_total);
            for (const auto& L : levels) flat.insert(flat.end(), L.begin(), L.end());
            BHIP(hipMemcpy(d_segs.p, flat.data(), flat.size() * sizeof(DevSeg), hipMemcpyHostToDevice));
        }
        BHIP(hipMemcpy(d_tmin.p, tmin, (size_t)n * 12, hipMemcpyHostToDevice));
        BHIP(hipMemcpy(d_tmax.p, tmax, (size_t)n * 12, hipMemcpyHostToDevice));
        BHIP(hipMemcpy(d_cen.p, centroid, (size_t)n * 12, hipMemcpyHostToDevice));
        {
            std::vector<uint32_t> iota(n);
            for (uint32_t i = 0; i < n; i++) iota[i] = i;
            BHIP(hipMemcpy(d_idx[0].p, iota.data(), (size_t)n * 4, hipMemcpyHostToDevice));
        }
        BHIP(hipMemset(d_nodes.p, 0, (size_t)total * sizeof(crt_bvh_node)));
        size_t temp_bytes = 0;
        BHIP(rocprim::radix_sort_pairs(nullptr, temp_bytes, d_keys[0].p, d_keys[1].p, d_idx[0].p, d_idx[1].p, (size_t)n, 0u, (unsigned)key_bits, (hipStream_t) nullptr));
        d_temp.alloc(temp_bytes);
        hipEvent_t e0, e1;
        BHIP(hipEventCreate(&e0)); BHIP(hipEventCreate(&e1));
        BHIP(hipEventRecord(e0, nullptr));
        const dim3 gridN((n + 255) / 256);
        int cur = 0;
        for (size_t lv = 0; lv < levels.size(); lv++) {
            const int ns = (int)levels[lv].size();
            const DevSeg* segs = d_segs.p + seg_off[lv];
            uint8_t* dead = d_dead.p + seg_off[lv];
            const uint8_t* pdead = lv ? d_dead.p + seg_off[lv - 1] : nullptr;
            const dim3 gridS((ns + 255) / 256);
            hipLaunchKernelGGL(k_level_begin, gridS, dim3(256), 0, nullptr, segs, ns, pdead, dead, d_box.p);
            hipLaunchKernelGGL(k_bounds, gridN, dim3(256), 0, nullptr, segs, ns, dead, d_idx[cur].p, d_tmin.p, d_tmax.p, (int)n, d_box.p);
            hipLaunchKernelGGL(k_nodes, gridS, dim3(256), 0, nullptr, segs, ns, dead, d_box.p, thresh, d_nodes.p, d_axis.p);
            if (lv + 1 == levels.size()) break; // the deepest level holds leaves only
            hipLaunchKernelGGL(k_keys, gridN, dim3(256), 0, nullptr, segs, ns, dead, d_axis.p, d_idx[cur].p, d_cen.p, (int)n, d_keys[0].p);
            BHIP(rocprim::radix_sort_pairs(d_temp.p, temp_bytes, d_keys[0].p, d_keys[1].p, d_idx[cur].p, d_idx[cur ^ 1].p, (size_t)n, 0u, (unsigned)key_bits,
                                           (hipStream_t) nullptr));
            BHIP(hipMemsetAsync(d_tie.p, 0, (size_t)ns, nullptr));
            hipLaunchKernelGGL(k_ties, gridN, dim3(256), 0, nullptr, segs, ns, dead, d_axis.p, d_keys[1].p, (int)n, d_tie.p);
            hipLaunchKernelGGL(k_restore, gridN, dim3(256), 0, nullptr, segs, ns, dead, d_tie.p, d_idx[cur].p, d_idx[cur ^ 1].p, (int)n);
            BHIP(hipGetLastError());
            cur ^= 1;
        }
        BHIP(hipEventRecord(e1, nullptr));
        BHIP(hipDeviceSynchronize());
        float dev_ms = 0.0f;
        BHIP(hipEventElapsedTime(&dev_ms, e0, e1));
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
        BHIP(hipMemcpy(out_perm, d_idx[cur].p, (size_t)n * 4, hipMemcpyDeviceToHost));
        BHIP(hipMemcpy(out_nodes, d_nodes.p, (size_t)total * sizeof(crt_bvh_node), hipMemcpyDeviceToHost));
        std::vector<uint8_t> dead(n_segs_total);
        BHIP(hipMemcpy(dead.data(), d_dead.p, n_segs_total, hipMemcpyDeviceToHost));
        host_ranges->clear();
        uint64_t host_tris = 0;
        for (size_t lv = 0; lv < levels.size(); lv++)
            for (size_t i = 0; i < levels[lv].size(); i++) {
                const DevSeg& g = levels[lv][i];
                const bool d = dead[seg_off[lv] + i] != 0;
                const bool pd = g.parent >= 0 && dead[seg_off[lv - 1] + (size_t)g.parent] != 0;
                if (d && !pd) {
                    host_ranges->push_back(crt_bvh_host_range{(uint32_t)g.l, (uint32_t)g.r, (uint32_t)((int64_t)g.node - shape.count(g.r - g.l) + 1), (uint32_t)lv + 1});
                    host_tris += (uint64_t)(g.r - g.l);
                }
            }
        if (info) {
            std::memset(info, 0, sizeof(*info));
            info->n_triangles = n; info->n_nodes = (uint32_t)total; info->levels = (uint32_t)levels.size();
            info->device_ms = dev_ms;
            info->host_ranges = (uint32_t)host_ranges->size();
            info->host_triangles = (uint32_t)host_tris;
            info->total_ms = std::chrono::duration<float, std::milli>(clk::now() - t_begin).count();
        }
        return CRT_OK;
    } catch (const BErr& f) {
        return bfail(CRT_ERR_HIP, std::string("crt_bvh_build_device: ") + f.what + ": " + hipGetErrorString(f.e));
    } catch (const std::bad_alloc&) {
        return bfail(CRT_ERR_OOM, "crt_bvh_build_device: out of host memory");
    }
}

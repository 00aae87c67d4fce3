// cudaraytracing_amd/csrc/crt_bvh_build.hip -- the reference's BVH built on the GPU.
//
// reference: include/BVH.h:37-84 -- recursive median split: box of the range, leaf iff n <= thresh_n, else std::sort of the
// range by centroid along the longest axis (ties x >= y >= z), split at mid = (l + r) / 2, nodes appended in post-order.
//
// Two facts make a level-synchronous device build byte-identical to that recursion:
//   * the SHAPE of the tree depends on the triangle count alone (left child floor(n / 2), right child ceil(n / 2) triangles), so
//     every range, its node index in the post-order array and its children's indices are known before a triangle is touched --
//     the host lays the levels out (a few thousand integers) and the device fills them in;
//   * the reference sorts with std::sort, and centroid coordinates are full of EQUAL keys on real meshes (the triangles of a grid
//     column, of a wall plane), where the result of an unstable sort is its own.  libstdc++'s std::sort is introsort: a quicksort
//     phase (median of three to the front, unguarded Hoare partition, recursion until a part has <= 16 elements, depth limit
//     2 * floor(log2 n) then heapsort) followed by ONE insertion sort over the whole range.  Insertion sort is stable, so
//         std::sort(range) == stable_sort(arrangement the quicksort phase left),
//     and the Hoare partition has a closed parallel form: with Lo = positions (ascending) whose key is >= pivot and Ro = positions
//     (descending) whose key is <= pivot, the k-th swap exchanges Lo[k] and Ro[k] for as long as Lo[k] < Ro[k], and the cut is
//     min(Lo[K], Ro[K-1]).  The device therefore REPLAYS the quicksort phase of every range of a level (one workgroup per
//     partition task, task lists ping-ponged round by round) and finishes with a stable LSD radix sort (rocPRIM) on (range start,
//     order-preserving bits of the key): exactly std::sort's permutation, ties included (the CPU model of this replay was checked
//     against std::sort on 3 000 tie-heavy arrays; the GPU test compares whole trees byte for byte).  A partition that runs into
//     the depth limit (std::sort then heapsorts that part: sequential) flags its range; the host std::sort's exactly that one range
//     between two levels and the device carries on.  Scenes with -0.0 or non-finite coordinates (where std::min / std::max and the comparator depend on scan
//     order) are not started on the device at all.
// Per level: range boxes (wave-aggregated atomic min / max on order-preserving integer keys), node records, sort keys, the replay
// rounds, one radix sort of the whole index array (elements outside the level's sorting ranges carry their position as key and
// stay put), restore of flagged ranges.
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

// float key of every element of a sorting range (centroid along the range's axis), and a working copy of the order
__global__ __launch_bounds__(256) void k_fkeys(const DevSeg* segs, int n_segs, const uint8_t* dead, const int8_t* axis, const uint32_t* idx, const float* cen, int n,
                                               float* fkey, uint32_t* work)
{
    const int pos = blockIdx.x * 256 + threadIdx.x;
    if (pos >= n) return;
    const int s = find_seg(segs, n_segs, pos);
    const uint32_t t = idx[pos];
    work[pos] = t;
    fkey[pos] = (s >= 0 && !dead[s] && axis[s] >= 0) ? cen[t * 3 + axis[s]] : 0.0f;
}

struct QTask {
    int32_t l, r, depth, seg;
};
__device__ inline int floor_log2(uint32_t n) { return 31 - __clz((int)n); }

// the partition tasks a level starts with: every live sorting range of more than 16 elements (smaller ones are insertion-sorted
// by std::sort, i.e. stably: the radix sort alone reproduces that), depth limit 2 * floor(log2 n) (std::__lg(n) * 2)
__global__ void k_qs_init(const DevSeg* segs, int n_segs, const uint8_t* dead, const int8_t* axis, uint8_t* tie, QTask* tasks, uint32_t* n_tasks)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_segs || dead[s] || axis[s] < 0 || tie[s] != 3) return; // only ranges that hold equal keys need the replay
    tie[s] = 0;
    const int n = segs[s].r - segs[s].l;
    if (n <= 16) return;
    tasks[atomicAdd(n_tasks, 1u)] = QTask{segs[s].l, segs[s].r, 2 * floor_log2((uint32_t)n), s};
}

// block-wide exclusive prefix of a flag (256 threads = 4 waves); returns this thread's rank among the flagged ones before it,
// and the block's total through `total`
__device__ inline uint32_t block_rank(bool flag, uint32_t* s_wave, uint32_t& total)
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

// One round of the quicksort phase: every task = one __unguarded_partition_pivot call of libstdc++ (stl_algo.h) on [l, r):
// median of (l+1, mid, r-1) swapped to l, Hoare partition of (l, r) around it, both parts with more than 16 elements become
// tasks of the next round with the depth limit lowered by one.
__global__ __launch_bounds__(256) void k_qs_round(const QTask* tasks, const uint32_t* n_tasks, QTask* next, uint32_t* n_next, float* fkey, uint32_t* work,
                                                  uint32_t* Lo, uint32_t* Ro, uint8_t* heap)
{
    __shared__ uint32_t s_wave[4];
    __shared__ uint32_t s_K;
    const uint32_t nt = *n_tasks;
    for (uint32_t ti = blockIdx.x; ti < nt; ti += gridDim.x) {
        const QTask t = tasks[ti];
        const int l = t.l, r = t.r, n = r - l;
        if (t.depth == 0) { // std::sort would heapsort this part: the whole range goes to the host builder
            if (threadIdx.x == 0) heap[t.seg] = 1;
            continue;
        }
        if (threadIdx.x == 0) { // __move_median_to_first(first, first + 1, mid, last - 1)
            const int A = l + 1, B = l + n / 2, C = r - 1;
            const float a = fkey[A], b = fkey[B], c = fkey[C];
            int m;
            if (a < b) m = (b < c) ? B : ((a < c) ? C : A);
            else m = (a < c) ? A : ((b < c) ? C : B);
            const float kf = fkey[l]; fkey[l] = fkey[m]; fkey[m] = kf;
            const uint32_t ki = work[l]; work[l] = work[m]; work[m] = ki;
            s_K = 0;
        }
        __syncthreads();
        const float p = fkey[l];
        // Lo: positions of (l, r) whose key is not below the pivot, ascending; Ro: positions whose key is not above it, descending
        uint32_t mL = 0, mR = 0;
        for (int base = 0; base < n - 1; base += 256) {
            const int k = base + (int)threadIdx.x;
            const int i = l + 1 + k, j = r - 1 - k;
            const bool in = k < n - 1;
            const bool ge = in && !(fkey[i] < p), le = in && !(p < fkey[j]);
            uint32_t tot;
            const uint32_t rl = block_rank(ge, s_wave, tot);
            if (ge) Lo[l + mL + rl] = (uint32_t)i;
            mL += tot;
            const uint32_t rr = block_rank(le, s_wave, tot);
            if (le) Ro[l + mR + rr] = (uint32_t)j;
            mR += tot;
        }
        __syncthreads();
        // the swaps: pair k for as long as Lo[k] < Ro[k] (Lo ascends, Ro descends: the condition holds on a prefix)
        const uint32_t mm = min(mL, mR);
        uint32_t cnt = 0;
        for (uint32_t k = threadIdx.x; k < mm; k += 256) cnt += Lo[l + k] < Ro[l + k] ? 1u : 0u;
        for (int o = 32; o > 0; o >>= 1) cnt += (uint32_t)__shfl_xor((int)cnt, o, 64);
        if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(&s_K, cnt);
        __syncthreads();
        const uint32_t K = s_K;
        for (uint32_t k = threadIdx.x; k < K; k += 256) {
            const uint32_t i = Lo[l + k], j = Ro[l + k];
            const float kf = fkey[i]; fkey[i] = fkey[j]; fkey[j] = kf;
            const uint32_t ki = work[i]; work[i] = work[j]; work[j] = ki;
        }
        if (threadIdx.x == 0) {
            const uint32_t inf = 0xffffffffu;
            const uint32_t lo = K < mL ? Lo[l + K] : inf;
            const uint32_t cut = K > 0 ? min(lo, Ro[l + K - 1]) : lo; // what __unguarded_partition returns
            if (r - (int)cut > 16) next[atomicAdd(n_next, 1u)] = QTask{(int)cut, r, t.depth - 1, t.seg};
            if ((int)cut - l > 16) next[atomicAdd(n_next, 1u)] = QTask{l, (int)cut, t.depth - 1, t.seg};
        }
        __syncthreads();
    }
}

// 64-bit radix keys after the replay: (range start, order-preserving bits of the key); elements outside sorting ranges stay put
// (tie[s] == 2: the host has sorted that range already -- its elements stay where the host put them)
__global__ __launch_bounds__(256) void k_keys(const DevSeg* segs, int n_segs, const uint8_t* dead, const int8_t* axis, const uint8_t* tie, const float* fkey, int n,
                                              unsigned long long* keys)
{
    const int pos = blockIdx.x * 256 + threadIdx.x;
    if (pos >= n) return;
    const int s = find_seg(segs, n_segs, pos);
    unsigned long long k = (unsigned long long)(uint32_t)pos << 32;
    if (s >= 0 && !dead[s] && axis[s] >= 0 && tie[s] != 2) k = ((unsigned long long)(uint32_t)segs[s].l << 32) | ord(fkey[pos]);
    keys[pos] = k;
}

// after the first sort of a level: two equal keys side by side in one range -> that range's std::sort order has to be replayed (3)
__global__ __launch_bounds__(256) void k_ties(const DevSeg* segs, int n_segs, const uint8_t* dead, const int8_t* axis, const unsigned long long* keys, int n, uint8_t* tie)
{
    const int pos = blockIdx.x * 256 + threadIdx.x;
    if (pos >= n) return;
    const int s = find_seg(segs, n_segs, pos);
    if (s < 0 || dead[s] || axis[s] < 0 || pos == segs[s].l || segs[s].r - segs[s].l <= 16) return; // (<= 16 elements: insertion sort = stable)
    if ((uint32_t)keys[pos] == (uint32_t)keys[pos - 1]) tie[s] = 3;
}

// a range flagged by the replay (heapsort territory) goes back to its pre-sort order and to the host builder
__global__ __launch_bounds__(256) void k_restore(const DevSeg* segs, int n_segs, uint8_t* dead, const uint8_t* tie, const uint32_t* idx_in, uint32_t* idx_out, int n)
{
    const int pos = blockIdx.x * 256 + threadIdx.x;
    if (pos >= n) return;
    const int s = find_seg(segs, n_segs, pos);
    if (s >= 0 && tie[s] == 1) {
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
        Buf<float> d_fkey;
        Buf<uint32_t> d_work, d_Lo, d_Ro, d_ntasks;
        Buf<QTask> d_tasks[2];
        size_t n_segs_total = 0, max_segs = 0;
        std::vector<size_t> seg_off;
        for (const auto& L : levels) { seg_off.push_back(n_segs_total); n_segs_total += L.size(); max_segs = std::max(max_segs, L.size()); }
        d_tmin.alloc((size_t)n * 3); d_tmax.alloc((size_t)n * 3); d_cen.alloc((size_t)n * 3);
        d_idx[0].alloc(n); d_idx[1].alloc(n); d_keys[0].alloc(n); d_keys[1].alloc(n);
        d_box.alloc(max_segs * 6); d_nodes.alloc((size_t)total); d_segs.alloc(n_segs_total); d_dead.alloc(n_segs_total); d_tie.alloc(max_segs);
        d_axis.alloc(max_segs);
        d_fkey.alloc(n); d_work.alloc(n); d_Lo.alloc(n); d_Ro.alloc(n); d_ntasks.alloc(2);
        d_tasks[0].alloc((size_t)n / 16 + 2); d_tasks[1].alloc((size_t)n / 16 + 2); // (parts of more than 16 elements, disjoint)
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
        BHIP(rocprim::radix_sort_pairs(nullptr, temp_bytes, d_keys[0].p, d_keys[1].p, d_work.p, d_idx[1].p, (size_t)n, 0u, (unsigned)key_bits, (hipStream_t) nullptr));
        d_temp.alloc(temp_bytes);
        hipEvent_t e0, e1;
        BHIP(hipEventCreate(&e0)); BHIP(hipEventCreate(&e1));
        BHIP(hipEventRecord(e0, nullptr));
        const dim3 gridN((n + 255) / 256);
        int cur = 0;
        uint32_t host_sorts = 0;
        uint64_t host_sort_elems = 0;
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
            // ---- std::sort of every range ----
            // (1) a stable radix sort of the level: final for every range whose keys are pairwise different (one sorted order exists)
            hipLaunchKernelGGL(k_fkeys, gridN, dim3(256), 0, nullptr, segs, ns, dead, d_axis.p, d_idx[cur].p, d_cen.p, (int)n, d_fkey.p, d_work.p);
            BHIP(hipMemsetAsync(d_tie.p, 0, (size_t)ns, nullptr));
            hipLaunchKernelGGL(k_keys, gridN, dim3(256), 0, nullptr, segs, ns, dead, d_axis.p, d_tie.p, d_fkey.p, (int)n, d_keys[0].p);
            BHIP(rocprim::radix_sort_pairs(d_temp.p, temp_bytes, d_keys[0].p, d_keys[1].p, d_work.p, d_idx[cur ^ 1].p, (size_t)n, 0u, (unsigned)key_bits,
                                           (hipStream_t) nullptr));
            hipLaunchKernelGGL(k_ties, gridN, dim3(256), 0, nullptr, segs, ns, dead, d_axis.p, d_keys[1].p, (int)n, d_tie.p);
            // (2) ranges with equal keys: replay of the quicksort phase of std::sort, then the stable sort again
            BHIP(hipMemsetAsync(d_ntasks.p, 0, 2 * sizeof(uint32_t), nullptr));
            hipLaunchKernelGGL(k_qs_init, gridS, dim3(256), 0, nullptr, segs, ns, dead, d_axis.p, d_tie.p, d_tasks[0].p, d_ntasks.p);
            uint32_t live[2] = {0, 0};
            BHIP(hipMemcpy(live, d_ntasks.p, sizeof(live), hipMemcpyDeviceToHost));
            if (live[0] == 0) { cur ^= 1; continue; } // no range of this level holds equal keys
            {
                int64_t biggest = 0;
                for (const DevSeg& g : levels[lv]) biggest = std::max<int64_t>(biggest, g.r - g.l);
                int rounds = 0;
                if (biggest > 16) { int lg = 0; while ((2ll << lg) <= biggest) lg++; rounds = 2 * lg + 1; } // the depth limit ends every chain
                const uint32_t qs_blocks = (uint32_t)std::min<int64_t>(1024, std::max<int64_t>(1, (int64_t)n / 17));
                for (int rd = 0; rd < rounds; rd++) {
                    const int in = rd & 1, outb = in ^ 1;
                    BHIP(hipMemsetAsync(d_ntasks.p + outb, 0, sizeof(uint32_t), nullptr));
                    hipLaunchKernelGGL(k_qs_round, dim3(qs_blocks), dim3(256), 0, nullptr, d_tasks[in].p, d_ntasks.p + in, d_tasks[outb].p, d_ntasks.p + outb,
                                       d_fkey.p, d_work.p, d_Lo.p, d_Ro.p, d_tie.p);
                    if ((rd & 3) == 3) { // every fourth round: are there parts left?
                        BHIP(hipMemcpy(live, d_ntasks.p, sizeof(live), hipMemcpyDeviceToHost));
                        if (live[outb] == 0) break;
                    }
                }
            }
            // A partition that ran into std::sort's depth limit flagged its range (heapsort is a sequential algorithm): the host sorts
            // exactly that range with std::sort -- from the order the range had before this level, which is the reference's -- and the
            // device carries on with everything else and with the levels below.
            {
                std::vector<uint8_t> flags((size_t)ns);
                BHIP(hipMemcpy(flags.data(), d_tie.p, (size_t)ns, hipMemcpyDeviceToHost));
                bool any = false;
                for (int q = 0; q < ns; q++) any = any || flags[q] != 0;
                if (any) {
                    std::vector<int8_t> ax((size_t)ns);
                    BHIP(hipMemcpy(ax.data(), d_axis.p, (size_t)ns, hipMemcpyDeviceToHost));
                    for (int q = 0; q < ns; q++) {
                        if (!flags[q]) continue;
                        const DevSeg& g = levels[lv][q];
                        const size_t cnt = (size_t)(g.r - g.l);
                        std::vector<uint32_t> order(cnt);
                        BHIP(hipMemcpy(order.data(), d_idx[cur].p + g.l, cnt * 4, hipMemcpyDeviceToHost));
                        struct KI { float k; uint32_t i; };
                        std::vector<KI> ki(cnt);
                        for (size_t e = 0; e < cnt; e++) ki[e] = KI{centroid[(size_t)order[e] * 3 + ax[q]], order[e]};
                        std::sort(ki.begin(), ki.end(), [](const KI& a, const KI& b) { return a.k < b.k; }); // BVH.h:67-75
                        for (size_t e = 0; e < cnt; e++) order[e] = ki[e].i;
                        BHIP(hipMemcpy(d_work.p + g.l, order.data(), cnt * 4, hipMemcpyHostToDevice));
                        flags[q] = 2;
                        host_sorts++;
                        host_sort_elems += cnt;
                    }
                    BHIP(hipMemcpy(d_tie.p, flags.data(), (size_t)ns, hipMemcpyHostToDevice));
                }
            }
            hipLaunchKernelGGL(k_keys, gridN, dim3(256), 0, nullptr, segs, ns, dead, d_axis.p, d_tie.p, d_fkey.p, (int)n, d_keys[0].p);
            BHIP(rocprim::radix_sort_pairs(d_temp.p, temp_bytes, d_keys[0].p, d_keys[1].p, d_work.p, d_idx[cur ^ 1].p, (size_t)n, 0u, (unsigned)key_bits,
                                           (hipStream_t) nullptr));
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
            info->host_sorts = host_sorts;
            info->host_sort_elements = host_sort_elems;
            info->total_ms = std::chrono::duration<float, std::milli>(clk::now() - t_begin).count();
        }
        return CRT_OK;
    } catch (const BErr& f) {
        return bfail(CRT_ERR_HIP, std::string("crt_bvh_build_device: ") + f.what + ": " + hipGetErrorString(f.e));
    } catch (const std::bad_alloc&) {
        return bfail(CRT_ERR_OOM, "crt_bvh_build_device: out of host memory");
    }
}

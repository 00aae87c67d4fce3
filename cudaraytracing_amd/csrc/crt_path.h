// cudaraytracing_amd/csrc/crt_path.h -- what the render kernels share: the path pool and the launch parameters (filled by the host
// code of crt_render.hip), work-item cursors, the path logic both pipelines are made of (next-event set-up, backward recursion:
// include/Render.cuh:199-326), the commit ring of CRT_FLAG_BOUNDED_RADIANCE, and the short exact reciprocal.
// Included by crt_mega3.hip (the megakernel), crt_wavefront.hip (the fallback pipeline), crt_frame.hip (frame kernels, test
// kernels) and crt_render.hip (host).
#ifndef CRT_PATH_H
#define CRT_PATH_H
#include "../../include/crt.h"
#include "crt_accel.h"
#include "crt_device.h"
#include "crt_trace.h"

#include <hip/hip_runtime.h>

#include <cfloat>
#include <cstddef>
#include <cstdint>
#include <type_traits>

namespace crtk {
using namespace crtdev;

// Statistics counters are sharded over CNT_SHARDS cache lines (CNT_STRIDE x u64 each): thousands of
// atomics per launch on ONE address serialise at the memory side (~12 ns each) and cost more
// than the kernel itself.  The host sums the shards.
enum { C_RAYS = 0, C_SHADOW, C_PROBE, C_INNER, C_LEAF, C_TESTS, C_HITS, C_PATHS, C_ALIVE, C_MAXSP, C_SUMSP, C_CYC_LOGIC, C_CYC_LEAF, C_CYC_INNER, C_CYC_OTHER,
       C_DIAG /* CRT_DIAG_N more diagnostic slots (-DCRT_STAMPS builds) */, C_UNTRACED = C_DIAG + 20, C_COUNT };
#define CNT_SHARDS 256
#define CNT_STRIDE 40
// The work-item cursor is sharded too: shard s hands out items [s*per, (s+1)*per); a wave
// starts at its home shard and moves on when a shard is exhausted.
#define ITEM_SHARDS 64
#define ITEM_STRIDE 32 /* u32 per shard = one 128 B line */

// Stage of a path (4 bits of the `la` plane's word).  The wavefront pipeline uses the first five; k_mega3 adds: ST_FIN = path
// complete, backward recursion pending (q bit 0: the deepest vertex is an emitter); ST_NEED = vertex entered with zero next-event
// samples, straight to roulette; ST_WAIT (commit ring) = the ray slot holds a work item it may not start yet.
enum { ST_DEAD = 0, ST_NEW = 1, ST_HIT = 2, ST_PROBE = 3, ST_SHADOW = 4, ST_FIN = 5, ST_NEED = 6, ST_WAIT = 7, ST_COUNT_ };
static_assert(ST_COUNT_ <= 16, "a stage is kept in 4 bits");
enum { RAY_NONE = 0, RAY_CLOSEST = 1, RAY_SHADOW = 2 };
#define ITEM_NONE 0xffffffffu

// Path pool, structure of arrays; every plane has `n` entries.
struct Pool {
    float4* ro;   // ray origin.xyz, t_limit (shadow rays: Render.cuh:272)
    float4* rd;   // ray direction.xyz (normalised as Ray does), bits(ray kind)
    float4* vx;   // current vertex position.xyz, bits(triangle) (k_mega3, round 6: only the vertex that waits for a SPECULAR probe's result)
    float4* la;   // next-event accumulator L_dir.xyz of the current vertex, bits(depth | stage << 8 | sample << 16)
    float4* cc;   // contribution of the in-flight shadow ray .xyz, distance to the light sample (k_mega3 but for REFERENCE: bits(the vertex's triangle))
    float4* vn;   // normal.xyz and bits(material) of the current vertex (of the PREVIOUS vertex while a bounce ray is in flight)
    uint4* id;    // pixel index, sample index, work item, unused -- written once per path (k_mega3: the work item alone, 4 B; REFERENCE: + triangle, 8 B)
    float2* res;  // result of the slot's last ray: t, bits(triangle or -1)
    float4* rec_a; // [depth][n]: L_dir.xyz of that vertex, cos to the next vertex                (k_mega3: L_dir.xyz, bits(material word) -- finish_path_m3)
    float4* rec_b; // [depth][n]: incoming direction.xyz, bits(material)                           (k_mega3: direction of SPECULAR vertices only, cos to the next vertex)
    uint32_t n;
};

struct LParams {
    DevScene sc;
    Pool pool;
    float eye[3];
    float inv_view[9];
    float scale, ar;
    uint32_t width, height;
    float p_rr;
    int32_t lsn;
    float inv_lsn_pow2;     // 1 / lsn when lsn is a power of two (x / 2^k and x * 2^-k round the same real number: same bits), else 0
    uint32_t pad_;
    uint64_t seed;
    uint32_t rank, world, tiles_x, n_tiles;
    uint32_t nslots;        // pixel slots of this shard (local tiles * 64)
    uint32_t sample_begin;  // first sample index of this chunk
    uint32_t n_items;       // nslots * samples in this chunk
    uint32_t items_per_shard;
    uint32_t n_mats;
    FastDiv lsn_div, nslots_div, tiles_x_div;
    unsigned int* item_next; // [ITEM_SHARDS * ITEM_STRIDE] cursors, relative to the shard start
    float4* L;              // per work item radiance (crt_intersect: per query ray (t, bits(triangle), -, -))
    unsigned long long* counters;
    const float4* q_o;      // crt_intersect: origins / normalised directions of the query rays (work item = ray index)
    const float4* q_d;
    const uint32_t* item_list; // NULL, or the order in which the LAST order_window work items of every cursor shard are handed out
                               // (k_order_items): [ITEM_SHARDS][order_window]
    uint32_t order_window;
    FastDiv items_per_shard_div;
    // ---- in-order commit through a ring of samples (bounded radiance storage, see commit_ring below); ring_mask == 0: off ----
    uint32_t ring_mask;       // ring samples - 1 (a power of two)
    uint32_t spsh;            // pixel slots per cursor shard (a multiple of 64: whole tiles)
    FastDiv spsh_div;
    uint32_t ring_shards;     // cursor shards of a ring launch (a power of two >= ITEM_SHARDS: a shard's commits are a serial chain,
                              // so there are more and smaller ones than without the ring)
    uint32_t ring_stride;     // slots per ring sample (= ring_shards * spsh)
    uint32_t n_samples;       // samples of this launch
    uint32_t tail_first;      // in-shard position where the ordered tail window begins (its items may belong to any later sample)
    float spp_f;              // (float)spp
    uint32_t ring_pad_;
    unsigned int* ring_done;  // [ring_shards][ring samples]: finished work items of (shard, sample mod ring)
    unsigned int* ring_state; // [ring_shards * ITEM_STRIDE]: word 0 = committed samples | busy << 31, word 1 = valid pixel slots of the shard
    float* accum;             // 3 planes of nslots: sum of L_k / spp over the committed samples (Render.cuh:348)
};

struct TParams {
    DevScene sc;
    Pool pool;
    unsigned long long* counters;
    unsigned int* slot_next;   // [SLOT_SHARDS * SLOT_STRIDE] cursors of the persistent trace kernel
    uint32_t slots_per_shard;
    int32_t stack_cap;         // traversal stack entries per lane kept in LDS
    int2* spill;               // [level][grid lanes] overflow of deeper entries (rare), L2 resident
    uint32_t spill_stride;     // grid lanes
    int32_t refill_min, leaf_min;
};

// work item slot -> pixel.  false for padding slots (ragged image edge / tile beyond the image).
__device__ __forceinline__ bool slot_to_pixel(uint32_t slot, uint32_t rank, uint32_t world, uint32_t n_tiles, uint32_t tiles_x, FastDiv tiles_x_div,
                                              uint32_t width, uint32_t height, uint32_t& i, uint32_t& j)
{
    uint32_t tile = (slot >> 6) * world + rank;
    uint32_t pix = slot & 63u;
    if (tile >= n_tiles) return false;
    uint32_t ty = fast_div(tile, tiles_x_div.m, tiles_x_div.sh), tx = tile - ty * tiles_x;
    i = tx * CRT_TILE + (pix & 7u);
    j = ty * CRT_TILE + (pix >> 3);
    return i < width && j < height;
}

// Loads / stores that say "global memory" in their type.  The logic phases of k_mega3 take their pointers out of a copy of the kernel
// arguments (LOGIC_PARAMS), where the compiler no longer sees that they are kernel arguments: plain accesses through them are FLAT
// instructions, which count on the LDS / scalar-memory counter too -- every s_waitcnt for an s_load then waits for the path-state
// loads in flight.
#define CRT_GAS __attribute__((address_space(1)))
typedef float crt_f4v_ __attribute__((ext_vector_type(4)));
typedef uint32_t crt_u4v_ __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 gld(const float4* p) { const crt_f4v_ v = *(const CRT_GAS crt_f4v_*)p; return make_float4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ uint4 gld(const uint4* p) { const crt_u4v_ v = *(const CRT_GAS crt_u4v_*)p; return make_uint4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ int32_t gld(const int32_t* p) { return *(const CRT_GAS int32_t*)p; }
__device__ __forceinline__ uint32_t gld(const uint32_t* p) { return *(const CRT_GAS uint32_t*)p; }
__device__ __forceinline__ void gst(float4* p, const float4 v) { crt_f4v_ t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w; *(CRT_GAS crt_f4v_*)p = t; }
__device__ __forceinline__ void gst(uint4* p, const uint4 v) { crt_u4v_ t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w; *(CRT_GAS crt_u4v_*)p = t; }
__device__ __forceinline__ void gst(float* p, const float v) { *(CRT_GAS float*)p = v; }
// The vertex records (rec_a / rec_b): written when a vertex is entered, read once when the path ends.  (Streaming them past the L2 -- nt
// loads and stores -- was measured in round 5: fabric traffic 348.6 -> 330.8 GB, L2 miss rate 0.475 -> 0.418, and the frame +14 %: the reads
// at a path's end then always go to the memory side, and LC waits for them.  Round 6, the STORES alone streamed, the loads plain: C2 +7.4 %,
// veach-mis +6.0 % -- the records' lines are still in L2 when the path ends.  Plain accesses.)
__device__ __forceinline__ float4 gld_rec(const float4* p) { return gld(p); }
__device__ __forceinline__ void gst_rec(float4* p, const float4 v) { gst(p, v); }
__device__ __forceinline__ void gst_rec(float* p, const float v) { gst(p, v); }

// Takes the next work item for every lane that is active here with ONE atomic per wave and
// shard (ballot of the active lanes, the first one adds their count, prefix rank per lane).
__device__ __forceinline__ uint32_t grab_item(const unsigned int* /*unused*/, unsigned int* item_next, uint32_t per, uint32_t n_items,
                                              uint32_t home)
{
    const int lane = threadIdx.x & 63;
    uint32_t item = ITEM_NONE;
    for (uint32_t t = 0; t < ITEM_SHARDS; t++) {
        const uint32_t sh = (home + t) & (ITEM_SHARDS - 1);
        const uint32_t lo = sh * per;
        if (lo >= n_items) continue;
        const uint32_t hi = min(lo + per, n_items);
        unsigned int* cur = item_next + sh * ITEM_STRIDE;
        // cursors only grow, so a stale read can at worst cost one fruitless atomic
        if (lo + __hip_atomic_load((CRT_GAS unsigned int*)cur, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= hi) continue;
        const unsigned long long mask = __ballot(1);
        const int leader = __ffsll((long long)mask) - 1;
        const uint32_t rank = (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
        unsigned int base = 0;
        if (lane == leader) base = __hip_atomic_fetch_add((CRT_GAS unsigned int*)cur, (unsigned int)__popcll(mask), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        base = (unsigned int)__builtin_amdgcn_readfirstlane((int)base); // leader is the first active lane
        const unsigned long long idx = (unsigned long long)lo + base + rank;
        if (idx < hi) { item = (uint32_t)idx; break; }
    }
    return item;
}

// The same for a commit-ring launch (P.ring_shards cursor shards): the cursors of as many shards as there are lanes asking are looked
// at in one round trip, not one after the other -- at the end of a launch every ray slot walks all shards once.
__device__ __forceinline__ uint32_t grab_item_ring(unsigned int* item_next, const uint32_t per, const uint32_t n_shards, const uint32_t home)
{
    const int lane = threadIdx.x & 63;
    uint32_t item = ITEM_NONE;
    uint32_t t = 0;
    for (;;) {
        // the lanes asking, numbered 0 .. n - 1, each look at one shard: home + t + number
        const unsigned long long mask = __ballot(1);
        const uint32_t n = (uint32_t)__popcll(mask), rank = (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
        if (t >= n_shards) break; // (t is the same in every lane that is still here)
        const uint32_t look = (home + t + rank) & (n_shards - 1u);
        const bool has = t + rank < n_shards && __hip_atomic_load((CRT_GAS unsigned int*)(item_next + look * ITEM_STRIDE), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < per;
        const unsigned long long found = __ballot(has);
        if (!found) { t += n; continue; }
        // the first shard in walking order that has items: the lane with the lowest number among `found`
        const int src = __ffsll((long long)found) - 1;
        const uint32_t sh = (uint32_t)__builtin_amdgcn_readlane((int)look, src);
        const int leader = __ffsll((long long)mask) - 1;
        unsigned int base = 0;
        if (lane == leader) base = __hip_atomic_fetch_add((CRT_GAS unsigned int*)(item_next + sh * ITEM_STRIDE), n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        base = (unsigned int)__builtin_amdgcn_readlane((int)base, leader);
        const unsigned long long idx = (unsigned long long)base + rank;
        if (idx < per) { item = sh * per + (uint32_t)idx; break; }
        // (the shard ran dry under this wave's hands: those without an item look again, from the same place)
    }
    return item;
}

// ---------------------------------------------------------------- logic ----
struct PathCounters {
    uint32_t rays, shadow, probe, paths;
    uint32_t untraced; // next-event samples answered without traversal (contribution exactly zero); counted in rays / shadow too
};

struct Lane {
    F3 ro, rd, pos, Ld, c, nrm;
    float tl;
    uint32_t kind, vtri, mat, depth, stage, q, item;
    uint32_t pixel_index, k;
};

template <bool RING = false>
__device__ __forceinline__ void decode_item(const LParams& P, uint32_t item, uint32_t& pixel_index, uint32_t& k, bool& valid,
                                            uint32_t& pi, uint32_t& pj)
{
    uint32_t s, slot;
    if (RING) { // cursor shard = a range of pixel slots, sample-major inside it (commit ring)
        const uint32_t sh = fast_div(item, P.items_per_shard_div.m, P.items_per_shard_div.sh), c = item - sh * P.items_per_shard;
        s = fast_div(c, P.spsh_div.m, P.spsh_div.sh);
        slot = sh * P.spsh + (c - s * P.spsh);
    } else {
        s = fast_div(item, P.nslots_div.m, P.nslots_div.sh);
        slot = item - s * P.nslots;
    }
    k = P.sample_begin + s;
    valid = (!RING || slot < P.nslots) && slot_to_pixel(slot, P.rank, P.world, P.n_tiles, P.tiles_x, P.tiles_x_div, P.width, P.height, pi, pj);
    pixel_index = pj * P.width + pi; // Render.cuh:336
}

// ---- commit ring: the frame's sum c += L_k / spp in sample order (Render.cuh:348) INSIDE the launch, with storage for a window of
// samples instead of one radiance per work item.  The cursor shards are ranges of pixel slots; every shard walks its samples in order,
// so the work items in flight lie within a few samples of one another.  L[(sample mod ring)][slot] holds a finished path's radiance;
// ring_done counts the finished items of (shard, sample); the wave whose count completes a sample commits it -- and the samples
// after it that are complete -- if it is the next one of its shard, else leaves it to the wave that commits the one before (one word
// per shard: committed samples | busy).  A work item of sample s is started only while s < committed + ring samples, so a slot of the
// ring is never overwritten before it has been read; a ray slot that is handed an item beyond that holds it (stage ST_WAIT) and asks
// again on its next turn.  Nothing waits on a wave that is not resident: what a shard's next commit needs are items already handed
// out, and the waves holding them go on by themselves.  Visibility across the XCDs' L2 caches: the ring, the accumulator and the
// protocol words live in uncached device memory (hipDeviceMallocUncached) AND are accessed with agent-scope atomics only (plain
// accesses to uncached memory were seen to return stale accumulator values); a wave orders its accesses with s_waitcnt.
typedef CRT_GAS unsigned int* ring_word_ptr;
__device__ __forceinline__ unsigned int ring_load(const unsigned int* p) { return __hip_atomic_load((ring_word_ptr)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ring_loadf(const float* p) { return __uint_as_float(__hip_atomic_load((ring_word_ptr)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)); }
__device__ __forceinline__ void ring_storef(float* p, const float v) { __hip_atomic_store((ring_word_ptr)p, __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void ring_wait_mem() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// A path's radiance as ONE 16-byte store with the scope bits of an agent-scope atomic store (what ring_storef's instruction carries,
// four times as wide: three 4-byte write-through stores per path were a fifth of the ring's cost).  The compiler does not count it;
// ring_publish waits for everything outstanding before the path is counted.
__device__ __forceinline__ void ring_store16(float4* p, const float x, const float y, const float z)
{
    typedef float v4f_ __attribute__((ext_vector_type(4)));
    v4f_ v; v.x = x; v.y = y; v.z = z; v.w = 0.0f;
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" : : "v"(p), "v"(v) : "memory");
}
#define RING_BUSY 0x80000000u

// May the work item be started?  (in-shard position c: items of the ordered tail window stand for the last sample)
__device__ __forceinline__ bool ring_gate_open(const LParams& P, const uint32_t item, const uint32_t home, const uint32_t home_word)
{
    const uint32_t sh = fast_div(item, P.items_per_shard_div.m, P.items_per_shard_div.sh), c = item - sh * P.items_per_shard;
    const uint32_t need = c >= P.tail_first ? P.n_samples - 1u : fast_div(c, P.spsh_div.m, P.spsh_div.sh);
    // (the word of the wave's home shard was fetched ahead, with the cursor: an older value only says "wait" where "go" was possible)
    const uint32_t committed = (sh == home ? home_word : ring_load(P.ring_state + sh * ITEM_STRIDE)) & ~RING_BUSY;
    return need - committed <= P.ring_mask; // need < committed + ring samples (need >= committed: its own sample is not committed yet)
}

// The wave (all 64 lanes) commits what is complete and next in shard sh.
__device__ __forceinline__ void ring_commit(const LParams& P, const uint32_t sh)
{
    const int lane = threadIdx.x & 63;
    unsigned int* word = P.ring_state + sh * ITEM_STRIDE;
    const uint32_t n_valid = ring_load(word + 1);
    for (;;) {
        const uint32_t w = ring_load(word);
        if (w & RING_BUSY) return;           // the wave that holds the shard looks again when it is done
        if (w >= P.n_samples) return;        // every sample of the launch is committed
        unsigned int* done = P.ring_done + sh * (P.ring_mask + 1u) + (w & P.ring_mask);
        if (ring_load(done) != n_valid) return; // the next sample is not complete
        unsigned int got = 0;
        if (lane == 0) {
            unsigned int expect = w;
            got = __hip_atomic_compare_exchange_strong((ring_word_ptr)word, &expect, w | RING_BUSY, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? 1u : 0u;
        }
        if (!__builtin_amdgcn_readfirstlane((int)got)) continue;
        // ---- sample w of the shard: c += L / spp for every pixel slot (Render.cuh:348); four slots per lane and round trip ----
        const float4* Lr = P.L + (size_t)(w & P.ring_mask) * P.ring_stride + (size_t)sh * P.spsh;
        const bool from_zero = P.sample_begin + w == 0u;
        for (uint32_t i0 = (uint32_t)lane; i0 < P.spsh; i0 += 256u) {
            float l[4][3], c[4][3];
            bool ok[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t i = i0 + 64u * (uint32_t)u, slot = sh * P.spsh + i;
                uint32_t pi, pj;
                ok[u] = i < P.spsh && slot < P.nslots && slot_to_pixel(slot, P.rank, P.world, P.n_tiles, P.tiles_x, P.tiles_x_div, P.width, P.height, pi, pj);
                l[u][0] = l[u][1] = l[u][2] = 0.0f; c[u][0] = c[u][1] = c[u][2] = 0.0f;
                if (ok[u]) {
                    l[u][0] = ring_loadf(&Lr[i].x); l[u][1] = ring_loadf(&Lr[i].y); l[u][2] = ring_loadf(&Lr[i].z);
                    if (!from_zero) { c[u][0] = ring_loadf(P.accum + slot); c[u][1] = ring_loadf(P.accum + P.nslots + slot); c[u][2] = ring_loadf(P.accum + 2ull * P.nslots + slot); }
                }
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t slot = sh * P.spsh + i0 + 64u * (uint32_t)u;
                if (ok[u]) {
                    ring_storef(P.accum + slot, c[u][0] + l[u][0] / P.spp_f);
                    ring_storef(P.accum + P.nslots + slot, c[u][1] + l[u][1] / P.spp_f);
                    ring_storef(P.accum + 2ull * P.nslots + slot, c[u][2] + l[u][2] / P.spp_f);
                }
            }
        }
        if (lane == 0) __hip_atomic_store((ring_word_ptr)done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // ready for sample w + ring samples
        ring_wait_mem(); // accumulator and counter are written before the shard is handed on
        unsigned int prev = 0;
        if (lane == 0) prev = __hip_atomic_exchange((ring_word_ptr)word, w + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        (void)__builtin_amdgcn_readfirstlane((int)prev); // (returned: the hand-over is performed before the next look at the counters)
    }
}

// The lanes with fin_key != ~0u have written the radiance of a finished work item of (shard, ring slot) = (fin_key >> 16, fin_key & 0xffff):
// one atomic per distinct key, and the commit of whatever that completes.  All 64 lanes.
__device__ __forceinline__ void ring_publish(const LParams& P, const uint32_t fin_key)
{
    unsigned long long todo = __ballot(fin_key != ~0u);
    if (!todo) return;
    // (the pixel count of the first key's shard -- nearly always the only key -- is fetched under the same wait as the stores)
    const uint32_t first_sh = (uint32_t)__builtin_amdgcn_readlane((int)fin_key, __ffsll((long long)todo) - 1) >> 16;
    const uint32_t first_valid = ring_load(P.ring_state + first_sh * ITEM_STRIDE + 1);
    ring_wait_mem(); // the radiance is in memory before it is counted
    const int lane = threadIdx.x & 63;
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const uint32_t key = (uint32_t)__builtin_amdgcn_readlane((int)fin_key, leader);
        const unsigned long long m = __ballot(fin_key == key);
        const uint32_t sh = key >> 16, n = (uint32_t)__popcll(m);
        unsigned int old = 0;
        if (lane == leader) old = __hip_atomic_fetch_add((ring_word_ptr)(P.ring_done + sh * (P.ring_mask + 1u) + (key & 0xffffu)), n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        old = (unsigned int)__builtin_amdgcn_readlane((int)old, leader);
        // (told to the compiler as next to never: the loops of the commit would otherwise weigh in its register allocation like the
        // traversal loops and push their scalars out into vector-register lanes)
        if (__builtin_expect_with_probability(old + n == (sh == first_sh ? first_valid : ring_load(P.ring_state + sh * ITEM_STRIDE + 1)), 0, 0.999999))
            ring_commit(P, sh);
        todo &= ~m;
    }
}

#define LOGIC_TABLE_MAX 64 /* materials / lights kept in LDS when they fit */

// Materials and lights are tiny tables read by every lane: LDS copies when they fit.
template <bool LDS_TABLES> struct Tables {
    const float4* mats;
    const uint4* lights;
};
template <bool LDS_TABLES>
__device__ __forceinline__ float4 mat_row(const Tables<LDS_TABLES>& tb, uint32_t mat, int row)
{
    if (LDS_TABLES) return tb.mats[mat * 3 + row]; // (k_logic's copies in LDS: not global memory)
    return gld(&tb.mats[mat * 3 + row]);
}

// Sets up next-event sample q of the current vertex: Render.cuh:262-272 (+ :274-283 evaluated
// ahead of the visibility test; the value is only added if the shadow ray is not blocked).
// lg = lights[q / lsn] (sample q = light li, repetition sj; the draw index li * lsn + sj is q itself), loaded by the caller.
__device__ __forceinline__ void setup_shadow_lg(const LParams& P, Lane& s, F3 f_r, const uint4 lg)
{
    const DevScene& sc = P.sc;
    U4 rl = rng_draw(P.seed, s.pixel_index, s.k, s.depth, RNG_NEE, s.q);
    const uint32_t ti = rl.x - fast_div(rl.x, lg.z, lg.w) * lg.y; // rand % triangle count (DeviceLights.cuh:35)
    const float4* lt = sc.ltri + (size_t)(lg.x + ti) * 4;
    float4 l0 = gld(lt), l1 = gld(lt + 1), l2 = gld(lt + 2), l3 = gld(lt + 3);
    float alpha = rng_uniform(rl.y); // DeviceTriangle.cuh:69-71
    float beta = rng_uniform(rl.z) * (1 - alpha);
    float gamma = 1 - alpha - beta;
    F3 lv1 = f3(l0.x, l0.y, l0.z), lv2 = f3(l0.w, l1.x, l1.y), lv3 = f3(l1.z, l1.w, l2.x);
    F3 lpos = add3(add3(scalel3(alpha, lv1), scalel3(beta, lv2)), scalel3(gamma, lv3));
    F3 dist = sub3(lpos, s.pos);
    F3 dir = unit3(dist);
    s.ro = s.pos;
    s.rd = unit3(dir);     // Ray normalises again (Ray.cuh:13)
    s.tl = dist.x / dir.x; // Render.cuh:272
    s.kind = RAY_SHADOW;
    float tl = norm3(dist);
    float t2 = tl * tl;
    float cos_theta = dot3(dir, s.nrm);
    float cos_theta_2 = -dot3(dir, f3(l2.y, l2.z, l2.w));
    cos_theta = cos_theta > 0.0f ? cos_theta : 0.0f;
    cos_theta_2 = cos_theta_2 > 0.0f ? cos_theta_2 : 0.0f;
    // ((((Le*fr)*cos)*cos2)*inv_pdf)/t2)/lsn  (Render.cuh:283)
    F3 c = mul3(f3(l3.x, l3.y, l3.z), f_r);
    c = scale3(c, cos_theta);
    c = scale3(c, cos_theta_2);
    c = scale3(c, l3.w);
    c = div3(c, t2);
    if (P.inv_lsn_pow2 != 0.0f) c = scale3(c, P.inv_lsn_pow2); // == c / lsn bit for bit (LParams)
    else c = div3(c, (float)P.lsn);
    s.c = c;
}
template <bool LDS_TABLES>
__device__ __forceinline__ void setup_shadow(const LParams& P, const Tables<LDS_TABLES>& tb, Lane& s, F3 f_r)
{
    const uint32_t li = fast_div(s.q, P.lsn_div.m, P.lsn_div.sh);
    setup_shadow_lg(P, s, f_r, LDS_TABLES ? tb.lights[li] : gld(&tb.lights[li]));
}

// The id plane of a path in k_mega3 is its work item alone (4 B); pixel and sample are worked out from it again wherever they are
// needed (two multiply-shift divisions and the tile arithmetic).  Against keeping (pixel, sample, item) as 16 B, measured on C2 with
// one --pmc pass per counter: memory-side traffic 431.7 -> 382.7 GB per launch, L2 miss rate 0.495 -> 0.467, vector instructions
// + 1.2 %, frame 96.7 -> 95.5 ms (profiles/r03_traffic_id_plane.txt).  (The wavefront pipeline keeps the 16-byte entries.)
// CRT_X_NOVN (round 5; 0 = the round-4 layout): k_mega3 keeps no `vn` plane (normal, material of the current vertex:
// 16 B read in every LA and LB visit, written at every vertex) -- both are a function of the vertex's triangle, which travels in the second
// word of an 8-byte id plane, and come from the L2-resident 16 B / triangle table tri_nm, one dependent load later.  Measured (C2, one
// box): L2 <-> fabric traffic 348.6 -> 320.0 GB per frame (-8.2 %), L2 miss rate 0.475 -> 0.449, frame 78.0 -> 77.2 ms; veach-mis spp 256
// 77.9 -> 77.5 ms.  (The wavefront pipeline keeps its vn plane.)
#ifndef CRT_X_NOVN
#define CRT_X_NOVN 1
#endif
// Round 6: in every mode but CRT_TRAVERSAL_REFERENCE the triangle rides in the w word of the cc plane instead -- written
// with every next-event sample anyway -- and the id plane is read as the work item alone (TRI = false); see logic_A.
#if CRT_X_NOVN
template <bool RING = false, bool TRI = true>
__device__ __forceinline__ uint4 load_path_id(const LParams& P, const uint32_t g)
{
    typedef uint32_t u2v_ __attribute__((ext_vector_type(2)));
    u2v_ w;
    if (TRI) w = *(const CRT_GAS u2v_*)((const uint2*)P.pool.id + g);
    else { w.x = *(const CRT_GAS uint32_t*)((const uint32_t*)P.pool.id + g); w.y = 0u; } // (the work items alone, densely: the first half of the plane)
    uint32_t pixel_index, k, pi, pj;
    bool valid;
    decode_item<RING>(P, w.x, pixel_index, k, valid, pi, pj);
    return make_uint4(pixel_index, k, w.x, w.y);
}
template <bool TRI = true>
__device__ __forceinline__ void store_path_id(const LParams& P, const uint32_t g, const uint32_t item)
{
    typedef uint32_t u2v_ __attribute__((ext_vector_type(2)));
    if (!TRI) { *(CRT_GAS uint32_t*)((uint32_t*)P.pool.id + g) = item; return; }
    u2v_ w; w.x = item; w.y = 0u; // (triangle 0: a valid row of tri_nm for the speculative load of a path's first visit)
    *(CRT_GAS u2v_*)((uint2*)P.pool.id + g) = w;
}
__device__ __forceinline__ void store_path_tri(const LParams& P, const uint32_t g, const uint32_t tri)
{
    *(CRT_GAS uint32_t*)((uint32_t*)((uint2*)P.pool.id + g) + 1) = tri;
}
#else
template <bool RING = false, bool TRI = true>
__device__ __forceinline__ uint4 load_path_id(const LParams& P, const uint32_t g)
{
    const uint32_t item = gld((const uint32_t*)P.pool.id + g);
    uint32_t pixel_index, k, pi, pj;
    bool valid;
    decode_item<RING>(P, item, pixel_index, k, valid, pi, pj);
    return make_uint4(pixel_index, k, item, 0u);
}
template <bool TRI = true>
__device__ __forceinline__ void store_path_id(const LParams& P, const uint32_t g, const uint32_t item)
{
    *(CRT_GAS uint32_t*)((uint32_t*)P.pool.id + g) = item;
}
#endif

// Backward recursion over the vertex records, deepest first: Render.cuh:238-326.
template <bool LDS_TABLES>
__device__ __forceinline__ F3 finish_path(const LParams& P, const Tables<LDS_TABLES>& tb, uint32_t slot, int deepest, bool emissive, F3 ke)
{
    const Pool& pl = P.pool;
    F3 L = f3(0.0f, 0.0f, 0.0f);
    if (deepest < 0) return L;
    const float inv_pdf_sphere = (float)(2.0f * 3.14159265358979323846); // Global.h:96-99
    if (emissive) {
        L = deepest == 0 ? add3(f3(0.0f, 0.0f, 0.0f), ke) : f3(0.0f, 0.0f, 0.0f); // :249-255, :323
    } else {
        float4 a = gld_rec(&pl.rec_a[(size_t)deepest * pl.n + slot]);
        L = add3(f3(0.0f, 0.0f, 0.0f), f3(a.x, a.y, a.z)); // final hit: direct light only (:316-319)
    }
    // The recursion is a serial chain, but its loads are not: the records (and material rows) of CRT_FINISH_PF vertices are
    // fetched together, so a chunk costs two memory round trips instead of two per vertex (lanes with fewer vertices re-read
    // vertex 0 and skip the arithmetic).
#ifndef CRT_FINISH_PF
#define CRT_FINISH_PF 4
#endif
    for (int v = deepest - 1; v >= 0; v -= CRT_FINISH_PF) {
        float4 a[CRT_FINISH_PF], fm[CRT_FINISH_PF];
        uint32_t mat[CRT_FINISH_PF];
#pragma unroll
        for (int j = 0; j < CRT_FINISH_PF; j++) {
            const int vj = v - j > 0 ? v - j : 0;
            a[j] = gld_rec(&pl.rec_a[(size_t)vj * pl.n + slot]);
            mat[j] = __float_as_uint(gld_rec(&pl.rec_b[(size_t)vj * pl.n + slot]).w);
        }
#pragma unroll
        for (int j = 0; j < CRT_FINISH_PF; j++) fm[j] = mat_row(tb, mat[j], 0);
#pragma unroll
        for (int j = 0; j < CRT_FINISH_PF; j++) {
            if (v - j >= 0) {
                F3 ind = mul3(L, f3(fm[j].x, fm[j].y, fm[j].z)); // L (.) f_r * cos * inv_pdf / P_RR  (:293)
                ind = scale3(ind, a[j].w);
                ind = scale3(ind, inv_pdf_sphere);
                ind = div3(ind, P.p_rr);
                L = add3(ind, f3(a[j].x, a[j].y, a[j].z)); // :323
            }
        }
    }
    return L;
}

struct AParams {
    uint32_t width, height, spp;
    uint32_t rank, world, tiles_x, n_tiles;
    uint32_t nslots;
    uint32_t chunk_samples;
    uint32_t first_chunk, last_chunk, tiled_output;
    const float4* L;
    float* accum;      // 3 planes of nslots (running sum across chunks)
    uint8_t* out_rgb;
    float* out_mean;   // may be null
};

// 1.0f / x, bit for bit, in 3 instructions + a guard instead of the 12 of the IEEE division expansion: v_rcp_f32 and one
// Newton step in FMA.  Verified EXHAUSTIVELY on gfx950 (tools/exhaustive/rcp_check.hip, all 2^32 inputs; crt_selftest()
// repeats the check through the C ABI): the bits differ from the division's only for zero / denormal x, |x| >= 2^126
// (denormal quotient) and infinities -- those lanes take the division itself behind a wave-uniform branch.
__device__ __forceinline__ bool rcp_short_ok(const float x)
{
    const float ax = absf(x);
    return (ax >= 0x1p-126f) & (ax < 0x1p126f); // exponent field in [1, 252]; false for NaN
}
// the same guard for two / three values at once: IEEE 754-2019 minimum / maximum (v_minimum3_f32 / v_maximum3_f32) return NaN
// if any operand is one, and a NaN fails both comparisons
__device__ __forceinline__ bool rcp_short_ok2(const float x, const float y)
{
    const float ax = absf(x), ay = absf(y);
    return (__builtin_elementwise_minimum(ax, ay) >= 0x1p-126f) & (__builtin_elementwise_maximum(ax, ay) < 0x1p126f);
}
__device__ __forceinline__ bool rcp_short_ok3(const float x, const float y, const float z)
{
    const float ax = absf(x), ay = absf(y), az = absf(z);
    return (__builtin_elementwise_minimum(__builtin_elementwise_minimum(ax, ay), az) >= 0x1p-126f) &
           (__builtin_elementwise_maximum(__builtin_elementwise_maximum(ax, ay), az) < 0x1p126f);
}
// |x|, |y|, |z| all <= FLT_MAX (false for a NaN, as the three comparisons are)
__device__ __forceinline__ bool finite3(const float x, const float y, const float z)
{
    return __builtin_elementwise_maximum(__builtin_elementwise_maximum(absf(x), absf(y)), absf(z)) <= FLT_MAX;
}
__device__ __forceinline__ float rcp_short(const float x)
{
    const float r0 = __builtin_amdgcn_rcpf(x);
    return __builtin_fmaf(__builtin_fmaf(-x, r0, 1.0f), r0, r0);
}
__device__ __forceinline__ float rcp_ieee(const float x)
{
    float r = rcp_short(x);
    const bool ok = rcp_short_ok(x);
    if (__builtin_amdgcn_ballot_w64(!ok)) {
        if (!ok) r = 1.0f / x;
    }
    return r;
}


} // namespace crtk
#endif

// cudaraytracing_amd/csrc/crt_kernels.hip -- device layer of libcrt.so.
//
// Hand-written HIP for gfx950 (MI355X).  Replaces the reference's
// view_render_kernel / cast_ray_v2 / DeviceBVH::intersect
// (include/Render.cuh:199-354, include/DeviceBVH.cuh:87-170).
//
// The reference runs one thread per pixel through the whole spp loop; measured
// on MI355X that shape keeps only ~14 % of the lanes busy (paths end at
// different depths, rays at different node counts).  This build is a wavefront
// path tracer instead, with the path state resident in HBM (MI355X has the
// capacity and bandwidth to trade ~200 B of streaming per ray for full waves):
//
//   pool          N path slots (structure-of-arrays float4 planes, coalesced).
//   k_logic       one thread per slot: consumes the result of the slot's last
//                 ray, advances the path (hit processing, next-event sampling,
//                 Russian roulette, regeneration of finished paths from a global
//                 work counter) until it needs another ray, and writes that ray.
//   k_trace       one ray per slot: LDS-stack BVH traversal, writes (t, tri).
//   k_accumulate  per pixel, sums L_k / spp in sample order (Render.cuh:348),
//                 tone-maps (Render.cuh:350), writes RGB8 + float mean.
//
// Every random draw is addressed explicitly (Philox counter = sample / depth /
// purpose / index, crt_detmath.h), so next-event estimation runs when a vertex
// is found while the radiance recursion is still evaluated deepest-vertex-first
// in the reference's float order (Render.cuh:238-326) from 32 B per-vertex
// records.  Results are bit-identical to the CPU oracle (oracle/crt_oracle.cpp).
// Build: -ffp-contract=off, correctly rounded fp32 divide/sqrt (build.py).
#include "../../include/crt.h"
#include "crt_accel.h"
#include "crt_device.h"
#include "crt_trace.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cfloat>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <string>
#include <type_traits>
#include <vector>

using namespace crtdev;

extern "C" void crt_set_last_error_(const char* msg);

namespace {

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

enum { ST_DEAD = 0, ST_NEW = 1, ST_HIT = 2, ST_PROBE = 3, ST_SHADOW = 4 };
enum { RAY_NONE = 0, RAY_CLOSEST = 1, RAY_SHADOW = 2 };
#define ITEM_NONE 0xffffffffu

// Path pool, structure of arrays; every plane has `n` entries.
struct Pool {
    float4* ro;   // ray origin.xyz, t_limit (shadow rays: Render.cuh:272)
    float4* rd;   // ray direction.xyz (normalised as Ray does), bits(ray kind)
    float4* vx;   // current vertex position.xyz, bits(triangle)
    float4* la;   // next-event accumulator L_dir.xyz of the current vertex, bits(depth | stage << 8 | sample << 16)
    float4* cc;   // contribution of the in-flight shadow ray .xyz, bits(work item)
    float4* vn;   // normal.xyz and bits(material) of the current vertex (of the PREVIOUS vertex while a bounce ray is in flight)
    uint4* id;    // pixel index, sample index, work item, unused -- written once per path
    float2* res;  // result of the slot's last ray: t, bits(triangle or -1)
    float4* rec_a; // [depth][n]: L_dir.xyz of that vertex, cos to the next vertex
    float4* rec_b; // [depth][n]: incoming direction.xyz, bits(material)
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
#define ST_WAIT 6 /* the ray slot holds a work item it may not start yet */
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
template <bool RING = false>
__device__ __forceinline__ uint4 load_path_id(const LParams& P, const uint32_t g)
{
    const uint32_t item = gld((const uint32_t*)P.pool.id + g);
    uint32_t pixel_index, k, pi, pj;
    bool valid;
    decode_item<RING>(P, item, pixel_index, k, valid, pi, pj);
    return make_uint4(pixel_index, k, item, 0u);
}
__device__ __forceinline__ void store_path_id(const LParams& P, const uint32_t g, const uint32_t item)
{
    *(CRT_GAS uint32_t*)((uint32_t*)P.pool.id + g) = item;
}

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
        float4 a = gld(&pl.rec_a[(size_t)deepest * pl.n + slot]);
        L = add3(f3(0.0f, 0.0f, 0.0f), f3(a.x, a.y, a.z)); // final hit: direct light only (:316-319)
    }
    // The recursion is a serial chain, but its loads are not: the records (and material rows) of CRT_FINISH_PF vertices are
    // fetched together, so a chunk costs two memory round trips instead of two per vertex (lanes with fewer vertices re-read
    // vertex 0 and skip the arithmetic).
#define CRT_FINISH_PF 4
    for (int v = deepest - 1; v >= 0; v -= CRT_FINISH_PF) {
        float4 a[CRT_FINISH_PF], fm[CRT_FINISH_PF];
        uint32_t mat[CRT_FINISH_PF];
#pragma unroll
        for (int j = 0; j < CRT_FINISH_PF; j++) {
            const int vj = v - j > 0 ? v - j : 0;
            a[j] = gld(&pl.rec_a[(size_t)vj * pl.n + slot]);
            mat[j] = __float_as_uint(gld(&pl.rec_b[(size_t)vj * pl.n + slot]).w);
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
#define REFILL_MIN 32
#define LEAF_MIN 24
#define SLOT_SHARDS 64
#define SLOT_STRIDE 32

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

// ----------------------------------------------------------- megakernels ----
// Fused forms of the two kernels above: one persistent launch per chunk, path logic and traversal in the same
// waves, rays and results never leave the chip (no rounds, no relaunches, no per-round drain); the path state planes
// (80 B per path + 32 B vertex records) stay L2 / MALL resident because there are only as many paths as resident
// rays.  (The first such kernel kept one ray per lane in registers: 42 % lane utilisation, removed.)
struct MParams {
    LParams P;
    DevScene sc;
    unsigned long long* counters;
    int2* spill;
    uint32_t spill_stride;
    int32_t stack_cap;
    int32_t logic_min, leaf_min;
};

#ifndef POOL_LV
#define POOL_LV 4 /* 32-bit traversal stack levels kept in LDS per ray (twice as many of 16 bits, Pool3LdsT); deeper levels spill to global memory */
#endif

// --------------------------------------- megakernel, queued sub-phases ----
// Its predecessor k_mega2 (wave-private LDS pool, rays regrouped by phase with ballot / prefix rank / ds_permute, one logic
// phase; 223 ms on C2, removed) spent 22 % of its cycles in a logic phase whose sections each serve 20-60 % of the gathered
// lanes, 11 % in the census / permute gather, and its inner-node batches average 47 of 64 lanes
// (-DCRT_STAMPS counters).  Sensitivity probes (tools/diag_sens.sh) show the kernel is bound by vector
// instruction ISSUE: every wave instruction added to the inner step costs ~5 SIMD cycles, additively, so
// the design goal of k_mega3 is instructions per ray:
//   * regrouping by QUEUES: every phase owns a ring of ray ids in LDS; a batch is the 64 oldest ids of the
//     chosen ring (one ds_read_u8), and a processed ray is appended to the ring of its new phase at
//     count + prefix-popcount of the ballot (one ds_write_b8).  Counts and heads are wave-uniform scalars;
//     the cost is independent of the pool size.
//   * the path logic is three phases of its own, so that a gathered lane only runs what its path needs:
//       LA  result of a shadow / closest / probe ray -> (enter the vertex) -> next next-event sample
//       LB  last next-event sample consumed -> vertex record, Russian roulette, bounce ray
//       LC  path ends (miss / emitter / roulette) -> backward recursion, next work item, camera ray
//     the traversal step routes a finished ray from flag bits in its LDS record; a phase that finds the
//     path belongs elsewhere (emitter found in LA, roulette stop in LB) parks it there without a ray.
//   * 1/direction lives in the LDS record (64 B per ray: origin, direction, 1/direction, one distance,
//     best triangle, node, flags, 3 stack levels, ring slots), so the inner step has no divisions;
//   * the nodes of the 2-wide trees are stored as (left, right) PAIRS per coordinate, those of the 4-wide tree plane-major (four
//     children per float4), so that child boxes go through v_pk_add_f32 / v_pk_mul_f32 two at a time; rays whose plane
//     distances are all finite (start_ray; all but a handful) walk the 4-wide tree, whose near / far planes are picked by the
//     load address (the reference's sign swap) and combined with v_max3 / v_min3; the others keep the reference formula
//     with its NaN behaviour (DeviceBVH.cuh:97-121) and walk the reference topology;
//   * the (<= 2) triangles of a leaf are one 80 B record, both Moeller-Trumbore tests run as one packed
//     computation (same operations per triangle, two at a time).
#define PH3_INNER 0
#define PH3_LEAF 1
#define PH3_LA 2
#define PH3_LB 3
#define PH3_LC 4
#define PH3_N 5
#define PH3_WAIT 5 /* commit ring: ray slots that hold a work item they may not start yet -- a ring like the others, but outside PH3_N:
                      only the LC phase feeds it and only the LC phase looks at it */
#define PH3_NONE 7
#define ST_FIN 5   /* path complete, backward recursion pending (q bit 0: the deepest vertex is an emitter) */
#define ST_NEED 6  /* vertex entered with zero next-event samples: straight to roulette */
// word D of the ray record: traversal stack depth (bits 0-7), best triangle - first triangle of its leaf (bits 8-23), flags
#define RF_ANYHIT 0x1000000u   /* traversal stops at the first accepted hit closer than the light */
#define RF_SHADOW 0x2000000u   /* (NewRay only) the ray is a next-event sample ... */
#define RF_LAST 0x4000000u     /* (NewRay only) ... and the last one of its vertex */
#define RF_PROBE 0x8000000u    /* (NewRay only) SPECULAR emitter probe */
/* In the RECORD bits 25-26 hold instead where the ray goes once its traversal is over, as phase - PH3_LA: 0 = LA (a next-event sample
   that is not the last of its vertex, a probe, a closest-hit ray that found a surface), 1 = LB (the last next-event sample), 2 = LC
   (a closest-hit ray that has found nothing so far).  A closest-hit ray that records a hit clears bit 26 -- LC becomes LA, LA and LB
   stay -- so the route of a finished ray is two instructions (it was a four-way select over five flag bits, eight). */
#define RR_ROUTE_SHIFT 25
#define RR_ROUTE_MASK 0x6000000u
#define RR_ROUTE_LC_BIT 0x4000000u
#define RF_EXACT 0x10000000u   /* reference box arithmetic (non-finite operands) */
#define RF_HASHIT 0x20000000u  /* closest-hit ray: a hit is recorded (T = its distance) */
#define RF_SKIP 0x40000000u    /* (NewRay only) next-event sample with a zero contribution: answered without traversal */
#define RF_QUERY 0x80000000u   /* crt_intersect: a bare closest-hit query; its result goes straight back to LC */
#ifndef POOL3_P
#define POOL3_P 164         /* 164 x 56 B + rings = 10 004 B: 16 waves per CU (measured with the 16-bit stack layout: 148 rays x 64 B records
                               with 1/d and six levels 102.5 ms, 176 x 52 B with six levels 100.2, 164 x 56 B with eight levels 98.8, 156 x 60 B with
                               ten 100.0) */
#endif
#define POOL3_QCAP ((POOL3_P + 3) & ~3) /* ring capacity (any number >= POOL3_P: indices wrap by compare, not by mask); ids fit a byte */
static_assert(POOL3_P <= 256, "ray ids of a pool must fit a byte (ring entries are uint8_t)");
static_assert(POOL3_QCAP >= POOL3_P, "a ring must hold every ray of the pool");
#define CRT_MEGA3_MAX_STACK 255 /* the traversal stack depth is kept in 8 bits of the record's word D */
#define CRT_MEGA3_MAX_LEAF 65535 /* best-triangle offset inside its leaf is kept in 16 bits */

typedef float v2f __attribute__((ext_vector_type(2)));

#ifndef CRT_WAVES
#define CRT_WAVES 4   /* waves per SIMD the kernel is compiled for; the LDS footprint of a pool must allow it (160 KiB per CU) */
#endif
// R16: the traversal stack holds 16-bit node refs, twice as many levels in the same bytes (scenes whose 4-wide tree and leaf records
// number at most 32 768 each: crt_scene::ref16_ok).  The levels beyond LDS cost a wave-uniform branch with 64-bit address arithmetic,
// global stores and -- in the pop -- an exposed global load whenever ANY ray of a batch is that deep, which with three levels is
// most batches (stamps: 830 of 5 800 cycles of an inner step, 560 of 4 100 of a leaf step); with six it is rare.  A ray on the
// reference-arithmetic path (RF_EXACT: refs of the 2-wide trees, which do not fit) keeps its whole stack in the global area then.
template <bool R16_>
struct Pool3LdsT {
    static constexpr bool R16 = R16_;
    static constexpr bool DEC = false;
    static constexpr int P = POOL3_P, QCAP = POOL3_QCAP;
    static constexpr int LV = R16_ ? 2 * POOL_LV : POOL_LV;
    typedef typename std::conditional<R16_, short, int>::type stk_t;
    float4 A[POOL3_P];           // origin.xyz, T = distance to the light (any-hit rays) | best hit distance (closest-hit rays)
    float4 B[POOL3_P];           // direction.xyz, bits(best triangle, -1 = none)
    int node[POOL3_P];           // current node ref
    stk_t stk[LV][POOL3_P];      // traversal stack (node refs); deeper levels spill to global memory
    uint32_t D[POOL3_P];         // stack depth | leaf offset << 8 | RF_* flags
    uint8_t ring[PH3_N + 1][POOL3_QCAP];
    uint32_t waitq;              // ring PH3_WAIT: entries | head << 8 | tail << 16 (kept here, not in scalar registers: only the LC phase uses it)
    __device__ __forceinline__ uint8_t* rq(const int p) { return ring[p]; }
};
typedef Pool3LdsT<false> Pool3Lds;
static_assert(sizeof(Pool3LdsT<true>) == sizeof(Pool3Lds), "16-bit stack entries: twice the levels in the same bytes");
static_assert(sizeof(Pool3Lds) * 4 * CRT_WAVES <= 160 * 1024, "the pool does not fit CRT_WAVES waves per SIMD into 160 KiB of LDS");

// ---- decoupled leaves (DEC): the pool of the kernels whose leaf tests are work items of their own ----
// A ray walks the INNER nodes only.  Every leaf child whose box it hits becomes an entry (ray, leaf record) of the wave's leaf queue,
// and the ray goes on at once; the leaf step takes 64 entries -- always a full batch -- tests the record's triangles against the
// entry's ray and folds an accepted hit into the ray's record with one LDS atomic minimum over (distance, ~triangle), which is the
// reference's own tie rule (crt_trace.h: among equal distances the largest leaf start wins; inside a leaf the first triangle, which
// the step resolves in registers), so the order in which a ray's leaves are tested cannot matter.  The record counts its entries in
// flight; the ray is finished when its stack is empty and that count is zero, and whichever step sees that routes it.  What it buys:
// the leaf step has no stack, no node, no ring push (it was as long as the arithmetic it carries), its batches are full, a ray
// leaves the inner ring once instead of once per leaf, and the stack holds inner nodes only -- six 16-bit levels cover scenes of
// 32 768 four-wide nodes whatever the number of leaves (24-bit leaf refs travel in the queue entries).
#ifndef LEAFQ_CAP
#define LEAFQ_CAP 256 /* entries of the leaf queue, a power of two; an inner batch is cut to (free entries) / 4 rays */
#endif
static_assert((LEAFQ_CAP & (LEAFQ_CAP - 1)) == 0 && LEAFQ_CAP >= 128, "leaf queue: a power of two, room for half a batch of inner steps");
#define RD_PEND_SHIFT 8
#define RD_PEND_MASK 0x3ff00u  /* word D, bits 8-17: leaf-queue entries of the ray that have not been tested yet */
#define RD_FIN 0x8000000u      /* word D: the traversal of the inner nodes is over */
#define LEAF_REC_MAX 0x7fffffu /* a queue entry is ray | leaf record << 8, and a node's row [7] keeps the sign bit for "leaf" */
template <bool R16_, bool RING_>
struct Pool4LdsT {
    static constexpr bool R16 = R16_;
    static constexpr bool DEC = true;
#ifdef POOL4_P
    static constexpr int P = POOL4_P;
#else
    static constexpr int P = RING_ ? 148 : 152;
#endif
    static constexpr int QCAP = (P + 3) & ~3;
    static constexpr int LV = R16_ ? 6 : 3;
    typedef typename std::conditional<R16_, short, int>::type stk_t;
    float4 A[P];                 // origin.xyz, the distance an accepted hit must stay below by more than EPSILON: the light's for an any-hit ray, +inf otherwise
    float4 B[P];                 // direction.xyz, bits(current node ref)
    unsigned long long best[P];  // the ray's answer so far: bits(distance) << 32 | ~triangle; FLT_MAX << 32 | 0 = nothing
    stk_t stk[LV][P];            // traversal stack (inner nodes only); deeper levels spill to global memory
    uint32_t D[P];               // stack depth (bits 0-7) | entries in flight (RD_PEND_MASK) | RF_* flags, RD_FIN
    uint32_t leafq[LEAFQ_CAP];
    uint8_t ring[(RING_ ? 5 : 4)][QCAP]; // INNER, LA, LB, LC (, WAIT)
    uint32_t waitq;
    __device__ __forceinline__ uint8_t* rq(const int p) { return ring[p == PH3_INNER ? 0 : p - 1]; }
};
static_assert(sizeof(Pool4LdsT<true, false>) * 4 * CRT_WAVES <= 160 * 1024 && sizeof(Pool4LdsT<true, true>) * 4 * CRT_WAVES <= 160 * 1024 &&
              sizeof(Pool4LdsT<false, false>) * 4 * CRT_WAVES <= 160 * 1024 && sizeof(Pool4LdsT<false, true>) * 4 * CRT_WAVES <= 160 * 1024,
              "the DEC pool does not fit CRT_WAVES waves per SIMD into 160 KiB of LDS");
static_assert(Pool4LdsT<true, false>::P <= 256, "ray ids of a pool must fit a byte");

struct MParams3 {
    MParams M;
    int* spill;                  // [level - POOL_LV][pool slot] stack entries beyond the LDS levels
    uint32_t force_exact;        // CRT_FLAG_FORCE_EXACT
    int32_t dbg_loads, dbg_valu; // unused by the kernel; tools/bbprof passes the address of its counter buffer in these two dwords
};

static_assert(offsetof(MParams3, dbg_loads) == 668 && offsetof(MParams3, dbg_valu) == 672, "tools/bbprof/instrument.py reads the counter buffer's address from these two kernel-argument dwords");

struct NewRay {
    F3 o, d;
    float tl;
    uint32_t kind, flags;
};

__device__ __forceinline__ float fmin3(float a, float b, float c) { return __builtin_fminf(__builtin_fminf(a, b), c); }
__device__ __forceinline__ float fmax3(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }
__device__ __forceinline__ v2f v2(float a, float b) { v2f r; r.x = a; r.y = b; return r; }
__device__ __forceinline__ v2f v2s(float a) { v2f r; r.x = a; r.y = a; return r; }

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

// max of two wave-uniform integers on the scalar unit
__device__ __forceinline__ int smax(const int a, const int b)
{
    int r;
    asm("s_max_i32 %0, %1, %2" : "=s"(r) : "s"(a), "s"(b) : "scc");
    return r;
}

// Ring index in [0, 2 * QCAP) -> [0, QCAP).
template <int QCAP>
__device__ __forceinline__ uint32_t ring_wrap(const uint32_t x) { return min(x, x - (uint32_t)QCAP); }

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
template <int MODE, bool QUERY = false, class LDS = Pool3Lds>
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
    const int ref = (MODE != 1 && finite && !force_exact) ? sc.root4 : sc.root3_exact;
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
    const float4 la = gld(&pl.la[g]);
    const uint4 idv = load_path_id<RING>(P, g);
    const float4 vn = gld(&pl.vn[g]);
    const float4 cc = gld(&pl.cc[g]); // pending next-event contribution, .w = distance to the light sample (ST_SHADOW)
    const float res_t = qa.w;
    const int res_tri = __float_as_int(qb.w);
    const float4 gq_hit = gld(&sc.tri_nm[res_tri >= 0 ? res_tri : 0]);
    const uint32_t st = __float_as_uint(la.w);
    const uint32_t stage = (st >> 8) & 15u;
    //   round 2: material rows of the vertex the samples belong to after this visit (the new one for ST_HIT), row 1 of the
    //   vertex the ray left (specular flag, ST_HIT), and the light of the sample that is set up below
    //   (the vn plane of a slot's very first vertex has never been written: the speculative index is clamped into the table)
    const uint32_t mat_old = min(__float_as_uint(vn.w), P.n_mats - 1u);
    const uint32_t mat_cur = stage == ST_HIT ? __float_as_uint(gq_hit.w) : mat_old;
    float4 m0_cur = mat_row(tb, mat_cur, 0), m1_cur = mat_row(tb, mat_cur, 1);
    const float4 pm1_old = mat_row(tb, mat_old, 1);
    const uint32_t n_nee = (uint32_t)(sc.n_lights * P.lsn);
    const uint32_t q_next = stage == ST_SHADOW ? (st >> 16) + 1 : 0u;
    uint4 lg_next = make_uint4(0u, 1u, 0u, 0u);
    if (n_nee > 0) lg_next = gld(&tb.lights[fast_div(q_next < n_nee ? q_next : 0u, P.lsn_div.m, P.lsn_div.sh)]);
    Lane s;
    s.depth = st & 255u; s.q = st >> 16; s.stage = stage;
    s.Ld = f3(la.x, la.y, la.z);
    s.nrm = f3(vn.x, vn.y, vn.z); s.mat = __float_as_uint(vn.w);
    s.pixel_index = idv.x; s.k = idv.y; s.item = idv.z;
    s.ro = f3(qa.x, qa.y, qa.z); s.tl = 0.0f;
    s.rd = f3(qb.x, qb.y, qb.z);
    s.pos = s.ro; s.vtri = 0; s.c = f3(0.0f, 0.0f, 0.0f); s.kind = RAY_NONE;
    bool do_enter = false;
    if (stage == ST_SHADOW) {
        // visibility of next-event sample q (Render.cuh:19-27, :272-284); shadow rays start at the vertex: s.pos == s.ro
        if (!shadow_blocked<MODE>(cc.w, res_t, res_tri)) s.Ld = add3(s.Ld, f3(cc.x, cc.y, cc.z));
        s.q++;
    } else if (stage == ST_HIT) {
        // the camera / bounce ray found vertex `depth` (Render.cuh:207-213)
        const F3 pos = add3(s.ro, scalel3(res_t, s.rd)); // DeviceTriangle.cuh:50
        do_enter = true;
        if (s.depth > 0) {
            // the previous vertex (normal / material still in the vn plane) is not the deepest one: cosine of its indirect term (Render.cuh:291)
            const size_t pr = (size_t)(s.depth - 1) * pl.n + g;
            const F3 pn = s.nrm;
            float cos_prev = dot3(unit3(sub3(pos, s.ro)), pn); // prev.pos == origin of this ray
            cos_prev = cos_prev > 0.0f ? cos_prev : 0.0f;
            gst(&pl.rec_a[pr].w, cos_prev);
            if (__float_as_uint(pm1_old.w) & 2u) { // SPECULAR: emitter probe, Render.cuh:294-303
                const float ns = mat_row(tb, s.mat, 0).w;
                const float4 pb = gld(&pl.rec_b[pr]); // direction that arrived at the previous vertex
                const float delta_coeff = (float)((double)(det_expf(25 / ns) - 1) / (2.71828182845904523536 - 1));
                const F3 in = unit3(f3(pb.x, pb.y, pb.z));
                const F3 out = sub3(in, scale3(pn, 2.f * dot3(in, pn)));
                const float d_theta = (float)((double)(delta_coeff * 30) * 3.14159265358979323846 / 180);
                const float d_phi = (float)((double)(delta_coeff * 120) * 3.14159265358979323846 / 180);
                const U4 rp = rng_draw(P.seed, s.pixel_index, s.k, s.depth - 1, RNG_PROBE, 0);
                const F3 refd = unit3(sample_lobe(out, d_theta, d_phi, rng_uniform(rp.x), rng_uniform(rp.y)));
                // the probe leaves from prev.pos (= this ray's origin); the bounce direction waits in rec_b[depth]
                gst(&pl.rec_b[(size_t)s.depth * pl.n + g], make_float4(s.rd.x, s.rd.y, s.rd.z, 0.0f));
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
                float4 a = gld(&pl.rec_a[pr]);
                a.x = a.x + temp.x; a.y = a.y + temp.y; a.z = a.z + temp.z;
                gst(&pl.rec_a[pr], a);
            }
        }
        const float4 pb = gld(&pl.rec_b[(size_t)s.depth * pl.n + g]); // the bounce direction that found the current vertex
        s.rd = f3(pb.x, pb.y, pb.z);
        do_enter = true;
    }
    if (do_enter) { // a new vertex (pos, vtri) at `depth`, reached along s.rd
        float4 gq = gq_hit;
        if (stage == ST_PROBE) { // the vertex was found by the ray before the probe: its triangle waits in the vx plane
            gq = gld(&sc.tri_nm[s.vtri]);
            m0_cur = mat_row(tb, __float_as_uint(gq.w), 0); m1_cur = mat_row(tb, __float_as_uint(gq.w), 1);
        }
        s.nrm = f3(gq.x, gq.y, gq.z);
        s.mat = __float_as_uint(gq.w);
        gst(&pl.rec_b[(size_t)s.depth * pl.n + g], make_float4(s.rd.x, s.rd.y, s.rd.z, __uint_as_float(s.mat)));
        gst(&pl.vx[g], make_float4(s.pos.x, s.pos.y, s.pos.z, __uint_as_float(s.vtri)));
        gst(&pl.vn[g], make_float4(s.nrm.x, s.nrm.y, s.nrm.z, __uint_as_float(s.mat)));
        if (__float_as_uint(m1_cur.w) & 1u) { // emitter: the path ends here (Render.cuh:210)
            gst(&pl.la[g], make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(s.depth | ((uint32_t)ST_FIN << 8) | (1u << 16))));
            return PH3_LC;
        }
        s.Ld = f3(0.0f, 0.0f, 0.0f);
        s.q = 0;
        if (n_nee == 0) {
            gst(&pl.la[g], make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(s.depth | ((uint32_t)ST_NEED << 8))));
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
            return PH3_LB;
        }
    }
    gst(&pl.la[g], make_float4(s.Ld.x, s.Ld.y, s.Ld.z, __uint_as_float(s.depth | ((uint32_t)ST_SHADOW << 8) | (s.q << 16))));
    gst(&pl.cc[g], make_float4(s.c.x, s.c.y, s.c.z, s.tl));
    nr.o = s.ro; nr.d = s.rd; nr.tl = s.tl; nr.kind = RAY_SHADOW;
    nr.flags = RF_SHADOW | (s.q + 1 == n_nee ? RF_LAST : 0u) | (skip ? RF_SKIP : 0u);
    return PH3_NONE;
}

// LB: direct light of vertex `depth` is complete -> vertex record, Russian roulette, bounce (Render.cuh:210-228).
template <int MODE, bool RING = false>
__device__ __forceinline__ uint32_t logic_B(const LParams& P, const uint32_t g, const float4 qa, const float4 qb, NewRay& nr)
{
    const Pool& pl = P.pool;
    const float4 la = gld(&pl.la[g]);
    const uint4 idv = load_path_id<RING>(P, g);
    const float4 cc = gld(&pl.cc[g]); // (with the other planes, not after the stage is known: one round trip less, see logic_A)
    const uint32_t st = __float_as_uint(la.w);
    const uint32_t stage = (st >> 8) & 15u;
    uint32_t depth = st & 255u;
    F3 Ld = f3(la.x, la.y, la.z);
    if (stage == ST_SHADOW) { // the last next-event sample (Render.cuh:272-284)
        if (!shadow_blocked<MODE>(cc.w, qa.w, __float_as_int(qb.w))) Ld = add3(Ld, f3(cc.x, cc.y, cc.z));
    }
    gst(&pl.rec_a[(size_t)depth * pl.n + g], make_float4(Ld.x, Ld.y, Ld.z, 0.0f));
    bool stop = depth == CRT_BOUNCE_STACK_SIZE - 1; // bounce stack full
    U4 rb;
    rb.x = rb.y = rb.z = rb.w = 0;
    if (!stop) {
        rb = rng_draw(P.seed, idv.x, idv.y, depth, RNG_BOUNCE, 0);
        stop = rng_uniform(rb.x) > P.p_rr;
    }
    if (stop) {
        gst(&pl.la[g], make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(depth | ((uint32_t)ST_FIN << 8))));
        return PH3_LC;
    }
    const float4 vn = gld(&pl.vn[g]), vx = gld(&pl.vx[g]);
    const F3 ndir = unit3(sample_hemisphere(f3(vn.x, vn.y, vn.z), rng_uniform(rb.y), rng_uniform(rb.z)));
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
template <bool RING>
__device__ __forceinline__ int logic_C(const LParams& P, const Tables<false>& tb, const uint32_t g, PathCounters& cnt, NewRay& nr, uint32_t& fin_key)
{
    const Pool& pl = P.pool;
    const float4 la = gld(&pl.la[g]);
    const uint4 idv = load_path_id<RING>(P, g); // (with la, not after the stage is known: one round trip less, see logic_A)
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
            const float4 m2 = mat_row(tb, __float_as_uint(gld(&pl.vn[g]).w), 2);
            ke = f3(m2.x, m2.y, m2.z);
        }
        const F3 L = finish_path(P, tb, g, deepest, emissive, ke);
        if (ring) {
            const uint32_t sh = fast_div(idv.z, P.items_per_shard_div.m, P.items_per_shard_div.sh), c = idv.z - sh * P.items_per_shard;
            const uint32_t s = fast_div(c, P.spsh_div.m, P.spsh_div.sh), rs = s & P.ring_mask;
            float4* Lr = P.L + (size_t)rs * P.ring_stride + (size_t)sh * P.spsh + (c - s * P.spsh);
            ring_store16(Lr, L.x, L.y, L.z);
            fin_key = (sh << 16) | rs;
        } else {
            // written once, read once by k_accumulate after the launch: a streaming store keeps it from displacing the path state
            // and the scene in L2
            __builtin_nontemporal_store(L.x, (CRT_GAS float*)&P.L[idv.z].x);
            __builtin_nontemporal_store(L.y, (CRT_GAS float*)&P.L[idv.z].y);
            __builtin_nontemporal_store(L.z, (CRT_GAS float*)&P.L[idv.z].z);
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
                if (item >= wlo_) item = gld(&P.item_list[sh_ * P.order_window + (item - wlo_)]);
            }
        }
        bool valid; uint32_t pi, pj, pixel_index, k;
        decode_item<RING>(P, item, pixel_index, k, valid, pi, pj);
        if (!valid) continue; // padding slot of a ragged tile: take another item
        if (ring && !ring_gate_open(P, item, home_, home_word_)) { // its sample's slot of the ring is not free yet: hold the item
            if (!waiting) {
                store_path_id(P, g, item);
                gst(&pl.la[g], make_float4(0.0f, 0.0f, 0.0f, __uint_as_float((uint32_t)ST_WAIT << 8)));
            }
            return LC_WAIT;
        }
        cnt.paths++;
        if (!waiting) store_path_id(P, g, item);
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

// The two triangles of a leaf record at once: Moeller-Trumbore exactly as DeviceTriangle.cuh:39-56 + inside() :58-65 +
// the t > EPSILON filter of DeviceBVHNode::hit (DeviceBVH.cuh:37); lane .x = first triangle, .y = second.
__device__ __forceinline__ void tri_pair(const float4 g0, const float4 g1, const float4 g2, const float4 g3, const float4 g4, const F3 o, const F3 d,
                                         bool& a0, bool& a1, float& t0, float& t1)
{
    const v2f v1x = v2(g0.x, g0.y), v1y = v2(g0.z, g0.w), v1z = v2(g1.x, g1.y);
    const v2f e1x = v2(g1.z, g1.w), e1y = v2(g2.x, g2.y), e1z = v2(g2.z, g2.w);
    const v2f e2x = v2(g3.x, g3.y), e2y = v2(g3.z, g3.w), e2z = v2(g4.x, g4.y);
    const v2f sx = v2s(o.x) - v1x, sy = v2s(o.y) - v1y, sz = v2s(o.z) - v1z;
    // s1 = d x e2, s2 = s x e1 (OrthoMethods.h:106-108)
    const v2f s1x = v2s(d.y) * e2z - v2s(d.z) * e2y, s1y = v2s(d.z) * e2x - v2s(d.x) * e2z, s1z = v2s(d.x) * e2y - v2s(d.y) * e2x;
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
    const v2f gamma = (s2x * v2s(d.x) + (s2y * v2s(d.y) + s2z * v2s(d.z))) * rcp;
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
template <bool STATS, bool SORT = true, class LDS = Pool3Lds>
__device__ __forceinline__ bool inner4_step(const DevScene& sc, LDS& S, const MParams3& M, const uint32_t id, const uint32_t g, const F3 o, const F3 inv,
                                            const float bound, int& ref, int& sp, TravCounters& tc, uint32_t& max_sp
                                            )
{
    const char* nb = (const char*)sc.nodes4; // 32-bit byte offsets: scalar base + vector offset addressing
    const uint32_t noff = (uint32_t)ref * 128u;
    const uint32_t ox = noff + ((__float_as_uint(inv.x) >> 27) & 16u), oy = noff + ((__float_as_uint(inv.y) >> 27) & 16u),
                   oz = noff + ((__float_as_uint(inv.z) >> 27) & 16u); // + 16: the ray runs towards -axis, its near plane is hi
    const float4 a0 = *(const float4*)(nb + ox), a1 = *(const float4*)(nb + (ox ^ 16u));
    const float4 a2 = *(const float4*)((nb + oy) + 32), b0 = *(const float4*)((nb + (oy ^ 16u)) + 32);
    const float4 b1 = *(const float4*)((nb + oz) + 64), b2 = *(const float4*)((nb + (oz ^ 16u)) + 64);
    const float4 rf = *(const float4*)((nb + noff) + 96);
    const int top = stack_top_ahead(S, id, sp, LDS::LV);
    if (STATS) tc.inner++;
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
template <bool STATS, class LDS>
__device__ __forceinline__ bool inner4_step_dec(const DevScene& sc, LDS& S, const MParams3& M, const uint32_t id, const uint32_t g, const F3 o, const F3 inv,
                                                int& ref, int& sp, TravCounters& tc, uint32_t& max_sp, const uint32_t lq_t, uint32_t& added, bool& any_leaf,
                                                const bool enable)
{
    // (every lane of the batch runs the step, so that the appends -- ballots, the running count `added` -- stay wave-uniform; a lane
    // without `enable`, an any-hit ray that has its answer or a ray of the reference-arithmetic path, steps at the EMPTY node the host
    // puts behind the tree -- four inverted boxes: nothing is hit, appended or pushed)
    const char* nb = (const char*)sc.nodes4;
    const uint32_t noff = enable ? (uint32_t)ref * 128u : sc.empty4_off;
    const uint32_t ox = noff + ((__float_as_uint(inv.x) >> 27) & 16u), oy = noff + ((__float_as_uint(inv.y) >> 27) & 16u),
                   oz = noff + ((__float_as_uint(inv.z) >> 27) & 16u);
    const float4 a0 = *(const float4*)(nb + ox), a1 = *(const float4*)(nb + (ox ^ 16u));
    const float4 a2 = *(const float4*)((nb + oy) + 32), b0 = *(const float4*)((nb + (oy ^ 16u)) + 32);
    const float4 b1 = *(const float4*)((nb + oz) + 64), b2 = *(const float4*)((nb + (oz ^ 16u)) + 64);
    const float4 rf = *(const float4*)((nb + noff) + 112);
    const int top = stack_top_ahead(S, id, sp, LDS::LV);
    if (STATS && enable) tc.inner++;
    float t0, t1, t2, t3;
    slab_quad_pruned<false>(a0, a1, a2, b0, b1, b2, o, inv, pinf(), t0, t1, t2, t3);
    asm volatile("" : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3));
    const float inf = pinf();
    int r0 = __float_as_int(rf.x), r1 = __float_as_int(rf.y), r2 = __float_as_int(rf.z), r3 = __float_as_int(rf.w);
    const bool l0 = (t0 < inf) & (r0 < 0), l1 = (t1 < inf) & (r1 < 0), l2 = (t2 < inf) & (r2 < 0), l3 = (t3 < inf) & (r3 < 0);
#define CRT_LEAF_MASK(t_, r_) (__builtin_amdgcn_ballot_w64(t_ < inf) & __builtin_amdgcn_ballot_w64(r_ < 0))
    leafq_push(S, id, l0, CRT_LEAF_MASK(t0, r0), ((uint32_t)r0 & 0x7fffff00u) | id, lq_t, added);
    leafq_push(S, id, l1, CRT_LEAF_MASK(t1, r1), ((uint32_t)r1 & 0x7fffff00u) | id, lq_t, added);
    leafq_push(S, id, l2, CRT_LEAF_MASK(t2, r2), ((uint32_t)r2 & 0x7fffff00u) | id, lq_t, added);
    leafq_push(S, id, l3, CRT_LEAF_MASK(t3, r3), ((uint32_t)r3 & 0x7fffff00u) | id, lq_t, added);
#undef CRT_LEAF_MASK
    any_leaf = l0 | l1 | l2 | l3;
    t0 = r0 < 0 ? inf : t0; t1 = r1 < 0 ? inf : t1; t2 = r2 < 0 ? inf : t2; t3 = r3 < 0 ? inf : t3;
#define CRT_CE(ta, ra, tb, rb) { const bool sw_ = tb < ta; const float tt_ = sw_ ? tb : ta; tb = sw_ ? ta : tb; ta = tt_; const int rr_ = sw_ ? rb : ra; rb = sw_ ? ra : rb; ra = rr_; }
#ifndef CRT_DEC_UNSORTED
    CRT_CE(t0, r0, t1, r1) CRT_CE(t2, r2, t3, r3) CRT_CE(t0, r0, t2, r2)
#endif
#undef CRT_CE
    const bool c0 = t0 < inf, c1 = t1 < inf, c2 = t2 < inf, c3 = t3 < inf;
    const int l3_ = sp, l2_ = l3_ + (c3 ? 1 : 0), l1_ = l2_ + (c2 ? 1 : 0);
    constexpr int LV = LDS::LV;
    typedef typename LDS::stk_t stk_t;
    if (c3 & (l3_ < LV)) S.stk[l3_][id] = (stk_t)r3;
    if (c2 & (l2_ < LV)) S.stk[l2_][id] = (stk_t)r2;
    if (c1 & (l1_ < LV)) S.stk[l1_][id] = (stk_t)r1;
    const int sp_new = l1_ + (c1 ? 1 : 0);
    if (__builtin_amdgcn_ballot_w64((sp_new > l3_) & (sp_new > LV))) {
        if (c3 & (l3_ >= LV)) M.spill[(size_t)(l3_ - LV) * M.M.spill_stride + g] = r3;
        if (c2 & (l2_ >= LV)) M.spill[(size_t)(l2_ - LV) * M.M.spill_stride + g] = r2;
        if (c1 & (l1_ >= LV)) M.spill[(size_t)(l1_ - LV) * M.M.spill_stride + g] = r1;
    }
    sp = sp_new;
    if (STATS && (uint32_t)sp > max_sp) max_sp = (uint32_t)sp;
    if (c0) { ref = r0; return false; }
    return stack_pop_ahead(S, M, id, g, sp, ref, top, LV);
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
template <int MODE, bool STATS, bool ALL = false, bool QUERY = false, bool R16 = false, bool RING = false, bool DEC = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(CRT_WAVES, CRT_WAVES))) void k_mega3(const MParams3 M3)
{
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
    uint32_t lq_h = 0, lq_t = 0; // DEC: the leaf queue's head and tail, free-running (entries = tail - head, index = counter mod LEAFQ_CAP)
    constexpr bool commit_ring = RING;
    if (commit_ring && lane == 0) S.waitq = 0u;
    // every ray of the pool starts in LC with a path in stage NEW
    {
        const int n_valid = (int)min((uint32_t)LDS3::P, pl.n > base ? pl.n - base : 0u);
        for (int i = lane; i < n_valid; i += 64) {
            S.rq(PH3_LC)[i] = (uint8_t)i;
            pl.la[base + i] = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float((uint32_t)ST_NEW << 8));
        }
        qn[PH3_LC] = n_valid;
        qt[PH3_LC] = n_valid >= QCAP ? n_valid - QCAP : n_valid;
    }

// appends the processed rays (lane active = `on`, ray `id`) to the ring of their new phase
#define PUSH3()                                                                                                            \
    _Pragma("unroll") for (int p = 0; p < PH3_N; p++) {                                                                    \
        const bool mine = nph == (uint32_t)p; /* lanes without a ray carry PH3_NONE */                                                                        \
        const unsigned long long m = __ballot(mine);                                                                       \
        if (m) {                                                                                                           \
            /* slot = tail + number of lanes below this one that go the same way: the tail rides in as mbcnt's addend */      \
            const uint32_t slot = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, (uint32_t)qt[p])); \
            if (mine) S.rq(p)[ring_wrap<QCAP>(slot)] = (uint8_t)id;                                                            \
            const int add = (int)__popcll(m);                                                                              \
            qn[p] += add;                                                                                                  \
            qt[p] += add;                                                                                                  \
            if (qt[p] >= QCAP) qt[p] -= QCAP;                                                                  \
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
    const uint32_t id = S.rq(p)[ring_wrap<QCAP>((uint32_t)(qh[p] + lane))];                                                    \
    qh[p] += take;                                                                                                         \
    if (qh[p] >= QCAP) qh[p] -= QCAP;                                                                          \
    qn[p] -= take;                                                                                                         \
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
            // (DEC: the leaf queue counts entries, not rays; with 64 or more it is the fullest there can be and wins over the inner ring,
            // so an inner batch always finds room for 4 entries per ray of at least 48 rays)
            const int kL = min(DEC ? (int)(lq_t - lq_h) : qn[PH3_LEAF], 64) * 8 + PH3_LEAF, kI = min(qn[PH3_INNER], 64) * 8 + PH3_INNER;
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
                const F3 o = f3(qa.x, qa.y, qa.z), inv = inv3_exact(f3(qbd.x, qbd.y, qbd.z));
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
                    done = inner4_step<STATS, CRT_SORT4(MODE)>(sc, S, M3, id, g, o, inv, bound, ref, sp, tc, max_sp);
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
                    tri_pair(g0, g1, g2, g3, g4, o, d, a0, a1, t0, t1);
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
            const uint32_t slot_ = __builtin_amdgcn_mbcnt_hi((uint32_t)(m_ >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m_, (uint32_t)qt[p_])); \
            if (mine_) S.rq(p_)[ring_wrap<QCAP>(slot_)] = (uint8_t)id;                                                     \
            const int add_ = (int)__popcll(m_);                                                                            \
            qn[p_] += add_;                                                                                                \
            qt[p_] += add_;                                                                                                \
            if (qt[p_] >= QCAP) qt[p_] -= QCAP;                                                                            \
        }                                                                                                                  \
    }
        auto inner_arm_dec = [&](auto may_exact_) __attribute__((always_inline)) {
            constexpr bool MAY_EXACT = decltype(may_exact_)::value;
            if constexpr (DEC) {
            // ---- inner-node step: the child boxes; leaf children that are hit -> queue entries; nearest inner child next ----
            // The whole step, the appends to the leaf queue and to the inner ring included, runs under the mask of the batch's lanes
            // (a ballot there sees those lanes only); what the appends add to the wave-uniform cursors comes out of the region in a
            // vector register of lane 0, which is always one of them.
            const int room = (int)((uint32_t)LEAFQ_CAP - (lq_t - lq_h)) >> 2;
            const int take = min(min(64, qn[PH3_INNER]), room);
            if (STATS) { dg_b[PH3_INNER]++; dg_l[PH3_INNER] += (uint32_t)take; }
            const bool on = lane < take;
            const uint32_t id = S.rq(PH3_INNER)[ring_wrap<QCAP>((uint32_t)(qh[PH3_INNER] + lane))];
            qh[PH3_INNER] += take;
            if (qh[PH3_INNER] >= QCAP) qh[PH3_INNER] -= QCAP;
            qn[PH3_INNER] -= take;
            const uint32_t g = base + id;
            uint32_t xfer = 0;            // entries appended to the leaf queue | rays re-queued << 16 | rays of the reference-arithmetic path that ended << 24
            uint32_t nph = PH3_NONE;      // a ray that is complete (its walk is over and none of its entries is in flight): where it goes
            if (on) {
                const float4 qa = S.A[id], qb = S.B[id];
                const uint32_t qd = S.D[id];
                const uint32_t blo = (uint32_t)__hip_atomic_load(&S.best[id], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                int ref = __float_as_int(qb.w);
                const F3 o = f3(qa.x, qa.y, qa.z), inv = inv3_exact(f3(qb.x, qb.y, qb.z));
                int sp = (int)(qd & 0xffu);
                // an any-hit ray that has its answer looks no further (entries of it still in the queue are tested and change nothing)
                bool done = (qd & RF_ANYHIT) != 0 && blo != 0u;
                const bool go = !done;
                const bool ex = MAY_EXACT && (qd & RF_EXACT) != 0;
                uint32_t added = 0;
                bool any_leaf = false;
                {
                    // (a lane that does not take the step -- see inner4_step_dec -- keeps its node, depth and `done`)
                    const bool en = go && !ex;
                    int ref4 = ref, sp4 = sp;
                    const bool done4 = inner4_step_dec<STATS>(sc, S, M3, id, g, o, inv, ref4, sp4, tc, max_sp, lq_t, added, any_leaf, en);
                    if (en) { ref = ref4; sp = sp4; done = done4; }
                }
                if (MAY_EXACT) {
                    if (__builtin_amdgcn_ballot_w64(go && ex)) { // reference arithmetic on the reference topology, one thing per visit:
                        const bool lf = go && ex && ref < 0;     // a leaf ref becomes a queue entry, an inner node is stepped
                        leafq_push(S, id, lf, __builtin_amdgcn_ballot_w64(lf), ((uint32_t)~ref << 8) | id, lq_t, added);
                        if (go && ex) {
                            if (lf) {
                                any_leaf = true;
                                done = stack_pop(S, M3, id, g, sp, ref, lds_levels<LDS3>(true));
                            } else {
                                done = inner2_step<0, STATS>(sc, S, M3, id, g, o, inv, f3(qb.x, qb.y, qb.z), pinf(), ref, sp, tc, max_sp);
                            }
                        }
                    }
                }
                // the record: node, stack depth, "the walk is over" -- the count of entries in flight in between is touched by atomics only
                // (leafq_push above: those additions are in LDS before this one, same wave, in order)
                S.B[id].w = __int_as_float(ref);
                __hip_atomic_fetch_add(&S.D[id], (uint32_t)sp - (qd & 0xffu) + (done ? RD_FIN : 0u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                const bool comp = done & !any_leaf & ((qd & RD_PEND_MASK) == 0u);
                if (STATS && comp && blo != 0u) tc.hits++;
                if (__builtin_amdgcn_ballot_w64(comp)) nph = comp ? route_complete<QUERY>(qd, blo != 0u) : nph;
                // re-queue the rays that go on
                const unsigned long long mc = __builtin_amdgcn_ballot_w64(!done);
                if (mc) {
                    const uint32_t slot = __builtin_amdgcn_mbcnt_hi((uint32_t)(mc >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mc, (uint32_t)qt[PH3_INNER]));
                    if (!done) S.rq(PH3_INNER)[ring_wrap<QCAP>(slot)] = (uint8_t)id;
                }
                xfer = added | ((uint32_t)__popcll(mc) << 16);
                if (MAY_EXACT) xfer |= (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(done && (qd & RF_EXACT) != 0)) << 24;
            }
            xfer = (uint32_t)__builtin_amdgcn_readfirstlane((int)xfer);
            lq_t += xfer & 0xffffu;
            {
                const int c = (int)((xfer >> 16) & 0xffu);
                qn[PH3_INNER] += c;
                qt[PH3_INNER] += c;
                if (qt[PH3_INNER] >= QCAP) qt[PH3_INNER] -= QCAP;
            }
            if (MAY_EXACT) n_exact -= (int)(xfer >> 24);
            if (__builtin_amdgcn_ballot_w64(nph != PH3_NONE)) {
                PUSH_ONE(PH3_LA, nph == PH3_LA) PUSH_ONE(PH3_LB, nph == PH3_LB) PUSH_ONE(PH3_LC, nph == PH3_LC)
            }
            }
        };
        auto leaf_arm_dec = [&]() __attribute__((always_inline)) {
            if constexpr (DEC) {
            // ---- leaf step: 64 entries of the queue, the record's two triangles in one packed computation ----
            const int take = min(64, (int)(lq_t - lq_h));
            if (STATS) { dg_b[PH3_LEAF]++; dg_l[PH3_LEAF] += (uint32_t)take; }
            const bool on = lane < take;
            const uint32_t item = S.leafq[(lq_h + (uint32_t)lane) & (uint32_t)(LEAFQ_CAP - 1)];
            lq_h += (uint32_t)take;
            const uint32_t id = item & 0xffu;
            bool comp = false;
            uint32_t t_flags = 0, blo = 0;
            if (on) {
                const float4 qa = S.A[id], qb = S.B[id];
                const F3 o = f3(qa.x, qa.y, qa.z), d = f3(qb.x, qb.y, qb.z);
                const float Tl = qa.w;
                uint32_t rec = item >> 8;
                // the leaf's candidate: the first of its triangles among equal distances (ascending index, strict <: DeviceBVH.cuh:34-41)
                bool have = false;
                float bt = 0.0f;
                int bi = 0;
                int left = 1;
                for (int k = 0; left > 0; k++, rec++) { // one record per pair of triangles: a single pass with bvh_thresh_n <= 2
                    const float4* lg = (const float4*)((const char*)sc.leaf_geo + rec * 80u);
                    const float4 g0 = lg[0], g1 = lg[1], g2 = lg[2], g3 = lg[3], g4 = lg[4];
                    const int it = __float_as_int(g4.z);
                    if (k == 0) left = __float_as_int(g4.w);
                    const bool two = left > 1;
                    bool a0, a1;
                    float t0, t1;
                    tri_pair(g0, g1, g2, g3, g4, o, d, a0, a1, t0, t1);
                    if (STATS) { tc.tests += two ? 2u : 1u; }
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
                if (STATS) tc.leaf++;
                // across leaves: the smaller distance, among equal ones the larger leaf start (crt_trace.h) = the larger triangle index, as
                // the leaves own disjoint ascending ranges -- one 64-bit minimum over (bits(t), ~triangle); t > EPSILON > 0, so its bits order as it does
                if (__builtin_amdgcn_ballot_w64(have)) {
                    if (have) __hip_atomic_fetch_min(&S.best[id], ((unsigned long long)__float_as_uint(bt) << 32) | (unsigned long long)(uint32_t)~bi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                const uint32_t od = __hip_atomic_fetch_sub(&S.D[id], 1u << RD_PEND_SHIFT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                comp = (od & (RD_PEND_MASK | RD_FIN)) == ((1u << RD_PEND_SHIFT) | RD_FIN); // the last entry of a ray whose walk is over
                t_flags = od;
            }
            if (__builtin_amdgcn_ballot_w64(comp)) {
                if (comp) blo = (uint32_t)__hip_atomic_load(&S.best[id], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (STATS && comp && blo != 0u) tc.hits++;
                const uint32_t nph = comp ? route_complete<QUERY>(t_flags, blo != 0u) : (uint32_t)PH3_NONE;
                PUSH_ONE(PH3_LA, nph == PH3_LA) PUSH_ONE(PH3_LB, nph == PH3_LB) PUSH_ONE(PH3_LC, nph == PH3_LC)
            }
            }
        };
#undef PUSH_ONE
        const bool plain = MODE == 1 || n_exact == 0;
        if constexpr (DEC) {
            if (act == PH3_INNER) {
                if (plain)
                    inner_arm_dec(std::false_type{});
                else
                    inner_arm_dec(std::true_type{});
            } else {
                leaf_arm_dec();
            }
        } else {
        if (act == PH3_INNER) {
            if (plain)
                inner_arm(std::false_type{});
            else
                inner_arm(std::true_type{});
        } else {
            if (plain)
                leaf_arm(std::false_type{});
            else
                leaf_arm(std::true_type{});
        }
        }
      }
        // (commit ring: a pool with nothing to do but slots that are held back looks at those)
        if (act == PH3_NONE && !(commit_ring && (__builtin_amdgcn_readfirstlane((int)S.waitq) & 0xff) != 0)) break;
        if (act == PH3_LA) {
            POP3(PH3_LA)
            bool new_exact = false;
            if (on) {
                LOGIC_PARAMS()
                NewRay nr;
                float4 ra_, rb_;
                ray_result(S, id, ra_, rb_);
                nph = logic_A<MODE, RING>(Pl, tl, g, ra_, rb_, nr, cnt, ALL);
                if (nph == PH3_NONE) nph = start_ray<MODE, false, LDS3>(Pl.sc, S, id, nr, cnt, M3.force_exact != 0, new_exact);
            }
            n_exact += (int)__popcll(__ballot(new_exact));
            PUSH3()
        } else if (act == PH3_LB) {
            POP3(PH3_LB)
            bool new_exact = false;
            if (on) {
                LOGIC_PARAMS()
                NewRay nr;
                float4 ra_, rb_;
                ray_result(S, id, ra_, rb_);
                nph = logic_B<MODE, RING>(Pl, g, ra_, rb_, nr);
                if (nph == PH3_NONE) nph = start_ray<MODE, false, LDS3>(Pl.sc, S, id, nr, cnt, M3.force_exact != 0, new_exact);
            }
            n_exact += (int)__popcll(__ballot(new_exact));
            PUSH3()
        } else {
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
            const int src_n = held ? wn : qn[PH3_LC], src_h = held ? wh : qh[PH3_LC];
            const int take = min(64, src_n);
            if (STATS) { dg_b[PH3_LC]++; dg_l[PH3_LC] += (uint32_t)take; }
            const bool on = lane < take;
            const uint32_t id = (held ? S.rq(PH3_WAIT) : S.rq(PH3_LC))[ring_wrap<QCAP>((uint32_t)(src_h + lane))];
            {
                int nh = src_h + take;
                if (nh >= QCAP) nh -= QCAP;
                if (held) { wh = nh; wn -= take; } else { qh[PH3_LC] = nh; qn[PH3_LC] -= take; }
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
                const int got = QUERY ? (query_C(Pl, g, ra_, rb_, nr) ? LC_RAY : LC_DEAD) : logic_C<RING>(Pl, tl, g, cnt, nr, fin_key);
                if (got == LC_RAY) nph = start_ray<MODE, QUERY, LDS3>(Pl.sc, S, id, nr, cnt, M3.force_exact != 0, new_exact);
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
                    if (wait) S.rq(PH3_WAIT)[ring_wrap<QCAP>(slot)] = (uint8_t)id;
                    const int add = (int)__popcll(mw);
                    wn += add;
                    wt += add;
                    if (wt >= QCAP) wt -= QCAP;
                }
                if (held && mw == __ballot(on)) __builtin_amdgcn_s_sleep(64); // (none of them may start yet: no hurry)
                if (lane == 0) S.waitq = (uint32_t)wn | ((uint32_t)wh << 8) | ((uint32_t)wt << 16) | (held ? 0u : 1u << 24);
            }
        }
    }
#undef PUSH3
#undef PUSH_TRAV
#undef POP3
#undef LOGIC_PARAMS

    // ---- counters ----
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
    }
}

// ----------------------------------------------------------- accumulate ----
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

__device__ __forceinline__ FastDiv make_fastdiv_dev(uint32_t d)
{
    // k_accumulate runs once per pixel: derive the magic on the fly (same formula as make_fastdiv)
    uint32_t l = d > 1 ? 32u - (uint32_t)__clz((int)(d - 1)) : 0u;
    FastDiv f;
    f.m = (uint32_t)((((1ull << l) - d) << 32) / d + 1);
    f.sh = (l < 1 ? l : 1u) | ((l > 0 ? l - 1 : 0u) << 8);
    return f;
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
        if (on_j) out[base_on + (uint32_t)__popcll(m_on & below)] = i;
        else if (ex_j) out[wn - 1u - (base_off + (uint32_t)__popcll(m_off & below))] = i;
        base_on += (unsigned int)__popcll(m_on);
        base_off += (unsigned int)__popcll(m_off);
    }
}

__global__ __launch_bounds__(256) void k_accumulate(const AParams A)
{
    uint32_t slot = blockIdx.x * 256u + threadIdx.x;
    if (slot >= A.nslots) return;
    uint32_t i = 0, j = 0;
    bool valid = slot_to_pixel(slot, A.rank, A.world, A.n_tiles, A.tiles_x, make_fastdiv_dev(A.tiles_x), A.width, A.height, i, j);
    F3 c = f3(0.0f, 0.0f, 0.0f);
    if (valid) {
        if (!A.first_chunk) c = f3(A.accum[slot], A.accum[A.nslots + slot], A.accum[2ull * A.nslots + slot]);
        const float fspp = (float)A.spp;
        for (uint32_t s = 0; s < A.chunk_samples; s++) { // temp_color += L / spp, in sample order (Render.cuh:348)
            float4 l = A.L[(uint64_t)s * A.nslots + slot];
            c.x = c.x + l.x / fspp;
            c.y = c.y + l.y / fspp;
            c.z = c.z + l.z / fspp;
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

// crt_preview: the frame a progressive render would show now.  The accumulator holds sum_{k < done} L_k / spp (Render.cuh:348
// with the samples so far); its estimate of the mean is that sum * spp / done.  Reads the accumulator only.
__global__ __launch_bounds__(256) void k_preview(const AParams A, const float scale)
{
    uint32_t slot = blockIdx.x * 256u + threadIdx.x;
    if (slot >= A.nslots) return;
    uint32_t i = 0, j = 0;
    const bool valid = slot_to_pixel(slot, A.rank, A.world, A.n_tiles, A.tiles_x, make_fastdiv_dev(A.tiles_x), A.width, A.height, i, j);
    if (!valid && !A.tiled_output) return;
    F3 c = f3(0.0f, 0.0f, 0.0f);
    if (valid) c = f3(A.accum[slot] * scale, A.accum[A.nslots + slot] * scale, A.accum[2ull * A.nslots + slot] * scale);
    const uint64_t o = A.tiled_output ? (uint64_t)slot : (uint64_t)j * A.width + i;
    A.out_rgb[o * 3 + 0] = valid ? tonemap(c.x) : 0;
    A.out_rgb[o * 3 + 1] = valid ? tonemap(c.y) : 0;
    A.out_rgb[o * 3 + 2] = valid ? tonemap(c.z) : 0;
    if (A.out_mean) { A.out_mean[o * 3 + 0] = c.x; A.out_mean[o * 3 + 1] = c.y; A.out_mean[o * 3 + 2] = c.z; }
}

// ------------------------------------------------------------ test kernels --
// crt_intersect: loads n host rays into the first n pool slots (direction normalised as Ray's
// constructor does, Ray.cuh:12-13) so that the production trace kernel answers them.
// limits != nullptr: the rays are visibility rays (blocked(), Render.cuh:19-27) with these t_to_light values.
__global__ __launch_bounds__(256) void k_fill_rays(Pool pl, uint32_t n, const float* o, const float* d, const bool raw_dir, const float* limits)
{
    uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    F3 dir = f3(d[3 * i], d[3 * i + 1], d[3 * i + 2]);
    if (!raw_dir) dir = unit3(dir);
    pl.ro[i] = make_float4(o[3 * i], o[3 * i + 1], o[3 * i + 2], limits ? limits[i] : 0.0f);
    pl.rd[i] = make_float4(dir.x, dir.y, dir.z, __uint_as_float((uint32_t)(limits ? RAY_SHADOW : RAY_CLOSEST)));
    pl.res[i] = make_float2(FLT_MAX, __int_as_float(-1));
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
    case 11: r = quot3_exact(f3(x, x, x), y, false).y; break;              // the short exact division against x / y (tests)
    case 12: r = quot3_exact(f3(x, 0.0f, -0.0f), y, true).x; break;        // ... in the form unit3 uses
    default: r = qnan();
    }
    out[i] = r;
}
// crt_device_rcp_check: every fp32 bit pattern through rcp_short and through the division
__global__ void k_rcp_check(unsigned long long* counts)
{
    const unsigned long long tid = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    unsigned int bad_in = 0, bad_out = 0;
    for (unsigned long long b = tid; b < (1ull << 32); b += stride) {
        const float x = __uint_as_float((uint32_t)b);
        const float ref = 1.0f / x, got = rcp_short(x);
        const bool same = __float_as_uint(ref) == __float_as_uint(got) || (ref != ref && got != got);
        if (!same) { if (rcp_short_ok(x)) bad_in++; else bad_out++; }
    }
    bad_in = wave_sum(bad_in); bad_out = wave_sum(bad_out);
    if ((threadIdx.x & 63) == 0 && (bad_in | bad_out)) { atomicAdd(&counts[0], (unsigned long long)bad_in); atomicAdd(&counts[1], (unsigned long long)bad_out); }
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
    void ensure(size_t count)
    {
        if (n < count) alloc(count);
    }
    // Uncached device memory: every access goes to memory, past the L2 caches of the XCDs, which are not coherent with one another
    // inside a launch (the commit ring's buffers: written by one wave, read by another during the same launch).
    void ensure_uncached(size_t count)
    {
        if (n >= count && uncached) return;
        release();
        if (count == 0) count = 1;
        HIP_CHECK(hipExtMallocWithFlags((void**)&p, count * sizeof(T), hipDeviceMallocUncached));
        n = count;
        uncached = true;
    }
    bool uncached = false;
    void upload(const std::vector<T>& v)
    {
        alloc(v.size());
        if (!v.empty()) HIP_CHECK(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    }
    void release()
    {
        if (p) { (void)hipFree(p); p = nullptr; n = 0; uncached = false; }
    }
    ~DevBuf() { release(); }
};

const int kMaxBatch = 64;

} // namespace

struct crt_scene {
    int device = 0;
    DevBuf<float4> nodes, tri_geo, mats, ltri, nodes3, leaf_geo, tri_nm, nodes4;
    int depth4 = 1; // depth of the 4-wide tree
    bool ref16_ok = false; // refs of the 4-wide tree and of the leaf records fit 16 bits (k_mega3's 16-bit stack layout)
    bool ref16_inner_ok = false; // refs of the 4-wide tree alone fit 16 bits (decoupled leaves: the stack holds inner nodes only)
    bool dec_ok = false;         // leaf records fit the 24 bits of a leaf-queue entry
    uint32_t max_leaf = 0; // triangles in the largest leaf
    DevBuf<int32_t> tri_mat, leaf_count;
    DevBuf<uint4> lights;
    // path pool + per-item radiance + cross-chunk accumulator
    DevBuf<float4> p_ro, p_rd, p_vx, p_la, p_cc, p_vn, p_rec_a, p_rec_b, L;
    DevBuf<uint4> p_id;
    uint32_t n_mats = 0;
    DevBuf<float2> p_res;
    DevBuf<float> accum;
    DevBuf<unsigned long long> counters;      // [CNT_SHARDS][CNT_STRIDE]
    DevBuf<unsigned int> item_next;           // [ITEM_SHARDS][ITEM_STRIDE]
    DevBuf<uint32_t> item_list;               // k_order_items: the order of the work items of a launch (small launches only)
    DevBuf<unsigned int> ring_done, ring_state; // commit ring: finished items per (shard, sample), shard words
    DevBuf<float4> ring_L;                      // commit ring: radiance of [ring samples][shards * slots per shard] (uncached memory)
    std::vector<unsigned int> ring_state_host;
    uint64_t last_radiance_bytes = 0;           // per-work-item (or ring) radiance storage the last render used
    uint32_t last_ring_samples = 0;             // its ring size in samples (0: one radiance per work item)
    DevBuf<unsigned int> order_cnt;           // [ITEM_SHARDS][2] counters, one 128 B line each
    DevBuf<unsigned int> slot_next[2];        // [SLOT_SHARDS][SLOT_STRIDE], one per pool half
    DevBuf<int2> spill[2];                    // traversal stack overflow, one per pool half
    hipStream_t aux_stream = nullptr;         // second pool half runs here so that k_logic overlaps k_trace
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipEvent_t ev_k0 = nullptr, ev_k1 = nullptr; // around the k_mega3 launches of the last frame, recorded without synchronizing (crt_last_launch_ms)
    uint32_t last_launches = 0;
    int n_cus = 0;
    unsigned long long* h_counters = nullptr; // pinned copy of counters
    DevScene dev{};
    int stack_cap = 0;
    uint32_t n_tris = 0;
    crt_accel_info accel{};
    // progressive render in flight: what the accumulator holds (crt_preview)
    struct { uint32_t samples = 0, spp = 0, width = 0, height = 0, rank = 0, world = 1, tiled = 0; } acc;
    std::vector<hipEvent_t> ev;
    ~crt_scene()
    {
        for (hipEvent_t e : ev) (void)hipEventDestroy(e);
        if (ev_k0) (void)hipEventDestroy(ev_k0);
        if (ev_k1) (void)hipEventDestroy(ev_k1);
        if (ev_fork) (void)hipEventDestroy(ev_fork);
        if (ev_join) (void)hipEventDestroy(ev_join);
        if (aux_stream) (void)hipStreamDestroy(aux_stream);
        if (h_counters) (void)hipHostFree(h_counters);
    }
};

// 16-bit stack entries: the scene allows it (crt_scene::ref16_ok) and CRT_REF16=0 does not forbid it
static bool use_ref16(const crt_scene* sc, int mode, bool dec = false)
{
    if (mode == 1 || !(dec ? sc->ref16_inner_ok : sc->ref16_ok)) return false;
    // CRT_REF16=0: the coupled form in its 32-bit layout whatever the scene; CRT_REF32=1: 32-bit stack entries in either form (tests, A/B)
    const char* e = std::getenv("CRT_REF16");
    const char* f = std::getenv("CRT_REF32");
    if (f && f[0] == '1') return false;
    return dec || !(e && e[0] == '0');
}

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

// Builds the device node array (layout: crt_device.h) holding TWO trees over the same leaves:
//   [0, A)      the SAH tree of crt_accel.h, used by CRT_TRAVERSAL_FAST for rays with finite inv_dir
//   [A, A + R)  the reference's own topology (post-order BVH re-laid breadth-first), used by
//               CRT_TRAVERSAL_REFERENCE and by FAST rays whose inv_dir is not finite.
// Returns the larger tree depth (root = 1).
struct AccelInfo {
    uint32_t n_leaves = 0, n_nodes2 = 0, on_device = 0, index_splits = 0;
    float sah_ms = 0.0f, sah_device_ms = 0.0f;
};
int convert_bvh(const crt_scene_desc& d, std::vector<float4>& nodes, std::vector<int32_t>& leaf_count, int32_t& root_fast, int32_t& root_exact, AccelInfo* ai = nullptr)
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
    if (is_leaf(d.root)) { root_fast = root_exact = leaf_ref(d.root); return 1; }

    // ---- SAH tree over the reference leaves ----
    std::vector<crtaccel::Prim> prims;
    for (uint32_t i = 0; i < d.n_nodes; i++) {
        if (!is_leaf((int32_t)i)) continue;
        crtaccel::Prim p;
        for (int a = 0; a < 3; a++) { p.box.lo[a] = d.nodes[i].aa[a]; p.box.hi[a] = d.nodes[i].bb[a]; }
        p.ref = leaf_ref((int32_t)i);
        prims.push_back(p);
    }
    std::vector<crtaccel::Node> acc;
    int32_t acc_root = 0;
    // the SAH tree over the reference leaves: on the device (crt_accel_build.hip; CRT_SAH_HOST=1 forces the host builder, which is
    // also the fallback); the 4-wide collapse below stays on the host (linear, a few hundred microseconds)
    const auto sah_t0 = std::chrono::steady_clock::now();
    int depth_fast = -1;
    float dev_ms = 0.0f;
    const bool want_device = !(std::getenv("CRT_SAH_HOST") && std::getenv("CRT_SAH_HOST")[0] == '1');
    uint32_t index_splits = 0;
    if (want_device) depth_fast = crtaccel::build_sah_device(prims, acc, acc_root, &dev_ms, &index_splits);
    const bool on_device = depth_fast >= 0;
    if (!on_device) depth_fast = crtaccel::build_sah(prims, acc, acc_root, &index_splits);
    if (const char* opt_ = std::getenv("CRT_SAH_OPT")) // experiment hook: insertion-based optimisation passes over the built tree
        if (!acc.empty() && std::atoi(opt_) > 0) depth_fast = crtaccel::optimize_sah(acc, std::atoi(opt_));
    if (ai) {
        ai->n_leaves = (uint32_t)prims.size(); ai->n_nodes2 = (uint32_t)acc.size(); ai->on_device = on_device ? 1u : 0u;
        ai->sah_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - sah_t0).count();
        ai->sah_device_ms = dev_ms;
        ai->index_splits = index_splits;
    }
    const int32_t A = (int32_t)acc.size();
    nodes.resize((size_t)A * 4);
    for (int32_t q = 0; q < A; q++) {
        const crtaccel::Node& n = acc[q];
        nodes[q * 4 + 0] = make_float4(n.box[0].lo[0], n.box[0].lo[1], n.box[0].lo[2], as_float(n.child[0]));
        nodes[q * 4 + 1] = make_float4(n.box[0].hi[0], n.box[0].hi[1], n.box[0].hi[2], as_float(n.child[1]));
        nodes[q * 4 + 2] = make_float4(n.box[1].lo[0], n.box[1].lo[1], n.box[1].lo[2], 0.0f);
        nodes[q * 4 + 3] = make_float4(n.box[1].hi[0], n.box[1].hi[1], n.box[1].hi[2], 0.0f);
    }
    root_fast = acc_root; // 0 (two or more leaves here)

    // ---- the reference topology, breadth-first numbering of inner nodes ----
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
    nodes.resize(((size_t)A + order.size()) * 4);
    for (size_t q = 0; q < order.size(); q++) {
        const crt_bvh_node& n = d.nodes[order[q]];
        const crt_bvh_node& l = d.nodes[n.lc];
        const crt_bvh_node& r = d.nodes[n.rc];
        int32_t lref = is_leaf(n.lc) ? leaf_ref(n.lc) : A + index[n.lc];
        int32_t rref = is_leaf(n.rc) ? leaf_ref(n.rc) : A + index[n.rc];
        const size_t o = ((size_t)A + q) * 4;
        nodes[o + 0] = make_float4(l.aa[0], l.aa[1], l.aa[2], as_float(lref));
        nodes[o + 1] = make_float4(l.bb[0], l.bb[1], l.bb[2], as_float(rref));
        nodes[o + 2] = make_float4(r.aa[0], r.aa[1], r.aa[2], 0.0f);
        nodes[o + 3] = make_float4(r.bb[0], r.bb[1], r.bb[2], 0.0f);
    }
    root_exact = A;
    return std::max(max_depth, depth_fast);
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
    {
        // every node must hang under the root exactly once (the reference builder emits a tree in post-order, BVH.h:37-84), and the
        // leaves must own disjoint triangle ranges: the FAST traversal builds its own tree over ALL leaves of the description, so a
        // leaf the reference topology cannot reach, or two leaves sharing triangles, would make the two modes disagree
        std::vector<uint8_t> seen(d->n_nodes, 0), owned(d->n_tris, 0);
        std::vector<int32_t> todo(1, d->root);
        uint32_t visited = 0;
        while (!todo.empty()) {
            const int32_t i = todo.back();
            todo.pop_back();
            if (seen[i]) return fail(CRT_ERR_INVALID_ARG, "crt_scene_create: a node is reachable from the root more than once");
            seen[i] = 1;
            visited++;
            const crt_bvh_node& n = d->nodes[i];
            if (n.lc < 0 && n.rc < 0) {
                for (uint32_t k = 0; k < n.n; k++) {
                    if (owned[(uint32_t)n.it + k]) return fail(CRT_ERR_INVALID_ARG, "crt_scene_create: two leaves share a triangle");
                    owned[(uint32_t)n.it + k] = 1;
                }
            } else { todo.push_back(n.lc); todo.push_back(n.rc); }
        }
        if (visited != d->n_nodes) return fail(CRT_ERR_INVALID_ARG, "crt_scene_create: nodes that the root does not reach");
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
struct RingPlan { uint32_t samples, spsh, shards; }; // commit ring of a launch: samples held (0 = one radiance per work item), pixel slots per cursor shard
Shard make_shard(uint32_t w, uint32_t h, uint32_t world)
{
    Shard s;
    s.tiles_x = (w + CRT_TILE - 1) / CRT_TILE;
    s.tiles_y = (h + CRT_TILE - 1) / CRT_TILE;
    s.n_tiles = s.tiles_x * s.tiles_y;
    s.local_tiles = (s.n_tiles + world - 1) / world; // padded so every rank writes the same number of slots
    s.nslots = s.local_tiles * 64u;
    return s;
}

// 1 / n for n a power of two (exactly representable), else 0
float inv_if_pow2(int32_t n) { return (n > 0 && (n & (n - 1)) == 0) ? 1.0f / (float)n : 0.0f; }

uint32_t env_u32(const char* name, uint32_t dflt)
{
    const char* v = std::getenv(name);
    if (!v || !*v) return dflt;
    long x = std::strtol(v, nullptr, 10);
    return x > 0 ? (uint32_t)x : dflt;
}

// The instantiation of k_mega3 for a traversal mode (0 FAST, 1 REFERENCE, 2 EXACT), with or without counters, every sample traced
// or not (FAST only), render or query form, 32- or 16-bit stack entries (never for REFERENCE)
typedef void (*Mega3Kernel)(const MParams3);
// The decoupled-leaves form (Pool4LdsT) of a launch: CRT_TRAVERSAL_EXACT on a scene whose leaf records fit a queue entry.  It is
// the layout of the scenes whose leaf records no longer fit 16-bit stack entries while their four-wide nodes do (about 50 000 to
// 160 000 triangles): its stack holds inner nodes only.  On the smaller scenes the coupled form is 1 - 3 % faster and
// stays the default.  CRT_DEC=1 / 0 forces / forbids it (tests, A/B); CRT_REF16=0 ("the leaf records do not fit") selects it too.
static bool use_dec(const crt_scene* sc, int mode)
{
    if (mode != 2 || !sc->dec_ok) return false;
    const char* e = std::getenv("CRT_DEC");
    if (e && e[0] == '0') return false;
    if (e && e[0] == '1') return true;
    const char* r = std::getenv("CRT_REF16");
    const bool fits16 = sc->ref16_ok && !(r && r[0] == '0');
    return !fits16 && sc->ref16_inner_ok;
}
template <bool R16, bool DEC> Mega3Kernel mega3_exact_kernel(bool stats, bool all, bool query, bool ring)
{
    if (ring) return all ? (Mega3Kernel)k_mega3<2, false, true, false, R16, true, DEC> : (Mega3Kernel)k_mega3<2, false, false, false, R16, true, DEC>;
    if (query) return (Mega3Kernel)k_mega3<2, false, false, true, R16, false, DEC>;
    if (all) return stats ? (Mega3Kernel)k_mega3<2, true, true, false, R16, false, DEC> : (Mega3Kernel)k_mega3<2, false, true, false, R16, false, DEC>;
    return stats ? (Mega3Kernel)k_mega3<2, true, false, false, R16, false, DEC> : (Mega3Kernel)k_mega3<2, false, false, false, R16, false, DEC>;
}
Mega3Kernel mega3_kernel(int mode, bool stats, bool all, bool query, bool r16, bool ring = false, bool dec = false)
{
    if (mode == 2) {
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
// rays per wave / stack levels in LDS of a launch's kernel
static uint32_t mega3_pool_p(bool dec, bool ring) { return dec ? (ring ? (uint32_t)Pool4LdsT<true, true>::P : (uint32_t)Pool4LdsT<true, false>::P) : (uint32_t)POOL3_P; }
static int mega3_lds_levels(bool dec, bool r16) { return dec ? (r16 ? Pool4LdsT<true, false>::LV : Pool4LdsT<false, false>::LV) : POOL_LV; }
// Diagnostic hook (tools/bbprof): CRT_BBPROF_CO names a code object holding the default instantiation of k_mega3 with a counting
// prologue in every basic block (tools/bbprof/instrument.py applied to the compiler's assembly of THIS file); the launch then goes
// to that copy, the address of its counter buffer travels in MParams3::dbg_loads / dbg_valu, and the summed counters
// (one u64 per block: executions << 32 | active lanes) are written to CRT_BBPROF_OUT after every launch.  Returns false when the
// variable is not set or the kernel is another instantiation: the caller launches as usual.
bool bbprof_launch(Mega3Kernel kern, MParams3 M3, uint32_t blocks, hipStream_t st)
{
    static const char* co = std::getenv("CRT_BBPROF_CO");
    // (the default instantiations: 16-bit stack entries, CRT_TRAVERSAL_EXACT, with the leaves decoupled or not)
    const char* sym = kern == (Mega3Kernel)k_mega3<2, false, false, false, true, false, true>    ? "_ZN12_GLOBAL__N_17k_mega3ILi2ELb0ELb0ELb0ELb1ELb0ELb1EEEvNS_8MParams3E"
                      : kern == (Mega3Kernel)k_mega3<2, false, false, false, true, false, false> ? "_ZN12_GLOBAL__N_17k_mega3ILi2ELb0ELb0ELb0ELb1ELb0ELb0EEEvNS_8MParams3E"
                                                                                                  : nullptr;
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

// Which pipeline renders: 4 = k_mega3 (the product), 2 = the wavefront pipeline (k_logic + k_trace).  k_mega3 keeps the best
// triangle's offset inside its leaf in 16 bits, addresses nodes and leaf records with 32-bit byte offsets and the traversal stack
// depth in 8 bits; scenes beyond any of these fall back to the wavefront pipeline, which has no such limits.  The CRT_TEST_*
// variables lower the limits so that the tests can force each fallback on a small scene.
uint32_t choose_pipeline(const crt_scene* sc);

const uint64_t kMaxChunkItems = 1ull << 30; // paths per chunk (17 GB of per-path radiance: sized for 288 GB of HBM, every launch ends with a 2 ms tail)

template <int MODE, bool STATS> void launch_trace(const TParams& T, uint32_t blocks, size_t lds, hipStream_t st)
{
    hipLaunchKernelGGL((k_trace<MODE, STATS>), dim3(blocks), dim3(256), lds, st, T);
}
template <int MODE, bool STATS> int trace_blocks_per_cu(size_t lds)
{
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_trace<MODE, STATS>, 256, lds) != hipSuccess || nb < 1) nb = 1;
    return nb;
}

struct TraceSetup {
    TParams T;
    size_t lds;
    int mode_id;
    uint32_t blocks;
};
// Everything a k_trace launch over `pool` needs (grid sized to the device's residency: the kernel is persistent).
TraceSetup make_trace_setup(crt_scene* sc, const Pool& pool, uint32_t traversal, bool want_stats, int half = 0, int n_halves = 1)
{
    TraceSetup S;
    std::memset(&S.T, 0, sizeof(S.T));
    TParams& T = S.T;
    T.sc = sc->dev; T.pool = pool; T.counters = sc->counters.p;
    T.slot_next = sc->slot_next[half].p;
    T.refill_min = (int32_t)std::min<uint32_t>(64, env_u32("CRT_REFILL_MIN", REFILL_MIN));
    T.leaf_min = (int32_t)std::min<uint32_t>(64, env_u32("CRT_LEAF_MIN", LEAF_MIN));
    T.slots_per_shard = ((pool.n + SLOT_SHARDS - 1) / SLOT_SHARDS + 63u) & ~63u;
    // LDS holds the first levels of the traversal stack; the rest (rarely touched) spills to HBM/L2
    const int lds_cap = (int)std::min<uint32_t>((uint32_t)sc->stack_cap, std::max(2u, env_u32("CRT_STACK_LDS", 8)));
    T.stack_cap = lds_cap;
    S.lds = (size_t)lds_cap * 256 * sizeof(int2);
    S.mode_id = (traversal == CRT_TRAVERSAL_REFERENCE ? 2 : traversal == CRT_TRAVERSAL_EXACT ? 4 : 0) + (want_stats ? 1 : 0);
    int per_cu = S.mode_id == 5 ? trace_blocks_per_cu<2, true>(S.lds) : S.mode_id == 4 ? trace_blocks_per_cu<2, false>(S.lds)
               : S.mode_id == 3 ? trace_blocks_per_cu<1, true>(S.lds) : S.mode_id == 2 ? trace_blocks_per_cu<1, false>(S.lds)
               : S.mode_id == 1 ? trace_blocks_per_cu<0, true>(S.lds) : trace_blocks_per_cu<0, false>(S.lds);
    // with two pool halves in flight leave room for the other half's k_logic blocks
    const uint32_t dflt_per_cu = n_halves > 1 ? 3u : 64u; // measured best on MI355X (C2): 3 trace blocks + logic blocks per CU
    per_cu = (int)std::min<uint32_t>((uint32_t)per_cu, env_u32("CRT_TRACE_BLOCKS_PER_CU", dflt_per_cu));
    S.blocks = std::min<uint32_t>((pool.n + 255) / 256, (uint32_t)(sc->n_cus * per_cu));
    const int spill_levels = std::max(1, sc->stack_cap - lds_cap);
    T.spill_stride = S.blocks * 256u;
    sc->spill[half].ensure((size_t)spill_levels * T.spill_stride);
    T.spill = sc->spill[half].p;
    return S;
}
void launch_trace_pass(crt_scene* sc, const TraceSetup& S, hipStream_t st)
{
    HIP_CHECK(hipMemsetAsync(S.T.slot_next, 0, (size_t)SLOT_SHARDS * SLOT_STRIDE * sizeof(unsigned int), st));
    if (S.mode_id == 5) launch_trace<2, true>(S.T, S.blocks, S.lds, st);
    else if (S.mode_id == 4) launch_trace<2, false>(S.T, S.blocks, S.lds, st);
    else if (S.mode_id == 3) launch_trace<1, true>(S.T, S.blocks, S.lds, st);
    else if (S.mode_id == 2) launch_trace<1, false>(S.T, S.blocks, S.lds, st);
    else if (S.mode_id == 1) launch_trace<0, true>(S.T, S.blocks, S.lds, st);
    else launch_trace<0, false>(S.T, S.blocks, S.lds, st);
}

// Renders samples [s_begin, s_begin + s_count) of the prm->spp samples per pixel into the scene's accumulator
// (temp_color += L_k / spp in sample order, Render.cuh:348); the range that ends at spp also tone-maps and writes the frame.
int render_impl(crt_scene* sc, const crt_camera* cam, const crt_params* prm, void* d_rgb, void* d_mean, hipStream_t st, crt_stats* stats,
                uint32_t s_begin = 0, uint32_t s_count = 0xffffffffu)
{
    if (!sc || !cam || !prm) return fail(CRT_ERR_INVALID_ARG, "crt_render: null argument");
    if (s_count == 0xffffffffu) s_count = prm->spp > s_begin ? prm->spp - s_begin : 0;
    if (s_count == 0 || (uint64_t)s_begin + s_count > prm->spp) return fail(CRT_ERR_INVALID_ARG, "crt_render: sample range outside [0, spp)");
    const uint32_t s_end = s_begin + s_count;
    if (!d_rgb && s_end == prm->spp) return fail(CRT_ERR_INVALID_ARG, "crt_render: null frame buffer");
    if (prm->width == 0 || prm->height == 0 || prm->spp == 0) return fail(CRT_ERR_INVALID_ARG, "crt_render: width, height and spp must be positive");
    if (prm->world == 0 || prm->rank >= prm->world) return fail(CRT_ERR_INVALID_ARG, "crt_render: need rank < world");
    if (prm->light_sample_n < 0 || prm->light_sample_n > 4096) return fail(CRT_ERR_INVALID_ARG, "crt_render: light_sample_n must be in [0, 4096]");
    if ((uint64_t)prm->width * prm->height > 0xffffffffull) return fail(CRT_ERR_UNSUPPORTED, "crt_render: more than 2^32 pixels");
    if (prm->traversal != CRT_TRAVERSAL_FAST && prm->traversal != CRT_TRAVERSAL_REFERENCE && prm->traversal != CRT_TRAVERSAL_EXACT)
        return fail(CRT_ERR_INVALID_ARG, "crt_render: unknown traversal mode");
    const bool want_stats = (prm->flags & CRT_FLAG_STATS) != 0;
    const bool tiled = (prm->flags & CRT_FLAG_TILED_OUTPUT) != 0;
    if (prm->world > 1 && !tiled) return fail(CRT_ERR_INVALID_ARG, "crt_render: world > 1 needs CRT_FLAG_TILED_OUTPUT");
    if ((uint64_t)sc->dev.n_lights * (uint64_t)prm->light_sample_n > 0xffffu) return fail(CRT_ERR_UNSUPPORTED, "crt_render: more than 65535 next-event samples per vertex");
    try {
        HIP_CHECK(hipSetDevice(sc->device));
        Shard sh = make_shard(prm->width, prm->height, prm->world);
        const uint64_t max_items = std::min<uint64_t>(kMaxChunkItems, 1ull << std::min(30u, env_u32("CRT_CHUNK_LOG2", 30))); // (test hook: small chunks)
        uint32_t chunk = (uint32_t)std::min<uint64_t>(s_count, std::max<uint64_t>(1, max_items / sh.nslots));
        uint64_t cap = (uint64_t)chunk * sh.nslots;
        const uint32_t pipeline = choose_pipeline(sc);
        // ---- commit ring (megakernel only, CRT_FLAG_BOUNDED_RADIANCE): radiance storage for a window of samples, the sum
        // c += L_k / spp made inside the launch; the whole sample range is then ONE launch.  ring samples = 4 x the depth of the work in
        // flight (pool slots / pixel slots), at least 32: a shard is held back only when one of its paths takes four times as long as
        // the rest of the pool.
        RingPlan ring;
        std::memset(&ring, 0, sizeof(ring));
        if (pipeline == 4 && !want_stats) {
            // cursor shards: the commits of a shard are a serial chain (one wave, a memory round trip per 256 pixel slots), so a ring
            // launch has more and smaller shards than the 64 of a launch without: about 1 024 pixel slots each, at most 1 024 shards
            uint32_t shards = ITEM_SHARDS;
            while (shards < 1024u && sh.nslots / (shards * 2u) >= 1024u) shards *= 2u;
            const uint32_t spsh = ((sh.nslots + shards - 1) / shards + 63u) & ~63u;
            const uint64_t pool_slots = (uint64_t)sc->n_cus * 16u * (uint64_t)POOL3_P;
            uint32_t rs = 32;
            while (rs < 65536u && (uint64_t)rs * sh.nslots < 4ull * pool_slots) rs <<= 1;
            const uint32_t forced = env_u32("CRT_COMMIT_RING_LOG2", 0); // (test hook: a ring of 2^n samples, with or without the flag)
            if (forced) rs = 1u << std::min(16u, forced);
            const uint64_t per_shard = (uint64_t)spsh * s_count;
            const bool fits32 = per_shard * shards < 0xffffffffull;
            if ((forced || (prm->flags & CRT_FLAG_BOUNDED_RADIANCE)) && rs < s_count && fits32) {
                ring.samples = rs; ring.spsh = spsh; ring.shards = shards;
                chunk = s_count;
                cap = (uint64_t)rs * spsh * shards;
            }
        }
        const uint32_t pool_log2 = std::min(26u, std::max(8u, env_u32("CRT_POOL_LOG2", 22)));
        const uint32_t pool_n = (uint32_t)std::min<uint64_t>((cap + 255) / 256 * 256, 1ull << pool_log2);
        const int batch_max = (int)std::min<uint32_t>(kMaxBatch, env_u32("CRT_ROUND_BATCH", 16));
        int batch = batch_max;

        if (ring.samples) sc->ring_L.ensure_uncached(cap);
        else sc->L.ensure(cap);
        sc->last_radiance_bytes = cap * sizeof(float4);
        sc->last_ring_samples = ring.samples;
        sc->accum.ensure_uncached((size_t)sh.nslots * 3); // (always uncached: a progressive render may switch between launches with and without the ring)
        const bool timing = stats != nullptr;
        if (timing && sc->ev.size() < (size_t)(4 * kMaxBatch + 4)) {
            while (sc->ev.size() < (size_t)(4 * kMaxBatch + 4)) {
                hipEvent_t e;
                HIP_CHECK(hipEventCreate(&e));
                sc->ev.push_back(e);
            }
        }
        const size_t counters_bytes = (size_t)CNT_SHARDS * CNT_STRIDE * sizeof(unsigned long long);
        HIP_CHECK(hipMemsetAsync(sc->counters.p, 0, counters_bytes, st));
        auto counter_sum = [&](int c) {
            unsigned long long v = 0;
            for (int s = 0; s < CNT_SHARDS; s++) v += sc->h_counters[s * CNT_STRIDE + c];
            return v;
        };
        unsigned long long alive_seen = 0;

        if (pipeline == 4) {
            // ---------- fused persistent megakernel: one launch per chunk ----------
            const bool reference = prm->traversal == CRT_TRAVERSAL_REFERENCE;
            const bool exact = prm->traversal == CRT_TRAVERSAL_EXACT;
            const int mode_id = (reference ? 2 : exact ? 4 : 0) + (want_stats ? 1 : 0);
            const int mode3 = reference ? 1 : exact ? 2 : 0;
            const bool dec = use_dec(sc, mode3);
            const bool r16 = use_ref16(sc, mode3, dec);
            const Mega3Kernel kern3 = mega3_kernel(mode3, want_stats, mode3 != 1 && (prm->flags & CRT_FLAG_TRACE_ALL) != 0, false, r16, ring.samples != 0, dec);
            const uint32_t pool_p = mega3_pool_p(dec, ring.samples != 0);
            MParams M;
            std::memset(&M, 0, sizeof(M));
            int per_cu = 1;
            uint32_t blocks, lanes;
            {
                // one wave per workgroup, pool_p rays per wave
                auto q3 = [&](int* n) {
                    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(n, kern3, 64, 0);
                    if (e != hipSuccess || *n < 1) *n = 1;
                };
                q3(&per_cu);
                per_cu = (int)std::min<uint32_t>((uint32_t)per_cu, env_u32("CRT_MEGA_BLOCKS_PER_CU", 64));
                const uint64_t most_items = ring.samples ? (uint64_t)s_count * sh.nslots : cap;
                blocks = std::min<uint32_t>((uint32_t)std::min<uint64_t>((most_items + pool_p - 1) / pool_p, 0x7fffffffull), (uint32_t)(sc->n_cus * per_cu));
                lanes = blocks * pool_p; // pool slots
            }
            sc->p_vx.ensure(lanes); sc->p_la.ensure(lanes); sc->p_cc.ensure(lanes); sc->p_vn.ensure(lanes); sc->p_id.ensure(lanes);
            sc->p_rec_a.ensure((size_t)lanes * CRT_BOUNCE_STACK_SIZE);
            sc->p_rec_b.ensure((size_t)lanes * CRT_BOUNCE_STACK_SIZE);
            // (16-bit layout: a ray on the reference-arithmetic path keeps its whole stack in the global area)
            const int lds_cap = mega3_lds_levels(dec, r16);
            const int spill_levels = r16 ? std::max(1, sc->stack_cap) : std::max(1, sc->stack_cap - lds_cap);
            sc->spill[0].ensure((size_t)spill_levels * lanes);
            Pool pool;
            std::memset(&pool, 0, sizeof(pool));
            pool.vx = sc->p_vx.p; pool.la = sc->p_la.p; pool.cc = sc->p_cc.p; pool.vn = sc->p_vn.p; pool.id = sc->p_id.p;
            pool.rec_a = sc->p_rec_a.p; pool.rec_b = sc->p_rec_b.p; pool.n = lanes;
            LParams P;
            std::memset(&P, 0, sizeof(P));
            P.sc = sc->dev; P.pool = pool;
            std::memcpy(P.eye, cam->eye, sizeof(P.eye));
            std::memcpy(P.inv_view, cam->inv_view, sizeof(P.inv_view));
            P.scale = det_tanf(cam->fov_y / 2);                       // Render.cuh:338
            P.ar = (float)prm->width / (float)prm->height;            // Render.cuh:339
            P.width = prm->width; P.height = prm->height;
            P.p_rr = prm->p_rr; P.lsn = prm->light_sample_n; P.seed = prm->seed;
            P.rank = prm->rank; P.world = prm->world; P.tiles_x = sh.tiles_x; P.n_tiles = sh.n_tiles;
            P.nslots = sh.nslots;
            P.inv_lsn_pow2 = inv_if_pow2(prm->light_sample_n); P.lsn_div = make_fastdiv((uint32_t)std::max(1, prm->light_sample_n)); P.nslots_div = make_fastdiv(sh.nslots); P.tiles_x_div = make_fastdiv(sh.tiles_x);
            P.L = sc->L.p; P.counters = sc->counters.p; P.item_next = sc->item_next.p; P.n_mats = sc->n_mats;
            M.sc = sc->dev; M.counters = sc->counters.p; M.spill = sc->spill[0].p; M.spill_stride = lanes; M.stack_cap = lds_cap;
            AParams A;
            std::memset(&A, 0, sizeof(A));
            A.width = prm->width; A.height = prm->height; A.spp = prm->spp;
            A.rank = prm->rank; A.world = prm->world; A.tiles_x = sh.tiles_x; A.n_tiles = sh.n_tiles;
            A.nslots = sh.nslots; A.tiled_output = tiled ? 1 : 0;
            A.L = sc->L.p; A.accum = sc->accum.p;
            A.out_rgb = (uint8_t*)d_rgb; A.out_mean = (float*)d_mean;
            hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr, e3 = nullptr;
            if (timing) { e0 = sc->ev[0]; e1 = sc->ev[1]; e2 = sc->ev[2]; e3 = sc->ev[3]; HIP_CHECK(hipEventRecord(e0, st)); }
            double kernel_ms = 0.0;
            uint32_t launches = 0;
            for (uint32_t s0 = s_begin; s0 < s_end; s0 += chunk) {
                uint32_t ns = std::min(chunk, s_end - s0);
                P.sample_begin = s0;
                P.n_items = (uint32_t)((uint64_t)ns * sh.nslots);
                P.items_per_shard = ((P.n_items + ITEM_SHARDS - 1) / ITEM_SHARDS + 63u) & ~63u;
                if (ring.samples) { // cursor shard = ring.spsh pixel slots x ns samples
                    P.items_per_shard = ring.spsh * ns;
                    P.n_items = P.items_per_shard * ring.shards;
                    P.ring_mask = ring.samples - 1u; P.spsh = ring.spsh; P.spsh_div = make_fastdiv(ring.spsh); P.ring_shards = ring.shards;
                    P.ring_stride = ring.spsh * ring.shards; P.n_samples = ns; P.tail_first = P.items_per_shard; P.spp_f = (float)prm->spp;
                    sc->ring_done.ensure_uncached((size_t)ring.shards * ring.samples);
                    sc->ring_state.ensure_uncached((size_t)ring.shards * ITEM_STRIDE);
                    P.ring_done = sc->ring_done.p; P.ring_state = sc->ring_state.p; P.accum = sc->accum.p; P.L = sc->ring_L.p;
                    std::vector<unsigned int>& state = sc->ring_state_host; // (a member: the copy below may still read it after this scope)
                    state.assign((size_t)ring.shards * ITEM_STRIDE, 0u);
                    for (uint32_t slot = 0; slot < sh.nslots; slot++) { // word 1: the pixel slots of the shard that are pixels
                        const uint32_t tile = (slot >> 6) * prm->world + prm->rank, pix = slot & 63u;
                        if (tile >= sh.n_tiles) continue;
                        const uint32_t ty = tile / sh.tiles_x, tx = tile - ty * sh.tiles_x;
                        if (tx * CRT_TILE + (pix & 7u) < prm->width && ty * CRT_TILE + (pix >> 3) < prm->height) state[(size_t)(slot / ring.spsh) * ITEM_STRIDE + 1]++;
                    }
                    HIP_CHECK(hipMemcpyAsync(sc->ring_state.p, state.data(), state.size() * sizeof(unsigned int), hipMemcpyHostToDevice, st));
                    HIP_CHECK(hipMemsetAsync(sc->ring_done.p, 0, (size_t)ring.shards * ring.samples * sizeof(unsigned int), st));
                }
                // the paths that stop at their first vertex are handed out last (k_order_items): 1 % of a whole C2 frame on one GPU,
                // 8 % of a rank's share on eight.  CRT_ITEM_ORDER=0 switches it off.
                P.item_list = nullptr;
                {
                    const char* eo = std::getenv("CRT_ITEM_ORDER");
                    const bool order = !(eo && eo[0] == '0');
                    if (order && P.n_items > 0) {
                        // the window: the last 2^19 work items of every shard (measured on C2, wall time of a rank's share at 1 / 2 / 4 / 8 ranks: no
                        // order 107.3 / 54.9 / 28.4 / 15.9 ms; 2^17: 107.4 / 54.5 / 28.3 / 15.2; 2^19: 106.8 / 54.2 / 27.7 / 14.6; whole shards:
                        // 107.2 / 54.1 / 27.5 / 14.6 -- the pass itself costs 0.9 ms for the 245.8 M items of a whole frame)
                        P.order_window = std::min<uint32_t>(P.items_per_shard, env_u32("CRT_ORDER_WINDOW", 1u << 19));
                        if (ring.samples) { // the window may span half the ring: its items stand for the launch's last sample at the gate
                            P.order_window = std::min<uint32_t>(P.order_window, (ring.samples / 2u) * ring.spsh);
                            P.tail_first = P.items_per_shard - P.order_window;
                        }
                        P.items_per_shard_div = make_fastdiv(std::max(1u, P.items_per_shard));
                        const uint32_t n_sh = ring.samples ? ring.shards : (uint32_t)ITEM_SHARDS;
                        sc->item_list.ensure((size_t)n_sh * P.order_window);
                        sc->order_cnt.ensure((size_t)n_sh * 2 * 32);
                        HIP_CHECK(hipMemsetAsync(sc->order_cnt.p, 0, (size_t)n_sh * 2 * 32 * sizeof(unsigned int), st));
                        const uint32_t spans = (P.order_window + 1023u) / 1024u;
                        if (ring.samples) hipLaunchKernelGGL(k_order_items<true>, dim3(ring.shards * spans), dim3(64), 0, st, P, sc->item_list.p, sc->order_cnt.p);
                        else hipLaunchKernelGGL(k_order_items<false>, dim3(ITEM_SHARDS * spans), dim3(64), 0, st, P, sc->item_list.p, sc->order_cnt.p);
                        HIP_CHECK(hipGetLastError());
                        P.item_list = sc->item_list.p;
                    }
                }
                P.items_per_shard_div = make_fastdiv(std::max(1u, P.items_per_shard));
                M.P = P;
                HIP_CHECK(hipMemsetAsync(sc->item_next.p, 0, (size_t)(ring.samples ? ring.shards : (uint32_t)ITEM_SHARDS) * ITEM_STRIDE * sizeof(unsigned int), st));
                if (timing) HIP_CHECK(hipEventRecord(e1, st));
                if (s0 == s_begin) HIP_CHECK(hipEventRecord(sc->ev_k0, st));
                {
                    MParams3 M3;
                    M3.M = M;
                    M3.spill = (int*)sc->spill[0].p; // (one word per entry; the buffer is sized for the two-word entries of k_trace)
                    M3.force_exact = (prm->flags & CRT_FLAG_FORCE_EXACT) ? 1u : 0u;
                    M3.dbg_loads = 0; M3.dbg_valu = 0;
                    if (!bbprof_launch(kern3, M3, blocks, st)) hipLaunchKernelGGL(kern3, dim3(blocks), dim3(64), 0, st, M3);
                }
                HIP_CHECK(hipGetLastError());
                if (s0 + ns >= s_end) HIP_CHECK(hipEventRecord(sc->ev_k1, st));
                if (timing) {
                    HIP_CHECK(hipEventRecord(e2, st));
                    HIP_CHECK(hipStreamSynchronize(st));
                    float ms = 0.0f;
                    HIP_CHECK(hipEventElapsedTime(&ms, e1, e2));
                    kernel_ms += ms;
                }
                launches++;
                sc->last_launches = launches;
                A.chunk_samples = ns;
                A.first_chunk = s0 == 0; A.last_chunk = s0 + ns >= prm->spp;
                if (ring.samples) { A.chunk_samples = 0; A.first_chunk = 0; } // the sum is in the accumulator already: tone mapping only
                if (!ring.samples || A.last_chunk) {
                    hipLaunchKernelGGL(k_accumulate, dim3((sh.nslots + 255) / 256), dim3(256), 0, st, A);
                    HIP_CHECK(hipGetLastError());
                }
                sc->acc.samples = A.last_chunk ? 0u : s0 + ns; sc->acc.spp = prm->spp; sc->acc.width = prm->width; sc->acc.height = prm->height;
                sc->acc.rank = prm->rank; sc->acc.world = prm->world; sc->acc.tiled = tiled ? 1u : 0u;
            }
            if (stats) {
                HIP_CHECK(hipEventRecord(e3, st));
                HIP_CHECK(hipMemcpyAsync(sc->h_counters, sc->counters.p, counters_bytes, hipMemcpyDeviceToHost, st));
                HIP_CHECK(hipStreamSynchronize(st));
                std::memset(stats, 0, sizeof(*stats));
                stats->paths = counter_sum(C_PATHS); stats->rays = counter_sum(C_RAYS); stats->shadow_rays = counter_sum(C_SHADOW);
                stats->probe_rays = counter_sum(C_PROBE);
                stats->rays_untraced = counter_sum(C_UNTRACED);
                stats->inner_pops = counter_sum(C_INNER); stats->leaf_pops = counter_sum(C_LEAF); stats->tri_tests = counter_sum(C_TESTS);
                stats->hits = counter_sum(C_HITS);
                stats->stack_sum = counter_sum(C_SUMSP);
                for (int sh2 = 0; sh2 < CNT_SHARDS; sh2++) stats->stack_max = std::max<uint64_t>(stats->stack_max, sc->h_counters[sh2 * CNT_STRIDE + C_MAXSP]);
                float total = 0.0f;
                HIP_CHECK(hipEventElapsedTime(&total, e0, e3));
                stats->phase_cycles[0] = counter_sum(C_CYC_LOGIC); stats->phase_cycles[1] = counter_sum(C_CYC_LEAF);
                stats->phase_cycles[2] = counter_sum(C_CYC_INNER); stats->phase_cycles[3] = counter_sum(C_CYC_OTHER);
                for (int i = 0; i < 20; i++) stats->phase_cycles[4 + i] = counter_sum(C_DIAG + i);
                stats->kernel_ms = (float)kernel_ms;
                stats->logic_ms = 0.0f;
                stats->total_ms = total;
                stats->kernel_launches = launches;
            }
            return CRT_OK;
        }

        // The pool is split into halves that run on two streams: the HBM-bound k_logic of one half
        // overlaps the issue-bound k_trace of the other.
        const int n_halves = (pool_n >= 2 * 65536u && env_u32("CRT_STREAMS", 2) >= 2) ? 2 : 1;
        const uint32_t half_n = n_halves == 2 ? ((pool_n / 2 + 255) / 256 * 256) : pool_n;
        sc->p_ro.ensure((size_t)half_n * n_halves); sc->p_rd.ensure((size_t)half_n * n_halves); sc->p_vx.ensure((size_t)half_n * n_halves);
        sc->p_la.ensure((size_t)half_n * n_halves); sc->p_cc.ensure((size_t)half_n * n_halves); sc->p_res.ensure((size_t)half_n * n_halves);
        sc->p_vn.ensure((size_t)half_n * n_halves); sc->p_id.ensure((size_t)half_n * n_halves);
        sc->p_rec_a.ensure((size_t)half_n * n_halves * CRT_BOUNCE_STACK_SIZE);
        sc->p_rec_b.ensure((size_t)half_n * n_halves * CRT_BOUNCE_STACK_SIZE);
        Pool pools[2];
        for (int h = 0; h < n_halves; h++) {
            Pool& pool = pools[h];
            const size_t o = (size_t)h * half_n;
            pool.ro = sc->p_ro.p + o; pool.rd = sc->p_rd.p + o; pool.vx = sc->p_vx.p + o; pool.la = sc->p_la.p + o; pool.cc = sc->p_cc.p + o;
            pool.vn = sc->p_vn.p + o; pool.id = sc->p_id.p + o; pool.res = sc->p_res.p + o;
            pool.rec_a = sc->p_rec_a.p + o * CRT_BOUNCE_STACK_SIZE; pool.rec_b = sc->p_rec_b.p + o * CRT_BOUNCE_STACK_SIZE;
            pool.n = half_n;
        }
        hipStream_t streams[2] = {st, sc->aux_stream};

        LParams P;
        std::memset(&P, 0, sizeof(P));
        P.sc = sc->dev;
        std::memcpy(P.eye, cam->eye, sizeof(P.eye));
        std::memcpy(P.inv_view, cam->inv_view, sizeof(P.inv_view));
        P.scale = det_tanf(cam->fov_y / 2);                       // Render.cuh:338
        P.ar = (float)prm->width / (float)prm->height;            // Render.cuh:339
        P.width = prm->width; P.height = prm->height;
        P.p_rr = prm->p_rr; P.lsn = prm->light_sample_n; P.seed = prm->seed;
        P.rank = prm->rank; P.world = prm->world; P.tiles_x = sh.tiles_x; P.n_tiles = sh.n_tiles;
        P.nslots = sh.nslots;
        P.inv_lsn_pow2 = inv_if_pow2(prm->light_sample_n); P.lsn_div = make_fastdiv((uint32_t)std::max(1, prm->light_sample_n)); P.nslots_div = make_fastdiv(sh.nslots); P.tiles_x_div = make_fastdiv(sh.tiles_x);
        P.L = sc->L.p;
        P.counters = sc->counters.p;
        P.item_next = sc->item_next.p;
        P.n_mats = sc->n_mats;
        const bool lds_tables = sc->n_mats <= LOGIC_TABLE_MAX && (uint32_t)sc->dev.n_lights <= LOGIC_TABLE_MAX;

        TraceSetup TS[2];
        LParams PH[2];
        for (int h = 0; h < n_halves; h++) TS[h] = make_trace_setup(sc, pools[h], prm->traversal, want_stats, h, n_halves);

        AParams A;
        std::memset(&A, 0, sizeof(A));
        A.width = prm->width; A.height = prm->height; A.spp = prm->spp;
        A.rank = prm->rank; A.world = prm->world; A.tiles_x = sh.tiles_x; A.n_tiles = sh.n_tiles;
        A.nslots = sh.nslots; A.tiled_output = tiled ? 1 : 0;
        A.L = sc->L.p; A.accum = sc->accum.p;
        A.out_rgb = (uint8_t*)d_rgb; A.out_mean = (float*)d_mean;

        double trace_ms = 0.0, logic_ms = 0.0;
        uint32_t trace_launches = 0;
        hipEvent_t ev_begin = nullptr, ev_end = nullptr;
        if (timing) {
            ev_begin = sc->ev[4 * kMaxBatch + 2];
            ev_end = sc->ev[4 * kMaxBatch + 3];
            HIP_CHECK(hipEventRecord(ev_begin, st));
        }
        const dim3 pool_grid((half_n + 255) / 256);
        const int evs_per_half = 2 * kMaxBatch + 1;
        for (uint32_t s0 = s_begin; s0 < s_end; s0 += chunk) {
            uint32_t ns = std::min(chunk, s_end - s0);
            P.sample_begin = s0;
            P.n_items = (uint32_t)((uint64_t)ns * sh.nslots);
            P.items_per_shard = ((P.n_items + ITEM_SHARDS - 1) / ITEM_SHARDS + 63u) & ~63u;
            HIP_CHECK(hipMemsetAsync(sc->item_next.p, 0, (size_t)ITEM_SHARDS * ITEM_STRIDE * sizeof(unsigned int), st));
            for (int h = 0; h < n_halves; h++) {
                PH[h] = P;
                PH[h].pool = pools[h];
                hipLaunchKernelGGL(k_pool_init, pool_grid, dim3(256), 0, st, pools[h]);
            }
            HIP_CHECK(hipGetLastError());
            for (;;) {
                if (n_halves == 2) { // fork: the second half's chain follows what is queued on st so far
                    HIP_CHECK(hipEventRecord(sc->ev_fork, st));
                    HIP_CHECK(hipStreamWaitEvent(sc->aux_stream, sc->ev_fork, 0));
                }
                for (int h = 0; h < n_halves; h++)
                    if (timing) HIP_CHECK(hipEventRecord(sc->ev[h * evs_per_half], streams[h]));
                for (int b = 0; b < batch; b++) {
                    for (int h = 0; h < n_halves; h++) {
                        hipStream_t hs = streams[h];
                        hipEvent_t* ev = sc->ev.data() + h * evs_per_half;
                        if (lds_tables) hipLaunchKernelGGL(k_logic<true>, pool_grid, dim3(256), 0, hs, PH[h]);
                        else hipLaunchKernelGGL(k_logic<false>, pool_grid, dim3(256), 0, hs, PH[h]);
                        if (timing) HIP_CHECK(hipEventRecord(ev[2 * b + 1], hs));
                        launch_trace_pass(sc, TS[h], hs);
                        if (timing) HIP_CHECK(hipEventRecord(ev[2 * b + 2], hs));
                    }
                }
                HIP_CHECK(hipGetLastError());
                if (n_halves == 2) { // join
                    HIP_CHECK(hipEventRecord(sc->ev_join, sc->aux_stream));
                    HIP_CHECK(hipStreamWaitEvent(st, sc->ev_join, 0));
                }
                HIP_CHECK(hipMemcpyAsync(sc->h_counters, sc->counters.p, counters_bytes, hipMemcpyDeviceToHost, st));
                HIP_CHECK(hipStreamSynchronize(st));
                if (timing) {
                    double bl = 0.0, bt = 0.0;
                    for (int h = 0; h < n_halves; h++) {
                        hipEvent_t* ev = sc->ev.data() + h * evs_per_half;
                        for (int b = 0; b < batch; b++) {
                            float a = 0.0f, c = 0.0f;
                            HIP_CHECK(hipEventElapsedTime(&a, ev[2 * b], ev[2 * b + 1]));
                            HIP_CHECK(hipEventElapsedTime(&c, ev[2 * b + 1], ev[2 * b + 2]));
                            bl += a; bt += c;
                        }
                    }
                    logic_ms += bl; trace_ms += bt;
                    if (std::getenv("CRT_TRACE_LOG"))
                        fprintf(stderr, "[crt] rounds %u..%u: rays in batch %llu, logic %.3f ms, trace %.3f ms\n", trace_launches, trace_launches + batch - 1,
                                (unsigned long long)(counter_sum(C_ALIVE) - alive_seen), bl, bt);
                }
                trace_launches += (uint32_t)(batch * n_halves);
                unsigned long long alive_now = counter_sum(C_ALIVE);
                if (alive_now == alive_seen) break; // no slot emitted a ray during the whole batch: chunk done
                // once the pool runs dry (no more regeneration) check more often, so that few empty rounds are launched
                const unsigned long long per_round = (alive_now - alive_seen) / (unsigned long long)batch;
                batch = per_round * 8 < (unsigned long long)half_n * n_halves ? std::min(batch_max, 4) : batch_max;
                if (per_round * 512 < (unsigned long long)half_n * n_halves) batch = std::min(batch_max, 2);
                alive_seen = alive_now;
            }
            A.chunk_samples = ns;
            A.first_chunk = s0 == 0; A.last_chunk = s0 + ns >= prm->spp;
            hipLaunchKernelGGL(k_accumulate, dim3((sh.nslots + 255) / 256), dim3(256), 0, st, A);
            HIP_CHECK(hipGetLastError());
            sc->acc.samples = A.last_chunk ? 0u : s0 + ns; sc->acc.spp = prm->spp; sc->acc.width = prm->width; sc->acc.height = prm->height;
            sc->acc.rank = prm->rank; sc->acc.world = prm->world; sc->acc.tiled = tiled ? 1u : 0u;
        }
        if (stats) {
            HIP_CHECK(hipEventRecord(ev_end, st));
            HIP_CHECK(hipStreamSynchronize(st));
            std::memset(stats, 0, sizeof(*stats));
            stats->paths = counter_sum(C_PATHS); stats->rays = counter_sum(C_RAYS); stats->shadow_rays = counter_sum(C_SHADOW);
            stats->probe_rays = counter_sum(C_PROBE);
            stats->inner_pops = counter_sum(C_INNER); stats->leaf_pops = counter_sum(C_LEAF); stats->tri_tests = counter_sum(C_TESTS);
            stats->hits = counter_sum(C_HITS);
            stats->stack_sum = counter_sum(C_SUMSP);
            for (int sh2 = 0; sh2 < CNT_SHARDS; sh2++) stats->stack_max = std::max<uint64_t>(stats->stack_max, sc->h_counters[sh2 * CNT_STRIDE + C_MAXSP]);
            float total = 0.0f;
            HIP_CHECK(hipEventElapsedTime(&total, ev_begin, ev_end));
            stats->kernel_ms = (float)trace_ms;
            stats->logic_ms = (float)logic_ms;
            stats->total_ms = total;
            stats->kernel_launches = trace_launches;
        }
        return CRT_OK;
    } catch (const HipFail& f) {
        return fail_hip(f);
    }
}

uint32_t choose_pipeline(const crt_scene* sc)
{
    uint32_t pipeline = env_u32("CRT_PIPELINE", 4);
    if (pipeline != 2) pipeline = 4;
    const uint64_t max_leaf = env_u32("CRT_TEST_MAX_LEAF", CRT_MEGA3_MAX_LEAF);
    const uint64_t max_bytes = std::getenv("CRT_TEST_MAX_BYTES") ? (uint64_t)env_u32("CRT_TEST_MAX_BYTES", 0xffffffffu) : (1ull << 32);
    const uint32_t max_stack = env_u32("CRT_TEST_MAX_STACK", CRT_MEGA3_MAX_STACK);
    if (pipeline == 4 && sc->max_leaf > max_leaf) pipeline = 2;
    if (pipeline == 4 && (sc->nodes4.n * sizeof(float4) >= max_bytes || sc->nodes3.n * sizeof(float4) >= max_bytes || sc->leaf_geo.n * sizeof(float4) >= max_bytes))
        pipeline = 2; // (33 M nodes / 53 M records)
    if (pipeline == 4 && (uint32_t)sc->stack_cap > max_stack) pipeline = 2; // a deeper stack would spill into the flag bits of word D
    return pipeline;
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
    *slots = make_shard(width, height, world).nslots;
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
        int32_t root_fast = 0, root_exact = 0;
        AccelInfo ai;
        int depth = convert_bvh(*d, nodes, leaf_count, root_fast, root_exact, &ai);
        sc->accel.n_leaves = ai.n_leaves; sc->accel.n_nodes2 = ai.n_nodes2; sc->accel.sah_on_device = ai.on_device;
        sc->accel.sah_ms = ai.sah_ms; sc->accel.sah_device_ms = ai.sah_device_ms; sc->accel.index_splits = ai.index_splits;
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
        // ---- k_mega3 layouts (crt_device.h): triangle-pair records per leaf, child boxes as (left, right) pairs ----
        std::vector<float4> leaf_geo, nodes3(nodes.size());
        std::vector<int32_t> rec_of_it(d->n_tris, -1);
        uint32_t max_leaf = 0;
        for (uint32_t i = 0; i < d->n_nodes; i++) {
            const crt_bvh_node& nn = d->nodes[i];
            if (!(nn.lc < 0 && nn.rc < 0)) continue;
            max_leaf = std::max(max_leaf, nn.n);
            rec_of_it[nn.it] = (int32_t)(leaf_geo.size() / 5);
            for (uint32_t k = 0; k < nn.n; k += 2) {
                const uint32_t ia = (uint32_t)nn.it + k, ib = k + 1 < nn.n ? ia + 1 : ia; // odd tail: the second lane repeats a and is masked
                const float4 a0 = geo[ia * 3ull], a1 = geo[ia * 3ull + 1], a2 = geo[ia * 3ull + 2];
                const float4 b0 = geo[ib * 3ull], b1 = geo[ib * 3ull + 1], b2 = geo[ib * 3ull + 2];
                leaf_geo.push_back(make_float4(a0.x, b0.x, a0.y, b0.y));  // v1.x, v1.y
                leaf_geo.push_back(make_float4(a0.z, b0.z, a0.w, b0.w));  // v1.z, e1.x
                leaf_geo.push_back(make_float4(a1.x, b1.x, a1.y, b1.y));  // e1.y, e1.z
                leaf_geo.push_back(make_float4(a1.z, b1.z, a1.w, b1.w));  // e2.x, e2.y
                leaf_geo.push_back(make_float4(a2.x, b2.x, as_float((int32_t)ia), as_float((int32_t)(nn.n - k)))); // e2.z, index, remaining
            }
        }
        auto ref3 = [&](int32_t r) -> int32_t { // old child ref -> k_mega3 child ref
            if (r >= 0) return r;
            return ~rec_of_it[(uint32_t)~r >> 4];
        };
        for (size_t q = 0; q * 4 < nodes.size(); q++) {
            const float4 a = nodes[q * 4], b = nodes[q * 4 + 1], c = nodes[q * 4 + 2], e = nodes[q * 4 + 3];
            int32_t lr, rr;
            std::memcpy(&lr, &a.w, 4); std::memcpy(&rr, &b.w, 4);
            nodes3[q * 4 + 0] = make_float4(a.x, c.x, a.y, c.y);
            nodes3[q * 4 + 1] = make_float4(a.z, c.z, b.x, e.x);
            nodes3[q * 4 + 2] = make_float4(b.y, e.y, b.z, e.z);
            nodes3[q * 4 + 3] = make_float4(as_float(ref3(lr)), as_float(ref3(rr)), 0.0f, 0.0f);
        }
        // ---- 4-wide tree for the rays with finite operands: the SAH tree collapsed (crt_device.h, nodes4) ----
        std::vector<float4> nodes4;
        float coord_max = 0.0f; // largest |coordinate| of a box of the 4-wide tree (+inf if any is not finite): start_ray's overflow test
        int32_t root4 = ref3(root_fast);
        int depth4 = 1;
        if (root4 >= 0) {
            struct Child { float lo[3], hi[3]; int32_t ref; }; // ref: nodes3 index (>= 0) or leaf ref (< 0)
            auto children_of = [&](int32_t q, Child out[2]) {
                const float4 n0 = nodes3[q * 4ull], n1 = nodes3[q * 4ull + 1], n2 = nodes3[q * 4ull + 2], n3 = nodes3[q * 4ull + 3];
                out[0] = Child{{n0.x, n0.z, n1.x}, {n1.z, n2.x, n2.z}, 0};
                out[1] = Child{{n0.y, n0.w, n1.y}, {n1.w, n2.y, n2.w}, 0};
                std::memcpy(&out[0].ref, &n3.x, 4); std::memcpy(&out[1].ref, &n3.y, 4);
            };
            auto area = [](const Child& c) {
                const double dx = (double)c.hi[0] - c.lo[0], dy = (double)c.hi[1] - c.lo[1], dz = (double)c.hi[2] - c.lo[2];
                return dx * dy + dy * dz + dz * dx;
            };
            // ---- which binary nodes become 4-wide nodes: the collapse that minimises the summed area of the 4-wide nodes -- the SAH
            //      cost of the inner steps, every step costing the same whatever the number of children used (Ylitie et al. 2017,
            //      the leaves being fixed here): cost(q) = A(q) + min_k D(left, k) + D(right, 4 - k), D(c, j) = cheapest cover of
            //      subtree c by at most j children of the node above = min(D(c, j - 1), min_k D(c.left, k) + D(c.right, j - k)),
            //      D(c, 1) = cost(c), D(leaf, .) = 0.  Against the round-1 rule (CRT_COLLAPSE=greedy: open the child with the largest
            //      area until there are four): cornell-box 9 967 instead of 11 993 nodes, 4.62 instead of 5.02 inner steps per ray,
            //      C2 -3.7 %; veach-mis 842 / 974 nodes, 5.05 / 5.13 steps, same time.
            const char* collapse_env = std::getenv("CRT_COLLAPSE");
            const bool collapse_dp = !(collapse_env && std::strcmp(collapse_env, "greedy") == 0);
            struct DpNode { double D[3]; uint8_t kw, c2, c3; }; // D[j-1]; kw: left share of the node's own four; c2 / c3: choice for j = 2 / 3
            std::vector<DpNode> dpn;
            if (collapse_dp) {
                dpn.assign(nodes3.size() / 4, DpNode{{0, 0, 0}, 1, 0, 0});
                struct Fr { int32_t q; double area; int state; };
                std::vector<Fr> st;
                st.push_back(Fr{root4, 0.0, 0});
                auto Dof = [&](const Child& c, int j) { return c.ref < 0 ? 0.0 : dpn[(size_t)c.ref].D[j - 1]; };
                while (!st.empty()) {
                    Fr& f = st.back();
                    Child two[2];
                    children_of(f.q, two);
                    if (f.state == 0) {
                        f.state = 1;
                        const int32_t q = f.q; (void)q;
                        for (int i = 0; i < 2; i++)
                            if (two[i].ref >= 0) st.push_back(Fr{two[i].ref, area(two[i]), 0}); // (invalidates f: not used below)
                        continue;
                    }
                    DpNode& n = dpn[(size_t)f.q];
                    // the node as a 4-wide node: its own step + the cheapest forest of four under it
                    double best = 0.0; int bk = 1;
                    for (int k = 1; k <= 3; k++) {
                        const double v = Dof(two[0], k) + Dof(two[1], 4 - k);
                        if (k == 1 || v < best) { best = v; bk = k; }
                    }
                    n.kw = (uint8_t)bk;
                    n.D[0] = f.area + best;
                    const double open2 = Dof(two[0], 1) + Dof(two[1], 1);
                    n.c2 = open2 < n.D[0] ? 1 : 0;
                    n.D[1] = n.c2 ? open2 : n.D[0];
                    const double o12 = Dof(two[0], 1) + Dof(two[1], 2), o21 = Dof(two[0], 2) + Dof(two[1], 1);
                    n.c3 = 0; n.D[2] = n.D[1];
                    if (o12 < n.D[2]) { n.D[2] = o12; n.c3 = 1; }
                    if (o21 < n.D[2]) { n.D[2] = o21; n.c3 = 2; }
                    st.pop_back();
                }
            }
            // the (at most j) roots that cover the subtree of child c in the cheapest way
            std::vector<Child> cover;
            struct Ex { Child c; int j; };
            auto expand = [&](const Child& c0, int j0) {
                std::vector<Ex> ex;
                ex.push_back(Ex{c0, j0});
                while (!ex.empty()) {
                    Ex e = ex.back(); ex.pop_back();
                    if (e.c.ref < 0 || e.j == 1) { cover.push_back(e.c); continue; }
                    const DpNode& n = dpn[(size_t)e.c.ref];
                    const int choice = e.j == 2 ? (n.c2 ? 1 : 0) : n.c3;
                    if (choice == 0) { ex.push_back(Ex{e.c, e.j - 1}); continue; }
                    Child two[2];
                    children_of(e.c.ref, two);
                    ex.push_back(Ex{two[1], e.j - choice});
                    ex.push_back(Ex{two[0], choice});
                }
            };
            struct Todo { int32_t node2; int32_t slot; int depth; }; // slot: index of the BVH4 node to fill
            std::vector<Todo> todo;
            nodes4.resize(8);
            todo.push_back(Todo{root4, 0, 1});
            root4 = 0;
            const float qn_ = std::numeric_limits<float>::quiet_NaN();
            (void)qn_;
            for (size_t t = 0; t < todo.size(); t++) {
                const Todo cur = todo[t];
                depth4 = std::max(depth4, cur.depth + 1);
                std::vector<Child> ch(2);
                children_of(cur.node2, ch.data());
                if (collapse_dp) {
                    const Child l = ch[0], r = ch[1];
                    const int k = dpn[(size_t)cur.node2].kw;
                    cover.clear();
                    expand(l, k);
                    expand(r, 4 - k);
                    ch = cover;
                }
                while (!collapse_dp && ch.size() < 4) { // open the largest inner child
                    int best = -1;
                    double ba = -1.0;
                    for (size_t i = 0; i < ch.size(); i++)
                        if (ch[i].ref >= 0 && area(ch[i]) > ba) { ba = area(ch[i]); best = (int)i; }
                    if (best < 0) break;
                    Child two[2];
                    children_of(ch[best].ref, two);
                    ch[best] = two[0];
                    ch.push_back(two[1]);
                }
                int32_t refs[4];
                float lo[4][3], hi[4][3];
                for (int i = 0; i < 4; i++) {
                    if (i < (int)ch.size()) {
                        for (int a = 0; a < 3; a++) { lo[i][a] = ch[i].lo[a]; hi[i][a] = ch[i].hi[a]; }
                        if (ch[i].ref >= 0) {
                            refs[i] = (int32_t)(nodes4.size() / 8);
                            nodes4.resize(nodes4.size() + 8);
                            todo.push_back(Todo{ch[i].ref, refs[i], cur.depth + 1});
                        } else refs[i] = ch[i].ref;
                    } else {
                        // empty slot: the inverted box (t_enter = +inf, t_exit = -inf whatever the signs of the direction)
                        for (int a = 0; a < 3; a++) { lo[i][a] = std::numeric_limits<float>::infinity(); hi[i][a] = -std::numeric_limits<float>::infinity(); }
                        refs[i] = ~0x7ffffff0; // (never followed)
                    }
                }
                for (int i = 0; i < (int)ch.size() && i < 4; i++)
                    for (int a = 0; a < 3; a++) {
                        const float m = std::max(std::fabs(lo[i][a]), std::fabs(hi[i][a]));
                        coord_max = (m <= FLT_MAX && coord_max <= FLT_MAX) ? std::max(coord_max, m) : std::numeric_limits<float>::infinity();
                    }
                float4* o = &nodes4[(size_t)cur.slot * 8];
                for (int a = 0; a < 3; a++) { // plane-major: [2a] = lo of axis a of the four children, [2a + 1] = hi
                    o[2 * a + 0] = make_float4(lo[0][a], lo[1][a], lo[2][a], lo[3][a]);
                    o[2 * a + 1] = make_float4(hi[0][a], hi[1][a], hi[2][a], hi[3][a]);
                }
                o[6] = make_float4(as_float(refs[0]), as_float(refs[1]), as_float(refs[2]), as_float(refs[3]));
                // the refs as the decoupled-leaves step wants them: a leaf as 0x80000000 | record << 8, ready to take the ray id
                // (records beyond 2^23 - 1 do not fit: crt_scene::dec_ok)
                auto dref = [](int32_t r) -> int32_t { return r >= 0 ? r : (int32_t)(0x80000000u | (((uint32_t)~r & 0x7fffffu) << 8)); };
                o[7] = make_float4(as_float(dref(refs[0])), as_float(dref(refs[1])), as_float(dref(refs[2])), as_float(dref(refs[3])));
            }
        }
        if (nodes4.empty()) nodes4.resize(8);
        const size_t n_nodes4 = nodes4.size() / 8;
        {
            // the empty node behind the tree (DevScene::empty4_off): four inverted boxes, refs that are never followed
            const float pinf_ = std::numeric_limits<float>::infinity();
            for (int a = 0; a < 3; a++) { nodes4.push_back(make_float4(pinf_, pinf_, pinf_, pinf_)); nodes4.push_back(make_float4(-pinf_, -pinf_, -pinf_, -pinf_)); }
            const float er = as_float(~0x7ffffff0);
            nodes4.push_back(make_float4(er, er, er, er)); nodes4.push_back(make_float4(er, er, er, er));
            sc->dev.empty4_off = (uint32_t)(n_nodes4 * 128);
        }
        sc->nodes4.upload(nodes4);
        sc->dev.nodes4 = sc->nodes4.p;
        sc->dev.root4 = root4;
        sc->dev.coord_max = coord_max;
        sc->depth4 = depth4;
        sc->accel.n_nodes4 = (uint32_t)n_nodes4; sc->accel.depth2 = (uint32_t)depth; sc->accel.depth4 = (uint32_t)depth4;
        sc->ref16_ok = n_nodes4 <= 32768 && leaf_geo.size() / 5 <= 32768; // node refs 0 .. 32767, leaf refs ~0 .. ~32767
        sc->ref16_inner_ok = n_nodes4 <= 32768;
        sc->dec_ok = leaf_geo.size() / 5 <= (size_t)LEAF_REC_MAX + 1;
        std::vector<float4> tri_nm(d->n_tris);
        for (uint32_t i = 0; i < d->n_tris; i++) tri_nm[i] = make_float4(d->tris[i].normal[0], d->tris[i].normal[1], d->tris[i].normal[2], as_float(d->tris[i].material));
        sc->nodes3.upload(nodes3); sc->leaf_geo.upload(leaf_geo); sc->tri_nm.upload(tri_nm);
        sc->dev.tri_nm = sc->tri_nm.p;
        sc->max_leaf = max_leaf;
        sc->dev.nodes3 = sc->nodes3.p; sc->dev.leaf_geo = sc->leaf_geo.p;
        sc->dev.root3_fast = ref3(root_fast); sc->dev.root3_exact = ref3(root_exact);
        std::vector<uint4> lights(d->n_lights);
        for (uint32_t i = 0; i < d->n_lights; i++) {
            FastDiv fd = make_fastdiv(d->lights[i].count);
            lights[i] = make_uint4(d->lights[i].first_tri, d->lights[i].count, fd.m, fd.sh);
        }
        sc->nodes.upload(nodes); sc->tri_geo.upload(geo); sc->tri_mat.upload(tri_mat); sc->mats.upload(mats);
        sc->ltri.upload(ltri); sc->lights.upload(lights); sc->leaf_count.upload(leaf_count);
        sc->counters.alloc((size_t)CNT_SHARDS * CNT_STRIDE);
        sc->item_next.alloc((size_t)1024 * ITEM_STRIDE); // (a commit-ring launch has up to 1 024 cursor shards)
        sc->slot_next[0].alloc((size_t)SLOT_SHARDS * SLOT_STRIDE);
        sc->slot_next[1].alloc((size_t)SLOT_SHARDS * SLOT_STRIDE);
        HIP_CHECK(hipStreamCreateWithFlags(&sc->aux_stream, hipStreamNonBlocking));
        HIP_CHECK(hipEventCreateWithFlags(&sc->ev_fork, hipEventDisableTiming));
        HIP_CHECK(hipEventCreateWithFlags(&sc->ev_join, hipEventDisableTiming));
        HIP_CHECK(hipEventCreate(&sc->ev_k0));
        HIP_CHECK(hipEventCreate(&sc->ev_k1));
        {
            hipDeviceProp_t prop;
            HIP_CHECK(hipGetDeviceProperties(&prop, device));
            sc->n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        }
        HIP_CHECK(hipHostMalloc((void**)&sc->h_counters, (size_t)CNT_SHARDS * CNT_STRIDE * sizeof(unsigned long long), hipHostMallocDefault));
        sc->dev.nodes = sc->nodes.p; sc->dev.tri_geo = sc->tri_geo.p; sc->dev.tri_mat = sc->tri_mat.p; sc->dev.mats = sc->mats.p;
        sc->dev.ltri = sc->ltri.p; sc->dev.lights = sc->lights.p; sc->dev.leaf_count = sc->leaf_count.p;
        sc->dev.root_fast = root_fast; sc->dev.root_exact = root_exact; sc->dev.n_lights = (int32_t)d->n_lights;
        sc->n_tris = d->n_tris;
        sc->n_mats = d->n_materials;
        // Both traversal modes hold at most one pending sibling per tree level.
        sc->stack_cap = std::max(depth + 2, 3 * sc->depth4 + 2); // BVH2: one pending sibling per level; BVH4: up to three
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

int crt_scene_accel_info(crt_scene* sc, crt_accel_info* out)
{
    if (!sc || !out) return fail(CRT_ERR_INVALID_ARG, "crt_scene_accel_info: null argument");
    *out = sc->accel;
    return CRT_OK;
}

int crt_scene_destroy(crt_scene* sc)
{
    if (!sc) return CRT_OK;
    (void)hipSetDevice(sc->device);
    delete sc;
    return CRT_OK;
}

int crt_render_device(crt_scene* sc, const crt_camera* cam, const crt_params* prm, void* d_rgb, void* d_mean, void* stream, crt_stats* stats)
{
    return render_impl(sc, cam, prm, d_rgb, d_mean, (hipStream_t)stream, stats);
}

int crt_render(crt_scene* sc, const crt_camera* cam, const crt_params* prm, uint8_t* out_rgb, float* out_mean, crt_stats* stats)
{
    if (!sc || !prm || !out_rgb) return fail(CRT_ERR_INVALID_ARG, "crt_render: null argument");
    if (prm->world == 0 || prm->rank >= prm->world || prm->width == 0 || prm->height == 0) return fail(CRT_ERR_INVALID_ARG, "crt_render: bad shard or size");
    try {
        HIP_CHECK(hipSetDevice(sc->device));
        const bool tiled = (prm->flags & CRT_FLAG_TILED_OUTPUT) != 0;
        uint64_t npix = tiled ? make_shard(prm->width, prm->height, prm->world).nslots : (uint64_t)prm->width * prm->height;
        DevBuf<uint8_t> d_rgb;
        DevBuf<float> d_mean;
        d_rgb.alloc(npix * 3);
        if (out_mean) d_mean.alloc(npix * 3);
        int rc = render_impl(sc, cam, prm, d_rgb.p, out_mean ? d_mean.p : nullptr, nullptr, stats);
        if (rc != CRT_OK) return rc;
        HIP_CHECK(hipDeviceSynchronize()); // Render.cuh:440
        HIP_CHECK(hipMemcpy(out_rgb, d_rgb.p, npix * 3, hipMemcpyDeviceToHost)); // Render.cuh:464
        if (out_mean) HIP_CHECK(hipMemcpy(out_mean, d_mean.p, npix * 3 * sizeof(float), hipMemcpyDeviceToHost));
        return CRT_OK;
    } catch (const HipFail& f) {
        return fail_hip(f);
    }
}

int crt_render_range_device(crt_scene* sc, const crt_camera* cam, const crt_params* prm, uint32_t sample_begin, uint32_t sample_count,
                            void* d_rgb, void* d_mean, void* stream, crt_stats* stats)
{
    if (sample_count == 0xffffffffu) return fail(CRT_ERR_INVALID_ARG, "crt_render_range: bad sample count");
    return render_impl(sc, cam, prm, d_rgb, d_mean, (hipStream_t)stream, stats, sample_begin, sample_count);
}

int crt_render_range(crt_scene* sc, const crt_camera* cam, const crt_params* prm, uint32_t sample_begin, uint32_t sample_count,
                     uint8_t* out_rgb, float* out_mean, crt_stats* stats)
{
    if (!sc || !prm) return fail(CRT_ERR_INVALID_ARG, "crt_render_range: null argument");
    if (prm->world == 0 || prm->rank >= prm->world || prm->width == 0 || prm->height == 0) return fail(CRT_ERR_INVALID_ARG, "crt_render_range: bad shard or size");
    if (sample_count == 0xffffffffu || sample_count == 0 || (uint64_t)sample_begin + sample_count > prm->spp)
        return fail(CRT_ERR_INVALID_ARG, "crt_render_range: sample range outside [0, spp)");
    const bool last = sample_begin + sample_count == prm->spp;
    if (last && !out_rgb) return fail(CRT_ERR_INVALID_ARG, "crt_render_range: the range that ends at spp needs a frame buffer");
    try {
        HIP_CHECK(hipSetDevice(sc->device));
        const bool tiled = (prm->flags & CRT_FLAG_TILED_OUTPUT) != 0;
        uint64_t npix = tiled ? make_shard(prm->width, prm->height, prm->world).nslots : (uint64_t)prm->width * prm->height;
        DevBuf<uint8_t> d_rgb;
        DevBuf<float> d_mean;
        if (last) d_rgb.alloc(npix * 3);
        if (last && out_mean) d_mean.alloc(npix * 3);
        int rc = render_impl(sc, cam, prm, last ? d_rgb.p : nullptr, last && out_mean ? d_mean.p : nullptr, nullptr, stats, sample_begin, sample_count);
        if (rc != CRT_OK) return rc;
        HIP_CHECK(hipDeviceSynchronize());
        if (last) {
            HIP_CHECK(hipMemcpy(out_rgb, d_rgb.p, npix * 3, hipMemcpyDeviceToHost));
            if (out_mean) HIP_CHECK(hipMemcpy(out_mean, d_mean.p, npix * 3 * sizeof(float), hipMemcpyDeviceToHost));
        }
        return CRT_OK;
    } catch (const HipFail& f) {
        return fail_hip(f);
    }
}

int crt_last_launch_ms(crt_scene* sc, float* ms, uint32_t* launches)
{
    if (!sc || !ms) return fail(CRT_ERR_INVALID_ARG, "crt_last_launch_ms: null argument");
    if (sc->last_launches == 0) return fail(CRT_ERR_INVALID_ARG, "crt_last_launch_ms: no frame has been rendered by the megakernel on this handle");
    hipError_t e = hipEventElapsedTime(ms, sc->ev_k0, sc->ev_k1);
    if (e != hipSuccess) return fail(CRT_ERR_HIP, std::string("crt_last_launch_ms: hipEventElapsedTime: ") + hipGetErrorString(e) + " (synchronize the stream first)");
    if (launches) *launches = sc->last_launches;
    return CRT_OK;
}

int crt_radiance_storage(crt_scene* sc, uint64_t* bytes, uint32_t* ring_samples)
{
    if (!sc || !bytes) return fail(CRT_ERR_INVALID_ARG, "crt_radiance_storage: null argument");
    *bytes = sc->last_radiance_bytes;
    if (ring_samples) *ring_samples = sc->last_ring_samples;
    return CRT_OK;
}

int crt_preview_device(crt_scene* sc, void* d_rgb, void* d_mean, void* stream, uint32_t* samples_done)
{
    if (!sc || !d_rgb) return fail(CRT_ERR_INVALID_ARG, "crt_preview: null argument");
    if (sc->acc.samples == 0) return fail(CRT_ERR_INVALID_ARG, "crt_preview: no progressive render in flight (submit a range that ends before spp first)");
    try {
        HIP_CHECK(hipSetDevice(sc->device));
        Shard sh = make_shard(sc->acc.width, sc->acc.height, sc->acc.world);
        AParams A;
        std::memset(&A, 0, sizeof(A));
        A.width = sc->acc.width; A.height = sc->acc.height; A.spp = sc->acc.spp;
        A.rank = sc->acc.rank; A.world = sc->acc.world; A.tiles_x = sh.tiles_x; A.n_tiles = sh.n_tiles;
        A.nslots = sh.nslots; A.tiled_output = sc->acc.tiled;
        A.accum = sc->accum.p;
        A.out_rgb = (uint8_t*)d_rgb; A.out_mean = (float*)d_mean;
        const float scale = (float)sc->acc.spp / (float)sc->acc.samples;
        hipLaunchKernelGGL(k_preview, dim3((sh.nslots + 255) / 256), dim3(256), 0, (hipStream_t)stream, A, scale);
        HIP_CHECK(hipGetLastError());
        if (samples_done) *samples_done = sc->acc.samples;
        return CRT_OK;
    } catch (const HipFail& f) {
        return fail_hip(f);
    }
}

int crt_preview(crt_scene* sc, uint8_t* out_rgb, float* out_mean, uint32_t* samples_done)
{
    if (!sc || !out_rgb) return fail(CRT_ERR_INVALID_ARG, "crt_preview: null argument");
    if (sc->acc.samples == 0) return fail(CRT_ERR_INVALID_ARG, "crt_preview: no progressive render in flight (submit a range that ends before spp first)");
    try {
        HIP_CHECK(hipSetDevice(sc->device));
        const uint64_t npix = sc->acc.tiled ? make_shard(sc->acc.width, sc->acc.height, sc->acc.world).nslots : (uint64_t)sc->acc.width * sc->acc.height;
        DevBuf<uint8_t> d_rgb;
        DevBuf<float> d_mean;
        d_rgb.alloc(npix * 3);
        if (out_mean) d_mean.alloc(npix * 3);
        if (sc->acc.tiled) HIP_CHECK(hipMemset(d_rgb.p, 0, npix * 3));
        int rc = crt_preview_device(sc, d_rgb.p, out_mean ? d_mean.p : nullptr, nullptr, samples_done);
        if (rc != CRT_OK) return rc;
        HIP_CHECK(hipDeviceSynchronize());
        HIP_CHECK(hipMemcpy(out_rgb, d_rgb.p, npix * 3, hipMemcpyDeviceToHost));
        if (out_mean) HIP_CHECK(hipMemcpy(out_mean, d_mean.p, npix * 3 * sizeof(float), hipMemcpyDeviceToHost));
        return CRT_OK;
    } catch (const HipFail& f) {
        return fail_hip(f);
    }
}

int crt_intersect(crt_scene* sc, uint32_t n, const float* origins, const float* dirs, uint32_t traversal, int32_t* out_tri, float* out_t)
{
    if (!sc || !origins || !dirs || !out_tri || !out_t) return fail(CRT_ERR_INVALID_ARG, "crt_intersect: null argument");
    const bool raw_dir = (traversal & CRT_INTERSECT_RAW_DIRECTIONS) != 0;
    const bool force_exact = (traversal & CRT_INTERSECT_FORCE_EXACT) != 0;
    const bool any_hit = (traversal & CRT_INTERSECT_VISIBILITY) != 0;
    traversal &= ~(uint32_t)(CRT_INTERSECT_RAW_DIRECTIONS | CRT_INTERSECT_FORCE_EXACT | CRT_INTERSECT_VISIBILITY);
    if (traversal != CRT_TRAVERSAL_FAST && traversal != CRT_TRAVERSAL_REFERENCE && traversal != CRT_TRAVERSAL_EXACT)
        return fail(CRT_ERR_INVALID_ARG, "crt_intersect: unknown traversal mode");
    if (n == 0) return CRT_OK;
    try {
        HIP_CHECK(hipSetDevice(sc->device));
        DevBuf<float> o, d, lim;
        o.alloc(n * 3ull); d.alloc(n * 3ull);
        HIP_CHECK(hipMemcpy(o.p, origins, n * 12ull, hipMemcpyHostToDevice));
        HIP_CHECK(hipMemcpy(d.p, dirs, n * 12ull, hipMemcpyHostToDevice));
        if (any_hit) { lim.alloc(n); HIP_CHECK(hipMemcpy(lim.p, out_t, n * 4ull, hipMemcpyHostToDevice)); }
        // blocked() of Render.cuh:19-27 from a finished visibility ray (limit = out_t[i] on entry): REFERENCE compares the closest
        // hit, FAST recorded a hit only if it passes the comparison (shadow_blocked)
        const bool reference_mode = traversal == CRT_TRAVERSAL_REFERENCE;
        auto answer = [&](uint32_t i, float T, int32_t tri) {
            if (!any_hit) { out_t[i] = T; out_tri[i] = tri; return; }
            const float tl = out_t[i];
            const bool blocked = reference_mode ? (tl - T > CRT_EPSILON) : (tri >= 0 || tl - FLT_MAX > CRT_EPSILON);
            out_t[i] = blocked ? 1.0f : 0.0f;
            out_tri[i] = blocked ? tri : -1;
        };
        sc->p_ro.ensure(n); sc->p_rd.ensure(n); sc->p_res.ensure(n);
        Pool pool;
        std::memset(&pool, 0, sizeof(pool));
        pool.ro = sc->p_ro.p; pool.rd = sc->p_rd.p; pool.res = sc->p_res.p; pool.n = n;
        hipLaunchKernelGGL(k_fill_rays, dim3((n + 255) / 256), dim3(256), 0, 0, pool, n, o.p, d.p, raw_dir, any_hit ? lim.p : (const float*)nullptr);
        HIP_CHECK(hipGetLastError());
        if (choose_pipeline(sc) == 4) {
            // the rays walk the traversal phases of the render kernel itself (k_mega3 in query form: work item = ray)
            const bool reference = traversal == CRT_TRAVERSAL_REFERENCE;
            int per_cu = 1;
            const bool exact = traversal == CRT_TRAVERSAL_EXACT;
            const int mode3 = reference ? 1 : exact ? 2 : 0;
            const bool dec = use_dec(sc, mode3);
            const bool r16 = use_ref16(sc, mode3, dec);
            const Mega3Kernel kern3 = mega3_kernel(mode3, false, false, true, r16, false, dec);
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern3, 64, 0) != hipSuccess || per_cu < 1) per_cu = 1;
            const uint32_t pool_p = mega3_pool_p(dec, false);
            const uint32_t blocks = std::min<uint32_t>((n + pool_p - 1) / pool_p, (uint32_t)(sc->n_cus * per_cu));
            const uint32_t lanes = blocks * pool_p;
            sc->p_la.ensure(lanes); sc->p_id.ensure(lanes); sc->L.ensure(n);
            sc->spill[0].ensure((size_t)(r16 ? std::max(1, sc->stack_cap) : std::max(1, sc->stack_cap - mega3_lds_levels(dec, r16))) * lanes);
            MParams3 M3;
            std::memset(&M3, 0, sizeof(M3));
            LParams& P = M3.M.P;
            P.sc = sc->dev;
            P.pool.la = sc->p_la.p; P.pool.id = sc->p_id.p; P.pool.n = lanes;
            P.n_items = n;
            P.items_per_shard = ((n + ITEM_SHARDS - 1) / ITEM_SHARDS + 63u) & ~63u;
            P.item_next = sc->item_next.p; P.L = sc->L.p; P.counters = sc->counters.p;
            P.q_o = sc->p_ro.p; P.q_d = sc->p_rd.p;
            P.nslots = 1; P.nslots_div = make_fastdiv(1); P.tiles_x = 1; P.tiles_x_div = make_fastdiv(1); P.lsn_div = make_fastdiv(1);
            M3.M.sc = sc->dev; M3.M.counters = sc->counters.p; M3.M.spill_stride = lanes; M3.M.stack_cap = mega3_lds_levels(dec, r16);
            M3.spill = (int*)sc->spill[0].p;
            M3.force_exact = force_exact ? 1u : 0u;
            HIP_CHECK(hipMemsetAsync(sc->item_next.p, 0, (size_t)ITEM_SHARDS * ITEM_STRIDE * sizeof(unsigned int), nullptr));
            hipLaunchKernelGGL(kern3, dim3(blocks), dim3(64), 0, nullptr, M3);
            HIP_CHECK(hipGetLastError());
            HIP_CHECK(hipDeviceSynchronize());
            std::vector<float4> res(n);
            HIP_CHECK(hipMemcpy(res.data(), sc->L.p, n * sizeof(float4), hipMemcpyDeviceToHost));
            for (uint32_t i = 0; i < n; i++) {
                int32_t tri;
                std::memcpy(&tri, &res[i].y, 4);
                answer(i, res[i].x, tri);
            }
            return CRT_OK;
        }
        TraceSetup TS = make_trace_setup(sc, pool, traversal, false);
        launch_trace_pass(sc, TS, nullptr);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipDeviceSynchronize());
        std::vector<float2> res(n);
        HIP_CHECK(hipMemcpy(res.data(), sc->p_res.p, n * sizeof(float2), hipMemcpyDeviceToHost));
        for (uint32_t i = 0; i < n; i++) {
            int32_t tri;
            std::memcpy(&tri, &res[i].y, 4);
            answer(i, res[i].x, tri);
        }
        return CRT_OK;
    } catch (const HipFail& f) {
        return fail_hip(f);
    }
}

int crt_device_math(int device, const char* fn, uint32_t n, const float* a, const float* b, float* out)
{
    if (!fn || !a || !out) return fail(CRT_ERR_INVALID_ARG, "crt_device_math: null argument");
    static const char* names[] = {"sin", "cos", "tan", "acos", "atan2", "exp", "log10", "pow", "uniform", "sincos_s", "sincos_c", "div_short", "div_short_bounded"};
    int id = -1;
    for (int i = 0; i < 13; i++)
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

int crt_device_rcp_check(int device, uint64_t* mismatches, uint64_t* outside)
{
    if (!mismatches || !outside) return fail(CRT_ERR_INVALID_ARG, "crt_device_rcp_check: null argument");
    try {
        HIP_CHECK(hipSetDevice(device));
        DevBuf<unsigned long long> c;
        c.alloc(2);
        HIP_CHECK(hipMemset(c.p, 0, 2 * sizeof(unsigned long long)));
        hipLaunchKernelGGL(k_rcp_check, dim3(256 * 32), dim3(256), 0, 0, c.p);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipDeviceSynchronize());
        unsigned long long h[2];
        HIP_CHECK(hipMemcpy(h, c.p, sizeof(h), hipMemcpyDeviceToHost));
        *mismatches = h[0]; *outside = h[1];
        return CRT_OK;
    } catch (const HipFail& f) {
        return fail_hip(f);
    }
}

} // extern "C"

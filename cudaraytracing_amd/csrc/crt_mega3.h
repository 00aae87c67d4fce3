// cudaraytracing_amd/csrc/crt_mega3.h -- the megakernel's launch parameters, pool layouts in LDS and limits, and what crt_mega3.hip
// exports to the host code (crt_render.hip).
#ifndef CRT_MEGA3_H
#define CRT_MEGA3_H
#include "crt_path.h"

namespace crtk {

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
#ifndef CRT_RING_MODE
#define CRT_RING_MODE 2 /* 2: the rings are STACKS -- a batch is the newest ids, one scalar per ring, no wrap (round 5: C2 -1.3 %, veach-mis -1.1 % on top of
                           the mask form of the inner step); 0: FIFO rings that wrap by compare (rounds 2 - 4); 1: FIFO rings of 256 entries, wrap by mask,
                           146 rays per wave (measured: +0.5 % / +0.1 % against 0 -- the smaller pool costs what the mask saves) (crt_mega3.hip: ring_wrap) */
#endif
#ifndef POOL3_P
#if CRT_RING_MODE == 1
#define POOL3_P 154
#else
#define POOL3_P 164         /* 164 x 56 B + rings = 10 004 B: 16 waves per CU (measured with the 16-bit stack layout: 148 rays x 64 B records
                               with 1/d and six levels 102.5 ms, 176 x 52 B with six levels 100.2, 164 x 56 B with eight levels 98.8, 156 x 60 B with
                               ten 100.0) */
#endif
#endif
#if CRT_RING_MODE == 1
#define POOL3_QCAP 256
#else
#define POOL3_QCAP ((POOL3_P + 3) & ~3) /* ring capacity (any number >= POOL3_P: indices wrap by compare, not by mask); ids fit a byte */
#endif
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
#define LEAFQ_CAP 256 /* entries of the leaf queue, a power of two; the inner step counts a visit's entries before it writes any (crt_mega3.hip: inner4_step_dec) */
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
    static constexpr int P = CRT_RING_MODE == 1 ? (RING_ ? 140 : 146) : (RING_ ? 148 : 152);
#endif
    static constexpr int QCAP = CRT_RING_MODE == 1 ? 256 : (P + 3) & ~3;
#ifdef POOL4_LV
    static constexpr int LV = R16_ ? POOL4_LV : POOL4_LV / 2;
#else
    static constexpr int LV = R16_ ? 6 : 3;
#endif
    typedef typename std::conditional<R16_, short, int>::type stk_t;
    float4 A[P];                 // origin.xyz, the distance an accepted hit must stay below by more than EPSILON: the light's for an any-hit ray, +inf otherwise
    float4 B[P];                 // direction.xyz, bits(current node ref)
    unsigned long long best[P];  // the ray's answer so far: bits(distance) << 32 | ~triangle; FLT_MAX << 32 | 0 = nothing
    stk_t stk[LV][P];            // traversal stack (inner nodes only); deeper levels spill to global memory
    uint32_t D[P];               // stack depth (bits 0-7) | entries in flight (RD_PEND_MASK) | RF_* flags, RD_FIN
    uint32_t leafq[LEAFQ_CAP + 1]; // (+ a spare dword: where the lanes without an entry write in the branch-free append)
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

static_assert(offsetof(MParams3, dbg_loads) == 748 && offsetof(MParams3, dbg_valu) == 752, "tools/bbprof/instrument.py reads the counter buffer's address from these two kernel-argument dwords");


// ---- exported by crt_mega3.hip ----
typedef void (*Mega3Kernel)(const MParams3);
// The instantiation of k_mega3 for a traversal mode (0 FAST, 1 REFERENCE, 2 EXACT), with or without counters, every sample traced
// or not, render or query form, 32- or 16-bit stack entries (never for REFERENCE), commit ring, decoupled leaves (EXACT only)
Mega3Kernel mega3_kernel(int mode, bool stats, bool all, bool query, bool r16, bool ring = false, bool dec = false, bool impl = false); // impl: the tree without its rows of refs (dec && r16 only)
uint32_t mega3_pool_p(bool dec, bool ring);   // rays per wave of that kernel's pool
int mega3_lds_levels(bool dec, bool r16);     // traversal-stack levels it keeps in LDS
bool bbprof_launch(Mega3Kernel kern, MParams3 M3, uint32_t blocks, hipStream_t st); // tools/bbprof hook (false: launch as usual)
void launch_order_items(bool ring, uint32_t blocks, hipStream_t st, const LParams& P, uint32_t* list, unsigned int* cnt);

} // namespace crtk
#endif

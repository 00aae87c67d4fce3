// cudaraytracing_amd/csrc/crt_device.h -- device-side data model and small
// vector helpers shared by the HIP kernels and the upload code.
//
// HBM layout (all read-only during a render, replicated per GPU):
//   nodes      4 x float4 per INNER node; two trees over the same leaves live in this array, the
//              SAH tree of crt_accel.h (root_fast) and the reference topology (root_exact):
//                [0] = (left.lo.xyz , bits(left_ref))   [1] = (left.hi.xyz , bits(right_ref))
//                [2] = (right.lo.xyz, 0)                [3] = (right.hi.xyz, 0)
//              both child boxes live in the parent: one 64 B fetch per inner-node
//              visit instead of the reference's 3 x 40 B (DeviceBVH.cuh:140,151-152).
//              child ref >= 0: inner node index; < 0: leaf, ~ref = (first_tri << 4) | n
//              with n in 1..15 (n = 0: count looked up in leaf_count[first_tri]).
//   tri_geo    3 x float4 per triangle (BVH order): (v1.xyz, e1.x) (e1.yz, e2.xy) (e2.z, n.xyz)
//              48 B instead of the reference's 140 B DeviceTriangle.
//   tri_mat    int32 material index per triangle
//   mats       3 x float4 per material: (kd/pi .xyz, ns) (kd.xyz, bits(flags)) (ke.xyz, 0)
//              flags bit0 = has_emit, bit1 = SPECULAR
//   ltri       4 x float4 per light triangle (shape order): (v1.xyz, v2.x) (v2.yz, v3.xy)
//              (v3.z, n.xyz) (ke.xyz, area_of_obj)
//   lights     uint4 (first, count, division magic, shifts) per light object
#ifndef CRT_DEVICE_H
#define CRT_DEVICE_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "crt_detmath.h"

namespace crtdev {

#define CRT_EPSILON 0.00001f       /* reference: include/Global.h:11 */
#define CRT_BOUNCE_STACK_SIZE 64   /* reference: include/Global.h:18 */
#define CRT_TILE 8                 /* pixel tile edge used for sharding */

#ifndef NODE4I_F4
#define NODE4I_F4 6 /* float4 per node of nodes4i: six rows of planes (96 B); 8 = padded to one 128-byte line per node (an A/B of round 6) */
#endif
#define TNM_MAT(w_) ((w_) & 0x3fffffffu)       /* tri_nm row, word 3: the material index ... */
#define TNM_EMITTER(w_) (((w_) >> 30) & 1u)  /* ... "its material emits" (Render.cuh:210) ... */
#define TNM_SPECULAR(w_) ((w_) >> 31)          /* ... "its material is SPECULAR" (Render.cuh:294) */

struct DevScene {
    const float4* nodes;
    const float4* tri_geo;
    const int32_t* tri_mat;
    const float4* mats;
    const float4* ltri;
    const uint4* lights;
    const int32_t* leaf_count; // only read for leaves with more than 15 triangles
    int32_t root_fast;   // root of the SAH tree over the reference leaves (crt_accel.h)
    int32_t root_exact;  // root of the reference-topology tree
    int32_t n_lights;
    // k_mega3's copies of the two trees: child pairs packed per coordinate, leaves as triangle-pair records
    const float4* nodes3;    // 4 x float4 per inner node (same node numbering as `nodes`):
                             //   [0] = (lo.x L, lo.x R, lo.y L, lo.y R)  [1] = (lo.z L, lo.z R, hi.x L, hi.x R)
                             //   [2] = (hi.y L, hi.y R, hi.z L, hi.z R)  [3] = (bits(ref L), bits(ref R), 0, 0)
                             // child ref >= 0: inner node; < 0: ~ref = first record of the leaf in leaf_geo
    const float4* leaf_geo;  // 5 x float4 per record = two consecutive triangles a, b of one leaf:
                             //   (v1.x a, v1.x b, v1.y a, v1.y b) (v1.z a, v1.z b, e1.x a, e1.x b) (e1.y a, e1.y b, e1.z a, e1.z b)
                             //   (e2.x a, e2.x b, e2.y a, e2.y b) (e2.z a, e2.z b, bits(index of a), bits(triangles of the leaf from a on))
                             // a leaf of n triangles owns ceil(n / 2) consecutive records
    int32_t root3_fast, root3_exact;
    const float4* nodes4;    // the SAH tree collapsed to 4 children per node, 8 x float4 (128 B) per node, plane-major: [2a] = lo of axis
                             // a of children 0..3, [2a + 1] = hi of axis a -- a ray loads the near plane [2a + (d_a < 0)] and the far
                             // plane [2a + 1 - (d_a < 0)] of each axis, which is hit_AABB's swap done by the address;
                             // [6] = bits(ref 0..3) (>= 0: nodes4 index, < 0: leaf as in nodes3), [7] padding; an empty slot is the
                             // inverted box (lo = +inf, hi = -inf: entered at +inf, left at -inf for every direction).
                             // (-DCRT_NODE_SIGNSEL=0: the round-1 layout, child pairs as in nodes3, NaN boxes for empty slots)
    int32_t root4;           // root of the 4-wide tree (a leaf ref if the scene is a single leaf)
    float coord_max;         // largest |coordinate| of a box of the 4-wide tree, +inf if one is not finite (start_ray: which rays may walk it)
    const float4* tri_nm;    // (normal.xyz, bits(material | emitter << 30 | SPECULAR << 31): TNM_*) per triangle: what entering a vertex needs, 16 B instead of 48 + 4
    uint32_t empty4_off;     // byte offset in nodes4 of a node of four empty slots, behind the tree (the decoupled-leaves step parks idle lanes there)
    // the 4-wide tree without its rows of refs (round 6, crt_render.hip "nodes4i"): 6 x float4 (96 B) per node, the same plane-major rows [0..5];
    // nodes [0, n_mixed4i) have an inner child, the others only leaves; child refs and leaf records are implied (crt_mega3.hip: inner4_step_dec)
    const float4* nodes4i;
    const float4* leaf_geo_i; // leaf child k of node n: record 4 n + k (a sparse copy of leaf_geo's records, 5 x float4 each)
    const int32_t* rec_map;   // leaf_geo record -> leaf_geo_i record (the reference-arithmetic arm walks nodes3, whose leaf refs are leaf_geo's)
    uint32_t n_mixed4i;
    int32_t root4i;
    uint32_t empty4i_off;    // byte offset in nodes4i of the node of four empty slots (numbered n: a fringe node)
};

// Exact unsigned 32-bit division by a run-time constant without the ~40-instruction hardware-less
// divide sequence (Granlund & Montgomery / Hacker's Delight 10-9): q = (t + ((n - t) >> sh1)) >> sh2,
// t = mulhi(m, n).  Valid for every n and every d >= 1 (tests/test_host_layer.py checks the host maths).
struct FastDiv {
    uint32_t m, sh; // sh = sh1 | sh2 << 8
};
inline FastDiv make_fastdiv(uint32_t d)
{
    uint32_t l = 0;
    while (l < 32 && (1ull << l) < d) l++;
    FastDiv f;
    f.m = (uint32_t)((((1ull << l) - d) << 32) / d + 1);
    f.sh = (l < 1 ? l : 1u) | ((l > 0 ? l - 1 : 0u) << 8);
    return f;
}
__host__ __device__ __forceinline__ uint32_t fast_div(uint32_t n, uint32_t m, uint32_t sh)
{
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t t = __umulhi(m, n);
#else
    uint32_t t = (uint32_t)(((unsigned long long)m * n) >> 32);
#endif
    return (t + ((n - t) >> (sh & 255u))) >> (sh >> 8);
}

struct F3 {
    float x, y, z;
};
__device__ __forceinline__ F3 f3(float x, float y, float z) { F3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ F3 add3(F3 a, F3 b) { return f3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ F3 sub3(F3 a, F3 b) { return f3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ F3 mul3(F3 a, F3 b) { return f3(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ F3 scale3(F3 a, float s) { return f3(a.x * s, a.y * s, a.z * s); }   // v * s
__device__ __forceinline__ F3 scalel3(float s, F3 a) { return f3(s * a.x, s * a.y, s * a.z); }  // s * v
// ---- three quotients by one positive divisor, bit for bit the IEEE quotients (a.x / n, a.y / n, a.z / n), in 12 instructions + a guard
// instead of the 30 of three division expansions.  y = 1 / n by v_rcp_f32 + one Newton step (the correctly rounded reciprocal:
// crt_device_rcp_check, all 2^32 inputs), then Markstein's correction: q0 = a y, r = fma(n, q0, -a), q = fma(-r, y, q0).  That this IS
// the IEEE quotient was checked for ALL 2^23 x 2^23 pairs of mantissas on gfx950 (tools/exhaustive/div_pair_check.hip,
// profiles/r03_div_pair_check.txt: 7.04e13 pairs, 0 mismatches); scaling numerator and divisor by powers of two scales every
// intermediate exactly while nothing leaves the normal range, which the guard ensures: 2^-62 <= n < 2^60 and every numerator zero
// or 2^-64 <= |a| < 2^62 -- then 2^-124 < |a / n| < 2^124, and the residual, a multiple of 2^(exponent(a) - 47), is representable.
// (The residual is formed with the opposite sign so that a zero numerator keeps its sign.)  Lanes outside the guard -- zero, negative
// or non-finite divisors, tiny, huge or non-finite numerators -- take the division itself behind a wave-uniform branch.
// `bounded`: the caller knows |a| <= n (1 + 2^-22) (normalisation): the upper test of the numerators is implied by the divisor's.
__device__ __forceinline__ F3 quot3_exact(const F3 a, const float n, const bool bounded)
{
    const float y0 = __builtin_amdgcn_rcpf(n);
    const float y = __builtin_fmaf(__builtin_fmaf(-n, y0, 1.0f), y0, y0);
    const float qx = a.x * y, qy = a.y * y, qz = a.z * y;
    F3 q = f3(__builtin_fmaf(-__builtin_fmaf(n, qx, -a.x), y, qx), __builtin_fmaf(-__builtin_fmaf(n, qy, -a.y), y, qy),
              __builtin_fmaf(-__builtin_fmaf(n, qz, -a.z), y, qz));
    asm volatile("" : "+v"(q.x), "+v"(q.y), "+v"(q.z)); // (keeps the short form ahead of the branch instead of in an else-arm)
    const uint32_t ex = f2u(a.x) & 0x7fffffffu, ey = f2u(a.y) & 0x7fffffffu, ez = f2u(a.z) & 0x7fffffffu;
    // zero -> 0xffffffff: passes the lower test; anything else below 2^-64 fails it
    const uint32_t lo = min(min(ex - 1u, ey - 1u), ez - 1u);
    bool ok = (f2u(n) - 0x20800000u /* 2^-62 */ < 0x5d800000u - 0x20800000u /* .. 2^60 */) & (lo >= 0x1f800000u /* 2^-64 */ - 1u);
    if (!bounded) ok = ok & (max(max(ex, ey), ez) < 0x5e800000u /* 2^62 */);
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(!ok) != 0, 0)) {
        if (!ok) q = f3(a.x / n, a.y / n, a.z / n);
    }
    return q;
}
__device__ __forceinline__ F3 div3(F3 a, float s) { return quot3_exact(a, s, false); }
// Eigen reduction order: p0 + (p1 + p2)
__device__ __forceinline__ float dot3(F3 a, F3 b) { return a.x * b.x + (a.y * b.y + a.z * b.z); }
__device__ __forceinline__ F3 cross3(F3 a, F3 b)
{
    return f3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
__device__ __forceinline__ float norm3(F3 a) { return sqrt_f(dot3(a, a)); }
__device__ __forceinline__ F3 unit3(F3 a)
{
    float z = dot3(a, a);
    if (z > 0.0f) { float s = sqrt_f(z); return quot3_exact(a, s, true); } // (|a| <= |a|: the divisor bounds the numerators)
    return a;
}
__device__ __forceinline__ float maxf_ref(float x, float y) { return x > y ? x : y; } // Global.h:111-114
__device__ __forceinline__ float minf_ref(float x, float y) { return x < y ? x : y; } // Global.h:116-119

struct RayT {
    F3 o, d, inv;
};
// reference: include/Ray.cuh:12-15 (direction is normalised again, inv_dir may be +-inf)
__device__ __forceinline__ RayT make_ray(F3 o, F3 d)
{
    RayT r;
    r.o = o;
    r.d = unit3(d);
    r.inv = f3(1 / r.d.x, 1 / r.d.y, 1 / r.d.z);
    return r;
}

} // namespace crtdev
#endif

// cudaraytracing_amd/csrc/crt_frame.hip -- the kernels around the render kernels: k_accumulate (per pixel c += L_k / spp in sample order, tone map:
// include/Render.cuh:348-350), k_preview, and the kernels behind crt_intersect's ray upload and the crt_device_* self-tests.
#include "crt_internal.h"

namespace crtk {

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

__device__ __forceinline__ float acc_load(const float* p) { return __uint_as_float(__hip_atomic_load((const unsigned int*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)); }
__device__ __forceinline__ void acc_store(float* p, const float v) { __hip_atomic_store((unsigned int*)p, __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__global__ __launch_bounds__(256) void k_accumulate(const AParams A)
{
    uint32_t slot = blockIdx.x * 256u + threadIdx.x;
    if (slot >= A.nslots) return;
    uint32_t i = 0, j = 0;
    bool valid = slot_to_pixel(slot, A.rank, A.world, A.n_tiles, A.tiles_x, make_fastdiv_dev(A.tiles_x), A.width, A.height, i, j);
    F3 c = f3(0.0f, 0.0f, 0.0f);
    if (valid) {
        // (the accumulator is uncached memory that commit-ring launches read and write with agent-scope atomics: the same accesses here)
        if (!A.first_chunk) c = f3(acc_load(A.accum + slot), acc_load(A.accum + A.nslots + slot), acc_load(A.accum + 2ull * A.nslots + slot));
        const float fspp = (float)A.spp;
        for (uint32_t s = 0; s < A.chunk_samples; s++) { // temp_color += L / spp, in sample order (Render.cuh:348)
            // (agent-scope loads: the radiance was written by the launch before this one, from other XCDs -- the same kind of hand-off as
            // k_order_items -> k_mega3, whose plain loads were seen to return what an earlier kernel had left at the address, docs/experiments.md 6)
            const float* lp = (const float*)&A.L[(uint64_t)s * A.nslots + slot];
#ifdef CRT_ACCUM_PLAIN
            const float lx = lp[0], ly = lp[1], lz = lp[2];
#else
            const float lx = acc_load(lp), ly = acc_load(lp + 1), lz = acc_load(lp + 2);
#endif
            c.x = c.x + lx / fspp;
            c.y = c.y + ly / fspp;
            c.z = c.z + lz / fspp;
        }
        if (!A.last_chunk) {
            acc_store(A.accum + slot, c.x); acc_store(A.accum + A.nslots + slot, c.y); acc_store(A.accum + 2ull * A.nslots + slot, c.z);
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


// ---- exported to crt_render.hip ----
void launch_accumulate(const AParams& A, hipStream_t st) { hipLaunchKernelGGL(k_accumulate, dim3((A.nslots + 255) / 256), dim3(256), 0, st, A); }
void launch_preview(const AParams& A, float scale, hipStream_t st) { hipLaunchKernelGGL(k_preview, dim3((A.nslots + 255) / 256), dim3(256), 0, st, A, scale); }
void launch_fill_rays(const Pool& pool, uint32_t n, const float* o, const float* d, bool raw_dir, const float* limits)
{
    hipLaunchKernelGGL(k_fill_rays, dim3((n + 255) / 256), dim3(256), 0, 0, pool, n, o, d, raw_dir, limits);
}
void launch_math(int fn, uint32_t n, const float* a, const float* b, float* out) { hipLaunchKernelGGL(k_math, dim3((n + 255) / 256), dim3(256), 0, 0, fn, n, a, b, out); }
void launch_philox(uint32_t n, const uint32_t* ctr, const uint32_t* key, uint32_t* out) { hipLaunchKernelGGL(k_philox, dim3((n + 255) / 256), dim3(256), 0, 0, n, ctr, key, out); }
void launch_rcp_check(unsigned long long* counts) { hipLaunchKernelGGL(k_rcp_check, dim3(256 * 32), dim3(256), 0, 0, counts); }

} // namespace crtk

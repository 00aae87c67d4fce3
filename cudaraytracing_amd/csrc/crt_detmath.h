// cudaraytracing_amd/csrc/crt_detmath.h
//
// Bit-reproducible single-precision transcendental functions and the Philox
// counter RNG used by the HIP kernels (and by the host side of libcrt for
// tan(fovY/2)).  The reference calls CUDA libdevice (sinf, cosf, acosf,
// atan2f, expf, log10f, powf, tanf: include/Global.h:40-93,
// include/Render.cuh:297,306,338,350) and cuRAND (Global.h:52-55,106-109);
// neither exists on this platform, and results must not depend on which libm
// evaluates them, so the kernels evaluate the published Cephes float
// algorithms with IEEE + - * / sqrt, floor and integer ops only.
// Compile with -ffp-contract=off (no FMA contraction) and correctly rounded
// division / sqrt (hipcc default, -fhip-fp32-correctly-rounded-divide-sqrt).
#ifndef CRT_DETMATH_H
#define CRT_DETMATH_H

#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define CRT_HD __host__ __device__ __forceinline__
#else
#include <cmath>
#define CRT_HD inline
#endif

namespace crtdev {

CRT_HD uint32_t f2u(float f) { return __builtin_bit_cast(uint32_t, f); }
CRT_HD float u2f(uint32_t u) { return __builtin_bit_cast(float, u); }
CRT_HD float qnan() { return u2f(0x7fc00000u); }
CRT_HD float pinf() { return u2f(0x7f800000u); }
CRT_HD float absf(float x) { return u2f(f2u(x) & 0x7fffffffu); }
CRT_HD float floor_f(float x) { return __builtin_floorf(x); }
CRT_HD float sqrt_f(float x) { return __builtin_sqrtf(x); }

// ---- sin / cos : Cody-Waite reduction by pi/4 octants + degree-7/8 minimax ----
struct Octant { float r; int q; };
CRT_HD Octant reduce_octant(float ax)
{
    const float FOPI = 1.27323954473516f;
    float j = floor_f(ax * FOPI);
    float h = floor_f(j * 0.5f);
    if (j - 2.0f * h != 0.0f) j = j + 1.0f;
    float q = j - 8.0f * floor_f(j * 0.125f);
    Octant o;
    o.r = ((ax - j * 0.78515625f) - j * 2.4187564849853515625e-4f) - j * 3.77489497744594108e-8f;
    o.q = (int)q;
    return o;
}
CRT_HD float poly_sin(float r, float z)
{
    return ((-1.9515295891E-4f * z + 8.3321608736E-3f) * z - 1.6666654611E-1f) * z * r + r;
}
CRT_HD float poly_cos(float z)
{
    float y = ((2.443315711809948E-5f * z - 1.388731625493765E-3f) * z + 4.166664568298827E-2f) * z * z;
    y = y - 0.5f * z;
    return y + 1.0f;
}
CRT_HD float det_sinf(float x)
{
    float ax = absf(x);
    if (!(ax <= 1.0e30f)) return qnan();
    float sign = x < 0.0f ? -1.0f : 1.0f;
    Octant o = reduce_octant(ax);
    int q = o.q;
    if (q > 3) { sign = -sign; q -= 4; }
    float z = o.r * o.r;
    float y = (q == 1 || q == 2) ? poly_cos(z) : poly_sin(o.r, z);
    return sign * y;
}
CRT_HD float det_cosf(float x)
{
    float ax = absf(x);
    if (!(ax <= 1.0e30f)) return qnan();
    float sign = 1.0f;
    Octant o = reduce_octant(ax);
    int q = o.q;
    if (q > 3) { sign = -sign; q -= 4; }
    if (q > 1) sign = -sign;
    float z = o.r * o.r;
    float y = (q == 1 || q == 2) ? poly_sin(o.r, z) : poly_cos(z);
    return sign * y;
}
// sin and cos of the same argument share one reduction
CRT_HD void det_sincosf(float x, float* s, float* c)
{
    float ax = absf(x);
    if (!(ax <= 1.0e30f)) { *s = qnan(); *c = qnan(); return; }
    Octant o = reduce_octant(ax);
    float z = o.r * o.r;
    float ps = poly_sin(o.r, z), pc = poly_cos(z);
    int q = o.q;
    float ssign = x < 0.0f ? -1.0f : 1.0f, csign = 1.0f;
    if (q > 3) { ssign = -ssign; csign = -csign; q -= 4; }
    if (q > 1) csign = -csign;
    bool swap = (q == 1 || q == 2);
    *s = ssign * (swap ? pc : ps);
    *c = csign * (swap ? ps : pc);
}
CRT_HD float det_tanf(float x) { return det_sinf(x) / det_cosf(x); }

// ---- asin / acos ----
CRT_HD float det_asinf(float x)
{
    float sign = x < 0.0f ? -1.0f : 1.0f;
    float a = absf(x);
    if (!(a <= 1.0f)) return qnan();
    if (a < 1.0e-4f) return sign * a;
    bool big = a > 0.5f;
    float z, w;
    if (big) { z = 0.5f * (1.0f - a); w = sqrt_f(z); }
    else { w = a; z = w * w; }
    float p = ((((4.2163199048E-2f * z + 2.4181311049E-2f) * z + 4.5470025998E-2f) * z + 7.4953002686E-2f) * z
               + 1.6666752422E-1f) * z * w + w;
    if (big) { p = p + p; p = 1.57079632679489661923f - p; }
    return sign * p;
}
CRT_HD float det_acosf(float x)
{
    if (!(x >= -1.0f && x <= 1.0f)) return qnan();
    if (x < -0.5f) return 3.14159265358979323846f - 2.0f * det_asinf(sqrt_f(0.5f * (1.0f + x)));
    if (x > 0.5f) return 2.0f * det_asinf(sqrt_f(0.5f * (1.0f - x)));
    return 1.57079632679489661923f - det_asinf(x);
}
// ---- atan / atan2 ----
CRT_HD float det_atanf(float x)
{
    float sign = x < 0.0f ? -1.0f : 1.0f;
    float a = absf(x);
    if (a != a) return qnan();
    float y;
    if (a > 2.414213562373095f) { y = 1.57079632679489661923f; a = -(1.0f / a); }
    else if (a > 0.4142135623730950f) { y = 0.78539816339744830962f; a = (a - 1.0f) / (a + 1.0f); }
    else y = 0.0f;
    float z = a * a;
    y = y + ((((8.05374449538e-2f * z - 1.38776856032E-1f) * z + 1.99777106478E-1f) * z - 3.33329491539E-1f) * z * a + a);
    return sign * y;
}
CRT_HD float det_atan2f(float y, float x)
{
    const float PIF = 3.14159265358979323846f, PIO2F = 1.57079632679489661923f;
    if (x != x || y != y) return qnan();
    int code = 0;
    if (x < 0.0f) code = 2;
    if (y < 0.0f) code |= 1;
    if (x == 0.0f) {
        if (code & 1) return -PIO2F;
        if (y == 0.0f) return 0.0f;
        return PIO2F;
    }
    if (y == 0.0f) return (code & 2) ? PIF : 0.0f;
    float w = code == 2 ? PIF : (code == 3 ? -PIF : 0.0f);
    return w + det_atanf(y / x);
}
// ---- exp / log / pow ----
CRT_HD float pow2i(int n) { return u2f((uint32_t)(n + 127) << 23); }
CRT_HD float det_expf(float x)
{
    if (x != x) return qnan();
    if (x > 88.72283905206835f) return pinf();
    if (x < -87.0f) return 0.0f;
    float zf = floor_f(1.44269504088896341f * x + 0.5f);
    float r = x - zf * 0.693359375f;
    r = r - zf * -2.12194440e-4f;
    float z = r * r;
    z = (((((1.9875691500E-4f * r + 1.3981999507E-3f) * r + 8.3334519073E-3f) * r + 4.1665795894E-2f) * r
          + 1.6666665459E-1f) * r + 5.0000001201E-1f) * z + r + 1.0f;
    int n = (int)zf;
    if (n > 127) return (z * 2.0f) * pow2i(n - 1);
    return z * pow2i(n);
}
CRT_HD float det_logf(float x)
{
    if (x != x || x < 0.0f) return qnan();
    if (x == 0.0f) return -pinf();
    if (x == pinf()) return x;
    int e = 0;
    uint32_t u = f2u(x);
    if (u < 0x00800000u) { x = x * 16777216.0f; e = -24; u = f2u(x); }
    e += (int)(u >> 23) - 126;
    float m = u2f((u & 0x007fffffu) | 0x3f000000u);
    if (m < 0.707106781186547524f) { e -= 1; m = m + m - 1.0f; }
    else m = m - 1.0f;
    float z = m * m;
    float y = ((((((((7.0376836292E-2f * m - 1.1514610310E-1f) * m + 1.1676998740E-1f) * m - 1.2420140846E-1f) * m
                   + 1.4249322787E-1f) * m - 1.6668057665E-1f) * m + 2.0000714765E-1f) * m - 2.4999993993E-1f) * m
               + 3.3333331174E-1f) * m * z;
    float fe = (float)e;
    if (e != 0) y = y + -2.12194440e-4f * fe;
    y = y + -0.5f * z;
    z = m + y;
    if (e != 0) z = z + 0.693359375f * fe;
    return z;
}
CRT_HD float det_log10f(float x) { return det_logf(x) * 0.43429448190325176f; }
CRT_HD float det_powf(float x, float y)
{
    if (x != x || y != y) return qnan();
    if (x < 0.0f) return qnan();
    if (x == 0.0f) return y > 0.0f ? 0.0f : (y == 0.0f ? 1.0f : pinf());
    if (x == 1.0f) return 1.0f;
    return det_expf(y * det_logf(x));
}

// ---- Philox4x32-10 (Salmon et al., SC'11) with explicit draw addressing ----
// key = {pixel_index, seed_lo}; counter = {sample_k, depth | purpose << 16, idx, seed_hi}
enum { RNG_JITTER = 0, RNG_BOUNCE = 1, RNG_NEE = 2, RNG_PROBE = 3 };
struct U4 { uint32_t x, y, z, w; };
CRT_HD U4 philox4x32_10(U4 c, uint32_t k0, uint32_t k1)
{
#pragma unroll
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c.x;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c.z;
        U4 n;
        n.x = (uint32_t)(p1 >> 32) ^ c.y ^ k0;
        n.y = (uint32_t)p1;
        n.z = (uint32_t)(p0 >> 32) ^ c.w ^ k1;
        n.w = (uint32_t)p0;
        c = n;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return c;
}
CRT_HD U4 rng_draw(uint64_t seed, uint32_t pixel, uint32_t k, uint32_t depth, uint32_t purpose, uint32_t idx)
{
    U4 c;
    c.x = k; c.y = depth | (purpose << 16); c.z = idx; c.w = (uint32_t)(seed >> 32);
    return philox4x32_10(c, pixel, (uint32_t)seed);
}
// curand_uniform's (0, 1] mapping
CRT_HD float rng_uniform(uint32_t x) { return (float)x * 2.3283064365386963e-10f + 1.1641532182693481e-10f; }

} // namespace crtdev
#endif

/*
 * oracle/det_math.h -- TEST INFRASTRUCTURE (CPU oracle), not product code.
 *
 * Deterministic single-precision replacements for the libm / libdevice calls
 * the reference makes on its hot path (reference: include/Global.h:40-93
 * sqrt/cos/sin/acos/atan2f; include/Render.cuh:297 exp, :306 log10f,
 * :338 tanf, :350 pow).  The reference's own implementations (CUDA libdevice,
 * MSVC libm) are unpinned third-party code that does not exist in this image,
 * so neither the oracle nor the HIP kernel can call them; both sides instead
 * evaluate the SAME published polynomial algorithms (Cephes single-precision
 * library, S. Moshier, sinf.c/asinf.c/atanf.c/expf.c/logf.c) using only
 * IEEE-754 correctly-rounded + - * / sqrt, floor and integer bit operations,
 * compiled with -ffp-contract=off.  The HIP kernel carries its own copy of the
 * same algorithms (cudaraytracing_amd/csrc/crt_detmath.h); tests compare the
 * two bit for bit and both against libm within a few ulp.
 */
#ifndef ORACLE_DET_MATH_H
#define ORACLE_DET_MATH_H

#include <math.h>
#include <stdint.h>
#include <string.h>

static inline uint32_t om_f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float om_u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline float om_nan(void) { return om_u2f(0x7fc00000u); }
static inline float om_inf(void) { return om_u2f(0x7f800000u); }

#define OM_PIF 3.14159265358979323846f
#define OM_PIO2F 1.57079632679489661923f
#define OM_PIO4F 0.78539816339744830962f
#define OM_FOPI 1.27323954473516f /* 4/pi */
#define OM_DP1 0.78515625f
#define OM_DP2 2.4187564849853515625e-4f
#define OM_DP3 3.77489497744594108e-8f

/* Octant reduction shared by sin and cos.  Returns octant in 0..7 (after the
 * "odd -> next even" step) and the reduced argument in [-pi/4, pi/4]. */
static inline int om_reduce_pio4(float ax, float* r)
{
    float j = floorf(ax * OM_FOPI);
    float half = floorf(j * 0.5f);
    if (j - 2.0f * half != 0.0f) j = j + 1.0f; /* odd -> even */
    float jm = j - 8.0f * floorf(j * 0.125f);  /* exact, in [0,8) */
    *r = ((ax - j * OM_DP1) - j * OM_DP2) - j * OM_DP3;
    return (int)jm;
}
static inline float om_sin_poly(float x, float z)
{
    return ((-1.9515295891E-4f * z + 8.3321608736E-3f) * z - 1.6666654611E-1f) * z * x + x;
}
static inline float om_cos_poly(float z)
{
    float y = ((2.443315711809948E-5f * z - 1.388731625493765E-3f) * z + 4.166664568298827E-2f) * z * z;
    y = y - 0.5f * z;
    return y + 1.0f;
}
static inline float om_sinf(float x)
{
    float ax = fabsf(x);
    if (!(ax <= 1.0e30f)) return om_nan();
    float sign = x < 0.0f ? -1.0f : 1.0f;
    float r;
    int jm = om_reduce_pio4(ax, &r);
    if (jm > 3) { sign = -sign; jm -= 4; }
    float z = r * r;
    float y = (jm == 1 || jm == 2) ? om_cos_poly(z) : om_sin_poly(r, z);
    return sign * y;
}
static inline float om_cosf(float x)
{
    float ax = fabsf(x);
    if (!(ax <= 1.0e30f)) return om_nan();
    float sign = 1.0f;
    float r;
    int jm = om_reduce_pio4(ax, &r);
    if (jm > 3) { sign = -sign; jm -= 4; }
    if (jm > 1) sign = -sign;
    float z = r * r;
    float y = (jm == 1 || jm == 2) ? om_sin_poly(r, z) : om_cos_poly(z);
    return sign * y;
}
static inline float om_tanf(float x) { return om_sinf(x) / om_cosf(x); }

static inline float om_asinf(float x)
{
    float sign = x < 0.0f ? -1.0f : 1.0f;
    float a = fabsf(x);
    if (!(a <= 1.0f)) return om_nan();
    if (a < 1.0e-4f) return sign * a;
    int flag = 0;
    float z, w;
    if (a > 0.5f) { z = 0.5f * (1.0f - a); w = sqrtf(z); flag = 1; }
    else { w = a; z = w * w; }
    float p = ((((4.2163199048E-2f * z + 2.4181311049E-2f) * z + 4.5470025998E-2f) * z + 7.4953002686E-2f) * z
               + 1.6666752422E-1f) * z * w + w;
    if (flag) { p = p + p; p = OM_PIO2F - p; }
    return sign * p;
}
static inline float om_acosf(float x)
{
    if (!(x >= -1.0f && x <= 1.0f)) return om_nan();
    if (x < -0.5f) return OM_PIF - 2.0f * om_asinf(sqrtf(0.5f * (1.0f + x)));
    if (x > 0.5f) return 2.0f * om_asinf(sqrtf(0.5f * (1.0f - x)));
    return OM_PIO2F - om_asinf(x);
}
static inline float om_atanf(float x)
{
    float sign = x < 0.0f ? -1.0f : 1.0f;
    float a = fabsf(x);
    if (a != a) return om_nan();
    float y;
    if (a > 2.414213562373095f) { y = OM_PIO2F; a = -(1.0f / a); }
    else if (a > 0.4142135623730950f) { y = OM_PIO4F; a = (a - 1.0f) / (a + 1.0f); }
    else y = 0.0f;
    float z = a * a;
    y = y + ((((8.05374449538e-2f * z - 1.38776856032E-1f) * z + 1.99777106478E-1f) * z - 3.33329491539E-1f) * z * a + a);
    return sign * y;
}
static inline float om_atan2f(float y, float x)
{
    if (x != x || y != y) return om_nan();
    int code = 0;
    if (x < 0.0f) code = 2;
    if (y < 0.0f) code |= 1;
    if (x == 0.0f) {
        if (code & 1) return -OM_PIO2F;
        if (y == 0.0f) return 0.0f;
        return OM_PIO2F;
    }
    if (y == 0.0f) return (code & 2) ? OM_PIF : 0.0f;
    float w = code == 2 ? OM_PIF : (code == 3 ? -OM_PIF : 0.0f);
    return w + om_atanf(y / x);
}
/* 2^n for integer n in [-126, 127] */
static inline float om_pow2i(int n) { return om_u2f((uint32_t)(n + 127) << 23); }
static inline float om_expf(float x)
{
    if (x != x) return om_nan();
    if (x > 88.72283905206835f) return om_inf();
    if (x < -87.0f) return 0.0f; /* flush: results below ~1.6e-38 are returned as 0 */
    float zf = floorf(1.44269504088896341f * x + 0.5f);
    float r = x - zf * 0.693359375f;
    r = r - zf * -2.12194440e-4f;
    float z = r * r;
    z = (((((1.9875691500E-4f * r + 1.3981999507E-3f) * r + 8.3334519073E-3f) * r + 4.1665795894E-2f) * r
          + 1.6666665459E-1f) * r + 5.0000001201E-1f) * z + r + 1.0f;
    int n = (int)zf; /* in [-126, 128] */
    if (n > 127) return (z * 2.0f) * om_pow2i(n - 1);
    return z * om_pow2i(n);
}
static inline float om_logf(float x)
{
    if (x != x || x < 0.0f) return om_nan();
    if (x == 0.0f) return -om_inf();
    if (x == om_inf()) return x;
    int e = 0;
    uint32_t u = om_f2u(x);
    if (u < 0x00800000u) { x = x * 16777216.0f; e = -24; u = om_f2u(x); } /* denormal */
    e += (int)(u >> 23) - 126;
    float m = om_u2f((u & 0x007fffffu) | 0x3f000000u); /* [0.5, 1) */
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
static inline float om_log10f(float x) { return om_logf(x) * 0.43429448190325176f; }
/* pow for the tone-map (reference: include/Render.cuh:350, base clamped to [0,1]);
 * defined for x >= 0 only. */
static inline float om_powf(float x, float y)
{
    if (x != x || y != y) return om_nan();
    if (x < 0.0f) return om_nan();
    if (x == 0.0f) return y > 0.0f ? 0.0f : (y == 0.0f ? 1.0f : om_inf());
    if (x == 1.0f) return 1.0f;
    return om_expf(y * om_logf(x));
}

#endif

// Exhaustive check on the GPU box: for a FIXED divisor b and every fp32 bit pattern x, do the short forms
//   r = 1.0f / b (once);  q0 = x * r;  q1 = fma(fma(-b, q0, x), r, q0);  [q2 = fma(fma(-b, q1, x), r, q1)]
// return the bits of the IEEE division x / b?  Prints mismatch counts per exponent field of x.
// build: hipcc -O3 -ffp-contract=off --offload-arch=gfx950 -fhip-fp32-correctly-rounded-divide-sqrt div_check.hip -o div_check
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void k_check(float b, float r, unsigned long long* per_exp1, unsigned long long* per_exp2)
{
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = tid; i < (1ull << 32); i += stride) {
        const float x = __uint_as_float((uint32_t)i);
        const float ref = x / b;
        const float q0 = x * r;
        const float q1 = __builtin_fmaf(__builtin_fmaf(-b, q0, x), r, q0);
        const float q2 = __builtin_fmaf(__builtin_fmaf(-b, q1, x), r, q1);
        const uint32_t ex = ((uint32_t)i >> 23) & 0xffu;
        if (!(__float_as_uint(ref) == __float_as_uint(q1) || (ref != ref && q1 != q1))) atomicAdd(&per_exp1[ex], 1ull);
        if (!(__float_as_uint(ref) == __float_as_uint(q2) || (ref != ref && q2 != q2))) atomicAdd(&per_exp2[ex], 1ull);
    }
}

int main(int argc, char** argv)
{
    unsigned long long *d1, *d2;
    (void)hipMalloc(&d1, 256 * 8); (void)hipMalloc(&d2, 256 * 8);
    for (int a = 1; a < argc; a++) {
        const float b = (float)atof(argv[a]);
        volatile float one = 1.0f;
        const float r = one / b;
        (void)hipMemset(d1, 0, 256 * 8); (void)hipMemset(d2, 0, 256 * 8);
        hipLaunchKernelGGL(k_check, dim3(256 * 32), dim3(256), 0, 0, b, r, d1, d2);
        if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
        std::vector<unsigned long long> e1(256), e2(256);
        (void)hipMemcpy(e1.data(), d1, 256 * 8, hipMemcpyDeviceToHost); (void)hipMemcpy(e2.data(), d2, 256 * 8, hipMemcpyDeviceToHost);
        unsigned long long t1 = 0, t2 = 0;
        int lo1 = 256, hi1 = -1, lo2 = 256, hi2 = -1;
        unsigned long long mid1 = 0, mid2 = 0;
        for (int i = 0; i < 256; i++) {
            t1 += e1[i]; t2 += e2[i];
            if (e1[i]) { if (i < lo1) lo1 = i; if (i > hi1) hi1 = i; }
            if (e2[i]) { if (i < lo2) lo2 = i; if (i > hi2) hi2 = i; }
            if (i >= 30 && i <= 225) { mid1 += e1[i]; mid2 += e2[i]; }
        }
        printf("b=%.9g  one step: %llu mismatches (exponent fields %d..%d, %llu in 30..225)   two steps: %llu (fields %d..%d, %llu in 30..225)\n",
               b, t1, lo1, hi1, mid1, t2, lo2, hi2, mid2);
        printf("   two-step mismatching exponent fields:");
        for (int i = 0; i < 256; i++) if (e2[i]) printf(" %d:%llu", i, e2[i]);
        printf("\n");
    }
    return 0;
}

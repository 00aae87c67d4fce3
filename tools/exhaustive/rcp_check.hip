// Exhaustive check on the GPU box: for every fp32 bit pattern x, does the short reciprocal
//   r0 = v_rcp_f32(x); e = fma(-x, r0, 1); r = fma(e, r0, r0)
// return the bits of the IEEE division 1.0f / x (the expansion hipcc emits with
// -fhip-fp32-correctly-rounded-divide-sqrt)?  Prints the mismatch count per exponent of x and a few examples.
// build: hipcc -O3 -ffp-contract=off --offload-arch=gfx950 -fhip-fp32-correctly-rounded-divide-sqrt rcp_check.hip -o rcp_check
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

__device__ __forceinline__ float rcp_short(float x)
{
    const float r0 = __builtin_amdgcn_rcpf(x);
    const float e = __builtin_fmaf(-x, r0, 1.0f);
    return __builtin_fmaf(e, r0, r0);
}

__global__ void k_check(unsigned long long* per_exp, unsigned long long* total, uint32_t* examples)
{
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t b = tid; b < (1ull << 32); b += stride) {
        const float x = __uint_as_float((uint32_t)b);
        const float ref = 1.0f / x;
        const float got = rcp_short(x);
        const bool same = __float_as_uint(ref) == __float_as_uint(got) || (ref != ref && got != got);
        if (!same) {
            const uint32_t ex = ((uint32_t)b >> 23) & 0xffu;
            atomicAdd(&per_exp[ex], 1ull);
            const unsigned long long n = atomicAdd(total, 1ull);
            if (n < 16) { examples[3 * n] = (uint32_t)b; examples[3 * n + 1] = __float_as_uint(ref); examples[3 * n + 2] = __float_as_uint(got); }
        }
    }
}

int main()
{
    unsigned long long *d_exp, *d_tot;
    uint32_t* d_ex;
    hipMalloc(&d_exp, 256 * 8); hipMalloc(&d_tot, 8); hipMalloc(&d_ex, 48 * 4);
    hipMemset(d_exp, 0, 256 * 8); hipMemset(d_tot, 0, 8); hipMemset(d_ex, 0, 48 * 4);
    hipLaunchKernelGGL(k_check, dim3(256 * 32), dim3(256), 0, 0, d_exp, d_tot, d_ex);
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
    std::vector<unsigned long long> e(256);
    unsigned long long tot = 0;
    uint32_t ex[48];
    hipMemcpy(e.data(), d_exp, 256 * 8, hipMemcpyDeviceToHost); hipMemcpy(&tot, d_tot, 8, hipMemcpyDeviceToHost); hipMemcpy(ex, d_ex, 48 * 4, hipMemcpyDeviceToHost);
    printf("mismatches: %llu of 2^32\n", tot);
    for (int i = 0; i < 256; i++) if (e[i]) printf("  exponent field %3d: %llu\n", i, e[i]);
    for (unsigned long long i = 0; i < tot && i < 16; i++) printf("  x=%08x ieee=%08x short=%08x\n", ex[3 * i], ex[3 * i + 1], ex[3 * i + 2]);
    return 0;
}

// Exhaustive check on the GPU box: is the short division
//     y = 1.0f / b (correctly rounded);  q0 = a * y;  r = fma(b, q0, -a);  q1 = fma(-r, y, q0)
// the IEEE quotient a / b for EVERY pair of fp32 mantissas?  (Markstein's correction step; the residual is negated so that a
// zero numerator keeps its sign.)  Scaling a and b by powers of two scales every intermediate exactly as long as nothing leaves
// the normal range, so the 2^23 x 2^23 pairs a, b in [1, 2) stand for all exponents of the guarded range
// (numerators 0 or 2^-100 <= |a| < 2^100, divisors 2^-40 <= b < 2^40).
//   hipcc -O3 -ffp-contract=off --offload-arch=gfx950 -fhip-fp32-correctly-rounded-divide-sqrt div_pair_check.hip -o div_pair_check
//   div_pair_check [first b mantissa] [count of b mantissas]      (default: all 2^23; prints progress every 2^18 divisors)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

__global__ __launch_bounds__(256) void k_check(uint32_t b_first, uint32_t b_count, unsigned long long* mism, uint32_t* example)
{
    // one divisor per wave-row: blockIdx.x selects the divisor, the 256 threads stride over the 2^23 numerators
    for (uint32_t bi = blockIdx.x; bi < b_count; bi += gridDim.x) {
        const float b = __uint_as_float(0x3f800000u | (b_first + bi));
        const float y = 1.0f / b;
        unsigned long long bad = 0;
        for (uint32_t am = threadIdx.x; am < (1u << 23); am += 256u) {
            const float a = __uint_as_float(0x3f800000u | am);
            const float ref = a / b;
            const float q0 = a * y;
            const float r = __builtin_fmaf(b, q0, -a);
            const float q1 = __builtin_fmaf(-r, y, q0);
            if (__float_as_uint(ref) != __float_as_uint(q1)) {
                if (bad == 0 && atomicAdd(&example[0], 1u) < 8u) { const uint32_t k = atomicAdd(&example[1], 2u); if (k < 16u) { example[2 + k] = am; example[3 + k] = b_first + bi; } }
                bad++;
            }
        }
        if (bad) atomicAdd(mism, bad);
    }
}

int main(int argc, char** argv)
{
    const uint32_t first = argc > 1 ? (uint32_t)strtoul(argv[1], nullptr, 0) : 0u;
    const uint32_t count = argc > 2 ? (uint32_t)strtoul(argv[2], nullptr, 0) : (1u << 23) - first;
    unsigned long long* d_m; uint32_t* d_ex;
    (void)hipMalloc(&d_m, 8); (void)hipMalloc(&d_ex, 32 * 4);
    (void)hipMemset(d_m, 0, 8); (void)hipMemset(d_ex, 0, 32 * 4);
    const uint32_t step = 1u << 18;
    unsigned long long total = 0;
    for (uint32_t done = 0; done < count; done += step) {
        const uint32_t n = count - done < step ? count - done : step;
        hipLaunchKernelGGL(k_check, dim3(256 * 16), dim3(256), 0, 0, first + done, n, d_m, d_ex);
        if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
        (void)hipMemcpy(&total, d_m, 8, hipMemcpyDeviceToHost);
        printf("divisor mantissas [%u, %u): %llu mismatching pairs so far\n", first, first + done + n, total);
        fflush(stdout);
    }
    uint32_t ex[32];
    (void)hipMemcpy(ex, d_ex, sizeof(ex), hipMemcpyDeviceToHost);
    for (uint32_t k = 0; k + 1 < 16 && k < ex[1]; k += 2) printf("  example: a mantissa 0x%06x, b mantissa 0x%06x\n", ex[2 + k], ex[3 + k]);
    printf("{\"pairs\": %llu, \"mismatches\": %llu}\n", (unsigned long long)count << 23, total);
    return total ? 2 : 0;
}

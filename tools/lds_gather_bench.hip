// tools/lds_gather_bench.hip -- what a gather of ray records from LDS costs by layout (SQ_LDS_BANK_CONFLICT is 50 % of the
// LDS-active cycles of k_mega3; is a component-major layout or an id swizzle the cure?).
// One wave per workgroup, 16 waves per CU as k_mega3, a pool of P records per wave; every iteration the 64 lanes read the records of 64
// DISTINCT pseudo-random ids (what a ring pop hands out) in one of the layouts:
//   aos128     float4 A[P], B[P]: two ds_read_b128 per lane (the production layout)
//   soa32      eight dword planes: eight ds_read_b32
//   soa64      four 8-byte planes: four ds_read_b64
//   aos128_swz as aos128 with id ^ (id >> 3) & 7 folded into the slot (an XOR swizzle)
//   linear     aos128 with ids = lane (the conflict-free floor)
// Reports cycles per iteration (s_memtime around the loop, per wave, averaged) -- the LDS pipe's cost of the gather with nothing else
// competing, which is what a layout can change.
// build: hipcc -O2 --offload-arch=gfx950 tools/lds_gather_bench.hip -o /tmp/lds_gather_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 2; } } while (0)

static constexpr int P = 152, ITERS = 4096;

template <int MODE>
__global__ __launch_bounds__(64) void k_gather(const unsigned char* ids, float* out, unsigned long long* cyc)
{
    __shared__ float4 rec[2 * P + 16];
    float* plane = (float*)rec;
    for (int i = threadIdx.x; i < 2 * P; i += 64) rec[i] = make_float4((float)i, 1.0f, 2.0f, 3.0f);
    __syncthreads();
    float acc = 0.0f;
    const unsigned char* my = ids + (size_t)(blockIdx.x & 63) * ITERS * 64;
    unsigned idv = my[threadIdx.x];
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < ITERS; it++) {
        unsigned id = MODE == 4 ? threadIdx.x : idv;
        if (MODE == 3) id = id ^ ((id >> 3) & 7u);
        if (MODE == 0 || MODE == 3 || MODE == 4) {
            const float4 a = rec[id], b = rec[P + id];
            acc += a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w;
        } else if (MODE == 1) {
#pragma unroll
            for (int c = 0; c < 8; c++) acc += plane[c * P + id];
        } else {
            const float2* p2 = (const float2*)rec;
#pragma unroll
            for (int c = 0; c < 4; c++) { const float2 v = p2[c * P + id]; acc += v.x + v.y; }
        }
        // the next iteration's ids: a dependent global load would dominate; permute the ids in registers instead (a fixed bijection on 0 .. P-1)
        idv = (idv * 37u + 11u + (unsigned)it) % (unsigned)P;
        asm volatile("" : "+v"(acc));
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + threadIdx.x] = acc;
    if (threadIdx.x == 0) atomicAdd(cyc, t1 - t0);
}

// (idv * 37 + 11 + it) % P keeps ids distinct across lanes only if they start distinct and the map is a bijection: x -> 37 x + c mod 152 is one (gcd(37, 152) = 1)

template <int MODE>
static int run(const char* name, const unsigned char* d_ids, float* d_out, unsigned long long* d_cyc, int blocks)
{
    CK(hipMemset(d_cyc, 0, 8));
    hipLaunchKernelGGL(k_gather<MODE>, dim3(blocks), dim3(64), 0, 0, d_ids, d_out, d_cyc);
    CK(hipDeviceSynchronize());
    unsigned long long c = 0;
    CK(hipMemcpy(&c, d_cyc, 8, hipMemcpyDeviceToHost));
    std::printf("{\"layout\": \"%s\", \"cycles_per_gather_per_wave\": %.1f}\n", name, (double)c / blocks / ITERS);
    return 0;
}

int main()
{
    hipDeviceProp_t pr;
    CK(hipGetDeviceProperties(&pr, 0));
    const int blocks = pr.multiProcessorCount * 16;
    std::vector<unsigned char> ids((size_t)64 * ITERS * 64);
    unsigned s = 12345u;
    for (size_t b = 0; b < 64; b++) { // distinct ids per wave: a shuffled 0 .. P-1, first 64
        unsigned char perm[P];
        for (int i = 0; i < P; i++) perm[i] = (unsigned char)i;
        for (int i = P - 1; i > 0; i--) { s = s * 1664525u + 1013904223u; const int j = (int)((s >> 8) % (unsigned)(i + 1)); const unsigned char t = perm[i]; perm[i] = perm[j]; perm[j] = t; }
        for (int l = 0; l < 64; l++) ids[b * ITERS * 64 + l] = perm[l];
    }
    unsigned char* d_ids; float* d_out; unsigned long long* d_cyc;
    CK(hipMalloc(&d_ids, ids.size())); CK(hipMalloc(&d_out, (size_t)blocks * 64 * 4)); CK(hipMalloc(&d_cyc, 8));
    CK(hipMemcpy(d_ids, ids.data(), ids.size(), hipMemcpyHostToDevice));
    std::printf("{\"device\": \"%s\", \"waves_per_cu\": 16, \"pool\": %d, \"note\": \"s_memtime cycles (100 MHz * ratio on this part: compare rows, not absolute)\"}\n", pr.gcnArchName, P);
    for (int rep = 0; rep < 2; rep++) {
        if (run<4>("linear ids, 2 x b128 (floor)", d_ids, d_out, d_cyc, blocks)) return 2;
        if (run<0>("aos128: 2 x ds_read_b128 (production)", d_ids, d_out, d_cyc, blocks)) return 2;
        if (run<3>("aos128 + xor swizzle", d_ids, d_out, d_cyc, blocks)) return 2;
        if (run<2>("soa64: 4 x ds_read_b64", d_ids, d_out, d_cyc, blocks)) return 2;
        if (run<1>("soa32: 8 x ds_read_b32 (component-major)", d_ids, d_out, d_cyc, blocks)) return 2;
    }
    return 0;
}

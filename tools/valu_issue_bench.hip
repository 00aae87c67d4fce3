// tools/valu_issue_bench.hip -- how many cycles does one SIMD of gfx950 need per wave64 vector instruction?
//
//   hipcc -O2 --offload-arch=gfx950 tools/valu_issue_bench.hip -o gpurun_out/valu_issue_bench && gpurun_out/valu_issue_bench
//
// VERDICT r01 asked for this number at the residency k_mega3 runs at (4 waves per SIMD), measured with INDEPENDENT
// instruction streams (the old sensitivity probe used a dependent chain and so measured latency).  Every wave runs
// `iters` iterations of an unrolled block of 32 instructions of one kind whose destinations rotate over 8 registers (no
// instruction reads a result younger than 8 instructions), timed with s_memtime; the kernel is launched with 1, 2, 4
// and 8 waves per SIMD on every CU.  Reported per kind and residency:
//     cycles_per_instr_per_wave = median wave's elapsed shader cycles / instructions it issued (what one wave sees)
//     cycles_per_instr_per_simd = grid span (s_memrealtime, first start to last end) x measured shader clock / (waves per SIMD x
//                                 instructions per wave): the SIMD's issue cost, the roofline constant
// plus a dependent chain of the same instruction (latency) and a mixed stream with one s_add_u32 per two vector instructions
// (does scalar issue share the port?).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

enum Kind { K_ADD = 0, K_FMA, K_PK_MUL, K_PK_ADD, K_MIN3, K_CNDMASK, K_MUL_LO, K_RCP, K_ADD_DEP, K_ADD_SALU, K_MUL, K_MAX, K_MOV, K_AND, K_CMP_CND, K_SQRT, K_FMA_K, K_COUNT };
static const char* kNames[K_COUNT] = {"v_add_f32", "v_fma_f32", "v_pk_mul_f32", "v_pk_add_f32", "v_max3_f32", "v_cndmask_b32", "v_mul_lo_u32",
                                      "v_rcp_f32", "v_add_f32 (dependent chain)", "2 v_add_f32 + 1 s_add_u32", "v_mul_f32", "v_max_f32", "v_mov_b32", "v_and_b32",
                                      "v_cmp_lt_f32 + v_cndmask_b32 (vcc)", "v_sqrt_f32", "v_fmac_f32 (VOP2)"};

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define BLOCK32(X) REP8(X) REP8(X) REP8(X) REP8(X)

template <int KIND>
__global__ __launch_bounds__(256) void k_issue(int iters, unsigned long long* cycles, float* sink)
{
    float r[8];
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f p[8];
    unsigned u[8];
    for (int i = 0; i < 8; i++) { r[i] = threadIdx.x * 0.001f + i; p[i].x = r[i]; p[i].y = r[i] + 0.5f; u[i] = threadIdx.x + i; }
    float a = 1.0001f, b = 0.9999f;
    v2f pa; pa.x = a; pa.y = b;
    unsigned sacc = 0;
    const unsigned long long cmask = 0x5555aaaa3333ccccull;
    __builtin_amdgcn_s_barrier();
    const unsigned long long w0 = __builtin_amdgcn_s_memrealtime(); // constant 100 MHz, the same on every CU
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();     // shader clock
    for (int it = 0; it < iters; it++) {
        if (KIND == K_ADD) {
#define X(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
            BLOCK32(X)
#undef X
        } else if (KIND == K_FMA) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));
            BLOCK32(X)
#undef X
        } else if (KIND == K_PK_MUL) {
#define X(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pa));
            BLOCK32(X)
#undef X
        } else if (KIND == K_PK_ADD) {
#define X(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pa));
            BLOCK32(X)
#undef X
        } else if (KIND == K_MIN3) {
#define X(i) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));
            BLOCK32(X)
#undef X
        } else if (KIND == K_CNDMASK) {
#define X(i) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "s"(cmask));
            BLOCK32(X)
#undef X
        } else if (KIND == K_MUL_LO) {
#define X(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
            BLOCK32(X)
#undef X
        } else if (KIND == K_RCP) {
#define X(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(r[i]));
            BLOCK32(X)
#undef X
        } else if (KIND == K_ADD_DEP) {
#define X(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(r[0]) : "v"(a));
            BLOCK32(X)
#undef X
        } else if (KIND == K_MUL) {
#define X(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
            BLOCK32(X)
#undef X
        } else if (KIND == K_MAX) {
#define X(i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
            BLOCK32(X)
#undef X
        } else if (KIND == K_MOV) {
#define X(i) asm volatile("v_mov_b32 %0, %1" : "+v"(r[i]) : "v"(r[(i + 3) & 7]));
            BLOCK32(X)
#undef X
        } else if (KIND == K_AND) {
#define X(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
            BLOCK32(X)
#undef X
        } else if (KIND == K_CMP_CND) { // 16 compare + 16 select per block
#define X(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(r[i]) : "v"(a) : "vcc");
            REP8(X) REP8(X)
#undef X
        } else if (KIND == K_SQRT) {
#define X(i) asm volatile("v_sqrt_f32 %0, %0" : "+v"(r[i]));
            BLOCK32(X)
#undef X
        } else if (KIND == K_FMA_K) {
#define X(i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));
            BLOCK32(X)
#undef X
        } else if (KIND == K_ADD_SALU) { // 32 vector + 16 scalar instructions per block
#define X(i) asm volatile("v_add_f32 %0, %0, %2\n\ts_add_u32 %1, %1, 1" : "+v"(r[i]), "+s"(sacc) : "v"(a) : "scc"); asm volatile("v_add_f32 %0, %0, %1" : "+v"(r[(i + 4) & 7]) : "v"(a));
            REP8(X) REP8(X)
#undef X
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long w1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.0f;
    for (int i = 0; i < 8; i++) s += r[i] + p[i].x + p[i].y + (float)u[i];
    if (s == 1.2345e-30f) sink[0] = s + (float)sacc;
    if ((threadIdx.x & 63) == 0) {
        const size_t wv = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
        cycles[wv * 3] = t1 - t0; cycles[wv * 3 + 1] = w0; cycles[wv * 3 + 2] = w1;
    }
}

struct Res { double cyc_wave, ms, ghz, span_us; };
template <int KIND> void run(int n_cus, int waves_per_simd, int iters, unsigned long long* d_cyc, float* d_sink, Res& R)
{
    const int blocks = n_cus * waves_per_simd; // 256 threads = one wave per SIMD of a CU
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_issue<KIND>, dim3(blocks), dim3(256), 0, 0, 16, d_cyc, d_sink); // warm-up
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_issue<KIND>, dim3(blocks), dim3(256), 0, 0, iters, d_cyc, d_sink);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float t = 0;
    CK(hipEventElapsedTime(&t, e0, e1));
    std::vector<unsigned long long> h((size_t)blocks * 4 * 3);
    CK(hipMemcpy(h.data(), d_cyc, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    std::vector<unsigned long long> cyc;
    std::vector<double> ghz;
    unsigned long long w_lo = ~0ull, w_hi = 0;
    for (size_t i = 0; i < (size_t)blocks * 4; i++) {
        cyc.push_back(h[i * 3]);
        w_lo = std::min(w_lo, h[i * 3 + 1]); w_hi = std::max(w_hi, h[i * 3 + 2]);
        if (h[i * 3 + 2] > h[i * 3 + 1]) ghz.push_back((double)h[i * 3] / ((double)(h[i * 3 + 2] - h[i * 3 + 1]) * 10.0)); // cycles per ns
    }
    std::sort(cyc.begin(), cyc.end());
    std::sort(ghz.begin(), ghz.end());
    R.cyc_wave = (double)cyc[cyc.size() / 2]; // median wave
    R.ghz = ghz.empty() ? 0.0 : ghz[ghz.size() / 2];
    R.span_us = (double)(w_hi - w_lo) / 100.0; // first wave's start to last wave's end
    R.ms = t;
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
}

int main(int argc, char** argv)
{
    const int iters = argc > 1 ? std::atoi(argv[1]) : 4000;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int n_cus = prop.multiProcessorCount;
    unsigned long long* d_cyc;
    float* d_sink;
    CK(hipMalloc(&d_cyc, (size_t)n_cus * 8 * 4 * 3 * sizeof(unsigned long long)));
    CK(hipMalloc(&d_sink, 4));
    std::printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz\": %d, \"iters\": %d, \"block\": 32, \"results\": [\n", prop.name, n_cus, prop.clockRate / 1000, iters);
    bool first = true;
    for (int kind = 0; kind < K_COUNT; kind++) {
        for (int w : {1, 2, 4, 8}) {
            Res R{};
            switch (kind) {
#define CASE(K) case K: run<K>(n_cus, w, iters, d_cyc, d_sink, R); break;
                CASE(K_ADD) CASE(K_FMA) CASE(K_PK_MUL) CASE(K_PK_ADD) CASE(K_MIN3) CASE(K_CNDMASK) CASE(K_MUL_LO) CASE(K_RCP) CASE(K_ADD_DEP) CASE(K_ADD_SALU) CASE(K_MUL) CASE(K_MAX) CASE(K_MOV) CASE(K_AND) CASE(K_CMP_CND) CASE(K_SQRT) CASE(K_FMA_K)
#undef CASE
            }
            const double n_vec = (double)iters * 32.0;
            // the SIMD's cost per instruction: the whole grid's span (first start to last end, constant clock) x the shader clock the
            // waves measured, over the instructions one SIMD issued (w waves x n_vec); waves of one SIMD are served oldest first, so a
            // single wave's own cycles say little at w > 2
            std::printf("%s {\"instr\": \"%s\", \"waves_per_simd\": %d, \"cycles_per_instr_per_wave\": %.3f, \"shader_ghz\": %.3f, \"span_us\": %.2f, "
                        "\"cycles_per_instr_per_simd\": %.3f, \"kernel_ms\": %.3f}",
                        first ? " " : ",\n ", kNames[kind], w, R.cyc_wave / n_vec, R.ghz, R.span_us, R.span_us * 1e3 * R.ghz / (n_vec * w), R.ms);
            std::fflush(stdout);
            first = false;
        }
    }
    std::printf("\n]}\n");
    return 0;
}

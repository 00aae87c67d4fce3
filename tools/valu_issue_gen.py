#!/usr/bin/env python3
"""Writes tools/valu_issue_ops.hip: the issue cost of one SIMD per wave64 vector instruction for every opcode (family) that the render
kernel executes, at the kernel's residency (4 waves per SIMD) -- the price list of the vector-issue roofline.

    python3 tools/valu_issue_gen.py && hipcc -O2 --offload-arch=gfx950 tools/valu_issue_ops.hip -o /tmp/valu_issue_ops && /tmp/valu_issue_ops > out.json

Method as tools/valu_issue_bench.hip (round 2): every wave runs `iters` iterations of an unrolled block of 32 instructions of ONE kind
whose destinations rotate over 8 registers (independent streams), 4 waves per SIMD on every CU; cost = grid span (s_memrealtime) x
measured shader clock / instructions per SIMD.  The output maps opcode -> cycles; tools/bbprof/report.py --costs prices the
kernel's dynamic opcode census (exact, from the basic-block profile) with it.
Operands: %0 r[i] (float, rotating), %1 a, %2 b (floats); u = unsigned rotating, q = 64-bit rotating, p = float2 rotating."""
import os

OPS = [
    # name,                asm,                                              operands
    ("v_mov_b32",          "v_mov_b32 %0, %1",                               'r "+v"(r[i]) : "v"(r[(i + 3) & 7])'),
    ("v_mov_b64",          "v_mov_b64 %0, %1",                               'q "+v"(q[i]) : "v"(q[(i + 3) & 7])'),
    ("v_add_f32",          "v_add_f32 %0, %0, %1",                           'r "+v"(r[i]) : "v"(a)'),
    ("v_sub_f32",          "v_sub_f32 %0, %0, %1",                           'r "+v"(r[i]) : "v"(a)'),
    ("v_mul_f32",          "v_mul_f32 %0, %0, %1",                           'r "+v"(r[i]) : "v"(a)'),
    ("v_fma_f32",          "v_fma_f32 %0, %0, %1, %2",                       'r "+v"(r[i]) : "v"(a), "v"(b)'),
    ("v_fmac_f32",         "v_fmac_f32 %0, %1, %2",                          'r "+v"(r[i]) : "v"(a), "v"(b)'),
    ("v_pk_mul_f32",       "v_pk_mul_f32 %0, %0, %1",                        'p "+v"(p[i]) : "v"(pa)'),
    ("v_pk_add_f32",       "v_pk_add_f32 %0, %0, %1",                        'p "+v"(p[i]) : "v"(pa)'),
    ("v_pk_fma_f32",       "v_pk_fma_f32 %0, %0, %1, %1",                    'p "+v"(p[i]) : "v"(pa)'),
    ("v_max3_f32",         "v_max3_f32 %0, %0, %1, %2",                      'r "+v"(r[i]) : "v"(a), "v"(b)'),
    ("v_min3_f32",         "v_min3_f32 %0, %0, %1, %2",                      'r "+v"(r[i]) : "v"(a), "v"(b)'),
    ("v_maximum3_f32",     "v_maximum3_f32 %0, %0, %1, %2",                  'r "+v"(r[i]) : "v"(a), "v"(b)'),
    ("v_minimum3_f32",     "v_minimum3_f32 %0, %0, %1, %2",                  'r "+v"(r[i]) : "v"(a), "v"(b)'),
    ("v_max_f32",          "v_max_f32 %0, %0, %1",                           'r "+v"(r[i]) : "v"(a)'),
    ("v_rcp_f32",          "v_rcp_f32 %0, %0",                               'r "+v"(r[i])'),
    ("v_sqrt_f32",         "v_sqrt_f32 %0, %0",                              'r "+v"(r[i])'),
    ("v_div_scale_f32",    "v_div_scale_f32 %0, vcc, %0, %1, %2",            'r "+v"(r[i]) : "v"(a), "v"(b) : "vcc"'),
    ("v_div_fmas_f32",     "v_div_fmas_f32 %0, %0, %1, %2",                  'r "+v"(r[i]) : "v"(a), "v"(b) : "vcc"'),
    ("v_div_fixup_f32",    "v_div_fixup_f32 %0, %0, %1, %2",                 'r "+v"(r[i]) : "v"(a), "v"(b)'),
    ("v_cvt_f32_u32",      "v_cvt_f32_u32 %0, %1",                           'r "+v"(r[i]) : "v"(u[i])'),
    ("v_add_u32",          "v_add_u32 %0, %0, %1",                           'u "+v"(u[i]) : "v"(u[(i + 1) & 7])'),
    ("v_sub_u32",          "v_sub_u32 %0, %0, %1",                           'u "+v"(u[i]) : "v"(u[(i + 1) & 7])'),
    ("v_and_b32",          "v_and_b32 %0, %0, %1",                           'u "+v"(u[i]) : "v"(u[(i + 1) & 7])'),
    ("v_or_b32",           "v_or_b32 %0, %0, %1",                            'u "+v"(u[i]) : "v"(u[(i + 1) & 7])'),
    ("v_xor_b32",          "v_xor_b32 %0, %0, %1",                           'u "+v"(u[i]) : "v"(u[(i + 1) & 7])'),
    ("v_not_b32",          "v_not_b32 %0, %0",                               'u "+v"(u[i])'),
    ("v_min_u32",          "v_min_u32 %0, %0, %1",                           'u "+v"(u[i]) : "v"(u[(i + 1) & 7])'),
    ("v_lshrrev_b32",      "v_lshrrev_b32 %0, 3, %0",                        'u "+v"(u[i])'),
    ("v_lshlrev_b32",      "v_lshlrev_b32 %0, 3, %0",                        'u "+v"(u[i])'),
    ("v_lshlrev_b64",      "v_lshlrev_b64 %0, 3, %0",                        'q "+v"(q[i])'),
    ("v_lshl_add_u32",     "v_lshl_add_u32 %0, %0, 2, %1",                   'u "+v"(u[i]) : "v"(u[(i + 1) & 7])'),
    ("v_lshl_add_u64",     "v_lshl_add_u64 %0, %0, 2, %1",                   'q "+v"(q[i]) : "v"(q[(i + 1) & 7])'),
    ("v_and_or_b32",       "v_and_or_b32 %0, %0, %1, %2",                    'u "+v"(u[i]) : "v"(u[(i + 1) & 7]), "v"(u[(i + 2) & 7])'),
    ("v_bitop3_b32",       "v_bitop3_b32 %0, %0, %1, %2 bitop3:0x26",        'u "+v"(u[i]) : "v"(u[(i + 1) & 7]), "v"(u[(i + 2) & 7])'),
    ("v_bfe_u32",          "v_bfe_u32 %0, %0, 3, 5",                         'u "+v"(u[i])'),
    ("v_bfe_i32",          "v_bfe_i32 %0, %0, 3, 5",                         'u "+v"(u[i])'),
    ("v_perm_b32",         "v_perm_b32 %0, %0, %1, %2",                      'u "+v"(u[i]) : "v"(u[(i + 1) & 7]), "v"(u[(i + 2) & 7])'),
    ("v_mad_i32_i24",      "v_mad_i32_i24 %0, %0, %1, %2",                   'u "+v"(u[i]) : "v"(u[(i + 1) & 7]), "v"(u[(i + 2) & 7])'),
    ("v_mad_u32_u24",      "v_mad_u32_u24 %0, %0, %1, %2",                   'u "+v"(u[i]) : "v"(u[(i + 1) & 7]), "v"(u[(i + 2) & 7])'),
    ("v_mul_i32_i24",      "v_mul_i32_i24 %0, %0, %1",                       'u "+v"(u[i]) : "v"(u[(i + 1) & 7])'),
    ("v_mul_lo_u32",       "v_mul_lo_u32 %0, %0, %1",                        'u "+v"(u[i]) : "v"(u[(i + 1) & 7])'),
    ("v_mul_hi_u32",       "v_mul_hi_u32 %0, %0, %1",                        'u "+v"(u[i]) : "v"(u[(i + 1) & 7])'),
    ("v_mad_u64_u32",      "v_mad_u64_u32 %0, vcc, %1, %2, %0",              'q "+v"(q[i]) : "v"(u[(i + 1) & 7]), "v"(u[(i + 2) & 7]) : "vcc"'),
    ("v_cndmask_b32_e32",  "v_cndmask_b32 %0, %0, %1, vcc",                  'r "+v"(r[i]) : "v"(a) : "vcc"'),
    ("v_cndmask_b32_e64",  "v_cndmask_b32 %0, %0, %1, %2",                   'r "+v"(r[i]) : "v"(a), "s"(cmask)'),
    ("v_cmp_f32_e32",      "v_cmp_lt_f32 vcc, %0, %1",                       'n : "v"(r[i]), "v"(a) : "vcc"'),
    ("v_cmp_f32_e64",      "v_cmp_lt_f32_e64 %0, %1, %2",                    's "=s"(sc) : "v"(r[i]), "v"(a)'),
    ("v_cmp_u32_e32",      "v_cmp_eq_u32 vcc, %0, %1",                       'n : "v"(u[i]), "v"(u[(i + 1) & 7]) : "vcc"'),
    ("v_cmp_u32_e64",      "v_cmp_eq_u32_e64 %0, %1, %2",                    's "=s"(sc) : "v"(u[i]), "v"(u[(i + 1) & 7])'),
    ("v_cmp_class_f32",    "v_cmp_class_f32 vcc, %0, %1",                    'n : "v"(r[i]), "v"(u[0]) : "vcc"'),
    ("v_mbcnt_lo_u32_b32", "v_mbcnt_lo_u32_b32 %0, %1, %0",                  'u "+v"(u[i]) : "s"(m32)'),
    ("v_mbcnt_hi_u32_b32", "v_mbcnt_hi_u32_b32 %0, %1, %0",                  'u "+v"(u[i]) : "s"(m32)'),
    ("v_readlane_b32",     "v_readlane_b32 %0, %1, 3",                       's32 "=s"(s32) : "v"(r[i])'),
    ("v_readfirstlane_b32", "v_readfirstlane_b32 %0, %1",                    's32 "=s"(s32) : "v"(r[i])'),
    ("v_writelane_b32",    "v_writelane_b32 %0, %1, 3",                      'r "+v"(r[i]) : "s"(m32)'),
]

HEAD = r'''// GENERATED by tools/valu_issue_gen.py -- do not edit.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define BLOCK32(X) REP8(X) REP8(X) REP8(X) REP8(X)
typedef float v2f __attribute__((ext_vector_type(2)));
#define PROLOG \
    float r[8]; v2f p[8]; unsigned u[8]; unsigned long long q[8]; \
    for (int i = 0; i < 8; i++) { r[i] = threadIdx.x * 0.001f + i + 1.0f; p[i].x = r[i]; p[i].y = r[i] + 0.5f; u[i] = threadIdx.x + i + 1; q[i] = (unsigned long long)u[i] * 77u; } \
    float a = 1.0001f, b = 0.9999f; v2f pa; pa.x = a; pa.y = b; \
    const unsigned long long cmask = 0x5555aaaa3333ccccull; unsigned long long sc = 0; unsigned m32 = 0x0f0f3355u; unsigned s32 = 0; \
    asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(r[0]), "v"(a) : "vcc"); \
    __builtin_amdgcn_s_barrier(); \
    const unsigned long long w0 = __builtin_amdgcn_s_memrealtime(); const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#define EPILOG \
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(); const unsigned long long w1 = __builtin_amdgcn_s_memrealtime(); \
    float s = 0.0f; for (int i = 0; i < 8; i++) s += r[i] + p[i].x + p[i].y + (float)u[i] + (float)q[i]; \
    if (s == 1.2345e-30f) sink[0] = s + (float)sc + (float)s32; \
    if ((threadIdx.x & 63) == 0) { const size_t wv = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; cycles[wv * 3] = t1 - t0; cycles[wv * 3 + 1] = w0; cycles[wv * 3 + 2] = w1; }
'''

TAIL = r'''
typedef void (*Kern)(int, unsigned long long*, float*);
struct Op { const char* name; Kern k; };
int main(int argc, char** argv)
{
    const int iters = argc > 1 ? std::atoi(argv[1]) : 3000, wps = argc > 2 ? std::atoi(argv[2]) : 4;
    hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
    const int n_cus = pr.multiProcessorCount, blocks = n_cus * wps;
    unsigned long long* d_cyc; float* d_sink;
    const size_t n_waves = (size_t)blocks * 4;
    CK(hipMalloc(&d_cyc, n_waves * 3 * sizeof(unsigned long long))); CK(hipMalloc(&d_sink, 64));
    std::vector<unsigned long long> h(n_waves * 3);
    std::printf("{\"device\": \"%s\", \"cus\": %d, \"waves_per_simd\": %d, \"iters\": %d, \"block\": 32, \"cycles_per_instr\": {", pr.name, n_cus, wps, iters);
    bool first = true;
    for (const Op& op : kOps) {
        hipLaunchKernelGGL(op.k, dim3(blocks), dim3(256), 0, 0, 16, d_cyc, d_sink);
        CK(hipDeviceSynchronize());
        hipLaunchKernelGGL(op.k, dim3(blocks), dim3(256), 0, 0, iters, d_cyc, d_sink);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h.data(), d_cyc, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        unsigned long long lo = ~0ull, hi = 0; std::vector<unsigned long long> cyc;
        for (size_t w = 0; w < n_waves; w++) { cyc.push_back(h[w * 3]); lo = std::min(lo, h[w * 3 + 1]); hi = std::max(hi, h[w * 3 + 2]); }
        std::nth_element(cyc.begin(), cyc.begin() + cyc.size() / 2, cyc.end());
        const double med = (double)cyc[cyc.size() / 2];
        // shader clock from the median wave: shader cycles it counted / real time it ran (100 MHz s_memrealtime)
        double ghz_sum = 0; for (size_t w = 0; w < n_waves; w++) ghz_sum += (double)h[w * 3] / ((double)(h[w * 3 + 2] - h[w * 3 + 1]) * 10.0);
        const double ghz = ghz_sum / n_waves;
        const double span_ns = (double)(hi - lo) * 10.0;
        const double instr_per_simd = (double)wps * iters * 32.0;
        std::printf("%s\"%s\": %.3f", first ? "" : ", ", op.name, span_ns * ghz / instr_per_simd);
        first = false;
        (void)med;
    }
    std::printf("}}\n");
    return 0;
}
'''


def main():
    out = [HEAD]
    for n, (name, asm, ops) in enumerate(OPS):
        kind, operands = ops.split(" ", 1)
        out.append("__global__ __launch_bounds__(256) void k_%d(int iters, unsigned long long* cycles, float* sink)\n{\n    PROLOG\n    for (int it = 0; it < iters; it++) {\n" % n)
        out.append('#define X(i) asm volatile("%s" : %s);\n        BLOCK32(X)\n#undef X\n    }\n    EPILOG\n}\n' % (asm, operands if not operands.startswith(":") else operands))
    out.append("typedef void (*KernT)(int, unsigned long long*, float*);\nstruct OpT { const char* name; KernT k; };\n")
    out.append("#define Op OpT\n#define Kern KernT\nstatic const OpT kOps[] = {\n" + "".join('    {"%s", k_%d},\n' % (name, n) for n, (name, _, _) in enumerate(OPS)) + "};\n")
    t = TAIL.replace("typedef void (*Kern)(int, unsigned long long*, float*);\nstruct Op { const char* name; Kern k; };\n", "")
    out.append(t)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "valu_issue_ops.hip")
    open(path, "w").write("".join(out))
    print(path)


if __name__ == "__main__":
    main()

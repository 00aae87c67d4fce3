// tools/init_probe.cpp -- what the first HIP calls of a process cost, with and without libcrt.so's code objects in the process (VERDICT r04
// item 7: `runtime_init_ms` 141 ms -- is it the library's 4.4 MB of kernels, fifty instantiations of k_mega3, or the runtime itself?).
// Times, in one fresh process: hipInit, hipSetDevice + hipFree(0) (context), the first hipMalloc, the first hipMemcpy (4 bytes up and down),
// the first kernel launch (a one-instruction kernel of THIS binary: its own module load).  Built twice:
//   hipcc -O2 --offload-arch=gfx950 tools/init_probe.cpp -o /tmp/init_probe_plain
//   hipcc -O2 --offload-arch=gfx950 -DWITH_LIBCRT tools/init_probe.cpp -Iinclude -Lcudaraytracing_amd/lib -lcrt -Wl,-rpath,$PWD/cudaraytracing_amd/lib -o /tmp/init_probe_crt
// The second also times the library's first kernel launch (crt_device_math: the module that holds it is loaded then).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#ifdef WITH_LIBCRT
#include "crt.h"
#endif
__global__ void k_nop(int* p) { if (p) *p = 1; }
static double ms_since(std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
int main()
{
    auto t = std::chrono::steady_clock::now();
    hipError_t e = hipInit(0);
    const double t_init = ms_since(t);
    t = std::chrono::steady_clock::now();
    e = hipSetDevice(0); e = hipFree(nullptr);
    const double t_ctx = ms_since(t);
    t = std::chrono::steady_clock::now();
    int* d = nullptr;
    e = hipMalloc((void**)&d, 4);
    const double t_malloc = ms_since(t);
    t = std::chrono::steady_clock::now();
    int h = 7;
    e = hipMemcpy(d, &h, 4, hipMemcpyHostToDevice); e = hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
    const double t_copy = ms_since(t);
    t = std::chrono::steady_clock::now();
    hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, 0, d);
    e = hipDeviceSynchronize();
    const double t_launch = ms_since(t);
    double t_crt = -1.0;
#ifdef WITH_LIBCRT
    t = std::chrono::steady_clock::now();
    float a = 0.5f, b = 0.0f, o = 0.0f;
    (void)crt_device_math(0, "sin", 1, &a, &b, &o);
    t_crt = ms_since(t);
#endif
    std::printf("{\"libcrt_linked\": %s, \"hipInit_ms\": %.2f, \"context_ms\": %.2f, \"first_malloc_ms\": %.2f, \"first_copies_ms\": %.2f, \"first_launch_own_kernel_ms\": %.2f, \"first_libcrt_kernel_ms\": %.2f, \"err\": %d}\n",
#ifdef WITH_LIBCRT
                "true",
#else
                "false",
#endif
                t_init, t_ctx, t_malloc, t_copy, t_launch, t_crt, (int)e);
    return 0;
}

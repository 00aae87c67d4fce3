// tools/copy_probe.cpp -- what the FIRST copies of each size cost in a process (round 6: the SAH build's first upload of 400 KB took
// 7.6 ms and its first download of 3 MB 8.7 ms in the first scene of a process, 0.12 / 0.65 ms in the second: whose cost is that?).
// After the runtime is up (hipInit, a 4-byte copy each way, a launch) it times, each on fresh buffers: H2D of 4 KB / 400 KB / 3 MB from
// pageable memory, twice each; D2H likewise; then the same from hipHostMalloc'ed memory.
//   hipcc -O2 --offload-arch=gfx950 tools/copy_probe.cpp -o /tmp/copy_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
static double ms_since(std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
__global__ void k_nop(int* p) { if (p) *p = 1; }
int main()
{
    hipError_t e = hipInit(0);
    e = hipSetDevice(0);
    int* d4 = nullptr;
    int h = 7;
    e = hipMalloc((void**)&d4, 4);
    e = hipMemcpy(d4, &h, 4, hipMemcpyHostToDevice); e = hipMemcpy(&h, d4, 4, hipMemcpyDeviceToHost);
    hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, 0, d4);
    e = hipDeviceSynchronize();
    const size_t sizes[3] = {4096, 400 * 1024, 3 * 1024 * 1024};
    std::printf("{");
    for (int dir = 0; dir < 2; dir++)
        for (int pinned = 0; pinned < 2; pinned++)
            for (size_t sz : sizes)
                for (int rep = 0; rep < 2; rep++) {
                    void* d = nullptr;
                    auto t = std::chrono::steady_clock::now();
                    e = hipMalloc(&d, sz);
                    const double t_malloc = ms_since(t);
                    std::vector<char> v;
                    char* hp = nullptr;
                    t = std::chrono::steady_clock::now();
                    if (pinned) e = hipHostMalloc((void**)&hp, sz, 0);
                    else { v.resize(sz); hp = v.data(); }
                    if (dir == 0) std::memset(hp, 1, sz);
                    const double t_host = ms_since(t);
                    t = std::chrono::steady_clock::now();
                    e = dir == 0 ? hipMemcpy(d, hp, sz, hipMemcpyHostToDevice) : hipMemcpy(hp, d, sz, hipMemcpyDeviceToHost);
                    const double t_copy = ms_since(t);
                    std::printf("\"%s_%s_%zu_%d\": [%.3f, %.3f, %.3f], ", dir ? "d2h" : "h2d", pinned ? "pinned" : "pageable", sz, rep, t_malloc, t_host, t_copy);
                    if (pinned) e = hipHostFree(hp);
                    e = hipFree(d);
                }
    std::printf("\"err\": %d}\n", (int)e);
    return 0;
}

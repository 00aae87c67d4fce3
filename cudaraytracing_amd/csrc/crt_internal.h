// cudaraytracing_amd/csrc/crt_internal.h -- between the translation units of libcrt.so's device layer: error plumbing, device
// buffers, and the launch entry points each kernel file exports to the host code in crt_render.hip.
#ifndef CRT_INTERNAL_H
#define CRT_INTERNAL_H
#include <cstdlib>
#include "crt_mega3.h"

#include <cstddef>
#include <vector>

extern "C" void crt_set_last_error_(const char* msg);

namespace crtk {

struct HipFail {
    hipError_t e;
    const char* what;
};
#define HIP_CHECK(call)                                          \
    do {                                                         \
        hipError_t e_ = (call);                                  \
        if (e_ != hipSuccess) throw HipFail{e_, #call};          \
    } while (0)

template <typename T> struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    void alloc(size_t count)
    {
        release();
        if (count == 0) count = 1;
        HIP_CHECK(hipMalloc((void**)&p, count * sizeof(T)));
        n = count;
        debug_fill();
    }
    // CRT_DEBUG_FILL=<byte>: every new allocation is filled with that byte (tests: a read of memory nobody wrote shows up whatever the
    // allocator hands out)
    void debug_fill()
    {
        static const char* e = std::getenv("CRT_DEBUG_FILL");
        if (!e || !*e) return;
        if (e[0] == 's') { HIP_CHECK(hipDeviceSynchronize()); return; }       // synchronise only
        if (e[0] == 'u' && !uncached) return;                                  // u<byte>: uncached allocations only
        if (e[0] == 'c' && uncached) return;                                   // c<byte>: cached allocations only
        const char* v = (e[0] == 'u' || e[0] == 'c') ? e + 1 : e;
        HIP_CHECK(hipMemset(p, std::atoi(v) & 0xff, n * sizeof(T)));
    }
    void ensure(size_t count)
    {
        if (n < count) alloc(count);
    }
    // Uncached device memory: every access goes to memory, past the L2 caches of the XCDs, which are not coherent with one another
    // inside a launch (the commit ring's buffers: written by one wave, read by another during the same launch).
    void ensure_uncached(size_t count)
    {
        if (n >= count && uncached) return;
        release();
        if (count == 0) count = 1;
        HIP_CHECK(hipExtMallocWithFlags((void**)&p, count * sizeof(T), hipDeviceMallocUncached));
        n = count;
        uncached = true;
        debug_fill();
    }
    bool uncached = false;
    void upload(const std::vector<T>& v)
    {
        alloc(v.size());
        if (!v.empty()) HIP_CHECK(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    }
    void release()
    {
        if (p) { (void)hipFree(p); p = nullptr; n = 0; uncached = false; }
    }
    ~DevBuf() { release(); }
};

const int kMaxBatch = 64;


// wavefront pipeline (crt_wavefront.hip)
#define REFILL_MIN 32
#define LEAF_MIN 24
#define SLOT_SHARDS 64
#define SLOT_STRIDE 32
void launch_pool_init(uint32_t blocks, hipStream_t st, const Pool& pool);
void launch_logic(bool lds_tables, uint32_t blocks, hipStream_t st, const LParams& P);
int trace_blocks_per_cu(int mode_id, size_t lds);
void launch_trace(int mode_id, const TParams& T, uint32_t blocks, size_t lds, hipStream_t st);
// frame and test kernels (crt_frame.hip)
void launch_accumulate(const AParams& A, hipStream_t st);
void launch_preview(const AParams& A, float scale, hipStream_t st);
void launch_fill_rays(const Pool& pool, uint32_t n, const float* o, const float* d, bool raw_dir, const float* limits);
void launch_math(int fn, uint32_t n, const float* a, const float* b, float* out);
void launch_philox(uint32_t n, const uint32_t* ctr, const uint32_t* key, uint32_t* out);
void launch_rcp_check(unsigned long long* counts);

} // namespace crtk
#endif

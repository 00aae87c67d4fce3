// cudaraytracing_amd/csrc/crt_multi.hip -- one process, N MI355X: the multi-device entry of libcrt.so.
//
// The reference selects device 0 and stops there (config_CUDA, src/main.cu:92-105; one kernel launch per
// frame, Render.cuh:435-440).  Pixels are independent and every random draw is keyed by the global pixel
// index, so the frame is sharded by interleaved 8x8 pixel tiles: tile t belongs to rank
// t % N.  A crt_multi owns one device replica of the scene per rank (crt_scene_create on each device),
// renders all shards concurrently (one host thread per device drives its own HIP stream), exchanges the
// compact tile buffers with ONE collective -- ncclAllGather over RCCL / xGMI, every link carries each
// peer's slice once -- and de-interleaves the gathered tiles into the row-major frame on rank 0.
//
// RCCL is bound at run time (dlopen "librccl.so.1"): libcrt.so stays loadable on a box without it, and a
// process that has already loaded PyTorch's RCCL shares that copy instead of mapping a second one.
// CRT_GATHER_COPY replaces the collective by peer copies into rank 0's buffer (hipMemcpyPeerAsync); it is
// also what runs when two ranks share one device (test configuration: RCCL refuses duplicate devices).
#include "../../include/crt.h"

#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

// The handful of RCCL declarations this file needs, restated so that libcrt.so builds on a box without the RCCL headers (the
// library itself is bound with dlopen below).  Values as in rccl.h / nccl.h: ncclSuccess = 0, ncclUint8 = 1.
extern "C" {
typedef struct ncclComm* ncclComm_t;
typedef int ncclResult_t;
typedef int ncclDataType_t;
enum { ncclSuccess = 0 };
enum { ncclUint8 = 1 };
ncclResult_t ncclGetVersion(int* version);
ncclResult_t ncclCommInitAll(ncclComm_t* comms, int ndev, const int* devlist);
ncclResult_t ncclCommDestroy(ncclComm_t comm);
ncclResult_t ncclCommCount(const ncclComm_t comm, int* count);
ncclResult_t ncclAllGather(const void* sendbuff, void* recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm, hipStream_t stream);
ncclResult_t ncclGroupStart(void);
ncclResult_t ncclGroupEnd(void);
const char* ncclGetErrorString(ncclResult_t result);
}

extern "C" void crt_set_last_error_(const char* msg);

namespace {

int mfail(int status, const std::string& msg)
{
    crt_set_last_error_(msg.c_str());
    return status;
}

struct HipErr {
    hipError_t e;
    const char* what;
};
#define MHIP(call)                                          \
    do {                                                    \
        hipError_t e_ = (call);                             \
        if (e_ != hipSuccess) throw HipErr{e_, #call};      \
    } while (0)

// ---- RCCL entry points, resolved once per process ----
struct Rccl {
    void* handle = nullptr;
    decltype(&ncclGetVersion) GetVersion = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string error;
    bool ok = false;
};

void rccl_load(Rccl& R);
Rccl& rccl()
{
    static Rccl R;
    static std::once_flag once;
    std::call_once(once, [] { rccl_load(R); });
    return R;
}
void rccl_load(Rccl& R)
{
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
        R.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (R.handle) break;
    }
    if (!R.handle) {
        const char* e = dlerror();
        R.error = std::string("cannot load librccl.so.1: ") + (e ? e : "?");
        return;
    }
#define SYM(field, name)                                                      \
    R.field = (decltype(R.field))dlsym(R.handle, name);                       \
    if (!R.field) { R.error = std::string("librccl: missing symbol ") + name; return; }
    SYM(GetVersion, "ncclGetVersion")
    SYM(CommInitAll, "ncclCommInitAll")
    SYM(CommDestroy, "ncclCommDestroy")
    SYM(CommCount, "ncclCommCount")
    SYM(AllGather, "ncclAllGather")
    SYM(GroupStart, "ncclGroupStart")
    SYM(GroupEnd, "ncclGroupEnd")
    SYM(GetErrorString, "ncclGetErrorString")
#undef SYM
    R.ok = true;
}

// Restores the calling thread's current device on every way out (the entry points walk over the ranks' devices).
struct DeviceGuard {
    int dev = -1;
    DeviceGuard() { if (hipGetDevice(&dev) != hipSuccess) dev = -1; }
    ~DeviceGuard() { if (dev >= 0) (void)hipSetDevice(dev); }
};

struct NcclErr {
    ncclResult_t r;
    const char* what;
};
#define MNCCL(call)                                         \
    do {                                                    \
        ncclResult_t r_ = (call);                           \
        if (r_ != ncclSuccess) throw NcclErr{r_, #call};    \
    } while (0)

// Gathered tile buffers -> row-major frame.  Rank r stores its k-th tile (global tile k * world + r) at slots
// [64 k, 64 k + 64) of its block; a block is `stride` bytes: RGB8 of `slots` pixels, padded to 16 B, then (optionally) the
// float mean of the same pixels.
__global__ __launch_bounds__(256) void k_untile(const uint8_t* gathered, uint32_t world, uint64_t stride, uint64_t mean_off, uint32_t width,
                                                uint32_t height, uint32_t tiles_x, uint8_t* out_rgb, float* out_mean)
{
    const uint32_t i = blockIdx.x * 32u + (threadIdx.x & 31u);
    const uint32_t j = blockIdx.y * 8u + (threadIdx.x >> 5);
    if (i >= width || j >= height) return;
    const uint32_t tile = (j >> 3) * tiles_x + (i >> 3);
    const uint32_t r = tile % world, k = tile / world;
    const uint64_t slot = (uint64_t)k * 64u + (j & 7u) * 8u + (i & 7u);
    const uint8_t* block = gathered + (uint64_t)r * stride;
    const uint64_t o = ((uint64_t)j * width + i) * 3u;
    out_rgb[o] = block[slot * 3]; out_rgb[o + 1] = block[slot * 3 + 1]; out_rgb[o + 2] = block[slot * 3 + 2];
    if (out_mean) {
        const float* m = (const float*)(block + mean_off) + slot * 3;
        out_mean[o] = m[0]; out_mean[o + 1] = m[1]; out_mean[o + 2] = m[2];
    }
}

struct Rank {
    int device = 0;
    crt_scene* scene = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;
    uint8_t* local = nullptr;    // this rank's block (stride bytes)
    uint8_t* gathered = nullptr; // world blocks (RCCL: every rank; COPY: rank 0 only)
    size_t local_cap = 0, gathered_cap = 0;
    ncclComm_t comm = nullptr;
};

} // namespace

struct crt_multi {
    std::vector<Rank> ranks;
    uint32_t gather = CRT_GATHER_COPY; // what runs
    int rccl_version = 0;
    int rccl_ranks = 0;
    uint8_t* frame = nullptr;  // rank 0: row-major RGB8
    float* mean = nullptr;     // rank 0: row-major mean
    size_t frame_cap = 0, mean_cap = 0;
    bool last_had_mean = false; // the last crt_multi_render wrote `mean`
    std::string fallback_reason; // CRT_GATHER_AUTO only: why RCCL was not used although the devices are distinct (empty: no fallback happened)
};

namespace {

void ensure(uint8_t*& p, size_t& cap, size_t bytes)
{
    if (cap >= bytes) return;
    if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
    MHIP(hipMalloc((void**)&p, bytes));
    cap = bytes;
}

void destroy(crt_multi* m)
{
    if (!m) return;
    Rccl& R = rccl();
    for (Rank& rk : m->ranks) {
        (void)hipSetDevice(rk.device);
        if (rk.comm && R.ok) (void)R.CommDestroy(rk.comm);
        if (rk.scene) (void)crt_scene_destroy(rk.scene);
        if (rk.local) (void)hipFree(rk.local);
        if (rk.gathered) (void)hipFree(rk.gathered);
        if (rk.done) (void)hipEventDestroy(rk.done);
        if (rk.stream) (void)hipStreamDestroy(rk.stream);
    }
    if (!m->ranks.empty()) (void)hipSetDevice(m->ranks[0].device);
    if (m->frame) (void)hipFree(m->frame);
    if (m->mean) (void)hipFree(m->mean);
    delete m;
}

} // namespace

extern "C" {

int crt_multi_create(const crt_scene_desc* desc, const int* devices, uint32_t n_devices, uint32_t gather, crt_multi** out)
{
    if (!out) return mfail(CRT_ERR_INVALID_ARG, "crt_multi_create: null output");
    *out = nullptr;
    if (!desc || !devices || n_devices == 0 || n_devices > 64) return mfail(CRT_ERR_INVALID_ARG, "crt_multi_create: need 1..64 devices");
    if (gather > CRT_GATHER_COPY) return mfail(CRT_ERR_INVALID_ARG, "crt_multi_create: unknown gather mode");
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) return mfail(CRT_ERR_NO_DEVICE, "crt_multi_create: no HIP device available");
    bool distinct = true;
    for (uint32_t a = 0; a < n_devices; a++) {
        if (devices[a] < 0 || devices[a] >= n_dev) return mfail(CRT_ERR_INVALID_ARG, "crt_multi_create: device index out of range");
        for (uint32_t b = 0; b < a; b++) distinct = distinct && devices[a] != devices[b];
    }
    if (gather == CRT_GATHER_RCCL && !distinct) return mfail(CRT_ERR_INVALID_ARG, "crt_multi_create: RCCL needs one rank per device (duplicate device index)");
    const uint32_t requested = gather;
    // Test hooks (the fallback below could not run on a one-GPU box).  CRT_TEST_RCCL_AUTO_ONE_DEVICE=1: AUTO attempts RCCL with a
    // single device too (a one-rank communicator).  CRT_TEST_RCCL_FAIL=load | init | count: the binding of librccl is treated as failed /
    // the communicator is made, then treated as failed and destroyed / ncclCommCount is treated as reporting another number of ranks.
    const char* t_one_ = std::getenv("CRT_TEST_RCCL_AUTO_ONE_DEVICE");
    const char* t_fail_ = std::getenv("CRT_TEST_RCCL_FAIL");
    const std::string inject = t_fail_ ? t_fail_ : "";
    const bool auto_one = t_one_ && t_one_[0] == '1';
    if (gather == CRT_GATHER_AUTO) gather = (distinct && (n_devices > 1 || auto_one)) ? CRT_GATHER_RCCL : CRT_GATHER_COPY;
    DeviceGuard guard;
    crt_multi* m = nullptr;
    try {
        m = new crt_multi();
        m->gather = gather;
        m->ranks.resize(n_devices);
        for (uint32_t r = 0; r < n_devices; r++) {
            Rank& rk = m->ranks[r];
            rk.device = devices[r];
            int rc = crt_scene_create(desc, rk.device, &rk.scene); // (leaves the device current)
            if (rc != CRT_OK) { destroy(m); return rc; }
            MHIP(hipSetDevice(rk.device));
            MHIP(hipStreamCreateWithFlags(&rk.stream, hipStreamNonBlocking));
            MHIP(hipEventCreateWithFlags(&rk.done, hipEventDisableTiming));
        }
        // CRT_GATHER_AUTO prefers RCCL but does not depend on it: if the library cannot be bound or the communicator cannot be made (a
        // node whose fabric RCCL refuses, a container without the IPC mode it needs), the frame is gathered by peer copies instead and
        // crt_multi_info::fallback_reason says why.  An explicit CRT_GATHER_RCCL fails loudly, as before.
        const bool auto_mode = requested == CRT_GATHER_AUTO;
        if (gather == CRT_GATHER_RCCL) {
            std::string why;
            Rccl& R = rccl();
            if (!R.ok) why = R.error;
            else if (inject == "load") why = "cannot load librccl.so.1: (injected by CRT_TEST_RCCL_FAIL=load)";
            else {
                std::vector<ncclComm_t> comms(n_devices, nullptr);
                std::vector<int> devs(devices, devices + n_devices);
                ncclResult_t rv = R.GetVersion(&m->rccl_version);
                if (rv == ncclSuccess) rv = R.CommInitAll(comms.data(), (int)n_devices, devs.data());
                if (rv == ncclSuccess) {
                    for (uint32_t r = 0; r < n_devices; r++) m->ranks[r].comm = comms[r];
                    rv = R.CommCount(comms[0], &m->rccl_ranks);
                }
                if (rv == ncclSuccess && inject == "count") m->rccl_ranks = (int)n_devices + 1;
                if (rv != ncclSuccess || inject == "init") {
                    why = std::string("ncclCommInitAll / ncclCommCount: ") + (rv != ncclSuccess ? R.GetErrorString(rv) : "(injected by CRT_TEST_RCCL_FAIL=init)");
                    for (uint32_t r = 0; r < n_devices; r++) {
                        if (m->ranks[r].comm) { (void)hipSetDevice(m->ranks[r].device); (void)R.CommDestroy(m->ranks[r].comm); m->ranks[r].comm = nullptr; }
                    }
                    (void)hipGetLastError();
                } else if (m->rccl_ranks != (int)n_devices) {
                    why = "ncclCommCount reports " + std::to_string(m->rccl_ranks) + " ranks for " + std::to_string(n_devices) + " devices";
                }
            }
            if (!why.empty()) {
                if (!auto_mode) { destroy(m); return mfail(R.ok ? CRT_ERR_HIP : CRT_ERR_UNSUPPORTED, "crt_multi_create: " + why); }
                for (uint32_t r = 0; r < n_devices; r++) {
                    if (m->ranks[r].comm) { (void)hipSetDevice(m->ranks[r].device); (void)R.CommDestroy(m->ranks[r].comm); m->ranks[r].comm = nullptr; }
                }
                m->fallback_reason = why;
                m->rccl_ranks = 0;
                gather = CRT_GATHER_COPY;
                m->gather = gather;
            }
        }
        if (gather == CRT_GATHER_COPY && distinct && n_devices > 1) {
            // rank 0 receives peer writes
            for (uint32_t r = 1; r < n_devices; r++) {
                int can = 0;
                MHIP(hipDeviceCanAccessPeer(&can, m->ranks[r].device, m->ranks[0].device));
                if (can) {
                    MHIP(hipSetDevice(m->ranks[r].device));
                    hipError_t e = hipDeviceEnablePeerAccess(m->ranks[0].device, 0);
                    if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) throw HipErr{e, "hipDeviceEnablePeerAccess"};
                    (void)hipGetLastError();
                }
            }
        }
        *out = m;
        return CRT_OK;
    } catch (const HipErr& f) {
        std::string msg = std::string("crt_multi_create: ") + f.what + ": " + hipGetErrorString(f.e);
        destroy(m);
        return mfail(CRT_ERR_HIP, msg);
    } catch (const NcclErr& f) {
        std::string msg = std::string("crt_multi_create: ") + f.what + ": " + rccl().GetErrorString(f.r);
        destroy(m);
        return mfail(CRT_ERR_HIP, msg);
    } catch (const std::bad_alloc&) {
        destroy(m);
        return mfail(CRT_ERR_OOM, "crt_multi_create: out of host memory");
    }
}

int crt_multi_destroy(crt_multi* m)
{
    destroy(m);
    return CRT_OK;
}

int crt_multi_render(crt_multi* m, const crt_camera* cam, const crt_params* prm, uint8_t* out_rgb, float* out_mean, crt_stats* stats,
                     crt_multi_info* info)
{
    if (!m || !cam || !prm) return mfail(CRT_ERR_INVALID_ARG, "crt_multi_render: null argument");
    if (prm->width == 0 || prm->height == 0) return mfail(CRT_ERR_INVALID_ARG, "crt_multi_render: width and height must be positive");
    if (out_mean && !out_rgb) return mfail(CRT_ERR_INVALID_ARG, "crt_multi_render: out_mean without out_rgb");
    const uint32_t world = (uint32_t)m->ranks.size();
    const bool want_mean = out_mean != nullptr;
    using clk = std::chrono::steady_clock;
    const auto t0 = clk::now();
    DeviceGuard guard;
    // an error must not leave work in flight on buffers the next call may free or reallocate
    auto drain = [&]() {
        for (Rank& rk : m->ranks)
            if (hipSetDevice(rk.device) == hipSuccess && rk.stream) (void)hipStreamSynchronize(rk.stream);
    };
    m->last_had_mean = false;
    try {
        uint64_t slots = 0;
        int rc = crt_shard_slots(prm->width, prm->height, 0, world, &slots);
        if (rc != CRT_OK) return rc;
        const uint64_t mean_off = (slots * 3 + 15) & ~15ull;
        const uint64_t stride = want_mean ? mean_off + slots * 12 : mean_off;
        for (uint32_t r = 0; r < world; r++) {
            Rank& rk = m->ranks[r];
            MHIP(hipSetDevice(rk.device));
            ensure(rk.local, rk.local_cap, stride);
            if (m->gather == CRT_GATHER_RCCL || r == 0) ensure(rk.gathered, rk.gathered_cap, stride * world);
        }
        MHIP(hipSetDevice(m->ranks[0].device));
        {
            size_t fb = (size_t)prm->width * prm->height * 3;
            ensure(m->frame, m->frame_cap, fb);
            if (want_mean) { // (through a local: the member must never hold a freed pointer if the allocation throws)
                uint8_t* p = (uint8_t*)m->mean;
                m->mean = nullptr;
                ensure(p, m->mean_cap, fb * 4);
                m->mean = (float*)p;
            }
        }
        // ---- every rank renders its tiles: one host thread per device, each on its own stream ----
        std::vector<int> rcs(world, CRT_OK);
        std::vector<std::string> errs(world);
        std::vector<crt_stats> st(world);
        auto work = [&](uint32_t r) {
            Rank& rk = m->ranks[r];
            crt_params p = *prm;
            p.rank = r; p.world = world;
            p.flags |= CRT_FLAG_TILED_OUTPUT;
            if (hipSetDevice(rk.device) != hipSuccess) { rcs[r] = CRT_ERR_HIP; errs[r] = "hipSetDevice failed"; return; }
            rcs[r] = crt_render_device(rk.scene, cam, &p, rk.local, want_mean ? rk.local + mean_off : nullptr, rk.stream, &st[r]);
            if (rcs[r] != CRT_OK) errs[r] = crt_last_error();
        };
        if (world == 1) work(0);
        else {
            std::vector<std::thread> th;
            for (uint32_t r = 0; r < world; r++) th.emplace_back(work, r);
            for (std::thread& t : th) t.join();
        }
        for (uint32_t r = 0; r < world; r++)
            if (rcs[r] != CRT_OK) { drain(); return mfail(rcs[r], "crt_multi_render: rank " + std::to_string(r) + ": " + errs[r]); }
        const auto t1 = clk::now();
        // ---- one exchange: all-gather of the compact tile blocks ----
        if (m->gather == CRT_GATHER_RCCL) {
            Rccl& R = rccl();
            MNCCL(R.GroupStart());
            for (uint32_t r = 0; r < world; r++) {
                Rank& rk = m->ranks[r];
                MNCCL(R.AllGather(rk.local, rk.gathered, (size_t)stride, ncclUint8, rk.comm, rk.stream));
            }
            MNCCL(R.GroupEnd());
        } else {
            Rank& r0 = m->ranks[0];
            for (uint32_t r = 0; r < world; r++) {
                Rank& rk = m->ranks[r];
                MHIP(hipSetDevice(rk.device));
                if (rk.device == r0.device) MHIP(hipMemcpyAsync(r0.gathered + (uint64_t)r * stride, rk.local, stride, hipMemcpyDeviceToDevice, rk.stream));
                else MHIP(hipMemcpyPeerAsync(r0.gathered + (uint64_t)r * stride, r0.device, rk.local, rk.device, stride, rk.stream));
                if (r > 0) MHIP(hipEventRecord(rk.done, rk.stream));
            }
            MHIP(hipSetDevice(r0.device));
            for (uint32_t r = 1; r < world; r++) MHIP(hipStreamWaitEvent(r0.stream, m->ranks[r].done, 0));
        }
        // ---- rank 0 de-interleaves the tiles into the frame ----
        {
            Rank& r0 = m->ranks[0];
            MHIP(hipSetDevice(r0.device));
            const uint32_t tiles_x = (prm->width + 7) / 8;
            dim3 grid((prm->width + 31) / 32, (prm->height + 7) / 8);
            hipLaunchKernelGGL(k_untile, grid, dim3(256), 0, r0.stream, r0.gathered, world, stride, mean_off, prm->width, prm->height, tiles_x, m->frame,
                               want_mean ? m->mean : nullptr);
            MHIP(hipGetLastError());
            if (out_rgb) MHIP(hipMemcpyAsync(out_rgb, m->frame, (size_t)prm->width * prm->height * 3, hipMemcpyDeviceToHost, r0.stream)); // Render.cuh:464
            if (out_mean) MHIP(hipMemcpyAsync(out_mean, m->mean, (size_t)prm->width * prm->height * 12, hipMemcpyDeviceToHost, r0.stream));
        }
        for (uint32_t r = 0; r < world; r++) {
            MHIP(hipSetDevice(m->ranks[r].device));
            MHIP(hipStreamSynchronize(m->ranks[r].stream));
        }
        const auto t2 = clk::now();
        m->last_had_mean = want_mean;
        if (stats) std::memcpy(stats, st.data(), sizeof(crt_stats) * world);
        if (info) {
            std::memset(info, 0, sizeof(*info));
            info->n_ranks = world;
            info->gather = m->gather;
            info->rccl_ranks = m->gather == CRT_GATHER_RCCL ? (uint32_t)m->rccl_ranks : 0u;
            info->rccl_version = m->rccl_version;
            info->render_ms = std::chrono::duration<float, std::milli>(t1 - t0).count();
            info->gather_ms = std::chrono::duration<float, std::milli>(t2 - t1).count();
            info->frame_ms = std::chrono::duration<float, std::milli>(t2 - t0).count();
            info->bytes_per_rank = stride;
            std::snprintf(info->fallback_reason, sizeof(info->fallback_reason), "%s", m->fallback_reason.c_str());
            for (uint32_t r = 0; r < world; r++) {
                info->rays += st[r].rays; info->paths += st[r].paths; info->rays_untraced += st[r].rays_untraced;
                info->max_kernel_ms = std::max(info->max_kernel_ms, st[r].kernel_ms);
            }
        }
        return CRT_OK;
    } catch (const HipErr& f) {
        drain();
        return mfail(CRT_ERR_HIP, std::string("crt_multi_render: ") + f.what + ": " + hipGetErrorString(f.e));
    } catch (const NcclErr& f) {
        drain();
        return mfail(CRT_ERR_HIP, std::string("crt_multi_render: ") + f.what + ": " + rccl().GetErrorString(f.r));
    } catch (const std::bad_alloc&) {
        drain();
        return mfail(CRT_ERR_OOM, "crt_multi_render: out of host memory");
    }
}

int crt_multi_frame_device(crt_multi* m, void** d_rgb, void** d_mean, int* device)
{
    if (!m) return mfail(CRT_ERR_INVALID_ARG, "crt_multi_frame_device: null argument");
    if (d_rgb) *d_rgb = m->frame;
    if (d_mean) *d_mean = m->last_had_mean ? m->mean : nullptr;
    if (device) *device = m->ranks[0].device;
    return CRT_OK;
}

} // extern "C"

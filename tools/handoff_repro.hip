// tools/handoff_repro.hip -- a reproducer for the stale kernel-to-kernel hand-off of DESIGN.md 6, outside the renderer.
//
// What the renderer saw: k_order_items writes a list with plain stores, k_mega3 -- next kernel but one on the same stream -- reads it
// with plain loads and finds, for a few entries, what an EARLIER kernel had left at the same address.  This program replays the
// pattern with a 4 MB buffer and counts stale reads per scenario, telling apart where the stale copy sat:
//   READ   every block reads "its" chunk (and a later CHECK with the same block -> chunk map reads it on the same XCD, often the same CU)
//   WRITE  new values, written by OTHER blocks (chunk map rotated by 1 block = another XCD)
//   CHECK  compares; map as READ (stale per-CU vector L1 or per-XCD L2), rotated by 8 blocks (same XCD, another CU: L2 only), rotated
//          by 3 (another XCD: nothing stale can be there)
// Scenarios: what stands between WRITE and CHECK (nothing / a small hipMemsetAsync / an event record / a host synchronisation), whether
// the buffer is freed and allocated again between READ and WRITE (cached again, or uncached in between), and whether CHECK loads with
// agent scope.  Each block records the XCC and CU it ran on (HW_REG_XCC_ID, HW_REG_HW_ID).
// build: hipcc -O2 --offload-arch=gfx950 tools/handoff_repro.hip -o /tmp/handoff_repro ; run: /tmp/handoff_repro [rounds]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); std::exit(2); } } while (0)

static constexpr int BLOCKS = 2048, THREADS = 64, PER_BLOCK = 512; // 2048 chunks of 512 words = 4 MB
static constexpr int N = BLOCKS * PER_BLOCK;

__device__ __forceinline__ unsigned where_am_i()
{
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    return (xcc & 0xfu) << 16 | (hw & 0xffffu); // HW_ID: wave 0-3, simd 4-5, pipe 6-7, cu 8-11, sh 12, se 13-15
}

__global__ void k_write(unsigned* buf, unsigned val, int rot)
{
    const int chunk = (blockIdx.x + rot) % BLOCKS;
    for (int i = threadIdx.x; i < PER_BLOCK; i += THREADS) buf[chunk * PER_BLOCK + i] = val + (unsigned)(chunk * PER_BLOCK + i);
}
__global__ void k_write_agent(unsigned* buf, unsigned val, int rot)
{
    const int chunk = (blockIdx.x + rot) % BLOCKS;
    for (int i = threadIdx.x; i < PER_BLOCK; i += THREADS)
        __hip_atomic_store(&buf[chunk * PER_BLOCK + i], val + (unsigned)(chunk * PER_BLOCK + i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ void k_read(const unsigned* buf, unsigned* sink, unsigned* where, int rot)
{
    const int chunk = (blockIdx.x + rot) % BLOCKS;
    unsigned s = 0;
    for (int i = threadIdx.x; i < PER_BLOCK; i += THREADS) s += buf[chunk * PER_BLOCK + i];
    if (s == 0x12345u) sink[0] = s;
    if (threadIdx.x == 0) where[blockIdx.x] = where_am_i();
}
template <bool AGENT>
__global__ void k_check(const unsigned* buf, unsigned val, int rot, unsigned* bad, unsigned* bad_old, unsigned old_val, unsigned* where)
{
    const int chunk = (blockIdx.x + rot) % BLOCKS;
    unsigned nb = 0, no = 0;
    for (int i = threadIdx.x; i < PER_BLOCK; i += THREADS) {
        const unsigned* p = &buf[chunk * PER_BLOCK + i];
        const unsigned v = AGENT ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *p;
        const unsigned want = val + (unsigned)(chunk * PER_BLOCK + i);
        if (v != want) { nb++; if (v == old_val + (unsigned)(chunk * PER_BLOCK + i)) no++; }
    }
    if (nb) { atomicAdd(&bad[0], nb); atomicAdd(&bad_old[0], no); atomicAdd(&bad[1], 1u); }
    if (threadIdx.x == 0) where[blockIdx.x] = where_am_i();
}
__global__ void k_touch(unsigned* p, int n) { for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = 0xdeadbeefu; }

enum Between { B_NONE, B_MEMSET, B_EVENT, B_SYNC, B_MEMSET_EVENT };
enum Realloc { R_KEEP, R_CACHED, R_VIA_UNCACHED };
static const char* between_name[] = {"nothing", "memsetAsync(256 B)", "eventRecord", "hipStreamSynchronize", "memsetAsync + eventRecord"};
static const char* realloc_name[] = {"same buffer", "free + malloc", "free + uncached malloc, touched, free + malloc"};

int main(int argc, char** argv)
{
    const int rounds = argc > 1 ? std::atoi(argv[1]) : 40;
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t ev;
    CK(hipEventCreate(&ev));
    unsigned *sink, *bad, *bad_old, *small, *where_r, *where_c;
    CK(hipMalloc(&sink, 64)); CK(hipMalloc(&bad, 64)); CK(hipMalloc(&bad_old, 64)); CK(hipMalloc(&small, 4096));
    CK(hipMalloc(&where_r, BLOCKS * 4)); CK(hipMalloc(&where_c, BLOCKS * 4));
    unsigned* buf = nullptr;
    CK(hipMalloc(&buf, (size_t)N * 4));
    hipDeviceProp_t pr;
    CK(hipGetDeviceProperties(&pr, 0));
    std::printf("{\"device\": \"%s\", \"cus\": %d, \"rounds\": %d}\n", pr.gcnArchName, pr.multiProcessorCount, rounds);
    const int check_rots[3] = {0, 8, 3};
    const char* rot_name[3] = {"same block map as READ (same XCD, CU by chance)", "blocks rotated by 8 (same XCD, other CU)", "blocks rotated by 3 (other XCD)"};
    unsigned val = 1000;
    for (int agent = 0; agent < 2; agent++)
        for (int ra = 0; ra < 3; ra++)
            for (int be = 0; be < 5; be++)
                for (int cr = 0; cr < 3; cr++) {
                    if (agent && cr != 0) continue;
                    unsigned long long stale_words = 0, stale_old = 0, stale_blocks = 0, moved = 0, same_cu = 0;
                    for (int r = 0; r < rounds; r++) {
                        const unsigned old_val = val;
                        val += 0x01000193u;
                        hipLaunchKernelGGL(k_write, dim3(BLOCKS), dim3(THREADS), 0, st, buf, old_val, 5);
                        hipLaunchKernelGGL(k_read, dim3(BLOCKS), dim3(THREADS), 0, st, buf, sink, where_r, 0);
                        if (ra != R_KEEP) {
                            CK(hipStreamSynchronize(st));
                            unsigned* was = buf;
                            CK(hipFree(buf));
                            if (ra == R_VIA_UNCACHED) {
                                unsigned* u = nullptr;
                                CK(hipExtMallocWithFlags((void**)&u, (size_t)N * 4, hipDeviceMallocUncached));
                                hipLaunchKernelGGL(k_touch, dim3(256), dim3(256), 0, st, u, N);
                                CK(hipStreamSynchronize(st));
                                CK(hipFree(u));
                            }
                            CK(hipMalloc(&buf, (size_t)N * 4));
                            if (buf != was) moved++;
                        }
                        CK(hipMemsetAsync(bad, 0, 8, st));
                        CK(hipMemsetAsync(bad_old, 0, 4, st));
                        if (agent) hipLaunchKernelGGL(k_write_agent, dim3(BLOCKS), dim3(THREADS), 0, st, buf, val, 1);
                        else hipLaunchKernelGGL(k_write, dim3(BLOCKS), dim3(THREADS), 0, st, buf, val, 1);
                        if (be == B_MEMSET || be == B_MEMSET_EVENT) CK(hipMemsetAsync(small, 0, 256, st));
                        if (be == B_EVENT || be == B_MEMSET_EVENT) CK(hipEventRecord(ev, st));
                        if (be == B_SYNC) CK(hipStreamSynchronize(st));
                        if (agent) hipLaunchKernelGGL(k_check<true>, dim3(BLOCKS), dim3(THREADS), 0, st, buf, val, check_rots[cr], bad, bad_old, old_val, where_c);
                        else hipLaunchKernelGGL(k_check<false>, dim3(BLOCKS), dim3(THREADS), 0, st, buf, val, check_rots[cr], bad, bad_old, old_val, where_c);
                        CK(hipGetLastError());
                        unsigned h[2], ho;
                        CK(hipMemcpyAsync(h, bad, 8, hipMemcpyDeviceToHost, st));
                        CK(hipMemcpyAsync(&ho, bad_old, 4, hipMemcpyDeviceToHost, st));
                        CK(hipStreamSynchronize(st));
                        stale_words += h[0]; stale_blocks += h[1]; stale_old += ho;
                        if (r == 0) {
                            std::vector<unsigned> wr(BLOCKS), wc(BLOCKS);
                            CK(hipMemcpy(wr.data(), where_r, BLOCKS * 4, hipMemcpyDeviceToHost));
                            CK(hipMemcpy(wc.data(), where_c, BLOCKS * 4, hipMemcpyDeviceToHost));
                            for (int b = 0; b < BLOCKS; b++) { // the block that CHECKs chunk c and the block that READ it
                                const int c = (b + check_rots[cr]) % BLOCKS;
                                const unsigned a = wc[b], o = wr[c];
                                if ((a >> 16) == (o >> 16) && ((a >> 8) & 0xffu) == ((o >> 8) & 0xffu)) same_cu++;
                            }
                        }
                    }
                    std::printf("{\"check_loads\": \"%s\", \"buffer\": \"%s\", \"between_write_and_check\": \"%s\", \"check\": \"%s\", \"stale_words\": %llu, \"of_them_the_old_value\": %llu, "
                                "\"stale_blocks\": %llu, \"words_checked\": %llu, \"reallocs_that_moved\": %llu, \"chunks_checked_on_the_cu_that_read_them\": %llu}\n",
                                agent ? "agent scope" : "plain", realloc_name[ra], between_name[be], rot_name[cr], stale_words, stale_old, stale_blocks,
                                (unsigned long long)rounds * N, moved, same_cu);
                    std::fflush(stdout);
                }
    return 0;
}

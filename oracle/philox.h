/*
 * oracle/philox.h -- TEST INFRASTRUCTURE (CPU oracle), not product code.
 *
 * Counter-based RNG that replaces the reference's cuRAND XORWOW stream
 * (reference: include/Global.h:52-55,106-109, include/Render.cuh:340-341 --
 * seeded from clock(), hence unreproducible; cuRAND itself is an unpinned CUDA
 * toolkit dependency that is absent from /root/reference).
 *
 * Algorithm: Philox4x32-10 exactly as published (Salmon, Moraes, Dror, Shaw,
 * "Parallel random numbers: as easy as 1, 2, 3", SC'11; Random123 known-answer
 * vectors are checked in tests/test_oracle_rng.py).
 *
 * Addressing contract shared by oracle and HIP kernel (DESIGN.md "RNG"):
 *   key     = { pixel_index, seed_lo }
 *   counter = { sample_k, depth | (purpose << 16), idx, seed_hi }
 *   purpose 0 camera jitter   : out[0] -> x jitter, out[1] -> y jitter      (Render.cuh:344-345)
 *   purpose 1 bounce at depth : out[0] -> RR, out[1] -> x_1, out[2] -> x_2  (Render.cuh:216, Global.h:59-60)
 *   purpose 2 NEE at depth    : idx = light*lsn + j; out[0] -> raw u32 triangle pick,
 *                               out[1] -> alpha, out[2] -> beta             (DeviceLights.cuh:35, DeviceTriangle.cuh:69-70)
 *   purpose 3 probe at depth  : out[0] -> eta_1, out[1] -> eta_2            (Global.h:70-71)
 * Every draw the reference takes from its sequential stream is therefore an
 * independently addressed uniform; the mapping is distribution-neutral and
 * makes results independent of evaluation order.
 * Uniform conversion follows curand_uniform's documented (0,1] mapping:
 *   u = x * 2^-32 + 2^-33 evaluated in float.
 */
#ifndef ORACLE_PHILOX_H
#define ORACLE_PHILOX_H
#include <stdint.h>

enum { ORC_RNG_JITTER = 0, ORC_RNG_BOUNCE = 1, ORC_RNG_NEE = 2, ORC_RNG_PROBE = 3 };

static inline void orc_philox4x32_10(const uint32_t ctr_in[4], const uint32_t key_in[2], uint32_t out[4])
{
    uint32_t c0 = ctr_in[0], c1 = ctr_in[1], c2 = ctr_in[2], c3 = ctr_in[3];
    uint32_t k0 = key_in[0], k1 = key_in[1];
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

static inline void orc_draw(uint64_t seed, uint32_t pixel_index, uint32_t sample_k, uint32_t depth,
                            uint32_t purpose, uint32_t idx, uint32_t out[4])
{
    uint32_t ctr[4] = { sample_k, depth | (purpose << 16), idx, (uint32_t)(seed >> 32) };
    uint32_t key[2] = { pixel_index, (uint32_t)seed };
    orc_philox4x32_10(ctr, key, out);
}

/* curand_uniform mapping: (0, 1] */
static inline float orc_uniform(uint32_t x)
{
    return (float)x * 2.3283064365386963e-10f + 1.1641532182693481e-10f;
}
#endif

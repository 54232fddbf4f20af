// philox.h -- the device-side noise source (Philox4x32-10, Salmon et al. 2011) and its draw conventions, shared by
// the stand-alone fill kernel (misc.hip: mvae_philox_fill / mvae_randn / mvae_bernoulli) and the PoE forward
// (poe.hip: mvae_poe_fwd_draw), which must produce the SAME value for element i of launch index L:
//   counter = (i / 4, L), key = seed;  standard normals: Box-Muller on the pairs (r0, r1) -> elements 4g, 4g + 1 and
//   (r2, r3) -> 4g + 2, 4g + 3.
// The reference draws eps = std.data.new(size).normal_() on the tensor's own generator (mnist/model.py:29-33);
// parity runs feed host-drawn noise instead (engine.set_noise), this is the throughput path.
#pragma once
#include "common.h"

__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
    const uint32_t hi0 = __umulhi(M0, c[0]), lo0 = M0 * c[0];
    const uint32_t hi1 = __umulhi(M1, c[2]), lo1 = M1 * c[2];
    const uint32_t n0 = hi1 ^ c[1] ^ k0, n1 = lo1, n2 = hi0 ^ c[3] ^ k1, n3 = lo0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}

__device__ __forceinline__ void philox4x32_10(uint64_t ctr_lo, uint64_t ctr_hi, uint64_t seed, uint32_t (&out)[4]) {
    uint32_t c[4] = {(uint32_t)ctr_lo, (uint32_t)(ctr_lo >> 32), (uint32_t)ctr_hi, (uint32_t)(ctr_hi >> 32)};
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        philox_round(c, k0, k1);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = c[3];
}

__device__ __forceinline__ float u01(uint32_t x) { return ((float)(x >> 8) + 0.5f) * (1.0f / 16777216.0f); }

// two independent standard normals from two 32-bit draws
__device__ __forceinline__ void box_muller(uint32_t a, uint32_t b, float &n0, float &n1) {
    const float rad = sqrtf(-2.0f * logf(u01(a)));
    float sn, cs;
    sincosf(6.2831853071795864f * u01(b), &sn, &cs);
    n0 = rad * cs; n1 = rad * sn;
}

// element i of the standard-normal stream of launch index `launch` -- what philox_fill_kernel writes to out[i]
__device__ __forceinline__ float philox_normal_at(size_t i, uint64_t launch, uint64_t seed) {
    uint32_t r[4];
    philox4x32_10(i >> 2, launch, seed, r);
    const bool second = (i & 2) != 0;
    float n0, n1;
    box_muller(second ? r[2] : r[0], second ? r[3] : r[1], n0, n1);
    return (i & 1) ? n1 : n0;
}

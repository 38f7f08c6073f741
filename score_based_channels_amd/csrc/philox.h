// Counter-based random numbers shared by the Langevin kernels (ops.hip) and the training perturbation (train.hip).
#pragma once
#include "common.h"

namespace sbc {

// Philox4x32-10 (Salmon et al. 2011), counter (c0..c3), key (k0, k1)
__device__ __forceinline__ uint4 philox4x32(uint4 c, uint2 k) {
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c.x), lo0 = 0xD2511F53u * c.x;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c.z), lo1 = 0xCD9E8D57u * c.z;
        c = make_uint4(hi1 ^ c.y ^ k.x, lo1, hi0 ^ c.w ^ k.y, lo0);
        k.x += 0x9E3779B9u;
        k.y += 0xBB67AE85u;
    }
    return c;
}

// two independent N(0,1) draws from counter (elem, step, stream) and key `seed`: Box-Muller on two uniforms in (0, 1)
__device__ __forceinline__ float2 normal_pair(uint64_t seed, int64_t stream, int step, int elem) {
    const uint4 r = philox4x32(make_uint4((uint32_t)elem, (uint32_t)step, (uint32_t)stream, (uint32_t)((uint64_t)stream >> 32)),
                               make_uint2((uint32_t)seed, (uint32_t)(seed >> 32)));
    const float u1 = ((float)(r.x >> 8) + 0.5f) * (1.f / 16777216.f);
    const float u2 = ((float)(r.y >> 8) + 0.5f) * (1.f / 16777216.f);
    const float rad = sqrtf(-2.f * logf(u1));
    float s, c;
    sincosf(6.283185307179586f * u2, &s, &c);
    return make_float2(rad * c, rad * s);
}

// CN(0,1) draws of the sampling loop (test_score.py:115,124,160-161: randn_like of a complex tensor = re, im ~ N(0, 1/2)).
// Stream definition (restated on the host by oracle/ald_oracle.py::device_complex_normal): elements 2q and 2q + 1 of trajectory
// `traj` at step `step` share ONE Philox4x32-10 block, counter (q, step, traj_lo, traj_hi), key = seed; element 2q takes words
// (x, y), element 2q + 1 words (z, w); per element u1 = ((a >> 8) + 0.5) / 2^24, u2 = ((b >> 8) + 0.5) / 2^24 and
// (re, im) = sqrt(-ln u1) * (cos 2 pi u2, sin 2 pi u2)   [Box-Muller with the 1/sqrt(2) folded in].
__device__ __forceinline__ float2 box_muller_half(uint32_t a, uint32_t b) {
    const float u1 = ((float)(a >> 8) + 0.5f) * (1.f / 16777216.f);   // (0, 1)
    const float u2 = ((float)(b >> 8) + 0.5f) * (1.f / 16777216.f);
    const float rad = sqrtf(-logf(u1));                                // sqrt(-2 ln u1) * sqrt(1/2)
    float s, c;
    sincosf(6.283185307179586f * u2, &s, &c);
    return make_float2(rad * c, rad * s);
}
__device__ __forceinline__ uint4 noise_block(uint64_t seed, int64_t traj, int step, int pair) {
    return philox4x32(make_uint4((uint32_t)pair, (uint32_t)step, (uint32_t)traj, (uint32_t)((uint64_t)traj >> 32)),
                      make_uint2((uint32_t)seed, (uint32_t)(seed >> 32)));
}
// both elements of pair q (one Philox block)
__device__ __forceinline__ void complex_normal_pair(uint64_t seed, int64_t traj, int step, int q, float2& n0, float2& n1) {
    const uint4 r = noise_block(seed, traj, step, q);
    n0 = box_muller_half(r.x, r.y);
    n1 = box_muller_half(r.z, r.w);
}
// one element (kernels whose threads do not own adjacent elements)
__device__ __forceinline__ float2 complex_normal(uint64_t seed, int64_t traj, int step, int elem) {
    const uint4 r = noise_block(seed, traj, step, elem >> 1);
    return (elem & 1) ? box_muller_half(r.z, r.w) : box_muller_half(r.x, r.y);
}

}  // namespace sbc

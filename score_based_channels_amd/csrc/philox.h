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

}  // namespace sbc

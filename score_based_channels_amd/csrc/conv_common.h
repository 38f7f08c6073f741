// Shared between the direct (conv_mfma.hip) and Winograd (conv_wino.hip) convolution kernels.
#pragma once
#include "tile.h"

namespace sbc {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ConvParams {
    const float* __restrict__ in;
    float* __restrict__ out;
    const float4* __restrict__ wpk;
    const float* __restrict__ bias;
    const float* __restrict__ stats;
    const float* __restrict__ res1;
    const float* __restrict__ res2;
    const float* __restrict__ up;
    int B, H, W, dil, flags, up_h, up_w, total_px;
    int hsh, wsh;         // log2(H), log2(W) for the power-of-two builds
};


// Winograd F(2x2,3x3) path (conv_wino.hip): returns SBC_OK after launching, or 1 when the shape is not eligible
// (the caller then uses the direct kernel).  `p.wpk` must point at the Winograd-packed weights.
int launch_conv_wino(const ConvParams& p, int cin, int cout, hipStream_t stream, bool dry);

}  // namespace sbc

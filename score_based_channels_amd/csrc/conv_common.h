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
    int plane;            // conv_x3: 16-bit elements per LDS plane
    int stats_off;        // conv_wx3: float offset of the statistics copy in LDS
    int top;              // op.tag == 1: a full-resolution ngf -> ngf layer (own kernel symbol for per-kernel profiles)
    // SBC_EPI_MOMENTS_OUT (tile.h): the output's tile moments [B][HW/128][COUT][2], for SBC_OP_INORM_STATS + SBC_PRO_NORM_MOMENTS
    float* __restrict__ pm_out;
    // conv_mode f16x2: per-device word the kernels OR a 1 into when a staged activation leaves the fp16 range (tile.h)
    unsigned* __restrict__ range_flag;
    // sbc_f16x2_calibrate: the layer's amax slot (max |x| of everything staged), NULL in ordinary launches
    float* __restrict__ calib;
};

// Trailer of the f16x2 weight forms (sbc_pack_conv_weight_f16x2 / _winograd_f16x2): one 16-byte record behind the last
// fragment, (act_scale, descale, 0, 0): activations are staged as x * act_scale, accumulators leave as acc * descale
// (descale = 1 / (act_scale * weight_scale), all powers of two).
__device__ __forceinline__ float4 f16x2_trailer(const float4* wpk, int n_frag16) {   // n_frag16: 16-byte fragments per lane slot
    return wpk[(size_t)n_frag16 * 64];
}


// Winograd F(2x2,3x3) path (conv_wino.hip): returns SBC_OK after launching, or 1 when the shape is not eligible
// (the caller then uses the direct kernel).  `p.wpk` must point at the Winograd-packed weights.
int launch_conv_wino(const ConvParams& p, int cin, int cout, hipStream_t stream, bool dry);

// Split-bf16 path (conv_x3.hip): fp32 operands as three bf16 terms each, six bf16 MFMAs per fp32 product block.
// `p.wpk` must point at sbc_pack_conv_weight_split weights.
int launch_conv_x3(const ConvParams& p, int cin, int cout, int ksize, hipStream_t stream, bool dry);

// Winograd F(2x2,3x3) with split-bf16 products (conv_wx3.hip): SBC_OK after launching, 1 when the shape is not eligible.
// `p.wpk` must point at sbc_pack_conv_weight_winograd_split weights.
int launch_conv_wx3(const ConvParams& p, int cin, int cout, hipStream_t stream, bool dry);

// 64 -> 64 channels, 3x3, images 8 / 4 / 2 pixels wide, conv_mode f16x2 (conv_dp.hip: direct persistent kernel with the filter
// fragments resident in registers): SBC_OK after launching, 1 when the layer is not eligible (-> conv_wx3), < 0 on error.
int launch_conv_dp(const sbc_op& op, unsigned* range_flag, hipStream_t stream, bool dry);

// Experiment (tools/experiments/conv_wp.hip, built only with -DSBC_WITH_WP): Winograd F(2x2,3x3) 64 -> 64 with the transformed
// filter resident in registers (conv_mode f16x2): SBC_OK after launching, 1 when the layer is not eligible (-> conv_wx3).
int launch_conv_wp(const ConvParams& p, int cin, int cout, hipStream_t stream, bool dry);

}  // namespace sbc

// Parameters of the fused RCU / CRP kernels (conv_pair.hip).
#pragma once
#include "conv_common.h"

namespace sbc {

struct PairParams {
    const float* __restrict__ in;
    float* __restrict__ out;
    const uint4* __restrict__ w1;       // sbc_pack_conv_weight_f16x2 / _f16 layout of conv1 (32 -> 32, 3x3)
    const uint4* __restrict__ w2;
    unsigned* __restrict__ range_flag;
    float* __restrict__ calib;          // sbc_f16x2_calibrate: two amax slots (conv1's input, the intermediate), else NULL
    const float* __restrict__ res1;     // conv_pool_kernel: residual operands of the CONV epilogue (or NULL)
    const float* __restrict__ res2;
    int flags;                          // conv_pool_kernel: SBC_PRO_ELU, SBC_EPI_RES1_ELU
    int B, H, ntiles, tiles_per_sample, wgs_per_xcd, tiles_per_xcd;
    unsigned long long* dbg;            // SBC_PAIR_TIMING builds: per-phase cycle sums of wave 0 of every workgroup
};

// (tools/experiments/conv_pair32.hip -- the RCU block on v_mfma_f32_32x32x16_f16 with the vector work between the matrix instructions,
// round 6: correct, slower; DESIGN.md section 9 -- declares its launcher against this struct)
int launch_pair32(const PairParams& p0, hipStream_t stream, bool dry);

}  // namespace sbc

// Shared declarations of libsbc_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/sbc_hip.h"

namespace sbc {

// thread-local error text behind sbc_last_error()
void set_error(const char* fmt, ...);

#define SBC_CHECK_HIP(expr)                                                                   \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            ::sbc::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
                             __LINE__);                                                       \
            return SBC_ERR_HIP;                                                               \
        }                                                                                     \
    } while (0)

#define SBC_REQUIRE(cond, ...)            \
    do {                                  \
        if (!(cond)) {                    \
            ::sbc::set_error(__VA_ARGS__); \
            return SBC_ERR_INVALID;       \
        }                                 \
    } while (0)

// Raise a kernel's dynamic-LDS limit to at least `bytes` on the CURRENT device.  hipFuncSetAttribute is per device, so the
// high-water mark is cached per (device, kernel) behind a mutex: safe for a process that drives several devices or launches
// from several host threads (include/sbc_hip.h: threading).  Returns SBC_OK or SBC_ERR_HIP.
// A/B probe: SBC_PERSIST_CUS=<n> makes the persistent kernels (conv_pair, conv_pool, conv_dp, conv_res) size their grids for n CUs
// instead of all of them, so that two streams' launches can be resident side by side.
int persistent_cus(int cus);
int balanced_sample_grid(int samples, int cus);   // grid of the sample-per-workgroup persistent kernels (api.hip)
int ensure_dyn_lds(const void* kernel, size_t bytes);

// conv_mode f16x2: the current device's range-flag word (allocated and zeroed on first use; api.hip)
int range_flag_ptr(unsigned** out);

// per-kind launchers (each validates its op, then launches asynchronously on `stream`)
// dry = true: validate, resolve the kernel variant and set its function attributes, but do not launch
int launch_conv(const sbc_op& op, hipStream_t stream, bool dry = false);
int launch_conv_pair(const sbc_op& op, hipStream_t stream, bool dry = false);
int launch_conv_pool(const sbc_op& op, hipStream_t stream, bool dry = false);
int launch_res_block(const sbc_op& op, hipStream_t stream, bool dry = false);   // conv_res.hip
int launch_chain(const sbc_op& op, const sbc_chain& ext, hipStream_t stream, bool dry = false);   // conv_chain.hip
int launch_conv_down(const sbc_op& op, hipStream_t stream, bool dry = false);                    // conv_down.hip
int launch_begin_conv(const sbc_op& op, hipStream_t stream);
int launch_inorm_stats(const sbc_op& op, hipStream_t stream);
int launch_maxpool5(const sbc_op& op, hipStream_t stream);
int launch_end_conv(const sbc_op& op, const sbc_endconv& ext, hipStream_t stream, bool dry = false);
int launch_langevin(const sbc_op& op, const sbc_langevin& ext, hipStream_t stream, bool dry = false);
int launch_measure(const sbc_op& op, const sbc_langevin& ext, hipStream_t stream);
int launch_step_inc(const sbc_op& op, hipStream_t stream);
// training operators (train.hip, train_conv.hip)
int launch_dsm_perturb(const sbc_op& op, const sbc_dsm& ext, hipStream_t stream);
int launch_dsm_loss(const sbc_op& op, const sbc_dsm& ext, hipStream_t stream);
int launch_grad_add(const sbc_op& op, hipStream_t stream);
int launch_inorm_bwd(const sbc_op& op, hipStream_t stream);
int launch_maxpool5_bwd(const sbc_op& op, hipStream_t stream);
int launch_upsample_bwd(const sbc_op& op, hipStream_t stream);
int launch_pool_bwd(const sbc_op& op, hipStream_t stream);
int launch_conv_wgrad(const sbc_op& op, hipStream_t stream);
int launch_pack_weight(const sbc_op& op, hipStream_t stream);
int launch_end_conv_bwd(const sbc_op& op, const sbc_endconv& ext, hipStream_t stream);
int launch_begin_conv_bwd(const sbc_op& op, hipStream_t stream);
int launch_adam_ema(const sbc_op& op, const sbc_adam& ext, hipStream_t stream);

#if defined(__HIPCC__)
// nn.ELU(alpha=1): x > 0 ? x : exp(x) - 1   (ncsnv2/models/layers.py:12-13).  Written the way PyTorch's own ELU
// kernels evaluate it, exp(x) - 1, on the hardware exponential (v_exp_f32 of x*log2(e), ~1 ulp) rather than a
// ~20-instruction expm1: on gfx950 fp32 VALU work and fp32 MFMA share the same ALUs (tools/mfma_valu_coissue.hip:
// 146 TF + 130 TF alone, 83 + 42 TF together), so every vector instruction of the staging path is MFMA time.
// The select is a median: exp(x) - 1 >= x everywhere, so for x > 0 the middle of (x, exp(x) - 1, 0) is x and for x < 0 it
// is exp(x) - 1 -- one v_med3_f32 instead of compare + select.  (x = +inf gives med3(inf, inf, 0) = inf; NaN stays NaN.)
// Round 4: the median form is not exact for small POSITIVE x -- the rounded exp(x) - 1 can fall just below x, and the median then
// returns it (6e-8 off: 4e-5 of an activation of 1.6e-3).  Same four instructions, exact where ELU is the identity:
//       c = clamp(1 - exp(x), 0, 1)  (the clamp rides on the subtraction; c = 0 for x >= 0),   elu = max(x, -c).
// For x < 0 this is max(x, exp(x) - 1): the hardware exponential's absolute error (6e-8) is all that is left, and where it exceeds
// the second-order term x^2 / 2 (|x| < 3e-4) the max falls back to x itself.  (A NaN input leaves as -0: v_max returns its
// number operand; the Langevin state X keeps its NaN, so a diverged run still shows in the NMSE log.)
__device__ __forceinline__ float elu1(float x) {
    const float e = __expf(x);
    float y;
    // (by hand: from C++ hipcc adds a canonicalising v_max of x and a v_xor for the negation -- six instructions instead of four.
    // The s_nop is the wait state a vector instruction needs behind the transcendental unit's result: hipcc inserts it between
    // its own instructions but does not look inside inline assembly.)
    asm("s_nop 0\n\tv_sub_f32_e64 %0, 1.0, %1 clamp\n\tv_max_f32_e64 %0, %2, -%0" : "=&v"(y) : "v"(e), "v"(x));
    return y;
}
// d ELU(x) / dx = 1 (x > 0), exp(x) (x <= 0)
__device__ __forceinline__ float elu_grad1(float x) { return x > 0.f ? 1.f : __expf(x); }
// four values in one block, exponentials first: every consumer sits at least three instructions behind its producer (no wait
// states needed), 16 instructions in all
__device__ __forceinline__ float4 elu4(float4 v) {
    float4 y;
    float e0, e1, e2, e3;
    asm("v_mul_f32_e32 %4, 0x3fb8aa3b, %8\n\t"
        "v_mul_f32_e32 %5, 0x3fb8aa3b, %9\n\t"
        "v_mul_f32_e32 %6, 0x3fb8aa3b, %10\n\t"
        "v_mul_f32_e32 %7, 0x3fb8aa3b, %11\n\t"
        "v_exp_f32_e32 %4, %4\n\t"
        "v_exp_f32_e32 %5, %5\n\t"
        "v_exp_f32_e32 %6, %6\n\t"
        "v_exp_f32_e32 %7, %7\n\t"
        "v_sub_f32_e64 %0, 1.0, %4 clamp\n\t"
        "v_sub_f32_e64 %1, 1.0, %5 clamp\n\t"
        "v_sub_f32_e64 %2, 1.0, %6 clamp\n\t"
        "v_sub_f32_e64 %3, 1.0, %7 clamp\n\t"
        "v_max_f32_e64 %0, %8, -%0\n\t"
        "v_max_f32_e64 %1, %9, -%1\n\t"
        "v_max_f32_e64 %2, %10, -%2\n\t"
        "v_max_f32_e64 %3, %11, -%3"
        : "=&v"(y.x), "=&v"(y.y), "=&v"(y.z), "=&v"(y.w), "=&v"(e0), "=&v"(e1), "=&v"(e2), "=&v"(e3)
        : "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w));
    return y;
}
// ELU that keeps fp32's RELATIVE accuracy for small negative x (SBC_PRO_ELU_ACC): exp(x) - 1 on the hardware exponential carries an
// absolute error of 6e-8, i.e. 6e-5 of an activation of -1e-3, where the reference's expm1 is good to 1e-7 of the value.  For
// -1/32 < x < 0 a fourth-order Taylor polynomial (truncation x^4 / 120 < 1e-8 relative), below that exp(x) - 1 (relative error
// 6e-8 / |x| <= 2e-6, falling to 1e-7 at -0.5).  Ten vector instructions per value instead of four: used by the layers whose
// calibrated input maximum is below 2^-4 (sbc_f16x2_calibrate sets the request in the weight trailer), by the exact modes (the host
// sets the flag: bf16x3 / f32) and by the two places ELU sits outside a convolution prologue (max pool, CRP residual operand).
__device__ __forceinline__ float elu1_acc(float x) {
    // (the two coefficients that are not inline constants are pinned to scalar registers: hipcc would otherwise hoist them into
    // VECTOR registers for the whole kernel -- the fused pair kernel has none to spare)
    const float c4 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(0x3d2aaaab));   // 1 / 24
    const float c3 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(0x3e2aaaab));   // 1 / 6
    const float p = x * fmaf(x, fmaf(x, fmaf(x, c4, c3), 0.5f), 1.f);
    const float e = __expf(x) - 1.f;
    return x > 0.f ? x : (x > -0.03125f ? p : e);
}
__device__ __forceinline__ float4 elu4_acc(float4 v) {
    // one value after the other (the barriers keep the scheduler from interleaving the four chains): a rare path that must not
    // raise the register demand of the kernels it sits in
    float4 y;
    y.x = elu1_acc(v.x);
    __builtin_amdgcn_sched_barrier(0);
    y.y = elu1_acc(v.y);
    __builtin_amdgcn_sched_barrier(0);
    y.z = elu1_acc(v.z);
    __builtin_amdgcn_sched_barrier(0);
    y.w = elu1_acc(v.w);
    return y;
}
// `acc` is uniform over the launch: a scalar branch
__device__ __forceinline__ float4 elu4(float4 v, bool acc) { return acc ? elu4_acc(v) : elu4(v); }
#endif

}  // namespace sbc

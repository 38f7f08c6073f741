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
int ensure_dyn_lds(const void* kernel, size_t bytes);

// conv_mode f16x2: the current device's range-flag word (allocated and zeroed on first use; api.hip)
int range_flag_ptr(unsigned** out);

// per-kind launchers (each validates its op, then launches asynchronously on `stream`)
// dry = true: validate, resolve the kernel variant and set its function attributes, but do not launch
int launch_conv(const sbc_op& op, hipStream_t stream, bool dry = false);
int launch_conv_pair(const sbc_op& op, hipStream_t stream, bool dry = false);
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
__device__ __forceinline__ float elu1(float x) { return __builtin_amdgcn_fmed3f(x, __expf(x) - 1.f, 0.f); }
// d ELU(x) / dx = 1 (x > 0), exp(x) (x <= 0)
__device__ __forceinline__ float elu_grad1(float x) { return x > 0.f ? 1.f : __expf(x); }
__device__ __forceinline__ float4 elu4(float4 v) {
    return make_float4(elu1(v.x), elu1(v.y), elu1(v.z), elu1(v.w));
}
#endif

}  // namespace sbc

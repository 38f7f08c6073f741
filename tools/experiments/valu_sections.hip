// Tuning aid (not part of the product): how fast do the operand sections of the fused kernels run when ONE wave per SIMD executes them
// (the young waves of a workgroup after their K loop, round-6 timelines) against two?  Per unit of four values: elu4 (common.h),
// scale_track + split_f16x2 (tile.h), as the kernels call them, over 8 units from registers.
//   build: hipcc --offload-arch=gfx950 -O3 -I ../../score_based_channels_amd/csrc -I ../../include -o valu_sections valu_sections.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "common.h"
#include "tile.h"
using namespace sbc;

template <int WHAT, int INTERLEAVE>
__global__ __launch_bounds__(512) void sect(float* out, const float* in, int iters, float scale) {
    float4 v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = reinterpret_cast<const float4*>(in)[threadIdx.x * 8 + i];
    uint2 h[8], l[8];
    float ta = 0.f;
    unsigned acc = 0;
    for (int it = 0; it < iters; ++it) {
        if constexpr (INTERLEAVE == 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float4 e = v[i];
                if (WHAT & 1) e = elu4(e);
                if (WHAT & 2) {
                    StageScale ss{scale, ta};
                    scale_track(e, &ss);
                    ta = ss.amax;
                    split_f16x2(e, scale, h[i], l[i]);
                } else { h[i] = make_uint2(__float_as_uint(e.x), __float_as_uint(e.y)); l[i] = make_uint2(__float_as_uint(e.z), __float_as_uint(e.w)); }
            }
        } else {
            float4 e[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) e[i] = (WHAT & 1) ? elu4(v[i]) : v[i];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (WHAT & 2) {
                    StageScale ss{scale, ta};
                    scale_track(e[i], &ss);
                    ta = ss.amax;
                    split_f16x2(e[i], scale, h[i], l[i]);
                } else { h[i] = make_uint2(__float_as_uint(e[i].x), __float_as_uint(e[i].y)); l[i] = make_uint2(__float_as_uint(e[i].z), __float_as_uint(e[i].w)); }
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) { acc ^= h[i].x ^ h[i].y ^ l[i].x ^ l[i].y; v[i].x += 1e-7f * (float)(acc & 1); }
    }
    out[blockIdx.x * 512 + threadIdx.x] = ta + (float)acc;
}

template <class K> float timeit(K kern, int threads, float* out, const float* in, int iters) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, out, in, iters, 2.f);
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, out, in, iters, 2.f);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best * 1e3f;
}
template <int WHAT, int IL> void row(float* out, const float* in, int it, const char* name, int instr) {
    const float t1 = timeit(sect<WHAT, IL>, 256, out, in, it), t2 = timeit(sect<WHAT, IL>, 512, out, in, it);
    printf("  %-44s %s: 1 wave/SIMD %7.1f us = %5.1f ns per unit;  2 waves/SIMD %7.1f us = %5.1f ns per unit and wave   (%d vector instructions per unit + the loop's own ~5)\n",
           name, IL ? "stages apart " : "unit by unit", t1, t1 * 1e3 / it / 8, t2, t2 * 1e3 / it / 8 / 2, instr);
}
int main() {
    float *out, *in;
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&in, 512 * 8 * 16);
    (void)hipMemset(in, 0x3c, 512 * 8 * 16);
    const int it = 20000;
    row<1, 0>(out, in, it, "elu4", 16); row<1, 1>(out, in, it, "elu4", 16);
    row<2, 0>(out, in, it, "scale_track + split_f16x2", 10); row<2, 1>(out, in, it, "scale_track + split_f16x2", 10);
    row<3, 0>(out, in, it, "elu4 + scale_track + split_f16x2", 26); row<3, 1>(out, in, it, "elu4 + scale_track + split_f16x2", 26);
    return 0;
}

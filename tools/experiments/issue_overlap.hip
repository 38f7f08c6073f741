// Micro-benchmark (not part of the product): do matrix (MFMA) and vector-ALU instructions of DIFFERENT waves of one SIMD overlap on
// gfx950?  One 8-wave workgroup per CU (two waves per SIMD): waves 0-3 run an MFMA loop, waves 4-7 an FMA loop.
// Timed alone and together; also a dependent MFMA chain (each instruction accumulates into the previous result) against four
// independent accumulators.      hipcc --offload-arch=gfx950 -O3 issue_overlap.hip -o issue_overlap && ./issue_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>   // bit 4: every wave is a matrix wave; bit 5 / 6: vector / matrix waves raise their priority; bit 0: matrix waves active, bit 1: vector waves active, bit 2: dependent MFMA chain, bit 3: vector waves do exp
__global__ __launch_bounds__(1024) void k(float* out, int iters) {
    const int wave = threadIdx.x >> 6;
    f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    float v0 = threadIdx.x, v1 = 1.f, v2 = 2.f, v3 = 3.f;
    if ((MODE & 16) || (wave & 4) == 0) {   // waves 0-3 (one per SIMD: a workgroup's waves go round the SIMDs) matrix, 4-7 vector
        if (!(MODE & 1)) return;
        if (MODE & 64) __builtin_amdgcn_s_setprio(3);
        f16x8 x = {1, 2, 3, 4, 5, 6, 7, 8}, w = {1, 1, 1, 1, 1, 1, 1, 1};
        for (int i = 0; i < iters; ++i) {
            if (MODE & 4) {
                a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w, x, a0, 0, 0, 0);
                a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w, x, a0, 0, 0, 0);
                a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w, x, a0, 0, 0, 0);
                a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w, x, a0, 0, 0, 0);
            } else {
                a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w, x, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w, x, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w, x, a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w, x, a3, 0, 0, 0);
            }
        }
    } else {
        if (!(MODE & 2)) return;
        if (MODE & 32) __builtin_amdgcn_s_setprio(3);
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (MODE & 8) { v0 = __builtin_amdgcn_exp2f(v0); v1 = __builtin_amdgcn_exp2f(v1); v2 = __builtin_amdgcn_exp2f(v2); v3 = __builtin_amdgcn_exp2f(v3); }
                else { v0 = fmaf(v0, 1.0001f, 0.5f); v1 = fmaf(v1, 1.0001f, 0.5f); v2 = fmaf(v2, 1.0001f, 0.5f); v3 = fmaf(v3, 1.0001f, 0.5f); }
            }
        }
    }
    out[blockIdx.x * 1024 + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3] + v0 + v1 + v2 + v3;
}

template <int MODE> float run(float* out, int iters, int threads = 512) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, iters);
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 5 * 1e3f;
}
int main() {
    float* out; hipMalloc(&out, 256 * 1024 * 4);
    const int it = 20000;   // per wave: 80 000 MFMAs (16 cycles each at full rate = 1.28 M cycles) / 320 000 vector instructions (4 cycles each)
    printf("matrix waves alone, 4 independent accumulators: %8.1f us\n", run<1>(out, it));
    printf("matrix waves alone, one dependent chain:        %8.1f us\n", run<1 | 4>(out, it));
    printf("vector waves alone (fma):                       %8.1f us\n", run<2>(out, it));
    printf("both (independent MFMAs + fma):                 %8.1f us\n", run<3>(out, it));
    printf("both (dependent chain + fma):                   %8.1f us\n", run<3 | 4>(out, it));
    printf("vector waves alone (exp):                       %8.1f us\n", run<2 | 8>(out, it));
    printf("both (independent MFMAs + exp):                 %8.1f us\n", run<3 | 8>(out, it));
    printf("both, vector waves at priority 3:               %8.1f us\n", run<3 | 32>(out, it));
    printf("both, matrix waves at priority 3:               %8.1f us\n", run<3 | 64>(out, it));
    printf("both (dependent chain), vector waves at prio 3: %8.1f us\n", run<3 | 4 | 32>(out, it));
    printf("matrix waves only, 1 / 2 / 4 per SIMD (independent): %8.1f %8.1f %8.1f us\n", run<1 | 16>(out, it, 256), run<1 | 16>(out, it, 512), run<1 | 16>(out, it, 1024));
    printf("matrix waves only, 1 / 2 / 4 per SIMD (dependent):   %8.1f %8.1f %8.1f us\n", run<1 | 4 | 16>(out, it, 256), run<1 | 4 | 16>(out, it, 512), run<1 | 4 | 16>(out, it, 1024));
    return 0;
}

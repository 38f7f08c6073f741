// Tuning aid (not part of the product): do a matrix wave and a vector wave on the SAME SIMD overlap when their priorities differ?
// issue_overlap2.hip measured "both = matrix alone + vector alone" with equal priorities.  Here: s_setprio on one kind or the other, a dense
// (4 independent chains) and a sparse (1 dependent chain) vector stream, and the matrix stream written with explicit s_nop gaps.
//   build: hipcc --offload-arch=gfx950 -O3 -o issue_overlap3 issue_overlap3.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// waves 0-3 matrix (one per SIMD), waves 4-7 vector.  MODE bit 0 matrix on, bit 1 vector on.  PRIO 0 none, 1 vector high, 2 matrix high.
// CHAINS independent fma chains in the vector wave (4 = dense issue, 1 = one instruction per dependent-issue latency).
template <int SHAPE, int MODE, int PRIO, int CHAINS>
__global__ __launch_bounds__(512) void two_kinds(float* out, int iters) {
    const int wave = threadIdx.x >> 6;
    float r = 0.f;
    if ((wave & 4) == 0) {
        if (!(MODE & 1)) return;
        if (PRIO == 2) __builtin_amdgcn_s_setprio(3);
        f16x8 x = {1, 2, 3, 4, 5, 6, 7, 8}, w = {1, 1, 1, 1, 1, 1, 1, 1};
        if constexpr (SHAPE == 0) {
            f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
            for (int i = 0; i < iters; ++i) {
                a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w, x, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w, x, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w, x, a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w, x, a3, 0, 0, 0);
            }
            r = a0[0] + a1[1] + a2[2] + a3[3];
        } else {
            f32x16 a0 = {0}, a1 = {0};
            for (int i = 0; i < iters; ++i) {
                a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w, x, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w, x, a1, 0, 0, 0);
            }
            r = a0[0] + a1[5];
        }
    } else {
        if (!(MODE & 2)) return;
        if (PRIO == 1) __builtin_amdgcn_s_setprio(3);
        float v0 = threadIdx.x, v1 = 1.f, v2 = 2.f, v3 = 3.f;
        for (int i = 0; i < iters; ++i) {
            if constexpr (CHAINS == 4) {
#pragma unroll
                for (int j = 0; j < 4; ++j) { v0 = fmaf(v0, 1.0001f, 0.5f); v1 = fmaf(v1, 1.0001f, 0.5f); v2 = fmaf(v2, 1.0001f, 0.5f); v3 = fmaf(v3, 1.0001f, 0.5f); }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) v0 = fmaf(v0, 1.0001f, 0.5f);
            }
        }
        r = v0 + v1 + v2 + v3;
    }
    out[blockIdx.x * 512 + threadIdx.x] = r;
}

template <class K> float timeit(K kern, int threads, float* out, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, out, iters);
    hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, out, iters);
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    hipEventDestroy(e0); hipEventDestroy(e1);
    return best * 1e3f;
}
template <int SHAPE, int PRIO, int CHAINS> void row(float* out, int it) {
    const char* pn[] = {"equal priority", "vector waves high", "matrix waves high"};
    printf("  %s  %-18s %s vector stream:  matrix alone %8.1f  vector alone %8.1f  both %8.1f us\n", SHAPE ? "32x32x16" : "16x16x32", pn[PRIO],
           CHAINS == 4 ? "dense " : "sparse", timeit(two_kinds<SHAPE, 1, PRIO, CHAINS>, 512, out, it), timeit(two_kinds<SHAPE, 2, PRIO, CHAINS>, 512, out, it),
           timeit(two_kinds<SHAPE, 3, PRIO, CHAINS>, 512, out, it));
}
int main() {
    float* out; hipMalloc(&out, 256 * 512 * 4);
    const int it = 60000;
    printf("separate matrix and vector waves on one SIMD (per iteration: 64 matrix cycles / 16 (dense) or 8 (sparse) v_fma_f32):\n");
    row<0, 0, 4>(out, it); row<0, 1, 4>(out, it); row<0, 2, 4>(out, it);
    row<0, 0, 1>(out, it); row<0, 1, 1>(out, it); row<0, 2, 1>(out, it);
    row<1, 0, 4>(out, it); row<1, 1, 4>(out, it); row<1, 2, 4>(out, it);
    row<1, 0, 1>(out, it); row<1, 1, 1>(out, it); row<1, 2, 1>(out, it);
    return 0;
}

// Micro-benchmark (not part of the product), round 6: how many vector-ALU instructions does a matrix instruction HIDE on gfx950, as a
// function of the MFMA shape?  Round 4's issue_overlap.hip tested only v_mfma_f32_16x16x32_f16 (16 cycles) against vector waves on the
// same SIMD and found the two streams' times ADD; /opt/skills/guides/MI355X_MICROARCH.md reports up to 5 single-issue fillers hidden under
// one 32-cycle v_mfma_f32_32x32x16.  This sweep settles it for the shapes the hot path can use:
//   same-wave interleave: every wave runs  { MFMA ; F x v_fma_f32 } , F = 0 .. 10, for both shapes, one and two waves per SIMD;
//   separate waves:       matrix waves + vector waves on the same SIMD (round 4's arrangement), both shapes.
// Equal matrix WORK per iteration: one 32x32x16 = two 16x16x32 (16384 MACs).  hipcc --offload-arch=gfx950 -O3 issue_overlap2.hip -o issue_overlap2
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define FILL(n) if (F > n) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[n & 7]) : "v"(c1), "v"(c2));
#define FILLB(n) if (F > n) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[(n + 4) & 7]) : "v"(c1), "v"(c2));
#define FILLERSB FILLB(0) FILLB(1) FILLB(2) FILLB(3) FILLB(4) FILLB(5) FILLB(6) FILLB(7) FILLB(8) FILLB(9) FILLB(10) FILLB(11)
#define FILLERS FILL(0) FILL(1) FILL(2) FILL(3) FILL(4) FILL(5) FILL(6) FILL(7) FILL(8) FILL(9) FILL(10) FILL(11)

// SHAPE 0: per iteration 4 x { v_mfma_f32_16x16x32_f16 ; F fillers }        (4 x 8192 MACs)
// SHAPE 1: per iteration 2 x { v_mfma_f32_32x32x16_f16 ; 2 F fillers }       (2 x 16384 MACs: the same matrix work and the same fillers)
template <int SHAPE, int F>
__global__ __launch_bounds__(512) void same_wave(float* out, int iters) {
    float v[8] = {(float)threadIdx.x, 1.f, 2.f, 3.f, 4.f, 5.f, 6.f, 7.f};
    const float c1 = 1.0001f, c2 = 0.5f;
    f16x8 x = {1, 2, 3, 4, 5, 6, 7, 8}, w = {1, 1, 1, 1, 1, 1, 1, 1};
    float r = 0.f;
    if constexpr (SHAPE == 0) {
        f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
        for (int i = 0; i < iters; ++i) {
            a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w, x, a0, 0, 0, 0); FILLERS __builtin_amdgcn_sched_barrier(0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w, x, a1, 0, 0, 0); FILLERS __builtin_amdgcn_sched_barrier(0);
            a2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w, x, a2, 0, 0, 0); FILLERS __builtin_amdgcn_sched_barrier(0);
            a3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w, x, a3, 0, 0, 0); FILLERS __builtin_amdgcn_sched_barrier(0);
        }
        r = a0[0] + a1[1] + a2[2] + a3[3];
    } else {
        f32x16 a0 = {0}, a1 = {0};
        for (int i = 0; i < iters; ++i) {
            a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w, x, a0, 0, 0, 0); FILLERS FILLERSB __builtin_amdgcn_sched_barrier(0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w, x, a1, 0, 0, 0); FILLERS FILLERSB __builtin_amdgcn_sched_barrier(0);
        }
        r = a0[0] + a1[5];
    }
    out[blockIdx.x * 512 + threadIdx.x] = r + v[0] + v[1] + v[2] + v[3] + v[4] + v[5] + v[6] + v[7];
}

// separate waves on one SIMD: waves 0-3 matrix (one per SIMD), waves 4-7 vector; MODE bit 0 matrix on, bit 1 vector on
template <int SHAPE, int MODE>
__global__ __launch_bounds__(512) void two_kinds(float* out, int iters) {
    const int wave = threadIdx.x >> 6;
    float r = 0.f;
    if ((wave & 4) == 0) {
        if (!(MODE & 1)) return;
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
        float v0 = threadIdx.x, v1 = 1.f, v2 = 2.f, v3 = 3.f;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { v0 = fmaf(v0, 1.0001f, 0.5f); v1 = fmaf(v1, 1.0001f, 0.5f); v2 = fmaf(v2, 1.0001f, 0.5f); v3 = fmaf(v3, 1.0001f, 0.5f); }
        }
        r = v0 + v1 + v2 + v3;
    }
    out[blockIdx.x * 512 + threadIdx.x] = r;
}

// dependent accumulator chains: every MFMA accumulates onto the one before it (DEP = 1), or two chains alternate (DEP = 2)
template <int SHAPE, int DEP>
__global__ __launch_bounds__(512) void dep_chain(float* out, int iters) {
    f16x8 x = {1, 2, 3, 4, 5, 6, 7, 8}, w = {1, 1, 1, 1, 1, 1, 1, 1};
    float r = 0.f;
    if constexpr (SHAPE == 0) {
        f32x4 a0 = {0, 0, 0, 0}, a1 = a0;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (DEP == 1 || (k & 1) == 0) a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w, x, a0, 0, 0, 0);
                else a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w, x, a1, 0, 0, 0);
            }
        }
        r = a0[0] + a1[1];
    } else {
        f32x16 a0 = {0}, a1 = {0};
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                if (DEP == 1 || (k & 1) == 0) a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w, x, a0, 0, 0, 0);
                else a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w, x, a1, 0, 0, 0);
            }
        }
        r = a0[0] + a1[5];
    }
    out[blockIdx.x * 512 + threadIdx.x] = r;
}

template <class K> float timeit(K kern, int threads, float* out, int iters) {
    // minimum of 7 timed launches after two warm-up launches (the chip's clock moves with the power state: the minimum is the
    // launch that ran at the steadiest high clock)
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, out, iters);
    hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, out, iters);
    float best = 1e30f;
    for (int r = 0; r < 7; ++r) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    hipEventDestroy(e0); hipEventDestroy(e1);
    return best * 1e3f;
}

template <int SHAPE, int F> void row(float* out, int it) {
    const float t1 = timeit(same_wave<SHAPE, F>, 256, out, it), t2 = timeit(same_wave<SHAPE, F>, 512, out, it);
    // per iteration: 4 x 16-cycle (or 2 x 32-cycle) MFMAs = 64 matrix cycles and 4 F fillers
    printf("  %s  F = %2d fillers per 16 matrix cycles:  1 wave/SIMD %8.1f us   2 waves/SIMD %8.1f us   (per iteration of 64 matrix cycles + %2d fillers: %6.2f / %6.2f ns)\n",
           SHAPE ? "32x32x16" : "16x16x32", F, t1, t2, 4 * F, t1 * 1e3 / it, t2 * 1e3 / it / 2);
}
template <int SHAPE> void sweep(float* out, int it) {
    row<SHAPE, 0>(out, it); row<SHAPE, 1>(out, it); row<SHAPE, 2>(out, it); row<SHAPE, 3>(out, it); row<SHAPE, 4>(out, it); row<SHAPE, 5>(out, it);
    row<SHAPE, 6>(out, it); row<SHAPE, 8>(out, it); row<SHAPE, 10>(out, it); row<SHAPE, 12>(out, it);
}
int main() {
    float* out; hipMalloc(&out, 256 * 512 * 4);
    const int it = 60000;
    printf("same-wave interleave { MFMA ; fillers } (v_fma_f32, independent), %d iterations per wave:\n", it);
    sweep<0>(out, it);
    sweep<1>(out, it);
    printf("dependent accumulator chains, no fillers (us; 1 wave/SIMD, 2 waves/SIMD):\n");
    printf("  16x16x32: one chain %8.1f %8.1f   two alternating chains %8.1f %8.1f\n", timeit(dep_chain<0, 1>, 256, out, it), timeit(dep_chain<0, 1>, 512, out, it), timeit(dep_chain<0, 2>, 256, out, it), timeit(dep_chain<0, 2>, 512, out, it));
    printf("  32x32x16: one chain %8.1f %8.1f   two alternating chains %8.1f %8.1f\n", timeit(dep_chain<1, 1>, 256, out, it), timeit(dep_chain<1, 1>, 512, out, it), timeit(dep_chain<1, 2>, 256, out, it), timeit(dep_chain<1, 2>, 512, out, it));
    printf("separate matrix and vector waves on one SIMD (4 it x 16 matrix cycles / 16 it v_fma_f32 per wave):\n");
    printf("  16x16x32: matrix alone %8.1f  vector alone %8.1f  both %8.1f us\n", timeit(two_kinds<0, 1>, 512, out, it), timeit(two_kinds<0, 2>, 512, out, it), timeit(two_kinds<0, 3>, 512, out, it));
    printf("  32x32x16: matrix alone %8.1f  vector alone %8.1f  both %8.1f us\n", timeit(two_kinds<1, 1>, 512, out, it), timeit(two_kinds<1, 2>, 512, out, it), timeit(two_kinds<1, 3>, 512, out, it));
    return 0;
}

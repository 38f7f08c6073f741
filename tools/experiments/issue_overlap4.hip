// Tuning aid (not part of the product): how dense may a vector wave's instruction stream be before it stops overlapping with a matrix wave
// on the same SIMD?  issue_overlap3.hip: a 4-chain v_fma stream adds to the matrix time whatever the priorities, a 1-chain stream hides
// completely when the vector wave has the higher priority.  Here: 1 .. 4 chains, s_nop-diluted dense streams, and two matrix waves + one
// vector wave per SIMD (the arrangement of conv_pair_roll_kernel).
//   build: hipcc --offload-arch=gfx950 -O3 -o issue_overlap4 issue_overlap4.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// NM matrix waves per SIMD (waves 0 .. 4 NM - 1), then 4 vector waves.  MODE bit 0 matrix on, bit 1 vector on.  PRIO 1: vector waves high.
// CHAINS independent v_fma chains; NOP: s_nop NOP-1 behind every v_fma (0: none).
template <int NM, int MODE, int PRIO, int CHAINS, int NOP>
__global__ __launch_bounds__(768) void mix(float* out, int iters) {
    const int wave = threadIdx.x >> 6;
    float r = 0.f;
    if (wave < 4 * NM) {
        if (!(MODE & 1)) return;
        f16x8 x = {1, 2, 3, 4, 5, 6, 7, 8}, w = {1, 1, 1, 1, 1, 1, 1, 1};
        f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
        for (int i = 0; i < iters / NM; ++i) {
            a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w, x, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w, x, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w, x, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w, x, a3, 0, 0, 0);
        }
        r = a0[0] + a1[1] + a2[2] + a3[3];
    } else {
        if (!(MODE & 2)) return;
        if (PRIO == 1) __builtin_amdgcn_s_setprio(3);
        float v[4] = {(float)threadIdx.x, 1.f, 2.f, 3.f};
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {                     // 8 v_fma per iteration (per 64 matrix cycles), whatever the chain count
                v[j % CHAINS] = fmaf(v[j % CHAINS], 1.0001f, 0.5f);
                if constexpr (NOP > 0) asm volatile("s_nop %0" :: "n"(NOP - 1));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        r = v[0] + v[1] + v[2] + v[3];
    }
    out[blockIdx.x * 768 + threadIdx.x] = r;
}

template <class K> float timeit(K kern, int threads, float* out, int iters) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, out, iters);
    hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, out, iters);
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, out, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best * 1e3f;
}
template <int NM, int PRIO, int CHAINS, int NOP> void row(float* out, int it) {
    const int th = 64 * (4 * NM + 4);
    const float m = timeit(mix<NM, 1, PRIO, CHAINS, NOP>, th, out, it), v = timeit(mix<NM, 2, PRIO, CHAINS, NOP>, th, out, it), b = timeit(mix<NM, 3, PRIO, CHAINS, NOP>, th, out, it);
    printf("  %d matrix wave(s) + 1 vector wave per SIMD, %s, %d chain(s), s_nop %d:  matrix alone %7.1f  vector alone %7.1f  both %7.1f us   hidden %4.0f %% of the shorter\n",
           NM, PRIO ? "vector high" : "equal prio ", CHAINS, NOP, m, v, b, 100.f * (m + v - b) / (m < v ? m : v));
}
int main() {
    float* out; (void)hipMalloc(&out, 256 * 768 * 4);
    const int it = 60000;
    printf("per iteration: 64 matrix cycles (4 x v_mfma_f32_16x16x32_f16) per SIMD and 8 v_fma_f32 in the vector wave\n");
    row<1, 0, 1, 0>(out, it); row<1, 1, 1, 0>(out, it);
    row<1, 0, 2, 0>(out, it); row<1, 1, 2, 0>(out, it);
    row<1, 0, 3, 0>(out, it); row<1, 1, 3, 0>(out, it);
    row<1, 0, 4, 0>(out, it); row<1, 1, 4, 0>(out, it);
    row<1, 0, 4, 1>(out, it); row<1, 1, 4, 1>(out, it);
    row<1, 0, 4, 2>(out, it); row<1, 1, 4, 2>(out, it);
    row<1, 0, 4, 4>(out, it); row<1, 1, 4, 4>(out, it);
    row<2, 0, 1, 0>(out, it); row<2, 1, 1, 0>(out, it);
    row<2, 0, 4, 0>(out, it); row<2, 1, 4, 0>(out, it);
    row<2, 0, 4, 2>(out, it); row<2, 1, 4, 2>(out, it);
    return 0;
}

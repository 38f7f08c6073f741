// Micro-benchmark (tooling, not product): sustained v_mfma_f32_32x32x16_bf16 rate as a function of the number of
// independent accumulators per wave and of waves per SIMD, with and without LDS operand reads between MFMA blocks.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NACC, int LDSREADS, int GLOADS>
__global__ __launch_bounds__(256) void k(float* out, int iters, const uint4* src, const uint4* wsrc) {
    __shared__ uint4 sm[2048];
    for (int i = threadIdx.x; i < 2048; i += 256) sm[i] = src[i];
    __syncthreads();
    f32x16 acc[NACC];
    for (int n = 0; n < NACC; ++n) for (int i = 0; i < 16; ++i) acc[n][i] = 0.f;
    bf16x8 a[4], b[3];
    for (int i = 0; i < 4; ++i) a[i] = __builtin_bit_cast(bf16x8, sm[threadIdx.x + 256 * i]);
    for (int i = 0; i < 3; ++i) b[i] = __builtin_bit_cast(bf16x8, sm[threadIdx.x + 1024 + 256 * i]);
    int off = threadIdx.x;
    for (int it = 0; it < iters; ++it) {
        if (GLOADS) {
            // 54 KB weight array walked like the convolution's B fragments (L2-resident, larger than the 32 KB L1)
            const uint4* w = wsrc + (size_t)((it % 18) * 3) * 64 + (threadIdx.x & 63);
#pragma unroll
            for (int i = 0; i < GLOADS; ++i) b[i % 3] = __builtin_bit_cast(bf16x8, w[i * 64]);
        }
        if (LDSREADS) {
#pragma unroll
            for (int i = 0; i < LDSREADS; ++i) a[i & 3] = __builtin_bit_cast(bf16x8, sm[(off + 64 * i) & 2047]);
            off += 17;
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int q = 0; q < 12 / NACC; ++q)
#pragma unroll
            for (int n = 0; n < NACC; ++n)
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(q + n) & 3], b[q % 3], acc[n], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    float r = 0;
    for (int n = 0; n < NACC; ++n) for (int i = 0; i < 16; ++i) r += acc[n][i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int NACC, int LDSREADS, int GLOADS = 0>
void run(int wgs_per_cu) {
    float* out; uint4* src; uint4* wsrc; hipMalloc(&wsrc, 64 * 1024); hipMemset(wsrc, 0, 64 * 1024);
    hipMalloc(&out, 8192 * 256 * 4); hipMalloc(&src, 2048 * 16); hipMemset(src, 0, 2048 * 16);
    const int iters = 4000, grid = 256 * wgs_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NACC, LDSREADS, GLOADS>), dim3(grid), dim3(256), 0, 0, out, 10, src, wsrc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NACC, LDSREADS, GLOADS>), dim3(grid), dim3(256), 0, 0, out, iters, src, wsrc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double fl = 12.0 * 2 * 32 * 32 * 16 * iters * grid * 4;
    printf("accs %d  lds reads/iter %d  global loads/iter %d  waves/SIMD %d : %7.3f ms  %7.1f TF\n", NACC, LDSREADS, GLOADS, wgs_per_cu, ms, fl / ms / 1e9);
    hipFree(out); hipFree(src);
}

int main() {
    for (int w = 1; w <= 4; w *= 2) { run<1, 0>(w); run<2, 0>(w); run<4, 0>(w); run<2, 6>(w); run<4, 6>(w); run<2, 12>(w); run<2, 0, 3>(w); run<2, 6, 3>(w); }
    return 0;
}

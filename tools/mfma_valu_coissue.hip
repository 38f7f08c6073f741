// Micro-benchmark (tooling, not product): can fp32 MFMA and fp32 VALU FMA streams of the same CU overlap?
// mode 0: MFMA only; mode 1: VALU v_fma only; mode 2: both interleaved in one wave; mode 3: pk_fma only; mode 4: MFMA+pk_fma
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float s) {
    f32x16 acc0 = {0}, acc1 = {0}, acc2 = {0}, acc3 = {0};
    float v[16];
    f32x2 pk[8];
    for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 1e-3f + i;
    for (int i = 0; i < 8; ++i) pk[i] = f32x2{v[2 * i], v[2 * i + 1]};
    float a = threadIdx.x * 1e-4f + 1.0f, b = 0.999f + s;
    unsigned u[16];
    for (int i = 0; i < 16; ++i) u[i] = threadIdx.x * 7u + i;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0 || MODE == 2 || MODE == 4 || MODE == 6 || MODE == 8) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc2, 0, 0, 0);
            acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc3, 0, 0, 0);
        }
        if (MODE == 1 || MODE == 2) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) v[i] = __builtin_fmaf(v[i], b, s);
        }
        if (MODE == 5 || MODE == 6) {      // integer VALU: v_mad_u32_u24 / v_xor chains
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) u[i] = (u[i] ^ (unsigned)it) + 0x9e3779b9u;
        }
        if (MODE == 7 || MODE == 8) {      // transcendental: v_exp_f32
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = __builtin_amdgcn_exp2f(v[i] * 1e-3f);
        }
        if (MODE == 3 || MODE == 4) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) pk[i] = __builtin_elementwise_fma(pk[i], f32x2{b, b}, f32x2{s, s});
        }
    }
    float r = 0;
    for (int i = 0; i < 16; ++i) r += acc0[i] + acc1[i] + acc2[i] + acc3[i] + v[i];
    for (int i = 0; i < 8; ++i) r += pk[i].x + pk[i].y;
    for (int i = 0; i < 16; ++i) r += (float)u[i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int MODE>
void run(const char* name, double mfma_flop_per_iter, double valu_flop_per_iter) {
    float* out;
    hipMalloc(&out, 4096 * 256 * 4);
    const int iters = 20000, grid = 1024;   // 4 workgroups of 4 waves per CU: 4 waves per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, out, 100, 0.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, out, iters, 1e-9f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double waves = grid * 4.0;
    printf("%-28s %8.3f ms  MFMA %7.1f TF  VALU %7.1f TF\n", name, ms, mfma_flop_per_iter * iters * waves / ms / 1e9,
           valu_flop_per_iter * iters * waves / ms / 1e9);
    hipFree(out);
}

int main() {
    const double mf = 4 * 2.0 * 32 * 32 * 2, vf = 64.0 * 64 * 2;   // per wave per iteration
    run<0>("MFMA only", mf, 0);
    run<1>("v_fma only", 0, vf);
    run<2>("MFMA + v_fma interleaved", mf, vf);
    run<3>("v_pk_fma only", 0, vf);
    run<4>("MFMA + v_pk_fma interleaved", mf, vf);
    run<5>("int VALU only (128 ops/iter)", 0, 128.0 * 64);
    run<6>("MFMA + int VALU", mf, 128.0 * 64);
    run<7>("v_exp only (16+16mul/iter)", 0, 16.0 * 64);
    run<8>("MFMA + v_exp", mf, 16.0 * 64);
    return 0;
}

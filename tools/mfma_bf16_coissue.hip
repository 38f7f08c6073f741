// Micro-benchmark (tooling, not product): on one SIMD, wave A issues bf16 MFMAs back to back while wave B runs a
// VALU / LDS-write / transcendental stream.  How much does each slow the other?  (fp32 MFMA: tools/mfma_valu_coissue.hip)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// 512 threads = 8 waves: waves 0-3 (one per SIMD) role A, waves 4-7 role B
template <int ROLE_A, int ROLE_B>   // 0 = idle, 1 = MFMA, 2 = v_fma chain, 3 = cvt/shift/sub split stream, 4 = v_exp
__global__ __launch_bounds__(512) void k(float* out, int iters, long long* cyc) {
    __shared__ float sm[4096];
    const int wave = threadIdx.x >> 6;
    const int role = wave < 4 ? ROLE_A : ROLE_B;
    f32x16 acc = {0};
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 1e-3f + i); b[i] = (__bf16)(i * 0.5f); }
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 1e-3f + i;
    const long long t0 = __builtin_readcyclecounter();
    if (role == 1) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int q = 0; q < 16; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
        }
    } else if (role == 2) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) v[i] = __builtin_fmaf(v[i], 0.999f, 1e-9f);
        }
    } else if (role == 3) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const __bf16 h = (__bf16)v[i];
                const float r1 = v[i] - (float)h;
                const __bf16 m = (__bf16)r1;
                v[i] = (r1 - (float)m) + v[i] * 0.5f + 1.f;
            }
        }
    } else if (role == 4) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = __builtin_amdgcn_exp2f(v[i] * 1e-3f);
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float r = 0;
    for (int i = 0; i < 16; ++i) r += acc[i] + v[i];
    sm[threadIdx.x] = r;
    out[blockIdx.x * 512 + threadIdx.x] = sm[threadIdx.x];
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) cyc[wave] = t1 - t0;
}

template <int A, int B>
void run(const char* name, int instrA, int instrB) {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 64);
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<A, B>), dim3(256), dim3(512), 0, 0, out, 10, cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<A, B>), dim3(256), dim3(512), 0, 0, out, iters, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[8]; hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
    printf("%-34s %7.3f ms | wave A: %6.1f ticks/instr  wave B: %6.1f ticks/instr  (ticks per us: %.0f)\n", name, ms,
           instrA ? (double)h[0] / iters / instrA : 0.0, instrB ? (double)h[4] / iters / instrB : 0.0,
           (double)(h[0] > h[4] ? h[0] : h[4]) / (ms * 1e3));
    hipFree(out); hipFree(cyc);
}

int main() {
    run<1, 0>("MFMA alone", 16, 0);
    run<0, 2>("v_fma alone", 0, 128);
    run<1, 2>("MFMA + v_fma", 16, 128);
    run<0, 3>("split stream alone", 0, 32);
    run<1, 3>("MFMA + split stream", 16, 32);
    run<0, 4>("v_exp alone", 0, 32);
    run<1, 4>("MFMA + v_exp", 16, 32);
    run<1, 1>("MFMA + MFMA", 16, 16);
    return 0;
}

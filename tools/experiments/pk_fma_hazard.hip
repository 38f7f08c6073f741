// Stand-alone reproducer attempt for DESIGN.md section 9 ("packed-fp32 instructions and concurrently running kernels"),
// asked for by the round-4 review: NO library code -- two trivial kernels on two HIP streams.
//
//   victim     : an LDS-staged tile, one barrier, then a dependent chain of fused multiply-adds per thread on two adjacent
//                registers -- either as v_pk_fma_f32 (packed, one instruction for the pair) or as two v_fma_f32; result stored.
//                (The shape of begin_conv_kernel, the smallest victim section 9 names: LDS tile, weights in registers, no atomics.)
//   aggressor  : waves that issue v_mfma_f32_32x32x16_bf16 back to back, optionally with v_pk_fma_f32 / v_pk_mul_f32 between them
//                (the shape of the Winograd kernel's K loop), long enough to overlap many victim launches.
// Every victim launch is compared BIT FOR BIT with the victim's own solo result.  The program prints, per (victim form, aggressor
// form), how many of the launches differed.  Build and run on the GPU box:
//     hipcc --offload-arch=gfx950 -O2 -o /tmp/pk_fma_hazard tools/experiments/pk_fma_hazard.hip && /tmp/pk_fma_hazard [rounds]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// ---- victim: out[i] = chain of N fmas over an LDS-staged tile -------------------------------------------------------------
template <bool PACKED>
__global__ __launch_bounds__(256) void victim_kernel(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ out, int n_chain) {
    __shared__ float2 tile[256 + 2];
    const int tid = threadIdx.x, g = blockIdx.x * 256 + tid;
    tile[tid + 1] = make_float2(2.f * x[2 * g] - 1.f, 2.f * x[2 * g + 1] - 1.f);
    if (tid == 0) { tile[0] = make_float2(0.f, 0.f); tile[257] = make_float2(0.f, 0.f); }
    float wr[18];
#pragma unroll
    for (int k = 0; k < 18; ++k) wr[k] = w[k];
    __syncthreads();
    f32x2 acc = {0.25f, -0.5f};
    for (int it = 0; it < n_chain; ++it) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float2 v = tile[tid + k];
            const f32x2 vv = {v.x, v.y};
            const f32x2 w0 = {wr[6 * k], wr[6 * k + 1]}, w1 = {wr[6 * k + 2], wr[6 * k + 3]}, w2 = {wr[6 * k + 4], wr[6 * k + 5]};
            if (PACKED) {
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(w0), "v"(vv));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(w1), "v"(vv));
                asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(acc) : "v"(w2));
            } else {
                float a0 = acc[0], a1 = acc[1];
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a0) : "v"(w0[0]), "v"(vv[0]));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a1) : "v"(w0[1]), "v"(vv[1]));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a0) : "v"(w1[0]), "v"(vv[0]));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a1) : "v"(w1[1]), "v"(vv[1]));
                asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a0) : "v"(w2[0]));
                asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a1) : "v"(w2[1]));
                acc[0] = a0; acc[1] = a1;
            }
        }
    }
    out[2 * g] = acc[0];
    out[2 * g + 1] = acc[1];
}

// ---- aggressor: MFMA stream, optionally with packed-fp32 arithmetic between the matrix instructions ---------------------------
template <int MODE>   // 0: MFMA only, 1: MFMA + v_pk_fma_f32 / v_pk_mul_f32, 2: packed arithmetic only
__global__ __launch_bounds__(256) void aggressor_kernel(float* __restrict__ sink, int iters) {
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = (float)(threadIdx.x & 7) * 0.125f;
    bf16x8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.5f + 0.01f * i); b[i] = (__bf16)(0.25f - 0.01f * i); }
    f32x2 p = {1.0f, 0.5f}, q = {0.999f, 1.001f}, r = {1e-3f, -1e-3f};
    for (int it = 0; it < iters; ++it) {
        if (MODE != 2) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
        if (MODE != 0) {
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p) : "v"(q), "v"(r));
            asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p) : "v"(q));
            asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p) : "v"(r));
        }
        if (MODE != 2) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, acc, 0, 0, 0);
    }
    float s = p[0] + p[1];
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i];
    if (s == 123.456f) sink[0] = s;        // never true: keeps the work alive
}

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 200;
    const int NB = 1024, N = NB * 256;                       // victim: 1024 workgroups
    std::vector<float> hx(2 * N), hw(18);
    unsigned sd = 12345u;
    auto rnd = [&]() { sd = sd * 1664525u + 1013904223u; return (float)((sd >> 8) & 0xffff) / 65536.f; };
    for (auto& v : hx) v = rnd();
    for (auto& v : hw) v = 0.9f + 0.2f * rnd();
    float *dx, *dw, *dout, *dsink;
    CHECK(hipMalloc(&dx, hx.size() * 4)); CHECK(hipMalloc(&dw, 18 * 4)); CHECK(hipMalloc(&dout, 2 * N * 4)); CHECK(hipMalloc(&dsink, 64));
    CHECK(hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dw, hw.data(), 18 * 4, hipMemcpyHostToDevice));
    hipStream_t sa, sb;
    CHECK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    CHECK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    std::vector<float> ref(2 * N), got(2 * N);
    const char* vname[2] = {"victim: v_pk_fma_f32", "victim: v_fma_f32 x 2"};
    const char* aname[4] = {"no aggressor", "aggressor: MFMA only", "aggressor: MFMA + v_pk_*_f32", "aggressor: v_pk_*_f32 only"};
    int total_bad = 0;
    for (int v = 0; v < 2; ++v) {
        auto launch_victim = [&]() {
            if (v == 0) hipLaunchKernelGGL(victim_kernel<true>, dim3(NB), dim3(256), 0, sa, dx, dw, dout, 24);
            else hipLaunchKernelGGL(victim_kernel<false>, dim3(NB), dim3(256), 0, sa, dx, dw, dout, 24);
        };
        launch_victim();
        CHECK(hipStreamSynchronize(sa));
        CHECK(hipMemcpy(ref.data(), dout, 2 * N * 4, hipMemcpyDeviceToHost));
        for (int a = 0; a < 4; ++a) {
            int bad_launches = 0;
            long bad_words = 0;
            for (int r = 0; r < rounds; ++r) {
                CHECK(hipMemsetAsync(dout, 0xff, 2 * N * 4, sa));
                // the aggressor: enough workgroups to share every CU with the victim, ~300 us of work
                if (a == 1) hipLaunchKernelGGL(aggressor_kernel<0>, dim3(2048), dim3(256), 0, sb, dsink, 6000);
                if (a == 2) hipLaunchKernelGGL(aggressor_kernel<1>, dim3(2048), dim3(256), 0, sb, dsink, 6000);
                if (a == 3) hipLaunchKernelGGL(aggressor_kernel<2>, dim3(2048), dim3(256), 0, sb, dsink, 20000);
                for (int k = 0; k < 4; ++k) launch_victim();                 // four victim launches inside the aggressor's life
                CHECK(hipStreamSynchronize(sa));
                CHECK(hipMemcpy(got.data(), dout, 2 * N * 4, hipMemcpyDeviceToHost));
                CHECK(hipStreamSynchronize(sb));
                long d = 0;
                for (int i = 0; i < 2 * N; ++i) d += memcmp(&got[i], &ref[i], 4) != 0;
                bad_words += d;
                bad_launches += d != 0;
            }
            printf("%-24s | %-30s : %d of %d rounds differ from the solo run (%ld words)\n", vname[v], aname[a], bad_launches, rounds, bad_words);
            total_bad += bad_launches;
        }
    }
    printf("RESULT pk_fma_hazard: %s\n", total_bad ? "REPRODUCED (see the lines above)" : "not reproduced: every concurrent launch equals the solo run bit for bit");
    return 0;
}

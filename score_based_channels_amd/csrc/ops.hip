// The non-GEMM kernels of the annealed-Langevin hot path (gfx950): 2-channel begin/end convolutions on the
// vector ALU, InstanceNorm++ statistics, 5x5 max pooling, and the fused data-consistency gradient + Langevin
// update + NMSE kernel.  All activations NHWC float32; complex tensors are interleaved (re, im) pairs.
#include "tile.h"
#include "philox.h"

namespace sbc {

// ------------------------------------------------------------------------------------------------ begin conv
// h = 2x - 1 (ncsnv2.py:270-273), then begin_conv: Conv2d(2 -> COUT, 3x3, pad 1) + bias (ncsnv2.py:209,275).
// Zero padding applies to h, not x.  The kernel is bound by its 128-byte-per-pixel output stream, so the arithmetic
// has to stay out of the way: a workgroup owns a block of rows of one sample (256 pixels), stages h for those rows +
// 1 halo row/column (zeros outside the image) in LDS, and a thread owns one channel quad (its 72 weights live in
// registers) and walks 8 pixels 32 apart: every store instruction of a workgroup covers 32 consecutive pixels x
// 128 bytes, and the nine taps of a pixel are 8-byte LDS reads shared by the 8 threads of the pixel.
__global__ __launch_bounds__(256) void begin_conv_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ bias, float* __restrict__ out,
                                                          int H, int W, int rows_per_wg, float* __restrict__ pm_out) {
    constexpr int COUT = 32;
    extern __shared__ __attribute__((aligned(16))) float2 hs[];       // [(rows_per_wg + 2)][W + 2]
    const int tid = threadIdx.x, c4 = tid & 7, slot = tid >> 3;
    const int wgs_per_sample = H / rows_per_wg;
    const int n = blockIdx.x / wgs_per_sample, r0 = (blockIdx.x % wgs_per_sample) * rows_per_wg;
    const int WP = W + 2;
    for (int i = tid; i < (rows_per_wg + 2) * WP; i += 256) {
        const int rr = i / WP, cc = i - rr * WP;
        const int r = r0 - 1 + rr, c = cc - 1;
        float2 v = make_float2(0.f, 0.f);                            // zero padding applies to h = 2x - 1
        if (r >= 0 && r < H && c >= 0 && c < W) {
            v = *reinterpret_cast<const float2*>(x + (((size_t)n * H + r) * W + c) * 2);
            v.x = 2.f * v.x - 1.f;
            v.y = 2.f * v.y - 1.f;
        }
        hs[i] = v;
    }
    // the 576 weights through LDS, transposed to [tap][co] (torch order [co][ci][kh][kw]): three coalesced loads per thread
    // and 18 ds_read_b128 instead of 72 scattered global loads per thread -- with one 128-pixel tile per workgroup those
    // loads, not the 223 MB the launch writes, set its time (82 us; the vector-memory address unit, 288 wave-loads per tile)
    __shared__ __attribute__((aligned(16))) float wl[18 * COUT];
    for (int i = tid; i < 18 * COUT; i += 256) wl[(i % 18) * COUT + i / 18] = w[i];
    const float4 b4 = *reinterpret_cast<const float4*>(bias + c4 * 4);
    __syncthreads();
    float4 wr[18];                                               // [ci*9 + kh*3 + kw] for channels c4*4 .. c4*4+3
#pragma unroll
    for (int k = 0; k < 18; ++k) wr[k] = *reinterpret_cast<const float4*>(wl + k * COUT + c4 * 4);
    const int npx = rows_per_wg * W;
    float* obase = out + ((size_t)n * H + r0) * W * COUT + c4 * 4;
    auto pixel = [&](int lp) {
        const int rr = lp / W, cc = lp - rr * W;                     // local row / column
        float4 acc = b4;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const float2 v = hs[(rr + kh) * WP + cc + kw];
                const float4 w0 = wr[kh * 3 + kw], w1 = wr[9 + kh * 3 + kw];
                acc.x = fmaf(w0.x, v.x, fmaf(w1.x, v.y, acc.x));
                acc.y = fmaf(w0.y, v.x, fmaf(w1.y, v.y, acc.y));
                acc.z = fmaf(w0.z, v.x, fmaf(w1.z, v.y, acc.z));
                acc.w = fmaf(w0.w, v.x, fmaf(w1.w, v.y, acc.w));
            }
        }
        st_stream(obase + (size_t)lp * COUT, acc);
        return acc;
    };
    if (pm_out) {
        // SBC_EPI_MOMENTS_OUT: the workgroup is one 128-pixel tile (the host chose rows_per_wg * W == 128); every thread
        // keeps its four outputs and the tile's per-channel (mean, M2) go to pm_out[tile][32][2] (tile.h)
        __shared__ __attribute__((aligned(16))) float red[8 * 8 * 8];
        float4 yk[4];
#pragma unroll
        for (int it = 0; it < 4; ++it) yk[it] = pixel(slot + 32 * it);
        tile_moments_out32(yk, red, pm_out + (size_t)blockIdx.x * COUT * 2, tid);
        return;
    }
#pragma unroll 2
    for (int lp = slot; lp < npx; lp += 32) pixel(lp);
}

int launch_begin_conv(const sbc_op& op, hipStream_t stream) {
    SBC_REQUIRE(op.in && op.out && op.weight && op.bias, "begin_conv: in/out/weight/bias must be set");
    SBC_REQUIRE(op.cin == 2 && op.cout == 32, "begin_conv: cin=%d cout=%d (kernel is built for 2 -> 32)", op.cin, op.cout);
    // rows per workgroup: ~256 pixels, a divisor of H (exactly 128 pixels when the tile moments are wanted)
    const bool moments = (op.flags & SBC_EPI_MOMENTS_OUT) != 0;
    SBC_REQUIRE(!moments || (op.aux && 128 % op.W == 0 && op.H % (128 / op.W) == 0), "begin_conv: EPI_MOMENTS_OUT needs aux and whole 128-pixel tiles (rows of 128 / W pixels)");
    int rows = (moments ? 128 : 256) / op.W > 0 ? (moments ? 128 : 256) / op.W : 1;
    if (rows > op.H) rows = op.H;
    while (op.H % rows) --rows;
    const size_t lds = (size_t)(rows + 2) * (op.W + 2) * sizeof(float2);
    SBC_REQUIRE(lds <= 64 * 1024, "begin_conv: image row of %d pixels too wide", op.W);
    hipLaunchKernelGGL(begin_conv_kernel, dim3(op.B * (op.H / rows)), dim3(256), lds, stream, (const float*)op.in,
                       (const float*)op.weight, (const float*)op.bias, (float*)op.out, op.H, op.W, rows,
                       moments ? (float*)op.aux : (float*)nullptr);
    SBC_CHECK_HIP(hipGetLastError());
    return SBC_OK;
}

// ------------------------------------------------------------------------------------------------ IN++ stats
// InstanceNorm2dPlus (normalization.py:163-176), one workgroup per sample:
//   mu_c = mean_HW(x), var_c = biased variance, m = mean_C(mu), v = unbiased var_C(mu),
//   out = gamma * ((x - mu_c)/sqrt(var_c + 1e-5) + (mu_c - m)/sqrt(v + 1e-5) * alpha) + beta
// is stored as (mu, scale = gamma * rstd, shift = gamma * mhat * alpha + beta) so consumers apply
// (x - mu) * scale + shift while staging.
// The tensor is read ONCE: every thread accumulates sum and sum of squares of (x - pivot) with its own first value as
// pivot (so the squares stay at variance scale), turns them into (count, mean, M2), and the partials are merged with
// the exact pairwise update of Chan et al. (delta = mean_b - mean_a; M2 = M2_a + M2_b + delta^2 n_a n_b / n).
struct Moments4 { float4 mean, m2; };

__device__ __forceinline__ void chan_merge(float4& mean_a, float4& m2_a, float na, float4 mean_b, float4 m2_b, float nb) {
    const float n = na + nb, wb = nb / n, wab = na * nb / n;
    float d;
    d = mean_b.x - mean_a.x; mean_a.x += d * wb; m2_a.x += m2_b.x + d * d * wab;
    d = mean_b.y - mean_a.y; mean_a.y += d * wb; m2_a.y += m2_b.y + d * d * wab;
    d = mean_b.z - mean_a.z; mean_a.z += d * wb; m2_a.z += m2_b.z + d * d * wab;
    d = mean_b.w - mean_a.w; mean_a.w += d * wb; m2_a.w += m2_b.w + d * d * wab;
}

template <int C>
__global__ __launch_bounds__(256) void inorm_stats_kernel(const float* __restrict__ x, const float* __restrict__ agb,
                                                           float* __restrict__ stats, int HW) {
    constexpr int C4 = C / 4;
    constexpr int J = 256 / C4;              // pixel lanes per channel quad
    __shared__ float4 red_mean[256], red_m2[256];
    __shared__ float red_n[256];
    __shared__ float mean_s[C], var_s[C];
    const int n = blockIdx.x, tid = threadIdx.x;
    const int c4 = tid % C4, j = tid / C4;
    const float* base = x + (size_t)n * HW * C + c4 * 4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f), q = s, pivot = s;
    int cnt = 0;
    if (j < HW) pivot = *reinterpret_cast<const float4*>(base + (size_t)j * C);
    constexpr int UNR = 8;                   // independent 16-byte loads in flight per thread
    for (int px0 = j; px0 < HW; px0 += J * UNR) {
        float4 v[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u)
            if (px0 + u * J < HW) v[u] = *reinterpret_cast<const float4*>(base + (size_t)(px0 + u * J) * C);
#pragma unroll
        for (int u = 0; u < UNR; ++u)
            if (px0 + u * J < HW) {
                const float dx = v[u].x - pivot.x, dy = v[u].y - pivot.y, dz = v[u].z - pivot.z, dw = v[u].w - pivot.w;
                s.x += dx; s.y += dy; s.z += dz; s.w += dw;
                q.x = fmaf(dx, dx, q.x); q.y = fmaf(dy, dy, q.y); q.z = fmaf(dz, dz, q.z); q.w = fmaf(dw, dw, q.w);
                ++cnt;
            }
    }
    const float fn = (float)cnt, inv_n = cnt ? 1.f / fn : 0.f;
    float4 mean = make_float4(pivot.x + s.x * inv_n, pivot.y + s.y * inv_n, pivot.z + s.z * inv_n, pivot.w + s.w * inv_n);
    float4 m2 = make_float4(q.x - s.x * s.x * inv_n, q.y - s.y * s.y * inv_n, q.z - s.z * s.z * inv_n, q.w - s.w * s.w * inv_n);
    red_mean[tid] = mean; red_m2[tid] = m2; red_n[tid] = fn;
    __syncthreads();
    for (int st = J / 2; st > 0; st >>= 1) {
        if (j < st) {
            const float nb = red_n[tid + st * C4];
            if (nb > 0.f) {
                float4 ma = red_mean[tid], qa = red_m2[tid];
                const float na = red_n[tid];
                if (na > 0.f) chan_merge(ma, qa, na, red_mean[tid + st * C4], red_m2[tid + st * C4], nb);
                else { ma = red_mean[tid + st * C4]; qa = red_m2[tid + st * C4]; }
                red_mean[tid] = ma; red_m2[tid] = qa; red_n[tid] = na + nb;
            }
        }
        __syncthreads();
    }
    if (j == 0) {
        const float4 m = red_mean[tid], qq = red_m2[tid];
        const float inv = 1.f / (float)HW;
        mean_s[c4 * 4 + 0] = m.x; mean_s[c4 * 4 + 1] = m.y; mean_s[c4 * 4 + 2] = m.z; mean_s[c4 * 4 + 3] = m.w;
        var_s[c4 * 4 + 0] = qq.x * inv; var_s[c4 * 4 + 1] = qq.y * inv;
        var_s[c4 * 4 + 2] = qq.z * inv; var_s[c4 * 4 + 3] = qq.w * inv;
    }
    __syncthreads();
    if (tid < C) {
        float m = 0.f;
#pragma unroll 8
        for (int c = 0; c < C; ++c) m += mean_s[c];
        m *= 1.f / (float)C;
        float v = 0.f;
#pragma unroll 8
        for (int c = 0; c < C; ++c) { const float d = mean_s[c] - m; v = fmaf(d, d, v); }
        v *= 1.f / (float)(C - 1);
        const float mhat = (mean_s[tid] - m) / sqrtf(v + 1e-5f);
        const float rstd = 1.f / sqrtf(fmaxf(var_s[tid], 0.f) + 1e-5f);
        const float alpha = agb[tid], gamma = agb[C + tid], beta = agb[2 * C + tid];
        float* o = stats + (size_t)n * 3 * C;
        o[tid] = mean_s[tid];
        o[C + tid] = gamma * rstd;
        o[2 * C + tid] = fmaf(gamma, mhat * alpha, beta);
    }
}

// SBC_PRO_NORM_MOMENTS: the same (mu, scale, shift) from the TILE MOMENTS a producing launch left (SBC_EPI_MOMENTS_OUT:
// pm[B][NT][C][2] = (mean, M2) of each 128-pixel tile) instead of from the tensor itself -- a read of NT * C * 8 bytes per
// sample where the statistics launch above reads H * W * C * 4.  Equal tile counts: mean = (1 / NT) sum mean_t and
// M2 = sum M2_t + 128 sum (mean_t - mean)^2, both in a fixed order (256 / C partial sums per channel over interleaved tiles, combined
// in ascending order): reproducible bit for bit and independent of what else is in the batch.
template <int C>
__global__ __launch_bounds__(256) void inorm_from_moments_kernel(const float* __restrict__ pm, const float* __restrict__ agb,
                                                                  float* __restrict__ stats, int NT, int HW) {
    static_assert(C == 32 || C == 64, "256 / C tile groups x C channels per workgroup");
    constexpr int G = 256 / C;                                  // tile groups (8 at 32 channels, 4 at 64)
    __shared__ float part[G][C];
    __shared__ float mean_s[C], var_s[C];
    const int n = blockIdx.x, tid = threadIdx.x, c = tid & (C - 1), grp = tid / C;
    const float2* src = reinterpret_cast<const float2*>(pm + (size_t)n * NT * C * 2) + c;
    float sm = 0.f, sq = 0.f;
    for (int t = grp; t < NT; t += G) { const float2 v = src[(size_t)t * C]; sm += v.x; sq += v.y; }
    part[grp][c] = sm;
    __syncthreads();
    float mean = 0.f;
#pragma unroll
    for (int g = 0; g < G; ++g) mean += part[g][c];
    mean *= 1.f / (float)NT;
    __syncthreads();
    float dd = 0.f;
    for (int t = grp; t < NT; t += G) { const float d = src[(size_t)t * C].x - mean; dd = fmaf(d, d, dd); }
    part[grp][c] = fmaf(128.f, dd, sq);
    __syncthreads();
    if (grp == 0) {
        float m2 = 0.f;
#pragma unroll
        for (int g = 0; g < G; ++g) m2 += part[g][c];
        mean_s[c] = mean;
        var_s[c] = m2 * (1.f / (float)HW);
    }
    __syncthreads();
    if (tid < C) {                                            // the "++" part, as in inorm_stats_kernel
        float m = 0.f;
#pragma unroll 8
        for (int k = 0; k < C; ++k) m += mean_s[k];
        m *= 1.f / (float)C;
        float v = 0.f;
#pragma unroll 8
        for (int k = 0; k < C; ++k) { const float d = mean_s[k] - m; v = fmaf(d, d, v); }
        v *= 1.f / (float)(C - 1);
        const float mhat = (mean_s[tid] - m) / sqrtf(v + 1e-5f);
        const float rstd = 1.f / sqrtf(fmaxf(var_s[tid], 0.f) + 1e-5f);
        const float alpha = agb[tid], gamma = agb[C + tid], beta = agb[2 * C + tid];
        float* o = stats + (size_t)n * 3 * C;
        o[tid] = mean_s[tid];
        o[C + tid] = gamma * rstd;
        o[2 * C + tid] = fmaf(gamma, mhat * alpha, beta);
    }
}

int launch_inorm_stats(const sbc_op& op, hipStream_t stream) {
    SBC_REQUIRE(op.in && op.out && op.weight, "inorm_stats: in/out/weight must be set");
    if (op.flags & SBC_PRO_NORM_MOMENTS) {
        SBC_REQUIRE((op.cin == 32 || op.cin == 64) && (op.H * op.W) % 128 == 0,
                    "inorm_stats: tile moments exist for 32 / 64 channels and whole 128-pixel tiles");
        if (op.cin == 32)
            hipLaunchKernelGGL(inorm_from_moments_kernel<32>, dim3(op.B), dim3(256), 0, stream, (const float*)op.in,
                               (const float*)op.weight, (float*)op.out, op.H * op.W / 128, op.H * op.W);
        else
            hipLaunchKernelGGL(inorm_from_moments_kernel<64>, dim3(op.B), dim3(256), 0, stream, (const float*)op.in,
                               (const float*)op.weight, (float*)op.out, op.H * op.W / 128, op.H * op.W);
        SBC_CHECK_HIP(hipGetLastError());
        return SBC_OK;
    }
    const int HW = op.H * op.W;
    const float* x = (const float*)op.in;
    const float* agb = (const float*)op.weight;
    float* st = (float*)op.out;
    switch (op.cin) {
        case 32: hipLaunchKernelGGL(inorm_stats_kernel<32>, dim3(op.B), dim3(256), 0, stream, x, agb, st, HW); break;
        case 64: hipLaunchKernelGGL(inorm_stats_kernel<64>, dim3(op.B), dim3(256), 0, stream, x, agb, st, HW); break;
        case 128: hipLaunchKernelGGL(inorm_stats_kernel<128>, dim3(op.B), dim3(256), 0, stream, x, agb, st, HW); break;
        default: set_error("inorm_stats: %d channels (only 32/64/128)", op.cin); return SBC_ERR_UNSUPPORTED;
    }
    SBC_CHECK_HIP(hipGetLastError());
    return SBC_OK;
}

// ------------------------------------------------------------------------------------------------ max pool
// nn.MaxPool2d(5, stride 1, padding 2) with -inf padding (layers.py:69).  With SBC_PRO_ELU the result is
// ELU(max) == max(ELU) because ELU is monotone (CRPBlock: x = act(x); path = maxpool(x), layers.py:77-80).
// One thread per (sample, row segment, column w, channel quad) walks down its rows with a sliding window of 5 row
// maxima, so each output costs ~7 (L1-resident) 16-byte loads instead of 25.
__device__ __forceinline__ float4 max4(float4 a, float4 b) {
    return make_float4(fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z), fmaxf(a.w, b.w));
}

__global__ __launch_bounds__(256) void maxpool5_kernel(const float* __restrict__ in, float* __restrict__ out, int B,
                                                        int H, int W, int C4, int flags, int seg, int nseg) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= B * nseg * W * C4) return;
    const int c4 = idx % C4, w = (idx / C4) % W, sg = (idx / (C4 * W)) % nseg, n = idx / (C4 * W * nseg);
    const size_t row_stride = (size_t)W * C4 * 4;
    const float* base = in + (size_t)n * H * row_stride + ((size_t)w * C4 + c4) * 4;
    float* obase = out + (size_t)n * H * row_stride + ((size_t)w * C4 + c4) * 4;
    const float NEG = -INFINITY;
    const float4 ninf = make_float4(NEG, NEG, NEG, NEG);
    const bool l2 = w >= 2, l1 = w >= 1, r1 = w + 1 < W, r2 = w + 2 < W;
    auto rowmax = [&](int r) {
        if (r < 0 || r >= H) return ninf;
        const float* q = base + (size_t)r * row_stride;
        float4 m = *reinterpret_cast<const float4*>(q);
        if (l1) m = max4(m, *reinterpret_cast<const float4*>(q - C4 * 4));
        if (l2) m = max4(m, *reinterpret_cast<const float4*>(q - 2 * C4 * 4));
        if (r1) m = max4(m, *reinterpret_cast<const float4*>(q + C4 * 4));
        if (r2) m = max4(m, *reinterpret_cast<const float4*>(q + 2 * C4 * 4));
        return m;
    };
    // this thread's rows [h0, h1): a sliding window of the row maxima h-2 .. h+2
    const int h0 = sg * seg, h1 = min(h0 + seg, H);
    float4 m0 = rowmax(h0 - 2), m1 = rowmax(h0 - 1), m2 = rowmax(h0), m3 = rowmax(h0 + 1), m4 = rowmax(h0 + 2);
    for (int h = h0; h < h1; ++h) {
        float4 m = max4(max4(max4(m0, m1), max4(m2, m3)), m4);
        if (flags & SBC_PRO_ELU) m = elu4_acc(m);
        st_stream(obase + (size_t)h * row_stride, m);
        m0 = m1; m1 = m2; m2 = m3; m3 = m4;
        if (h + 1 < h1) m4 = rowmax(h + 3);
    }
}

// Separable variant for images at least 16 rows high (the 64x16 and 32x8 levels, and every level of a 256x64 array, where
// the tensor is far larger than the caches): a workgroup owns 16 rows x up to 16 columns of one sample.  Pass 1: thread (row, quad) reads its
// row straight from global memory (every element once per workgroup, 128-byte segments) and writes the horizontal
// 5-max to LDS; pass 2: thread (row group, column, quad) slides the vertical 5-window over LDS and stores.  1.25 global
// loads per output instead of ~7 through the vector cache.
template <int WMAX>
__global__ __launch_bounds__(256) void maxpool5_rows_kernel(const float* __restrict__ in, float* __restrict__ out, int H,
                                                             int W, int C4, int flags, int wtiles) {
    // a workgroup owns R rows x WT = min(W, WMAX) columns of one sample; wider images are tiled along W with a
    // two-column halo on each side (wtiles column tiles per row block)
    constexpr int R = 16, RH = R + 4;
    extern __shared__ __attribute__((aligned(16))) float4 hm[];       // [RH][WT][C4] horizontal maxima
    const int tid = threadIdx.x;
    const int tiles_per_sample = (H / R) * wtiles;
    const int n = blockIdx.x / tiles_per_sample, rem = blockIdx.x % tiles_per_sample;
    const int r0 = (rem / wtiles) * R, w0 = (rem % wtiles) * WMAX;
    const int WT = min(WMAX, W - w0);
    const size_t row_stride = (size_t)W * C4 * 4;
    const float* base = in + (size_t)n * H * row_stride;
    const float NEG = -INFINITY;
    const float4 ninf = make_float4(NEG, NEG, NEG, NEG);
    // pass 1: RH rows x C4 quads
    for (int t = tid; t < RH * C4; t += 256) {
        const int c4 = t % C4, rr = t / C4, r = r0 - 2 + rr;
        float4 x[WMAX + 4];                                            // x[k] = column w0 - 2 + k
#pragma unroll
        for (int k = 0; k < WMAX + 4; ++k) x[k] = ninf;
        if (r >= 0 && r < H) {
            const float* q = base + (size_t)r * row_stride + c4 * 4;
#pragma unroll
            for (int k = 0; k < WMAX + 4; ++k) {
                const int col = w0 - 2 + k;
                if (col >= 0 && col < W && k < WT + 4) x[k] = *reinterpret_cast<const float4*>(q + (size_t)col * C4 * 4);
            }
        }
#pragma unroll
        for (int w = 0; w < WMAX; ++w)
            if (w < WT) hm[(rr * WT + w) * C4 + c4] = max4(max4(max4(x[w], x[w + 1]), max4(x[w + 2], x[w + 3])), x[w + 4]);
    }
    __syncthreads();
    // pass 2: (row group of 4, column, quad)
    float* obase = out + (size_t)n * H * row_stride;
    for (int t = tid; t < (R / 4) * WT * C4; t += 256) {
        const int c4 = t % C4, w = (t / C4) % WT, g = t / (C4 * WT);
        const float4* col = hm + w * C4 + c4;
        const int rr0 = g * 4;                                            // hm row rr corresponds to image row r0 - 2 + rr
        float4 m0 = col[(rr0 + 0) * WT * C4], m1 = col[(rr0 + 1) * WT * C4], m2 = col[(rr0 + 2) * WT * C4],
               m3 = col[(rr0 + 3) * WT * C4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float4 m4 = col[(rr0 + 4 + k) * WT * C4];
            float4 m = max4(max4(max4(m0, m1), max4(m2, m3)), m4);
            if (flags & SBC_PRO_ELU) m = elu4_acc(m);
            st_stream(obase + (size_t)(r0 + rr0 + k) * row_stride + ((size_t)(w0 + w) * C4 + c4) * 4, m);
            m0 = m1; m1 = m2; m2 = m3; m3 = m4;
        }
    }
}

int launch_maxpool5(const sbc_op& op, hipStream_t stream) {
    SBC_REQUIRE(op.in && op.out && op.cin % 4 == 0, "maxpool5: in/out must be set, channels %% 4 == 0");
    const int C4 = op.cin / 4;
    const int wt = op.W <= 8 ? op.W : 16;                              // column tile
    if (op.H % 16 == 0 && (op.W <= 16 || op.W % 16 == 0) && (size_t)20 * wt * C4 * 16 <= 64 * 1024) {
        const size_t lds = (size_t)20 * wt * C4 * sizeof(float4);
        const int wtiles = (op.W + wt - 1) / wt;
        const int grid = op.B * (op.H / 16) * wtiles;
        if (op.W <= 8)
            hipLaunchKernelGGL(maxpool5_rows_kernel<8>, dim3(grid), dim3(256), lds, stream, (const float*)op.in,
                               (float*)op.out, op.H, op.W, C4, op.flags, wtiles);
        else
            hipLaunchKernelGGL(maxpool5_rows_kernel<16>, dim3(grid), dim3(256), lds, stream, (const float*)op.in,
                               (float*)op.out, op.H, op.W, C4, op.flags, wtiles);
        SBC_CHECK_HIP(hipGetLastError());
        return SBC_OK;
    }
    // rows are walked in segments of 8 (4 extra row maxima per segment) so that tall images still give every CU
    // tens of waves; images of <= 8 rows are one segment
    const int seg = 8, nseg = (op.H + seg - 1) / seg;
    const long total = (long)op.B * nseg * op.W * C4;
    hipLaunchKernelGGL(maxpool5_kernel, dim3((int)((total + 255) / 256)), dim3(256), 0, stream, (const float*)op.in,
                       (float*)op.out, op.B, op.H, op.W, C4, op.flags, seg, nseg);
    SBC_CHECK_HIP(hipGetLastError());
    return SBC_OK;
}

// ------------------------------------------------------------------------------------------------ end conv
// normalizer (InstanceNorm++) -> ELU -> end_conv: Conv2d(CIN -> 2, 3x3, pad 1) + bias -> / sigma
// (ncsnv2.py:291-298).  256-pixel tiles staged like the MFMA conv; one thread per output pixel (both channels),
// weights [tap][c][2] in LDS (broadcast reads).
template <int CIN>
__global__ __launch_bounds__(256) void end_conv_kernel(const float* __restrict__ in, const float* __restrict__ stats,
                                                        const float* __restrict__ w, const float* __restrict__ bias,
                                                        float* __restrict__ out, sbc_endconv e, int B, int H, int W) {
    constexpr int TM = 256, S = CIN + 4;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const Dims<false> d{H, W, H * W, 0, 0};
    const TileGeom g = tile_geom(blockIdx.x, TM, B, d, 1);
    if (!g.multi) {
        // one sample per tile: the thread's (mu, scale, shift) in registers instead of three loads per staged chunk
        float4 pf[9];
        stage_issue<CIN, 256, 9>(pf, in, g, W, tid);
        const RegStats rs = load_reg_stats<CIN, 256>(stats, g, tid);
        stage_commit_reg<CIN, 256, 9>(lds, pf, in, rs, SBC_PRO_NORM | SBC_PRO_ELU, g, W, tid);
    } else {
        stage_tile<CIN, 256, 9, false>(lds, in, stats, SBC_PRO_NORM | SBC_PRO_ELU, g, d, tid);
    }
    __syncthreads();
    const int px = g.p0 + tid;
    if (px >= B * H * W) return;
    const int row = px / W, wq = px - row * W, h = row % H;
    // LDS offsets of the nine taps of this pixel (the zero pixel for taps outside the image)
    int toff[9];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int hh = h + tap / 3 - 1, ww = wq + tap % 3 - 1;
        const bool ok = hh >= 0 && hh < H && ww >= 0 && ww < W;
        toff[tap] = (ok ? (row + tap / 3 - 1 - g.rs0) * W + ww : g.nps) * S;
    }
    // The weights never enter LDS or vector registers: every index below is uniform, so the 72 weights of a channel quad
    // (torch [2][CIN][3][3]: the nine taps of a channel are contiguous) arrive by scalar loads and feed the FMAs as SGPR
    // operands.  (Round 2 kept them in LDS as [tap][c][2]: 16 broadcast ds_read_b128 per tap beside the 8 that fetch the
    // activations -- two thirds of the kernel's LDS instructions.)
    float a0 = 0.f, a1 = 0.f;
#pragma unroll 1
    for (int cb = 0; cb < CIN; cb += 4) {
        const float* w0 = w + cb * 9;                   // output 0, channels cb .. cb + 3
        const float* w1 = w + (CIN + cb) * 9;           // output 1
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const float4 v = *reinterpret_cast<const float4*>(lds + toff[tap] + cb);
            a0 = fmaf(v.x, w0[tap], a0);      a1 = fmaf(v.x, w1[tap], a1);
            a0 = fmaf(v.y, w0[9 + tap], a0);  a1 = fmaf(v.y, w1[9 + tap], a1);
            a0 = fmaf(v.z, w0[18 + tap], a0); a1 = fmaf(v.z, w1[18 + tap], a1);
            a0 = fmaf(v.w, w0[27 + tap], a0); a1 = fmaf(v.w, w1[27 + tap], a1);
        }
    }
    const int n = px / (H * W);
    const float sigma = e.labels ? e.sigmas[e.labels[n]] : e.sigma_of_step[*e.step];
    float2 o;
    o.x = (a0 + bias[0]) / sigma;
    o.y = (a1 + bias[1]) / sigma;
    *reinterpret_cast<float2*>(out + (size_t)px * 2) = o;
}

// The same tail with the normalizer's statistics formed IN the launch (SBC_PRO_NORM_SELF; round 5): a persistent 8-wave workgroup per
// CU owns whole samples.  The statistics launch read the 223 MB tensor once (40 us at 1700 x 64 x 16) and this kernel read it again;
// here a sample is read ONCE into registers (16 float4 a thread, requested while the previous sample's convolution runs), its
// InstanceNorm++ statistics (normalization.py:163-176: per-channel mean and biased variance over the pixels, two passes; mean and
// UNBIASED variance of the channel means) are formed from the registers -- in-thread, three xor shuffles over the lanes that share a
// channel quad, the eight waves through LDS, all in a fixed order --, the normalised + ELU'd sample goes to LDS ([pixel][CIN + 4]
// floats: 16 consecutive pixels hit 16 different 16-byte slots), and every thread convolves its two output pixels from there with the
// weights as scalar operands (end_conv_kernel's inner loop, two pixels sharing each scalar).
template <int CIN, int NTH>
__global__ __launch_bounds__(NTH) void end_conv_self_kernel(const float* __restrict__ in, const float* __restrict__ agb,
                                                             const float* __restrict__ w, const float* __restrict__ bias,
                                                             float* __restrict__ out, sbc_endconv e, int B, int H, int W) {
    constexpr int C4 = CIN / 4, S = CIN + 4, NWV = NTH / 64, PJ = NTH / C4;      // waves; pixels per round of chunks
    constexpr int NQ = 1024 * C4 / NTH, NPX = 1024 / NTH;                       // chunks / output pixels per thread at HW = 1024
    static_assert(CIN == 32, "eight channel quads: a thread keeps one quad for all its chunks");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int HW = 1024;                                                  // (H W = 1024: checked by the launcher -- a compile-time count
                                                                              // keeps the chunk loops free of branches and of full memory waits)
    float* const act = lds;                                                   // [HW + 1][S]: the activated sample + one zero pixel
    float* const red = lds + (HW + 1) * S;                                    // two buffers of [waves][CIN] partial sums
    float* const mean_s = red + 2 * NWV * CIN;                                // [CIN] channel means
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c4 = tid & (C4 - 1), j = tid >> 3;                              // channel quad; pixel k * PJ + j for chunk k
    const float inv_hw = 1.f / (float)HW;
    if (tid < S) act[HW * S + tid] = 0.f;                                     // the zero pixel (taps outside the image)
    const float4 al = *reinterpret_cast<const float4*>(agb + c4 * 4), ga = *reinterpret_cast<const float4*>(agb + CIN + c4 * 4),
                 be = *reinterpret_cast<const float4*>(agb + 2 * CIN + c4 * 4);
    const float b0 = bias[0], b1 = bias[1];
    float4 xv[NQ];
    auto request = [&](int n) {
        const float* base = in + ((size_t)n * HW + j) * CIN + c4 * 4;
#pragma unroll
        for (int k = 0; k < NQ; ++k)
            xv[k] = *reinterpret_cast<const float4*>(base + (size_t)k * PJ * CIN);
    };
    // sum over the threads that share a channel quad: lanes (bits 3..5), then the waves through `buf` -- every thread adds the
    // eight wave sums in the same order
    auto quad_total = [&](float4 t, float* buf) {
#pragma unroll
        for (int m = 8; m < 64; m <<= 1) {
            t.x += __shfl_xor(t.x, m); t.y += __shfl_xor(t.y, m); t.z += __shfl_xor(t.z, m); t.w += __shfl_xor(t.w, m);
        }
        if (lane < C4) *reinterpret_cast<float4*>(buf + wave * CIN + c4 * 4) = t;
        __syncthreads();
        float4 a = *reinterpret_cast<const float4*>(buf + c4 * 4);
#pragma unroll
        for (int wv = 1; wv < NWV; ++wv) {
            const float4 o = *reinterpret_cast<const float4*>(buf + wv * CIN + c4 * 4);
            a.x += o.x; a.y += o.y; a.z += o.z; a.w += o.w;
        }
        return a;
    };
    int n = blockIdx.x;
    if (n < B) request(n);
    for (; n < B; n += gridDim.x) {
        // the sample's noise level, requested FIRST: read where it is used (behind the convolution) its two dependent round trips were
        // exposed at the end of every sample, each behind a full memory wait that also covered the next sample's prefetch
        const float sigma = e.labels ? e.sigmas[e.labels[n]] : e.sigma_of_step[*e.step];
        // ---- statistics of the sample in the registers
        float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 0; k < NQ; ++k)
            { sum.x += xv[k].x; sum.y += xv[k].y; sum.z += xv[k].z; sum.w += xv[k].w; }
        sum = quad_total(sum, red);
        const float4 mu = make_float4(sum.x * inv_hw, sum.y * inv_hw, sum.z * inv_hw, sum.w * inv_hw);
        float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 0; k < NQ; ++k)
            {
                float d;
                d = xv[k].x - mu.x; q.x = fmaf(d, d, q.x); d = xv[k].y - mu.y; q.y = fmaf(d, d, q.y);
                d = xv[k].z - mu.z; q.z = fmaf(d, d, q.z); d = xv[k].w - mu.w; q.w = fmaf(d, d, q.w);
            }
        if (tid < C4) *reinterpret_cast<float4*>(mean_s + c4 * 4) = mu;
        q = quad_total(q, red + NWV * CIN);                                   // (its barrier also publishes mean_s)
        float m = 0.f;
#pragma unroll 8
        for (int c = 0; c < CIN; ++c) m += mean_s[c];
        m *= 1.f / (float)CIN;
        float v = 0.f;
#pragma unroll 8
        for (int c = 0; c < CIN; ++c) { const float d = mean_s[c] - m; v = fmaf(d, d, v); }
        const float rv = 1.f / sqrtf(v * (1.f / (float)(CIN - 1)) + 1e-5f);
        float4 sc, sh;                                                        // out = (x - mu) * sc + sh   (ops.hip: inorm_stats_kernel)
        sc.x = ga.x / sqrtf(fmaxf(q.x * inv_hw, 0.f) + 1e-5f); sh.x = fmaf(ga.x, (mu.x - m) * rv * al.x, be.x);
        sc.y = ga.y / sqrtf(fmaxf(q.y * inv_hw, 0.f) + 1e-5f); sh.y = fmaf(ga.y, (mu.y - m) * rv * al.y, be.y);
        sc.z = ga.z / sqrtf(fmaxf(q.z * inv_hw, 0.f) + 1e-5f); sh.z = fmaf(ga.z, (mu.z - m) * rv * al.z, be.z);
        sc.w = ga.w / sqrtf(fmaxf(q.w * inv_hw, 0.f) + 1e-5f); sh.w = fmaf(ga.w, (mu.w - m) * rv * al.w, be.w);
        // ---- normalise, ELU, to LDS; then the registers take the next sample
#pragma unroll
        for (int k = 0; k < NQ; ++k)
            {
                float4 y;
                y.x = (xv[k].x - mu.x) * sc.x + sh.x; y.y = (xv[k].y - mu.y) * sc.y + sh.y;
                y.z = (xv[k].z - mu.z) * sc.z + sh.z; y.w = (xv[k].w - mu.w) * sc.w + sh.w;
                *reinterpret_cast<float4*>(act + (k * PJ + j) * S + c4 * 4) = elu4(y);
            }
        if (n + (int)gridDim.x < B) request(n + gridDim.x);
        __syncthreads();
        // ---- end_conv: thread = output pixels tid, tid + NTH, ...
        int toff[NPX][9];
#pragma unroll
        for (int i = 0; i < NPX; ++i) {
            const int px = min(tid + i * NTH, HW - 1), h = px / W, wq = px - h * W;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int hh = h + tap / 3 - 1, ww = wq + tap % 3 - 1;
                toff[i][tap] = ((hh >= 0 && hh < H && ww >= 0 && ww < W) ? hh * W + ww : HW) * S;
            }
        }
        float a0[NPX], a1[NPX];
#pragma unroll
        for (int i = 0; i < NPX; ++i) { a0[i] = 0.f; a1[i] = 0.f; }
#pragma unroll 1
        for (int cb = 0; cb < CIN; cb += 4) {
            const float* w0 = w + cb * 9;                   // output 0, channels cb .. cb + 3 (uniform: scalar loads)
            const float* w1 = w + (CIN + cb) * 9;           // output 1
            // (the reads of a filter row first: left to itself hipcc waits for each one in front of its eight FMAs, and the LDS round trip
            // per read is not hidden)
#pragma unroll
            for (int tr = 0; tr < 3; ++tr) {
                float4 x4[NPX][3];
#pragma unroll
                for (int tq = 0; tq < 3; ++tq)
#pragma unroll
                    for (int i = 0; i < NPX; ++i) x4[i][tq] = *reinterpret_cast<const float4*>(act + toff[i][3 * tr + tq] + cb);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int tq = 0; tq < 3; ++tq) {
                    const int tap = 3 * tr + tq;
#pragma unroll
                    for (int i = 0; i < NPX; ++i) {
                        a0[i] = fmaf(x4[i][tq].x, w0[tap], a0[i]);      a1[i] = fmaf(x4[i][tq].x, w1[tap], a1[i]);
                        a0[i] = fmaf(x4[i][tq].y, w0[9 + tap], a0[i]);  a1[i] = fmaf(x4[i][tq].y, w1[9 + tap], a1[i]);
                        a0[i] = fmaf(x4[i][tq].z, w0[18 + tap], a0[i]); a1[i] = fmaf(x4[i][tq].z, w1[18 + tap], a1[i]);
                        a0[i] = fmaf(x4[i][tq].w, w0[27 + tap], a0[i]); a1[i] = fmaf(x4[i][tq].w, w1[27 + tap], a1[i]);
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < NPX; ++i) {
            {
                float2 o;
                o.x = (a0[i] + b0) / sigma;
                o.y = (a1[i] + b1) / sigma;
                *reinterpret_cast<float2*>(out + ((size_t)n * HW + tid + i * NTH) * 2) = o;
            }
        }
        // (the next sample's first statistics barrier keeps anyone from rewriting `act` before every thread is through here)
    }
}

int launch_end_conv(const sbc_op& op, const sbc_endconv& e, hipStream_t stream, bool dry) {
    SBC_REQUIRE(op.in && op.out && op.weight && op.bias && op.stats, "end_conv: in/out/weight/bias/stats must be set");
    SBC_REQUIRE(op.cout == 2, "end_conv: cout=%d", op.cout);
    SBC_REQUIRE((e.labels && e.sigmas) || (e.sigma_of_step && e.step), "end_conv: no noise-level source");
    if (op.flags & SBC_PRO_NORM_SELF) {
        // stats = the normalizer's (alpha | gamma | beta): the launch forms the statistics itself, a workgroup holds whole samples
        const int hw = op.H * op.W;
        SBC_REQUIRE(op.cin == 32 && hw == 1024,
                    "end_conv: SBC_PRO_NORM_SELF takes 32 channels and images of 1024 pixels (got %d channels, %dx%d)", op.cin, op.H, op.W);
        constexpr int NTH = 512;            // (1024 threads, one output pixel each: 128 registers a thread spill -- 212 us against 98)
        const size_t lds = ((size_t)(hw + 1) * (op.cin + 4) + 2 * (NTH / 64) * op.cin + op.cin) * sizeof(float);
        { auto k0 = end_conv_self_kernel<32, NTH>; const int rc = ensure_dyn_lds(reinterpret_cast<const void*>(k0), lds); if (rc) return rc; }
        if (dry) return SBC_OK;
        int dev = 0, cus = 256;
        SBC_CHECK_HIP(hipGetDevice(&dev));
        SBC_CHECK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        const int grid = balanced_sample_grid(op.B, cus);
        auto kern = end_conv_self_kernel<32, NTH>;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(NTH), lds, stream, (const float*)op.in, (const float*)op.stats,
                           (const float*)op.weight, (const float*)op.bias, (float*)op.out, e, op.B, op.H, op.W);
        SBC_CHECK_HIP(hipGetLastError());
        return SBC_OK;
    }
    const int HW = op.H * op.W, TM = 256;
    SBC_REQUIRE(TM % op.W == 0 && (HW % TM == 0 || TM % HW == 0), "end_conv: image %dx%d does not tile", op.H, op.W);
    const int total = op.B * HW;
    const int halo_px = TM >= HW ? 0 : 2 * op.W;
    const size_t lds = (size_t)(TM + halo_px + 1) * (op.cin + 4) * sizeof(float);
    SBC_REQUIRE(lds <= 160 * 1024, "end_conv: tile needs %zu bytes of LDS", lds);
    SBC_REQUIRE(op.cin == 32, "end_conv: %d input channels (only ngf = 32)", op.cin);
    { const int rc = ensure_dyn_lds(reinterpret_cast<const void*>(end_conv_kernel<32>), lds); if (rc) return rc; }
    if (dry) return SBC_OK;
    hipLaunchKernelGGL(end_conv_kernel<32>, dim3((total + TM - 1) / TM), dim3(256), lds, stream, (const float*)op.in,
                       (const float*)op.stats, (const float*)op.weight, (const float*)op.bias, (float*)op.out, e, op.B,
                       op.H, op.W);
    SBC_CHECK_HIP(hipGetLastError());
    return SBC_OK;
}

// ------------------------------------------------------------------------------------------------ Langevin
__device__ __forceinline__ float2 cmul(float2 a, float2 b) {          // a * b
    return make_float2(fmaf(a.x, b.x, -a.y * b.y), fmaf(a.x, b.y, a.y * b.x));
}
__device__ __forceinline__ float2 cfma(float2 a, float2 b, float2 c) {  // a * b + c
    return make_float2(fmaf(a.x, b.x, fmaf(-a.y, b.y, c.x)), fmaf(a.x, b.y, fmaf(a.y, b.x, c.y)));
}
__device__ __forceinline__ float2 cfma_conj(float2 a, float2 b, float2 c) {  // conj(a) * b + c
    return make_float2(fmaf(a.x, b.x, fmaf(a.y, b.y, c.x)), fmaf(a.x, b.y, fmaf(-a.y, b.x, c.y)));
}

// One workgroup per trajectory b:  R = P X - Y  [Np x Nr],  G = P^H R  [Nt x Nr]  (test_score.py:157-158),
// X <- X + alpha (S - G / dc_div) + noise_scale * n  (:160-165),  nmse[step][b] = |X - H|^2 / |H|^2  (:168-170).
// X and R live in LDS; P (per-sample pilots, L2 resident) is read through the vector cache: in both products the
// 16 lanes that share a pilot entry read the same address.
__global__ __launch_bounds__(256) void langevin_kernel(sbc_langevin a, int B, int x_in_lds, int p_in_lds) {
    extern __shared__ __attribute__((aligned(16))) float2 sm[];
    const int Nt = a.Nt, Nr = a.Nr, Np = a.Np;
    float2* Rs = sm;                   // [Np*Nr]
    float2* Xl = sm + Np * Nr;         // [Nt*Nr] when it fits (64x16: 8 KB); large arrays read X through L1/L2
    float2* Pl = Xl + Nt * Nr;         // [Np*Nt] when it fits (38x64: 19 KB): both products walk P (2 x 311 KB of 8-byte
                                       // vector-cache reads per trajectory otherwise, which made the kernel TA-bound)
    __shared__ float red[2][4];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int step = *a.step;
    float2* X = reinterpret_cast<float2*>(a.X) + (size_t)b * Nt * Nr;
    const float2* P = reinterpret_cast<const float2*>(a.P) + (size_t)(a.p_index ? a.p_index[b] : b) * Np * Nt;
    const float2* Y = reinterpret_cast<const float2*>(a.Y) + (size_t)b * Np * Nr;
    const float2* Ht = reinterpret_cast<const float2*>(a.Htrue) + (size_t)(a.h_index ? a.h_index[b] : b) * Nt * Nr;
    const float2* Sc = reinterpret_cast<const float2*>(a.score) + (size_t)b * Nt * Nr;
    // rows of P in LDS are Nt + 2 long: the first product reads pm[t] of four pilots m per wave, and Nt = 64 complex numbers are
    // a whole number of bank rows (all four on one bank, a 4-way conflict on every read)
    const int PS = p_in_lds ? Nt + 2 : Nt;
    if (p_in_lds) {
        const float4* src = reinterpret_cast<const float4*>(P);          // Nt is even: 16-byte copies that stay inside a row
        const int h = Nt / 2;
        for (int e = tid; e < Np * h; e += 256) {
            const int m = e / h, j = e - m * h;
            reinterpret_cast<float4*>(Pl + (size_t)m * PS)[j] = src[e];
        }
    }
    if (x_in_lds) {
        for (int e = tid; e < Nt * Nr; e += 256) Xl[e] = X[e];
    }
    if (x_in_lds || p_in_lds) __syncthreads();
    const float2* Xs = x_in_lds ? Xl : X;
    if (p_in_lds) P = Pl;
    if ((Nr & 3) == 0) {
        // four adjacent receive antennas per thread: one pilot value and two 16-byte reads of X feed four independent chains
        // (one output per thread was two 8-byte LDS reads per complex FMA and a single dependent chain of 4 Nt FMAs)
        for (int o = tid * 4; o < Np * Nr; o += 1024) {
            const int m = o / Nr, r = o - m * Nr;
            float2 c0 = make_float2(0.f, 0.f), c1 = c0, c2 = c0, c3 = c0;
            const float2* pm = P + (size_t)m * PS;
            for (int t = 0; t < Nt; ++t) {
                const float2 pv = pm[t];
                const float4 xa = *reinterpret_cast<const float4*>(Xs + t * Nr + r);
                const float4 xb = *reinterpret_cast<const float4*>(Xs + t * Nr + r + 2);
                c0 = cfma(pv, make_float2(xa.x, xa.y), c0);
                c1 = cfma(pv, make_float2(xa.z, xa.w), c1);
                c2 = cfma(pv, make_float2(xb.x, xb.y), c2);
                c3 = cfma(pv, make_float2(xb.z, xb.w), c3);
            }
            const float4 ya = *reinterpret_cast<const float4*>(Y + o), yb = *reinterpret_cast<const float4*>(Y + o + 2);
            *reinterpret_cast<float4*>(Rs + o) = make_float4(c0.x - ya.x, c0.y - ya.y, c1.x - ya.z, c1.y - ya.w);
            *reinterpret_cast<float4*>(Rs + o + 2) = make_float4(c2.x - yb.x, c2.y - yb.y, c3.x - yb.z, c3.y - yb.w);
        }
    } else {
        for (int o = tid; o < Np * Nr; o += 256) {
            const int m = o / Nr, r = o - m * Nr;
            float2 acc = make_float2(0.f, 0.f);
            const float2* pm = P + (size_t)m * PS;
            for (int t = 0; t < Nt; ++t) acc = cfma(pm[t], Xs[t * Nr + r], acc);
            const float2 y = Y[o];
            Rs[o] = make_float2(acc.x - y.x, acc.y - y.y);
        }
    }
    __syncthreads();
    const float* sc = a.sched + ((size_t)(a.group ? a.group[b] : 0) * a.n_steps + step) * 4;
    const float alpha = sc[0], dc_div = sc[1], nscale = sc[2];
    const float dcb = sc[3] != 0.f ? sc[3] : 1.f;            // dc_boost of test_mmse.py:231-233; 0 = not set (hosts of the 3-column era)
    const float2* ext = a.noise ? reinterpret_cast<const float2*>(a.noise) + ((size_t)step * B + b) * Nt * Nr : nullptr;
    const int64_t traj = a.traj_id ? a.traj_id[b] : b;
    float err = 0.f, den = 0.f;
    // two adjacent elements (same row t; Nr is even) per thread and round: one Philox block feeds both, one pilot load too
    for (int q = tid; q < Nt * Nr / 2; q += 256) {
        const int e = 2 * q, t = e / Nr, r = e - t * Nr;
        float2 g0 = make_float2(0.f, 0.f), g1 = make_float2(0.f, 0.f);
        for (int m = 0; m < Np; ++m) {
            const float2 pv = P[(size_t)m * PS + t];
            const float4 rr = *reinterpret_cast<const float4*>(Rs + m * Nr + r);      // (r even: 16-byte aligned)
            g0 = cfma_conj(pv, make_float2(rr.x, rr.y), g0);
            g1 = cfma_conj(pv, make_float2(rr.z, rr.w), g1);
        }
        const float4 s = reinterpret_cast<const float4*>(Sc)[q], x = reinterpret_cast<const float4*>(Xs)[q];
        const float4 h = reinterpret_cast<const float4*>(Ht)[q];
        float2 n0, n1;
        if (ext) {
            const float4 nn = reinterpret_cast<const float4*>(ext)[q];
            n0 = make_float2(nn.x, nn.y); n1 = make_float2(nn.z, nn.w);
        } else {
            complex_normal_pair(a.seed, traj, step, q, n0, n1);
        }
        float4 u;
        u.x = x.x + alpha * (s.x - (dcb * g0.x) / dc_div) + nscale * n0.x;     // x * 1.0f is exact: test_score semantics
        u.y = x.y + alpha * (s.y - (dcb * g0.y) / dc_div) + nscale * n0.y;
        u.z = x.z + alpha * (s.z - (dcb * g1.x) / dc_div) + nscale * n1.x;
        u.w = x.w + alpha * (s.w - (dcb * g1.y) / dc_div) + nscale * n1.y;
        reinterpret_cast<float4*>(X)[q] = u;
        float dx = u.x - h.x, dy = u.y - h.y;
        err += dx * dx + dy * dy;
        den += h.x * h.x + h.y * h.y;
        dx = u.z - h.z; dy = u.w - h.w;
        err += dx * dx + dy * dy;
        den += h.z * h.z + h.w * h.w;
    }
    for (int off = 32; off > 0; off >>= 1) {
        err += __shfl_down(err, off);
        den += __shfl_down(den, off);
    }
    if ((tid & 63) == 0) { red[0][tid >> 6] = err; red[1][tid >> 6] = den; }
    __syncthreads();
    if (tid == 0) {
        const float e4 = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        const float d4 = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
        a.nmse[(size_t)step * B + b] = e4 / d4;
    }
}

// Large arrays (256 x 64: X alone is 128 KB): the two products are register-blocked and K-chunked instead.  512 threads;
// thread (g, r) = (tid / Nr, tid % Nr) owns column r of up to J output rows g, g + G, g + 2G, ... (G = 512 / Nr groups) in
// registers; per K chunk the P slab [rows][KC] and the X (or R) slab [KC][Nr] go through LDS, so each complex FMA costs one
// broadcast LDS read of P (all lanes of a wave share the row) -- the X / R element is read once per chunk row and reused for
// all J rows.  R = P X - Y stays in LDS for the second product.  The summation order over Nt / Np is the same ascending
// order as langevin_kernel's, so both kernels agree to rounding of identical operation sequences.
template <int J, int KC>
__global__ __launch_bounds__(512) void langevin_tiled_kernel(sbc_langevin a, int B) {
    extern __shared__ __attribute__((aligned(16))) float2 sm[];
    constexpr int NT_ = 512;
    const int Nt = a.Nt, Nr = a.Nr, Np = a.Np;
    const int G = NT_ / Nr;                               // row groups
    float2* Rs = sm;                                      // [Np][Nr]
    float2* Ps = Rs + Np * Nr;                            // phase 1: [G*J][KC] (zero rows beyond Np); phase 2: [KC][G*J]
    float2* Xs = Ps + G * J * KC;                         // phase 1: [KC][Nr]
    __shared__ float red[2][8];
    const int b = blockIdx.x, tid = threadIdx.x, r = tid % Nr, g = tid / Nr;
    const int step = *a.step;
    float2* X = reinterpret_cast<float2*>(a.X) + (size_t)b * Nt * Nr;
    const float2* P = reinterpret_cast<const float2*>(a.P) + (size_t)(a.p_index ? a.p_index[b] : b) * Np * Nt;
    const float2* Y = reinterpret_cast<const float2*>(a.Y) + (size_t)b * Np * Nr;
    const float2* Ht = reinterpret_cast<const float2*>(a.Htrue) + (size_t)(a.h_index ? a.h_index[b] : b) * Nt * Nr;
    const float2* Sc = reinterpret_cast<const float2*>(a.score) + (size_t)b * Nt * Nr;
    // ---- R = P X - Y: output rows m = g + G*j (+ pass offset), K = Nt
    for (int m0 = 0; m0 < Np; m0 += G * J) {
        float2 acc[J];
#pragma unroll
        for (int j = 0; j < J; ++j) acc[j] = make_float2(0.f, 0.f);
        for (int t0 = 0; t0 < Nt; t0 += KC) {
            __syncthreads();
            for (int e = tid; e < G * J * KC; e += NT_) {
                const int mm = e / KC, tt = e - mm * KC;
                Ps[e] = (m0 + mm < Np && t0 + tt < Nt) ? P[(size_t)(m0 + mm) * Nt + t0 + tt] : make_float2(0.f, 0.f);
            }
            for (int e = tid; e < KC * Nr; e += NT_) {
                const int tt = e / Nr;
                Xs[e] = t0 + tt < Nt ? X[(size_t)(t0 + tt) * Nr + (e - tt * Nr)] : make_float2(0.f, 0.f);
            }
            __syncthreads();
#pragma unroll 4
            for (int tt = 0; tt < KC; ++tt) {
                const float2 x = Xs[tt * Nr + r];
#pragma unroll
                for (int j = 0; j < J; ++j) acc[j] = cfma(Ps[(g + G * j) * KC + tt], x, acc[j]);
            }
        }
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const int m = m0 + g + G * j;
            if (m < Np) {
                const float2 y = Y[m * Nr + r];
                Rs[m * Nr + r] = make_float2(acc[j].x - y.x, acc[j].y - y.y);
            }
        }
    }
    // ---- G = P^H R: output rows t = g + G*j, K = Np; then the update and the NMSE terms of those elements
    const float* sc = a.sched + ((size_t)(a.group ? a.group[b] : 0) * a.n_steps + step) * 4;
    const float alpha = sc[0], dc_div = sc[1], nscale = sc[2];
    const float dcb = sc[3] != 0.f ? sc[3] : 1.f;            // 0 = not set, as in langevin_kernel
    const float2* ext = a.noise ? reinterpret_cast<const float2*>(a.noise) + ((size_t)step * B + b) * Nt * Nr : nullptr;
    const int64_t traj = a.traj_id ? a.traj_id[b] : b;
    float err = 0.f, den = 0.f;
    for (int q0 = 0; q0 < Nt; q0 += G * J) {
        float2 acc[J];
#pragma unroll
        for (int j = 0; j < J; ++j) acc[j] = make_float2(0.f, 0.f);
        for (int k0 = 0; k0 < Np; k0 += KC) {
            __syncthreads();                              // also orders the Rs writes above before the first read
            for (int e = tid; e < KC * G * J; e += NT_) {
                const int mm = e / (G * J), tt = e - mm * (G * J);
                Ps[e] = (k0 + mm < Np && q0 + tt < Nt) ? P[(size_t)(k0 + mm) * Nt + q0 + tt] : make_float2(0.f, 0.f);
            }
            __syncthreads();
            const int kn = min(KC, Np - k0);
            for (int mm = 0; mm < kn; ++mm) {
                const float2 rv = Rs[(k0 + mm) * Nr + r];
#pragma unroll
                for (int j = 0; j < J; ++j) acc[j] = cfma_conj(Ps[mm * (G * J) + g + G * j], rv, acc[j]);
            }
        }
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const int t = q0 + g + G * j;
            if (t < Nt) {
                const int e = t * Nr + r;
                const float2 s = Sc[e], x = X[e], h = Ht[e];
                const float2 n = ext ? ext[e] : complex_normal(a.seed, traj, step, e);
                float2 u;
                u.x = x.x + alpha * (s.x - (dcb * acc[j].x) / dc_div) + nscale * n.x;
                u.y = x.y + alpha * (s.y - (dcb * acc[j].y) / dc_div) + nscale * n.y;
                X[e] = u;
                const float dx = u.x - h.x, dy = u.y - h.y;
                err += dx * dx + dy * dy;
                den += h.x * h.x + h.y * h.y;
            }
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        err += __shfl_down(err, off);
        den += __shfl_down(den, off);
    }
    if ((tid & 63) == 0) { red[0][tid >> 6] = err; red[1][tid >> 6] = den; }
    __syncthreads();
    if (tid == 0) {
        float e8 = 0.f, d8 = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) { e8 += red[0][w]; d8 += red[1][w]; }
        a.nmse[(size_t)step * B + b] = e8 / d8;
    }
}

static int check_langevin(const sbc_op& op, const sbc_langevin& a, bool measure) {
    SBC_REQUIRE(a.P && a.Y && a.Htrue, "langevin/measure: P, Y, Htrue must be set");
    SBC_REQUIRE(a.Nt > 0 && a.Nr > 0 && a.Np > 0 && op.B > 0, "langevin/measure: bad sizes");
    if (!measure) {
        SBC_REQUIRE(a.X && a.score && a.sched && a.nmse && a.step && a.n_steps > 0,
                    "langevin: X, score, sched, nmse, step must be set");
        const size_t lds = (size_t)a.Np * a.Nr * sizeof(float2);
        SBC_REQUIRE(lds <= 150 * 1024, "langevin: Nr=%d Np=%d needs %zu bytes of LDS", a.Nr, a.Np, lds);
        // the update loop takes the elements of a row in adjacent PAIRS (one Philox block, one pilot value, 16-byte accesses):
        // a pair must not straddle two rows, and every per-trajectory tensor must start on a 16-byte boundary
        SBC_REQUIRE(a.Nr % 2 == 0, "langevin: Nr = %d must be even (elements are updated in adjacent pairs of a row)", a.Nr);
        SBC_REQUIRE(!(((uintptr_t)a.X | (uintptr_t)a.score | (uintptr_t)a.Y | (uintptr_t)a.Htrue | (uintptr_t)a.noise) & 15),
                    "langevin: X, score, Y, Htrue and noise must be 16-byte aligned");
    } else {
        SBC_REQUIRE(a.meas_scale, "measure: meas_scale must be set");
    }
    return SBC_OK;
}

int launch_langevin(const sbc_op& op, const sbc_langevin& a, hipStream_t stream, bool dry) {
    const int rc = check_langevin(op, a, false);
    if (rc) return rc;
    const size_t lds_all = (size_t)(a.Nt + a.Np) * a.Nr * sizeof(float2);
    const int x_in_lds = lds_all <= 64 * 1024;
    const size_t lds_p = (size_t)a.Np * (a.Nt + 2) * sizeof(float2);                          // padded rows (langevin_kernel)
    const int p_in_lds = x_in_lds && a.Nt % 2 == 0 && lds_all + lds_p <= 40 * 1024;            // keeps 4 workgroups per CU
    const size_t lds = (x_in_lds ? lds_all : (size_t)a.Np * a.Nr * sizeof(float2)) + (p_in_lds ? lds_p : 0);
    if (!x_in_lds && 512 % a.Nr == 0) {
        // large arrays: register-blocked, K-chunked products (langevin_tiled_kernel)
        constexpr int J = 32, KC = 16;
        const int G = 512 / a.Nr;
        const size_t lds_t = ((size_t)a.Np * a.Nr + (size_t)G * J * KC + (size_t)KC * a.Nr) * sizeof(float2);
        if (lds_t <= 156 * 1024) {
            auto kern = langevin_tiled_kernel<J, KC>;
            { const int rc = ensure_dyn_lds(reinterpret_cast<const void*>(kern), lds_t); if (rc) return rc; }
            if (dry) return SBC_OK;
            hipLaunchKernelGGL(kern, dim3(op.B), dim3(512), lds_t, stream, a, op.B);
            SBC_CHECK_HIP(hipGetLastError());
            return SBC_OK;
        }
    }
    { const int rc = ensure_dyn_lds(reinterpret_cast<const void*>(langevin_kernel), lds); if (rc) return rc; }
    if (dry) return SBC_OK;
    hipLaunchKernelGGL(langevin_kernel, dim3(op.B), dim3(256), lds, stream, a, op.B, x_in_lds, p_in_lds);
    SBC_CHECK_HIP(hipGetLastError());
    return SBC_OK;
}

// Y = P H + sqrt(local_noise) * n   (test_score.py:122-124); n from `noise` [B][Np][Nr] or Philox (step = -1).
__global__ __launch_bounds__(256) void measure_kernel(sbc_langevin a, int B) {
    const int Nt = a.Nt, Nr = a.Nr, Np = a.Np;
    const int b = blockIdx.x, tid = threadIdx.x;
    const float2* P = reinterpret_cast<const float2*>(a.P) + (size_t)(a.p_index ? a.p_index[b] : b) * Np * Nt;
    const float2* Ht = reinterpret_cast<const float2*>(a.Htrue) + (size_t)(a.h_index ? a.h_index[b] : b) * Nt * Nr;
    float2* Y = reinterpret_cast<float2*>(a.Y) + (size_t)b * Np * Nr;
    const float2* ext = a.noise ? reinterpret_cast<const float2*>(a.noise) + (size_t)b * Np * Nr : nullptr;
    const float sn = a.meas_scale[b];
    const int64_t traj = a.traj_id ? a.traj_id[b] : b;
    for (int o = tid; o < Np * Nr; o += 256) {
        const int m = o / Nr, r = o - m * Nr;
        float2 acc = make_float2(0.f, 0.f);
        for (int t = 0; t < Nt; ++t) acc = cfma(P[(size_t)m * Nt + t], Ht[t * Nr + r], acc);
        const float2 n = ext ? ext[o] : complex_normal(a.seed, traj, -1, o);
        Y[o] = make_float2(acc.x + sn * n.x, acc.y + sn * n.y);
    }
}

int launch_measure(const sbc_op& op, const sbc_langevin& a, hipStream_t stream) {
    const int rc = check_langevin(op, a, true);
    if (rc) return rc;
    hipLaunchKernelGGL(measure_kernel, dim3(op.B), dim3(256), 0, stream, a, op.B);
    SBC_CHECK_HIP(hipGetLastError());
    return SBC_OK;
}

// ------------------------------------------------------------------------------------------------ step counter
__global__ void step_inc_kernel(int* step) { *step += 1; }

int launch_step_inc(const sbc_op& op, hipStream_t stream) {
    SBC_REQUIRE(op.out, "step_inc: out (device int32 counter) must be set");
    hipLaunchKernelGGL(step_inc_kernel, dim3(1), dim3(1), 0, stream, (int*)op.out);
    SBC_CHECK_HIP(hipGetLastError());
    return SBC_OK;
}

}  // namespace sbc

// ---- known-answer hooks of the in-kernel random numbers (include/sbc_hip.h: sbc_debug_philox4x32, sbc_debug_complex_normal) ----
namespace sbc {
__global__ void debug_philox_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, int n) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    const uint4 r = philox4x32(make_uint4(in[i * 6], in[i * 6 + 1], in[i * 6 + 2], in[i * 6 + 3]), make_uint2(in[i * 6 + 4], in[i * 6 + 5]));
    out[i * 4] = r.x; out[i * 4 + 1] = r.y; out[i * 4 + 2] = r.z; out[i * 4 + 3] = r.w;
}
__global__ void debug_normal_kernel(uint64_t seed, int64_t traj, int step, int n, float2* __restrict__ out) {
    const int e = blockIdx.x * 64 + threadIdx.x;
    if (e < n) out[e] = complex_normal(seed, traj, step, e);
}
}  // namespace sbc

extern "C" int sbc_debug_philox4x32(const uint32_t* counters_keys, int32_t n, uint32_t* out) {
    using namespace sbc;
    SBC_REQUIRE(counters_keys && out && n > 0, "sbc_debug_philox4x32: bad arguments");
    uint32_t *din = nullptr, *dout = nullptr;
    SBC_CHECK_HIP(hipMalloc((void**)&din, (size_t)n * 6 * 4));
    SBC_CHECK_HIP(hipMalloc((void**)&dout, (size_t)n * 4 * 4));
    SBC_CHECK_HIP(hipMemcpy(din, counters_keys, (size_t)n * 6 * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(debug_philox_kernel, dim3((n + 63) / 64), dim3(64), 0, nullptr, din, dout, n);
    SBC_CHECK_HIP(hipMemcpy(out, dout, (size_t)n * 4 * 4, hipMemcpyDeviceToHost));
    (void)hipFree(din); (void)hipFree(dout);
    return SBC_OK;
}

extern "C" int sbc_debug_complex_normal(uint64_t seed, int64_t traj, int32_t step, int32_t n_elem, float* out) {
    using namespace sbc;
    SBC_REQUIRE(out && n_elem > 0, "sbc_debug_complex_normal: bad arguments");
    float2* d = nullptr;
    SBC_CHECK_HIP(hipMalloc((void**)&d, (size_t)n_elem * 8));
    hipLaunchKernelGGL(debug_normal_kernel, dim3((n_elem + 63) / 64), dim3(64), 0, nullptr, seed, traj, step, n_elem, d);
    SBC_CHECK_HIP(hipMemcpy(out, d, (size_t)n_elem * 8, hipMemcpyDeviceToHost));
    (void)hipFree(d);
    return SBC_OK;
}

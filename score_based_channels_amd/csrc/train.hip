// Training operators, part 1: the denoising-score-matching loss (ncsnv2/losses/dsm.py:6-32) and the
// reverse-mode counterparts of the non-convolution operators of the score network, plus the optimiser step
// (torch.optim.Adam as configured by losses/__init__.py:3-7, EMAHelper.update of models/ema.py:17-22).  The
// convolution gradients are in train_conv.hip.  All tensors NHWC float32; every sum has a fixed order (no atomics), so a
// training step is reproducible bit for bit.
#include "philox.h"
#include "tile.h"

namespace sbc {

// elu_grad1 (common.h): d ELU(x) / dx    [nn.ELU(alpha = 1), layers.py:12-13]
__device__ __forceinline__ float4 elu_grad4(float4 v) {
    return make_float4(elu_grad1(v.x), elu_grad1(v.y), elu_grad1(v.z), elu_grad1(v.w));
}
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

// ------------------------------------------------------------------------------------------------ DSM perturbation
// noise = randn_like(samples) * used_sigmas; perturbed = samples + noise          (dsm.py:14-17)
__global__ __launch_bounds__(256) void dsm_perturb_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                           float* __restrict__ nz, sbc_dsm e, int B, int n) {
#pragma clang fp contract(off)                                      // noise is a rounded product, then added (dsm.py:15-17)
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;          // one pair of elements
    const int half = n / 2;
    if (idx >= (long)B * half) return;
    const int b = (int)(idx / half), k = (int)(idx - (long)b * half);
    const float sigma = e.sigmas[e.labels[b]];
    float2 z;
    if (e.noise) z = *reinterpret_cast<const float2*>(e.noise + (size_t)b * n + 2 * k);
    else z = normal_pair(e.seed, e.sample_id ? e.sample_id[b] : b, e.offset + (e.step ? *e.step : 0), k);
    const float2 xv = *reinterpret_cast<const float2*>(x + (size_t)b * n + 2 * k);
    const float2 nv = make_float2(z.x * sigma, z.y * sigma);
    *reinterpret_cast<float2*>(nz + (size_t)b * n + 2 * k) = nv;
    *reinterpret_cast<float2*>(out + (size_t)b * n + 2 * k) = make_float2(xv.x + nv.x, xv.y + nv.y);
}

int launch_dsm_perturb(const sbc_op& op, const sbc_dsm& e, hipStream_t stream) {
    SBC_REQUIRE(op.in && op.out && op.aux && e.sigmas && e.labels, "dsm_perturb: in/out/aux/sigmas/labels must be set");
    const int n = op.H * op.W * op.cin;
    SBC_REQUIRE(n > 0 && n % 2 == 0 && op.B > 0, "dsm_perturb: bad shape");
    const long pairs = (long)op.B * (n / 2);
    hipLaunchKernelGGL(dsm_perturb_kernel, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, stream, (const float*)op.in,
                       (float*)op.out, (float*)op.aux, e, op.B, n);
    SBC_CHECK_HIP(hipGetLastError());
    return SBC_OK;
}

// ------------------------------------------------------------------------------------------------ DSM loss
// target = -1 / sigma^2 * noise; loss_b = 1/2 * sum((scores - target)^2) * sigma^p          (dsm.py:19-30)
// d mean_b(loss_b) / d scores = (scores - target) * sigma^p / B.  One workgroup per sample, fixed summation order.
__global__ __launch_bounds__(256) void dsm_loss_kernel(const float* __restrict__ s, const float* __restrict__ nz,
                                                        float* __restrict__ loss, float* __restrict__ ds, sbc_dsm e,
                                                        int B, int n) {
    __shared__ float red[256];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float sigma = e.sigmas[e.labels[b]];
    const float inv = -(1.f / (sigma * sigma));
    const float wp = e.anneal_power == 2.f ? sigma * sigma : powf(sigma, e.anneal_power);
    const float gs = wp / (float)B * (e.grad_scale != 0.f ? e.grad_scale : 1.f);
    float acc = 0.f;
    for (int i = tid; i < n; i += 256) {
        const float d = s[(size_t)b * n + i] - inv * nz[(size_t)b * n + i];
        acc = fmaf(d, d, acc);
        if (ds) ds[(size_t)b * n + i] = d * gs;
    }
    red[tid] = acc;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (tid < st) red[tid] += red[tid + st];
        __syncthreads();
    }
    if (tid == 0) loss[b] = 0.5f * red[0] * wp;
}

int launch_dsm_loss(const sbc_op& op, const sbc_dsm& e, hipStream_t stream) {
    SBC_REQUIRE(op.in && op.out && op.grad && e.sigmas && e.labels, "dsm_loss: in/out/grad/sigmas/labels must be set");
    const int n = op.H * op.W * op.cin;
    SBC_REQUIRE(n > 0 && op.B > 0, "dsm_loss: bad shape");
    hipLaunchKernelGGL(dsm_loss_kernel, dim3(op.B), dim3(256), 0, stream, (const float*)op.in, (const float*)op.grad,
                       (float*)op.out, (float*)op.aux, e, op.B, n);
    SBC_CHECK_HIP(hipGetLastError());
    return SBC_OK;
}

// ------------------------------------------------------------------------------------------------ grad add / ELU backward
__global__ __launch_bounds__(256) void grad_add_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                        float* __restrict__ out, long n4, int flags) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    float4 v = ld4(g + i * 4);
    if (flags & SBC_PRO_ELU) {
        const float4 d = elu_grad4(ld4(x + i * 4));
        v.x *= d.x; v.y *= d.y; v.z *= d.z; v.w *= d.w;
    }
    if (flags & SBC_BWD_ACCUM) {
        const float4 o = ld4(out + i * 4);
        v.x = o.x + v.x; v.y = o.y + v.y; v.z = o.z + v.z; v.w = o.w + v.w;
    }
    st4(out + i * 4, v);
}

int launch_grad_add(const sbc_op& op, hipStream_t stream) {
    SBC_REQUIRE(op.grad && op.out && (!(op.flags & SBC_PRO_ELU) || op.in), "grad_add: grad/out (and in with PRO_ELU) must be set");
    const long n = (long)op.B * op.H * op.W * op.cin;
    SBC_REQUIRE(n > 0 && n % 4 == 0, "grad_add: element count must be a positive multiple of 4");
    hipLaunchKernelGGL(grad_add_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, stream, (const float*)op.in,
                       (const float*)op.grad, (float*)op.out, n / 4, op.flags);
    SBC_CHECK_HIP(hipGetLastError());
    return SBC_OK;
}

// ------------------------------------------------------------------------------------------------ InstanceNorm++ backward
// Forward (normalization.py:163-176), per sample, N = H*W, C channels:
//   mu_c = mean_N x,  rstd_c = 1/sqrt(var_N x + eps),  h = (x - mu_c) rstd_c
//   m = mean_C mu,  t = sqrt(var_C(mu, unbiased) + eps),  mhat_c = (mu_c - m) / t
//   n = gamma_c (h + mhat_c alpha_c) + beta_c  [= (x - mu) scale + shift of the forward statistics],  y = ELU(n)
// With g = dL/dn (= grad * ELU'(n)), S1 = sum_N g, S2 = sum_N g (x - mu):
//   d beta = S1, d gamma = rstd S2 + mhat alpha S1, d alpha = gamma mhat S1           (summed over the batch)
//   q_c = gamma_c alpha_c S1_c = dL/d mhat_c
//   d mu_c (through mhat) = (q_c - mean_C q)/t - (mu_c - m) sum_C(q (mu - m)) / ((C-1) t^3)
//   dL/dx = gamma rstd g - gamma rstd S1/N - gamma rstd^3 S2 (x - mu)/N + d mu_c / N  =  A g + Bc (x - mu) + Cc
// Kernel 1 (one workgroup per sample) reduces S1, S2, sum (x-mu)^2 and writes A, Bc, Cc, S1, rstd*S2, mhat per channel;
// kernel 2 applies the element-wise formula; kernel 3 sums the parameter gradients over the batch.
template <int C>
__global__ __launch_bounds__(256) void inorm_bwd_reduce_kernel(const float* __restrict__ x, const float* __restrict__ stats,
                                                                const float* __restrict__ agb, const float* __restrict__ grad,
                                                                float* __restrict__ aux, int HW, int flags) {
    constexpr int C4 = C / 4;
    constexpr int J = 256 / C4;
    __shared__ float4 r1[256], r2[256], r3[256];
    __shared__ float S1s[C], S2s[C], Qs[C], mus[C], qs[C];
    const int n = blockIdx.x, tid = threadIdx.x;
    const int c4 = tid % C4, j = tid / C4;
    const float* st = stats + (size_t)n * 3 * C + c4 * 4;
    const float4 mu = ld4(st), sc = ld4(st + C), sh = ld4(st + 2 * C);
    const float* xb = x + (size_t)n * HW * C + c4 * 4;
    const float* gb = grad + (size_t)n * HW * C + c4 * 4;
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1, q = s1;
    for (int px = j; px < HW; px += J) {
        const float4 xv = ld4(xb + (size_t)px * C);
        float4 g = ld4(gb + (size_t)px * C);
        const float4 d = make_float4(xv.x - mu.x, xv.y - mu.y, xv.z - mu.z, xv.w - mu.w);
        if (flags & SBC_PRO_ELU) {
            const float4 e = elu_grad4(make_float4(fmaf(d.x, sc.x, sh.x), fmaf(d.y, sc.y, sh.y), fmaf(d.z, sc.z, sh.z),
                                                   fmaf(d.w, sc.w, sh.w)));
            g.x *= e.x; g.y *= e.y; g.z *= e.z; g.w *= e.w;
        }
        s1.x += g.x; s1.y += g.y; s1.z += g.z; s1.w += g.w;
        s2.x = fmaf(g.x, d.x, s2.x); s2.y = fmaf(g.y, d.y, s2.y); s2.z = fmaf(g.z, d.z, s2.z); s2.w = fmaf(g.w, d.w, s2.w);
        q.x = fmaf(d.x, d.x, q.x); q.y = fmaf(d.y, d.y, q.y); q.z = fmaf(d.z, d.z, q.z); q.w = fmaf(d.w, d.w, q.w);
    }
    r1[tid] = s1; r2[tid] = s2; r3[tid] = q;
    __syncthreads();
    for (int stp = J / 2; stp > 0; stp >>= 1) {
        if (j < stp) {
            const int o = tid + stp * C4;
            float4 a = r1[tid], b = r1[o];
            r1[tid] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
            a = r2[tid]; b = r2[o];
            r2[tid] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
            a = r3[tid]; b = r3[o];
            r3[tid] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
        }
        __syncthreads();
    }
    if (j == 0) {
        const float4 a = r1[tid], b = r2[tid], c = r3[tid];
        S1s[c4 * 4 + 0] = a.x; S1s[c4 * 4 + 1] = a.y; S1s[c4 * 4 + 2] = a.z; S1s[c4 * 4 + 3] = a.w;
        S2s[c4 * 4 + 0] = b.x; S2s[c4 * 4 + 1] = b.y; S2s[c4 * 4 + 2] = b.z; S2s[c4 * 4 + 3] = b.w;
        Qs[c4 * 4 + 0] = c.x; Qs[c4 * 4 + 1] = c.y; Qs[c4 * 4 + 2] = c.z; Qs[c4 * 4 + 3] = c.w;
        mus[c4 * 4 + 0] = mu.x; mus[c4 * 4 + 1] = mu.y; mus[c4 * 4 + 2] = mu.z; mus[c4 * 4 + 3] = mu.w;
    }
    __syncthreads();
    if (tid < C) qs[tid] = agb[C + tid] * agb[tid] * S1s[tid];            // gamma alpha S1
    __syncthreads();
    if (tid < C) {
        float m = 0.f;
        for (int c = 0; c < C; ++c) m += mus[c];
        m *= 1.f / (float)C;
        float v = 0.f, qbar = 0.f, qdot = 0.f;
        for (int c = 0; c < C; ++c) {
            const float d = mus[c] - m;
            v = fmaf(d, d, v);
            qbar += qs[c];
            qdot = fmaf(qs[c], d, qdot);
        }
        v *= 1.f / (float)(C - 1);
        qbar *= 1.f / (float)C;
        const float t = sqrtf(v + 1e-5f), inv_n = 1.f / (float)HW;
        const float dc = mus[tid] - m;
        const float mhat = dc / t;
        const float dmu = (qs[tid] - qbar) / t - dc * qdot / ((float)(C - 1) * t * t * t);
        const float rstd = 1.f / sqrtf(Qs[tid] * inv_n + 1e-5f);
        const float gamma = agb[C + tid];
        float* o = aux + (size_t)n * 6 * C;
        o[tid] = gamma * rstd;                                              // A
        o[C + tid] = -gamma * rstd * rstd * rstd * S2s[tid] * inv_n;        // Bc
        o[2 * C + tid] = (dmu - gamma * rstd * S1s[tid]) * inv_n;           // Cc
        o[3 * C + tid] = S1s[tid];
        o[4 * C + tid] = rstd * S2s[tid];
        o[5 * C + tid] = mhat;
    }
}

__global__ __launch_bounds__(256) void inorm_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ stats,
                                                               const float* __restrict__ grad, const float* __restrict__ aux,
                                                               float* __restrict__ out, long n4, int HW, int C, int flags,
                                                               const float* __restrict__ agb, float* __restrict__ dagb, int B) {
    if (blockIdx.x == gridDim.x - 1) {
        // The last workgroup also sums the per-sample terms of d alpha | d gamma | d beta over the batch, in ascending order
        // (a launch of its own until round 3: 25 launches per optimiser step on a host-launch-bound path).  It reads only `aux`,
        // which the reduce launch finished before this one started.
        for (int c = threadIdx.x; c < C; c += 256) {
            const float alpha = agb[c], gamma = agb[C + c];
            float da = 0.f, dg = 0.f, db = 0.f;
            for (int b = 0; b < B; ++b) {
                const float* o = aux + (size_t)b * 6 * C;
                const float s1 = o[3 * C + c], s2h = o[4 * C + c], mhat = o[5 * C + c];
                db += s1;
                dg += fmaf(mhat * alpha, s1, s2h);
                da = fmaf(gamma * mhat, s1, da);
            }
            dagb[c] = da; dagb[C + c] = dg; dagb[2 * C + c] = db;
        }
    }
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const int C4 = C / 4;
    const int c4 = (int)(i % C4), n = (int)(i / ((long)HW * C4));
    const float* st = stats + (size_t)n * 3 * C + c4 * 4;
    const float* ax = aux + (size_t)n * 6 * C + c4 * 4;
    const float4 mu = ld4(st), sc = ld4(st + C), sh = ld4(st + 2 * C);
    const float4 A = ld4(ax), Bc = ld4(ax + C), Cc = ld4(ax + 2 * C);
    const float4 xv = ld4(x + i * 4);
    float4 g = ld4(grad + i * 4);
    const float4 d = make_float4(xv.x - mu.x, xv.y - mu.y, xv.z - mu.z, xv.w - mu.w);
    if (flags & SBC_PRO_ELU) {
        const float4 e = elu_grad4(make_float4(fmaf(d.x, sc.x, sh.x), fmaf(d.y, sc.y, sh.y), fmaf(d.z, sc.z, sh.z),
                                               fmaf(d.w, sc.w, sh.w)));
        g.x *= e.x; g.y *= e.y; g.z *= e.z; g.w *= e.w;
    }
    float4 v = make_float4(fmaf(A.x, g.x, fmaf(Bc.x, d.x, Cc.x)), fmaf(A.y, g.y, fmaf(Bc.y, d.y, Cc.y)),
                           fmaf(A.z, g.z, fmaf(Bc.z, d.z, Cc.z)), fmaf(A.w, g.w, fmaf(Bc.w, d.w, Cc.w)));
    if (flags & SBC_BWD_ACCUM) {
        const float4 o = ld4(out + i * 4);
        v.x = o.x + v.x; v.y = o.y + v.y; v.z = o.z + v.z; v.w = o.w + v.w;
    }
    st4(out + i * 4, v);
}

int launch_inorm_bwd(const sbc_op& op, hipStream_t stream) {
    SBC_REQUIRE(op.in && op.stats && op.weight && op.grad && op.out && op.aux && op.wgrad,
                "inorm_bwd: in/stats/weight/grad/out/aux/wgrad must be set");
    const int HW = op.H * op.W, C = op.cin;
    const float* x = (const float*)op.in;
    const float* st = (const float*)op.stats;
    const float* agb = (const float*)op.weight;
    const float* g = (const float*)op.grad;
    float* aux = (float*)op.aux;
    switch (C) {
        case 32: hipLaunchKernelGGL(inorm_bwd_reduce_kernel<32>, dim3(op.B), dim3(256), 0, stream, x, st, agb, g, aux, HW, op.flags); break;
        case 64: hipLaunchKernelGGL(inorm_bwd_reduce_kernel<64>, dim3(op.B), dim3(256), 0, stream, x, st, agb, g, aux, HW, op.flags); break;
        case 128: hipLaunchKernelGGL(inorm_bwd_reduce_kernel<128>, dim3(op.B), dim3(256), 0, stream, x, st, agb, g, aux, HW, op.flags); break;
        default: set_error("inorm_bwd: %d channels (only 32/64/128)", C); return SBC_ERR_UNSUPPORTED;
    }
    SBC_CHECK_HIP(hipGetLastError());
    const long n4 = (long)op.B * HW * C / 4;
    hipLaunchKernelGGL(inorm_bwd_apply_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, stream, x, st, g,
                       (const float*)aux, (float*)op.out, n4, HW, C, op.flags, agb, (float*)op.wgrad, op.B);
    SBC_CHECK_HIP(hipGetLastError());
    return SBC_OK;
}

// ------------------------------------------------------------------------------------------------ max pool backward
// nn.MaxPool2d(5, 1, 2): every output's gradient goes to the first maximum of its window in row-major order (the
// `val > maxval` scan of PyTorch's max_pool2d).  Pass 1 stores that position (0..24) per output element, pass 2 gathers:
// input element p collects grad[q] of the <= 25 windows q that contain it and chose it.  With SBC_PRO_ELU the forward
// was maxpool(ELU(x)): ELU is monotone, so the argmax is taken on x, and the result is multiplied by ELU'(x).
__global__ __launch_bounds__(256) void maxpool5_argmax_kernel(const float* __restrict__ in, unsigned char* __restrict__ idx,
                                                               int B, int H, int W, int C) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)B * H * W * C) return;
    const int c = (int)(i % C), w = (int)((i / C) % W), h = (int)((i / ((long)C * W)) % H), n = (int)(i / ((long)C * W * H));
    const float* base = in + (size_t)n * H * W * C + c;
    float best = -INFINITY;
    int bi = -1;
    for (int kh = 0; kh < 5; ++kh) {
        const int hh = h + kh - 2;
        if (hh < 0 || hh >= H) continue;
        for (int kw = 0; kw < 5; ++kw) {
            const int ww = w + kw - 2;
            if (ww < 0 || ww >= W) continue;
            const float v = base[((size_t)hh * W + ww) * C];
            if (v > best || bi < 0) { best = v; bi = kh * 5 + kw; }
        }
    }
    idx[i] = (unsigned char)bi;
}

__global__ __launch_bounds__(256) void maxpool5_bwd_kernel(const float* __restrict__ in, const float* __restrict__ grad,
                                                            const unsigned char* __restrict__ idx, float* __restrict__ out,
                                                            int B, int H, int W, int C, int flags) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)B * H * W * C) return;
    const int c = (int)(i % C), w = (int)((i / C) % W), h = (int)((i / ((long)C * W)) % H), n = (int)(i / ((long)C * W * H));
    const size_t sb = (size_t)n * H * W * C + c;
    float acc = 0.f;
    for (int dh = -2; dh <= 2; ++dh) {
        const int hh = h + dh;
        if (hh < 0 || hh >= H) continue;
        for (int dw = -2; dw <= 2; ++dw) {
            const int ww = w + dw;
            if (ww < 0 || ww >= W) continue;
            const size_t q = sb + ((size_t)hh * W + ww) * C;
            if (idx[q] == (2 - dh) * 5 + (2 - dw)) acc += grad[q];      // window q holds p at offset (-dh, -dw)
        }
    }
    if (flags & SBC_PRO_ELU) acc *= elu_grad1(in[i]);
    if (flags & SBC_BWD_ACCUM) acc = out[i] + acc;
    out[i] = acc;
}

int launch_maxpool5_bwd(const sbc_op& op, hipStream_t stream) {
    SBC_REQUIRE(op.in && op.grad && op.out && op.aux, "maxpool5_bwd: in/grad/out/aux must be set");
    const long n = (long)op.B * op.H * op.W * op.cin;
    const unsigned g = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(maxpool5_argmax_kernel, dim3(g), dim3(256), 0, stream, (const float*)op.in, (unsigned char*)op.aux,
                       op.B, op.H, op.W, op.cin);
    SBC_CHECK_HIP(hipGetLastError());
    hipLaunchKernelGGL(maxpool5_bwd_kernel, dim3(g), dim3(256), 0, stream, (const float*)op.in, (const float*)op.grad,
                       (const unsigned char*)op.aux, (float*)op.out, op.B, op.H, op.W, op.cin, op.flags);
    SBC_CHECK_HIP(hipGetLastError());
    return SBC_OK;
}

// ------------------------------------------------------------------------------------------------ resize backward
// Transpose of the SBC_EPI_UP resize (conv_epilogue.h): high-res pixel (r, s) reads low-res rows h0, h1 with weights
// (1 - l), l where f = r * (up_h - 1)/(H - 1), h0 = min(int(f), up_h - 1), h1 = min(h0 + 1, up_h - 1), l = f - h0; the
// same along the width.  A low-res element gathers from the few high-res pixels that read it, in ascending order.
__device__ __forceinline__ float up_weight(int r, float scale, int lo_n, int i) {
    const float f = scale * (float)r;
    const int h0 = min((int)f, lo_n - 1), h1 = min(h0 + 1, lo_n - 1);
    const float l1 = f - (float)h0;
    return (h0 == i ? 1.f - l1 : 0.f) + (h1 == i ? l1 : 0.f);
}

__global__ __launch_bounds__(256) void upsample_bwd_kernel(const float* __restrict__ grad, float* __restrict__ out, int B,
                                                            int H, int W, int uh, int uw, int C4, int flags) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)B * uh * uw * C4) return;
    const int c4 = (int)(i % C4), j = (int)((i / C4) % uw), ii = (int)((i / ((long)C4 * uw)) % uh);
    const int n = (int)(i / ((long)C4 * uw * uh));
    const float sh = H > 1 ? (float)(uh - 1) / (float)(H - 1) : 0.f;
    const float sw = W > 1 ? (float)(uw - 1) / (float)(W - 1) : 0.f;
    // candidate high-res rows: those whose source coordinate lies within (ii - 1, ii + 1), with a safety margin
    int r_lo = 0, r_hi = H - 1, s_lo = 0, s_hi = W - 1;
    if (sh > 0.f) { r_lo = max(0, (int)floorf((float)(ii - 1) / sh) - 1); r_hi = min(H - 1, (int)ceilf((float)(ii + 1) / sh) + 1); }
    if (sw > 0.f) { s_lo = max(0, (int)floorf((float)(j - 1) / sw) - 1); s_hi = min(W - 1, (int)ceilf((float)(j + 1) / sw) + 1); }
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const float* g = grad + (size_t)n * H * W * C4 * 4 + c4 * 4;
    for (int r = r_lo; r <= r_hi; ++r) {
        const float wr = up_weight(r, sh, uh, ii);
        if (wr == 0.f) continue;
        for (int s = s_lo; s <= s_hi; ++s) {
            const float ws = up_weight(s, sw, uw, j);
            if (ws == 0.f) continue;
            const float4 v = ld4(g + ((size_t)r * W + s) * C4 * 4);
            const float wt = wr * ws;
            acc.x = fmaf(wt, v.x, acc.x); acc.y = fmaf(wt, v.y, acc.y); acc.z = fmaf(wt, v.z, acc.z); acc.w = fmaf(wt, v.w, acc.w);
        }
    }
    if (flags & SBC_BWD_ACCUM) {
        const float4 o = ld4(out + i * 4);
        acc.x = o.x + acc.x; acc.y = o.y + acc.y; acc.z = o.z + acc.z; acc.w = o.w + acc.w;
    }
    st4(out + i * 4, acc);
}

int launch_upsample_bwd(const sbc_op& op, hipStream_t stream) {
    SBC_REQUIRE(op.grad && op.out && op.up_h > 0 && op.up_w > 0 && op.cin % 4 == 0, "upsample_bwd: grad/out/up_h/up_w must be set");
    const long n = (long)op.B * op.up_h * op.up_w * (op.cin / 4);
    hipLaunchKernelGGL(upsample_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, (const float*)op.grad,
                       (float*)op.out, op.B, op.H, op.W, op.up_h, op.up_w, op.cin / 4, op.flags);
    SBC_CHECK_HIP(hipGetLastError());
    return SBC_OK;
}

// ------------------------------------------------------------------------------------------------ mean-pool backward
// ConvMeanPool (layers.py:311-312): y = (a + b + c + d) / 4  =>  every one of the four inputs receives grad / 4.
__global__ __launch_bounds__(256) void pool_bwd_kernel(const float* __restrict__ grad, float* __restrict__ out, int B, int H,
                                                        int W, int C4) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)B * H * W * C4) return;
    const int c4 = (int)(i % C4), w = (int)((i / C4) % W), h = (int)((i / ((long)C4 * W)) % H), n = (int)(i / ((long)C4 * W * H));
    const float4 v = ld4(grad + ((((size_t)n * (H / 2) + h / 2) * (W / 2) + w / 2) * C4 + c4) * 4);
    st4(out + i * 4, make_float4(0.25f * v.x, 0.25f * v.y, 0.25f * v.z, 0.25f * v.w));
}

int launch_pool_bwd(const sbc_op& op, hipStream_t stream) {
    SBC_REQUIRE(op.grad && op.out && op.cin % 4 == 0 && op.H % 2 == 0 && op.W % 2 == 0, "pool_bwd: grad/out must be set, even H and W");
    const long n = (long)op.B * op.H * op.W * (op.cin / 4);
    hipLaunchKernelGGL(pool_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, (const float*)op.grad,
                       (float*)op.out, op.B, op.H, op.W, op.cin / 4);
    SBC_CHECK_HIP(hipGetLastError());
    return SBC_OK;
}

// ------------------------------------------------------------------------------------------------ weight packing on device
// sbc_pack_conv_weight_split on the device, for weights that change every optimiser step: one thread per packed 8-element
// group.  Layout [k*k][cin'/16][cout'/32][3][64 lanes][8] (uint16 bf16 patterns); lane l of block (tap, g, n) holds
// w'[n*32 + (l & 31)][g*16 + 8*(l >> 5) + j][tap].  SBC_PACK_ADJOINT: w'[ci][co][tap] = w[co][ci][taps - 1 - tap].
__device__ __forceinline__ unsigned short bf16_bits(float v) {
    const __bf16 b = (__bf16)v;
    return __builtin_bit_cast(unsigned short, b);
}

// `table` != NULL: one launch packs many weights, blockIdx.y = entry, 6 int32 per entry:
// (source offset in floats from w, destination offset in uint16 from dst, cout, cin, taps, adjoint)
__global__ __launch_bounds__(256) void pack_weight_kernel(const float* __restrict__ w, unsigned short* __restrict__ dst,
                                                           int cout, int cin, int taps, int adjoint,
                                                           const int* __restrict__ table) {
    if (table) {
        const int* e = table + 6 * blockIdx.y;
        w += e[0]; dst += e[1]; cout = e[2]; cin = e[3]; taps = e[4]; adjoint = e[5];
    }
    const int co_p = adjoint ? cin : cout, ci_p = adjoint ? cout : cin;      // packed (cout', cin')
    const int KG = ci_p / 16, NB = co_p / 32;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)taps * KG * NB * 64) return;
    const int lane = (int)(i % 64), n = (int)((i / 64) % NB), g = (int)((i / (64L * NB)) % KG), tap = (int)(i / (64L * NB * KG));
    const int co = n * 32 + (lane & 31);
    unsigned short h[8], m[8], l[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int ci = g * 16 + 8 * (lane >> 5) + j;
        float v;
        if (taps == 16) {
            // Winograd F(2x2,3x3) form of a 3x3 weight: U = G g G^T in double, rounded once (sbc_pack_conv_weight_winograd_split);
            // `tap` is the transform position (i, l); the adjoint convolution's filter is the flipped, transposed one
            const float* gsrc = adjoint ? w + ((size_t)ci * cin + co) * 9 : w + ((size_t)co * cin + ci) * 9;
            const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
            const int i = tap >> 2, q = tap & 3;
            double acc = 0;
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b) acc += G[i][a] * (double)gsrc[adjoint ? 8 - (a * 3 + b) : a * 3 + b] * G[q][b];
            v = (float)acc;
        } else {
            v = adjoint ? w[((size_t)ci * cin + co) * taps + (taps - 1 - tap)] : w[((size_t)co * cin + ci) * taps + tap];
        }
        const __bf16 bh = (__bf16)v;
        const float r1 = v - (float)bh;
        const __bf16 bm = (__bf16)r1;
        const float r2 = r1 - (float)bm;
        h[j] = __builtin_bit_cast(unsigned short, bh);
        m[j] = __builtin_bit_cast(unsigned short, bm);
        l[j] = bf16_bits(r2);
    }
    unsigned short* o = dst + ((((size_t)tap * KG + g) * NB + n) * 3) * 64 * 8 + (size_t)lane * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) { o[j] = h[j]; o[64 * 8 + j] = m[j]; o[2 * 64 * 8 + j] = l[j]; }
}

int launch_pack_weight(const sbc_op& op, hipStream_t stream) {
    SBC_REQUIRE(op.in && op.out, "pack_weight: in/out must be set");
    if (op.aux) {
        // batched form: B entries of the device table in `aux`; cin/cout/ksize = the LARGEST weight (sizes the grid)
        SBC_REQUIRE(op.B > 0 && op.cin > 0 && op.cout > 0 && op.ksize > 0, "pack_weight (batched): B, cin, cout, ksize must be set");
        const long n = 16L * (op.cin / 16) * (op.cout / 16) * 64;     // upper bound for every form (16 Winograd positions)
        hipLaunchKernelGGL(pack_weight_kernel, dim3((unsigned)((n + 255) / 256), op.B), dim3(256), 0, stream,
                           (const float*)op.in, (unsigned short*)op.out, 0, 0, 0, 0, (const int*)op.aux);
        SBC_CHECK_HIP(hipGetLastError());
        return SBC_OK;
    }
    SBC_REQUIRE(op.cin % 32 == 0 && op.cout % 32 == 0 && (op.ksize == 1 || op.ksize == 3), "pack_weight: cin, cout %% 32, ksize in {1, 3}");
    SBC_REQUIRE(!(op.flags & SBC_PACK_WINOGRAD) || op.ksize == 3, "pack_weight: the Winograd form exists for 3x3 weights");
    const int taps = (op.flags & SBC_PACK_WINOGRAD) ? 16 : op.ksize * op.ksize, adj = (op.flags & SBC_PACK_ADJOINT) ? 1 : 0;
    const long n = (long)taps * (op.cin / (adj ? 32 : 16)) * (op.cout / (adj ? 16 : 32)) * 64;
    hipLaunchKernelGGL(pack_weight_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, (const float*)op.in,
                       (unsigned short*)op.out, op.cout, op.cin, taps, adj, (const int*)nullptr);
    SBC_CHECK_HIP(hipGetLastError());
    return SBC_OK;
}

// ------------------------------------------------------------------------------------------------ Adam + EMA
// torch.optim.Adam (single-tensor form, weight_decay 0, amsgrad False):
//   exp_avg.lerp_(g, 1 - b1); exp_avg_sq = b2 exp_avg_sq + (1 - b2) g^2
//   p -= (lr / (1 - b1^t)) * exp_avg / (sqrt(exp_avg_sq) / sqrt(1 - b2^t) + eps)
// then EMAHelper.update: shadow = (1 - mu) p + mu shadow (models/ema.py:17-22).  The bias corrections are evaluated in
// double from the device step counter (torch computes them as python floats).
__global__ __launch_bounds__(256) void adam_ema_kernel(const float* __restrict__ g, float* __restrict__ p,
                                                        float* __restrict__ state, sbc_adam a) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.n) return;
    const double t = (double)(*a.step + 1);
    const double bc1 = 1.0 - pow(a.beta1, t), bc2 = 1.0 - pow(a.beta2, t);
    const float step_size = (float)(a.lr / bc1), bc2s = (float)sqrt(bc2);
    const float w1 = (float)(1.0 - a.beta1), b2 = (float)a.beta2, w2 = (float)(1.0 - a.beta2), eps = (float)a.eps;
    const float gv = g[i];
    float m = state[i], v = state[a.n + i];
    m = m + (gv - m) * w1;                                   // exp_avg.lerp_(grad, 1 - beta1)
    v = v * b2 + w2 * (gv * gv);                             // mul_(beta2).addcmul_(grad, grad, value = 1 - beta2)
    state[i] = m;
    state[a.n + i] = v;
    const float denom = sqrtf(v) / bc2s + eps;
    const float pn = p[i] - step_size * (m / denom);
    p[i] = pn;
    if (a.ema_mu >= 0.0) {
        const float mu = (float)a.ema_mu, omu = (float)(1.0 - a.ema_mu);
        state[2 * a.n + i] = omu * pn + mu * state[2 * a.n + i];
    }
}

int launch_adam_ema(const sbc_op& op, const sbc_adam& a, hipStream_t stream) {
    SBC_REQUIRE(op.in && op.out && op.aux && a.step && a.n > 0, "adam_ema: in/out/aux/step/n must be set");
    hipLaunchKernelGGL(adam_ema_kernel, dim3((unsigned)((a.n + 255) / 256)), dim3(256), 0, stream, (const float*)op.in,
                       (float*)op.out, (float*)op.aux, a);
    SBC_CHECK_HIP(hipGetLastError());
    return SBC_OK;
}

}  // namespace sbc

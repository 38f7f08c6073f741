// Epilogue shared by the fp32-MFMA (conv_mfma.hip) and split-bf16 (conv_x3.hip) direct convolution kernels: the
// 32x32 accumulator blocks go through LDS as [pixel][COUT + 4] so that every global access is a 16-byte access of
// four consecutive channels, then bias / 2x2 mean pool / residuals / bilinear resize-add are applied
// (ResidualBlock layers.py:456, RCUBlock :133, CRPBlock :82, ConvMeanPool :311-312, MSFBlock :182-183).
#pragma once
#include "conv_common.h"

namespace sbc {

// accumulators (32x32 MFMA map: column = lane & 31 = output channel, row = (r&3) + 8*(r>>2) + 4*(lane>>5)) -> LDS
template <int COUT, int MT, int NT>
__device__ __forceinline__ void conv_acc_to_lds(float* lds, const f32x16 (&acc)[MT][NT], const float* __restrict__ bias,
                                                int wm, int wn, int lane, float descale = 1.f) {
    constexpr int ES = COUT + 4;
    const int col = lane & 31, rhalf = 4 * (lane >> 5);
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) {
            const int co = (wn * NT + ni) * 32 + col;
            float* e = lds + ((wm * MT + mi) * 32 + rhalf) * ES + co;
            // (descale is an exact power of two: acc * descale + bias rounds once, like acc + bias)
            const float bv = bias ? bias[co] : 0.f;
            if (bias || descale != 1.f) {
#pragma unroll
                for (int r = 0; r < 16; ++r) e[((r & 3) + 8 * (r >> 2)) * ES] = fmaf(acc[mi][ni][r], descale, bv);
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) e[((r & 3) + 8 * (r >> 2)) * ES] = acc[mi][ni][r];
            }
        }
}

// LDS [TM][COUT + 4] (conv + bias) -> global, with the fused pool / residual / resize-add variants
template <int COUT, int TM, int NTHREADS, bool P2>
__device__ __forceinline__ void conv_epilogue(const float* lds, const ConvParams& p, const TileGeom& g,
                                              const Dims<P2>& dm, int tid) {
    constexpr int ES = COUT + 4;
    constexpr int C4 = COUT / 4;
    constexpr int ITER = TM * C4 / NTHREADS;            // 16-byte output chunks per thread
    static_assert(TM * C4 % NTHREADS == 0, "epilogue chunks must divide evenly");
    constexpr int EC = ITER < 4 ? ITER : 4;             // 16-byte requests in flight per thread and phase
    static_assert(ITER % EC == 0, "epilogue chunking");
    const int H = p.H, W = p.W, HW = H * W;

    if (p.flags & SBC_EPI_POOL) {
        // ((((0 + a) + b) + c) + d) / 4, a=[0::2,0::2] b=[1::2,0::2] c=[0::2,1::2] d=[1::2,1::2] (layers.py:311-312)
        constexpr int PTOT = (TM / 4) * C4;              // pooled 16-byte outputs of the tile
        constexpr int PC = 4;
        const int Wo = W / 2, Ho = H / 2;
        const int r0 = dm.div_w(g.p0);                 // first global row of the tile (even)
#pragma unroll 1
        for (int base = 0; base < PTOT; base += PC * NTHREADS) {
            float4 v[PC], rr[PC];
            unsigned o[PC];
            bool ok[PC];
#pragma unroll
            for (int i = 0; i < PC; ++i) {
                const int idx = base + i * NTHREADS + tid;
                const int c4 = idx % C4, q = idx / C4;
                const int qr = P2 ? q >> (p.wsh - 1) : q / Wo, qc = q - qr * Wo;
                const int grow = r0 + 2 * qr;
                ok[i] = idx < PTOT && grow < p.B * H;
                const float* e = lds + (ok[i] ? ((2 * qr) * W + 2 * qc) * ES + c4 * 4 : 0);
                const float4 a = *reinterpret_cast<const float4*>(e), b = *reinterpret_cast<const float4*>(e + W * ES);
                const float4 c = *reinterpret_cast<const float4*>(e + ES), d = *reinterpret_cast<const float4*>(e + (W + 1) * ES);
                v[i].x = (((a.x + b.x) + c.x) + d.x) * 0.25f;
                v[i].y = (((a.y + b.y) + c.y) + d.y) * 0.25f;
                v[i].z = (((a.z + b.z) + c.z) + d.z) * 0.25f;
                v[i].w = (((a.w + b.w) + c.w) + d.w) * 0.25f;
                const int n = dm.div_h(grow), ho = (grow - n * H) >> 1;
                o[i] = ((unsigned)(n * Ho + ho) * Wo + qc) * COUT + c4 * 4;
            }
            if (p.res1) {
#pragma unroll
                for (int i = 0; i < PC; ++i)
                    if (ok[i]) rr[i] = ld_stream(p.res1 + o[i]);
#pragma unroll
                for (int i = 0; i < PC; ++i) {
                    v[i].x = rr[i].x + v[i].x; v[i].y = rr[i].y + v[i].y;
                    v[i].z = rr[i].z + v[i].z; v[i].w = rr[i].w + v[i].w;
                }
            }
#pragma unroll
            for (int i = 0; i < PC; ++i)
                if (ok[i]) st_stream(p.out + o[i], v[i]);
        }
        return;
    }

    const float sh = (p.flags & SBC_EPI_UP) && H > 1 ? (float)(p.up_h - 1) / (float)(H - 1) : 0.f;
    const float sw = (p.flags & SBC_EPI_UP) && W > 1 ? (float)(p.up_w - 1) / (float)(W - 1) : 0.f;
#pragma unroll 1
    for (int c0 = 0; c0 < ITER; c0 += EC) {
        float4 v[EC], rr[EC];
        unsigned o[EC];
        bool ok[EC];
#pragma unroll
        for (int i = 0; i < EC; ++i) {
            const int idx = tid + (c0 + i) * NTHREADS;
            const int c4 = idx % C4, pl = idx / C4;
            v[i] = *reinterpret_cast<const float4*>(lds + pl * ES + c4 * 4);
            ok[i] = g.p0 + pl < p.total_px;
            o[i] = (unsigned)(g.p0 + pl) * COUT + c4 * 4;
        }
        if (p.flags & SBC_EPI_ELUGRAD) {
            // reverse pass: this convolution is the adjoint of a forward conv whose input went through ELU; `res2` holds that
            // forward input: v = conv * ELU'(res2) [+ res1, the gradient collected so far -- may alias `out`]
#pragma unroll
            for (int i = 0; i < EC; ++i)
                if (ok[i]) rr[i] = ld_stream(p.res2 + o[i]);
#pragma unroll
            for (int i = 0; i < EC; ++i) {
                v[i].x *= elu_grad1(rr[i].x); v[i].y *= elu_grad1(rr[i].y);
                v[i].z *= elu_grad1(rr[i].z); v[i].w *= elu_grad1(rr[i].w);
            }
        }
        if (p.res1) {
#pragma unroll
            for (int i = 0; i < EC; ++i)
                if (ok[i]) rr[i] = ld_stream(p.res1 + o[i]);
            if (p.flags & SBC_EPI_RES1_ELU) {
#pragma unroll
                for (int i = 0; i < EC; ++i) rr[i] = elu4_acc(rr[i]);
            }
            if (p.res2 && !(p.flags & SBC_EPI_ELUGRAD)) {
                float4 r2[EC];
#pragma unroll
                for (int i = 0; i < EC; ++i)
                    if (ok[i]) r2[i] = ld_stream(p.res2 + o[i]);
#pragma unroll
                for (int i = 0; i < EC; ++i) {
                    rr[i].x = r2[i].x + rr[i].x; rr[i].y = r2[i].y + rr[i].y;
                    rr[i].z = r2[i].z + rr[i].z; rr[i].w = r2[i].w + rr[i].w;
                }
            }
#pragma unroll
            for (int i = 0; i < EC; ++i) {
                v[i].x = v[i].x + rr[i].x; v[i].y = v[i].y + rr[i].y;
                v[i].z = v[i].z + rr[i].z; v[i].w = v[i].w + rr[i].w;
            }
        }
        if (p.flags & SBC_EPI_UP) {
            // F.interpolate(bilinear, align_corners=True) of `up` added on top (MSFBlock, layers.py:182-183)
#pragma unroll
            for (int i = 0; i < EC; ++i) {
                if (!ok[i]) continue;
                const int idx = tid + (c0 + i) * NTHREADS;
                const int c4 = idx % C4, px = g.p0 + idx / C4;
                const int n = dm.div_hw(px), rem = px - n * HW;
                const int h = dm.div_w(rem), w = rem - h * W;
                const float fh = sh * (float)h, fw = sw * (float)w;
                const int h0 = min((int)fh, p.up_h - 1), w0 = min((int)fw, p.up_w - 1);
                const int h1 = min(h0 + 1, p.up_h - 1), w1 = min(w0 + 1, p.up_w - 1);
                const float lh1 = fh - (float)h0, lw1 = fw - (float)w0;
                const float lh0 = 1.f - lh1, lw0 = 1.f - lw1;
                const float* u = p.up + (size_t)n * p.up_h * p.up_w * COUT + c4 * 4;
                const float4 v00 = *reinterpret_cast<const float4*>(u + (size_t)(h0 * p.up_w + w0) * COUT);
                const float4 v01 = *reinterpret_cast<const float4*>(u + (size_t)(h0 * p.up_w + w1) * COUT);
                const float4 v10 = *reinterpret_cast<const float4*>(u + (size_t)(h1 * p.up_w + w0) * COUT);
                const float4 v11 = *reinterpret_cast<const float4*>(u + (size_t)(h1 * p.up_w + w1) * COUT);
                v[i].x = v[i].x + (lh0 * (lw0 * v00.x + lw1 * v01.x) + lh1 * (lw0 * v10.x + lw1 * v11.x));
                v[i].y = v[i].y + (lh0 * (lw0 * v00.y + lw1 * v01.y) + lh1 * (lw0 * v10.y + lw1 * v11.y));
                v[i].z = v[i].z + (lh0 * (lw0 * v00.z + lw1 * v01.z) + lh1 * (lw0 * v10.z + lw1 * v11.z));
                v[i].w = v[i].w + (lh0 * (lw0 * v00.w + lw1 * v01.w) + lh1 * (lw0 * v10.w + lw1 * v11.w));
            }
        }
#pragma unroll
        for (int i = 0; i < EC; ++i)
            if (ok[i]) st_stream(p.out + o[i], v[i]);
        // SBC_EPI_MOMENTS_OUT (tile.h): 256 threads, 32 output channels -- a pass of four chunks per thread is one whole 128-pixel tile in
        // the layout tile_moments_out32 takes (thread (tid >> 3, tid & 7): four pixels of channel quad tid & 7); scratch behind the tile
        if constexpr (COUT == 32 && NTHREADS == 256 && EC == 4 && TM % 128 == 0) {
            if (p.flags & SBC_EPI_MOMENTS_OUT) {
                float* red = const_cast<float*>(lds) + TM * ES;
                tile_moments_out32(v, red, p.pm_out + ((size_t)((g.p0 >> 7) + c0 / 4) * COUT) * 2, tid);
                __syncthreads();                              // (the scratch is reused by the next pass)
            }
        }
    }
}

}  // namespace sbc

// EXPERIMENT (round 2), NOT part of the product build: Winograd F(4x4,3x3) with split-bf16 products for the 32 -> 32 layers.
// Result on MI355X: correct (all epilogue variants matched the oracle at the conv tolerance when it was wired in as
// `weight_wino4_split`), but 216-222 us per launch at T = 1700, 64x16 against 196 us for conv_wx3 (F(2x2,3x3)): the per-wave
// row transform (4 LDS reads + 4 FMAs per value, 48 ds_read_b128 per K step), the 55 KB of T planes (two workgroups = 12 waves
// per CU instead of 16) and twice as many, half-sized MFMAs eat the 1.8x saving in splits.  A one-column filter prefetch ring
// and precomputed patch offsets changed nothing.  Kept for the record; the host-side packer it needs is
// `U = G g G^T` with the 6x3 G below, split like sbc_pack_conv_weight_winograd_split, layout [36][cin/32][cout/16][3][64][8]
// (lane l: U[nb*16 + (l & 15)][kg*32 + 8*(l >> 4) + j]).
//   G = [[1/4,0,0],[-1/6,-1/6,-1/6],[-1/6,1/6,-1/6],[1/24,1/12,1/6],[1/24,-1/12,1/6],[0,0,1]]
//
// 3x3 stride-1 convolution (padding 1, no dilation) by Winograd F(4x4, 3x3) with the 36 element-wise products on the bf16
// matrix cores of gfx950, fp32 in / fp32 out / fp32 accumulate, fp32-accurate products (exact three-term bf16 split).
//
// Why a second Winograd kernel: conv_wx3.hip (F(2x2,3x3)) is bound by instruction issue, and 60 % of its K loop is the exact
// split of the transformed input -- 9 instructions per pair of values, 4 transformed values per output pixel and input
// channel.  F(4x4,3x3) needs 36 products per 16 outputs = 2.25 transformed values per output pixel: 1.8x fewer splits and
// MFMAs for a costlier input transform.  Numerically it is viable *because* the products are exact:
// the forward error against the reference stays at 1.3e-6 (F(2x2): 0.8e-6).
//
// Decomposition (Lavin & Gray 2016):  Y = A^T [ (G g G^T) .* (B^T d B) ] A  with 6x6 input patches d (stride 4), 4x4 outputs.
//   * a workgroup owns 16 Winograd tiles = 256 output pixels (whole image rows of one sample, or whole samples), staged
//     with a one-row halo as fp32 [pixel][CIN + 4] exactly like conv_wx3 (tile.h); six waves, wave xi owns row xi of B^T;
//   * the 36 GEMMs M[xi][nu] = V[xi][nu] U[xi][nu] run on v_mfma_f32_16x16x32_bf16: M = 16 tiles, K = 32 input channels,
//     N = 16 output channels; lane l holds tile l & 15, channels 8 (l >> 4) .. + 8 of the A operand;
//   * per 32 input channels a wave forms R_j = sum_i B^T[xi][i] d[i][j] (j = 0..5, at most four non-zero i), then the six
//     columns V_nu with shared sub-expressions, splits each V_nu into three bf16 terms and issues 6 MFMAs per 16 output
//     channels; the filter fragments U = G g G^T (split on the host, sbc_pack_conv_weight_winograd4_split) come from L2;
//   * A^T over nu in registers, the per-xi partial results through LDS [xi][b][tile][36] (overlaying the staged tile), and
//     each finish task (tile, output row a, channel quad) applies A^T over xi and bias / residual(s) / bilinear resize-add.
#include <stdlib.h>
#include "conv_common.h"

namespace sbc {

typedef float f32x4v __attribute__((ext_vector_type(4)));

template <int CIN, int COUT, bool P2>
__global__ __launch_bounds__(384, 3) void conv_wx4_kernel(ConvParams p) {
    constexpr int TM = 256;                      // output pixels per workgroup = 16 tiles of 4 x 4
    constexpr int S = CIN + 4;
    constexpr int KG = CIN / 32;                 // K steps of 32 input channels
    constexpr int NB = COUT / 16;                // 16-channel output blocks
    constexpr int TS = 36;                       // floats per (tile) row of a T plane: 32 channels + 4 pad
    constexpr int NTHREADS = 384;
    constexpr int NPF = ((TM + 32) * (CIN / 4) + NTHREADS - 1) / NTHREADS;
    static_assert(COUT == 32, "T planes / finish are laid out for 32 output channels");
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int xi = __builtin_amdgcn_readfirstlane(tid >> 6);          // transform row of this wave, 0..5
    const int H = p.H, W = p.W, HW = H * W;
    const Dims<P2> dm{H, W, HW, p.hsh, p.wsh};
    const int kq = 8 * (lane >> 4);                                   // first of this lane's 8 input channels (per K step)

    const TileGeom g = tile_geom(xcd_tile(blockIdx.x, gridDim.x), TM, p.B, dm, 1);
    {
        float4 pf[NPF];
        stage_issue<CIN, NTHREADS, NPF>(pf, p.in, g, W, tid);
        float* st_lds = lds + p.stats_off;
        if (p.flags & SBC_PRO_NORM) {
            stage_stats_to_lds<CIN, NTHREADS, P2>(st_lds, p.stats, g, dm, tid);
            __syncthreads();
        }
        stage_commit<CIN, NTHREADS, NPF, P2>(lds, pf, p.in, st_lds, p.flags, g, dm, tid, 0);
    }

    // B^T row xi as (row index, coefficient) pairs, at most four non-zeros (a zero-weight dummy for xi = 0, 5)
    //   xi0: 4 d0 - 5 d2 + d4        xi1: -4 d1 - 4 d2 + d3 + d4     xi2: 4 d1 - 4 d2 - d3 + d4
    //   xi3: -2 d1 - d2 + 2 d3 + d4  xi4: 2 d1 - d2 - 2 d3 + d4      xi5: 4 d1 - 5 d3 + d5
    const int ri0 = xi == 0 ? 0 : 1, ri1 = xi == 0 ? 2 : xi == 5 ? 3 : 2, ri2 = xi == 0 ? 4 : xi == 5 ? 5 : 3,
              ri3 = xi == 0 ? 4 : xi == 5 ? 5 : 4;
    const float c0 = xi == 0 ? 4.f : xi == 1 ? -4.f : xi == 2 ? 4.f : xi == 3 ? -2.f : xi == 4 ? 2.f : 4.f;
    const float c1 = xi == 0 ? -5.f : xi == 1 ? -4.f : xi == 2 ? -4.f : xi == 3 ? -1.f : xi == 4 ? -1.f : -5.f;
    const float c2 = xi == 0 ? 1.f : xi == 1 ? 1.f : xi == 2 ? -1.f : xi == 3 ? 2.f : xi == 4 ? -2.f : 1.f;
    const float c3 = (xi == 0 || xi == 5) ? 0.f : 1.f;

    // this lane's tile and the LDS offsets of the four patch rows it reads (zero pixel when outside the image)
    const int Wt = W >> 2;                                            // tiles per image row
    const int r0 = dm.div_w(g.p0);                                    // first output row of the workgroup
    const int t = lane & 15;
    const int tr = P2 ? t >> (p.wsh - 2) : t / Wt, tc = t - tr * Wt;
    const int grow = r0 + 4 * tr;                                     // first output row of the tile
    const int hrow = dm.mod_h(grow);
    const bool tile_ok = grow < p.B * H;
    const int zoff = g.nps * S + kq;
    int rowoff[4];
    {
        const int ris[4] = {ri0, ri1, ri2, ri3};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int hh = hrow - 1 + ris[k];
            rowoff[k] = (tile_ok && hh >= 0 && hh < H) ? ((grow - 1 + ris[k] - g.rs0) * W) * S + kq : -1;
        }
    }
    const int col0 = 4 * tc - 1;
    int off[4][6];                                                    // LDS float offsets of the 4 x 6 patch pixels this wave reads
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int ww = col0 + j;
            off[k][j] = (ww >= 0 && ww < W && rowoff[k] >= 0) ? rowoff[k] + ww * S : zoff;
        }
    __syncthreads();                                                  // staged tile visible

    f32x4v T[NB][4];                                                  // sum over nu of A^T[b][nu] M[xi][nu], b = 0..3
    {
        f32x4v acc[6][NB];
#pragma unroll
        for (int nu = 0; nu < 6; ++nu)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[nu][nb] = f32x4v{0.f, 0.f, 0.f, 0.f};
        // split U: [(xi*6 + nu)][kg][nb][term][lane] 16-byte fragments
        const uint4* wp = reinterpret_cast<const uint4*>(p.wpk) + (size_t)(xi * 6) * KG * NB * 3 * 64 + lane;
#pragma unroll
        for (int kg = 0; kg < KG; ++kg) {
            float R[6][8];
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                float d[4][8];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float* q = lds + off[k][j] + kg * 32;
                    const float4 a0 = *reinterpret_cast<const float4*>(__builtin_assume_aligned(q, 16));
                    const float4 a1 = *reinterpret_cast<const float4*>(__builtin_assume_aligned(q + 4, 16));
                    d[k][0] = a0.x; d[k][1] = a0.y; d[k][2] = a0.z; d[k][3] = a0.w;
                    d[k][4] = a1.x; d[k][5] = a1.y; d[k][6] = a1.z; d[k][7] = a1.w;
                }
#pragma unroll
                for (int c = 0; c < 8; ++c) R[j][c] = fmaf(c3, d[3][c], fmaf(c2, d[2][c], fmaf(c1, d[1][c], c0 * d[0][c])));
            }
            // columns of B: V0 = 4 R0 - 5 R2 + R4;  V1, V2 = (R4 - 4 R2) +- (R3 - 4 R1);  V3, V4 = (R4 - R2) +- 2 (R3 - R1);
            // V5 = 4 R1 - 5 R3 + R5
            float V[6][8];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float t1 = fmaf(-4.f, R[2][c], R[4][c]), t2 = fmaf(-4.f, R[1][c], R[3][c]);
                const float t3 = R[4][c] - R[2][c], t4 = 2.f * (R[3][c] - R[1][c]);
                V[0][c] = fmaf(4.f, R[0][c], fmaf(-5.f, R[2][c], R[4][c]));
                V[1][c] = t1 + t2; V[2][c] = t1 - t2;
                V[3][c] = t3 + t4; V[4][c] = t3 - t4;
                V[5][c] = fmaf(4.f, R[1][c], fmaf(-5.f, R[3][c], R[5][c]));
            }
            uint4 uR[2][NB][3];                                       // filter fragments, requested one column ahead
            auto u_load = [&](int nu_, int set) {
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                    for (int tm = 0; tm < 3; ++tm) uR[set][nb][tm] = wp[((size_t)((nu_ * KG + kg) * NB + nb) * 3 + tm) * 64];
            };
            u_load(0, 0);
#pragma unroll
            for (int nu = 0; nu < 6; ++nu) {
                if (nu + 1 < 6) u_load(nu + 1, (nu + 1) & 1);
                auto& u = uR[nu & 1];
                bf16x8 vh, vm, vl;
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const float v = V[nu][c];
                    const __bf16 h = (__bf16)v;
                    const float r1 = v - (float)h;
                    const __bf16 m = (__bf16)r1;
                    vh[c] = h; vm[c] = m; vl[c] = (__bf16)(r1 - (float)m);
                }
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    const bf16x8 uh = __builtin_bit_cast(bf16x8, u[nb][0]), um = __builtin_bit_cast(bf16x8, u[nb][1]),
                                 ul = __builtin_bit_cast(bf16x8, u[nb][2]);
                    // partial products, smallest first: (l,h) (h,l) (m,m) (m,h) (h,m) (h,h)
                    acc[nu][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vl, uh, acc[nu][nb], 0, 0, 0);
                    acc[nu][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vh, ul, acc[nu][nb], 0, 0, 0);
                    acc[nu][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vm, um, acc[nu][nb], 0, 0, 0);
                    acc[nu][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vm, uh, acc[nu][nb], 0, 0, 0);
                    acc[nu][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vh, um, acc[nu][nb], 0, 0, 0);
                    acc[nu][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vh, uh, acc[nu][nb], 0, 0, 0);
                }
            }
        }
        // A^T over nu:  [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float m0 = acc[0][nb][r], m1 = acc[1][nb][r], m2 = acc[2][nb][r], m3 = acc[3][nb][r],
                            m4 = acc[4][nb][r], m5 = acc[5][nb][r];
                const float s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
                T[nb][0][r] = (m0 + s12) + s34;
                T[nb][1][r] = fmaf(2.f, d34, d12);
                T[nb][2][r] = fmaf(4.f, s34, s12);
                T[nb][3][r] = fmaf(8.f, d34, d12) + m5;
            }
    }

    __syncthreads();                              // every wave is done with the staged tile: the T planes overlay it
    // T planes [xi][b][tile][TS]; C/D layout of the 16x16 MFMA: column (output channel) = lane & 15, row (tile) = 4 (lane >> 4) + r
    float* const tl = lds;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            float* e = tl + ((size_t)((xi * 4 + b) * 16 + 4 * (lane >> 4))) * TS + nb * 16 + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) e[r * TS] = T[nb][b][r];
        }
    __syncthreads();

    // finish: task = (tile, output row a of the tile, channel quad): Y[a][b] = sum_xi A^T[a][xi] T[xi][b]
    const float sh = (p.flags & SBC_EPI_UP) && H > 1 ? (float)(p.up_h - 1) / (float)(H - 1) : 0.f;
    const float sw = (p.flags & SBC_EPI_UP) && W > 1 ? (float)(p.up_w - 1) / (float)(W - 1) : 0.f;
#pragma unroll 1
    for (int task = tid; task < 16 * 4 * 8; task += NTHREADS) {
        const int c4 = task & 7, a = (task >> 3) & 3, tt = task >> 5;
        const int ttr = P2 ? tt >> (p.wsh - 2) : tt / Wt, ttc = tt - ttr * Wt;
        const int orow = r0 + 4 * ttr + a;                            // global output row (n * H + h)
        if (r0 + 4 * ttr >= p.B * H) continue;
        // A^T[a][xi]: a0: 1 1 1 1 1 0   a1: 0 1 -1 2 -2 0   a2: 0 1 1 4 4 0   a3: 0 1 -1 8 -8 1
        const float w1 = a == 0 ? 1.f : a == 1 ? 2.f : a == 2 ? 4.f : 8.f;      // weight of the (xi3, xi4) pair
        const float sg = (a & 1) ? -1.f : 1.f;                                    // xi2 and xi4 enter with this sign
        float4 y[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            float4 tx[6];
#pragma unroll
            for (int x = 0; x < 6; ++x)
                tx[x] = *reinterpret_cast<const float4*>(tl + ((size_t)((x * 4 + b) * 16 + tt)) * TS + c4 * 4);
            float4 v;
            v.x = (tx[1].x + sg * tx[2].x) + w1 * (tx[3].x + sg * tx[4].x);
            v.y = (tx[1].y + sg * tx[2].y) + w1 * (tx[3].y + sg * tx[4].y);
            v.z = (tx[1].z + sg * tx[2].z) + w1 * (tx[3].z + sg * tx[4].z);
            v.w = (tx[1].w + sg * tx[2].w) + w1 * (tx[3].w + sg * tx[4].w);
            if (a == 0) { v.x += tx[0].x; v.y += tx[0].y; v.z += tx[0].z; v.w += tx[0].w; }
            if (a == 3) { v.x += tx[5].x; v.y += tx[5].y; v.z += tx[5].z; v.w += tx[5].w; }
            y[b] = v;
        }
        const int co = c4 * 4;
        if (p.bias) {
            const float4 bv = *reinterpret_cast<const float4*>(p.bias + co);
#pragma unroll
            for (int b = 0; b < 4; ++b) { y[b].x += bv.x; y[b].y += bv.y; y[b].z += bv.z; y[b].w += bv.w; }
        }
        size_t o[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) o[b] = ((size_t)orow * W + 4 * ttc + b) * COUT + co;
        if (p.res1) {
            float4 rr[4];
#pragma unroll
            for (int b = 0; b < 4; ++b) rr[b] = ld_stream(p.res1 + o[b]);
            if (p.flags & SBC_EPI_RES1_ELU) {
#pragma unroll
                for (int b = 0; b < 4; ++b) rr[b] = elu4(rr[b]);
            }
            if (p.res2) {
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const float4 r2 = ld_stream(p.res2 + o[b]);
                    rr[b].x = r2.x + rr[b].x; rr[b].y = r2.y + rr[b].y; rr[b].z = r2.z + rr[b].z; rr[b].w = r2.w + rr[b].w;
                }
            }
#pragma unroll
            for (int b = 0; b < 4; ++b) { y[b].x += rr[b].x; y[b].y += rr[b].y; y[b].z += rr[b].z; y[b].w += rr[b].w; }
        }
        if (p.flags & SBC_EPI_UP) {
            // F.interpolate(bilinear, align_corners=True) of `up` added on top (MSFBlock, layers.py:182-183)
            const int n = dm.div_h(orow), hh = orow - n * H;
            const float* u = p.up + (size_t)n * p.up_h * p.up_w * COUT + co;
            const float fh = sh * (float)hh;
            const int h0 = min((int)fh, p.up_h - 1), h1 = min(h0 + 1, p.up_h - 1);
            const float lh1 = fh - (float)h0, lh0 = 1.f - lh1;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const float fw = sw * (float)(4 * ttc + b);
                const int w0 = min((int)fw, p.up_w - 1), w1i = min(w0 + 1, p.up_w - 1);
                const float lw1 = fw - (float)w0, lw0 = 1.f - lw1;
                const float4 v00 = *reinterpret_cast<const float4*>(u + (size_t)(h0 * p.up_w + w0) * COUT);
                const float4 v01 = *reinterpret_cast<const float4*>(u + (size_t)(h0 * p.up_w + w1i) * COUT);
                const float4 v10 = *reinterpret_cast<const float4*>(u + (size_t)(h1 * p.up_w + w0) * COUT);
                const float4 v11 = *reinterpret_cast<const float4*>(u + (size_t)(h1 * p.up_w + w1i) * COUT);
                y[b].x += lh0 * (lw0 * v00.x + lw1 * v01.x) + lh1 * (lw0 * v10.x + lw1 * v11.x);
                y[b].y += lh0 * (lw0 * v00.y + lw1 * v01.y) + lh1 * (lw0 * v10.y + lw1 * v11.y);
                y[b].z += lh0 * (lw0 * v00.z + lw1 * v01.z) + lh1 * (lw0 * v10.z + lw1 * v11.z);
                y[b].w += lh0 * (lw0 * v00.w + lw1 * v01.w) + lh1 * (lw0 * v10.w + lw1 * v11.w);
            }
        }
#pragma unroll
        for (int b = 0; b < 4; ++b) st_stream(p.out + o[b], y[b]);
    }
}

// ------------------------------------------------------------------------------------------------ dispatch
template <int CIN, int COUT>
static int launch_wx4(const ConvParams& p, hipStream_t stream, bool dry) {
    constexpr int TM = 256, S = CIN + 4;
    const int HW = p.H * p.W;
    const bool multi = TM >= HW;
    if (TM % (4 * p.W) != 0 || !(HW % TM == 0 || TM % HW == 0)) return 1;       // whole 4-row tile bands per workgroup
    const size_t staged = (size_t)(multi ? TM + 1 : TM + 2 * p.W + 1) * S * sizeof(float);
    const size_t tplanes = (size_t)6 * 4 * 16 * 36 * sizeof(float);
    const size_t lds = staged > tplanes ? staged : tplanes;
    const size_t nsamp = multi ? TM / HW : 1;
    const size_t stats_off = lds / sizeof(float);
    const size_t lds_all = lds + ((p.flags & SBC_PRO_NORM) ? nsamp * 3 * CIN * sizeof(float) : 0);
    if (lds_all > 160 * 1024) return 1;
    auto kern = conv_wx4_kernel<CIN, COUT, true>;
    { const int rc = ensure_dyn_lds(reinterpret_cast<const void*>(kern), lds_all); if (rc) return rc; }
    if (dry) return SBC_OK;
    ConvParams q = p;
    q.stats_off = (int)stats_off;
    const int ntiles = (p.total_px + TM - 1) / TM;
    hipLaunchKernelGGL(kern, dim3(ntiles), dim3(384), lds_all, stream, q);
    SBC_CHECK_HIP(hipGetLastError());
    return SBC_OK;
}

// SBC_OK after launching, 1 when the shape is not eligible (the caller falls back to conv_wx3 / conv_x3).
// `p.wpk` must point at sbc_pack_conv_weight_winograd4_split weights.
int launch_conv_wx4(const ConvParams& p, int cin, int cout, hipStream_t stream, bool dry) {
    // power-of-two images with sides divisible by 4; the pooled epilogue stays with F(2x2) (its finish owns the 2x2 window)
    if (p.dil != 1 || p.hsh < 2 || p.wsh < 2 || (p.flags & (SBC_EPI_POOL | SBC_CONV_F16W))) return 1;
    if (cin == 32 && cout == 32) return launch_wx4<32, 32>(p, stream, dry);
    return 1;
}

}  // namespace sbc

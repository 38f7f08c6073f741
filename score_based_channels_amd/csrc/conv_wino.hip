// 3x3 stride-1 convolution (padding 1, no dilation) by Winograd F(2x2, 3x3) on the fp32 matrix cores of gfx950.
//
// Same operator as conv_mfma.hip (nn.Conv2d 3x3 of ncsnv2/models/layers.py:37-44 with the fused prologue / epilogue
// of ResidualBlock, RCUBlock, CRPBlock, MSFBlock, ConvMeanPool) computed with 2.25x fewer multiplications:
//     Y = A^T [ (G g G^T) .* (B^T d B) ] A        per 2x2 output block, 4x4 input patch d, 3x3 filter g
// (Lavin & Gray 2016).  The filter transform U = G g G^T is done once on the host (weights.pack_conv_weight_winograd,
// packed as 16 "taps").  The 16 element-wise products are 16 GEMMs  M[xi][nu][tile][cout] = sum_cin V[xi][nu][tile]
// [cin] U[xi][nu][cin][cout]  over the Winograd tiles of the workgroup; they run on v_mfma_f32_32x32x2_f32 with
// M = 32 Winograd tiles, N = 32 output channels.  fp32 throughout; measured against the reference goldens the
// forward deviates 9e-7 relative and the NMSE trajectory 2e-7 (the direct kernel: 4e-7 / 1e-7).
//
// Work split: wave xi (0..3) owns transform row xi.  Because every row of B^T has exactly two non-zeros (+-1), the
// transformed input is never materialised: per 8-channel group a lane reads the 2 x 4 patch pixels its row needs
// from the staged LDS tile, forms R_j = d[ia][j] + sgn * d[ib][j] and the four V[xi][nu] = +-R_j +-R_j' (32 vector
// adds) and issues 16 MFMAs (4 nu x 4 channels) -- 2 vector instructions per MFMA, which matters because fp32 MFMA
// and VALU share the ALU on this chip.  After the K loops a wave reduces its four accumulators over nu with A
// (T_b = sum_nu M[xi][nu] A[nu][b], in registers), the T planes go through LDS, and every thread finishes
// Y[a][b] = sum_xi A^T[a][xi] T[xi][b] for one (tile, 4 channels): the four pixels of a 2x2 block -- which is also
// the window of the 2x2 mean pool.  Bias / residual / ELU / bilinear-add / pool / store follow as in the direct kernel.
#include <stdlib.h>
#include "conv_common.h"

namespace sbc {

template <int CIN, int COUT, int MB, bool P2>
__global__ __launch_bounds__(256) void conv_wino_kernel(ConvParams p) {
    constexpr int TM = 128 * MB;                 // output pixels per workgroup
    constexpr int NTW = 32 * MB;                 // Winograd tiles (2x2 output blocks) per workgroup
    constexpr int S = CIN + 4;
    constexpr int KG = CIN / 8;
    constexpr int NBLK = COUT / 32;
    constexpr int PH = NBLK;                     // phases: one 32-channel output block each (K loops, then output)
    constexpr int TS = 36;                       // floats per (tile) row of a T plane: 32 channels + 4 pad
    constexpr int NTHREADS = 256;
    constexpr int NPF_FULL = ((TM + 32) * (CIN / 4) + NTHREADS - 1) / NTHREADS;
    constexpr int NPF = NPF_FULL <= 10 ? NPF_FULL : 10;
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int xi = __builtin_amdgcn_readfirstlane(tid >> 6);         // transform row of this wave
    const int H = p.H, W = p.W, HW = H * W;
    const Dims<P2> dm{H, W, HW, p.hsh, p.wsh};
    const int khalf = 4 * (lane >> 5);
    const int col = lane & 31, rhalf = 4 * (lane >> 5);

    const TileGeom g = tile_geom(blockIdx.x, TM, p.B, dm, 1);
    stage_tile<CIN, NTHREADS, NPF, P2>(lds, p.in, p.stats, p.flags, g, dm, tid);
    // T planes [xi][b][tile][TS]: overlay the staged tile when there is a single phase, else live behind it
    float* const tl = PH == 1 ? lds : lds + (size_t)(g.multi ? TM + 1 : TM + 2 * W + 1) * S;

    // B^T rows: xi=0: d0 - d2, xi=1: d1 + d2, xi=2: d2 - d1, xi=3: d1 - d3   ->  R = d[ia] + sgn * d[ib]
    const int ia = xi == 0 ? 0 : xi == 2 ? 2 : 1;
    const int ib = xi == 0 ? 2 : xi == 1 ? 2 : xi == 2 ? 1 : 3;
    const float sgn = xi == 1 ? 1.f : -1.f;

    // per lane: LDS offsets of the 2 x 4 patch pixels of its tile (lane & 31) in each tile block
    const int Wt = W >> 1;                                            // tiles per image row
    const int r0 = dm.div_w(g.p0);                                    // first output row of the workgroup (even)
    const int zoff = g.nps * S + khalf;
    int off[MB][2][4];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        const int t = mb * 32 + (lane & 31);
        const int tr = P2 ? t >> (p.wsh - 1) : t / Wt, tc = t - tr * Wt;
        const int grow = r0 + 2 * tr;                                 // even output row of the tile
        const int h = dm.mod_h(grow);
        const bool tile_ok = grow < p.B * H;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int ii = k == 0 ? ia : ib;
            const int hh = h - 1 + ii;
            const bool rok = tile_ok && hh >= 0 && hh < H;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int ww = 2 * tc - 1 + j;
                off[mb][k][j] = (rok && ww >= 0 && ww < W) ? ((grow - 1 + ii - g.rs0) * W + ww) * S + khalf : zoff;
            }
        }
    }
    __syncthreads();                                                  // staged tile visible

    for (int ph = 0; ph < PH; ++ph) {
        const int nb = ph;                        // output-channel block of this phase
        f32x16 T[MB][2];
        f32x16 acc[4];
        // packed U: [xi*4 + nu][kg][nb][lane]
        const float4* wp = p.wpk + ((size_t)(xi * 4) * KG * NBLK + nb) * 64 + lane;
        // One software pipeline over all (tile block, 8-channel group) steps of the phase: the operands of step s+1
        // (4 B fragments from L2, 2 x 4 patch pixels from LDS) are requested before the transform + MFMAs of step
        // s, also across tile blocks; two statically indexed register sets, order pinned so the compiler's waits
        // are counted.
        float4 bS[2][4], dA[2][4], dB[2][4];
#pragma unroll
        for (int nu = 0; nu < 4; ++nu) bS[0][nu] = wp[(size_t)((nu * KG) * NBLK) * 64];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            dA[0][j] = *reinterpret_cast<const float4*>(__builtin_assume_aligned(lds + off[0][0][j], 16));
            dB[0][j] = *reinterpret_cast<const float4*>(__builtin_assume_aligned(lds + off[0][1][j], 16));
        }
#pragma unroll
        for (int s = 0; s < MB * KG; ++s) {
            const int mb = s / KG, kg = s % KG;
            const int cur = s & 1, nxt = cur ^ 1;
            const int sn = s + 1 < MB * KG ? s + 1 : s;
            const int mbn = sn / KG, kn = sn % KG;
            if (kg == 0) {
#pragma unroll
                for (int nu = 0; nu < 4; ++nu)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[nu][r] = 0.f;
            }
#pragma unroll
            for (int nu = 0; nu < 4; ++nu) bS[nxt][nu] = wp[(size_t)((nu * KG + kn) * NBLK) * 64];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                dA[nxt][j] = *reinterpret_cast<const float4*>(__builtin_assume_aligned(lds + off[mbn][0][j] + kn * 8, 16));
                dB[nxt][j] = *reinterpret_cast<const float4*>(__builtin_assume_aligned(lds + off[mbn][1][j] + kn * 8, 16));
            }
            __builtin_amdgcn_sched_barrier(0);
            float4 R[4], V[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 a4 = dA[cur][j], b4 = dB[cur][j];
                R[j].x = fmaf(sgn, b4.x, a4.x); R[j].y = fmaf(sgn, b4.y, a4.y);
                R[j].z = fmaf(sgn, b4.z, a4.z); R[j].w = fmaf(sgn, b4.w, a4.w);
            }
            // columns of B: nu=0: R0 - R2, nu=1: R1 + R2, nu=2: R2 - R1, nu=3: R1 - R3
            V[0] = make_float4(R[0].x - R[2].x, R[0].y - R[2].y, R[0].z - R[2].z, R[0].w - R[2].w);
            V[1] = make_float4(R[1].x + R[2].x, R[1].y + R[2].y, R[1].z + R[2].z, R[1].w + R[2].w);
            V[2] = make_float4(R[2].x - R[1].x, R[2].y - R[1].y, R[2].z - R[1].z, R[2].w - R[1].w);
            V[3] = make_float4(R[1].x - R[3].x, R[1].y - R[3].y, R[1].z - R[3].z, R[1].w - R[3].w);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int nu = 0; nu < 4; ++nu) {
                    const float4 b4 = bS[cur][nu];
                    const float av = j == 0 ? V[nu].x : j == 1 ? V[nu].y : j == 2 ? V[nu].z : V[nu].w;
                    const float bv = j == 0 ? b4.x : j == 1 ? b4.y : j == 2 ? b4.z : b4.w;
                    acc[nu] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[nu], 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
            if (kg == KG - 1) {
                // A^T = [[1, 1, 1, 0], [0, 1, -1, -1]] applied over nu
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    T[mb][0][r] = (acc[0][r] + acc[1][r]) + acc[2][r];
                    T[mb][1][r] = (acc[1][r] - acc[2][r]) - acc[3][r];
                }
            }
        }

        if (PH == 1) __syncthreads();             // all waves are done with the staged tile (T planes overlay it)
        {
            // T planes of this output block -> LDS [xi][b][tile][TS]
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    float* e = tl + ((size_t)((xi * 2 + b) * NTW + mb * 32 + rhalf)) * TS + col;
                    const f32x16 tv = T[mb][b];
#pragma unroll
                    for (int r = 0; r < 16; ++r) e[((r & 3) + 8 * (r >> 2)) * TS] = tv[r];
                }
            __syncthreads();
            // finish: one (tile, channel quad) per thread and round
#pragma unroll 1
            for (int task = tid; task < NTW * 8; task += NTHREADS) {
                const int t = task >> 3, c4 = task & 7;
                const int co = nb * 32 + c4 * 4;
                float4 y[2][2];                                       // [a][b]
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    float4 tx[4];
#pragma unroll
                    for (int x = 0; x < 4; ++x)
                        tx[x] = *reinterpret_cast<const float4*>(tl + ((size_t)((x * 2 + b) * NTW + t)) * TS + c4 * 4);
                    y[0][b] = make_float4((tx[0].x + tx[1].x) + tx[2].x, (tx[0].y + tx[1].y) + tx[2].y,
                                          (tx[0].z + tx[1].z) + tx[2].z, (tx[0].w + tx[1].w) + tx[2].w);
                    y[1][b] = make_float4((tx[1].x - tx[2].x) - tx[3].x, (tx[1].y - tx[2].y) - tx[3].y,
                                          (tx[1].z - tx[2].z) - tx[3].z, (tx[1].w - tx[2].w) - tx[3].w);
                }
                const int tr = P2 ? t >> (p.wsh - 1) : t / Wt, tc = t - tr * Wt;
                const int grow = r0 + 2 * tr;
                if (grow >= p.B * H) continue;
                if (p.bias) {
                    const float4 bv = *reinterpret_cast<const float4*>(p.bias + co);
#pragma unroll
                    for (int a = 0; a < 2; ++a)
#pragma unroll
                        for (int b = 0; b < 2; ++b) {
                            y[a][b].x += bv.x; y[a][b].y += bv.y; y[a][b].z += bv.z; y[a][b].w += bv.w;
                        }
                }
                if (p.flags & SBC_EPI_POOL) {
                    // ((((0 + a) + b) + c) + d) / 4 with a=[0::2,0::2] b=[1::2,0::2] c=[0::2,1::2] d=[1::2,1::2]
                    float4 v;
                    v.x = (((y[0][0].x + y[1][0].x) + y[0][1].x) + y[1][1].x) * 0.25f;
                    v.y = (((y[0][0].y + y[1][0].y) + y[0][1].y) + y[1][1].y) * 0.25f;
                    v.z = (((y[0][0].z + y[1][0].z) + y[0][1].z) + y[1][1].z) * 0.25f;
                    v.w = (((y[0][0].w + y[1][0].w) + y[0][1].w) + y[1][1].w) * 0.25f;
                    const int n = dm.div_h(grow), ho = (grow - n * H) >> 1;
                    const size_t o = ((size_t)(n * (H >> 1) + ho) * Wt + tc) * COUT + co;
                    if (p.res1) {
                        const float4 rr = ld_stream(p.res1 + o);
                        v.x = rr.x + v.x; v.y = rr.y + v.y; v.z = rr.z + v.z; v.w = rr.w + v.w;
                    }
                    st_stream(p.out + o, v);
                    continue;
                }
                size_t o[2][2];
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b) o[a][b] = ((size_t)(grow + a) * W + 2 * tc + b) * COUT + co;
                if (p.res1) {
                    float4 rr[2][2];
#pragma unroll
                    for (int a = 0; a < 2; ++a)
#pragma unroll
                        for (int b = 0; b < 2; ++b) rr[a][b] = ld_stream(p.res1 + o[a][b]);
                    if (p.flags & SBC_EPI_RES1_ELU) {
#pragma unroll
                        for (int a = 0; a < 2; ++a)
#pragma unroll
                            for (int b = 0; b < 2; ++b) rr[a][b] = elu4_acc(rr[a][b]);
                    }
                    if (p.res2) {
#pragma unroll
                        for (int a = 0; a < 2; ++a)
#pragma unroll
                            for (int b = 0; b < 2; ++b) {
                                const float4 r2 = ld_stream(p.res2 + o[a][b]);
                                rr[a][b].x = r2.x + rr[a][b].x; rr[a][b].y = r2.y + rr[a][b].y;
                                rr[a][b].z = r2.z + rr[a][b].z; rr[a][b].w = r2.w + rr[a][b].w;
                            }
                    }
#pragma unroll
                    for (int a = 0; a < 2; ++a)
#pragma unroll
                        for (int b = 0; b < 2; ++b) {
                            y[a][b].x += rr[a][b].x; y[a][b].y += rr[a][b].y;
                            y[a][b].z += rr[a][b].z; y[a][b].w += rr[a][b].w;
                        }
                }
                if (p.flags & SBC_EPI_UP) {
                    // F.interpolate(bilinear, align_corners=True) of `up` added on top (MSFBlock, layers.py:182-183)
                    const float sh = H > 1 ? (float)(p.up_h - 1) / (float)(H - 1) : 0.f;
                    const float sw = W > 1 ? (float)(p.up_w - 1) / (float)(W - 1) : 0.f;
                    const int n = dm.div_h(grow), hrow = grow - n * H;
                    const float* u = p.up + (size_t)n * p.up_h * p.up_w * COUT + co;
#pragma unroll
                    for (int a = 0; a < 2; ++a)
#pragma unroll
                        for (int b = 0; b < 2; ++b) {
                            const float fh = sh * (float)(hrow + a), fw = sw * (float)(2 * tc + b);
                            const int h0 = min((int)fh, p.up_h - 1), w0 = min((int)fw, p.up_w - 1);
                            const int h1 = min(h0 + 1, p.up_h - 1), w1 = min(w0 + 1, p.up_w - 1);
                            const float lh1 = fh - (float)h0, lw1 = fw - (float)w0;
                            const float lh0 = 1.f - lh1, lw0 = 1.f - lw1;
                            const float4 v00 = *reinterpret_cast<const float4*>(u + (size_t)(h0 * p.up_w + w0) * COUT);
                            const float4 v01 = *reinterpret_cast<const float4*>(u + (size_t)(h0 * p.up_w + w1) * COUT);
                            const float4 v10 = *reinterpret_cast<const float4*>(u + (size_t)(h1 * p.up_w + w0) * COUT);
                            const float4 v11 = *reinterpret_cast<const float4*>(u + (size_t)(h1 * p.up_w + w1) * COUT);
                            y[a][b].x += lh0 * (lw0 * v00.x + lw1 * v01.x) + lh1 * (lw0 * v10.x + lw1 * v11.x);
                            y[a][b].y += lh0 * (lw0 * v00.y + lw1 * v01.y) + lh1 * (lw0 * v10.y + lw1 * v11.y);
                            y[a][b].z += lh0 * (lw0 * v00.z + lw1 * v01.z) + lh1 * (lw0 * v10.z + lw1 * v11.z);
                            y[a][b].w += lh0 * (lw0 * v00.w + lw1 * v01.w) + lh1 * (lw0 * v10.w + lw1 * v11.w);
                        }
                }
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b) st_stream(p.out + o[a][b], y[a][b]);
            }
            if (ph + 1 < PH) __syncthreads();                          // T planes are rewritten next phase
        }
    }
}

// ------------------------------------------------------------------------------------------------ dispatch
template <int CIN, int COUT, int MB>
static int launch_wino(const ConvParams& p, hipStream_t stream, bool dry) {
    constexpr int TM = 128 * MB;
    constexpr int S = CIN + 4;
    constexpr int NBLK = COUT / 32;
    constexpr int PH = NBLK;
    const int HW = p.H * p.W;
    const bool multi = TM >= HW;
    const size_t staged = (size_t)(multi ? TM + 1 : TM + 2 * p.W + 1) * S * sizeof(float);
    const size_t tplanes = (size_t)8 * 32 * MB * 36 * sizeof(float);
    const size_t lds = PH == 1 ? max(staged, tplanes) : staged + tplanes;
    if (lds > 160 * 1024) return 1;
    auto kern = conv_wino_kernel<CIN, COUT, MB, true>;
    { const int rc = ensure_dyn_lds(reinterpret_cast<const void*>(kern), lds); if (rc) return rc; }
    if (dry) return SBC_OK;
    hipLaunchKernelGGL(kern, dim3((p.total_px + TM - 1) / TM), dim3(256), lds, stream, p);
    SBC_CHECK_HIP(hipGetLastError());
    return SBC_OK;
}

template <int CIN, int COUT>
static int launch_wino_sized(const ConvParams& p, hipStream_t stream, bool dry) {
    const int HW = p.H * p.W;
    auto fits = [&](int tm) { return tm % (2 * p.W) == 0 && (HW % tm == 0 || tm % HW == 0); };
    constexpr bool mb2_ok = (COUT == 32);          // 256-pixel tiles only where one output block keeps registers low
    static const bool force1 = getenv("SBC_WINO_MB1") != nullptr;                   // tuning aid
    // 256-pixel tiles for the full-resolution level only (>= 4096 tiles); 128-pixel tiles measure within 2 % below that
    // and keep one kernel symbol per level, so per-kernel profiler averages are per level too
    if (mb2_ok && !force1 && fits(256) && p.total_px >= 256L * 4096) return launch_wino<CIN, COUT, 2>(p, stream, dry);
    if (fits(128)) return launch_wino<CIN, COUT, 1>(p, stream, dry);
    if (mb2_ok && fits(256)) return launch_wino<CIN, COUT, 2>(p, stream, dry);
    return 1;
}

int launch_conv_wino(const ConvParams& p, int cin, int cout, hipStream_t stream, bool dry) {
    // power-of-two images with even sides only (every level the score network produces for Nt, Nr in {16, 64, 256})
    if (p.dil != 1 || p.hsh < 1 || p.wsh < 1) return 1;
    const int key = cin * 1000 + cout;
    switch (key) {
        case 32 * 1000 + 32: return launch_wino_sized<32, 32>(p, stream, dry);
        case 32 * 1000 + 64: return launch_wino_sized<32, 64>(p, stream, dry);
        case 64 * 1000 + 64: return launch_wino_sized<64, 64>(p, stream, dry);
        case 64 * 1000 + 32: return launch_wino_sized<64, 32>(p, stream, dry);
        case 64 * 1000 + 128: return launch_wino_sized<64, 128>(p, stream, dry);
        case 128 * 1000 + 128: return launch_wino_sized<128, 128>(p, stream, dry);
        case 128 * 1000 + 64: return launch_wino_sized<128, 64>(p, stream, dry);
        default: return 1;
    }
}

}  // namespace sbc

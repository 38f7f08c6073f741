// Fused NHWC convolution (1x1 / 3x3 / dilated 3x3, stride 1, zero padding = dilation) as an implicit GEMM on
// the fp32 matrix cores of gfx950 (v_mfma_f32_32x32x2_f32: exact IEEE fp32 FMA chain, 157 TFLOP/s peak).
//
// Replaces every nn.Conv2d the score network issues except the 2-channel begin/end convolutions
// (ncsnv2/models/layers.py:28-60) together with what surrounds it in the reference graph:
//   prologue  InstanceNorm++ affine and/or ELU applied while the input tile is staged into LDS
//             (ResidualBlock layers.py:444-449, RCUBlock :130-131);
//   epilogue  bias, residual adds (ResidualBlock :456, RCUBlock :133, CRPBlock :82), 2x2 mean pooling
//             (ConvMeanPool :311-312) and the bilinear(align_corners) resize-add of MSFBlock (:182-183).
//
// GEMM view: M = B*H*W pixels, N = COUT, K = taps*CIN.  A workgroup computes TM = 32*MT*WM pixels x COUT; wave
// (wm, wn) computes MT 32-pixel blocks x NT 32-channel blocks.  The A operand comes from the LDS tile (tile.h);
// the B operand (weights, pre-packed on the host into fragment order, weights.py:pack_conv_weight) is loaded
// straight from global/L2 as one 16-byte load per lane per (tap, 8-channel group) and feeds four MFMAs.
// K is walked in groups of 8 channels: within a group, lanes 0-31 supply channels g*8+0..3 and lanes 32-63
// channels g*8+4..7 as the two k-rows of four consecutive 32x32x2 MFMAs (any K permutation is legal as long
// as A and B agree).
#include "tile.h"

namespace sbc {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ConvParams {
    const float* __restrict__ in;
    float* __restrict__ out;
    const float4* __restrict__ wpk;
    const float* __restrict__ bias;
    const float* __restrict__ stats;
    const float* __restrict__ res1;
    const float* __restrict__ res2;
    const float* __restrict__ up;
    int B, H, W, dil, flags, up_h, up_w, total_px;
};

template <int CIN, int COUT, int KS, int MT, int NT, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN) void conv_mfma_kernel(ConvParams p) {
    constexpr int TM = 32 * MT * WM;
    constexpr int S = CIN + 4;
    constexpr int KG = CIN / 8;
    constexpr int NBLK = COUT / 32;
    constexpr int NTHREADS = 64 * WM * WN;
    static_assert(WN * NT == NBLK, "waves x blocks must cover COUT");
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int H = p.H, W = p.W;
    const TileGeom g = tile_geom(blockIdx.x, TM, p.B, H, W, KS == 3 ? p.dil : 0);

    stage_tile<CIN>(lds, p.in, p.stats, p.flags, g, H, W, tid, NTHREADS);

    // this lane's A rows: pixel (lane & 31) of each of the wave's MT blocks
    int row[MT], hh0[MT], ww0[MT];
    bool live[MT];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) {
        const int px = g.p0 + (wm * MT + mi) * 32 + (lane & 31);
        live[mi] = px < p.total_px;
        row[mi] = px / W;
        ww0[mi] = px - row[mi] * W;
        hh0[mi] = row[mi] % H;
    }
    const int khalf = 4 * (lane >> 5);

    f32x16 acc[MT][NT];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int ni = 0; ni < NT; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    __syncthreads();

    for (int tap = 0; tap < KS * KS; ++tap) {
        const int dh = KS == 3 ? (tap / 3 - 1) * p.dil : 0;
        const int dw = KS == 3 ? (tap % 3 - 1) * p.dil : 0;
        int aoff[MT];
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) {
            const int hh = hh0[mi] + dh, ww = ww0[mi] + dw;
            const bool ok = live[mi] && hh >= 0 && hh < H && ww >= 0 && ww < W;
            const int lp = ok ? (row[mi] + dh - g.rs0) * W + ww : g.nps;
            aoff[mi] = lp * S + khalf;
        }
        const float4* wp = p.wpk + ((size_t)tap * KG * NBLK + wn * NT) * 64 + lane;
#pragma unroll 2
        for (int kg = 0; kg < KG; ++kg) {
            float4 b[NT], a[MT];
#pragma unroll
            for (int ni = 0; ni < NT; ++ni) b[ni] = wp[(size_t)(kg * NBLK + ni) * 64];
#pragma unroll
            for (int mi = 0; mi < MT; ++mi) a[mi] = *reinterpret_cast<const float4*>(lds + aoff[mi] + kg * 8);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int mi = 0; mi < MT; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NT; ++ni) {
                        const float av = j == 0 ? a[mi].x : j == 1 ? a[mi].y : j == 2 ? a[mi].z : a[mi].w;
                        const float bv = j == 0 ? b[ni].x : j == 1 ? b[ni].y : j == 2 ? b[ni].z : b[ni].w;
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[mi][ni], 0, 0, 0);
                    }
        }
    }

    // ---------------------------------------------------------------- epilogue
    // accumulator map (32x32 MFMA): column = lane & 31 (output channel), row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const int col = lane & 31, rhalf = 4 * (lane >> 5);
    if (p.flags & SBC_EPI_POOL) {
        constexpr int ES = COUT + 1;
        __syncthreads();   // every wave is done reading the staged tile
#pragma unroll
        for (int mi = 0; mi < MT; ++mi)
#pragma unroll
            for (int ni = 0; ni < NT; ++ni) {
                const int co = (wn * NT + ni) * 32 + col;
                const float bv = p.bias ? p.bias[co] : 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int pl = (wm * MT + mi) * 32 + (r & 3) + 8 * (r >> 2) + rhalf;
                    lds[pl * ES + co] = acc[mi][ni][r] + bv;
                }
            }
        __syncthreads();
        const int Wo = W / 2, Ho = H / 2;
        const int r0 = g.p0 / W;                       // first global row of the tile (even)
        for (int idx = tid; idx < (TM / 4) * COUT; idx += NTHREADS) {
            const int co = idx % COUT, q = idx / COUT;
            const int qr = q / Wo, qc = q - qr * Wo;
            const int grow = r0 + 2 * qr;
            if (grow >= p.B * H) continue;
            const float* e = lds + ((2 * qr) * W + 2 * qc) * ES + co;
            // ((((0 + a) + b) + c) + d) / 4 with a=[0::2,0::2] b=[1::2,0::2] c=[0::2,1::2] d=[1::2,1::2]
            float v = (((e[0] + e[W * ES]) + e[ES]) + e[(W + 1) * ES]) * 0.25f;
            const int n = grow / H, ho = (grow - n * H) >> 1;
            const size_t o = ((size_t)(n * Ho + ho) * Wo + qc) * COUT + co;
            if (p.res1) v = p.res1[o] + v;
            p.out[o] = v;
        }
        return;
    }

    const int HW = H * W;
    float sh = 0.f, sw = 0.f;
    if (p.flags & SBC_EPI_UP) {
        sh = H > 1 ? (float)(p.up_h - 1) / (float)(H - 1) : 0.f;
        sw = W > 1 ? (float)(p.up_w - 1) / (float)(W - 1) : 0.f;
    }
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) {
            const int co = (wn * NT + ni) * 32 + col;
            const float bv = p.bias ? p.bias[co] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int px = g.p0 + (wm * MT + mi) * 32 + (r & 3) + 8 * (r >> 2) + rhalf;
                if (px >= p.total_px) continue;
                const size_t o = (size_t)px * COUT + co;
                float v = acc[mi][ni][r] + bv;
                if (p.res1) {
                    float rr = p.res1[o];
                    if (p.flags & SBC_EPI_RES1_ELU) rr = elu1(rr);
                    if (p.res2) rr = p.res2[o] + rr;
                    v = v + rr;
                }
                if (p.flags & SBC_EPI_UP) {
                    const int n = px / HW, rem = px - n * HW;
                    const int h = rem / W, w = rem - h * W;
                    const float fh = sh * (float)h, fw = sw * (float)w;
                    const int h0 = min((int)fh, p.up_h - 1), w0 = min((int)fw, p.up_w - 1);
                    const int h1 = min(h0 + 1, p.up_h - 1), w1 = min(w0 + 1, p.up_w - 1);
                    const float lh1 = fh - (float)h0, lw1 = fw - (float)w0;
                    const float lh0 = 1.f - lh1, lw0 = 1.f - lw1;
                    const float* u = p.up + (size_t)n * p.up_h * p.up_w * COUT + co;
                    const float v00 = u[(size_t)(h0 * p.up_w + w0) * COUT], v01 = u[(size_t)(h0 * p.up_w + w1) * COUT];
                    const float v10 = u[(size_t)(h1 * p.up_w + w0) * COUT], v11 = u[(size_t)(h1 * p.up_w + w1) * COUT];
                    v = v + (lh0 * (lw0 * v00 + lw1 * v01) + lh1 * (lw0 * v10 + lw1 * v11));
                }
                p.out[o] = v;
            }
        }
}

// ------------------------------------------------------------------------------------------------ dispatch
template <int CIN, int COUT, int KS, int MT, int NT, int WM, int WN>
static int launch_variant(const ConvParams& p, hipStream_t stream) {
    constexpr int TM = 32 * MT * WM;
    constexpr int S = CIN + 4;
    const int HW = p.H * p.W;
    SBC_REQUIRE(TM % p.W == 0 && (HW % TM == 0 || TM % HW == 0),
                "conv tile of %d pixels does not fit image %dx%d", TM, p.H, p.W);
    if (p.flags & SBC_EPI_POOL)
        SBC_REQUIRE(p.H % 2 == 0 && p.W % 2 == 0 && TM % (2 * p.W) == 0, "mean-pool needs even H, W (%dx%d)", p.H, p.W);
    const int halo_px = (TM >= HW || KS == 1) ? 0 : 2 * p.dil * p.W;
    size_t lds = (size_t)(TM + halo_px + 1) * S * sizeof(float);
    if (p.flags & SBC_EPI_POOL) lds = max(lds, (size_t)TM * (COUT + 1) * sizeof(float));
    SBC_REQUIRE(lds <= 160 * 1024, "conv tile needs %zu bytes of LDS (> 160 KiB)", lds);
    auto kern = conv_mfma_kernel<CIN, COUT, KS, MT, NT, WM, WN>;
    static size_t lds_attr = 0;   // per instantiation
    if (lds > lds_attr) {
        SBC_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        lds_attr = lds;
    }
    const int grid = (p.total_px + TM - 1) / TM;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * WM * WN), lds, stream, p);
    SBC_CHECK_HIP(hipGetLastError());
    return SBC_OK;
}

// Tile choice: as many pixels per workgroup as still leaves >= ~2 workgroups per CU (256 CUs), subject to the
// tile fitting the image (whole rows of one sample, or whole samples) and, for pooling, holding whole 2x2 blocks.
template <int CIN, int COUT, int KS>
static int launch_sized(const ConvParams& p, hipStream_t stream) {
    constexpr int NB = COUT / 32;
    const int HW = p.H * p.W;
    const long px = p.total_px;
    auto fits = [&](int tm) {
        return tm % p.W == 0 && (HW % tm == 0 || tm % HW == 0) && (!(p.flags & SBC_EPI_POOL) || tm % (2 * p.W) == 0);
    };
    int tm = 0;
    if (fits(256) && px >= 256L * 512) tm = 256;
    else if (fits(128) && px >= 128L * 512) tm = 128;
    else if (fits(64)) tm = 64;
    else if (fits(128)) tm = 128;
    else if (fits(256)) tm = 256;
    SBC_REQUIRE(tm != 0, "conv: no tile of 64/128/256 pixels fits image %dx%d (flags 0x%x)", p.H, p.W, p.flags);
    if (tm == 256) return launch_variant<CIN, COUT, KS, 2, NB, 4, 1>(p, stream);
    if constexpr (NB >= 2) {
        if (tm == 128) return launch_variant<CIN, COUT, KS, 2, NB / 2, 2, 2>(p, stream);
        if constexpr (NB >= 4) return launch_variant<CIN, COUT, KS, 2, 1, 1, 4>(p, stream);
        else return launch_variant<CIN, COUT, KS, 1, 1, 2, 2>(p, stream);
    } else {
        if (tm == 128) return launch_variant<CIN, COUT, KS, 1, 1, 4, 1>(p, stream);
        return launch_variant<CIN, COUT, KS, 1, 1, 2, 1>(p, stream);
    }
}

int launch_conv(const sbc_op& op, hipStream_t stream) {
    SBC_REQUIRE(op.in && op.out && op.weight, "conv: in/out/weight must be set");
    SBC_REQUIRE(op.B > 0 && op.H > 0 && op.W > 0, "conv: bad shape B=%d H=%d W=%d", op.B, op.H, op.W);
    SBC_REQUIRE(op.ksize == 1 || op.ksize == 3, "conv: ksize %d (only 1 and 3)", op.ksize);
    SBC_REQUIRE(op.dil >= 1, "conv: dilation %d", op.dil);
    SBC_REQUIRE(!(op.flags & SBC_PRO_NORM) || op.stats, "conv: PRO_NORM without stats");
    SBC_REQUIRE(!(op.flags & SBC_EPI_UP) || (op.up && op.up_h > 0 && op.up_w > 0), "conv: EPI_UP without up tensor");
    SBC_REQUIRE(!((op.flags & SBC_EPI_POOL) && (op.res2 || (op.flags & (SBC_EPI_UP | SBC_EPI_RES1_ELU)))),
                "conv: EPI_POOL combines only with res1");
    SBC_REQUIRE((long)op.B * op.H * op.W < (1L << 31) / 128, "conv: tensor too large for 32-bit pixel index");
    ConvParams p;
    p.in = (const float*)op.in; p.out = (float*)op.out; p.wpk = (const float4*)op.weight;
    p.bias = (const float*)op.bias; p.stats = (const float*)op.stats;
    p.res1 = (const float*)op.res1; p.res2 = (const float*)op.res2; p.up = (const float*)op.up;
    p.B = op.B; p.H = op.H; p.W = op.W; p.dil = op.dil; p.flags = op.flags;
    p.up_h = op.up_h; p.up_w = op.up_w; p.total_px = op.B * op.H * op.W;
    const int key = op.cin * 100000 + op.cout * 100 + op.ksize;
    switch (key) {
        case 32 * 100000 + 32 * 100 + 3: return launch_sized<32, 32, 3>(p, stream);
        case 32 * 100000 + 64 * 100 + 3: return launch_sized<32, 64, 3>(p, stream);
        case 32 * 100000 + 64 * 100 + 1: return launch_sized<32, 64, 1>(p, stream);
        case 64 * 100000 + 64 * 100 + 3: return launch_sized<64, 64, 3>(p, stream);
        case 64 * 100000 + 64 * 100 + 1: return launch_sized<64, 64, 1>(p, stream);
        case 64 * 100000 + 32 * 100 + 3: return launch_sized<64, 32, 3>(p, stream);
        case 64 * 100000 + 128 * 100 + 3: return launch_sized<64, 128, 3>(p, stream);
        case 128 * 100000 + 128 * 100 + 3: return launch_sized<128, 128, 3>(p, stream);
        case 128 * 100000 + 64 * 100 + 3: return launch_sized<128, 64, 3>(p, stream);
        default:
            set_error("conv: no kernel for cin=%d cout=%d ksize=%d (NCSNv2Deepest with ngf=32 needs "
                      "32/64/128 channels)", op.cin, op.cout, op.ksize);
            return SBC_ERR_UNSUPPORTED;
    }
}

}  // namespace sbc

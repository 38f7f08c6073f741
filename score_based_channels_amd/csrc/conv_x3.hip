// Fused NHWC convolution (1x1 / 3x3 / dilated 3x3) with fp32 results on the bf16 matrix cores of gfx950.
//
// Same operator, prologue and epilogue as conv_mfma.hip (nn.Conv2d of ncsnv2/models/layers.py:28-60 plus the
// InstanceNorm++/ELU prologue and bias/residual/pool/resize epilogue around it); only the multiply differs.  The fp32
// MFMA (v_mfma_f32_32x32x2_f32) runs at the vector-ALU rate, 1/16 of the bf16 matrix rate, and blocks the VALU while it
// does.  Here every fp32 operand is split EXACTLY into three bf16 terms
//       x = xh + xm + xl,   xh = bf16(x), xm = bf16(x - xh), xl = x - xh - xm        (8 + 8 + 8 significand bits)
// and x*w is evaluated as the six partial products whose weight is >= 2^-16 of the full product,
//       xl*wh + xh*wl + xm*wm + xm*wh + xh*wm + xh*wh,
// each an exact bf16 x bf16 product accumulated in fp32 by v_mfma_f32_32x32x16_bf16.  The three dropped terms
// (xm*wl, xl*wm, xl*wl) are below 2^-23 relative -- the size of the rounding an fp32 FMA chain makes anyway
// (tests: forward error vs the reference 0.9e-6, the fp32 kernels 1.0e-6).  Cost: 6 bf16 MFMAs of 32x32x16 (6 x 32
// cycles) replace 8 fp32 MFMAs of 32x32x2 (8 x 64 cycles) per 16 channels, i.e. 0.375x the matrix time, and the vector
// ALU is free for staging / epilogue work of the other waves meanwhile.
//
// Activations are split while the input tile is staged into LDS (tile.h: three [pixel][CIN + 8] bf16 planes); weights
// are split once on the host (sbc_pack_conv_weight_split) into B-operand fragment order
//       [tap][CIN/16][COUT/32][3 terms][64 lanes][8 bf16]:  lane l holds w[n*32 + (l&31)][g*16 + 8*(l>>5) + j].
#include <stdlib.h>
#include "conv_epilogue.h"

namespace sbc {

// TERMS = 3: the exact three-term bf16 split described above.  TERMS = 1 (conv_mode f16w, BASELINE config 5 "fp16
// score-net weights"): weights are fp16 (sbc_pack_conv_weight_f16), the fp32 activations are rounded to fp16 while they are
// staged, and a step is ONE v_mfma_f32_32x32x16_f16 with fp32 accumulation -- a sixth of the matrix work, a third of the LDS.
template <int CIN, int COUT, int KS, int MT, int NT, int WM, int WN, bool P2, int TERMS>
__global__ __launch_bounds__(64 * WM * WN) void conv_x3_kernel(ConvParams p) {
    constexpr int TM = 32 * MT * WM;
    constexpr int SH = CIN + 8;                         // bf16 elements per staged pixel
    constexpr int KG = CIN / 16;                        // K steps per tap
    constexpr int NBLK = COUT / 32;
    constexpr int NTHREADS = 64 * WM * WN;
    constexpr int TAPS = KS * KS;
    static_assert(WN * NT == NBLK, "waves x blocks must cover COUT");
    static_assert(KG % 2 == 0, "two register sets alternate per 16-channel group");
    constexpr int NPF_FULL = ((TM + 32) * (CIN / 4) + NTHREADS - 1) / NTHREADS;
    constexpr int NPF = NPF_FULL <= 10 ? NPF_FULL : 10;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    unsigned short* lds16 = reinterpret_cast<unsigned short*>(lds);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int H = p.H, W = p.W, HW = H * W;
    const Dims<P2> dm{H, W, HW, p.hsh, p.wsh};
    const int plane = p.plane;
    const int khalf = 8 * (lane >> 5);
    const uint4* wp = reinterpret_cast<const uint4*>(p.wpk) + (size_t)(wn * NT) * TERMS * 64 + lane;

    const TileGeom g = tile_geom(blockIdx.x, TM, p.B, dm, KS == 3 ? p.dil : 0);
    float descale = 1.f;
    const float* stats = p.stats;
    int n0 = -1;
    if (p.flags & SBC_PRO_NORM_SELF) {
        // whole samples per tile: the statistics are computed here, into LDS behind the planes / the epilogue tile (tile.h)
        float* st = lds + p.stats_off;
        self_stats_to_lds<CIN, NTHREADS, TM, P2>(st, p.in, p.stats, g, dm, tid);
        stats = st;
        n0 = 0;
    }
    if constexpr (TERMS == 2) {
        // conv_mode f16x2: scaled activations as two fp16 terms; the scales ride behind the packed weight (conv_common.h)
        const float4 tr = f16x2_trailer(p.wpk, TAPS * KG * NBLK * TERMS);
        StageScale ss{__builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, tr.x))), 0.f};
        descale = tr.y;
        const int sflags = p.flags | (__builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, tr.w)) != 0 ? SBC_PRO_ELU_ACC : 0);
        stage_tile_split<CIN, NTHREADS, NPF, P2, TERMS>(lds16, plane, p.in, stats, sflags, g, dm, tid, &ss, n0);
        f16x2_range_report(ss.amax, ss.scale, p.range_flag, p.calib);
    } else {
        stage_tile_split<CIN, NTHREADS, NPF, P2, TERMS>(lds16, plane, p.in, stats, p.flags, g, dm, tid, nullptr, n0);
    }

    // per 32-pixel block: this lane's LDS base offset and the 9-bit mask of taps inside the image (conv_mfma.hip)
    int abase[MT];
    unsigned amask[MT];
    const int zoff = g.nps * SH + khalf;
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) {
        const int px = g.p0 + (wm * MT + mi) * 32 + (lane & 31);
        const int row = dm.div_w(px), ww = dm.mod_w(px), hh = dm.mod_h(row);
        abase[mi] = ((row - g.rs0) * W + ww) * SH + khalf;
        unsigned rb = 2u, cb = 2u;
        if (KS == 3) {
            rb |= (hh - p.dil >= 0 ? 1u : 0u) | (hh + p.dil < H ? 4u : 0u);
            cb |= (ww - p.dil >= 0 ? 1u : 0u) | (ww + p.dil < W ? 4u : 0u);
        }
        unsigned m = ((rb & 1u) ? cb : 0u) | ((rb & 2u) ? cb << 3 : 0u) | ((rb & 4u) ? cb << 6 : 0u);
        if (KS == 1) m = 1u;
        amask[mi] = px < p.total_px ? m : 0u;
    }
    unsigned lane_or = 0;
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) lane_or |= amask[mi];
    unsigned tapmask = 0;
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
        if (__ballot((lane_or >> t) & 1u)) tapmask |= 1u << t;
    auto tap_offset = [&](int tap, int mi) {
        const int dh = KS == 3 ? (tap / 3 - 1) * p.dil : 0;
        const int dw = KS == 3 ? (tap % 3 - 1) * p.dil : 0;
        return ((amask[mi] >> tap) & 1u) ? abase[mi] + (dh * W + dw) * SH : zoff;
    };
    auto lds_frag = [&](int off, int term) {
        return *reinterpret_cast<const uint4*>(__builtin_assume_aligned(lds16 + off + term * plane, 16));
    };
    auto w_frag = [&](int it, int ni, int term) { return wp[((size_t)(it * NBLK + ni) * TERMS + term) * 64]; };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int ni = 0; ni < NT; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    // K loop, software-pipelined by one (tap, 16-channel group) step: the three weight fragments (global/L2) and three
    // activation fragments (LDS) of step it+1 are requested, then the MFMAs of step it are issued; two statically
    // indexed register sets alternate.
    uint4 aS[2][MT][TERMS], bS[2][NT][TERMS];          // 8 x 16-bit fragments (bf16 terms or fp16)
    auto mfma_step = [&](int cur) {
        if constexpr (TERMS == 1) {
#pragma unroll
            for (int mi = 0; mi < MT; ++mi)
#pragma unroll
                for (int ni = 0; ni < NT; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, aS[cur][mi][0]),
                                                                         __builtin_bit_cast(f16x8, bS[cur][ni][0]),
                                                                         acc[mi][ni], 0, 0, 0);
        } else if constexpr (TERMS == 2) {
            // two fp16 terms per operand: (l,h) (h,l) (h,h); the dropped (l,l) is below 2^-22 of the product
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int ta = q == 0 ? 1 : 0, tb = q == 1 ? 1 : 0;
#pragma unroll
                for (int mi = 0; mi < MT; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NT; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
                            __builtin_bit_cast(f16x8, aS[cur][mi][ta]), __builtin_bit_cast(f16x8, bS[cur][ni][tb]),
                            acc[mi][ni], 0, 0, 0);
            }
        } else {
            // partial products, smallest first: (l,h) (h,l) (m,m) (m,h) (h,m) (h,h)
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                const int ta = q == 0 ? 2 : (q == 2 || q == 3) ? 1 : 0;
                const int tb = q == 1 ? 2 : (q == 2 || q == 4) ? 1 : 0;
#pragma unroll
                for (int mi = 0; mi < MT; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NT; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                            __builtin_bit_cast(bf16x8, aS[cur][mi][ta]), __builtin_bit_cast(bf16x8, bS[cur][ni][tb]),
                            acc[mi][ni], 0, 0, 0);
            }
        }
    };
    if (tapmask == (1u << TAPS) - 1u) {
        // Every tap has work for this wave (always, except on images a few pixels high): the whole walk is unrolled, so
        // tap deltas, fragment indices and register sets are compile-time and the loop is loads + MFMAs only -- with
        // the dynamic tap walk below, ~35 scalar instructions per step sat between the MFMA blocks (-30 % K time).
#pragma unroll
        for (int ni = 0; ni < NT; ++ni)
#pragma unroll
            for (int t = 0; t < TERMS; ++t) bS[0][ni][t] = w_frag(0, ni, t);
        __syncthreads();                                               // staged tile visible
        int aoff[MT];
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) {
            aoff[mi] = tap_offset(0, mi);
#pragma unroll
            for (int t = 0; t < TERMS; ++t) aS[0][mi][t] = lds_frag(aoff[mi], t);
        }
        // (two nested fully unrolled loops: as ONE loop of 72 steps the 128-channel fp16 instantiations are peeled first and
        // then refused by the unroller as too large)
#pragma unroll
        for (int tap_c = 0; tap_c < TAPS; ++tap_c)
#pragma unroll
        for (int kg_c = 0; kg_c < KG; ++kg_c) {
            const int it = tap_c * KG + kg_c;
            const int cur = it & 1, nxt = cur ^ 1;
            if (it + 1 < TAPS * KG) {
                const int tap_n = (it + 1) / KG, kg_n = (it + 1) % KG;
                if (kg_n == 0) {
#pragma unroll
                    for (int mi = 0; mi < MT; ++mi) aoff[mi] = tap_offset(tap_n, mi);
                }
#pragma unroll
                for (int ni = 0; ni < NT; ++ni)
#pragma unroll
                    for (int t = 0; t < TERMS; ++t) bS[nxt][ni][t] = w_frag(it + 1, ni, t);
#pragma unroll
                for (int mi = 0; mi < MT; ++mi)
#pragma unroll
                    for (int t = 0; t < TERMS; ++t) aS[nxt][mi][t] = lds_frag(aoff[mi] + kg_n * 16, t);
            }
            mfma_step(cur);
        }
    } else {
        // general walk over the taps that have work (taps outside the image for the whole wave are skipped)
        int aoff[MT], aoff_n[MT];
        int tap = tapmask ? __builtin_ctz(tapmask) : TAPS;
        const int it0 = (tap < TAPS ? tap : 0) * KG;
#pragma unroll
        for (int ni = 0; ni < NT; ++ni)
#pragma unroll
            for (int t = 0; t < TERMS; ++t) bS[0][ni][t] = w_frag(it0, ni, t);
        __syncthreads();                                               // staged tile visible
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) {
            aoff[mi] = tap_offset(tap < TAPS ? tap : 0, mi);
#pragma unroll
            for (int t = 0; t < TERMS; ++t) aS[0][mi][t] = lds_frag(aoff[mi], t);
        }
#pragma unroll 1
        while (tap < TAPS) {
            const unsigned rest = tapmask >> (tap + 1);
            const int tap_n = rest ? tap + 1 + __builtin_ctz(rest) : tap;
#pragma unroll
            for (int mi = 0; mi < MT; ++mi) aoff_n[mi] = tap_offset(tap_n, mi);
#pragma unroll
            for (int kg = 0; kg < KG; ++kg) {
                const int cur = kg & 1, nxt = cur ^ 1;
                const int it_n = kg + 1 < KG ? tap * KG + kg + 1 : tap_n * KG + (tap_n == tap ? KG - 1 : 0);
#pragma unroll
                for (int ni = 0; ni < NT; ++ni)
#pragma unroll
                    for (int t = 0; t < TERMS; ++t) bS[nxt][ni][t] = w_frag(it_n, ni, t);
#pragma unroll
                for (int mi = 0; mi < MT; ++mi)
#pragma unroll
                    for (int t = 0; t < TERMS; ++t)
                        aS[nxt][mi][t] = lds_frag(kg + 1 < KG ? aoff[mi] + (kg + 1) * 16 : aoff_n[mi], t);
                __builtin_amdgcn_sched_barrier(0);
                mfma_step(cur);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (tap_n == tap) break;
#pragma unroll
            for (int mi = 0; mi < MT; ++mi) aoff[mi] = aoff_n[mi];
            tap = tap_n;
        }
    }

    // ---------------------------------------------------------------- epilogue (conv_epilogue.h)
    __syncthreads();   // every wave is done reading the staged planes
    conv_acc_to_lds<COUT, MT, NT>(lds, acc, p.bias, wm, wn, lane, descale);
    __syncthreads();
    conv_epilogue<COUT, TM, NTHREADS, P2>(lds, p, g, dm, tid);
}

// ------------------------------------------------------------------------------------------------ dispatch
template <int CIN, int COUT, int KS, int MT, int NT, int WM, int WN, int TERMS>
static size_t x3_lds_bytes(const ConvParams& p, int* plane_out) {
    constexpr int TM = 32 * MT * WM;
    const int HW = p.H * p.W;
    const int halo_px = (TM >= HW || KS == 1) ? 0 : 2 * p.dil * p.W;
    const int plane = (TM + halo_px + 1) * (CIN + 8);
    if (plane_out) *plane_out = plane;
    const size_t staged = (size_t)TERMS * plane * sizeof(unsigned short);
    // (+ 2 KB behind the epilogue tile: the reduction scratch of SBC_EPI_MOMENTS_OUT, conv_epilogue.h)
    const size_t epi = (size_t)TM * (COUT + 4) * sizeof(float) + ((p.flags & SBC_EPI_MOMENTS_OUT) ? 2048 : 0);
    return staged > epi ? staged : epi;
}

template <int CIN, int COUT, int KS, int MT, int NT, int WM, int WN, bool P2, int TERMS>
static int launch_kernel(ConvParams p, hipStream_t stream, bool dry) {
    constexpr int TM = 32 * MT * WM;
    size_t lds = x3_lds_bytes<CIN, COUT, KS, MT, NT, WM, WN, TERMS>(p, &p.plane);
    if (p.flags & SBC_PRO_NORM_SELF) {
        const int HW = p.H * p.W;
        SBC_REQUIRE(TM % HW == 0, "conv: SBC_PRO_NORM_SELF needs tiles of whole samples (H*W = %d, tile %d)", HW, TM);
        lds = (lds + 15) / 16 * 16;
        p.stats_off = (int)(lds / sizeof(float));
        lds += (size_t)(TM / HW) * (3 * CIN + 2) * sizeof(float);
    }
    SBC_REQUIRE(lds <= 160 * 1024, "conv tile needs %zu bytes of LDS (> 160 KiB)", lds);
    auto kern = conv_x3_kernel<CIN, COUT, KS, MT, NT, WM, WN, P2, TERMS>;
    { const int rc = ensure_dyn_lds(reinterpret_cast<const void*>(kern), lds); if (rc) return rc; }
    if (dry) return SBC_OK;
    const int grid = (p.total_px + TM - 1) / TM;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * WM * WN), lds, stream, p);
    SBC_CHECK_HIP(hipGetLastError());
    return SBC_OK;
}

template <int CIN, int COUT, int KS, int MT, int NT, int WM, int WN>
static int launch_variant(const ConvParams& p, hipStream_t stream, bool dry) {
    const bool p2 = p.hsh >= 0 && p.wsh >= 1;
    if (p.flags & SBC_CONV_F16W) {
        if (p2) return launch_kernel<CIN, COUT, KS, MT, NT, WM, WN, true, 1>(p, stream, dry);
        return launch_kernel<CIN, COUT, KS, MT, NT, WM, WN, false, 1>(p, stream, dry);
    }
    if (p.flags & SBC_CONV_F16X2) {
        if (p2) return launch_kernel<CIN, COUT, KS, MT, NT, WM, WN, true, 2>(p, stream, dry);
        return launch_kernel<CIN, COUT, KS, MT, NT, WM, WN, false, 2>(p, stream, dry);
    }
    if (p2) return launch_kernel<CIN, COUT, KS, MT, NT, WM, WN, true, 3>(p, stream, dry);
    return launch_kernel<CIN, COUT, KS, MT, NT, WM, WN, false, 3>(p, stream, dry);
}

// Tile choice: the largest tile that fits the image, keeps >= 2 workgroups per CU worth of work (256 CUs) and leaves
// room for two workgroups in the 160 KiB of LDS.
template <int CIN, int COUT, int KS>
static int launch_sized(const ConvParams& p, hipStream_t stream, bool dry) {
    constexpr int NB = COUT / 32;
    const int HW = p.H * p.W;
    const long px = p.total_px;
    auto fits = [&](int tm) {
        return tm % p.W == 0 && (HW % tm == 0 || tm % HW == 0) && (!(p.flags & SBC_EPI_POOL) || tm % (2 * p.W) == 0);
    };
    const int terms = (p.flags & SBC_CONV_F16W) ? 1 : (p.flags & SBC_CONV_F16X2) ? 2 : 3;
    auto lds_of = [&](int tm) -> size_t {
        const int halo_px = (tm >= HW || KS == 1) ? 0 : 2 * p.dil * p.W;
        const size_t staged = (size_t)terms * (tm + halo_px + 1) * (CIN + 8) * 2, epi = (size_t)tm * (COUT + 4) * 4;
        return staged > epi ? staged : epi;
    };
    auto good = [&](int tm) { return fits(tm) && px >= (long)tm * 512 && lds_of(tm) <= 80 * 1024; };
    static const int force = getenv("SBC_TILE") ? atoi(getenv("SBC_TILE")) : 0;     // tuning aid
    int tm = 0;
    if (force && fits(force)) tm = force;
    else if ((p.flags & SBC_EPI_MOMENTS_OUT) && !good(256)) {
        // tile moments come out of the 256-thread variants (whole 128-pixel tiles per pass): never the 64-pixel tile
        SBC_REQUIRE(COUT == 32 && fits(128), "conv: SBC_EPI_MOMENTS_OUT on the direct kernel needs 32 output channels and 128-pixel tiles (%dx%d)", p.H, p.W);
        tm = 128;
    }
    else if (good(256)) tm = 256;
    else if (good(128)) tm = 128;
    else if (fits(64)) tm = 64;
    else if (fits(128)) tm = 128;
    else if (fits(256)) tm = 256;
    SBC_REQUIRE(tm != 0, "conv: no tile of 64/128/256 pixels fits image %dx%d (flags 0x%x)", p.H, p.W, p.flags);
    if (tm == 256) return launch_variant<CIN, COUT, KS, 2, NB, 4, 1>(p, stream, dry);
    if constexpr (NB >= 2) {
        if (tm == 128) return launch_variant<CIN, COUT, KS, 2, NB / 2, 2, 2>(p, stream, dry);
        if constexpr (NB >= 4) return launch_variant<CIN, COUT, KS, 2, 1, 1, 4>(p, stream, dry);
        else return launch_variant<CIN, COUT, KS, 1, 1, 2, 2>(p, stream, dry);
    } else {
        if (tm == 128) return launch_variant<CIN, COUT, KS, 1, 1, 4, 1>(p, stream, dry);
        return launch_variant<CIN, COUT, KS, 1, 1, 2, 1>(p, stream, dry);
    }
}

int launch_conv_x3(const ConvParams& p, int cin, int cout, int ksize, hipStream_t stream, bool dry) {
    const int key = cin * 100000 + cout * 100 + ksize;
    switch (key) {
        case 32 * 100000 + 32 * 100 + 3: return launch_sized<32, 32, 3>(p, stream, dry);
        case 32 * 100000 + 64 * 100 + 3: return launch_sized<32, 64, 3>(p, stream, dry);
        case 32 * 100000 + 64 * 100 + 1: return launch_sized<32, 64, 1>(p, stream, dry);
        case 64 * 100000 + 64 * 100 + 3: return launch_sized<64, 64, 3>(p, stream, dry);
        case 64 * 100000 + 64 * 100 + 1: return launch_sized<64, 64, 1>(p, stream, dry);
        case 64 * 100000 + 32 * 100 + 3: return launch_sized<64, 32, 3>(p, stream, dry);
        case 64 * 100000 + 32 * 100 + 1: return launch_sized<64, 32, 1>(p, stream, dry);     // adjoint of the 32 -> 64 shortcut
        case 64 * 100000 + 128 * 100 + 3: return launch_sized<64, 128, 3>(p, stream, dry);
        case 128 * 100000 + 128 * 100 + 3: return launch_sized<128, 128, 3>(p, stream, dry);
        case 128 * 100000 + 64 * 100 + 3: return launch_sized<128, 64, 3>(p, stream, dry);
        default:
            set_error("conv: no kernel for cin=%d cout=%d ksize=%d (NCSNv2Deepest with ngf=32 needs "
                      "32/64/128 channels)", cin, cout, ksize);
            return SBC_ERR_UNSUPPORTED;
    }
}

}  // namespace sbc

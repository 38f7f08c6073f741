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
// GEMM view: M = B*H*W pixels, N = COUT, K = taps*CIN.  A workgroup computes one tile of TM = 32*MT*WM pixels x
// COUT; wave (wm, wn) computes MT 32-pixel blocks x NT 32-channel blocks.  The A operand comes from the LDS tile
// (tile.h); the B operand (weights, pre-packed on the host into fragment order, weights.py:pack_conv_weight) is
// loaded straight from global/L2 as one 16-byte load per lane per (tap, 8-channel group) and feeds four MFMAs.
// K is walked in groups of 8 channels: within a group, lanes 0-31 supply channels g*8+0..3 and lanes 32-63
// channels g*8+4..7 as the two k-rows of four consecutive 32x32x2 MFMAs (any K permutation is legal as long as A
// and B agree).
//
// On gfx950 the fp32 MFMA executes on the vector ALU (tools/mfma_valu_coissue.hip: MFMA 146 TF alone, v_fma 130 TF
// alone, 83 + 42 TF interleaved -- integer VALU and v_exp serialise with it in the same way), so every vector
// instruction outside the MFMAs is paid in MFMA time: index math uses shifts (power-of-two H, W build), per-tap
// offsets come from a bit mask, ELU uses the hardware exponential, taps that are outside the image for a whole
// wave are skipped, and all epilogue traffic is 16 bytes per lane.
//
// Experiments around this kernel that did not pay off (persistent workgroups, weights in LDS, staggering, ...) are
// recorded in DESIGN.md section 5.
#include <stdlib.h>
#include <string.h>
#include "conv_epilogue.h"

namespace sbc {

template <int CIN, int COUT, int KS, int MT, int NT, int WM, int WN, bool P2>
__global__ __launch_bounds__(64 * WM * WN) void conv_mfma_kernel(ConvParams p) {
    constexpr int TM = 32 * MT * WM;
    constexpr int S = CIN + 4;
    constexpr int KG = CIN / 8;
    constexpr int NBLK = COUT / 32;
    constexpr int NTHREADS = 64 * WM * WN;
    constexpr int TAPS = KS * KS;
    static_assert(WN * NT == NBLK, "waves x blocks must cover COUT");
    static_assert(KG % 2 == 0, "two register sets alternate per 8-channel group");
    // 16-byte staging requests in flight per thread: the tile + a 32-pixel halo allowance, at most 10 at a time
    constexpr int NPF_FULL = ((TM + 32) * (CIN / 4) + NTHREADS - 1) / NTHREADS;
    constexpr int NPF = NPF_FULL <= 10 ? NPF_FULL : 10;
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int H = p.H, W = p.W, HW = H * W;
    const Dims<P2> dm{H, W, HW, p.hsh, p.wsh};
    const int khalf = 4 * (lane >> 5);
    const float4* wp = p.wpk + (size_t)(wn * NT) * 64 + lane;      // + it * NBLK * 64 per K step

    const TileGeom g = tile_geom(blockIdx.x, TM, p.B, dm, KS == 3 ? p.dil : 0);
    stage_tile<CIN, NTHREADS, NPF, P2>(lds, p.in, p.stats, p.flags, g, dm, tid);

    // This lane's A rows: pixel (lane & 31) of each of the wave's MT blocks.  Per block one LDS base offset and a
    // 9-bit mask of the taps that stay inside the image, so that the per-tap work inside the MFMA loop is a bit
    // test and a select (tap deltas are wave-uniform); out-of-image taps read the shared zero pixel.
    int abase[MT];
    unsigned amask[MT];
    const int zoff = g.nps * S + khalf;
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) {
        const int px = g.p0 + (wm * MT + mi) * 32 + (lane & 31);
        const int row = dm.div_w(px), ww = dm.mod_w(px), hh = dm.mod_h(row);
        abase[mi] = ((row - g.rs0) * W + ww) * S + khalf;
        // 3 row bits x 3 column bits -> 9 tap bits (bit t = kh*3 + kw)
        unsigned rb = 2u, cb = 2u;                       // centre row / column always inside
        if (KS == 3) {
            rb |= (hh - p.dil >= 0 ? 1u : 0u) | (hh + p.dil < H ? 4u : 0u);
            cb |= (ww - p.dil >= 0 ? 1u : 0u) | (ww + p.dil < W ? 4u : 0u);
        }
        unsigned m = ((rb & 1u) ? cb : 0u) | ((rb & 2u) ? cb << 3 : 0u) | ((rb & 4u) ? cb << 6 : 0u);
        if (KS == 1) m = 1u;
        amask[mi] = px < p.total_px ? m : 0u;
    }
    // taps that are outside the image for every lane of this wave contribute exact zeros: skip them (at 8x2 with
    // dilation 2 / 4 that is 6 of the 9 taps)
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
        return ((amask[mi] >> tap) & 1u) ? abase[mi] + (dh * W + dw) * S : zoff;
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int ni = 0; ni < NT; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    // K loop over the taps that have work, software-pipelined by one (tap, 8-channel group) step with two statically
    // indexed register sets: the B fragment (global/L2) and the A fragment (LDS) of step it+1 are requested, then the
    // MFMAs of step it are issued.  sched_barrier pins that order so the waits the compiler inserts in front of the
    // MFMAs are counted ("everything but the requests just issued"), never a full drain.
    int aoff[MT], aoff_n[MT];
    float4 aS[2][MT], bS[2][NT];
    int tap = tapmask ? __builtin_ctz(tapmask) : TAPS;                 // first tap with work
    const int it0 = (tap < TAPS ? tap : 0) * KG;
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) bS[0][ni] = wp[(size_t)(it0 * NBLK + ni) * 64];
    __syncthreads();                                                   // staged tile visible
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) {
        aoff[mi] = tap_offset(tap < TAPS ? tap : 0, mi);
        aS[0][mi] = *reinterpret_cast<const float4*>(__builtin_assume_aligned(lds + aoff[mi], 16));
    }
#pragma unroll 1
    while (tap < TAPS) {
        const unsigned rest = tapmask >> (tap + 1);
        const int tap_n = rest ? tap + 1 + __builtin_ctz(rest) : tap;   // next tap with work (or stay)
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) aoff_n[mi] = tap_offset(tap_n, mi);
#pragma unroll
        for (int kg = 0; kg < KG; ++kg) {
            const int cur = kg & 1, nxt = cur ^ 1;
            const int it_n = kg + 1 < KG ? tap * KG + kg + 1 : tap_n * KG + (tap_n == tap ? KG - 1 : 0);
#pragma unroll
            for (int ni = 0; ni < NT; ++ni) bS[nxt][ni] = wp[(size_t)(it_n * NBLK + ni) * 64];
#pragma unroll
            for (int mi = 0; mi < MT; ++mi)
                aS[nxt][mi] = *reinterpret_cast<const float4*>(
                    __builtin_assume_aligned(lds + (kg + 1 < KG ? aoff[mi] + (kg + 1) * 8 : aoff_n[mi]), 16));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int mi = 0; mi < MT; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NT; ++ni) {
                        const float4 a4 = aS[cur][mi], b4 = bS[cur][ni];
                        const float av = j == 0 ? a4.x : j == 1 ? a4.y : j == 2 ? a4.z : a4.w;
                        const float bv = j == 0 ? b4.x : j == 1 ? b4.y : j == 2 ? b4.z : b4.w;
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[mi][ni], 0, 0, 0);
                    }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (tap_n == tap) break;
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) aoff[mi] = aoff_n[mi];
        tap = tap_n;
    }

    // ---------------------------------------------------------------- epilogue (conv_epilogue.h)
    __syncthreads();   // every wave is done reading the staged tile
    conv_acc_to_lds<COUT, MT, NT>(lds, acc, p.bias, wm, wn, lane);
    __syncthreads();
    conv_epilogue<COUT, TM, NTHREADS, P2>(lds, p, g, dm, tid);
}

// ------------------------------------------------------------------------------------------------ dispatch
template <int CIN, int COUT, int KS, int MT, int NT, int WM, int WN, bool P2>
static int launch_kernel(const ConvParams& p, size_t lds, hipStream_t stream, bool dry) {
    constexpr int TM = 32 * MT * WM;
    auto kern = conv_mfma_kernel<CIN, COUT, KS, MT, NT, WM, WN, P2>;
    { const int rc = ensure_dyn_lds(reinterpret_cast<const void*>(kern), lds); if (rc) return rc; }
    if (dry) return SBC_OK;
    const int grid = (p.total_px + TM - 1) / TM;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * WM * WN), lds, stream, p);
    SBC_CHECK_HIP(hipGetLastError());
    return SBC_OK;
}

template <int CIN, int COUT, int KS, int MT, int NT, int WM, int WN>
static int launch_variant(const ConvParams& p, hipStream_t stream, bool dry) {
    constexpr int TM = 32 * MT * WM;
    constexpr int S = CIN + 4;
    const int HW = p.H * p.W;
    SBC_REQUIRE(TM % p.W == 0 && (HW % TM == 0 || TM % HW == 0),
                "conv tile of %d pixels does not fit image %dx%d", TM, p.H, p.W);
    if (p.flags & SBC_EPI_POOL)
        SBC_REQUIRE(p.H % 2 == 0 && p.W % 2 == 0 && TM % (2 * p.W) == 0, "mean-pool needs even H, W (%dx%d)", p.H, p.W);
    const int halo_px = (TM >= HW || KS == 1) ? 0 : 2 * p.dil * p.W;
    size_t lds = (size_t)(TM + halo_px + 1) * S * sizeof(float);
    lds = max(lds, (size_t)TM * (COUT + 4) * sizeof(float));       // epilogue transposes through LDS
    SBC_REQUIRE(lds <= 160 * 1024, "conv tile needs %zu bytes of LDS (> 160 KiB)", lds);
    if (p.hsh >= 0 && p.wsh >= 1) return launch_kernel<CIN, COUT, KS, MT, NT, WM, WN, true>(p, lds, stream, dry);
    return launch_kernel<CIN, COUT, KS, MT, NT, WM, WN, false>(p, lds, stream, dry);
}

// Tile choice: as many pixels per workgroup as still leaves >= ~2 workgroups per CU (256 CUs), subject to the
// tile fitting the image (whole rows of one sample, or whole samples) and, for pooling, holding whole 2x2 blocks.
template <int CIN, int COUT, int KS>
static int launch_sized(const ConvParams& p, hipStream_t stream, bool dry) {
    constexpr int NB = COUT / 32;
    const int HW = p.H * p.W;
    const long px = p.total_px;
    auto fits = [&](int tm) {
        return tm % p.W == 0 && (HW % tm == 0 || tm % HW == 0) && (!(p.flags & SBC_EPI_POOL) || tm % (2 * p.W) == 0);
    };
    static const int force = getenv("SBC_TILE") ? atoi(getenv("SBC_TILE")) : 0;     // tuning aid
    int tm = 0;
    if (force && fits(force)) tm = force;
    else if (fits(256) && px >= 256L * 512) tm = 256;
    else if (fits(128) && px >= 128L * 512) tm = 128;
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

// dry = true: resolve the kernel variant and set its function attributes without launching (done at plan creation
// so that nothing but kernel launches happens inside a hipGraph capture)
int launch_conv(const sbc_op& op, hipStream_t stream, bool dry) {
    static const bool f32_only = getenv("SBC_CONV_MODE") && !strcmp(getenv("SBC_CONV_MODE"), "f32");   // A/B aid
    const bool x3 = op.weight_split && !f32_only;
    SBC_REQUIRE(op.in && op.out && (op.weight || x3), "conv: in/out/weight must be set");
    SBC_REQUIRE(op.B > 0 && op.H > 0 && op.W > 0, "conv: bad shape B=%d H=%d W=%d", op.B, op.H, op.W);
    SBC_REQUIRE(op.ksize == 1 || op.ksize == 3, "conv: ksize %d (only 1 and 3)", op.ksize);
    SBC_REQUIRE(op.dil >= 1, "conv: dilation %d", op.dil);
    SBC_REQUIRE(!(op.flags & SBC_PRO_NORM) || op.stats, "conv: PRO_NORM without stats");
    SBC_REQUIRE(!(op.flags & SBC_EPI_UP) || (op.up && op.up_h > 0 && op.up_w > 0), "conv: EPI_UP without up tensor");
    SBC_REQUIRE(!(op.flags & SBC_EPI_ELUGRAD) || (op.res2 && !(op.flags & (SBC_EPI_POOL | SBC_EPI_RES1_ELU))),
                "conv: EPI_ELUGRAD needs res2 (the forward input) and excludes EPI_POOL / EPI_RES1_ELU");
    SBC_REQUIRE(!((op.flags & SBC_EPI_POOL) && (op.res2 || (op.flags & (SBC_EPI_UP | SBC_EPI_RES1_ELU)))),
                "conv: EPI_POOL combines only with res1");
    // pixel indices are int32, element offsets of the epilogues uint32: bound the larger of the two tensors
    SBC_REQUIRE((long)op.B * op.H * op.W * (op.cin > op.cout ? op.cin : op.cout) <= 0x7fffffffL,
                "conv: %d x %d x %d x %d elements exceed the 32-bit element index", op.B, op.H, op.W,
                op.cin > op.cout ? op.cin : op.cout);
    ConvParams p;
    p.in = (const float*)op.in; p.out = (float*)op.out; p.wpk = (const float4*)op.weight;
    p.bias = (const float*)op.bias; p.stats = (const float*)op.stats;
    p.res1 = (const float*)op.res1; p.res2 = (const float*)op.res2; p.up = (const float*)op.up;
    p.B = op.B; p.H = op.H; p.W = op.W; p.dil = op.dil; p.flags = op.flags;
    p.up_h = op.up_h; p.up_w = op.up_w; p.total_px = op.B * op.H * op.W;
    p.hsh = log2_exact(op.H); p.wsh = log2_exact(op.W);
    p.plane = 0; p.stats_off = 0; p.top = op.tag == 1;
    p.pm_out = (float*)op.aux;
    p.range_flag = nullptr;
    p.calib = (float*)op.calib;
    if (op.flags & SBC_CONV_F16X2) {
        SBC_REQUIRE(x3 && !(op.flags & SBC_CONV_F16W), "conv: SBC_CONV_F16X2 needs weight_split (sbc_pack_conv_weight_f16x2) and excludes SBC_CONV_F16W");
        unsigned* word = nullptr;
        const int rc = range_flag_ptr(&word);
        if (rc) return rc;
        p.range_flag = word;
    }
    const bool moments = (op.flags & SBC_EPI_MOMENTS_OUT) != 0;
    if (moments) {
        // tile moments of the output are written by the Winograd split kernels' 128-pixel variants with 32 / 64 output channels (tile.h):
        // whole 128-pixel tiles inside one sample, unpooled output
        // (round 6: in conv_mode f16w the direct kernel writes them for 32 output channels, conv_epilogue.h)
        SBC_REQUIRE((op.weight_wino_split || ((op.flags & SBC_CONV_F16W) && x3 && op.cout == 32)) && op.ksize == 3 && op.dil == 1 && op.aux && (op.cout == 32 || op.cout == 64) && (op.H * op.W) % 128 == 0 &&
                    128 % (2 * op.W) == 0 && !(op.flags & SBC_EPI_POOL),
                    "conv: EPI_MOMENTS_OUT needs the Winograd split kernel (weight_wino_split), aux, 32 / 64 output channels, whole 128-pixel "
                    "tiles per sample and an unpooled output");
    }
    SBC_REQUIRE(!(op.flags & SBC_PRO_NORM_MOMENTS), "conv: SBC_PRO_NORM_MOMENTS belongs to SBC_OP_INORM_STATS (statistics from tile moments)");
    if (op.flags & SBC_PRO_NORM_SELF) {
        const int hw = op.H * op.W;
        SBC_REQUIRE((op.flags & SBC_PRO_NORM) && (x3 || (op.weight_wino_split && !f32_only)) && op.ksize == 3 && hw <= 64 && !(hw & (hw - 1)) && p.hsh >= 0 && p.wsh >= 1 &&
                    !(op.flags & SBC_EPI_ELUGRAD),
                    "conv: SBC_PRO_NORM_SELF needs SBC_PRO_NORM, a matrix-core weight form (weight_split), a 3x3 kernel and a "
                    "power-of-two image of at most 64 pixels (got %dx%d, ksize %d)", op.H, op.W, op.ksize);
    }
    static const bool no_wx3 = getenv("SBC_NO_WX3") != nullptr;                    // A/B aid: direct split-bf16 kernel everywhere
    const bool direct_only = (op.flags & SBC_EPI_ELUGRAD) != 0;       // the fp32 Winograd kernel's epilogue does not know the flag
    if ((op.flags & SBC_CONV_F16X2) && !f32_only) {
        const int rc = launch_conv_dp(op, p.range_flag, stream, dry);
        if (rc <= 0) return rc;                                                    // launched (0) or failed (< 0)
    }
    // conv_mode f16w (BASELINE config 5): ONE matrix instruction per product, so Winograd's 2.25 x fewer products buy nothing and its
    // transforms cost vector time -- measured at 256 x 64, 1024 samples (profiles/r06_big_direct_vs_winograd.txt): 32 -> 32 917 against
    // 1291 us, 64 -> 64 174 / 199, 128 -> 128 158 / 279, pooled 32 -> 64 1684 / 2143.  The direct kernel takes every f16w layer except
    // the 64-channel producers of tile moments (its epilogue writes moments for 32 output channels only).
    const bool f16w_direct = (op.flags & SBC_CONV_F16W) && x3 && !(moments && op.cout != 32);
    if (op.weight_wino_split && !f32_only && !no_wx3 && !f16w_direct && op.ksize == 3 && op.dil == 1) {
        ConvParams pw = p;
        pw.wpk = (const float4*)op.weight_wino_split;
#ifdef SBC_WITH_WP   // tools/experiments/conv_wp.hip (round 4: 64 -> 64 with the transformed filter resident in registers; measured slower)
        {
            const int rc = launch_conv_wp(pw, op.cin, op.cout, stream, dry);
            if (rc <= 0) return rc;
        }
#endif
        const int rc = launch_conv_wx3(pw, op.cin, op.cout, stream, dry);
        if (rc <= 0) return rc;                                                    // launched (0) or failed (< 0)
    }
    if (moments && !(f16w_direct && op.cout == 32)) {
        set_error("conv: tile moments requested but the Winograd split-bf16 kernel does not take this shape (%dx%d, %d -> %d)",
                  op.H, op.W, op.cin, op.cout);
        return SBC_ERR_UNSUPPORTED;
    }
    if (x3) {
        p.wpk = (const float4*)op.weight_split;
        return launch_conv_x3(p, op.cin, op.cout, op.ksize, stream, dry);
    }
    SBC_REQUIRE(!(op.flags & SBC_PRO_NORM_SELF), "conv: SBC_PRO_NORM_SELF: no matrix-core kernel takes this layer (%dx%d, %d -> %d, dilation %d)",
                op.H, op.W, op.cin, op.cout, op.dil);
    static const bool no_wino = getenv("SBC_NO_WINO") != nullptr;                  // A/B aid
    if (op.weight_wino && op.ksize == 3 && op.dil == 1 && !no_wino && !direct_only) {
        ConvParams pw = p;
        pw.wpk = (const float4*)op.weight_wino;
        const int rc = launch_conv_wino(pw, op.cin, op.cout, stream, dry);
        if (rc <= 0) return rc;                                                    // launched (0) or failed (< 0)
    }
    const int key = op.cin * 100000 + op.cout * 100 + op.ksize;
    switch (key) {
        case 32 * 100000 + 32 * 100 + 3: return launch_sized<32, 32, 3>(p, stream, dry);
        case 32 * 100000 + 64 * 100 + 3: return launch_sized<32, 64, 3>(p, stream, dry);
        case 32 * 100000 + 64 * 100 + 1: return launch_sized<32, 64, 1>(p, stream, dry);
        case 64 * 100000 + 64 * 100 + 3: return launch_sized<64, 64, 3>(p, stream, dry);
        case 64 * 100000 + 64 * 100 + 1: return launch_sized<64, 64, 1>(p, stream, dry);
        case 64 * 100000 + 32 * 100 + 3: return launch_sized<64, 32, 3>(p, stream, dry);
        case 64 * 100000 + 128 * 100 + 3: return launch_sized<64, 128, 3>(p, stream, dry);
        case 128 * 100000 + 128 * 100 + 3: return launch_sized<128, 128, 3>(p, stream, dry);
        case 128 * 100000 + 64 * 100 + 3: return launch_sized<128, 64, 3>(p, stream, dry);
        default:
            set_error("conv: no kernel for cin=%d cout=%d ksize=%d (NCSNv2Deepest with ngf=32 needs "
                      "32/64/128 channels)", op.cin, op.cout, op.ksize);
            return SBC_ERR_UNSUPPORTED;
    }
}

}  // namespace sbc

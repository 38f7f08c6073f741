// Training operators, part 2: weight / bias gradients of the convolutions (the input gradients run as
// ordinary SBC_OP_CONV launches with adjoint-packed weights, SBC_OP_PACK_WEIGHT | SBC_PACK_ADJOINT) and the reverse of the
// two 2-channel convolutions at the ends of the network.
//
//   dW[co][ci][tap] = sum_{b,h,w} a[b][h + dh(tap)][w + dw(tap)][ci] * dC[b][h][w][co],   a = pro(x) (IN++ affine, ELU),
//   db[co] = sum dC[b][h][w][co]
// is a [k*k*cin] x [cout] product with B*H*W as the contraction.  conv_wgrad_kernel: a workgroup owns one tap and a chunk
// of the 64-pixel tiles; per tile it stages a (with halo, through the forward prologue) and dC in LDS and its four waves
// accumulate 32 x 32 blocks of that tap's outputs on the fp32 matrix cores across the tiles.
#include "tile.h"

namespace sbc {

constexpr int WG_TM = 64;             // pixels per tile (divides or is a multiple of every image of the network)
constexpr int WG_MAX_CHUNKS = 64;

static int wgrad_chunks(long total_px) {
    const long tiles = (total_px + WG_TM - 1) / WG_TM;
    return (int)(tiles < WG_MAX_CHUNKS ? tiles : WG_MAX_CHUNKS);
}

typedef float f32x16 __attribute__((ext_vector_type(16)));

// Wave layout: WM x WN waves tile the [CIN] x [COUT] output blocks of 32 x 32, WK waves split the pixels of a tile (the
// contraction); WM * WN * WK = 4.  Each wave holds (CIN/32/WM) x (COUT/32/WN) accumulator blocks (v_mfma_f32_32x32x2_f32:
// exact fp32 products, fp32 accumulation; A = a[pixel][ci] with M = ci, B = dC[pixel][co] with N = co, K = 2 pixels per
// instruction, both operands one conflict-free ds_read_b32 per lane).  Partials go to scratch
// [chunk * WK + wk][tap][ci][co]; wgrad_reduce_kernel sums them in ascending order and writes the torch layout.  (A
// single-launch variant in which the last workgroup of a tap reduces it was 3x SLOWER: the device-scope fence it needs
// writes back the whole L2 of the XCD, once per workgroup.)
template <int CIN, int COUT, int KS, int WM, int WN, int WK>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const float* __restrict__ in, const float* __restrict__ stats,
                                                          const float* __restrict__ dC, float* __restrict__ scratch,
                                                          int B, int H, int W, int dil, int flags, int nchunks) {
    static_assert(WM * WN * WK == 4, "four waves");
    constexpr int S = CIN + 4, DS = COUT + 4;
    constexpr int MB = CIN / 32 / WM, NB = COUT / 32 / WN;       // accumulator blocks per wave
    static_assert(MB >= 1 && NB >= 1, "wgrad: wave tiling");
    constexpr int TAPS = KS * KS;
    constexpr int PXW = WG_TM / WK;                              // pixels of a tile per K-split wave
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wk = wave % WK, wn = (wave / WK) % WN, wm = wave / (WK * WN);
    const int tap = blockIdx.y, chunk = blockIdx.x;
    const Dims<false> d{H, W, H * W, 0, 0};
    const int total_px = B * H * W, ntiles = (total_px + WG_TM - 1) / WG_TM;
    const int halo_rows = KS == 3 ? dil : 0;
    const int max_nps = WG_TM >= d.HW ? WG_TM : WG_TM + 2 * halo_rows * W;
    float* at = lds;                                   // [max_nps + 1][S]
    float* dt = at + (size_t)(max_nps + 1) * S;        // [WG_TM][DS]
    int* aidx = reinterpret_cast<int*>(dt + (size_t)WG_TM * DS);   // [WG_TM] staged-pixel index of this tap, per pixel
    const int dh = KS == 3 ? (tap / 3 - 1) * dil : 0, dw = KS == 3 ? (tap % 3 - 1) * dil : 0;
    f32x16 acc[MB][NB];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float bacc = 0.f;
    const int half = lane >> 5, col = lane & 31;
    for (int tile = chunk; tile < ntiles; tile += nchunks) {
        const TileGeom g = tile_geom(tile, WG_TM, B, d, halo_rows);
        __syncthreads();                                // previous tile fully consumed
        stage_tile<CIN, 256, 4, false>(at, in, stats, flags & (SBC_PRO_NORM | SBC_PRO_ELU), g, d, tid);
        for (int i = tid; i < WG_TM * (COUT / 4); i += 256) {
            const int pl = i / (COUT / 4), c = i % (COUT / 4);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (g.p0 + pl < total_px) v = *reinterpret_cast<const float4*>(dC + (size_t)(g.p0 + pl) * COUT + c * 4);
            *reinterpret_cast<float4*>(dt + pl * DS + c * 4) = v;
        }
        if (tid < WG_TM) {
            const int px = g.p0 + tid;
            int a = g.nps;                               // the zero pixel
            if (px < total_px) {
                const int row = px / W, w = px - row * W, h = row % H;
                const int hh = h + dh, ww = w + dw;
                if (hh >= 0 && hh < H && ww >= 0 && ww < W) a = (row + dh - g.rs0) * W + ww;
            }
            aidx[tid] = a;
        }
        __syncthreads();
#pragma unroll 2
        for (int p2 = 0; p2 < PXW; p2 += 2) {
            const int pl = wk * PXW + p2 + half;
            const float* ap = at + aidx[pl] * S + wm * MB * 32 + col;
            const float* bp = dt + pl * DS + wn * NB * 32 + col;
            float av[MB], bv[NB];
#pragma unroll
            for (int i = 0; i < MB; ++i) av[i] = ap[i * 32];
#pragma unroll
            for (int j = 0; j < NB; ++j) bv[j] = bp[j * 32];
#pragma unroll
            for (int i = 0; i < MB; ++i)
#pragma unroll
                for (int j = 0; j < NB; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
        if (tap == 0 && tid < COUT) {
            for (int pl = 0; pl < WG_TM; ++pl) bacc += dt[pl * DS + tid];
        }
    }
    // scratch layout: 16 unused floats | partials [nchunks * WK][TAPS][CIN][COUT] | bias [nchunks][COUT]
    float* part = scratch + 16;
    const int per_tap = CIN * COUT, nparts = nchunks * WK;
    float* o = part + (((size_t)(chunk * WK + wk) * TAPS + tap) * CIN) * COUT;
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ci = (wm * MB + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                o[(size_t)ci * COUT + (wn * NB + j) * 32 + col] = acc[i][j][r];
            }
    float* bpart = part + (size_t)nparts * TAPS * per_tap;
    if (tap == 0 && tid < COUT) bpart[(size_t)chunk * COUT + tid] = bacc;
}

// sum the partials in ascending order (fixed order, no atomics); write the torch layout [co][ci][tap]
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dW,
                                                            float* __restrict__ db, int nparts, int nchunks, int taps, int cin,
                                                            int cout) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int per = taps * cin * cout;
    if (i < per) {
        float s = 0.f;
#pragma unroll 8
        for (int c = 0; c < nparts; ++c) s += part[(size_t)c * per + i];
        const int co = i % cout, ci = (i / cout) % cin, tap = i / (cout * cin);
        dW[((size_t)co * cin + ci) * taps + tap] = s;
    } else if (db && i < per + cout) {
        const int co = i - per;
        float s = 0.f;
        for (int c = 0; c < nchunks; ++c) s += part[(size_t)nparts * per + (size_t)c * cout + co];
        db[co] = s;
    }
}

template <int CIN, int COUT>
struct WgradWaves {        // (WM, WN, WK): blocks of 32 x 32 per wave <= 2 x 2, the rest of the four waves split the pixels
    static constexpr int WM = CIN >= 64 ? 2 : 1;
    static constexpr int WN = COUT >= 64 ? 2 : 1;
    static constexpr int WK = 4 / (WM * WN);
};

template <int CIN, int COUT, int KS>
static int launch_wgrad_t(const sbc_op& op, hipStream_t stream) {
    using Wv = WgradWaves<CIN, COUT>;
    const long total_px = (long)op.B * op.H * op.W;
    const int HW = op.H * op.W;
    SBC_REQUIRE(WG_TM % op.W == 0 && (HW % WG_TM == 0 || WG_TM % HW == 0), "conv_wgrad: image %dx%d does not tile by %d pixels",
                op.H, op.W, WG_TM);
    const int nchunks = wgrad_chunks(total_px);
    const int halo_rows = KS == 3 ? op.dil : 0;
    const int max_nps = WG_TM >= HW ? WG_TM : WG_TM + 2 * halo_rows * op.W;
    const size_t lds = ((size_t)(max_nps + 1) * (CIN + 4) + (size_t)WG_TM * (COUT + 4)) * sizeof(float) + WG_TM * sizeof(int);
    SBC_REQUIRE(lds <= 160 * 1024, "conv_wgrad: tile needs %zu bytes of LDS", lds);
    auto kern = conv_wgrad_kernel<CIN, COUT, KS, Wv::WM, Wv::WN, Wv::WK>;
    { const int rc = ensure_dyn_lds(reinterpret_cast<const void*>(kern), lds); if (rc) return rc; }
    hipLaunchKernelGGL(kern, dim3(nchunks, KS * KS), dim3(256), lds, stream, (const float*)op.in, (const float*)op.stats,
                       (const float*)op.grad, (float*)op.aux, op.B, op.H, op.W, op.dil, op.flags, nchunks);
    SBC_CHECK_HIP(hipGetLastError());
    const int outs = KS * KS * CIN * COUT + COUT;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((outs + 255) / 256), dim3(256), 0, stream, (const float*)op.aux + 16,
                       (float*)op.wgrad, (float*)op.bgrad, nchunks * Wv::WK, nchunks, KS * KS, CIN, COUT);
    SBC_CHECK_HIP(hipGetLastError());
    return SBC_OK;
}

int launch_conv_wgrad(const sbc_op& op, hipStream_t stream) {
    SBC_REQUIRE(op.in && op.grad && op.wgrad && op.aux, "conv_wgrad: in/grad/wgrad/aux must be set");
    SBC_REQUIRE(!(op.flags & SBC_PRO_NORM) || op.stats, "conv_wgrad: PRO_NORM needs stats");
    SBC_REQUIRE(op.B > 0 && op.H > 0 && op.W > 0 && (long)op.B * op.H * op.W * (op.cin > op.cout ? op.cin : op.cout) <= 0x7fffffffL,
                "conv_wgrad: bad or too large shape");
    const int key = op.cin * 100000 + op.cout * 100 + op.ksize;
    switch (key) {
        case 32 * 100000 + 32 * 100 + 3: return launch_wgrad_t<32, 32, 3>(op, stream);
        case 32 * 100000 + 64 * 100 + 3: return launch_wgrad_t<32, 64, 3>(op, stream);
        case 32 * 100000 + 64 * 100 + 1: return launch_wgrad_t<32, 64, 1>(op, stream);
        case 64 * 100000 + 64 * 100 + 3: return launch_wgrad_t<64, 64, 3>(op, stream);
        case 64 * 100000 + 64 * 100 + 1: return launch_wgrad_t<64, 64, 1>(op, stream);
        case 64 * 100000 + 32 * 100 + 3: return launch_wgrad_t<64, 32, 3>(op, stream);
        case 64 * 100000 + 128 * 100 + 3: return launch_wgrad_t<64, 128, 3>(op, stream);
        case 128 * 100000 + 128 * 100 + 3: return launch_wgrad_t<128, 128, 3>(op, stream);
        case 128 * 100000 + 64 * 100 + 3: return launch_wgrad_t<128, 64, 3>(op, stream);
        default:
            set_error("conv_wgrad: no kernel for cin=%d cout=%d ksize=%d", op.cin, op.cout, op.ksize);
            return SBC_ERR_UNSUPPORTED;
    }
}

// ------------------------------------------------------------------------------------------------ 2-channel ends
// begin_conv (2 -> 32, input h = 2x - 1 zero-padded) and end_conv (ELU(IN++(x)) 32 -> 2, output / sigma): one side of the
// product has two channels, so the 576 weight gradients are one thread each: a workgroup owns 256 pixels of whole rows,
// stages a (with a one-pixel border, zeros outside the image) and d in LDS, every thread sums its outputs over the pixels;
// partials per workgroup, summed in order by wgrad_reduce_small_kernel.
//   MODE 0 (begin): a = 2x - 1 [CA = 2], d = grad [CD = 32];   dW[co][ci][tap], co < 32, ci < 2
//   MODE 1 (end):   a = ELU((x - mu) scale + shift) [CA = 32], d = grad / sigma_b [CD = 2];  dW[co][ci][tap], co < 2, ci < 32
template <int MODE>
__global__ __launch_bounds__(256) void wgrad_small_kernel(const float* __restrict__ x, const float* __restrict__ stats,
                                                           const float* __restrict__ grad, float* __restrict__ scratch,
                                                           sbc_endconv e, int B, int H, int W, int rows) {
    constexpr int CA = MODE == 0 ? 2 : 32, CD = MODE == 0 ? 32 : 2;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int wgs_per_sample = H / rows;
    const int n = blockIdx.x / wgs_per_sample, r0 = (blockIdx.x % wgs_per_sample) * rows;
    const int WP = W + 2, npx = rows * W;
    float* at = lds;                                    // [(rows + 2)][WP][CA]
    float* dt = at + (size_t)(rows + 2) * WP * CA;      // [rows][W][CD]
    for (int i = tid; i < (rows + 2) * WP * CA; i += 256) {
        const int c = i % CA, cc = (i / CA) % WP, rr = i / (CA * WP);
        const int r = r0 - 1 + rr, col = cc - 1;
        float v = 0.f;
        if (r >= 0 && r < H && col >= 0 && col < W) {
            v = x[(((size_t)n * H + r) * W + col) * CA + c];
            if (MODE == 0) v = 2.f * v - 1.f;
            else {
                const float* st = stats + (size_t)n * 3 * CA;
                v = elu1((v - st[c]) * st[CA + c] + st[2 * CA + c]);
            }
        }
        at[i] = v;
    }
    float inv_sigma = 1.f;
    if (MODE == 1) inv_sigma = 1.f / (e.labels ? e.sigmas[e.labels[n]] : e.sigma_of_step[*e.step]);
#pragma clang loop vectorize(disable)                    // no packed-fp32 instructions in this library (Makefile)
    for (int i = tid; i < npx * CD; i += 256) {
        float v = grad[((size_t)n * H + r0) * W * CD + i];
        if (MODE == 1) v = v * inv_sigma;                // d (out / sigma) / d out  (reference: tensor / sigma)
        dt[i] = v;
    }
    __syncthreads();
    constexpr int NOUT = 9 * CA * CD;                    // 576
    float* o = scratch + 16 + (size_t)blockIdx.x * (NOUT + CD);     // the first 16 words belong to conv_wgrad_kernel's counters
    for (int q = tid; q < NOUT + CD; q += 256) {
        float s = 0.f;
        if (q < NOUT) {
            const int tap = q % 9, ci = (q / 9) % CA, co = q / (9 * CA);      // torch order [co][ci][tap]
            const int kh = tap / 3, kw = tap % 3;
            for (int pl = 0; pl < npx; ++pl) {
                const int rr = pl / W, cc = pl - rr * W;
                s = fmaf(at[((rr + kh) * WP + cc + kw) * CA + ci], dt[pl * CD + co], s);
            }
        } else {
            const int co = q - NOUT;
            for (int pl = 0; pl < npx; ++pl) s += dt[pl * CD + co];
        }
        o[q] = s;
    }
}

__global__ __launch_bounds__(256) void wgrad_reduce_small_kernel(const float* __restrict__ scratch, float* __restrict__ dW,
                                                                  float* __restrict__ db, int nwg, int nout, int cd) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= nout + cd) return;
    float s = 0.f;
    for (int g = 0; g < nwg; ++g) s += scratch[16 + (size_t)g * (nout + cd) + q];
    if (q < nout) dW[q] = s;
    else if (db) db[q - nout] = s;
}

static int small_rows(int H, int W) {
    int rows = 256 / W > 0 ? 256 / W : 1;
    if (rows > H) rows = H;
    while (H % rows) --rows;
    return rows;
}

template <int MODE>
static int launch_wgrad_small(const sbc_op& op, const sbc_endconv& e, hipStream_t stream) {
    constexpr int CA = MODE == 0 ? 2 : 32, CD = MODE == 0 ? 32 : 2;
    const int rows = small_rows(op.H, op.W);
    const size_t lds = ((size_t)(rows + 2) * (op.W + 2) * CA + (size_t)rows * op.W * CD) * sizeof(float);
    SBC_REQUIRE(lds <= 160 * 1024, "begin/end conv backward: image row of %d pixels too wide", op.W);
    auto kern = wgrad_small_kernel<MODE>;
    { const int rc = ensure_dyn_lds(reinterpret_cast<const void*>(kern), lds); if (rc) return rc; }
    const int nwg = op.B * (op.H / rows);
    hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), lds, stream, (const float*)op.in, (const float*)op.stats,
                       (const float*)op.grad, (float*)op.aux, e, op.B, op.H, op.W, rows);
    SBC_CHECK_HIP(hipGetLastError());
    hipLaunchKernelGGL(wgrad_reduce_small_kernel, dim3((9 * CA * CD + CD + 255) / 256), dim3(256), 0, stream,
                       (const float*)op.aux, (float*)op.wgrad, (float*)op.bgrad, nwg, 9 * CA * CD, CD);
    SBC_CHECK_HIP(hipGetLastError());
    return SBC_OK;
}

int launch_begin_conv_bwd(const sbc_op& op, hipStream_t stream) {
    SBC_REQUIRE(op.in && op.grad && op.wgrad && op.aux, "begin_conv_bwd: in/grad/wgrad/aux must be set");
    SBC_REQUIRE(op.cin == 2 && op.cout == 32, "begin_conv_bwd: cin=%d cout=%d (built for 2 -> 32)", op.cin, op.cout);
    sbc_endconv none{};
    return launch_wgrad_small<0>(op, none, stream);
}

// d / d ELU-output of the end conv: dA[px][ci] = sum_{tap, co} w[co][ci][tap] * dO[px - off(tap)][co], dO = grad / sigma_b
__global__ __launch_bounds__(256) void end_conv_dgrad_kernel(const float* __restrict__ grad, const float* __restrict__ w,
                                                              float* __restrict__ out, sbc_endconv e, int B, int H, int W,
                                                              int C4) {
    __shared__ float wl[2 * 32 * 9];
    const int CIN = C4 * 4;
    for (int i = threadIdx.x; i < 2 * CIN * 9; i += 256) wl[i] = w[i];
    __syncthreads();
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)B * H * W * C4) return;
    const int c4 = (int)(i % C4), wq = (int)((i / C4) % W), h = (int)((i / ((long)C4 * W)) % H), n = (int)(i / ((long)C4 * W * H));
    const float inv_sigma = 1.f / (e.labels ? e.sigmas[e.labels[n]] : e.sigma_of_step[*e.step]);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int tap = 0; tap < 9; ++tap) {
        const int hh = h - (tap / 3 - 1), ww = wq - (tap % 3 - 1);          // output pixel that read this input through `tap`
        if (hh < 0 || hh >= H || ww < 0 || ww >= W) continue;
        float2 d = *reinterpret_cast<const float2*>(grad + (((size_t)n * H + hh) * W + ww) * 2);
        d.x *= inv_sigma; d.y *= inv_sigma;
        const float* w0 = wl + (c4 * 4) * 9 + tap;                          // [co = 0][ci][tap]
        const float* w1 = wl + (CIN + c4 * 4) * 9 + tap;                    // [co = 1][ci][tap]
        acc.x = fmaf(w0[0], d.x, fmaf(w1[0], d.y, acc.x));
        acc.y = fmaf(w0[9], d.x, fmaf(w1[9], d.y, acc.y));
        acc.z = fmaf(w0[18], d.x, fmaf(w1[18], d.y, acc.z));
        acc.w = fmaf(w0[27], d.x, fmaf(w1[27], d.y, acc.w));
    }
    *reinterpret_cast<float4*>(out + i * 4) = acc;
}

int launch_end_conv_bwd(const sbc_op& op, const sbc_endconv& e, hipStream_t stream) {
    SBC_REQUIRE(op.in && op.stats && op.weight && op.grad && op.out && op.wgrad && op.aux,
                "end_conv_bwd: in/stats/weight/grad/out/wgrad/aux must be set");
    SBC_REQUIRE(op.cin == 32 && op.cout == 2, "end_conv_bwd: cin=%d cout=%d (built for 32 -> 2)", op.cin, op.cout);
    SBC_REQUIRE((e.labels && e.sigmas) || (e.sigma_of_step && e.step), "end_conv_bwd: no noise-level source");
    const int rc = launch_wgrad_small<1>(op, e, stream);
    if (rc) return rc;
    const long n = (long)op.B * op.H * op.W * (op.cin / 4);
    hipLaunchKernelGGL(end_conv_dgrad_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, (const float*)op.grad,
                       (const float*)op.weight, (float*)op.out, e, op.B, op.H, op.W, op.cin / 4);
    SBC_CHECK_HIP(hipGetLastError());
    return SBC_OK;
}

}  // namespace sbc

extern "C" int64_t sbc_wgrad_scratch_floats(int32_t B, int32_t H, int32_t W, int32_t cin, int32_t cout, int32_t ksize) {
    if (B <= 0 || H <= 0 || W <= 0 || cin <= 0 || cout <= 0 || ksize <= 0) return 0;
    if (cin == 2 || cout == 2) {                                             // begin / end conv: one partial per workgroup
        const int rows = sbc::small_rows(H, W);
        return 16 + (int64_t)B * (H / rows) * (9 * cin * cout + (cin == 2 ? cout : 2));
    }
    const int chunks = sbc::wgrad_chunks((long)B * H * W);
    return 16 + (int64_t)chunks * 4 * ((int64_t)ksize * ksize * cin * cout) + (int64_t)chunks * cout;   // <= 4 pixel splits
}

// ONE 3x3 32 -> 32 convolution of the full-resolution level (16-pixel rows) as a direct, pipelined, persistent kernel -- the two
// layers of a 64 x 16 array that are neither part of an RCU pair, a CRP stage nor a whole ResidualBlock:
//   res2.0.conv1         out = conv(ELU(norm1(x))) + bias, + the (mean, M2) of the output's 128-pixel tiles for normalize2
//                        (ncsnv2/models/layers.py:443-447; InstanceNorm2dPlus: normalization.py:150-176)
//   refine5.msf.convs.0  out = conv(x) + bias + bilinear resize of the low-resolution branch (layers.py:178-184)
// Both ran on the Winograd kernel (conv_wx3.hip: 35 vector instructions per matrix instruction, 150-156 us a launch).  This is
// conv_pair_roll_kernel (conv_pair.hip) with ONE convolution: a persistent workgroup per CU of twelve waves walks a contiguous run
// of 8-row tiles --
//   conversion (4 waves): keep the next item's raw rows in flight by LDS-DMA; raw fp32 -> [norm] -> [ELU] -> x act_scale -> two fp16
//           terms -> a ring of operand rows (24 rows in three slots of 8 + a copy of rows 22, 23 in front of row 0): a TILE item
//           converts the eight rows r0+1 .. r0+8, a PRE item (the run's first tile, the first tile of every sample) the two rows
//           r0-1, r0 -- every input row is fetched and converted once;
//   matrix (8 waves, two per SIMD): K loop of the convolution on the rows slot-2 .. slot+7 of the previous item, two units (image rows
//           sub, sub + 4) of 16 output channels per wave, filter fragments resident in registers (72); x descale + bias [+ resize];
//           store; [tile moments: the wave's 32 pixels per channel in registers and by DPP row sums, the four waves of a channel
//           block meet in LDS, merged one item later in a fixed order].
// One workgroup barrier per item.  Same products, same order of the K loop as conv_dp / conv_pair: direct f16x2 arithmetic.
//
// NOT PART OF THE LIBRARY (round 5, measured and left out).  Correct -- the test that went with it (six cases: plain, norm + ELU + bias
// + tile moments, bias + resize; 8 / 4 / 2 / 1 tiles per sample) passed against the oracle, numpy's tile moments and the Winograd
// kernel -- but no faster than the Winograd kernel it would replace: 159 / 161 us against 153 / 152 us per launch at 1700 x 64 x 16
// (81 / 76 against 69 / 75 at 850), the two-stream step 4.32-4.40 against 4.31-4.34 ms.  Three LDS-DMA items in flight instead of one
// changed nothing (164 us), nor did term-major matrix instructions in the pair kernel's K loop: these single-convolution pipelines
// sit at ~3 TB/s and half of their instruction-issue bound whatever the arithmetic in them (conv_pool_kernel without its pooling: 132
// against 143 us), so the 2.25 x matrix work of a direct convolution over F(2x2, 3x3) buys nothing here.
// To build it in: add conv_row.hip to csrc/Makefile's SRCS with -DSBC_WITH_CONV_ROW (conv_mfma.hip: launch_conv routes to it).
#include <stdlib.h>
#include <type_traits>
#include "../../score_based_channels_amd/csrc/conv_common.h"

namespace sbc {

typedef float f32x4v __attribute__((ext_vector_type(4)));

struct RowParams {
    const float* __restrict__ in;
    float* __restrict__ out;
    const uint4* __restrict__ w;         // sbc_pack_conv_weight_f16x2 layout (32 -> 32, 3x3)
    const float* __restrict__ bias;      // [32] or NULL
    const float* __restrict__ stats;     // SBC_PRO_NORM: [B][3][32] (mu, scale, shift) of the input's InstanceNorm++ (SBC_OP_INORM_STATS)
    const float* __restrict__ up;        // SBC_EPI_UP: [B][up_h][up_w][32]
    float* __restrict__ pm_out;          // SBC_EPI_MOMENTS_OUT: [B][H / 8][32][2] (mean, M2) of the output's 128-pixel tiles
    unsigned* __restrict__ range_flag;
    float* __restrict__ calib;           // sbc_f16x2_calibrate: amax slot of the (normalised, activated) input, else NULL
    int flags, up_h, up_w;
    int B, H, ntiles, tiles_per_sample, wgs_per_xcd, tiles_per_xcd;
};

// sum over the 16 lanes of a DPP row (= the 16 pixels of a unit for one k-quarter); every lane of the row gets the total
__device__ __forceinline__ float row_total16(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128 /* row_ror:8 */, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124 /* row_ror:4 */, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122 /* row_ror:2 */, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121 /* row_ror:1 */, 0xf, 0xf, false));
    return v;
}

__global__ __launch_bounds__(768) void conv_row_kernel(RowParams p) {
    constexpr int W = 16, R = 8, C = 32, KGS = C / 8, C4 = C / 4, NT = 2;
    constexpr int NWM = 8, NWC = 4, NTC = 64 * NWC;    // matrix waves, conversion waves (and their threads)
    constexpr int WP = W + 2, ROWB = WP * 16;          // bytes of one plane row
    constexpr int RING = 24, XMIR = 2;                 // ring rows; rows copied in front of row 0
    constexpr int XPS = ((RING + XMIR) * ROWB + 255) / 256 * 256;
    constexpr int RAW_BYTES = R * W * C * 4;
    constexpr int DEPTH = 3, NRAW = DEPTH + 1;        // LDS-DMA items in flight; raw buffers
    constexpr int X_OFF = NRAW * RAW_BYTES, RED_OFF = X_OFF + NT * KGS * XPS;   // red: [2 items][32 channels][4 unit groups][(mean, M2)]
    constexpr int NK = R * W * C4 / NTC;               // 16-byte chunks per conversion thread and TILE item (4); a PRE item: the last one
    constexpr int NU = 2;                              // units (image rows) per matrix wave
    extern __shared__ __attribute__((aligned(256))) unsigned char smem[];

    const int wv = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);        // 0 .. 7: matrix waves, 8 .. 11: conversion waves
    const bool matrix = wv < NWM;
    const int lane = threadIdx.x & 63;
    const int hf = wv & 1, sub = (wv >> 1) & 3;        // matrix: 16-output-channel half; unit group (rows sub, sub + 4 of the tile)
    const int kq = lane >> 4, c = lane & 15;
    const int H = p.H;

    const float4 tr = f16x2_trailer(reinterpret_cast<const float4*>(p.w), 9 * (C / 16) * (C / 32) * NT);
    const float scale = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, tr.x)));
    const float descale = tr.y;
    const bool elu_acc = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, tr.w)) != 0;   // small inputs: ELU in its accurate form (common.h)
    unsigned rbits = 0;

    // ---- zero the padding columns of every plane row once
    for (int i = threadIdx.x; i < NT * KGS * (RING + XMIR) * 2; i += 64 * (NWM + NWC)) {
        const int side = i & 1, row = (i >> 1) % (RING + XMIR), pl = (i >> 1) / (RING + XMIR);
        *reinterpret_cast<uint4*>(smem + X_OFF + pl * XPS + row * ROWB + side * (W + 1) * 16) = make_uint4(0, 0, 0, 0);
    }

    // ---- this workgroup's run (conv_pair_roll_kernel): XCD x owns tiles [t_begin, t_end), its workgroups equal contiguous pieces [a, b)
    const int xcd = blockIdx.x & 7, jw = blockIdx.x >> 3;
    const int t_begin = xcd * p.tiles_per_xcd;
    const int cnt = max(min(t_begin + p.tiles_per_xcd, p.ntiles) - t_begin, 0);
    const int a = t_begin + jw * cnt / p.wgs_per_xcd, b = t_begin + (jw + 1) * cnt / p.wgs_per_xcd;
    if (a >= b) return;                                               // (whole workgroup)
    const int tps = p.tiles_per_sample;
    const int n_items = (b - a) + 1 + ((b - 1) / tps - a / tps);    // tiles + the run's PRE + one PRE per sample that starts inside
    struct Cur { int n, j, pre; };                                    // the next item: PRE (pre != 0) or TILE of tile j of sample n
    auto cur_first = [&]() { Cur q; q.n = a / tps; q.j = a - q.n * tps; q.pre = 1; return q; };
    auto cur_next = [&](Cur& q) {
        if (q.pre) { q.pre = 0; return; }
        if (++q.j == tps) { q.j = 0; ++q.n; q.pre = 1; }
    };
    auto next_slot = [](int s) { return s == 16 ? 0 : s + 8; };

    if (!matrix) {
        // ================================================================= conversion: LDS-DMA one item ahead, raw -> ring slot
        const int wc = wv - NWM, tid = threadIdx.x - 64 * NWM;
        const int c4 = tid % C4;                                      // this thread's channel quad (the same for all its chunks)
        auto issue_dma = [&](const Cur& q, int buf) {
            const int rb = R * q.j - (q.pre ? R : 0) + 1;            // image row of the slot's row 0
#pragma unroll
            for (int k = 0; k < NK; ++k) {
                if (k < NK - 1 && q.pre) continue;                   // (uniform)
                const int j = k * NWC + wc;                           // piece: 64 chunks = half a row
                const int ri = j >> 1;
                const int grow = min(max(rb + ri, 0), H - 1);         // rows outside the image: any row inside (converted to zeros)
                const char* sbase = reinterpret_cast<const char*>(p.in) + ((size_t)(q.n * H + grow) * W * C) * 4 + (size_t)(j & 1) * 1024;
                const unsigned dst = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem + buf * RAW_BYTES + j * 1024;
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(lane * 16), "s"(dst), "s"(sbase) : "memory");
            }
        };
        const bool pro_norm = (p.flags & SBC_PRO_NORM) != 0, pro_elu = (p.flags & SBC_PRO_ELU) != 0;
        float4 mu = make_float4(0.f, 0.f, 0.f, 0.f), sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = mu;
        int stats_n = -1;
        auto convert = [&](const Cur& q, int buf, int slot) {
            const int rb = R * q.j - (q.pre ? R : 0) + 1;
            if (pro_norm && q.n != stats_n) {                          // (uniform) a new sample: its statistics for this thread's channels
                const float* st = p.stats + (size_t)q.n * 3 * C + c4 * 4;
                mu = *reinterpret_cast<const float4*>(st); sc = *reinterpret_cast<const float4*>(st + C); sh = *reinterpret_cast<const float4*>(st + 2 * C);
                stats_n = q.n;
            }
            float ta = 0.f;
#pragma unroll
            for (int kk = 0; kk < NK; ++kk) {
                if (kk < NK - 1 && q.pre) continue;
                const int qi = kk * NTC + tid;
                const int px = qi / C4;
                const int ri = px / W, col = px - ri * W;
                const int grow = rb + ri;
                float4 v = *reinterpret_cast<const float4*>(smem + buf * RAW_BYTES + qi * 16);
                if (pro_norm) {
                    v.x = (v.x - mu.x) * sc.x + sh.x; v.y = (v.y - mu.y) * sc.y + sh.y;
                    v.z = (v.z - mu.z) * sc.z + sh.z; v.w = (v.w - mu.w) * sc.w + sh.w;
                }
                if (pro_elu) v = elu4(v, elu_acc);
                if (grow < 0 || grow >= H) v = make_float4(0.f, 0.f, 0.f, 0.f);    // the convolution's zero padding
                unsigned char* dst = smem + X_OFF + (c4 >> 1) * XPS + ((slot + ri + XMIR) * WP + col + 1) * 16 + (c4 & 1) * 8;
                StageScale ss{scale, ta};
                scale_track(v, &ss);
                ta = ss.amax;
                uint2 h, l;
                split_f16x2(v, scale, h, l);
                *reinterpret_cast<uint2*>(dst) = h;
                *reinterpret_cast<uint2*>(dst + KGS * XPS) = l;
                if (kk == NK - 1 && slot == 16) {                     // rows 22, 23: the copy in front of row 0
                    *reinterpret_cast<uint2*>(dst - RING * ROWB) = h;
                    *reinterpret_cast<uint2*>(dst - RING * ROWB + KGS * XPS) = l;
                }
            }
            pair_range_tile(ta, scale, rbits, p.calib);
        };
        // DMA runs DEPTH items ahead (a single convolution leaves ~1.7 us between items: one 16 KB item in flight per CU is 4 MB on the
        // chip, less than the memory system's latency x bandwidth -- the pair kernel's tiles take twice as long and need no more).
        // The wave's requests complete in order: "item it + 1 has landed" = at most the requests of the items behind it outstanding.
        auto wait_outstanding = [](int n) {                          // (n: sums of 1 (PRE) and 4 (TILE) request counts of two items)
            if (n >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if (n >= 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
            else if (n >= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else if (n >= 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else if (n >= 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        };
        Cur qd = cur_first(), qc = qd;
        int c1 = 0, c2 = 0;                                            // requests of items it + 1, it + 2 (0: no such item)
        static_assert(DEPTH == 3, "the bookkeeping below names three items in flight");
        issue_dma(qd, 0);
        cur_next(qd);
        if (1 < n_items) { issue_dma(qd, 1); c1 = qd.pre ? 1 : NK; cur_next(qd); }
        if (2 < n_items) { issue_dma(qd, 2); c2 = qd.pre ? 1 : NK; cur_next(qd); }
        wait_outstanding(c1 + c2);                                     // item 0 has landed (the opening barrier publishes it)
        int slot = 0;
        for (int it = 0; it < n_items + 2; ++it) {
            lds_barrier();
            int c3 = 0;
            if (it + DEPTH < n_items) { issue_dma(qd, (it + DEPTH) % NRAW); c3 = qd.pre ? 1 : NK; cur_next(qd); }
            if (it < n_items) { convert(qc, it % NRAW, slot); cur_next(qc); slot = next_slot(slot); }
            wait_outstanding(c2 + c3);                                 // item it + 1 has landed before the next barrier publishes it
            c1 = c2; c2 = c3;
        }
        (void)c1;
    } else {
        // ================================================================= matrix: ring window -> K loop -> epilogue
        uint4 wf[9][NT];
        {
            const int lsrc = (16 * hf + c) + 32 * (kq & 1);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap)
#pragma unroll
                for (int t = 0; t < NT; ++t) wf[tap][t] = p.w[((tap * (C / 16) + (kq >> 1)) * NT + t) * 64 + lsrc];
        }
        const int cq = 4 * hf + kq;                                    // channel quad of this lane's four outputs
        float4 bias = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.bias) bias = *reinterpret_cast<const float4*>(p.bias + cq * 4);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        float* const red = reinterpret_cast<float*>(smem + RED_OFF);
        const bool epi_up = (p.flags & SBC_EPI_UP) != 0;
        const float fsh = epi_up && H > 1 ? (float)(p.up_h - 1) / (float)(H - 1) : 0.f;
        const float fsw = epi_up ? (float)(p.up_w - 1) / (float)(W - 1) : 0.f;
        Cur q = cur_first();
        int slot = 0, prev_tile = -1;                                  // prev_tile: the TILE item whose moment partials wait in `red`
        for (int it = 0; it < n_items + 2; ++it) {
            lds_barrier();
            // tile moments of the item before: wave 0, one channel per lane -- the four unit groups' partials (32 pixels each) in order:
            // mean of the means, M2 = sum M2_g + 32 sum (mean_g - mean)^2 (equal counts; tile.h)
            if (p.pm_out && prev_tile >= 0 && wv == 0 && lane < C) {
                const float* r = red + ((it & 1) * C + lane) * 8;      // (written in iteration it - 1: parity (it - 1 - 1) & 1 == it & 1)
                const float4 m01 = *reinterpret_cast<const float4*>(r), m23 = *reinterpret_cast<const float4*>(r + 4);
                const float m = ((m01.x + m01.z) + (m23.x + m23.z)) * 0.25f;
                float dd = 0.f, d;
                d = m01.x - m; dd = fmaf(d, d, dd); d = m01.z - m; dd = fmaf(d, d, dd);
                d = m23.x - m; dd = fmaf(d, d, dd); d = m23.z - m; dd = fmaf(d, d, dd);
                const float qq = ((m01.y + m01.w) + (m23.y + m23.w));
                *reinterpret_cast<float2*>(p.pm_out + ((size_t)prev_tile * C + lane) * 2) = make_float2(m, fmaf(32.f, dd, qq));
            }
            prev_tile = -1;
            if (it < 1 || it - 1 >= n_items) continue;
            if (!q.pre) {
                constexpr int DO = 4 * W * C, DU = 4 * ROWB;
                const int r0 = R * q.j;
                const unsigned o0 = (unsigned)(((q.n * H + r0 + sub) * W + c) * C + cq * 4);
                f32x4v acc[NU];
                // the resized operand: its four neighbours per output pixel, requested before the K loop (bilinear, align_corners = True:
                // layers.py:182-183)
                float4 u00[NU], u01[NU], u10[NU], u11[NU];
                float lh1[NU], lw1 = 0.f;
                if (epi_up) {
                    const float fw = fsw * (float)c;
                    const int w0 = min((int)fw, p.up_w - 1), w1 = min(w0 + 1, p.up_w - 1);
                    lw1 = fw - (float)w0;
                    const float* u = p.up + (size_t)q.n * p.up_h * p.up_w * C + cq * 4;
#pragma unroll
                    for (int i = 0; i < NU; ++i) {
                        const float fh = fsh * (float)(r0 + sub + 4 * i);
                        const int h0 = min((int)fh, p.up_h - 1), h1 = min(h0 + 1, p.up_h - 1);
                        lh1[i] = fh - (float)h0;
                        u00[i] = *reinterpret_cast<const float4*>(u + (size_t)(h0 * p.up_w + w0) * C);
                        u01[i] = *reinterpret_cast<const float4*>(u + (size_t)(h0 * p.up_w + w1) * C);
                        u10[i] = *reinterpret_cast<const float4*>(u + (size_t)(h1 * p.up_w + w0) * C);
                        u11[i] = *reinterpret_cast<const float4*>(u + (size_t)(h1 * p.up_w + w1) * C);
                    }
                }
                {
                    // output row o of the tile (image row r0 + o) reads ring rows slot + o - 2 .. slot + o
                    const int ub0 = X_OFF + kq * XPS + ((slot + sub - 2 + XMIR) * WP + c) * 16;
                    constexpr int NS = 9 * NU, D = 3;
                    f16x8 ring[D][NT];
                    auto ld = [&](int s) {
                        const int tap = s / NU, i = s % NU;
                        const int off = i * DU + ((tap / 3) * WP + (tap % 3)) * 16;
#pragma unroll
                        for (int t = 0; t < NT; ++t) ring[s % D][t] = *reinterpret_cast<const f16x8*>(smem + ub0 + (off + t * KGS * XPS));
                    };
#pragma unroll
                    for (int s = 0; s < D - 1; ++s) ld(s);
#pragma unroll
                    for (int s = 0; s < NS; ++s) {
                        const int tap = s / NU, i = s % NU;
                        if (s + D - 1 < NS) ld(s + D - 1);
                        const f16x8 xh = ring[s % D][0], xl = ring[s % D][1];
                        const f16x8 wh = __builtin_bit_cast(f16x8, wf[tap][0]), wl = __builtin_bit_cast(f16x8, wf[tap][1]);
                        const f32x4v c0 = tap == 0 ? f32x4v{0.f, 0.f, 0.f, 0.f} : acc[i];
                        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xl, c0, 0, 0, 0);
                        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, xh, acc[i], 0, 0, 0);
                        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xh, acc[i], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                float4 y[NU];
#pragma unroll
                for (int i = 0; i < NU; ++i) {
                    y[i] = make_float4(fmaf(acc[i][0], descale, bias.x), fmaf(acc[i][1], descale, bias.y),
                                       fmaf(acc[i][2], descale, bias.z), fmaf(acc[i][3], descale, bias.w));
                    if (epi_up) {
                        const float lh0 = 1.f - lh1[i], lw0 = 1.f - lw1;
                        y[i].x = y[i].x + (lh0 * (lw0 * u00[i].x + lw1 * u01[i].x) + lh1[i] * (lw0 * u10[i].x + lw1 * u11[i].x));
                        y[i].y = y[i].y + (lh0 * (lw0 * u00[i].y + lw1 * u01[i].y) + lh1[i] * (lw0 * u10[i].y + lw1 * u11[i].y));
                        y[i].z = y[i].z + (lh0 * (lw0 * u00[i].z + lw1 * u01[i].z) + lh1[i] * (lw0 * u10[i].z + lw1 * u11[i].z));
                        y[i].w = y[i].w + (lh0 * (lw0 * u00[i].w + lw1 * u01[i].w) + lh1[i] * (lw0 * u10[i].w + lw1 * u11[i].w));
                    }
                    st_out(p.out + o0 + i * DO, y[i]);
                }
                if (p.pm_out) {
                    // this wave's 32 pixels of the tile, per channel: (mean, M2) -- two values in the lane, the 16 pixel lanes by DPP
                    float mean[4], m2[4];
                    const float ya[2][4] = {{y[0].x, y[0].y, y[0].z, y[0].w}, {y[1].x, y[1].y, y[1].z, y[1].w}};
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        mean[r] = row_total16(ya[0][r] + ya[1][r]) * (1.f / 32.f);
                        const float d0 = ya[0][r] - mean[r], d1 = ya[1][r] - mean[r];
                        m2[r] = row_total16(fmaf(d0, d0, d1 * d1));
                    }
                    if (c == 0) {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            *reinterpret_cast<float2*>(red + ((((it - 1) & 1) * C + cq * 4 + r) * 4 + sub) * 2) = make_float2(mean[r], m2[r]);
                    }
                    prev_tile = q.n * tps + q.j;
                }
            }
            cur_next(q);
            slot = next_slot(slot);
        }
    }
    if (rbits && lane == 0) atomicOr(p.range_flag, rbits);
}

// launched (0), failed (< 0), or not this kernel's layer (1)
int launch_conv_row(const sbc_op& op, unsigned* range_flag, hipStream_t stream, bool dry) {
    static const bool off = getenv("SBC_NO_CONV_ROW") != nullptr;            // A/B aid: the Winograd kernel for these layers
    if (off || !(op.flags & SBC_CONV_F16X2) || (op.flags & SBC_CONV_F16W) || !op.weight_split) return 1;
    if (op.cin != 32 || op.cout != 32 || op.ksize != 3 || op.dil != 1 || op.W != 16 || op.H % 8 != 0) return 1;
    if ((long)op.B * (op.H / 8) < 4096) return 1;                             // (the pipeline wants many tiles per workgroup)
    const int known = SBC_CONV_F16X2 | SBC_PRO_NORM | SBC_PRO_ELU | SBC_EPI_MOMENTS_OUT | SBC_EPI_UP;
    if ((op.flags & ~known) || op.res1 || op.res2) return 1;
    if ((op.flags & SBC_PRO_NORM) && !op.stats) return 1;
    if ((op.flags & SBC_EPI_MOMENTS_OUT) && !op.aux) return 1;
    if ((op.flags & SBC_EPI_UP) && !(op.up && op.up_h > 0 && op.up_w > 0)) return 1;
    SBC_REQUIRE(op.out != op.in, "conv (row pipeline): out must not alias in");
    constexpr int ROWB = 18 * 16, XPS = (26 * ROWB + 255) / 256 * 256;
    constexpr size_t lds = (size_t)4 * 8 * 16 * 32 * 4 + (size_t)2 * 4 * XPS + 2 * 32 * 4 * 2 * 4;
    static_assert(lds <= 160 * 1024, "LDS of the one resident workgroup");
    { const int rc = ensure_dyn_lds(reinterpret_cast<const void*>(conv_row_kernel), lds); if (rc) return rc; }
    if (dry) return SBC_OK;
    RowParams p{};
    p.in = (const float*)op.in; p.out = (float*)op.out; p.w = (const uint4*)op.weight_split;
    p.bias = (const float*)op.bias;
    p.stats = (op.flags & SBC_PRO_NORM) ? (const float*)op.stats : nullptr;
    p.up = (op.flags & SBC_EPI_UP) ? (const float*)op.up : nullptr; p.up_h = op.up_h; p.up_w = op.up_w;
    p.pm_out = (op.flags & SBC_EPI_MOMENTS_OUT) ? (float*)op.aux : nullptr;
    p.range_flag = range_flag; p.calib = (float*)op.calib;
    p.flags = op.flags; p.B = op.B; p.H = op.H;
    p.tiles_per_sample = op.H / 8;
    p.ntiles = op.B * p.tiles_per_sample;
    int dev = 0, cus = 256;
    SBC_CHECK_HIP(hipGetDevice(&dev));
    SBC_CHECK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    p.tiles_per_xcd = (p.ntiles + 7) / 8;
    p.wgs_per_xcd = max(1, min(persistent_cus(cus) / 8, p.tiles_per_xcd));
    hipLaunchKernelGGL(conv_row_kernel, dim3(8 * p.wgs_per_xcd), dim3(768), lds, stream, p);
    SBC_CHECK_HIP(hipGetLastError());
    return SBC_OK;
}

}  // namespace sbc

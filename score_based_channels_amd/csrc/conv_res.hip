// One whole ResidualBlock of the score network in ONE launch (SBC_OP_RES_BLOCK), for the blocks without resampling at 64 x 16 and
// 32 channels (res1.0 / res1.1 of NCSNv2Deepest):
//       out = x + conv2(ELU(norm2(conv1(ELU(norm1(x))))))          ncsnv2/models/layers.py:443-456 (both convolutions 3x3 with bias)
// norm = InstanceNorm2dPlus (normalization.py:150-176).  InstanceNorm++ needs the statistics of the WHOLE intermediate sample before the
// second convolution can start, so a tile kernel cannot fuse the block -- a workgroup that owns a whole sample can: 1024 pixels x 32
// channels as two fp16 operand planes are 150 KB of LDS, and the first convolution's accumulators (64 registers a wave) wait in
// registers across the two barriers that form the statistics.
//
// One 8-wave workgroup per CU walks samples.  Wave (hf, sub) owns output channels 16 hf .. + 15 of image rows 16 sub .. + 15 (16 units
// of one row each).  Round 6: the sample's x stays in REGISTERS in the accumulators' layout from phase A to phase F (64 a wave) -- it is
// read from memory once, and the block's sum is formed behind the K loop as the unfused records form it -- and the filter fragments
// stream from L2 through a four-tap register ring (32 registers; round 5 kept one convolution's 72 resident and read x twice).  Per sample:
//   A  x (requested while the previous sample's epilogue ran) -> norm1 from the statistics table -> ELU -> x act_scale -> two fp16 terms
//      -> operand planes [term][8-channel group][66 rows][18 slots][8 halves];
//   B  conv1: direct implicit GEMM on v_mfma_f32_16x16x32_f16 (hh + hl + lh), K loop = LDS reads + filter requests + matrix instructions;
//   C  t = conv1 + bias; per 128-pixel tile and channel (mean, M2) -- a wave holds two whole tiles of its 16 channels: in-lane sums
//      over 8 rows, DPP sums over the 16 pixel lanes -- exchanged through LDS; 32 threads merge them per channel in a fixed order (the
//      formulas of ops.hip: inorm_from_moments_kernel), form the cross-channel "++" term and leave norm2's (mu, scale, shift) per
//      channel in LDS (round 5: every lane repeated the cross-channel sums, ~200 vector instructions of phase D);
//   D  t -> norm2 -> ELU -> x act_scale -> split -> the SAME planes (every wave is through conv1: phase C has two barriers);
//   E  conv2;
//   F  out = (conv2 x descale2 + bias2) + x; the next sample's x is requested into the registers x just left, AHEAD of the 16 stores
//      (the memory counter is in order); with SBC_EPI_MOMENTS_OUT also the (mean, M2) of the output's 128-pixel tiles for the
//      InstanceNorm++ that reads it next (tile.h) -- whole tiles per wave, no exchange.
// Five workgroup barriers per sample.  Everything is summed in an order that depends on the layer's shape only.
// Timeline of a sample (-DSBC_RES_TIMELINE, tools/prof_res.py; profiles/r06_timeline_conv_res.txt): of 66 k clocks the two K loops take
// 15 k and 22 k (the four waves that arrived first on their SIMDs finish each loop in 9 k, the other four get the matrix pipe afterwards --
// and run conv2 beside the first four's epilogue), phases A and D 7 k each, the statistics 4 k, the epilogue of the late waves 9 k: a
// CU moves its 256 KB per sample at ~50 GB/s whoever issues the requests.
#include <stdlib.h>
#include <type_traits>
#include "conv_common.h"
#ifndef RES_WD
#define RES_WD 4
#endif

namespace sbc {

typedef float f32x4v __attribute__((ext_vector_type(4)));

struct ResParams {
    const float* __restrict__ in;
    float* __restrict__ out;
    const uint4* __restrict__ w1;        // sbc_pack_conv_weight_f16x2 layout (32 -> 32, 3x3)
    const uint4* __restrict__ w2;
    const float* __restrict__ bias1;
    const float* __restrict__ bias2;
    const float* __restrict__ stats1;    // [B][3][32]: (mu, scale, shift) of norm1 (SBC_OP_INORM_STATS)
    const float* __restrict__ norm2;     // alpha | gamma | beta of norm2, [3][32]
    float* __restrict__ pm_out;          // [B][8][32][2] tile moments of the output, or NULL
    unsigned* __restrict__ range_flag;
    float* __restrict__ calib;           // sbc_f16x2_calibrate: two amax slots (conv1's input, conv2's input), else NULL
    int B;
    unsigned long long* dbg;             // SBC_RES_TIMELINE builds: clock stamps of block 0 (tools/prof_res.py)
};

#ifdef SBC_RES_TIMELINE
// block 0 only: every wave's clock at 12 points of every sample -> dbg[(iteration * 8 + wave) * 16 + k]
#define RS_T(k) do { if (blockIdx.x == 0 && p.dbg) { const unsigned long long _t = __builtin_readcyclecounter(); if (lane == 0) p.dbg[((n / gridDim.x) * 8 + wave) * 16 + (k)] = _t; } } while (0)
#else
#define RS_T(k) do { } while (0)
#endif

// sum over the 16 lanes of a DPP row (= the 16 pixels of a unit for one k-quarter); every lane of the row gets the total
__device__ __forceinline__ float row_sum16(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128 /* row_ror:8 */, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124 /* row_ror:4 */, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122 /* row_ror:2 */, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121 /* row_ror:1 */, 0xf, 0xf, false));
    return v;
}

__global__ __launch_bounds__(512, 2) void conv_res_kernel(ResParams p) {
    constexpr int C = 32, W = 16, H = 64, NT = 2, KGS = C / 8;
    constexpr int WP = W + 2, PR = H + 2;              // slots per plane row, plane rows
    constexpr int PS = (PR * WP * 16 + 255) / 256 * 256;   // bytes of one (term, k-group) plane
    constexpr int RED_OFF = NT * KGS * PS;             // [8 tiles][32 channels][(mean, M2)]
    constexpr int MV_OFF = RED_OFF + 8 * C * 2 * 4;    // mean_s[32], var_s[32]
    constexpr int NRM_OFF = MV_OFF + 2 * C * 4;        // norm2 of this sample: mu[32], scale[32], shift[32]
    constexpr int NU = 16;                             // units (image rows) per wave
    extern __shared__ __attribute__((aligned(256))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hf = wave & 1, sub = wave >> 1;          // 16-output-channel half; rows 16 sub .. + 15
    const int kq = lane >> 4, c = lane & 15;
    const int cq = 4 * hf + kq;                        // channel quad of this lane's four outputs (and of the x it stages)

    // ---- filter fragments STREAM from L2 through a ring of WD taps (8 registers a tap), re-read for every sample.  (Round 5 kept one
    // convolution's 72 registers resident; with x resident as well -- below -- that does not fit.)  The first WD - 1 taps are requested
    // before the barrier in front of the convolution, tap + WD - 1 when tap starts: 2 x 768 matrix cycles ahead of its first use.
    constexpr int WD = RES_WD;
    static_assert(WD % 2 == 0, "an odd ring depth makes hipcc merge the two K loops into one loop over a scratch-resident ring");
    uint4 wr[WD][NT];
    const uint4* wb = nullptr;                          // this lane's fragment of tap 0, term 0 of the convolution in flight
    auto ldw = [&](int tap) {
#pragma unroll
        for (int t = 0; t < NT; ++t) wr[tap % WD][t] = wb[(tap * (C / 16) * NT + t) * 64];
    };
    auto load_w = [&](const uint4* __restrict__ w) {
        int lo = lane;
        asm volatile("" : "+v"(lo));                   // (per-lane offsets formed here: hoisted out of the sample loop they are spilled)
        wb = w + ((lo >> 5) * NT) * 64 + (16 * hf + (lo & 15)) + 32 * ((lo >> 4) & 1);
#pragma unroll
        for (int tap = 0; tap < WD - 1; ++tap) ldw(tap);
    };
    const float4 t1 = f16x2_trailer(reinterpret_cast<const float4*>(p.w1), 9 * (C / 16) * (C / 32) * NT);
    const float4 t2 = f16x2_trailer(reinterpret_cast<const float4*>(p.w2), 9 * (C / 16) * (C / 32) * NT);
    const float scale1 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, t1.x)));
    const float scale2 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, t2.x)));
    const float descale1 = t1.y, descale2 = t2.y;
    // sample-invariant parameters of the lane's four output channels (a global load inside a phase is a round trip nobody hides)
    const float4 b1 = *reinterpret_cast<const float4*>(p.bias1 + cq * 4), b2 = *reinterpret_cast<const float4*>(p.bias2 + cq * 4);
    // ... and, in the 32 threads that form norm2's parameters, alpha | gamma | beta of their channel
    float al = 0.f, ga = 0.f, be = 0.f;
    if (tid < C) { al = p.norm2[tid]; ga = p.norm2[C + tid]; be = p.norm2[2 * C + tid]; }
    unsigned rbits = 0;
    // the calibration found one of the two convolutions' inputs below 2^-4: this kernel has the exp(x) - 1 form of ELU only
    if (t1.w != 0.f || t2.w != 0.f) rbits |= 4u;

    // ---- zero the planes once: the padding columns and the rows above and below the image are never written again
    for (int i = tid; i < NT * KGS * PS / 16; i += 512) *reinterpret_cast<uint4*>(smem + i * 16) = make_uint4(0, 0, 0, 0);

    // ---- x of the sample in flight stays in REGISTERS, in the accumulators' layout (lane (kq, c): pixel c of the wave's 16 rows, channel
    // quad cq): phase A stages it, phase F adds it to conv2 -- no second read.  The next sample's x is requested into the same registers
    // as soon as phase F has formed its sums.
    float4 xr[NU], mu, sc, sh;                         // ... and norm1's (mu, scale, shift) of the lane's channel quad
    auto request_x = [&](int n) {
        int lo = lane;
        asm volatile("" : "+v"(lo));
        const float* src = p.in + (size_t)(((n * H + sub * NU) * W + (lo & 15)) * C + (4 * hf + (lo >> 4)) * 4);
#pragma unroll
        for (int i = 0; i < NU; ++i) xr[i] = *reinterpret_cast<const float4*>(src + i * W * C);
        const float* st = p.stats1 + (size_t)n * 3 * C + (4 * hf + (lo >> 4)) * 4;
        mu = *reinterpret_cast<const float4*>(st); sc = *reinterpret_cast<const float4*>(st + C); sh = *reinterpret_cast<const float4*>(st + 2 * C);
    };
    int n = blockIdx.x;
    if (n < p.B) request_x(n);
    load_w(p.w1);

    for (; n < p.B; n += gridDim.x) {
        // every wave is through the previous sample's conv2 (first sample: the planes are zeroed)
        RS_T(0);
        __syncthreads();
        RS_T(1);
        // this lane's slot in the operand planes (the same in phases A and D): pixel c of plane row 16 sub + 1, k-group cq >> 1, half cq & 1
        int lq = lane;
        asm volatile("" : "+v"(lq));                       // (formed per sample, see load_w)
        unsigned char* const dst0 = smem + ((4 * hf + (lq >> 4)) >> 1) * PS + ((sub * NU + 1) * WP + (lq & 15) + 1) * 16 + ((lq >> 4) & 1) * 8;
        // ---- A: x -> norm1 -> ELU -> split -> planes
        float ta = 0.f, tb = 0.f;
#pragma unroll
        for (int i = 0; i < NU; ++i) {
            float4 v = make_float4(fmaf(xr[i].x - mu.x, sc.x, sh.x), fmaf(xr[i].y - mu.y, sc.y, sh.y),
                                   fmaf(xr[i].z - mu.z, sc.z, sh.z), fmaf(xr[i].w - mu.w, sc.w, sh.w));
            v = elu4(v);
            StageScale ss{scale1, ta};
            scale_track(v, &ss);
            ta = ss.amax;
            uint2 h, l;
            split_f16x2(v, scale1, h, l);
            *reinterpret_cast<uint2*>(dst0 + i * WP * 16) = h;
            *reinterpret_cast<uint2*>(dst0 + i * WP * 16 + KGS * PS) = l;
        }
        pair_range_tile(ta, scale1, rbits, p.calib);
        RS_T(2);
        lds_barrier();
        RS_T(3);

        // one convolution over this wave's 16 rows: acc[i] = D[16 couts][16 pixels of row 16 sub + i]
        f32x4v acc[NU];
        auto conv = [&]() __attribute__((always_inline)) {
            // top-left tap of row u = 16 sub + i, pixel c: plane row u (image row u - 1), slot c (column c - 1).  (One base per term
            // plane, opaque: folded into the 144 read addresses the second plane's offset exceeds the 16-bit immediate and hipcc keeps
            // ~50 precomputed addresses alive through the whole sample loop.)
            int ub[NT];
            ub[0] = kq * PS + ((sub * NU) * WP + c) * 16;
#pragma unroll
            for (int t = 1; t < NT; ++t) { ub[t] = ub[0] + t * KGS * PS; asm volatile("" : "+v"(ub[t])); }
            constexpr int NS = 9 * NU, D = 3;
            f16x8 ring[D][NT];
            auto ld = [&](int s) {                                            // s is a compile-time constant at every call
                const int tap = s / NU, i = s % NU;
                const int off = ((i + tap / 3) * WP + (tap % 3)) * 16;
#pragma unroll
                for (int t = 0; t < NT; ++t) ring[s % D][t] = *reinterpret_cast<const f16x8*>(smem + ub[t] + off);
            };
#pragma unroll
            for (int s = 0; s < D - 1; ++s) ld(s);
            __builtin_amdgcn_s_setprio(3);
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int tap = s / NU, i = s % NU;
                if (s + D - 1 < NS) ld(s + D - 1);
                if (i == 0 && tap + WD - 1 < 9) ldw(tap + WD - 1);
                const f16x8 xh = ring[s % D][0], xl = ring[s % D][1];
                const f16x8 wh = __builtin_bit_cast(f16x8, wr[tap % WD][0]), wl = __builtin_bit_cast(f16x8, wr[tap % WD][1]);
                const f32x4v c0 = tap == 0 ? f32x4v{0.f, 0.f, 0.f, 0.f} : acc[i];
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xl, c0, 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, xh, acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xh, acc[i], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            __builtin_amdgcn_s_setprio(0);
        };
        // (mean, M2) of the lane's 4 channels over one 128-pixel tile = units 8 T .. 8 T + 7 of this wave (every lane of a row gets them)
        auto tile_moments = [&](int T, float (&mean)[4], float (&m2)[4]) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float s = 0.f;
#pragma unroll
                for (int i = 0; i < 8; ++i) s += acc[8 * T + i][r];
                mean[r] = row_sum16(s) * (1.f / 128.f);
                float q = 0.f;
#pragma unroll
                for (int i = 0; i < 8; ++i) { const float d = acc[8 * T + i][r] - mean[r]; q = fmaf(d, d, q); }
                m2[r] = row_sum16(q);
            }
        };

        // ---- B: conv1
        conv();
        load_w(p.w2);
        RS_T(4);
        // ---- C: t = conv1 + bias1 (in place), statistics of t over the whole sample
        {
#pragma unroll
            for (int i = 0; i < NU; ++i) {
                acc[i][0] = fmaf(acc[i][0], descale1, b1.x); acc[i][1] = fmaf(acc[i][1], descale1, b1.y);
                acc[i][2] = fmaf(acc[i][2], descale1, b1.z); acc[i][3] = fmaf(acc[i][3], descale1, b1.w);
            }
        }
        float* red = reinterpret_cast<float*>(smem + RED_OFF);
        float* mv = reinterpret_cast<float*>(smem + MV_OFF);
        float* nrm = reinterpret_cast<float*>(smem + NRM_OFF);
#pragma unroll
        for (int T = 0; T < 2; ++T) {
            float mean[4], m2[4];
            tile_moments(T, mean, m2);
            if (c == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    *reinterpret_cast<float2*>(red + ((2 * sub + T) * C + cq * 4 + r) * 2) = make_float2(mean[r], m2[r]);
            }
        }
        RS_T(5);
        lds_barrier();
        RS_T(6);
        if (tid < C) {
            // one channel each: the 8 tiles in order (ops.hip: inorm_from_moments_kernel), then -- the same 32 lanes of wave 0, through
            // LDS, every lane summing all 32 channels in index order -- the cross-channel "++" term, and norm2's (mu, scale, shift)
            float mw[8], m_c = 0.f, q = 0.f;
#pragma unroll
            for (int t = 0; t < 8; ++t) { const float2 v = *reinterpret_cast<const float2*>(red + (t * C + tid) * 2); mw[t] = v.x; m_c += v.x; q += v.y; }
            m_c *= 0.125f;
            float dd = 0.f;
#pragma unroll
            for (int t = 0; t < 8; ++t) { const float d = mw[t] - m_c; dd = fmaf(d, d, dd); }
            const float var_c = fmaf(128.f, dd, q) * (1.f / (float)(H * W));
            mv[tid] = m_c;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                 // (one wave: its LDS operations complete in order)
            float m = 0.f;
#pragma unroll
            for (int k = 0; k < C; k += 4) { const float4 v = *reinterpret_cast<const float4*>(mv + k); m += v.x; m += v.y; m += v.z; m += v.w; }
            m *= 1.f / (float)C;
            float v = 0.f;
#pragma unroll
            for (int k = 0; k < C; k += 4) {
                const float4 u = *reinterpret_cast<const float4*>(mv + k);
                float d = u.x - m; v = fmaf(d, d, v);
                d = u.y - m; v = fmaf(d, d, v);
                d = u.z - m; v = fmaf(d, d, v);
                d = u.w - m; v = fmaf(d, d, v);
            }
            v *= 1.f / (float)(C - 1);
            const float rs = 1.f / sqrtf(v + 1e-5f);
            nrm[tid] = m_c;
            nrm[C + tid] = ga / sqrtf(fmaxf(var_c, 0.f) + 1e-5f);
            nrm[2 * C + tid] = fmaf(ga, (m_c - m) * rs * al, be);
        }
        lds_barrier();
        RS_T(7);
        const float4 nmu = *reinterpret_cast<const float4*>(nrm + cq * 4), nsc = *reinterpret_cast<const float4*>(nrm + C + cq * 4),
                     nsh = *reinterpret_cast<const float4*>(nrm + 2 * C + cq * 4);
        // ---- D: t -> norm2 -> ELU -> split -> the same planes (conv2's operand)
#pragma unroll
        for (int i = 0; i < NU; ++i) {
            float4 v = make_float4(fmaf(acc[i][0] - nmu.x, nsc.x, nsh.x), fmaf(acc[i][1] - nmu.y, nsc.y, nsh.y),
                                   fmaf(acc[i][2] - nmu.z, nsc.z, nsh.z), fmaf(acc[i][3] - nmu.w, nsc.w, nsh.w));
            v = elu4(v);
            StageScale ss{scale2, tb};
            scale_track(v, &ss);
            tb = ss.amax;
            uint2 h, l;
            split_f16x2(v, scale2, h, l);
            *reinterpret_cast<uint2*>(dst0 + i * WP * 16) = h;
            *reinterpret_cast<uint2*>(dst0 + i * WP * 16 + KGS * PS) = l;
        }
        pair_range_tile(tb, scale2, rbits, p.calib ? p.calib + 1 : nullptr);
        RS_T(8);
        lds_barrier();
        RS_T(9);
        // ---- E: conv2
        conv();
        load_w(p.w1);                                       // (conv1's first taps for the next sample; harmless behind the last one)
        RS_T(10);
        // ---- F: out = (conv2 x descale2 + bias2) + x -- the order of the unfused records: one rounding at the output's magnitude;
        // then the next sample's x is requested (AHEAD of the stores: the memory counter is in order, behind 16 stores phase A would
        // wait for every one of them to be acknowledged), the stores, the tile moments of the output
#pragma unroll
        for (int i = 0; i < NU; ++i) {
            acc[i][0] = fmaf(acc[i][0], descale2, b2.x) + xr[i].x; acc[i][1] = fmaf(acc[i][1], descale2, b2.y) + xr[i].y;
            acc[i][2] = fmaf(acc[i][2], descale2, b2.z) + xr[i].z; acc[i][3] = fmaf(acc[i][3], descale2, b2.w) + xr[i].w;
        }
        RS_T(12);
        __builtin_amdgcn_sched_barrier(0);
        request_x(min(n + (int)gridDim.x, p.B - 1));       // (unconditional: behind a branch the new x lands in other registers and is
                                                            // COPIED at the end of the loop body -- after a wait for all of it)
        RS_T(13);
        __builtin_amdgcn_sched_barrier(0);
        {
            float* const o = p.out + (size_t)(((n * H + sub * NU) * W + (lq & 15)) * C + (4 * hf + (lq >> 4)) * 4);
#pragma unroll
            for (int i = 0; i < NU; ++i) st_out(o + i * W * C, make_float4(acc[i][0], acc[i][1], acc[i][2], acc[i][3]));
        }
        RS_T(14);
        if (p.pm_out) {
#pragma unroll
            for (int T = 0; T < 2; ++T) {
                float mean[4], m2[4];
                tile_moments(T, mean, m2);
                if (c == 0) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        *reinterpret_cast<float2*>(p.pm_out + (((size_t)n * 8 + 2 * sub + T) * C + cq * 4 + r) * 2) = make_float2(mean[r], m2[r]);
                }
            }
        }
        RS_T(11);
    }
    if (rbits && lane == 0) atomicOr(p.range_flag, rbits);
}

int launch_res_block(const sbc_op& op, hipStream_t stream, bool dry) {
    SBC_REQUIRE(op.in && op.out && op.weight_split && op.weight2_split && op.bias && op.bias2 && op.stats && op.norm2,
                "res_block: in / out / weight_split / weight2_split / bias / bias2 / stats / norm2 must be set");
    SBC_REQUIRE(op.cin == 32 && op.cout == 32 && op.ksize == 3 && op.dil == 1 && op.H == 64 && op.W == 16,
                "res_block: 32 -> 32 -> 32 channels, 3x3, undilated, 64 x 16 samples (got %d -> %d, %dx%d)", op.cin, op.cout, op.H, op.W);
    SBC_REQUIRE(op.B > 0, "res_block: bad batch %d", op.B);
    SBC_REQUIRE(op.out != op.in, "res_block: out must not alias in (the next sample is requested while the epilogue stores)");
    SBC_REQUIRE((op.flags & SBC_CONV_F16X2) && !(op.flags & SBC_CONV_F16W), "res_block: SBC_CONV_F16X2 only (the weight forms it reads)");
    SBC_REQUIRE(!(op.flags & SBC_EPI_MOMENTS_OUT) || op.aux, "res_block: SBC_EPI_MOMENTS_OUT without aux");
    SBC_REQUIRE(!(op.flags & (SBC_EPI_POOL | SBC_EPI_UP | SBC_EPI_ELUGRAD | SBC_EPI_RES1_ELU)) && !op.res1 && !op.res2 && !op.up,
                "res_block: the residual operand is the input itself; no other epilogue");
    SBC_REQUIRE((long)op.B * op.H * op.W * op.cin <= 0x7fffffffL, "res_block: tensor exceeds the 32-bit element index");
    constexpr size_t lds = (size_t)2 * 4 * ((66 * 18 * 16 + 255) / 256 * 256) + 8 * 32 * 2 * 4 + 2 * 32 * 4 + 3 * 32 * 4;
    static_assert(lds <= 160 * 1024, "LDS of the one resident workgroup");
    { const int rc = ensure_dyn_lds(reinterpret_cast<const void*>(conv_res_kernel), lds); if (rc) return rc; }
    unsigned* word = nullptr;
    { const int rc = range_flag_ptr(&word); if (rc) return rc; }
    if (dry) return SBC_OK;
    ResParams p{};
    p.in = (const float*)op.in; p.out = (float*)op.out;
    p.w1 = (const uint4*)op.weight_split; p.w2 = (const uint4*)op.weight2_split;
    p.bias1 = (const float*)op.bias; p.bias2 = (const float*)op.bias2;
    p.stats1 = (const float*)op.stats; p.norm2 = (const float*)op.norm2;
    p.pm_out = (op.flags & SBC_EPI_MOMENTS_OUT) ? (float*)op.aux : nullptr;
    p.range_flag = word; p.calib = (float*)op.calib; p.B = op.B;
#ifdef SBC_RES_TIMELINE
    p.dbg = (op.flags & SBC_EPI_MOMENTS_OUT) ? nullptr : (unsigned long long*)op.aux;
#endif
    int dev = 0, cus = 256;
    SBC_CHECK_HIP(hipGetDevice(&dev));
    SBC_CHECK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    hipLaunchKernelGGL(conv_res_kernel, dim3(balanced_sample_grid(op.B, cus)), dim3(512), lds, stream, p);
    SBC_CHECK_HIP(hipGetLastError());
    return SBC_OK;
}

}  // namespace sbc

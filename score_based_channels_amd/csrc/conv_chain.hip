// A CHAIN of RCU / CRP blocks of one RefineBlock in ONE launch, for the lowest-resolution level of the score network
// (SBC_OP_CHAIN; 8 x 2 samples of 64 or 128 channels: refine1, refine2, refine31 and the low-resolution input of refine3,
// ncsnv2/models/ncsnv2.py:257-260,284-287; blocks: layers.py:76-83 (CRP), :126-134 (RCU), wiring :234-249).
//
// Why: 53 of the network's 113 convolutions run at 8 x 2 -- 27 200 pixels per launch at 1700 trajectories -- and each was its own
// launch (plus 12 max-pool launches): the level cost 1.05 ms of a 5.33 ms one-stream step and 0.77 ms of the two-stream step for
// 18 % of the FLOPs (profiles/r05_level_bounds.txt), every launch a load -> stage -> K loop -> store chain of 12-25 us that
// neither fills the chip nor overlaps with its neighbours.  Here a 512-thread workgroup owns EIGHT samples for a whole chain:
//   * the running tensor x lives in REGISTERS in the accumulator layout of v_mfma_f32_16x16x32_f16 (a wave owns 16 output
//     channels of its pixel units for the whole launch, so `x + conv(...)` never moves data);
//   * the convolution operand -- ELU / max-pool of x or of the previous convolution's accumulators, x act_scale, split into two
//     fp16 terms (conv_mode f16x2) -- goes to ONE set of LDS planes that every convolution of the chain re-uses;
//   * only the filters stream: 2 KB per (tap, 32 input channels) and wave from L2 through a four-deep register ring.
// Pixel units are arranged by COLUMN PARITY: a unit's 16 pixels are the 8 rows of one image column of two samples.  At a width of
// two, a tap with dx = -1 reads padding for every pixel of column 0 (and dx = +1 for column 1): whole units skip those taps -- 6
// of 9 taps per unit, a third of the matrix instructions gone, with no change to any sum (the skipped products are exact zeros).
// The same arrangement makes the 5 x 5 max pool lane-local: the window always spans both columns, which sit in the same lane of
// two units of the same wave; the five rows are four cross-lane reads inside a 16-lane group.
//
// LDS planes [term][8-channel group][sample][column][12 row slots][8 halves]: slots 1 .. 8 = rows, 0 and 9 stay zero (the padding
// above and below); 24 slots per sample put the second sample of a unit 8 slots (mod 16) behind the first, so the 16 lanes of each
// ds_read_b128 service group hit 16 different 16-byte slots: conflict-free, for every tap (conv_dp.hip has the rule).
//
// The 16 x 4 level (64 channels: refine3 and the low-resolution input of refine4) runs the same kernel with W = 4: a unit is ONE
// column of ONE sample (its 16 rows; 18 row slots per column), four samples per workgroup, a wave owns all four columns of two
// samples; the units of column 0 skip the dx = -1 taps and those of column 3 the dx = +1 taps (a sixth of the matrix work).
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include "conv_common.h"

namespace sbc {

typedef float f32x4v __attribute__((ext_vector_type(4)));

struct ChainParams {
    const float* __restrict__ in;
    float* __restrict__ out;
    const uint4* w[SBC_CHAIN_MAX_BLOCKS][3];        // conv 1, conv 2, [shortcut conv of a RES block or NULL]
    const float* bias[SBC_CHAIN_MAX_BLOCKS][3];     // RES blocks: their biases
    const float* norm[SBC_CHAIN_MAX_BLOCKS][2];     // RES blocks: alpha | gamma | beta of normalize1 / normalize2, [3][C]
    int type[SBC_CHAIN_MAX_BLOCKS];
    int dil[SBC_CHAIN_MAX_BLOCKS];
    int n_blocks;
    unsigned* __restrict__ range_flag;
    float* __restrict__ calib;          // sbc_f16x2_calibrate: 3 amax slots per block (inputs of conv 1, conv 2, shortcut conv), else NULL
    int B;
    unsigned long long* dbg;            // SBC_CHAIN_TIMELINE builds (tools/prof_chain.py): clock stamps of the middle workgroup
};

#ifdef SBC_CHAIN_TIMELINE
// the middle workgroup's waves stamp eight points of every phase -> dbg[wave][phase][8]:
//   0 phase parameters read, 1 operand values formed (ELU / pool / norm), 2 split + filter prologue requested, 3 planes free (barrier),
//   4 planes written (barrier), 5 K loop done, 6 result taken
#define CH_T(k) do { if (p.dbg && blockIdx.x == gridDim.x / 2) { const unsigned long long _t = __builtin_readcyclecounter(); if (lane == 0) p.dbg[(wave * 16 + phase_no) * 8 + (k)] = _t; } } while (0)
#else
#define CH_T(k) do { } while (0)
#endif

__device__ __forceinline__ float4 to_f4(f32x4v v) { return make_float4(v[0], v[1], v[2], v[3]); }
__device__ __forceinline__ f32x4v to_v4(float4 v) { return f32x4v{v.x, v.y, v.z, v.w}; }
// (element by element: arithmetic on the vector type compiles to v_pk_mul_f32 / v_pk_add_f32, which this library does not use -- Makefile)
__device__ __forceinline__ f32x4v vscale4(f32x4v a, float s) { return f32x4v{a[0] * s, a[1] * s, a[2] * s, a[3] * s}; }
__device__ __forceinline__ f32x4v vadd4(f32x4v a, f32x4v b) { return f32x4v{a[0] + b[0], a[1] + b[1], a[2] + b[2], a[3] + b[3]}; }
__device__ __forceinline__ f32x4v vmax4(f32x4v a, f32x4v b) {
    return f32x4v{__builtin_fmaxf(a[0], b[0]), __builtin_fmaxf(a[1], b[1]), __builtin_fmaxf(a[2], b[2]), __builtin_fmaxf(a[3], b[3])};
}

// compile-time loop: f(std::integral_constant<int, I>) for I in [B, E) -- a #pragma unroll of ~100 large iterations is only a hint
template <int B, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}

// The K loop's micro-steps.  K step s = (tap, 32-channel slice kh), tap-major; within a K step the units whose source column
// x + dx lies inside the image (x = unit index mod W) are taken in pairs, in ascending order:
//   W = 2:  dx = 0: all NU units;  dx = -1: the column-1 units;  dx = +1: the column-0 units (NU / 2 each)
//   W = 4:  dx = 0: all NU units;  dx = -1: columns 1 .. 3;      dx = +1: columns 0 .. 2     (3 NU / 4 each)
struct MicroStep { int s, u[4]; bool first_of_step; };        // up to four units per micro-step; u[k] < 0: unused slot
template <int W, int NU>
__host__ __device__ constexpr int valid_units(int dx) { return dx == 0 ? NU : NU * (W - 1) / W; }
template <int W, int NU>
__host__ __device__ constexpr int valid_unit(int dx, int k) {          // the k-th unit (ascending) with 0 <= x + dx < W, or -1
    int seen = 0;
    for (int i = 0; i < NU; ++i) {
        const int xx = i % W + dx;
        if (xx < 0 || xx >= W) continue;
        if (seen == k) return i;
        ++seen;
    }
    return -1;
}
template <int W, int NU, int KH>
__host__ __device__ constexpr int micro_steps() { return 3 * ((valid_units<W, NU>(0) + 3) / 4 + 2 * ((valid_units<W, NU>(1) + 3) / 4)) * KH; }
template <int W, int NU, int KH>
__host__ __device__ constexpr MicroStep micro_step(int m) {
    constexpr int M0 = (valid_units<W, NU>(0) + 3) / 4, M1 = (valid_units<W, NU>(1) + 3) / 4, ROW = (M0 + 2 * M1) * KH;
    const int row = m / ROW;
    int r = m % ROW, tapc = 0, kh = 0, q = 0;
    if (r < M1 * KH) { tapc = 0; kh = r / M1; q = r % M1; }
    else if (r < (M1 + M0) * KH) { r -= M1 * KH; tapc = 1; kh = r / M0; q = r % M0; }
    else { r -= (M1 + M0) * KH; tapc = 2; kh = r / M1; q = r % M1; }
    return MicroStep{(3 * row + tapc) * KH + kh,
                     {valid_unit<W, NU>(tapc - 1, 4 * q), valid_unit<W, NU>(tapc - 1, 4 * q + 1), valid_unit<W, NU>(tapc - 1, 4 * q + 2),
                      valid_unit<W, NU>(tapc - 1, 4 * q + 3)}, q == 0};
}

// samples per workgroup of NW waves (one 16-output-channel block per wave, 8 units per wave -- 4 at W = 2 with 64 channels)
__host__ __device__ constexpr int chain_samples(int C, int W, int NW) {
    const int groups = NW / (C / 16);                                   // unit groups among the waves
    return W == 2 ? (C == 128 ? 8 : 4 * groups) : W == 4 ? 2 * groups : (groups + 1) / 2;
}

// C channels in = out; NW waves; unit i of a wave = (sample group i / W, column i % W):
//   W = 2 (8 x 2 samples), G = 8 samples per workgroup, a unit = one column of TWO samples (lane n: sample n >> 3, row n & 7);
//          C = 128: wave = 16-output-channel block cb, all 8 units;  C = 64: wave = (cb, half), the 4 units of two sample pairs;
//   W = 4 (16 x 4 samples, C = 64), G = 4: a unit = one column of ONE sample (lane n: row n); wave = (cb, half), 8 units;
//   W = 8 (32 x 8 samples): a unit = one column of one HALF of a sample (lane n: row 16 half + n), 8 units = the eight columns --
//          statistics and the max pool's boundary rows cross the two waves of a sample through LDS;  C = 64, G = 1: wave = (cb, half);  C = 32, G = 2: wave = (cb, sample, half).
// GD = 2 (W = 2 only): HALF the samples per workgroup (four at 128 channels, two at 64) -- twice the workgroups for batches that would
// leave most of the chip idle otherwise (a launch of 425 samples is 54 workgroups of eight); the same sums in the same order.
// GD = 4 (128 channels): a QUARTER -- two samples, one unit pair per wave -- for a rank's share of a sharded run (213 samples: 107
// workgroups instead of 54; a convolution's time is its filter stream + barriers either way, half the matrix work per workgroup).
template <int C, int W, int NW, int GD = 1>
__global__ __launch_bounds__(64 * NW, 2) void conv_chain_kernel(ChainParams p) {
    constexpr int H = W == 2 ? 8 : W == 4 ? 16 : 32;
    constexpr int CG = C / 8, KH = C / 32, NCB = C / 16, NHALF = NW / NCB;   // NHALF: unit groups among the waves
    static_assert(NW % NCB == 0 && NHALF >= 1, "a wave owns one 16-output-channel block");
    // samples per workgroup: every unit group of waves takes 8 units (4 at W = 2, C = 64)
    constexpr int G = chain_samples(C, W, NW) / GD;
    static_assert(GD == 1 || (GD == 2 && W == 2) || (GD == 4 && W == 2 && C == 128), "half groups exist at a width of two, quarter groups at 128 channels");
    constexpr int NTH = 64 * NW;
    constexpr int SPU = W == 2 ? 2 : 1;                   // samples per unit
    constexpr int NSG = W == 8 ? 1 : G / SPU / NHALF;     // sample groups (pairs at W = 2, samples at W = 4, half samples at W = 8) per wave
    constexpr int NU = NSG * W;                           // units per wave
    constexpr int CP = H + (W == 2 ? 4 : 2);              // row slots per column: H rows + a zero slot above and below (+ alignment)
    constexpr int SP = W * CP;                            // slots per sample
    constexpr int PS = G * SP * 16;                       // bytes of one (term, channel group) plane
    constexpr int TERM = CG * PS;                         // low-term planes behind the high-term planes
    static_assert((W == 2 && (C == 64 || C == 128)) || (W == 4 && C == 64) || (W == 8 && (C == 32 || C == 64)), "shapes of the header");
    static_assert(PS % 256 == 0 && (W != 2 || (SP % 16) == 8), "plane stride = whole bank rows; second sample of a W = 2 unit 8 slots (mod 16) on");
    static_assert(NU == 8 || NU == 4 || NU == 2, "accumulator budget: at most 8 units per wave");
    extern __shared__ __attribute__((aligned(256))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cb = wave % NCB, uh = wave / NCB;
    // first sample (within the workgroup's G) of this wave, and the image row of its units' lane 0
    const int w0 = W == 8 ? uh / 2 : uh * NSG * SPU;
    const int yoff = W == 8 ? 16 * (uh & 1) : 0;
    const int kq = lane >> 4, n = lane & 15;
    const int sp = W == 2 ? n >> 3 : 0, y = yoff + (W == 2 ? n & 7 : n);   // sample of the unit, image row of this lane's pixel
    const int s0 = blockIdx.x * G;

    // ---- zero the planes once: the slots above and below every column are never written
    for (int i = tid; i < 2 * TERM / 16; i += NTH) *reinterpret_cast<uint4*>(smem + i * 16) = make_uint4(0, 0, 0, 0);

    // ---- the running tensor, accumulator layout: lane (kq, n) holds channels 16 cb + 4 kq .. + 3 of pixel n of each unit
    f32x4v xs[NU];
    const int ch0 = 16 * cb + 4 * kq;
#pragma unroll
    for (int i = 0; i < NU; ++i) {
        const int s = s0 + w0 + (i / W) * SPU + sp;
        xs[i] = f32x4v{0.f, 0.f, 0.f, 0.f};
        if (s < p.B) xs[i] = *reinterpret_cast<const f32x4v*>(p.in + ((size_t)(s * H + y) * W + (i % W)) * C + ch0);
    }
    // lane parts of the LDS addresses (bytes); the rest are compile-time constants: unit i, source column xx, tap row dy ->
    // (((i / W) * SPU) * SP + xx * CP + 1 + dy) * 16
    const int rd_base = kq * PS + ((w0 + sp) * SP + y) * 16;
    const int wr_base = (2 * cb + (kq >> 1)) * PS + ((w0 + sp) * SP + 1 + y) * 16 + (kq & 1) * 8;
    // lane part of the filter-fragment index (packed layout [tap][C/16 input groups][C/32 output blocks][2 terms][64 lanes], conv_pair.hip)
    const int wl_base = (((kq >> 1) * (C / 32) + (cb >> 1)) * 2) * 64 + (16 * (cb & 1) + n) + 32 * (kq & 1);

    f32x4v acc[NU];
#pragma unroll
    for (int i = 0; i < NU; ++i) acc[i] = f32x4v{0.f, 0.f, 0.f, 0.f};
    float prev_descale = 1.f;
    unsigned rbits = 0;
    float* const nscr = reinterpret_cast<float*>(smem + 2 * TERM);       // InstanceNorm++ scratch: mu [G][C], then (m, 1 / sqrt(v + eps)) [G][2]
    float* const hscr = nscr + G * C + 2 * G;                            // W = 8: half-sample sums, two buffers of [NW][16]
    float* const pscr = hscr + 2 * NW * 16;                              // W = 8: boundary rows of the max pool, [NW][2 rows][NU][4 kq] float4
    __syncthreads();

    // ---- InstanceNorm2dPlus (normalization.py:163-176) + ELU of a tensor held in the accumulator layout, for the RES blocks: every
    // statistic is formed in a fixed order that depends on the layout only (a sample's numbers do not depend on its neighbours):
    //   per (sample, channel): mean and biased variance over the H W pixels -- the W units of the sample group in this lane, then the
    //   rows (lanes of the same sample) by xor shuffles -- two passes, as F.instance_norm computes them;
    //   per sample: mean m and UNBIASED variance v of the channel means -- every wave publishes its 16 channels' means in LDS, wave s
    //   reduces sample s, everybody reads (m, 1 / sqrt(v + 1e-5)) back: two workgroup barriers.
    auto norm_elu = [&](f32x4v (&v)[NU], const float* __restrict__ agb, bool elu_accurate) {
        constexpr int RL = W == 2 ? 8 : 16;                           // lanes (rows) of one sample in a 16-lane group
        const float inv_hw = 1.f / (float)(H * W);
        f32x4v mu[NSG], rs[NSG];
        // W = 8: the rows of a sample are split over two waves (uh, uh ^ 1, same channel block): the two half sums meet in LDS and
        // both waves add them in the same order (rows 0..15 first), one workgroup barrier per statistic
        auto join_halves = [&](f32x4v& t, float* buf) {
            if constexpr (W == 8) {
                if (n == 0) *reinterpret_cast<f32x4v*>(buf + wave * 16 + 4 * kq) = t;
                lds_barrier();
                const f32x4v o = *reinterpret_cast<const f32x4v*>(buf + (wave ^ NCB) * 16 + 4 * kq);
                t = (uh & 1) ? vadd4(o, t) : vadd4(t, o);
            }
        };
#pragma unroll
        for (int g = 0; g < NSG; ++g) {
            f32x4v sum = v[g * W];
#pragma unroll
            for (int x = 1; x < W; ++x) sum = vadd4(sum, v[g * W + x]);
#pragma unroll
            for (int m = 1; m < RL; m <<= 1)
#pragma unroll
                for (int e = 0; e < 4; ++e) sum[e] += __shfl_xor(sum[e], m);
            join_halves(sum, hscr);
            mu[g] = vscale4(sum, inv_hw);
            f32x4v m2 = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int x = 0; x < W; ++x)
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float t = v[g * W + x][e] - mu[g][e]; m2[e] = fmaf(t, t, m2[e]); }
#pragma unroll
            for (int m = 1; m < RL; m <<= 1)
#pragma unroll
                for (int e = 0; e < 4; ++e) m2[e] += __shfl_xor(m2[e], m);
            join_halves(m2, hscr + NW * 16);
#pragma unroll
            for (int e = 0; e < 4; ++e) rs[g][e] = 1.f / sqrtf(m2[e] * inv_hw + 1e-5f);
            if (y == 0) *reinterpret_cast<f32x4v*>(nscr + (w0 + g * SPU + sp) * C + ch0) = mu[g];
        }
        lds_barrier();
        if (wave < G) {                                                // wave s: mean and unbiased variance over the channels of sample s
            float a = 0.f;
            for (int c = lane; c < C; c += 64) a += nscr[wave * C + c];
            for (int m = 1; m < 64; m <<= 1) a += __shfl_xor(a, m);
            const float mm = a * (1.f / (float)C);
            float q = 0.f;
            for (int c = lane; c < C; c += 64) { const float t = nscr[wave * C + c] - mm; q = fmaf(t, t, q); }
            for (int m = 1; m < 64; m <<= 1) q += __shfl_xor(q, m);
            if (lane == 0) { nscr[G * C + 2 * wave] = mm; nscr[G * C + 2 * wave + 1] = 1.f / sqrtf(q * (1.f / (float)(C - 1)) + 1e-5f); }
        }
        lds_barrier();
        const f32x4v al = *reinterpret_cast<const f32x4v*>(agb + ch0), ga = *reinterpret_cast<const f32x4v*>(agb + C + ch0),
                     be = *reinterpret_cast<const f32x4v*>(agb + 2 * C + ch0);
#pragma unroll
        for (int g = 0; g < NSG; ++g) {
            const float mm = nscr[G * C + 2 * (w0 + g * SPU + sp)], rv = nscr[G * C + 2 * (w0 + g * SPU + sp) + 1];
            f32x4v sc, sh;                                              // out = (x - mu) * sc + sh,  sc = gamma rstd,  sh = gamma mhat alpha + beta
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float mhat = (mu[g][e] - mm) * rv;
                sc[e] = ga[e] * rs[g][e];
                sh[e] = fmaf(ga[e], mhat * al[e], be[e]);
            }
#pragma unroll
            for (int x = 0; x < W; ++x) {
                f32x4v o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = fmaf(v[g * W + x][e] - mu[g][e], sc[e], sh[e]);
                v[g * W + x] = to_v4(elu4(to_f4(o), elu_accurate));
            }
        }
    };

    // phases of a block: every phase is one convolution; what feeds it and what happens to its result depends on the block type
    enum { PH_R1, PH_R2, PH_P1, PH_P2, PH_SC, PH_C1, PH_C2 };
    int phase_no = -1;
    (void)phase_no;
#pragma unroll 1
    for (int blk = 0; blk < p.n_blocks; ++blk) {
        const int type = p.type[blk];
        const bool has_sc = type == SBC_CHAIN_RES && p.w[blk][2] != nullptr;
        const int n_ph = has_sc ? 3 : 2;
        const int dil = p.dil[blk];
#pragma unroll 1
        for (int ph = 0; ph < n_ph; ++ph) {
            phase_no = min(phase_no + 1, 15);
            // phase kind and its filter: RCU (conv 1, conv 2), CRP (conv 1, conv 2), RES ([shortcut conv,] conv 1, conv 2)
            const int kind = type == SBC_CHAIN_RCU ? PH_R1 + ph : type == SBC_CHAIN_CRP ? PH_P1 + ph : (has_sc ? PH_SC + ph : PH_C1 + ph);
            const int wi = kind == PH_SC ? 2 : (kind == PH_R2 || kind == PH_P2 || kind == PH_C2) ? 1 : 0;
            const uint4* __restrict__ w = p.w[blk][wi];
            const float4 tr = f16x2_trailer(reinterpret_cast<const float4*>(w), 9 * (C / 16) * (C / 32) * 2);
            const float scale = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, tr.x)));
            const float descale = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, tr.y)));
            const bool elu_acc = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, tr.w)) != 0;
            const bool dx0 = W == 2 && dil > 1;                         // dilated at a width of two: only the dx = 0 taps touch the image

            // ---- filter ring: K step s = (tap, 32-channel slice kh); the first three steps are requested behind the operand work, in
            // front of the two barriers of the plane update
            // (ring depth: a K step is NU x 3 matrix instructions long -- with two or four units per wave (the half- and quarter-size
            // groups of small batches, 64 channels at 8 x 2) four steps in flight do not cover the L2 round trip of a filter fragment,
            // and those instantiations have the registers for more)
#ifndef SBC_CHAIN_WD_SMALL
#define SBC_CHAIN_WD_SMALL 8
#endif
#ifndef SBC_CHAIN_WD_MID
#define SBC_CHAIN_WD_MID 6
#endif
            constexpr int WD = NU <= 2 ? SBC_CHAIN_WD_SMALL : NU <= 4 ? SBC_CHAIN_WD_MID : 4;
            uint4 wr[WD][2];
            auto ldw = [&](int tap, int kh, int slot) {                 // compile-time constants at every call
#ifdef SBC_CHAIN_NO_WLOAD   // timing probe (tools/): every K step re-uses the first fragment's registers -- wrong results
                if (slot != 0) { wr[slot][0] = wr[0][0]; wr[slot][1] = wr[0][1]; return; }
#endif
                const int idx = wl_base + ((tap * (C / 16) + 2 * kh) * (C / 32) * 2) * 64;
                wr[slot][0] = w[idx];
                wr[slot][1] = w[idx + 64];
            };

            CH_T(0);
            // ---- the operand of this convolution, from registers
            f32x4v v[NU];
            if (kind == PH_R1 || kind == PH_R2) {
#pragma unroll
                for (int i = 0; i < NU; ++i) {
                    const f32x4v src = kind == PH_R1 ? xs[i] : vscale4(acc[i], prev_descale);
                    v[i] = to_v4(elu4(to_f4(src), elu_acc));
                }
            } else if (kind == PH_SC) {
                // the shortcut convolution of a RES block reads x as it is (layers.py:453-456)
#pragma unroll
                for (int i = 0; i < NU; ++i) v[i] = xs[i];
            } else if (kind == PH_C1 || kind == PH_C2) {
                if (kind == PH_C1) {
                    // normalize1 -> ELU of x; with a shortcut convolution its result (accumulators of the phase before, + bias) then
                    // takes x's place: x itself is not needed again
#pragma unroll
                    for (int i = 0; i < NU; ++i) v[i] = xs[i];
                    norm_elu(v, p.norm[blk][0], elu_acc);
                    if (has_sc) {
                        const f32x4v b3 = *reinterpret_cast<const f32x4v*>(p.bias[blk][2] + ch0);
#pragma unroll
                        for (int i = 0; i < NU; ++i)
#pragma unroll
                            for (int e = 0; e < 4; ++e) xs[i][e] = fmaf(acc[i][e], prev_descale, b3[e]);
                    }
                } else {
                    // t = conv1 + bias1; normalize2 -> ELU
                    const f32x4v b1 = *reinterpret_cast<const f32x4v*>(p.bias[blk][0] + ch0);
#pragma unroll
                    for (int i = 0; i < NU; ++i)
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[i][e] = fmaf(acc[i][e], prev_descale, b1[e]);
                    norm_elu(v, p.norm[blk][1], elu_acc);
                }
            } else {
                if (kind == PH_P1) {
                    // x = act(x) (layers.py:77): the activated tensor is both the running sum and the first pooling input.  ELU in
                    // its accurate form: it sits outside a convolution prologue here (common.h)
#pragma unroll
                    for (int i = 0; i < NU; ++i) { xs[i] = to_v4(elu4_acc(to_f4(xs[i]))); v[i] = xs[i]; }
                } else {
                    // p = conv_w1(...), x = p + x (layers.py:82), and p is the second pooling input
#pragma unroll
                    for (int i = 0; i < NU; ++i) { acc[i] = vscale4(acc[i], prev_descale); xs[i] = vadd4(acc[i], xs[i]); v[i] = acc[i]; }
                }
                // nn.MaxPool2d(5, 1, 2), separable: rows y - 2 .. y + 2 of the lane's own column (lanes n - 2 .. n + 2 of the same
                // sample; -inf outside, as PyTorch pads), then columns x - 2 .. x + 2: the same lane of the sample group's other units
                // (W = 8: the two rows on the other side of the half-sample boundary belong to the partner wave (uh ^ 1, same channel
                // block): both publish their two boundary rows in LDS first -- [wave][row][unit][kq] float4 -- one more barrier per pool)
                const bool lower = (uh & 1) == 0;
                if constexpr (W == 8) {
                    const int slot = lower ? n - 14 : n;                 // rows 14, 15 of the upper half / rows 0, 1 of the lower one
                    if ((unsigned)slot < 2u) {
#pragma unroll
                        for (int i = 0; i < NU; ++i) *reinterpret_cast<f32x4v*>(pscr + (((wave * 2 + slot) * NU + i) * 4 + kq) * 4) = v[i];
                    }
                    lds_barrier();
                }
#pragma unroll
                for (int i = 0; i < NU; ++i) {
                    const f32x4v m = v[i];
                    f32x4v r = m;
#pragma unroll
                    for (int d = -2; d <= 2; ++d) {
                        if (d == 0) continue;
                        const bool ok = (unsigned)(y + d) < (unsigned)H;
                        f32x4v o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = __shfl(m[e], lane + d);
                        if constexpr (W == 8) {
                            const int nd = n + d;
                            const int slot = lower ? nd - 16 : nd + 2;     // the partner's published row, when nd is outside this wave's 16
                            const f32x4v po = *reinterpret_cast<const f32x4v*>(pscr + ((((wave ^ NCB) * 2 + (slot & 1)) * NU + i) * 4 + kq) * 4);
                            if ((unsigned)nd >= 16u) o = po;
                        }
                        if (ok) r = vmax4(r, o);
                    }
                    v[i] = r;
                }
#pragma unroll
                for (int g = 0; g < NSG; ++g) {
                    f32x4v c[W];
#pragma unroll
                    for (int x = 0; x < W; ++x) c[x] = v[g * W + x];
#pragma unroll
                    for (int x = 0; x < W; ++x) {
                        f32x4v r = c[x];
#pragma unroll
                        for (int x2 = 0; x2 < W; ++x2)
                            if (x2 != x && x2 >= x - 2 && x2 <= x + 2) r = vmax4(r, c[x2]);
                        v[g * W + x] = r;
                    }
                }
            }
            CH_T(1);
            float ta = 0.f;
            uint2 vh[NU], vl[NU];
#ifdef SBC_CHAIN_NO_CONVERT   // timing probe: no ELU / pooling / norm / split (the values above are dead code then) -- wrong results
#pragma unroll
            for (int i = 0; i < NU; ++i) { vh[i] = make_uint2(blk, ph); vl[i] = make_uint2(ph, blk); }
#else
#pragma unroll
            for (int i = 0; i < NU; ++i) {
                StageScale ss{scale, ta};
                const float4 f = to_f4(v[i]);
                scale_track(f, &ss);
                ta = ss.amax;
                split_f16x2(f, scale, vh[i], vl[i]);
            }
#endif
            pair_range_tile(ta, scale, rbits, p.calib ? p.calib + 3 * blk + wi : nullptr);
            // ring position r of the K loop's order: (tap r / KH, slice r % KH); dilated at a width of two: (tap 3 (r / KH) + 1, slice r % KH)
            if (dx0) {
                static_for<0, WD - 1>([&](auto rc) {
                    constexpr int r = decltype(rc)::value;
                    if constexpr (r < 3 * KH) ldw(3 * (r / KH) + 1, r % KH, r);
                });
            } else {
                static_for<0, WD - 1>([&](auto rc) {
                    constexpr int r = decltype(rc)::value;
                    ldw(r / KH, r % KH, r);
                });
            }
            // every wave has left the previous K loop: the planes may be rewritten
            CH_T(2);
            lds_barrier();
            CH_T(3);
#pragma unroll
            for (int i = 0; i < NU; ++i) {
                unsigned char* dst = smem + wr_base + (((i / W) * SPU) * SP + (i % W) * CP) * 16;
                *reinterpret_cast<uint2*>(dst) = vh[i];
                *reinterpret_cast<uint2*>(dst + TERM) = vl[i];
            }
            lds_barrier();
            CH_T(4);

            // ---- K loop: D[16 couts][16 pixels] += W[16 couts][32 cin] X[32 cin][16 pixels] per (tap, slice), three fp16 products each.
            // Flat walk over micro-steps m = (K step, pair of units): the four X fragments of a micro-step are requested XD - 1
            // micro-steps ahead of its six matrix instructions through a ring of statically indexed registers (left to itself the
            // scheduler reads each fragment right in front of its first use: one LDS round trip per unit), and the two units' matrix
            // instructions alternate, so no instruction waits for the accumulator of the one before.
            // Lane base of tap row dy: row y + dy * dil of the lane's column, or the column's zero slot when that row is outside the
            // image (undilated: the slots above and below the column are the padding, no select needed).
            int rb[3];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const int yy = y + (r - 1) * dil;
                rb[r] = dil == 1 ? rd_base + r * 16 : ((unsigned)yy < (unsigned)H ? rd_base + (1 + (r - 1) * dil) * 16 : rd_base - y * 16);
            }
#ifndef SBC_CHAIN_XD
#define SBC_CHAIN_XD 2
#endif
            constexpr int XD = SBC_CHAIN_XD;                            // micro-steps of up to four units: 8 fragments each
            f16x8 xr[XD][8];
            auto kloop = [&](auto dx0c) {
                constexpr bool DX0 = decltype(dx0c)::value;
                constexpr int NS = (DX0 ? 3 : 9) * KH;
                constexpr int QD = (NU + 3) / 4;                         // micro-steps of a dilated K step (all units)
                constexpr int NM = DX0 ? 3 * KH * QD : micro_steps<W, NU, KH>();
                auto step_of = [](int m) constexpr -> MicroStep {
                    if constexpr (DX0) {                                 // K step s = (filter row, slice): tap 3 row + 1; units 4q .. 4q + 3
                        const int s_ = m / QD, q = m % QD;
                        return MicroStep{((3 * (s_ / KH) + 1) * KH + s_ % KH),
                                         {4 * q, 4 * q + 1 < NU ? 4 * q + 1 : -1, 4 * q + 2 < NU ? 4 * q + 2 : -1, 4 * q + 3 < NU ? 4 * q + 3 : -1}, q == 0};
                    } else {
                        return micro_step<W, NU, KH>(m);
                    }
                };
                // position of K step s in this loop's filter-ring order
                auto ring_of = [](int s_) constexpr -> int { return DX0 ? ((s_ / KH) / 3) * KH + s_ % KH : s_; };
                auto ldx = [&](int m) {
#ifdef SBC_CHAIN_NO_XLOAD   // timing probe: the operand ring is loaded once -- wrong results
                    if (m >= XD) return;
#endif
                    const MicroStep d = step_of(m);
                    const int tap = d.s / KH, kh = d.s % KH, r = tap / 3, dx = tap % 3 - 1;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        if (d.u[k] < 0) continue;
                        const int off = (4 * kh) * PS + (((d.u[k] / W) * SPU) * SP + (d.u[k] % W + dx) * CP) * 16;
                        xr[m % XD][2 * k] = *reinterpret_cast<const f16x8*>(smem + rb[r] + off);
                        xr[m % XD][2 * k + 1] = *reinterpret_cast<const f16x8*>(smem + rb[r] + off + TERM);
                    }
                };
#pragma unroll
                for (int m = 0; m < XD - 1; ++m) ldx(m);
                static_for<0, NM>([&](auto mc) {
                    constexpr int m = decltype(mc)::value;
                    constexpr MicroStep d = step_of(m);
                    constexpr int tap = d.s / KH, kh = d.s % KH;
                    constexpr int rp = ring_of(d.s);                     // this K step's position in the ring order
                    if constexpr (d.first_of_step && rp + WD - 1 < NS) {   // keep the filter ring full
                        constexpr int nx = rp + WD - 1;
                        constexpr int ntap = DX0 ? 3 * (nx / KH) + 1 : nx / KH;
                        ldw(ntap, nx % KH, nx % WD);
                    }
                    if constexpr (m + XD - 1 < NM) ldx(m + XD - 1);
                    const f16x8 wh = __builtin_bit_cast(f16x8, wr[rp % WD][0]);
                    const f16x8 wl = __builtin_bit_cast(f16x8, wr[rp % WD][1]);
                    // term-major over the micro-step's units: an accumulator's next matrix instruction is up to four instructions on.
                    // A unit's first one takes a literal zero addend: tap 0 (dx = -1) unless the unit is column 0, then tap 1
                    // (dilated: tap 1 for every unit)
                    static_for<0, 4>([&](auto kc) {
                        constexpr int k = decltype(kc)::value;
                        if constexpr (d.u[k] >= 0) {
                            constexpr bool first = kh == 0 && tap == ((d.u[k] % W) && !DX0 ? 0 : 1);
                            const f32x4v z = first ? f32x4v{0.f, 0.f, 0.f, 0.f} : acc[d.u[k]];
                            acc[d.u[k]] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xr[m % XD][2 * k + 1], z, 0, 0, 0);
                        }
                    });
                    static_for<0, 4>([&](auto kc) {
                        constexpr int k = decltype(kc)::value;
                        if constexpr (d.u[k] >= 0) acc[d.u[k]] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, xr[m % XD][2 * k], acc[d.u[k]], 0, 0, 0);
                    });
                    static_for<0, 4>([&](auto kc) {
                        constexpr int k = decltype(kc)::value;
                        if constexpr (d.u[k] >= 0) acc[d.u[k]] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xr[m % XD][2 * k], acc[d.u[k]], 0, 0, 0);
                    });
                    __builtin_amdgcn_sched_barrier(0);
                });
            };
#ifdef SBC_CHAIN_PRIO
            __builtin_amdgcn_s_setprio(SBC_CHAIN_PRIO);
#endif
            if (W == 2 && dx0) kloop(std::integral_constant<bool, W == 2>{});
            else kloop(std::false_type{});
#ifdef SBC_CHAIN_PRIO
            __builtin_amdgcn_s_setprio(0);
#endif
            CH_T(5);

            // ---- what the result is for
            if (kind == PH_R2 || kind == PH_P2) {
                // the running sum takes it (descale is a power of two: one rounding, as an add)
#pragma unroll
                for (int i = 0; i < NU; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e) xs[i][e] = fmaf(acc[i][e], descale, xs[i][e]);
            } else if (kind == PH_C2) {
                // shortcut + (conv2 + bias2)  (layers.py:456)
                const f32x4v b2 = *reinterpret_cast<const f32x4v*>(p.bias[blk][1] + ch0);
#pragma unroll
                for (int i = 0; i < NU; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e) xs[i][e] = xs[i][e] + fmaf(acc[i][e], descale, b2[e]);
            }
            prev_descale = descale;
            CH_T(6);
        }
    }
#pragma unroll
    for (int i = 0; i < NU; ++i) {
        const int s = s0 + w0 + (i / W) * SPU + sp;
        if (s < p.B) st_out(p.out + ((size_t)(s * H + y) * W + (i % W)) * C + ch0, to_f4(xs[i]));
    }
    if (rbits && lane == 0) atomicOr(p.range_flag, rbits);
}

template <int C, int W, int NW, int GD = 1>
static int launch_chain_t(const ChainParams& p, hipStream_t stream, bool dry) {
    constexpr int G = chain_samples(C, W, NW) / GD, SP = W * (W == 2 ? 12 : W == 4 ? 18 : 34);
    // operand planes + InstanceNorm++ scratch (+ at W = 8 the half-sample sums and the max pool's boundary rows)
    constexpr int LDS = 2 * (C / 8) * G * SP * 16 + (G * C + 2 * G) * 4 + (W == 8 ? 2 * NW * 16 * 4 + NW * 2 * 8 * 4 * 16 : 0);
    auto kern = conv_chain_kernel<C, W, NW, GD>;
    { const int rc = ensure_dyn_lds(reinterpret_cast<const void*>(kern), LDS); if (rc) return rc; }
    if (dry) return SBC_OK;
    hipLaunchKernelGGL(kern, dim3((p.B + G - 1) / G), dim3(64 * NW), LDS, stream, p);
    SBC_CHECK_HIP(hipGetLastError());
    return SBC_OK;
}

int launch_chain(const sbc_op& op, const sbc_chain& c, hipStream_t stream, bool dry) {
    SBC_REQUIRE(op.in && op.out && op.in != op.out, "chain: in / out must be set and distinct");
    SBC_REQUIRE(op.cin == op.cout && ((op.H == 8 && op.W == 2 && (op.cin == 64 || op.cin == 128)) || (op.H == 16 && op.W == 4 && op.cin == 64) ||
                                      (op.H == 32 && op.W == 8 && (op.cin == 32 || op.cin == 64))),
                "chain: 8 x 2 samples of 64 / 128 channels, 16 x 4 samples of 64, 32 x 8 samples of 32 / 64 (got %d x %d, %d -> %d)", op.H, op.W, op.cin, op.cout);
    SBC_REQUIRE((op.flags & SBC_CONV_F16X2) && !(op.flags & SBC_CONV_F16W), "chain: SBC_CONV_F16X2 only (the weight forms it reads)");
    SBC_REQUIRE(c.n_blocks >= 1 && c.n_blocks <= SBC_CHAIN_MAX_BLOCKS, "chain: %d blocks (1 .. %d)", c.n_blocks, SBC_CHAIN_MAX_BLOCKS);
    SBC_REQUIRE(op.B > 0 && (long)op.B * op.H * op.W * op.cin <= 0x7fffffffL, "chain: bad batch %d", op.B);
    ChainParams p;
    memset(&p, 0, sizeof(p));
    p.in = (const float*)op.in;
    p.out = (float*)op.out;
    p.n_blocks = c.n_blocks;
    for (int b = 0; b < c.n_blocks; ++b) {
        SBC_REQUIRE(c.w1[b] && c.w2[b] && c.type[b] >= SBC_CHAIN_RCU && c.type[b] <= SBC_CHAIN_RES, "chain: block %d: weights / type", b);
        p.w[b][0] = (const uint4*)c.w1[b];
        p.w[b][1] = (const uint4*)c.w2[b];
        p.type[b] = c.type[b];
        p.dil[b] = 1;
        if (c.type[b] == SBC_CHAIN_RES) {
            const int d = c.dil[b] > 1 ? c.dil[b] : 1;
            SBC_REQUIRE(c.bias1[b] && c.bias2[b] && c.norm1[b] && c.norm2[b] && (!c.w3[b] || c.bias3[b]), "chain: RES block %d: biases / norms", b);
            SBC_REQUIRE(d == 1 || (op.W == 2 && (d == 2 || d == 4)), "chain: RES block %d: dilation %d (1, or 2 / 4 at a width of two)", b, d);
            p.dil[b] = d;
            p.w[b][2] = (const uint4*)c.w3[b];
            p.bias[b][0] = c.bias1[b]; p.bias[b][1] = c.bias2[b]; p.bias[b][2] = c.bias3[b];
            p.norm[b][0] = c.norm1[b]; p.norm[b][1] = c.norm2[b];
        }
    }
    p.calib = (float*)op.calib;
    p.B = op.B;
#ifdef SBC_CHAIN_TIMELINE
    p.dbg = (unsigned long long*)op.aux;
#endif
    unsigned* flag = nullptr;
    { const int rc = range_flag_ptr(&flag); if (rc) return rc; }
    p.range_flag = flag;
    // Four-wave workgroups where a 16-output-channel block per wave allows it (64 channels at 8 x 2 / 16 x 4, 32 channels at 32 x 8):
    // two of them share a CU and drift apart, so one converts operands while the other multiplies (A/B aid: SBC_CHAIN_NW8)
    static const bool nw8 = getenv("SBC_CHAIN_NW8") != nullptr;
    if (op.W == 8) return op.cin == 64 ? launch_chain_t<64, 8, 8>(p, stream, dry) : nw8 ? launch_chain_t<32, 8, 8>(p, stream, dry) : launch_chain_t<32, 8, 4>(p, stream, dry);
    if (op.W == 4) return nw8 ? launch_chain_t<64, 4, 8>(p, stream, dry) : launch_chain_t<64, 4, 4>(p, stream, dry);
    // 8 x 2 samples, small batches: half groups when the full groups would occupy at most half of the CUs this launch may count on
    // (the plan's persistent-grid width: all CUs, or half of them when two sub-batch streams share the chip).  (A/B aid: SBC_CHAIN_GD=1 / 2)
    int dev = 0, cus = 256;
    SBC_CHECK_HIP(hipGetDevice(&dev));
    SBC_CHECK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    static const int force_gd = getenv("SBC_CHAIN_GD") ? atoi(getenv("SBC_CHAIN_GD")) : 0;
    const int avail = persistent_cus(cus);
    if (op.cin == 128) {
        const bool half = force_gd ? force_gd == 2 : 2 * ((op.B + 7) / 8) <= avail;
        // quarter groups while they occupy at most half of the CUs (213 samples: 107 workgroups, -2 % per step; 425 samples: 213 workgroups, +3 %: half groups)
        const bool quarter = force_gd ? force_gd == 4 : (op.B + 1) / 2 <= avail / 2;
        if (dry) {
            int rc = launch_chain_t<128, 2, 8, 4>(p, stream, true);
            if (!rc) rc = launch_chain_t<128, 2, 8, 2>(p, stream, true);
            return rc ? rc : launch_chain_t<128, 2, 8>(p, stream, true);
        }
        return quarter ? launch_chain_t<128, 2, 8, 4>(p, stream, dry) : half ? launch_chain_t<128, 2, 8, 2>(p, stream, dry) : launch_chain_t<128, 2, 8>(p, stream, dry);
    }
    if (nw8) return launch_chain_t<64, 2, 8>(p, stream, dry);
    const bool half = force_gd ? force_gd == 2 : 2 * ((op.B + 3) / 4) <= 2 * avail;      // (two 4-wave workgroups per CU)
    if (dry) { const int rc = launch_chain_t<64, 2, 4, 2>(p, stream, true); if (rc) return rc; return launch_chain_t<64, 2, 4>(p, stream, true); }
    return half ? launch_chain_t<64, 2, 4, 2>(p, stream, dry) : launch_chain_t<64, 2, 4>(p, stream, dry);
}

}  // namespace sbc

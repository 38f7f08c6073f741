// Input-tile staging shared by the MFMA convolution and the end-conv kernel.
//
// A workgroup owns TM consecutive output pixels of the flattened (n, h, w) space of an NHWC tensor.  TM is
// either a whole number of image rows inside one sample (TM < H*W, TM % W == 0) or a whole number of samples
// (TM % (H*W) == 0).  The rows a 3x3 (dilated) stencil needs are copied to LDS once -- transformed on the way
// (InstanceNorm++ affine, ELU) -- as [pixel][CIN + 4] floats: the +4 pad makes the pixel stride an odd number
// of 16-byte slots, so the ds_read_b128 A-fragment reads (16 lanes = 16 consecutive pixels, same channel
// offset) hit 16 distinct slots of the 64-bank row.  One extra all-zero pixel at index `nps` stands in for
// every out-of-image tap (zero padding of nn.Conv2d), so no border is materialised.
#pragma once
#include "common.h"

namespace sbc {

// 16-byte accesses for activations.  Rounds 1-3 made all of them non-temporal ("touched once per kernel").  Round 4 measured it
// (DESIGN.md section 13.6): a tensor is read by the NEXT launch, usually out of L2 / the 256 MB MALL, so
//   * loads are plain (cached) loads now: sustained two-stream step 5.05 -> 5.00 ms (-DSBC_NT_LD restores the hint);
//   * stores of the kernels that write whole 128-byte lines per thread group (Winograd, statistics, ...) stay non-temporal (plain:
//     5.05 -> 5.05, and with the loads plain as well 5.32-5.37 against 5.21 in the first A/B);
//   * stores of the direct kernels, which write lines in two pieces, are plain: st_out below.
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld_stream(const float* p) {
#ifndef SBC_NT_LD
    return *reinterpret_cast<const float4*>(p);
#else
    const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
#endif
}
// Output store of the direct kernels (conv_pair, conv_pool, conv_dp, conv_res): a wave writes 64 of a pixel's 128 or 256 bytes, the
// wave with the other output channels the rest a little later.  As plain (cached) stores the pieces meet in L2 and reach memory as
// whole lines; as non-temporal stores they did not: WRITE_SIZE 308 MB for a 223 MB tensor (pair), 279 (CRP stage), 276 (ResidualBlock),
// 144 for 111 (64 -> 64) -- 218 / 218 / 229 / 109 MB now, and the sustained two-stream step 5.42 -> 5.21 ms.  (The Winograd kernels
// write whole lines per thread group and keep st_stream: with plain stores everywhere the step was 5.32-5.37.)
__device__ __forceinline__ void st_out(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ void st_stream(float* p, float4 v) {
#if defined(SBC_NO_STREAM) || defined(SBC_PLAIN_ST)
    *reinterpret_cast<float4*>(p) = v;
#else
    f32x4 t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
    __builtin_nontemporal_store(t, reinterpret_cast<f32x4*>(p));
#endif
}

// conv_mode f16x2 (SBC_CONV_F16X2): activations enter the matrix cores as x * act_scale split into two fp16 terms.  act_scale is a
// power of two per LAYER, the first float of the packed weight's trailer: 1 as packed, set by sbc_f16x2_calibrate (api.hip) so that
// the layer's calibration-time max |x| lands in [2^8, 2^9) -- fp16 then holds the low term l = fp16(x s - h) as a NORMAL number
// for every |x s| >= 2^-3 (relative error 2^-22 over 12 binades below the calibrated maximum and 5 above it), where the unscaled
// split of rounds 2-3 degraded to an absolute 2^-25 as soon as |x| < 0.125.  The multiplication rides in the split itself
// (split_f16x2: fma(x, s, 0) / fma(x, s, -h) on the mixed-precision FMA), so it costs one instruction per PAIR of values.
// `amax` collects max |x| (unscaled) of what a thread stages.  Both ends are guarded through the device's range-flag word
// (sbc_range_flag): bit 0 when amax * act_scale >= F16X2_LIMIT (the high term or, in the Winograd kernel, a 4-term transform sum
// could overflow fp16), bit 1 when a whole wave's share of a tile is non-zero but stays below F16X2_SMALL after scaling (the low
// terms of that region are fp16 denormals: precision is no longer fp32-class).  The host re-runs such a batch in bf16x3.
struct StageScale { float scale; float amax; };
constexpr float F16X2_LIMIT = 16000.f;                   // 65504 / 4, rounded down
constexpr float F16X2_SMALL = 0.015625f;                 // 2^-6: below it the relative error of the split exceeds 2^-19
// fp32-tile staging (conv_wx3): track, then multiply by act_scale (an exact power of two)
__device__ __forceinline__ void scale_stage(float4& x, StageScale* ss) {
    if (!ss) return;
    ss->amax = __builtin_fmaxf(__builtin_fmaxf(ss->amax, __builtin_fabsf(x.x)), __builtin_fabsf(x.y));
    ss->amax = __builtin_fmaxf(__builtin_fmaxf(ss->amax, __builtin_fabsf(x.z)), __builtin_fabsf(x.w));
    x.x *= ss->scale; x.y *= ss->scale; x.z *= ss->scale; x.w *= ss->scale;
}
__device__ __forceinline__ void scale_track(const float4& x, StageScale* ss) {
    if (!ss) return;
    // (max(max(a, |x|), |y|): the shape hipcc turns into one v_max3_f32 with |.| source modifiers)
    ss->amax = __builtin_fmaxf(__builtin_fmaxf(ss->amax, __builtin_fabsf(x.x)), __builtin_fabsf(x.y));
    ss->amax = __builtin_fmaxf(__builtin_fmaxf(ss->amax, __builtin_fabsf(x.z)), __builtin_fabsf(x.w));
}
// range check of what this wave staged (all lanes call; `amax` = the lane's max |x|, unscaled): returns the bits to OR into the
// device's range-flag word, 0 in the normal case
__device__ __forceinline__ unsigned f16x2_range_bits(float amax, float scale) {
    const float am = amax * scale;
    unsigned bits = 0;
    if (__builtin_amdgcn_ballot_w64(am >= F16X2_LIMIT)) bits |= 1u;
    if (!__builtin_amdgcn_ballot_w64(am >= F16X2_SMALL) && __builtin_amdgcn_ballot_w64(amax > 0.f)) bits |= 2u;
    return bits;
}
// ... and the store: one lane per wave raises the flag; in calibration launches (calib != NULL, sbc_f16x2_calibrate) the wave also
// folds its max |x| into the layer's slot (non-negative floats order like their bit patterns)
__device__ __forceinline__ void f16x2_range_report(float amax, float scale, unsigned* __restrict__ flag, float* __restrict__ calib) {
    const unsigned bits = f16x2_range_bits(amax, scale);
    if (bits && (threadIdx.x & 63) == 0) atomicOr(flag, bits);
    if (calib) {
        float m = amax;
        for (int o = 32; o > 0; o >>= 1) m = __builtin_fmaxf(m, __shfl_xor(m, o));
        if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(reinterpret_cast<unsigned*>(calib), __float_as_uint(m));
    }
}

// Image dimensions with the divisions the kernels need.  P2 = true: H and W are powers of two (every shape the
// score network produces for Nt, Nr in {16, 64, 256}), so / and % are shifts and masks -- on gfx950 fp32 MFMA
// shares the vector ALU with every other vector instruction, and a runtime integer division is ~25 of them.
template <bool P2>
struct Dims {
    int H, W, HW, hsh, wsh;
    __device__ __forceinline__ int div_w(int x) const { return P2 ? x >> wsh : x / W; }
    __device__ __forceinline__ int mod_w(int x) const { return P2 ? x & (W - 1) : x % W; }
    __device__ __forceinline__ int div_h(int x) const { return P2 ? x >> hsh : x / H; }
    __device__ __forceinline__ int mod_h(int x) const { return P2 ? x & (H - 1) : x % H; }
    __device__ __forceinline__ int div_hw(int x) const { return P2 ? x >> (hsh + wsh) : x / HW; }
};

// Workgroup ids are dealt round-robin over the 8 XCDs of the chip (each with its own L2).  Mapping id -> tile so that
// every XCD walks a contiguous run of tiles lets the halo rows two neighbouring tiles share come through the same L2.
__device__ __forceinline__ int xcd_tile(int id, int n) {
    if (n < 64 || (n & 7)) return id;
    return (id & 7) * (n >> 3) + (id >> 3);
}

struct TileGeom {
    int p0;        // first output pixel (flattened n*H*W + h*W + w)
    int rs0;       // first staged global row (n*H + h)
    int nps;       // staged pixels; LDS pixel index nps is the zero pixel
    int n_first;   // sample index of the first staged row
    int multi;     // tile spans whole samples (no halo)
};

template <bool P2>
__device__ __forceinline__ TileGeom tile_geom(int tile, int TM, int B, const Dims<P2>& d, int halo_rows) {
    TileGeom g;
    g.p0 = tile * TM;
    const int r0 = d.div_w(g.p0);
    int r1 = r0 + d.div_w(TM);
    if (r1 > B * d.H) r1 = B * d.H;
    g.multi = TM >= d.HW;
    int rs1;
    if (g.multi) {
        g.rs0 = r0;
        rs1 = r1;
    } else {
        const int n = d.div_h(r0);
        g.rs0 = max(r0 - halo_rows, n * d.H);
        rs1 = min(r1 + halo_rows, (n + 1) * d.H);
    }
    g.nps = (rs1 - g.rs0) * d.W;
    g.n_first = d.div_h(g.rs0);
    return g;
}

// Staging is split in two so that other requests (statistics, prefetches) can be issued between the loads and their use:
//   stage_issue   requests up to NPF 16-byte chunks per thread of the staged rows (raw values, registers only);
//   stage_commit  applies the prologue (stats = [B][3][CIN]: mu, scale, shift; ELU) and writes the LDS tile; chunks
//                 beyond NPF * NTHREADS (unusually wide halos) are fetched synchronously here.
template <int CIN, int NTHREADS, int NPF>
__device__ __forceinline__ void stage_issue(float4 (&pf)[NPF], const float* __restrict__ in, const TileGeom& g,
                                            int W, int tid) {
    const float* src = in + (size_t)g.rs0 * W * CIN;
    const int total = g.nps * (CIN / 4);
#pragma unroll
    for (int u = 0; u < NPF; ++u) {
        const int idx = u * NTHREADS + tid;
        if (idx < total) pf[u] = ld_stream(src + (size_t)idx * 4);
    }
}

// `stats` is indexed by sample: global [B][3][CIN] with n0 = first sample of the staged rows, or a copy of the tile's
// samples in LDS with n0 = 0 (stage_stats_to_lds) -- the latter keeps three dependent L2 round trips per 16-byte chunk
// out of the staging loop.
template <int CIN, bool P2>
__device__ __forceinline__ void stage_put(float* lds, float4 x, int idx, const float* __restrict__ stats, int flags,
                                          const TileGeom& g, const Dims<P2>& d, int n0 = -1, StageScale* ss = nullptr) {
    constexpr int S = CIN + 4;
    constexpr int C4 = CIN / 4;
    const int pix = idx / C4, c4 = idx % C4;
    if (flags & SBC_PRO_NORM) {
        const int n = (n0 < 0 ? g.n_first : n0) + (g.multi ? d.div_hw(pix) : 0);
        const float* st = stats + (size_t)n * 3 * CIN + c4 * 4;
        const float4 mu = *reinterpret_cast<const float4*>(st);
        const float4 sc = *reinterpret_cast<const float4*>(st + CIN);
        const float4 sh = *reinterpret_cast<const float4*>(st + 2 * CIN);
        x.x = (x.x - mu.x) * sc.x + sh.x;
        x.y = (x.y - mu.y) * sc.y + sh.y;
        x.z = (x.z - mu.z) * sc.z + sh.z;
        x.w = (x.w - mu.w) * sc.w + sh.w;
    }
    if (flags & SBC_PRO_ELU) x = elu4(x, (flags & SBC_PRO_ELU_ACC) != 0);
    scale_stage(x, ss);
    *reinterpret_cast<float4*>(lds + pix * S + c4 * 4) = x;
}

// copy the InstanceNorm++ statistics of the samples a tile touches to LDS ([sample][3][CIN]); the caller puts a
// barrier between this and stage_commit(..., n0 = 0)
template <int CIN, int NTHREADS, bool P2>
__device__ __forceinline__ void stage_stats_to_lds(float* st_lds, const float* __restrict__ stats, const TileGeom& g,
                                                   const Dims<P2>& d, int tid) {
    const int nsamp = g.multi ? d.div_hw(g.nps) : 1;
    const float4* src = reinterpret_cast<const float4*>(stats + (size_t)g.n_first * 3 * CIN);
    for (int i = tid; i < nsamp * 3 * CIN / 4; i += NTHREADS) reinterpret_cast<float4*>(st_lds)[i] = src[i];
}

template <int CIN, int NTHREADS, int NPF, bool P2>
__device__ __forceinline__ void stage_commit(float* lds, const float4 (&pf)[NPF], const float* __restrict__ in,
                                             const float* __restrict__ stats, int flags, const TileGeom& g,
                                             const Dims<P2>& d, int tid, int n0 = -1, StageScale* ss = nullptr) {
    constexpr int S = CIN + 4;
    const int W = d.W;
    const int total = g.nps * (CIN / 4);
#pragma unroll
    for (int u = 0; u < NPF; ++u) {
        const int idx = u * NTHREADS + tid;
        if (idx < total) stage_put<CIN, P2>(lds, pf[u], idx, stats, flags, g, d, n0, ss);
    }
    const float* src = in + (size_t)g.rs0 * W * CIN;
    for (int idx = NPF * NTHREADS + tid; idx < total; idx += NTHREADS)
        stage_put<CIN, P2>(lds, ld_stream(src + (size_t)idx * 4), idx, stats, flags, g, d, n0, ss);
    for (int i = tid; i < S; i += NTHREADS) lds[g.nps * S + i] = 0.f;
}

// SBC_PRO_NORM_SELF: the InstanceNorm++ statistics of the tile's samples computed by the consumer itself, for tiles that hold
// whole samples (the 16x4 and 8x2 levels of the score network: 64 / 16 pixels per sample).  A statistics launch there reads a
// few MB and costs 9-13 us of pure launch latency on the critical path, 14 times per network evaluation; here the workgroup
// reads its own samples once more (they are on their way to L2 anyway: stage_issue has requested them) and spends ~1 us.
//   phase A: L adjacent lanes per (sample, channel quad) -- two passes over the sample's pixels (mean, then sum (x - mean)^2),
//            lanes combined by shuffles;   phase B: one wave per sample: mean and unbiased variance of the channel means (the
//            "++" term);   phase C: (mu, scale, shift) as ops.hip: inorm_stats_kernel defines them.
// Every sum has a fixed order that depends only on (H*W, CIN): results are independent of the batch, of the kernel variant a
// batch size selects, and reproducible.  st_lds: [samples][3][CIN] floats followed by 2 floats per sample of scratch.  All threads of the workgroup
// call; three barriers, the last one behind the finished table.  `agb` = [3][CIN] (alpha | gamma | beta).
template <int CIN, int NTHREADS, int TM, bool P2>   // (TM: unused, kept for the call sites)
__device__ __forceinline__ void self_stats_to_lds(float* st_lds, const float* __restrict__ in, const float* __restrict__ agb,
                                                  const TileGeom& g, const Dims<P2>& d, int tid) {
    constexpr int C4 = CIN / 4;
    const int HW = d.HW;
    const int ns = d.div_hw(g.nps);
    const int pairs = ns * C4;
    // lanes per (sample, quad): a function of the image size ONLY -- the kernel variant (threads per workgroup, pixels per tile)
    // depends on the batch size, and a sample's sums must not (two sub-batch streams against one: bit-identical results)
    const int L = HW < 8 ? HW : 8;
    const int l = tid & (L - 1), pidx = tid / L;
    const float* base = in + (size_t)g.rs0 * d.W * CIN;
    const float inv_hw = 1.f / (float)HW;
    float* tmp = st_lds + (size_t)ns * 3 * CIN;
    for (int pr0 = 0; pr0 < pairs; pr0 += NTHREADS / L) {
        const int pr = pr0 + pidx;
        const bool act = pr < pairs;
        const int s = act ? pr / C4 : 0, c4 = act ? pr % C4 : 0;
        const float* q = base + ((size_t)s * HW + l) * CIN + c4 * 4;
        float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int px = l; px < HW; px += L) {
            const float4 v = *reinterpret_cast<const float4*>(q + (size_t)(px - l) * CIN);
            sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
        }
        for (int m = 1; m < L; m <<= 1) {
            sum.x += __shfl_xor(sum.x, m); sum.y += __shfl_xor(sum.y, m);
            sum.z += __shfl_xor(sum.z, m); sum.w += __shfl_xor(sum.w, m);
        }
        const float4 mean = make_float4(sum.x * inv_hw, sum.y * inv_hw, sum.z * inv_hw, sum.w * inv_hw);
        float4 m2 = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int px = l; px < HW; px += L) {
            const float4 v = *reinterpret_cast<const float4*>(q + (size_t)(px - l) * CIN);
            float t;
            t = v.x - mean.x; m2.x = fmaf(t, t, m2.x); t = v.y - mean.y; m2.y = fmaf(t, t, m2.y);
            t = v.z - mean.z; m2.z = fmaf(t, t, m2.z); t = v.w - mean.w; m2.w = fmaf(t, t, m2.w);
        }
        for (int m = 1; m < L; m <<= 1) {
            m2.x += __shfl_xor(m2.x, m); m2.y += __shfl_xor(m2.y, m);
            m2.z += __shfl_xor(m2.z, m); m2.w += __shfl_xor(m2.w, m);
        }
        if (act && l == 0) {
            *reinterpret_cast<float4*>(st_lds + ((size_t)s * 3 + 0) * CIN + c4 * 4) = mean;
            *reinterpret_cast<float4*>(st_lds + ((size_t)s * 3 + 1) * CIN + c4 * 4) =
                make_float4(m2.x * inv_hw, m2.y * inv_hw, m2.z * inv_hw, m2.w * inv_hw);
        }
    }
    __syncthreads();
    {
        const int lane = tid & 63;
        for (int s = tid >> 6; s < ns; s += NTHREADS / 64) {
            const float* mu = st_lds + (size_t)s * 3 * CIN;
            float a = 0.f;
            for (int c = lane; c < CIN; c += 64) a += mu[c];
            for (int m = 1; m < 64; m <<= 1) a += __shfl_xor(a, m);
            const float mm = a * (1.f / (float)CIN);
            float b = 0.f;
            for (int c = lane; c < CIN; c += 64) { const float t = mu[c] - mm; b = fmaf(t, t, b); }
            for (int m = 1; m < 64; m <<= 1) b += __shfl_xor(b, m);
            if (lane == 0) { tmp[2 * s] = mm; tmp[2 * s + 1] = b * (1.f / (float)(CIN - 1)); }
        }
    }
    __syncthreads();
    for (int i = tid; i < ns * CIN; i += NTHREADS) {
        const int s = i / CIN, c = i % CIN;
        float* o = st_lds + (size_t)s * 3 * CIN;
        const float mhat = (o[c] - tmp[2 * s]) / sqrtf(tmp[2 * s + 1] + 1e-5f);
        const float rstd = 1.f / sqrtf(fmaxf(o[CIN + c], 0.f) + 1e-5f);
        const float alpha = agb[c], gamma = agb[CIN + c], beta = agb[2 * CIN + c];
        o[CIN + c] = gamma * rstd;
        o[2 * CIN + c] = fmaf(gamma, mhat * alpha, beta);
    }
    __syncthreads();
}

// Statistics in registers: when a tile lies inside ONE sample (TM <= H*W) and NTHREADS is a multiple of CIN / 4, every
// 16-byte chunk a thread stages belongs to the same channel quad of the same sample, so its (mu, scale, shift) are three
// float4 loads per THREAD -- issued together with the tile loads -- and the copy of the statistics through LDS with its
// workgroup barrier disappears from the prologue.
struct RegStats { float4 mu, sc, sh; };
template <int CIN, int NTHREADS>
__device__ __forceinline__ RegStats load_reg_stats(const float* __restrict__ stats, const TileGeom& g, int tid) {
    static_assert(NTHREADS % (CIN / 4) == 0, "a thread must keep its channel quad over all its chunks");
    const float* st = stats + (size_t)g.n_first * 3 * CIN + (tid % (CIN / 4)) * 4;
    RegStats r;
    r.mu = *reinterpret_cast<const float4*>(st);
    r.sc = *reinterpret_cast<const float4*>(st + CIN);
    r.sh = *reinterpret_cast<const float4*>(st + 2 * CIN);
    return r;
}
template <int CIN, int NTHREADS, int NPF>
__device__ __forceinline__ void stage_commit_reg(float* lds, const float4 (&pf)[NPF], const float* __restrict__ in,
                                                 const RegStats& rs, int flags, const TileGeom& g, int W, int tid,
                                                 StageScale* ss = nullptr) {
    constexpr int S = CIN + 4, C4 = CIN / 4;
    const int total = g.nps * C4;
    auto put = [&](float4 x, int idx) {
        x.x = (x.x - rs.mu.x) * rs.sc.x + rs.sh.x; x.y = (x.y - rs.mu.y) * rs.sc.y + rs.sh.y;
        x.z = (x.z - rs.mu.z) * rs.sc.z + rs.sh.z; x.w = (x.w - rs.mu.w) * rs.sc.w + rs.sh.w;
        if (flags & SBC_PRO_ELU) x = elu4(x, (flags & SBC_PRO_ELU_ACC) != 0);
        scale_stage(x, ss);
        *reinterpret_cast<float4*>(lds + (idx / C4) * S + (idx % C4) * 4) = x;
    };
#pragma unroll
    for (int u = 0; u < NPF; ++u) {
        const int idx = u * NTHREADS + tid;
        if (idx < total) put(pf[u], idx);
    }
    const float* src = in + (size_t)g.rs0 * W * CIN;
    for (int idx = NPF * NTHREADS + tid; idx < total; idx += NTHREADS) put(ld_stream(src + (size_t)idx * 4), idx);
    for (int i = tid; i < S; i += NTHREADS) lds[g.nps * S + i] = 0.f;
}

template <int CIN, int NTHREADS, int NPF, bool P2>
__device__ __forceinline__ void stage_tile(float* lds, const float* __restrict__ in, const float* __restrict__ stats,
                                           int flags, const TileGeom& g, const Dims<P2>& d, int tid) {
    float4 pf[NPF];
    stage_issue<CIN, NTHREADS, NPF>(pf, in, g, d.W, tid);
    stage_commit<CIN, NTHREADS, NPF, P2>(lds, pf, in, stats, flags, g, d, tid);
}


// ---- split-bf16 staging (conv_x3.hip) ---------------------------------------------------------------------------
// Every fp32 activation x is written as three bf16 terms h + m + l = x (h = bf16(x), m = bf16(x - h), l = x - h - m:
// 8 + 8 + 8 significand bits, the sum is exact), one LDS plane per term, each [pixel][CIN + 8] bf16.  The 16-byte pad
// makes the pixel stride an odd number of 16-byte slots (conflict-free ds_read_b128 over 16 consecutive pixels).
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split3(float4 x, bf16x4& h, bf16x4& m, bf16x4& l) {
    h[0] = (__bf16)x.x; h[1] = (__bf16)x.y; h[2] = (__bf16)x.z; h[3] = (__bf16)x.w;
    x.x -= (float)h[0]; x.y -= (float)h[1]; x.z -= (float)h[2]; x.w -= (float)h[3];
    m[0] = (__bf16)x.x; m[1] = (__bf16)x.y; m[2] = (__bf16)x.z; m[3] = (__bf16)x.w;
    x.x -= (float)m[0]; x.y -= (float)m[1]; x.z -= (float)m[2]; x.w -= (float)m[3];
    l[0] = (__bf16)x.x; l[1] = (__bf16)x.y; l[2] = (__bf16)x.z; l[3] = (__bf16)x.w;
}

// Single-term fp16 staging (conv_mode f16w, BASELINE config 5: fp16 score-network weights): the activation is rounded
// once to fp16 (round to nearest even, v_cvt_f16_f32) and only the first plane exists.
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// Two-term fp16 split (conv_mode f16x2) of a * s (s = the layer's act_scale, a power of two): a s = h + l + O(2^-22 |a s|),
// h = fp16(a s) (round to nearest even), l = fp16(a s - h); the product a s and the difference a s - h are exact in fp32, so each
// term is ONE mixed-precision FMA rounded once to fp16: v_fma_mixlo/mixhi_f16 h = fma(a, s, 0), l = fma(a, s, -h) with the third
// operand read from the fp16 half just written.  Four vector instructions per pair of values.
// h, l: the two values' terms packed low | high, as the matrix instructions take them.
// HAZARD: hipcc does not look inside inline assembly, so it inserts none of the wait states a vector-ALU write needs before a
// MATRIX instruction reads the register.  Where split terms feed v_mfma_* directly from registers (conv_wx3.hip, conv_wp.hip) the
// caller puts split_f16x2_settle() between the last split and the first matrix instruction: without it one instantiation (128 ->
// 64, two output blocks per phase), in which nothing else happened to sit between the two, multiplied a stale high half -- results
// off by 3e-4 ... 0.5, correct at -O1 and with the two wait states.  Terms that go through LDS first (conv_x3, conv_pair) are safe.
__device__ __forceinline__ void split_f16x2_settle(uint4& h, uint4& l) {
    asm volatile("s_nop 1" : "+v"(h.x), "+v"(h.y), "+v"(h.z), "+v"(h.w), "+v"(l.x), "+v"(l.y), "+v"(l.z), "+v"(l.w));
}
// Four values at a time: the two high terms are complete before the first low term reads them, so no instruction reads a
// register in the slot right behind a 16-bit (partial) write of it.
__device__ __forceinline__ void split_f16x2(float4 x, float s, uint2& h, uint2& l) {
#ifdef SBC_F16X2_UNSCALED   // timing aid (tools/): the three-instruction split of round 3, right only while every act_scale is 1
    asm("v_cvt_pk_f16_f32 %0, %2, %3\n\tv_cvt_pk_f16_f32 %1, %4, %5" : "=&v"(h.x), "=&v"(h.y) : "v"(x.x), "v"(x.y), "v"(x.z), "v"(x.w));
    asm("v_fma_mixlo_f16 %0, %2, 1.0, -%4 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixlo_f16 %1, %3, 1.0, -%5 op_sel:[0,0,0] op_sel_hi:[0,0,1]"
        : "=&v"(l.x), "=&v"(l.y) : "v"(x.x), "v"(x.z), "v"(h.x), "v"(h.y));
    asm("v_fma_mixhi_f16 %0, %2, 1.0, -%4 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %1, %3, 1.0, -%5 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
        : "+v"(l.x), "+v"(l.y) : "v"(x.y), "v"(x.w), "v"(h.x), "v"(h.y));
    (void)s;
    return;
#endif
    asm("v_fma_mixlo_f16 %0, %4, %8, 0 op_sel_hi:[0,0,0]\n\t"
        "v_fma_mixlo_f16 %1, %6, %8, 0 op_sel_hi:[0,0,0]\n\t"
        "v_fma_mixhi_f16 %0, %5, %8, 0 op_sel_hi:[0,0,0]\n\t"
        "v_fma_mixhi_f16 %1, %7, %8, 0 op_sel_hi:[0,0,0]\n\t"
        "v_fma_mixlo_f16 %2, %4, %8, -%0 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixlo_f16 %3, %6, %8, -%1 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %2, %5, %8, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %3, %7, %8, -%1 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
        : "=&v"(h.x), "=&v"(h.y), "=&v"(l.x), "=&v"(l.y) : "v"(x.x), "v"(x.y), "v"(x.z), "v"(x.w), "s"(s));
}

// ... and of four values that are ALREADY scaled (a = h + l): three instructions per pair.  The Winograd kernel stages its fp32
// tile as x * act_scale (one multiply per staged value, scale_stage below) and splits the transformed values with this form: a
// staged value feeds four transformed values, so the multiply at staging is half the price of the scaling split in the K loop.
__device__ __forceinline__ void split_f16x2_unit(float4 x, uint2& h, uint2& l) {
    asm("v_cvt_pk_f16_f32 %0, %4, %5\n\t"
        "v_cvt_pk_f16_f32 %1, %6, %7\n\t"
        "v_fma_mixlo_f16 %2, %4, 1.0, -%0 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixlo_f16 %3, %6, 1.0, -%1 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %2, %5, 1.0, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %3, %7, 1.0, -%1 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
        : "=&v"(h.x), "=&v"(h.y), "=&v"(l.x), "=&v"(l.y) : "v"(x.x), "v"(x.y), "v"(x.z), "v"(x.w));
}

// lds16: LDS viewed as 16-bit elements; `plane` = elements per plane; TERMS = 3 (exact bf16 split), 2 (fp16 pair of a scaled
// value, conv_mode f16x2) or 1 (fp16)
template <int CIN, bool P2, int TERMS = 3>
__device__ __forceinline__ void stage_put_split(unsigned short* lds16, int plane, float4 x, int idx,
                                                const float* __restrict__ stats, int flags, const TileGeom& g,
                                                const Dims<P2>& d, StageScale* ss = nullptr, int n0 = -1) {
    constexpr int SH = CIN + 8;
    constexpr int C4 = CIN / 4;
    const int pix = idx / C4, c4 = idx % C4;
    if (flags & SBC_PRO_NORM) {
        const int n = (n0 < 0 ? g.n_first : n0) + (g.multi ? d.div_hw(pix) : 0);   // n0 = 0: `stats` is the tile's own table
        const float* st = stats + (size_t)n * 3 * CIN + c4 * 4;
        const float4 mu = *reinterpret_cast<const float4*>(st);
        const float4 sc = *reinterpret_cast<const float4*>(st + CIN);
        const float4 sh = *reinterpret_cast<const float4*>(st + 2 * CIN);
        x.x = (x.x - mu.x) * sc.x + sh.x;
        x.y = (x.y - mu.y) * sc.y + sh.y;
        x.z = (x.z - mu.z) * sc.z + sh.z;
        x.w = (x.w - mu.w) * sc.w + sh.w;
    }
    if (flags & SBC_PRO_ELU) x = elu4(x, (flags & SBC_PRO_ELU_ACC) != 0);
    unsigned short* dst = lds16 + pix * SH + c4 * 4;
    if constexpr (TERMS == 1) {
        f16x4 h;
        h[0] = (_Float16)x.x; h[1] = (_Float16)x.y; h[2] = (_Float16)x.z; h[3] = (_Float16)x.w;
        *reinterpret_cast<f16x4*>(dst) = h;
    } else if constexpr (TERMS == 2) {
        scale_track(x, ss);
        uint2 h, l;
        split_f16x2(x, ss->scale, h, l);
        *reinterpret_cast<uint2*>(dst) = h;
        *reinterpret_cast<uint2*>(dst + plane) = l;
    } else {
        bf16x4 h, m, l;
        split3(x, h, m, l);
        *reinterpret_cast<bf16x4*>(dst) = h;
        *reinterpret_cast<bf16x4*>(dst + plane) = m;
        *reinterpret_cast<bf16x4*>(dst + 2 * plane) = l;
    }
}

template <int CIN, int NTHREADS, int NPF, bool P2, int TERMS = 3>
__device__ __forceinline__ void stage_tile_split(unsigned short* lds16, int plane, const float* __restrict__ in,
                                                 const float* __restrict__ stats, int flags, const TileGeom& g,
                                                 const Dims<P2>& d, int tid, StageScale* ss = nullptr, int n0 = -1) {
    constexpr int SH = CIN + 8;
    float4 pf[NPF];
    stage_issue<CIN, NTHREADS, NPF>(pf, in, g, d.W, tid);
    const int total = g.nps * (CIN / 4);
#pragma unroll
    for (int u = 0; u < NPF; ++u) {
        const int idx = u * NTHREADS + tid;
        if (idx < total) stage_put_split<CIN, P2, TERMS>(lds16, plane, pf[u], idx, stats, flags, g, d, ss, n0);
    }
    const float* src = in + (size_t)g.rs0 * d.W * CIN;
    for (int idx = NPF * NTHREADS + tid; idx < total; idx += NTHREADS)
        stage_put_split<CIN, P2, TERMS>(lds16, plane, ld_stream(src + (size_t)idx * 4), idx, stats, flags, g, d, ss, n0);
    // the zero pixel of each plane (SH / 2 dwords each)
    for (int i = tid; i < TERMS * (SH / 2); i += NTHREADS)
        reinterpret_cast<unsigned*>(lds16 + (i / (SH / 2)) * plane + g.nps * SH)[i % (SH / 2)] = 0u;
}

// ---- InstanceNorm++ statistics from tile moments (SBC_EPI_MOMENTS_OUT -> SBC_OP_INORM_STATS + SBC_PRO_NORM_MOMENTS) --------------
// Instead of a statistics launch that reads the whole tensor again, the PRODUCER of a tensor writes, per 128-pixel tile and
// channel, the tile's (mean, M2 = sum (x - mean)^2); a small launch (ops.hip: inorm_from_moments_kernel, one workgroup per
// sample) merges the tiles of a sample into the (mu, scale, shift) the statistics launch would have written, and consumers
// read those as always.  (Round 2 merged the moments in every CONSUMER workgroup: that cost each of them a dependent chain of
// loads before staging could start, and was limited to 8 tiles per sample.)  Everything has a fixed order, so results stay
// independent of batch composition and reproducible bit for bit.
__device__ __forceinline__ void chan_merge1(float& mean_a, float& m2_a, float na, float mean_b, float m2_b, float nb) {
    const float n = na + nb, d = mean_b - mean_a;
    mean_a += d * (nb / n);
    m2_a += m2_b + d * d * (na * nb / n);
}
__device__ __forceinline__ void merge_equal4(float4& ma, float4& qa, float4 mb, float4 qb, float n) {   // two partials of n each
    const float hn = 0.5f * n;
    float d;
    d = mb.x - ma.x; ma.x += 0.5f * d; qa.x += qb.x + d * d * hn;
    d = mb.y - ma.y; ma.y += 0.5f * d; qa.y += qb.y + d * d * hn;
    d = mb.z - ma.z; ma.z += 0.5f * d; qa.z += qb.z + d * d * hn;
    d = mb.w - ma.w; ma.w += 0.5f * d; qa.w += qb.w + d * d * hn;
}
// lane ^ 8 inside a row of 16 lanes is a rotation by 8 (one DPP move, no LDS); lane ^ 16 a swizzle inside 32 lanes
__device__ __forceinline__ float xor8(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128 /* row_ror:8 */, 0xf, 0xf, false));
}
__device__ __forceinline__ float xor16(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x401f /* and 0x1f, xor 0x10 */));
}
__device__ __forceinline__ float4 xor8_4(float4 v) { return make_float4(xor8(v.x), xor8(v.y), xor8(v.z), xor8(v.w)); }
__device__ __forceinline__ float4 xor16_4(float4 v) { return make_float4(xor16(v.x), xor16(v.y), xor16(v.z), xor16(v.w)); }

// Producer side, 256 threads, 32 channels: thread (tid >> 3, tid & 7) holds four pixels of channel quad tid & 7 (any four:
// together the threads cover the tile's 128 pixels once).  `red`: LDS scratch of 8 * 8 * 8 floats that nobody else touches;
// `pm_tile`: this tile's [32][2] output.  Contains one workgroup barrier (conv_wsp.hip calls the two halves around its own).
__device__ __forceinline__ void tile_moments_partials32(const float4 (&v)[4], float* red, int tid) {
    float4 mean, m2;
    mean.x = ((v[0].x + v[1].x) + (v[2].x + v[3].x)) * 0.25f; mean.y = ((v[0].y + v[1].y) + (v[2].y + v[3].y)) * 0.25f;
    mean.z = ((v[0].z + v[1].z) + (v[2].z + v[3].z)) * 0.25f; mean.w = ((v[0].w + v[1].w) + (v[2].w + v[3].w)) * 0.25f;
    m2 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float d;
        d = v[i].x - mean.x; m2.x = fmaf(d, d, m2.x); d = v[i].y - mean.y; m2.y = fmaf(d, d, m2.y);
        d = v[i].z - mean.z; m2.z = fmaf(d, d, m2.z); d = v[i].w - mean.w; m2.w = fmaf(d, d, m2.w);
    }
    // lanes that share the channel quad differ in lane bits 3..5: two pairwise merges of equal counts (4, 8) in registers,
    // then the eight 16-pixel partials of a quad (two per wave) meet in LDS
    merge_equal4(mean, m2, xor8_4(mean), xor8_4(m2), 4.f);
    merge_equal4(mean, m2, xor16_4(mean), xor16_4(m2), 8.f);
    const int lane = tid & 63, part = (tid >> 6) * 2 + (lane >> 5);
    if ((lane & 31) < 8) {
        *reinterpret_cast<float4*>(red + (part * 8 + (lane & 7)) * 8) = mean;
        *reinterpret_cast<float4*>(red + (part * 8 + (lane & 7)) * 8 + 4) = m2;
    }
}
// second half, for tid < 32 (one channel each) after a barrier: the eight partials (16 pixels each) in order
__device__ __forceinline__ void tile_moments_merge32(const float* red, float* __restrict__ pm_tile, int tid) {
    const int c4 = tid >> 2, k = tid & 3;
    // equal counts: mean of the means, and M2 = sum M2_w + 16 sum (mean_w - mean)^2 (fixed order, no dependent chain)
    float mw[8], mu = 0.f, q = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) { mw[w] = red[(w * 8 + c4) * 8 + k]; mu += mw[w]; q += red[(w * 8 + c4) * 8 + 4 + k]; }
    mu *= 0.125f;
    float dd = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) { const float d = mw[w] - mu; dd = fmaf(d, d, dd); }
    *reinterpret_cast<float2*>(pm_tile + tid * 2) = make_float2(mu, fmaf(16.f, dd, q));
}
__device__ __forceinline__ void tile_moments_out32(const float4 (&v)[4], float* red, float* __restrict__ pm_tile, int tid) {
    tile_moments_partials32(v, red, tid);
    __syncthreads();
    if (tid < 32) tile_moments_merge32(red, pm_tile, tid);
}

// ---- shared by the persistent direct kernels (conv_pair.hip, conv_dp.hip) ------------------------------------------------------
// range tracking of one tile (tile.h): the wave's verdict on what it just converted goes into `rbits` (wave-uniform, raised once
// at the end of the launch); calibration launches (calib != NULL) fold the wave's maximum into the record's slot right away
__device__ __forceinline__ void pair_range_tile(float t, float scale, unsigned& rbits, float* __restrict__ calib_slot) {
    rbits |= f16x2_range_bits(t, scale);
    if (calib_slot) {
        for (int o = 32; o > 0; o >>= 1) t = __builtin_fmaxf(t, __shfl_xor(t, o));
        if ((threadIdx.x & 63) == 0 && t > 0.f) atomicMax(reinterpret_cast<unsigned*>(calib_slot), __float_as_uint(t));
    }
}

__device__ __forceinline__ void lds_barrier() {
    // LDS traffic only: the tile in flight by LDS-DMA (vmcnt) must NOT be waited for here
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// host side: log2 of a power of two, or -1
inline int log2_exact(int v) {
    if (v <= 0 || (v & (v - 1))) return -1;
    int s = 0;
    while ((1 << s) < v) ++s;
    return s;
}

}  // namespace sbc

// EXPERIMENT (round 6; not part of the product: built only by tools/build_pair32_variant.sh, which adds it to a variant library and
// routes SBC_OP_CONV_PAIR launches of its shape to it).  RESULT: correct -- tests/test_gpu_ops.py::test_conv_pair_matches_oracle passes
// with it, batch-independent bit for bit -- and SLOWER than conv_pair_roll_kernel: 264 us against 230 us stand-alone at 1700 samples
// (the same with the vector work outside the K loops: 253 us).  Per item, wave 0 of each role (s_memtime ticks, profiles/r06_pair32_phases.txt):
// the conv1 role's epilogue is about half hidden between its matrix instructions (3082 against 2581 + 1139 ticks), the conv2 role's
// conversion not at all (4155 against 3961), and that role is the critical path (K loop 4155 + residual epilogue and stores 1251 of
// ~6100 ticks per item; the other role waits ~2280 at the barrier).  What the port taught about the matrix pipe is in DESIGN.md section 8.
//
// One RCU block of the score network in ONE launch (SBC_OP_CONV_PAIR), round 6: 16-pixel rows, 32 channels, conv_mode f16x2,
//       out = x + conv2(ELU(conv1(ELU(x))))        ncsnv2/models/layers.py:126-134 (n_stages = 2, 3x3, no bias)
// on v_mfma_f32_32x32x16_f16, with the vector work INSIDE the K loops.
//
// Why (tools/experiments/issue_overlap2.hip, profiles/r06_issue_overlap_shapes.txt): on gfx950 a vector instruction of ANOTHER wave
// does not issue while a wave's matrix instruction occupies the SIMD -- matrix waves and vector waves on one SIMD take the SUM of
// their times, whatever the MFMA shape (round 4's result, confirmed) -- but the wave that issued a 32-cycle v_mfma_f32_32x32x16 can
// itself issue up to ~6 independent vector instructions in its shadow at no cost (one, for the 16-cycle 16x16x32).  The pipeline of
// conv_pair_roll_kernel (conv_pair.hip) splits the work by ROLE -- conversion waves are all vector work, matrix waves all MFMA plus an
// epilogue -- so its time is the serial sum (issue model 0.80 of 169 us, matrix pipe 0.45 busy).  Here every wave is a matrix wave on
// the 32-cycle shape and carries the vector work as fillers between its own matrix instructions:
//   role 0 (4 waves, one per SIMD): conv1 of item k, and between its matrix instructions the EPILOGUE of item k - 1 (descale, ELU,
//           two fp16 terms, write to the intermediate rows M);
//   role 1 (4 waves): conv2 of item k - 3, and between its matrix instructions the CONVERSION of item k (raw fp32 rows that arrived by
//           LDS-DMA -> ELU -> two fp16 terms -> operand rows X); + x, store behind the K loop (requested in front of it).
// A wave owns ALL 32 output channels of 32 pixels (two adjacent image rows of the item's eight): D[32 couts][32 px] += W[32][16 cin]
// X[16 cin][32 px] per (tap, 16-channel half, term pair): 54 matrix instructions of 32 cycles per wave and item, the whole filter of
// its convolution resident in registers (144), the accumulators of the item before it (16) beside the current ones.
// Items, row rings (X: 24 rows + a 4-row copy in front of row 0; M: 24 + 2), PRE items and the workgroup's contiguous run of tiles are
// conv_pair_roll_kernel's (its header has the picture); the intermediate of item k now appears one iteration later, so conv2 runs
// three iterations behind the conversion.  One workgroup barrier per item.
// Pixel -> lane: lanes 0..15 of a 32-lane half are columns 0..15 of the unit's first row, lanes 16..31 columns (n - 18) mod 16 of its
// second row: a plane row is 18 slots = 288 bytes, so the second row's slots sit two 16-byte banks on, and the rotation by two makes
// each of ds_read_b128's four lane groups hit sixteen different slots (conflict-free for every tap).
// The sums differ from the 16x16x32 kernels' in their grouping (two 16-channel halves per tap instead of one 32-channel step), so this
// kernel takes EVERY launch of its shape, whatever the batch: results do not depend on the batch size or on where a run starts.
#include <stdlib.h>
#include <type_traits>
#include "../../score_based_channels_amd/csrc/conv_pair.h"

namespace sbc {

typedef float f32x16v __attribute__((ext_vector_type(16)));

template <int B, int E, class F>
__device__ __forceinline__ void static_for32(F&& f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for32<B + 1, E>(f);
    }
}

// the pieces of elu4 (common.h) and split_f16x2 (tile.h), one issue group each: they sit between matrix instructions
__device__ __forceinline__ void p_mul_log2e(float4 v, float4& e) {
    asm("v_mul_f32_e32 %0, 0x3fb8aa3b, %4\n\tv_mul_f32_e32 %1, 0x3fb8aa3b, %5\n\tv_mul_f32_e32 %2, 0x3fb8aa3b, %6\n\tv_mul_f32_e32 %3, 0x3fb8aa3b, %7"
        : "=&v"(e.x), "=&v"(e.y), "=&v"(e.z), "=&v"(e.w) : "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w));
}
__device__ __forceinline__ void p_exp2(float& a, float& b) { asm("v_exp_f32_e32 %0, %0\n\tv_exp_f32_e32 %1, %1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void p_one_minus_clamp(float4& e) {            // c = clamp(1 - exp(x), 0, 1)
    asm("s_nop 0\n\tv_sub_f32_e64 %0, 1.0, %0 clamp\n\tv_sub_f32_e64 %1, 1.0, %1 clamp\n\tv_sub_f32_e64 %2, 1.0, %2 clamp\n\tv_sub_f32_e64 %3, 1.0, %3 clamp"
        : "+v"(e.x), "+v"(e.y), "+v"(e.z), "+v"(e.w));
}
__device__ __forceinline__ void p_max_neg(float4& v, float4 c) {          // elu = max(x, -c)
    asm("v_max_f32_e64 %0, %0, -%4\n\tv_max_f32_e64 %1, %1, -%5\n\tv_max_f32_e64 %2, %2, -%6\n\tv_max_f32_e64 %3, %3, -%7"
        : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w) : "v"(c.x), "v"(c.y), "v"(c.z), "v"(c.w));
}
__device__ __forceinline__ void p_split_hi(float4 x, float s, uint2& h) {
    asm("v_fma_mixlo_f16 %0, %2, %6, 0 op_sel_hi:[0,0,0]\n\t"
        "v_fma_mixlo_f16 %1, %4, %6, 0 op_sel_hi:[0,0,0]\n\t"
        "v_fma_mixhi_f16 %0, %3, %6, 0 op_sel_hi:[0,0,0]\n\t"
        "v_fma_mixhi_f16 %1, %5, %6, 0 op_sel_hi:[0,0,0]"
        : "=&v"(h.x), "=&v"(h.y) : "v"(x.x), "v"(x.y), "v"(x.z), "v"(x.w), "s"(s));
}
__device__ __forceinline__ void p_split_lo(float4 x, float s, uint2 h, uint2& l) {
    asm("v_fma_mixlo_f16 %0, %2, %6, -%7 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixlo_f16 %1, %4, %6, -%8 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %0, %3, %6, -%7 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %1, %5, %6, -%8 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
        : "=&v"(l.x), "=&v"(l.y) : "v"(x.x), "v"(x.y), "v"(x.z), "v"(x.w), "s"(s), "v"(h.x), "v"(h.y));
}

#ifdef SBC_PAIR_TIMING
#define P32_MARK(k) do { const unsigned long long _t = __builtin_readcyclecounter(); pt[k] += _t - pt_last; pt_last = _t; } while (0)
#else
#define P32_MARK(k) do { } while (0)
#endif

// ILV = 1: vector work between the matrix instructions (the product); ILV = 0: the same schedule with the vector work behind / in
// front of the K loops (A/B aid: what the interleave buys; identical results)
template <int ILV>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_pair32_kernel(PairParams p) {
    constexpr int W = 16, R = 8, C = 32, NTH = 256;
    constexpr int KGS = C / 8, C4 = C / 4, NT = 2;
    constexpr int WP = W + 2, ROWB = WP * 16;          // bytes of one plane row
    constexpr int RING = 24, XMIR = 4, MMIR = 2;       // ring rows; rows copied in front of row 0
    constexpr int XPS = ((RING + XMIR) * ROWB + 255) / 256 * 256;
    constexpr int MPS = ((RING + MMIR) * ROWB + 255) / 256 * 256;
    constexpr int RAW_BYTES = R * W * C * 4;
    constexpr int X_OFF = 2 * RAW_BYTES, M_OFF = X_OFF + NT * KGS * XPS;
    constexpr int NK = R * W * C4 / NTH;               // 16-byte chunks per thread and TILE item (4); a PRE item: the last two
    constexpr int L2 = 3;                              // conv2 of an item runs this many iterations behind its conversion
    extern __shared__ __attribute__((aligned(256))) unsigned char smem[];

    const int role = __builtin_amdgcn_readfirstlane((int)threadIdx.x / NTH);      // 0: conv1 (+ its epilogue), 1: conversion + conv2
    const int tid = threadIdx.x - role * NTH, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);                   // rows 2 wave, 2 wave + 1 of an item
    const int n = lane & 31, kq = lane >> 5;
    const int urow = n >> 4, col = urow ? ((n - 18) & 15) : n;
    const int H = p.H;

    // ---- the whole filter of this role's convolution: [tap][16-channel half][term], sbc_pack_conv_weight_f16x2 layout
    uint4 wf[9][2][NT];
    {
        const uint4* w = role == 0 ? p.w1 : p.w2;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int t = 0; t < NT; ++t) wf[tap][g][t] = w[((tap * 2 + g) * NT + t) * 64 + lane];
    }
    const float4 t1 = f16x2_trailer(reinterpret_cast<const float4*>(p.w1), 9 * (C / 16) * (C / 32) * NT);
    const float4 t2 = f16x2_trailer(reinterpret_cast<const float4*>(p.w2), 9 * (C / 16) * (C / 32) * NT);
    const float scale1 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, t1.x)));
    const float scale2 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, t2.x)));
    const float descale1 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, t1.y)));
    const float descale2 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, t2.y)));
    unsigned rbits = 0;
    if (t1.w != 0.f || t2.w != 0.f) rbits |= 4u;     // (the four-instruction ELU only: conv_pair_kernel)

    // ---- zero the padding columns of every plane row once
    for (int i = threadIdx.x; i < NT * KGS * (RING + XMIR) * 2; i += 2 * NTH) {
        const int side = i & 1, row = (i >> 1) % (RING + XMIR), pl = (i >> 1) / (RING + XMIR);
        *reinterpret_cast<uint4*>(smem + X_OFF + pl * XPS + row * ROWB + side * (W + 1) * 16) = make_uint4(0, 0, 0, 0);
    }
    for (int i = threadIdx.x; i < NT * KGS * (RING + MMIR) * 2; i += 2 * NTH) {
        const int side = i & 1, row = (i >> 1) % (RING + MMIR), pl = (i >> 1) / (RING + MMIR);
        *reinterpret_cast<uint4*>(smem + M_OFF + pl * MPS + row * ROWB + side * (W + 1) * 16) = make_uint4(0, 0, 0, 0);
    }

    // ---- this workgroup's run: XCD x owns tiles [t_begin, t_end), its workgroups take equal contiguous pieces [a, b) of it
    const int xcd = blockIdx.x & 7, jw = blockIdx.x >> 3;
    const int t_begin = xcd * p.tiles_per_xcd;
    const int cnt = max(min(t_begin + p.tiles_per_xcd, p.ntiles) - t_begin, 0);
    const int a = t_begin + jw * cnt / p.wgs_per_xcd, b = t_begin + (jw + 1) * cnt / p.wgs_per_xcd;
    if (a >= b) return;                                               // (whole workgroup)
    const int tps = p.tiles_per_sample;
    const int n_items = (b - a) + 1 + ((b - 1) / tps - a / tps);    // tiles + the run's PRE + one PRE per sample that starts inside
    struct Cur { int n, j, pre; };
    auto cur_first = [&]() { Cur q; q.n = a / tps; q.j = a - q.n * tps; q.pre = 1; return q; };
    auto cur_next = [&](Cur& q) {
        if (q.pre) { q.pre = 0; return; }
        if (++q.j == tps) { q.j = 0; ++q.n; q.pre = 1; }
    };
    auto next_slot = [](int s) { return s == 16 ? 0 : s + 8; };

    // ---- one convolution of this wave's unit: 18 K steps (tap, half) of three matrix instructions; fill(step) sits behind matrix
    // instruction `step` (0 .. 53).  ub0: byte address of tap (0, 0), half 0, term 0 for this lane; PS: plane stride
    // Two accumulators, alternating by matrix instruction: a filler between two matrix instructions on the SAME accumulator breaks the
    // pipe's accumulator forwarding (~ +43 cycles each: /opt/skills/guides/MI355X_MICROARCH.md; measured here as fillers that hid
    // nothing), and a wave's matrix instruction does not issue before the pipe takes it, so fillers behind a back-to-back GROUP get the
    // shadow of its last instruction only.  A, B, A, B ... with up to six fillers behind each is the pattern
    // tools/experiments/issue_overlap2.hip measures as free.  The two partial sums are added at the end (a fixed order).
    auto kloop = [&](const int ub0, const int PS, f32x16v& acc, auto&& fill) {
        constexpr int NS = 18, D = 2;
        f16x8 ring[D][NT];
        f32x16v acc2;
        auto ld = [&](int s) {                                          // (compile-time constant at every call)
            const int tap = s >> 1, g = s & 1;
            const int off = ((tap / 3) * WP + (tap % 3)) * 16 + 2 * g * PS;
#pragma unroll
            for (int t = 0; t < NT; ++t) ring[s % D][t] = *reinterpret_cast<const f16x8*>(smem + ub0 + (off + t * KGS * PS));
        };
        static_for32<0, D - 1>([&](auto sc) { ld(decltype(sc)::value); });
        static_for32<0, NS>([&](auto sc) {
            constexpr int s = decltype(sc)::value, tap = s >> 1, g = s & 1;
            if constexpr (s + D - 1 < NS) ld(s + D - 1);                  // (in flight during this step's three matrix instructions)
            const f16x8 xh = ring[s % D][0], xl = ring[s % D][1];
            const f16x8 wh = __builtin_bit_cast(f16x8, wf[tap][g][0]), wl = __builtin_bit_cast(f16x8, wf[tap][g][1]);
            static_for32<0, 3>([&](auto ic) {
                constexpr int i = decltype(ic)::value, m = 3 * s + i;
                f32x16v& a = (m & 1) ? acc2 : acc;
                const f16x8 wa = i == 1 ? wl : wh, xb = i == 0 ? xl : xh;       // (wh, xl), (wl, xh), (wh, xh)
                if constexpr (m < 2) {
                    f32x16v z;
#pragma unroll
                    for (int k = 0; k < 16; ++k) z[k] = 0.f;
                    a = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa, xb, z, 0, 0, 0);
                } else {
                    a = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa, xb, a, 0, 0, 0);
                }
                fill(std::integral_constant<int, m>{});
                __builtin_amdgcn_sched_barrier(0);
            });
        });
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] += acc2[i];
    };
    auto no_fill = [](auto) {};

    // Vector work in NINE issue groups per float4 (group k of chunk c behind matrix instruction 9 c + k): shared by the conversion
    // (chunk = one 16-byte piece of the raw rows) and conv1's epilogue (chunk = one channel quad of the accumulators)
    struct Piece { float4 v, e; uint2 h, l; };
    auto piece_step = [&](auto kc, Piece& pc, float scale, float& amax, unsigned char* dst, int PS, bool mirror, bool zero) {
        constexpr int k = decltype(kc)::value;
        if constexpr (k == 1) {
            pc.v.x = zero ? 0.f : pc.v.x; pc.v.y = zero ? 0.f : pc.v.y; pc.v.z = zero ? 0.f : pc.v.z; pc.v.w = zero ? 0.f : pc.v.w;
            p_mul_log2e(pc.v, pc.e);
        }
        if constexpr (k == 2) p_exp2(pc.e.x, pc.e.y);
        if constexpr (k == 3) p_exp2(pc.e.z, pc.e.w);
        if constexpr (k == 4) p_one_minus_clamp(pc.e);
        if constexpr (k == 5) p_max_neg(pc.v, pc.e);
        if constexpr (k == 6) {
            amax = __builtin_fmaxf(__builtin_fmaxf(amax, __builtin_fabsf(pc.v.x)), __builtin_fabsf(pc.v.y));
            amax = __builtin_fmaxf(__builtin_fmaxf(amax, __builtin_fabsf(pc.v.z)), __builtin_fabsf(pc.v.w));
            p_split_hi(pc.v, scale, pc.h);
        }
        if constexpr (k == 7) p_split_lo(pc.v, scale, pc.h, pc.l);
        if constexpr (k == 8) {
            *reinterpret_cast<uint2*>(dst) = pc.h;
            *reinterpret_cast<uint2*>(dst + KGS * PS) = pc.l;
            if (mirror) {                                              // the copy in front of row 0
                *reinterpret_cast<uint2*>(dst - RING * ROWB) = pc.h;
                *reinterpret_cast<uint2*>(dst - RING * ROWB + KGS * PS) = pc.l;
            }
        }
    };

    // LDS-DMA of an item's raw input rows (8 rows x 2 KB; a PRE item: the last four) by the four waves of the role that calls it -- the
    // conv1 role: its waves have no other memory traffic, so their vmcnt counts these pieces only (the conv2 role's stores would make a
    // wait for the pieces a wait for the stores' acknowledgement: measured, 5 us per item)
    auto issue_dma = [&](const Cur& q, int buf) {
        const int rb = R * q.j - (q.pre ? R : 0) + 2;            // image row of the slot's row 0
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            if (k < NK / 2 && q.pre) continue;                   // (uniform)
            const int j = k * 4 + wave;                           // piece: 64 chunks = half a row
            const int ri = j >> 1;
            const int grow = min(max(rb + ri, 0), H - 1);         // rows outside the image: any row inside (converted to zeros)
            const char* sbase = reinterpret_cast<const char*>(p.in) + ((size_t)(q.n * H + grow) * W * C) * 4 + (size_t)(j & 1) * 1024;
            const unsigned dst = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem + buf * RAW_BYTES + j * 1024;
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(lane * 16), "s"(dst), "s"(sbase) : "memory");
        }
    };
#ifdef SBC_PAIR_TIMING
    unsigned long long pt[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pt_last = __builtin_readcyclecounter();
#endif
    if (role == 0) {
        // ================================================================= conv1 of item k; epilogue of item k - 1 between its matrix instructions
        // intermediate row m of the slot (image row r0 + 1 + m) reads X rows slot + m - 2 .. slot + m; this lane: m = 2 wave + urow
        const int mrow = 2 * wave + urow;
        f32x16v acc, accp;
        Cur q = cur_first(), qp = q;
        int slot = 0, slotp = 0;
        bool have_p = false;
        float tb = 0.f;
        // epilogue of (accp, qp, slotp): chunk c = channel quad 2 c + kq of this lane's pixel
        // (per pending item: the lane's write address of chunk 0 and whether its row lies outside the image -- set by epi_setup)
        unsigned char* epi_base = smem;
        bool epi_out = false, epi_mir = false;
        auto epi_setup = [&]() {
            epi_base = smem + M_OFF + ((slotp + mrow + MMIR) * WP + col + 1) * 16 + kq * 8;
            const int grow = R * qp.j - (qp.pre ? R : 0) + 1 + mrow;
            epi_out = grow < 0 || grow >= H;
            epi_mir = mrow >= 6 && slotp == 16;
        };
        Piece pc;
        Cur qd = cur_first();
        issue_dma(qd, 0);
        cur_next(qd);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        auto epi_fill = [&](auto stc) {
            constexpr int st = decltype(stc)::value;
            if constexpr (st < 36) {
                constexpr int c = st / 9, k = st % 9;
                if constexpr (k == 0) pc.v = make_float4(accp[4 * c] * descale1, accp[4 * c + 1] * descale1, accp[4 * c + 2] * descale1, accp[4 * c + 3] * descale1);
                else piece_step(std::integral_constant<int, k>{}, pc, scale2, tb, epi_base + c * MPS, MPS, epi_mir, epi_out);
            }
        };
        auto epi_plain = [&]() { static_for32<0, 36>([&](auto stc) { epi_fill(stc); }); };
        for (int it = 0; it < n_items + L2; ++it) {
            P32_MARK(0);
            lds_barrier();
            P32_MARK(1);
            if (it + 1 < n_items) { issue_dma(qd, (it + 1) & 1); cur_next(qd); }
            const bool item = it >= 1 && it - 1 < n_items;
            const bool active = item && (!q.pre || wave == 3);          // a PRE item: rows 6, 7 only -- the last wave's unit
            if (active) {
                const int ub0 = X_OFF + kq * XPS + ((slot + mrow - 2 + XMIR) * WP + col) * 16;
                if (ILV && have_p) kloop(ub0, XPS, acc, epi_fill);
                else { if (have_p) epi_plain(); kloop(ub0, XPS, acc, no_fill); }
            } else if (have_p) {
                epi_plain();
            }
            P32_MARK(2);
            if (have_p) pair_range_tile(tb, scale2, rbits, p.calib ? p.calib + 1 : nullptr);
            tb = 0.f;
            have_p = active;
            if (active) { accp = acc; qp = q; slotp = slot; epi_setup(); }
            if (!ILV && have_p) {                                       // (A/B build: the epilogue right behind its K loop)
                epi_plain();
                pair_range_tile(tb, scale2, rbits, p.calib ? p.calib + 1 : nullptr);
                tb = 0.f;
                have_p = false;
            }
            if (item) { cur_next(q); slot = next_slot(slot); }
            P32_MARK(0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // item it + 1 has landed before the next barrier publishes it
            P32_MARK(3);
        }
    } else {
        // ================================================================= conversion of item k (LDS-DMA one item ahead); conv2 of item k - L2
        // conversion of item qc into X slot `cslot` from raw buffer `cbuf`: chunk kk = 16-byte piece kk * 256 + tid
        Cur qc = cur_first(), q2 = qc;
        int cslot = 0, cbuf = 0, slot2 = 0;
        float ta = 0.f;
        Piece pc;
        // chunk kk of a thread = pixel kk * 32 + tid / 8 (two plane rows per chunk), channel quad tid % 8: everything is linear in kk
        const int px0 = tid / C4, c40 = tid % C4, ri0 = px0 / W, pcol0 = px0 - ri0 * W;
        unsigned char* cv_base = smem;
        const unsigned char* cv_raw = smem;
        int cv_row = 0;
        bool cv_pre = false, cv_mir = false;
        auto cvt_setup = [&]() {                                       // per converted item (qc, cslot, cbuf)
            cv_base = smem + X_OFF + (c40 >> 1) * XPS + ((cslot + ri0 + XMIR) * WP + pcol0 + 1) * 16 + (c40 & 1) * 8;
            cv_raw = smem + cbuf * RAW_BYTES + tid * 16;
            cv_row = R * qc.j - (qc.pre ? R : 0) + 2 + ri0;
            cv_pre = qc.pre != 0;
            cv_mir = cslot == 16;
        };
        auto cvt_fill = [&](auto stc) {
            constexpr int st = decltype(stc)::value;
            if constexpr (st < 9 * NK) {
                constexpr int kk = st / 9, k = st % 9;
                if constexpr (k == 0) pc.v = *reinterpret_cast<const float4*>(cv_raw + kk * NTH * 16);
                else {
                    // rows outside the image, and the first two chunks of a PRE item (rows nobody reads, raw data nobody requested): zeros
                    const int grow = cv_row + 2 * kk;
                    piece_step(std::integral_constant<int, k>{}, pc, scale1, ta, cv_base + kk * 2 * ROWB, XPS, kk >= NK / 2 && cv_mir,
                               grow < 0 || grow >= H || (kk < NK / 2 && cv_pre));
                }
            }
        };
        auto cvt_plain = [&]() { static_for32<0, 9 * NK>([&](auto stc) { cvt_fill(stc); }); };

        const int orow = 2 * wave + urow;                              // this lane's output row within a tile
        for (int it = 0; it < n_items + L2; ++it) {
            P32_MARK(4);
            lds_barrier();
            P32_MARK(5);
            const bool do_cvt = it < n_items;
            const bool item2 = it >= L2;
            const bool do_c2 = item2 && !q2.pre;
            cbuf = it & 1;
            if (do_cvt) cvt_setup();
            if (do_c2) {
                // output row o of the tile (image row r0 + o) reads M rows slot + o - 2 .. slot + o
                const int r0 = R * q2.j;
                f32x16v acc;
                float4 xr[4];
                const unsigned o0 = (unsigned)(((q2.n * H + r0 + orow) * W + col) * C + kq * 4);
#pragma unroll
                for (int g = 0; g < 4; ++g) xr[g] = *reinterpret_cast<const float4*>(p.in + o0 + 8 * g);
                const int ub0 = M_OFF + kq * MPS + ((slot2 + orow - 2 + MMIR) * WP + col) * 16;
                if (ILV && do_cvt) kloop(ub0, MPS, acc, cvt_fill);
                else { if (do_cvt) cvt_plain(); kloop(ub0, MPS, acc, no_fill); }
                P32_MARK(6);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the residual operands (and the stores of the tile before: long done)
                P32_MARK(7);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float4 y;
                    y.x = fmaf(acc[4 * g], descale2, xr[g].x); y.y = fmaf(acc[4 * g + 1], descale2, xr[g].y);
                    y.z = fmaf(acc[4 * g + 2], descale2, xr[g].z); y.w = fmaf(acc[4 * g + 3], descale2, xr[g].w);
                    st_out(p.out + o0 + 8 * g, y);
                }
            } else if (do_cvt) {
                cvt_plain();
            }
            if (do_cvt) {
                pair_range_tile(ta, scale1, rbits, p.calib);
                ta = 0.f;
                cur_next(qc);
                cslot = next_slot(cslot);
            }
            if (item2) { cur_next(q2); slot2 = next_slot(slot2); }
        }
    }
    if (rbits && (threadIdx.x & 63) == 0) atomicOr(p.range_flag, rbits);
#ifdef SBC_PAIR_TIMING
    // wave 0 of each role: [role 0: other, barrier, conv1 + epilogue, wait for the LDS-DMA | role 1: other (stores, cursors), barrier, conversion + conv2, residual wait]
    if (tid == 0 && p.dbg)
        for (int k = 0; k < 8; ++k) atomicAdd(p.dbg + k, pt[k]);
    if (threadIdx.x == 0 && p.dbg) atomicAdd(p.dbg + 8, (unsigned long long)n_items);
#endif
}

template <int ILV>
static int launch_pair32_t(const PairParams& p0, hipStream_t stream, bool dry) {
    constexpr int ROWB = 18 * 16;
    constexpr int XPS = (28 * ROWB + 255) / 256 * 256, MPS = (26 * ROWB + 255) / 256 * 256;
    constexpr size_t lds = (size_t)2 * 8 * 16 * 32 * 4 + (size_t)2 * 4 * (XPS + MPS);
    static_assert(lds <= 160 * 1024, "LDS of the one resident workgroup");
    auto kern = conv_pair32_kernel<ILV>;
    { const int rc = ensure_dyn_lds(reinterpret_cast<const void*>(kern), lds); if (rc) return rc; }
    if (dry) return SBC_OK;
    PairParams p = p0;
    p.tiles_per_sample = p.H / 8;
    p.ntiles = p.B * p.tiles_per_sample;
    int dev = 0, cus = 256;
    SBC_CHECK_HIP(hipGetDevice(&dev));
    SBC_CHECK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    p.tiles_per_xcd = (p.ntiles + 7) / 8;
    p.wgs_per_xcd = max(1, min(persistent_cus(cus) / 8, p.tiles_per_xcd));
    hipLaunchKernelGGL(kern, dim3(8 * p.wgs_per_xcd), dim3(512), lds, stream, p);
    SBC_CHECK_HIP(hipGetLastError());
    return SBC_OK;
}

int launch_pair32(const PairParams& p0, hipStream_t stream, bool dry) {
    static const bool no_ilv = getenv("SBC_PAIR32_NO_ILV") != nullptr;       // A/B aid: vector work outside the K loops (identical results)
    if (dry) { const int rc = launch_pair32_t<0>(p0, stream, true); if (rc) return rc; return launch_pair32_t<1>(p0, stream, true); }
    return no_ilv ? launch_pair32_t<0>(p0, stream, false) : launch_pair32_t<1>(p0, stream, false);
}

}  // namespace sbc

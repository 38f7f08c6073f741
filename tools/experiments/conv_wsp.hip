// EXPERIMENT (round 3), NOT part of the product build: Winograd F(2x2,3x3) with producer / consumer wave roles.
// Result on MI355X (T = 1700, conv_mode f16x2, against conv_wx3.hip on the same box; bit-identical outputs for 32 -> 32):
//   32 -> 32 at 64x16: 168 us against 167 us;  32 -> 32 at 32x8: 40.9 against 40.3;  64 -> 64 at 32x8: 130 against 121 with one group
//   of matrix waves, 140 against 126 with two groups (one output block each, the form below).  (The 64 -> 64 path's mismatch in a
//   quarter of its outputs was the inline-assembly hazard of DESIGN.md section 9: with split_f16x2_settle() it is bit-identical.)
// Why it does not pay (PMC, tools/prof_conv.py): the 32 -> 32 kernel is bound by instruction issue, not by latency -- 700 vector
// instructions per wave and tile (58 % of the SIMD cycles) plus 24 matrix instructions (14 %); splitting the work between roles
// removes the waiting but not one instruction, and a SIMD's one matrix wave + one memory wave take the same 2.3 us per tile as
// four conv_wx3 waves.  The 64 -> 64 kernel IS latency bound (vector ALU 32 %, matrix pipe 17 %), but on its filter fragments:
// a lone matrix wave per SIMD waits for every column of U (K loop 7.2 us per tile against 6.7), the fragments do not fit in
// registers (256 per wave), and a deeper ring does not fit beside 128 accumulators.  Kept for the record.
// To build it: tools/build_variant.sh wsp conv_wx3.hip -DSBC_WITH_WSP (conv_wx3.hip then includes this file and offers it every layer).
// One lesson that generalises: keep every request sequence of a prefetching wave straight-line (clamped
// indices, no branches around loads) -- with conditional loads hipcc puts s_waitcnt vmcnt(0) in front of every load.
//
// Winograd F(2x2, 3x3) convolution with the work of a tile split between two kinds of waves ("roles") of a persistent
// workgroup -- the second generation of conv_wx3.hip for the layers whose launches have many tiles per CU.
//
// conv_wx3.hip runs a tile as one chain per workgroup: request the input rows, wait, prologue (InstanceNorm++ affine, ELU),
// LDS, barrier, K loop (B^T d B in registers, split, matrix instructions; the filter fragments from L2), T planes through
// LDS, barrier, output transform, residuals, store.  Timestamps of every workgroup (tools/prof_conv.py WX3_TIMING=1) show
// where a 64 -> 64 launch at 32x8 spends its 14 us per tile: 3.2 us waiting for the rows, 6.7 us in the K loop, 4.2 us in
// the exchange and the finish -- with 1.7 workgroups resident per CU (228 VGPRs, 74 KB of LDS) the phases of different
// workgroups overlap only a little, and the kernel runs at a third of every roofline it has (HBM, matrix pipe, vector ALU).
//
// Here a workgroup is eight waves that stay on their CU and walk a run of tiles:
//   waves 0-3  ("matrix" role, one B^T row xi each, exactly the K loop of conv_wx3.hip): read the staged tile k from LDS,
//              accumulate, apply A^T over nu, write the T planes of tile k;
//   waves 4-7  ("memory" role): request the rows of tile k + 2 and the residual of tile k (registers, a full iteration
//              ahead of their use, two register sets used alternately), finish tile k - 1 from its T planes (output transform,
//              bias, residuals, store, tile moments), run the prologue on the rows of tile k + 1 and write them to the other
//              staging buffer.
// Memory latency is paid by waves that have nothing else to do, the matrix waves never wait for global memory except for
// their filter fragments -- and a wave's vmcnt is its own, so the in-order return of the memory role's loads does not
// hold the fragments back (what defeated the register prefetch tried inside conv_wx3.hip).  Staging is double buffered;
// the T planes are double buffered where LDS allows (one barrier per tile), otherwise single (two barriers per tile: the
// matrix waves write the planes between them).  With 32 -> 32 channels every filter fragment of a wave (4 nu x 2 k-steps x
// terms) stays in registers for the whole launch.
//
// The arithmetic -- order of every sum included -- is that of conv_wx3.hip, so both kernels return the same bits.
// Eligible: 32 -> 32 and 64 -> 64 in the fp16 modes, whole 128-pixel tiles inside one sample (H*W >= 256, 128 % 2W == 0),
// epilogues bias / res1 / RES1_ELU / MOMENTS_OUT; everything else (EPI_UP, EPI_POOL, EPI_ELUGRAD, res2, tiles spanning samples,
// the bf16x3 mode) stays with conv_wx3.hip.
#include <stdlib.h>
#include "../../score_based_channels_amd/csrc/conv_common.h"

namespace sbc {

struct WspWalk {
    int ntiles, tiles_per_xcd, wgs_per_xcd, stage_floats;
    unsigned long long* dbg;        // SBC_WSP_TIMING builds: per-phase 100 MHz tick sums of one wave of each role
};

#ifdef SBC_WSP_TIMING
#define WS_INIT() unsigned long long ws[6] = {0, 0, 0, 0, 0, 0}, ws_last = wall_clock64()
#define WS_MARK(k) do { const unsigned long long t_ = wall_clock64(); ws[k] += t_ - ws_last; ws_last = t_; } while (0)
#define WS_DUMP(role) do { if (lane == 0 && (wave & 3) == 0 && wk.dbg) { for (int k_ = 0; k_ < 6; ++k_) wk.dbg[((size_t)blockIdx.x * 2 + role) * 8 + k_] = ws[k_]; \
                                wk.dbg[((size_t)blockIdx.x * 2 + role) * 8 + 6] = n_my; } } while (0)
#else
#define WS_INIT() do { } while (0)
#define WS_MARK(k) do { } while (0)
#define WS_DUMP(role) do { } while (0)
#endif

// LDS traffic only: the loads a memory-role wave has in flight (vmcnt) must not be waited for at a barrier
__device__ __forceinline__ void role_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// TBUF: T-plane buffers.  RES: the layer adds a residual (res1).  One workgroup per CU (512 threads, 256 VGPRs each).
// TOP: instantiation tag (own kernel symbol for the full-resolution ngf -> ngf layers, as in conv_wx3.hip).
template <int CIN, int COUT, int MODE, int TBUF, bool RES, bool TOP, int NPF>
__global__ __launch_bounds__(256 * (COUT / 32 + 1), COUT / 32 + 1) void conv_wsp_kernel(ConvParams p, WspWalk wk) {
    constexpr int NTERM = MODE == 0 ? 3 : MODE;
    constexpr int TM = 128, NTW = 32, S = CIN + 4, KG = CIN / 16, NBLK = COUT / 32;
    static_assert(NBLK <= 2 && KG <= 4, "one phase of at most two output blocks");
    // KGR groups of four matrix waves, one output block each (two waves of a SIMD inside K loops, as in conv_wx3 with two workgroups)
    constexpr int KGR = NBLK, QN = NBLK / KGR;
    constexpr int TS = 36, TPL = 8 * NTW * TS;              // floats of one output block's T planes [xi][b][tile][TS]
    // filter fragments resident in registers: 4 nu x KG x NBLK x NTERM x 4 VGPRs
    constexpr bool WRES = (4 * KG * NBLK * NTERM * 4 <= 64);
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = p.H, W = p.W, HW = H * W;
    const Dims<true> dm{H, W, HW, p.hsh, p.wsh};
    const int Wt = W >> 1;                                  // Winograd tiles per image row
    float* const tpl = lds + 2 * wk.stage_floats;
    float* const red = tpl + TBUF * NBLK * TPL;             // 2 x 512 floats: tile-moment partials (tile.h), by tile parity

    // tile walk: XCD x (= blockIdx % 8) owns a contiguous run of tiles, its workgroups take consecutive tiles of it
    const int xcd = blockIdx.x & 7, jw = blockIdx.x >> 3;
    const int t_begin = xcd * wk.tiles_per_xcd, t_end = min(t_begin + wk.tiles_per_xcd, wk.ntiles);
    const int first = t_begin + jw;
    const int n_my = first < t_end ? (t_end - first + wk.wgs_per_xcd - 1) / wk.wgs_per_xcd : 0;
    if (n_my == 0) return;                                  // (whole workgroup: no barrier is left waiting)
    auto geom_of = [&](int k) { return tile_geom(first + k * wk.wgs_per_xcd, TM, p.B, dm, 1); };

    float descale = 1.f;
    if constexpr (MODE == 2) descale = f16x2_trailer(p.wpk, 16 * KG * NBLK * NTERM).y;
    WS_INIT();

    if (wave < 4 * KGR) {
        // ================================================================================ matrix role
        const int xi = wave & 3, q0 = (wave >> 2) * QN;               // B^T row; first output block of this wave group
        const int khalf = 8 * (lane >> 5), col = lane & 31, rhalf = 4 * (lane >> 5);
        // B^T rows: xi=0: d0 - d2, xi=1: d1 + d2, xi=2: d2 - d1, xi=3: d1 - d3   ->  R = d[ia] + sgn * d[ib]
        const int ia = xi == 0 ? 0 : xi == 2 ? 2 : 1;
        const int ib = xi == 0 ? 2 : xi == 1 ? 2 : xi == 2 ? 1 : 3;
        const float sgn = xi == 1 ? 1.f : -1.f;
        const int t = lane & 31;
        const int tr = t >> (p.wsh - 1), tc = t - tr * Wt;
        // split U: [(xi*4 + nu)][kg][nb][term][lane] 16-byte fragments
        const uint4* const wp0 = reinterpret_cast<const uint4*>(p.wpk) + ((size_t)(xi * 4) * KG * NBLK) * NTERM * 64;
        uint4 ures[WRES ? 4 * KG * NBLK * NTERM : 1];
        if constexpr (WRES) {
#pragma unroll
            for (int i = 0; i < 4 * KG * NBLK * NTERM; ++i) ures[i] = wp0[(size_t)i * 64 + lane];
        }
        role_barrier();                                       // tile 0 staged
        for (int k = 0; k <= n_my; ++k) {
            if (k < n_my) {
                const TileGeom g = geom_of(k);
                int lane_l = lane;
                asm volatile("" : "+v"(lane_l));              // keeps the fragment loads of the K loop inside the tile loop
                const uint4* wp = wp0 + lane_l;
                const float* st = lds + (k & 1) * wk.stage_floats;
                // per lane: LDS offsets of the 2 x 4 patch pixels of its tile
                const int r0 = dm.div_w(g.p0);
                const int zoff = g.nps * S + khalf;
                const int grow = r0 + 2 * tr;
                const int h = dm.mod_h(grow);
                int off[2][4];
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const int ii = kk == 0 ? ia : ib;
                    const int hh = h - 1 + ii;
                    const bool rok = hh >= 0 && hh < H;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int ww = 2 * tc - 1 + j;
                        off[kk][j] = (rok && ww >= 0 && ww < W) ? ((grow - 1 + ii - g.rs0) * W + ww) * S + khalf : zoff;
                    }
                }
                f32x16 T[QN][2];
                f32x16 acc[QN][4];
                auto u_frag = [&](int nu, int kg, int q, int tt) { return wp[((size_t)((nu * KG + kg) * NBLK + q0 + q) * NTERM + tt) * 64]; };
#ifndef SBC_WSP_SETS
#define SBC_WSP_SETS 2
#endif
                constexpr int SETS = QN == 1 ? 4 : SBC_WSP_SETS, D = SETS - 1, NSEQ = KG * 4;
                uint4 uB[WRES ? 1 : SETS][QN][NTERM];
                auto u_load = [&](int gq) {                       // gq is a compile-time constant at every call
                    if constexpr (!WRES) {
                        const int gg = gq % NSEQ, nu_g = gg & 3, kg_g = (gg >> 2) % KG;
#pragma unroll
                        for (int q = 0; q < QN; ++q)
#pragma unroll
                            for (int tt = 0; tt < NTERM; ++tt) uB[gq % SETS][q][tt] = u_frag(nu_g, kg_g, q, tt);
                    }
                };
                auto u_get = [&](int gq, int q, int tt) {
                    if constexpr (WRES) return ures[(((gq & 3) * KG + (gq >> 2)) * NBLK + q) * NTERM + tt];
                    else return uB[gq % SETS][q][tt];
                };
#pragma unroll
                for (int gq = 0; gq < D; ++gq) u_load(gq);
#pragma unroll
                for (int q = 0; q < QN; ++q)
#pragma unroll
                    for (int nu = 0; nu < 4; ++nu)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[q][nu][r] = 0.f;
#pragma unroll
                for (int kg = 0; kg < KG; ++kg) {
                    // rows of B^T d for this lane's tile: R_j = d[ia][j] + sgn * d[ib][j], 8 channels each
                    float R[4][8];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float* qa = st + off[0][j] + kg * 16;
                        const float* qb = st + off[1][j] + kg * 16;
                        const float4 a0 = *reinterpret_cast<const float4*>(__builtin_assume_aligned(qa, 16));
                        const float4 a1 = *reinterpret_cast<const float4*>(__builtin_assume_aligned(qa + 4, 16));
                        const float4 b0 = *reinterpret_cast<const float4*>(__builtin_assume_aligned(qb, 16));
                        const float4 b1 = *reinterpret_cast<const float4*>(__builtin_assume_aligned(qb + 4, 16));
                        R[j][0] = fmaf(sgn, b0.x, a0.x); R[j][1] = fmaf(sgn, b0.y, a0.y);
                        R[j][2] = fmaf(sgn, b0.z, a0.z); R[j][3] = fmaf(sgn, b0.w, a0.w);
                        R[j][4] = fmaf(sgn, b1.x, a1.x); R[j][5] = fmaf(sgn, b1.y, a1.y);
                        R[j][6] = fmaf(sgn, b1.z, a1.z); R[j][7] = fmaf(sgn, b1.w, a1.w);
                    }
#pragma unroll
                    for (int nu = 0; nu < 4; ++nu) {
                        const int gq = kg * 4 + nu;
                        u_load(gq + D);                                   // wraps to a harmless re-read at the very end
                        // columns of B: nu=0: R0 - R2, nu=1: R1 + R2, nu=2: R2 - R1, nu=3: R1 - R3; then the split
                        float v[8];
#pragma unroll
                        for (int c = 0; c < 8; ++c)
                            v[c] = nu == 0 ? R[0][c] - R[2][c] : nu == 1 ? R[1][c] + R[2][c]
                                 : nu == 2 ? R[2][c] - R[1][c] : R[1][c] - R[3][c];
                        if constexpr (MODE == 1) {
                            f16x8 vf;
#pragma unroll
                            for (int c = 0; c < 8; ++c) vf[c] = (_Float16)v[c];
                            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                            for (int q = 0; q < QN; ++q)
                                acc[q][nu] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, __builtin_bit_cast(f16x8, u_get(gq, q, 0)),
                                                                                    acc[q][nu], 0, 0, 0);
                        } else if constexpr (MODE == 2) {
                            uint4 vhu, vlu;
                            split_f16x2(v[0], v[1], vhu.x, vlu.x);
                            split_f16x2(v[2], v[3], vhu.y, vlu.y);
                            split_f16x2(v[4], v[5], vhu.z, vlu.z);
                            split_f16x2(v[6], v[7], vhu.w, vlu.w);
                            split_f16x2_settle(vhu, vlu);                  // wait states before the matrix instructions read the terms (tile.h)
                            const f16x8 vh = __builtin_bit_cast(f16x8, vhu), vl = __builtin_bit_cast(f16x8, vlu);
                            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                            for (int q = 0; q < QN; ++q) {
                                const f16x8 uh = __builtin_bit_cast(f16x8, u_get(gq, q, 0)),
                                            ul = __builtin_bit_cast(f16x8, u_get(gq, q, NTERM > 1 ? 1 : 0));
                                acc[q][nu] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, uh, acc[q][nu], 0, 0, 0);
                                acc[q][nu] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, ul, acc[q][nu], 0, 0, 0);
                                acc[q][nu] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, uh, acc[q][nu], 0, 0, 0);
                            }
                        } else {
                            bf16x8 vh, vm, vl;
#pragma unroll
                            for (int c = 0; c < 8; ++c) {
                                const __bf16 hb = (__bf16)v[c];
                                const float r1 = v[c] - (float)hb;
                                const __bf16 mb = (__bf16)r1;
                                vh[c] = hb; vm[c] = mb; vl[c] = (__bf16)(r1 - (float)mb);
                            }
                            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                            for (int q = 0; q < QN; ++q) {
                                const bf16x8 uh = __builtin_bit_cast(bf16x8, u_get(gq, q, 0)),
                                             um = __builtin_bit_cast(bf16x8, u_get(gq, q, NTERM > 1 ? 1 : 0)),
                                             ul = __builtin_bit_cast(bf16x8, u_get(gq, q, NTERM > 2 ? 2 : 0));
                                // partial products, smallest first: (l,h) (h,l) (m,m) (m,h) (h,m) (h,h)
                                acc[q][nu] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vl, uh, acc[q][nu], 0, 0, 0);
                                acc[q][nu] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, ul, acc[q][nu], 0, 0, 0);
                                acc[q][nu] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vm, um, acc[q][nu], 0, 0, 0);
                                acc[q][nu] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vm, uh, acc[q][nu], 0, 0, 0);
                                acc[q][nu] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, um, acc[q][nu], 0, 0, 0);
                                acc[q][nu] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, uh, acc[q][nu], 0, 0, 0);
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                // A^T = [[1, 1, 1, 0], [0, 1, -1, -1]] applied over nu
#pragma unroll
                for (int q = 0; q < QN; ++q)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        T[q][0][r] = (acc[q][0][r] + acc[q][1][r]) + acc[q][2][r];
                        T[q][1][r] = (acc[q][1][r] - acc[q][2][r]) - acc[q][3][r];
                    }
                WS_MARK(0);
                if constexpr (TBUF == 1) role_barrier();          // the memory role is done with the planes of tile k - 1
                WS_MARK(1);
                float* const tl = tpl + (TBUF == 2 ? (k & 1) * NBLK * TPL : 0);
#pragma unroll
                for (int q = 0; q < QN; ++q)
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        float* e = tl + (q0 + q) * TPL + ((size_t)((xi * 2 + b) * NTW + rhalf)) * TS + col;
                        const f32x16 tv = T[q][b];
#pragma unroll
                        for (int r = 0; r < 16; ++r) e[((r & 3) + 8 * (r >> 2)) * TS] = tv[r];
                    }
                WS_MARK(2);
            } else if constexpr (TBUF == 1) {
                role_barrier();
            }
            role_barrier();
            WS_MARK(3);
        }
        WS_DUMP(0);
        return;
    }

    // ==================================================================================== memory role
    // Every request sequence below is straight-line code with a fixed number of vector-memory instructions per iteration
    // (indices clamped instead of branches, the residual a template parameter, the bias held in registers): the compiler's
    // s_waitcnt placement then leaves the requests of the NEXT iteration in flight while this one's data is consumed.  With
    // conditional loads it falls back to vmcnt(0) in front of every load, and the role costs 2.6 us per tile instead of 0.9.
    const int it = tid - 256 * KGR;
    struct IoSet {
        float4 pf[NPF];
        RegStats rs;
        float4 r1[RES ? NBLK : 1][4];
    };
    StageScale ss{1.f, 0.f};
    StageScale* const ssp = MODE == 2 ? &ss : nullptr;
    const int t = it >> 3, c4 = it & 7;
    const int tr = t >> (p.wsh - 1), tc = t - tr * Wt;
    const bool norm = (p.flags & SBC_PRO_NORM) != 0;
    const float* const stats_src = norm ? p.stats : p.in;     // (no statistics: three harmless loads of valid memory)
    float4 bias4[NBLK];
#pragma unroll
    for (int nb = 0; nb < NBLK; ++nb)
        bias4[nb] = p.bias ? *reinterpret_cast<const float4*>(p.bias + nb * 32 + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    const int k_last = n_my - 1;

    // requests for iteration k: the rows of tile k + 1 (to stage) and the residual of tile k - 1 (to finish); past either end
    // of the walk the nearest valid tile is requested again and not used
    auto issue_rows = [&](IoSet& s, int k) {
        const TileGeom g = geom_of(min(k + 1, k_last));
        const float* src = p.in + (size_t)g.rs0 * W * CIN;
        const int last = g.nps * (CIN / 4) - 1;
#pragma unroll
        for (int u = 0; u < NPF; ++u) s.pf[u] = ld_stream(src + (size_t)min(u * 256 + it, last) * 4);
        s.rs = load_reg_stats<CIN, 256>(stats_src, g, it);
    };
    auto issue_res = [&](IoSet& s, int k) {
        if constexpr (RES) {
            const TileGeom g = geom_of(min(max(k - 1, 0), k_last));
            const int grow = dm.div_w(g.p0) + 2 * tr;
#pragma unroll
            for (int nb = 0; nb < NBLK; ++nb)
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
                        s.r1[nb][a * 2 + b] = ld_stream(p.res1 + ((size_t)(grow + a) * W + 2 * tc + b) * COUT + nb * 32 + c4 * 4);
        }
    };
    auto commit = [&](const IoSet& s, int kt) {               // prologue of tile kt -> its staging buffer
        const TileGeom g = geom_of(kt);
        float* st = lds + (kt & 1) * wk.stage_floats;
        const int total = g.nps * (CIN / 4);
        constexpr int C4 = CIN / 4;
#pragma unroll
        for (int u = 0; u < NPF; ++u) {
            const int idx = u * 256 + it;
            float4 x = s.pf[u];
            if (norm) {
                x.x = (x.x - s.rs.mu.x) * s.rs.sc.x + s.rs.sh.x; x.y = (x.y - s.rs.mu.y) * s.rs.sc.y + s.rs.sh.y;
                x.z = (x.z - s.rs.mu.z) * s.rs.sc.z + s.rs.sh.z; x.w = (x.w - s.rs.mu.w) * s.rs.sc.w + s.rs.sh.w;
            }
            if (p.flags & SBC_PRO_ELU) x = elu4(x);
            if (idx < total) {
                scale_track(x, ssp);
                *reinterpret_cast<float4*>(st + (idx / C4) * S + (idx % C4) * 4) = x;
            }
        }
        for (int i = it; i < S; i += 256) st[g.nps * S + i] = 0.f;
    };
    auto finish = [&](const IoSet& s, int kf) {               // tile kf from its T planes
        const TileGeom g = geom_of(kf);
        const int grow = dm.div_w(g.p0) + 2 * tr;
        const float* tb = tpl + (TBUF == 2 ? (kf & 1) * NBLK * TPL : 0);
#pragma unroll
        for (int nb = 0; nb < NBLK; ++nb) {
            const int co = nb * 32 + c4 * 4;
            float4 y[2][2];                                       // [a][b]
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                float4 tx[4];
#pragma unroll
                for (int x = 0; x < 4; ++x)
                    tx[x] = *reinterpret_cast<const float4*>(tb + nb * TPL + ((size_t)((x * 2 + b) * NTW + t)) * TS + c4 * 4);
                y[0][b] = make_float4((tx[0].x + tx[1].x) + tx[2].x, (tx[0].y + tx[1].y) + tx[2].y,
                                      (tx[0].z + tx[1].z) + tx[2].z, (tx[0].w + tx[1].w) + tx[2].w);
                y[1][b] = make_float4((tx[1].x - tx[2].x) - tx[3].x, (tx[1].y - tx[2].y) - tx[3].y,
                                      (tx[1].z - tx[2].z) - tx[3].z, (tx[1].w - tx[2].w) - tx[3].w);
            }
            const float4 bv = bias4[nb];
            if (MODE == 2) {
                // descale (an exact power of two) in the same rounding as the bias add
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        y[a][b].x = fmaf(y[a][b].x, descale, bv.x); y[a][b].y = fmaf(y[a][b].y, descale, bv.y);
                        y[a][b].z = fmaf(y[a][b].z, descale, bv.z); y[a][b].w = fmaf(y[a][b].w, descale, bv.w);
                    }
            } else if (p.bias) {
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        y[a][b].x += bv.x; y[a][b].y += bv.y; y[a][b].z += bv.z; y[a][b].w += bv.w;
                    }
            }
            if constexpr (RES) {
                float4 rr[2][2];
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b) rr[a][b] = s.r1[nb][a * 2 + b];
                if (p.flags & SBC_EPI_RES1_ELU) {
#pragma unroll
                    for (int a = 0; a < 2; ++a)
#pragma unroll
                        for (int b = 0; b < 2; ++b) rr[a][b] = elu4(rr[a][b]);
                }
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        y[a][b].x += rr[a][b].x; y[a][b].y += rr[a][b].y;
                        y[a][b].z += rr[a][b].z; y[a][b].w += rr[a][b].w;
                    }
            }
            const bool moments = COUT == 32 && (p.flags & SBC_EPI_MOMENTS_OUT);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    float* o = p.out + ((size_t)(grow + a) * W + 2 * tc + b) * COUT + co;
                    // tile moments: no statistics launch will read this tensor back in before its consumer does -- cacheable
                    if (moments) *reinterpret_cast<float4*>(o) = y[a][b];
                    else st_stream(o, y[a][b]);
                }
            if constexpr (COUT == 32) {
                if (moments) {
                    const float4 yk[4] = {y[0][0], y[0][1], y[1][0], y[1][1]};
                    tile_moments_partials32(yk, red + (kf & 1) * 512, it);
                }
            }
        }
    };
    auto moments_merge = [&](int kf) {                         // after the barrier that follows finish(kf)
        if constexpr (COUT == 32) {
            if ((p.flags & SBC_EPI_MOMENTS_OUT) && it < 32) {
                const TileGeom g = geom_of(kf);
                tile_moments_merge32(red + (kf & 1) * 512, p.pm_out + (size_t)(g.p0 >> 7) * 32 * 2, it);
            }
        }
    };

    // DSET: two register sets used alternately, each requested a full iteration before its use (32 input channels: the memory
    // role is the slower one there); one set, requested at the end of the iteration before its use, where two would spill
    constexpr bool DSET = CIN <= 32;
    IoSet A;
    // before the first iteration: tile 0 staged, the requests of iteration 0 (rows of tile 1) in flight
    issue_rows(A, -1);
    commit(A, 0);
    issue_rows(A, 0);
    issue_res(A, 0);
    role_barrier();
    WS_MARK(0);
    if constexpr (DSET) {
        IoSet B;
        // iteration k consumes the set requested during iteration k - 1 and requests the set of iteration k + 1
        auto iteration = [&](IoSet& cur, IoSet& next, int k) {
            issue_rows(next, k + 1);
            issue_res(next, k + 1);
            WS_MARK(1);
            if (k >= 2) moments_merge(k - 2);
            if (k >= 1) finish(cur, k - 1);
            WS_MARK(2);
            if (k + 1 < n_my) commit(cur, k + 1);
            WS_MARK(3);
            if constexpr (TBUF == 1) role_barrier();
            role_barrier();
            WS_MARK(4);
        };
        for (int k = 0; k <= n_my; k += 2) {
            iteration(A, B, k);
            if (k + 1 <= n_my) iteration(B, A, k + 1);
        }
    } else {
        for (int k = 0; k <= n_my; ++k) {
            if (k + 1 < n_my) commit(A, k + 1);
            WS_MARK(3);
            if (k >= 2) moments_merge(k - 2);
            if (k >= 1) finish(A, k - 1);
            WS_MARK(2);
            issue_rows(A, k + 1);
            issue_res(A, k + 1);
            WS_MARK(1);
            if constexpr (TBUF == 1) role_barrier();
            role_barrier();
            WS_MARK(4);
        }
    }
    moments_merge(n_my - 1);
    if constexpr (MODE == 2) {
        if (ss.amax >= F16X2_LIMIT) atomicOr(p.range_flag, 1u);
    }
    WS_DUMP(1);
}

// ------------------------------------------------------------------------------------------------ dispatch
template <int CIN, int COUT, int MODE, int NPF>
static int launch_wsp(const ConvParams& p, hipStream_t stream, bool dry, unsigned long long* dbg) {
    constexpr int S = CIN + 4, NBLK = COUT / 32, TPL = 8 * 32 * 36;
    constexpr int TBUF = CIN == 32 ? 2 : 1;                 // what 160 KB of LDS hold beside two staging buffers
    const int ntiles = p.total_px / 128;
    WspWalk wk;
    wk.ntiles = ntiles;
    wk.tiles_per_xcd = (ntiles + 7) / 8;
    wk.stage_floats = (128 + 2 * p.W + 1) * S;
    wk.dbg = dbg;
    const size_t lds_all = (size_t)2 * wk.stage_floats * 4 + 2 * 512 * 4 + (size_t)TBUF * NBLK * TPL * 4;
    if (lds_all > 160 * 1024) return 1;
    wk.wgs_per_xcd = min(32, wk.tiles_per_xcd);
    constexpr bool TOPV = CIN == 32 && COUT == 32;
    const bool top = TOPV && p.top;
    auto kern = p.res1 ? (top ? conv_wsp_kernel<CIN, COUT, MODE, TBUF, true, TOPV, NPF> : conv_wsp_kernel<CIN, COUT, MODE, TBUF, true, false, NPF>)
                       : (top ? conv_wsp_kernel<CIN, COUT, MODE, TBUF, false, TOPV, NPF> : conv_wsp_kernel<CIN, COUT, MODE, TBUF, false, false, NPF>);
    { const int rc = ensure_dyn_lds(reinterpret_cast<const void*>(kern), lds_all); if (rc) return rc; }
    if (dry) return SBC_OK;
    hipLaunchKernelGGL(kern, dim3(8 * wk.wgs_per_xcd), dim3(256 * (COUT / 32 + 1)), lds_all, stream, p, wk);
    SBC_CHECK_HIP(hipGetLastError());
    return SBC_OK;
}
template <int CIN, int COUT, int MODE>
static int launch_wsp_w(const ConvParams& p, hipStream_t stream, bool dry, unsigned long long* dbg) {
    const int chunks = ((128 + 2 * p.W) * (CIN / 4) + 255) / 256;   // request registers (16 bytes each) of a memory-role thread
    if (chunks <= 5) return launch_wsp<CIN, COUT, MODE, 5>(p, stream, dry, dbg);
    if (chunks <= 8) return launch_wsp<CIN, COUT, MODE, 8>(p, stream, dry, dbg);
    if (chunks <= 10 && CIN == 64) return launch_wsp<CIN, COUT, MODE, 10>(p, stream, dry, dbg);
    return 1;
}

// SBC_OK after launching, 1 when the layer is not eligible (conv_wx3.hip then takes it), < 0 on errors
int launch_conv_wsp(const ConvParams& p, int cin, int cout, hipStream_t stream, bool dry) {
    static const bool off = getenv("SBC_NO_WSP") != nullptr;                       // A/B aid
    if (off) return 1;
    const int HW = p.H * p.W;
    if (p.dil != 1 || p.hsh < 1 || p.wsh < 1 || HW < 256 || HW % 128 || 128 % (2 * p.W)) return 1;
    if ((p.flags & (SBC_EPI_UP | SBC_EPI_POOL | SBC_EPI_ELUGRAD)) || p.res2) return 1;
    if ((p.flags & SBC_EPI_MOMENTS_OUT) && cout != 32) return 1;
    // enough tiles for the walk to pipeline: at least four per workgroup (SBC_WSP_MIN_TILES: test aid, any count is valid)
    static const int min_tiles = getenv("SBC_WSP_MIN_TILES") ? atoi(getenv("SBC_WSP_MIN_TILES")) : 4 * 256;
    if (p.total_px / 128 < min_tiles) return 1;
    unsigned long long* dbg = nullptr;
#ifdef SBC_WSP_TIMING
    dbg = reinterpret_cast<unsigned long long*>(const_cast<float*>(p.up));
#endif
    const int mode = (p.flags & SBC_CONV_F16W) ? 1 : (p.flags & SBC_CONV_F16X2) ? 2 : 0;
    if (mode == 0) return 1;                                                       // (the bf16x3 mode stays with conv_wx3.hip)
    if (cin == 32 && cout == 32) return mode == 1 ? launch_wsp_w<32, 32, 1>(p, stream, dry, dbg) : launch_wsp_w<32, 32, 2>(p, stream, dry, dbg);
    if (cin == 64 && cout == 64) return mode == 1 ? launch_wsp_w<64, 64, 1>(p, stream, dry, dbg) : launch_wsp_w<64, 64, 2>(p, stream, dry, dbg);
    return 1;
}

}  // namespace sbc

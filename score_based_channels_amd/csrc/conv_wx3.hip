// 3x3 stride-1 convolution (padding 1, no dilation) by Winograd F(2x2, 3x3) with the 16 element-wise products on the
// bf16 matrix cores of gfx950, fp32 in / fp32 out / fp32 accumulate.
//
// Same decomposition, work split, T-plane exchange and epilogue as conv_wino.hip (read that header first); only the 16
// GEMMs M[xi][nu] = V[xi][nu] U[xi][nu] differ.  The fp32 MFMA of conv_wino.hip runs on the vector ALU, so its 16 MFMAs
// per step serialise with the transform adds, the staging and the epilogue of every wave on the SIMD.  Here each fp32
// operand is split exactly into three bf16 terms (conv_x3.hip): the transformed input V = B^T d B is formed in fp32
// from the staged fp32 tile exactly as before and split in registers, the transformed filter U = G g G^T is split on
// the host (sbc_pack_conv_weight_winograd_split), and a step of 16 input channels issues 4 nu x 6 =
// 24 v_mfma_f32_32x32x16_bf16 (768 matrix-core cycles) where conv_wino.hip needs 32 fp32 MFMAs (2048 vector-ALU
// cycles) -- and the vector ALU is free meanwhile for the ~240 transform + split instructions of the next step.
// Because the split happens AFTER the input transform, the LDS tile stays fp32 ([pixel][CIN + 4], 144 B per pixel at
// 32 channels) instead of the three bf16 planes of conv_x3.hip.
#include <stdlib.h>
#include "conv_common.h"
#ifdef SBC_WITH_WSP   // tools/build_variant.sh wsp conv_wx3.hip -DSBC_WITH_WSP: the role-split experiment takes the layers it can
namespace sbc { int launch_conv_wsp(const ConvParams& p, int cin, int cout, hipStream_t stream, bool dry); }
#include "../../tools/experiments/conv_wsp.hip"
#endif

namespace sbc {

constexpr int WX3_TS = 32;

#ifdef SBC_WX3_TIMING   // tuning aid (tools/prof_conv.py WX3_TIMING=1): 100 MHz timestamps of wave 0 of every workgroup through p.up
#define WT_MARK(k) do { if (tid == 0) wt[k] = wall_clock64(); } while (0)
#else
#define WT_MARK(k) do { } while (0)
#endif

// WPE: waves per SIMD the register allocation must allow (2 = 256 registers, 3 = 168).  A third resident workgroup per CU
// is worth ~20 % where the kernel fits without spilling (32 -> 32 with 128-pixel tiles); the wider variants would spill.
// TOP: instantiation tag without effect on the code -- the ngf -> ngf layers of the score network's full-resolution level
// (sbc_op.tag == 1) get their own kernel symbol, so per-symbol profiler statistics (rocprofv3 --stats) separate them from
// the same channel configuration at 32x8, and bench.py's hipEvent average of that level can be checked against them.
// NBP: output blocks (32 channels) per phase.  (One workgroup per (tile, phase) was tried for the low-resolution
// levels, where a launch has fewer tiles than the chip has CUs: slower, because staging the 128-channel tile dominates
// there and would be repeated per phase.)
// NG: wave groups of four waves (one per transform row).  With two groups the phases (output blocks) are dealt between
// them, so a 128-output-channel layer on the 8x2 level -- a launch with fewer workgroups than CUs, i.e. pure
// single-workgroup latency -- walks two phases per group instead of four, and twice the threads stage the tile.
// MODE 1 (conv_mode f16w, BASELINE config 5 "fp16 score-net weights"): the transformed filter is ONE fp16 term
// (sbc_pack_conv_weight_winograd_f16), the transformed input V is rounded to fp16 instead of split, and a transform column
// is one v_mfma_f32_32x32x16_f16 per output block instead of six bf16 MFMAs; everything else is shared.
// MODE 2 (conv_mode f16x2): fp32-class arithmetic with TWO fp16 terms per operand.  The tile is staged as x * act_scale (a
// power of two, so V = B^T d B scales exactly), V is split in registers as h = fp16(V), l = fp16(V - h) -- three vector
// instructions per pair of values where the exact bf16 split needs nine --, U = G g G^T was scaled and split on the host
// (sbc_pack_conv_weight_winograd_f16x2), and a transform column is three v_mfma_f32_32x32x16_f16 per output block:
// (l,h) (h,l) (h,h).  The finish multiplies by descale = 1 / (act_scale * weight_scale) in the fma that adds the bias.
// MODE 0: the exact three-term bf16 split.
template <int CIN, int COUT, int MB, bool P2, int WPE, bool TOP, int NBP, int NG, int MODE>
__global__ __launch_bounds__(256 * NG, WPE) void conv_wx3_kernel(ConvParams p) {
    constexpr bool F16 = MODE == 1;
    constexpr int NTERM = MODE == 0 ? 3 : MODE;  // 16-bit terms per operand
    constexpr int TM = 128 * MB;                 // output pixels per workgroup
    constexpr int NTW = 32 * MB;                 // Winograd tiles (2x2 output blocks) per workgroup
    constexpr int S = CIN + 4;
    constexpr int KG = CIN / 16;                 // K steps: 16 input channels each
    constexpr int NBLK = COUT / 32;
    // Output blocks of a phase share the transformed + split input (two for 64 output channels; with 128 the longer
    // unrolled walk spills at two blocks and measures slower).
    constexpr int PH = NBLK / NBP;               // phases (K loops, then the outputs of NBP blocks)
    static_assert(MB == 1 || NBP == 1, "two tile blocks only with one output block per phase (registers)");
    // floats per (tile) row of a T plane: the 32 channels, unpadded -- the finish's ds_read_b128 (thread = tile, channel quad) is
    // then conflict-free in the instruction's four 16-lane groups; the 36 of conv_wino.hip's layout made two lanes of a group meet
    constexpr int TS = WX3_TS;
    constexpr bool MOM = (COUT == 32 || COUT == 64) && MB == 1 && NG == 1;   // variants that can write tile moments (SBC_EPI_MOMENTS_OUT)
    constexpr int NTHREADS = 256 * NG;           // staging; everything after it works per group of 256
    static_assert(NG == 1 || (COUT / 32 / NBP) % NG == 0, "phases must divide evenly between the wave groups");
    constexpr int NPF_FULL = ((TM + 32) * (CIN / 4) + NTHREADS - 1) / NTHREADS;
    constexpr int NPF = NPF_FULL <= 10 ? NPF_FULL : 10;
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int tid = threadIdx.x, lane = tid & 63, gtid = tid & 255;
    const int xi = __builtin_amdgcn_readfirstlane((tid >> 6) & 3);   // transform row of this wave
    const int grp = __builtin_amdgcn_readfirstlane(tid >> 8);        // wave group
    const int H = p.H, W = p.W, HW = H * W;
    const Dims<P2> dm{H, W, HW, p.hsh, p.wsh};
    const int khalf = 8 * (lane >> 5);
    const int col = lane & 31, rhalf = 4 * (lane >> 5);

    const TileGeom g = tile_geom(xcd_tile(blockIdx.x, gridDim.x), TM, p.B, dm, 1);
    float descale = 1.f, act_scale = 1.f;
#ifdef SBC_WX3_TIMING
    unsigned long long wt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    WT_MARK(0);
    {
        StageScale ss{1.f, 0.f};
        StageScale* const ssp = MODE == 2 ? &ss : nullptr;
        int sflags = p.flags;                                             // prologue flags of the staging calls
        if constexpr (MODE == 2) {
            const float4 tr = f16x2_trailer(p.wpk, 16 * KG * NBLK * NTERM);
            ss.scale = act_scale = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, tr.x)));
            descale = tr.y;
            if (__builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, tr.w)) != 0) sflags |= SBC_PRO_ELU_ACC;   // calibration: small inputs
        }
        float4 pf[NPF];
        stage_issue<CIN, NTHREADS, NPF>(pf, p.in, g, W, tid);
        WT_MARK(1);
        // InstanceNorm++ statistics of the tile's samples through LDS (behind the staged tile and the T planes)
        float* st_lds = lds + p.stats_off;
        bool direct = false;
        if ((p.flags & SBC_PRO_NORM) && !g.multi) {
            // one sample per tile: statistics straight into registers, no LDS copy, no barrier (tile.h)
            const RegStats rs = load_reg_stats<CIN, NTHREADS>(p.stats, g, tid);
            stage_commit_reg<CIN, NTHREADS, NPF>(lds, pf, p.in, rs, sflags, g, W, tid, ssp);
            direct = true;
        } else if (p.flags & SBC_PRO_NORM_SELF) {
            // whole samples per tile: the statistics are computed here, `stats` = the norm's parameters (tile.h; ends in a barrier)
            self_stats_to_lds<CIN, NTHREADS, TM, P2>(st_lds, p.in, p.stats, g, dm, tid);
        } else if (p.flags & SBC_PRO_NORM) {
            stage_stats_to_lds<CIN, NTHREADS, P2>(st_lds, p.stats, g, dm, tid);
            __syncthreads();
        }
        if (!direct) stage_commit<CIN, NTHREADS, NPF, P2>(lds, pf, p.in, st_lds, sflags, g, dm, tid, 0, ssp);
        if constexpr (MODE == 2) f16x2_range_report(ss.amax, act_scale, p.range_flag, p.calib);
    }
    WT_MARK(2);
    // T planes [xi][b][tile][TS]: overlay the staged tile when there is a single output block, else live behind it
    float* const tl = (NBLK == 1 ? lds : lds + (size_t)(g.multi ? TM + 1 : TM + 2 * W + 1) * S) + (size_t)grp * 8 * NTW * TS;

    // B^T rows: xi=0: d0 - d2, xi=1: d1 + d2, xi=2: d2 - d1, xi=3: d1 - d3   ->  R = d[ia] + sgn * d[ib]
    const int ia = xi == 0 ? 0 : xi == 2 ? 2 : 1;
    const int ib = xi == 0 ? 2 : xi == 1 ? 2 : xi == 2 ? 1 : 3;
    const float sgn = xi == 1 ? 1.f : -1.f;

    // per lane: LDS offsets of the 2 x 4 patch pixels of its tile (lane & 31) in each tile block
    const int Wt = W >> 1;                                            // tiles per image row
    const int r0 = dm.div_w(g.p0);                                    // first output row of the workgroup (even)
    const int zoff = g.nps * S + khalf;
    int off[MB][2][4];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        const int t = mb * 32 + (lane & 31);
        const int tr = P2 ? t >> (p.wsh - 1) : t / Wt, tc = t - tr * Wt;
        const int grow = r0 + 2 * tr;                                 // even output row of the tile
        const int h = dm.mod_h(grow);
        const bool tile_ok = grow < p.B * H;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int ii = k == 0 ? ia : ib;
            const int hh = h - 1 + ii;
            const bool rok = tile_ok && hh >= 0 && hh < H;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int ww = 2 * tc - 1 + j;
                off[mb][k][j] = (rok && ww >= 0 && ww < W) ? ((grow - 1 + ii - g.rs0) * W + ww) * S + khalf : zoff;
            }
        }
    }
    __syncthreads();                                                  // staged tile visible
    WT_MARK(3);

    for (int ph = grp; ph < PH; ph += NG) {
        f32x16 T[MB][NBP][2];
        f32x16 acc[NBP][4];
        // split U: [(xi*4 + nu)][kg][nb][term][lane] 16-byte fragments
        const uint4* wp = reinterpret_cast<const uint4*>(p.wpk) + ((size_t)(xi * 4) * KG * NBLK + ph * NBP) * NTERM * 64 + lane;
        auto u_frag = [&](int nu, int kg, int q, int t) { return wp[((size_t)((nu * KG + kg) * NBLK + q) * NTERM + t) * 64]; };
        // The filter fragments are requested D = SETS - 1 transform columns ahead of the MFMAs that consume them (a
        // ring of SETS statically indexed register sets over the sequence g = step * 4 + nu), so their L2 latency hides
        // behind the MFMAs and splits in between: three columns ahead with one output block per phase, one column
        // ahead with two (twice the MFMAs per column; 48 registers either way).
                // (round 3, f16x2: rings of 3 / 4 sets with two blocks, 4 / 6 with one at three waves per SIMD: within 1-3 % either way)
        constexpr int SETS = (NBP == 1 && WPE < 4) ? 4 : 2, D = SETS - 1, NSEQ = MB * KG * 4;
        uint4 uB[SETS][NBP][NTERM];                                   // 8 x 16-bit fragments (bf16 terms or fp16)
        auto u_load = [&](int gq) {                                   // gq is a compile-time constant at every call
            const int gg = gq % NSEQ, nu_g = gg & 3, kg_g = (gg >> 2) % KG;
#pragma unroll
            for (int q = 0; q < NBP; ++q)
#pragma unroll
                for (int t = 0; t < NTERM; ++t) uB[gq % SETS][q][t] = u_frag(nu_g, kg_g, q, t);
        };
#pragma unroll
        for (int gq = 0; gq < D; ++gq) u_load(gq);
#pragma unroll
        for (int s = 0; s < MB * KG; ++s) {
            const int mb = s / KG, kg = s % KG;
            if (kg == 0) {
#pragma unroll
                for (int q = 0; q < NBP; ++q)
#pragma unroll
                    for (int nu = 0; nu < 4; ++nu)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[q][nu][r] = 0.f;
            }
            // rows of B^T d for this lane's tile: R_j = d[ia][j] + sgn * d[ib][j], 8 channels each
            float R[4][8];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float* qa = lds + off[mb][0][j] + kg * 16;
                const float* qb = lds + off[mb][1][j] + kg * 16;
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
                const int gq = s * 4 + nu, cur = gq % SETS;
                u_load(gq + D);                                       // wraps to a harmless re-read at the very end
                // columns of B: nu=0: R0 - R2, nu=1: R1 + R2, nu=2: R2 - R1, nu=3: R1 - R3; then the exact split
                float v[8];
#pragma unroll
                for (int c = 0; c < 8; ++c)
                    v[c] = nu == 0 ? R[0][c] - R[2][c] : nu == 1 ? R[1][c] + R[2][c]
                         : nu == 2 ? R[2][c] - R[1][c] : R[1][c] - R[3][c];
                if constexpr (F16) {
                    f16x8 vf;
#pragma unroll
                    for (int c = 0; c < 8; ++c) vf[c] = (_Float16)v[c];
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int q = 0; q < NBP; ++q)
                        acc[q][nu] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, __builtin_bit_cast(f16x8, uB[cur][q][0]),
                                                                            acc[q][nu], 0, 0, 0);
                } else if constexpr (MODE == 2) {
                    uint4 vhu, vlu;
                    {
                        uint2 ha, la, hb, lb;                     // (the tile was staged as x * act_scale: tile.h scale_stage)
                        split_f16x2_unit(make_float4(v[0], v[1], v[2], v[3]), ha, la);
                        split_f16x2_unit(make_float4(v[4], v[5], v[6], v[7]), hb, lb);
                        vhu = make_uint4(ha.x, ha.y, hb.x, hb.y);
                        vlu = make_uint4(la.x, la.y, lb.x, lb.y);
                    }
                    split_f16x2_settle(vhu, vlu);                  // wait states before the matrix instructions read the terms (tile.h)
                    const f16x8 vh = __builtin_bit_cast(f16x8, vhu), vl = __builtin_bit_cast(f16x8, vlu);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int q = 0; q < NBP; ++q) {
                        const f16x8 uh = __builtin_bit_cast(f16x8, uB[cur][q][0]),
                                    ul = __builtin_bit_cast(f16x8, uB[cur][q][NTERM > 1 ? 1 : 0]);
                        acc[q][nu] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, uh, acc[q][nu], 0, 0, 0);
                        acc[q][nu] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, ul, acc[q][nu], 0, 0, 0);
                        acc[q][nu] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, uh, acc[q][nu], 0, 0, 0);
                    }
                } else {
                    bf16x8 vh, vm, vl;
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        const __bf16 h = (__bf16)v[c];
                        const float r1 = v[c] - (float)h;
                        const __bf16 m = (__bf16)r1;
                        vh[c] = h; vm[c] = m; vl[c] = (__bf16)(r1 - (float)m);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int q = 0; q < NBP; ++q) {
                        const bf16x8 uh = __builtin_bit_cast(bf16x8, uB[cur][q][0]),
                                     um = __builtin_bit_cast(bf16x8, uB[cur][q][NTERM > 1 ? 1 : 0]),
                                     ul = __builtin_bit_cast(bf16x8, uB[cur][q][NTERM > 2 ? 2 : 0]);
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
            if (kg == KG - 1) {
                // A^T = [[1, 1, 1, 0], [0, 1, -1, -1]] applied over nu
#pragma unroll
                for (int q = 0; q < NBP; ++q)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        T[mb][q][0][r] = (acc[q][0][r] + acc[q][1][r]) + acc[q][2][r];
                        T[mb][q][1][r] = (acc[q][1][r] - acc[q][2][r]) - acc[q][3][r];
                    }
            }
        }

        WT_MARK(4);
        if (NBLK == 1) __syncthreads();           // all waves are done with the staged tile (T planes overlay it)
#pragma unroll
        for (int q = 0; q < NBP; ++q) {
            const int nb = ph * NBP + q;          // output-channel block
            // T planes of this output block -> LDS [xi][b][tile][TS]
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    float* e = tl + ((size_t)((xi * 2 + b) * NTW + mb * 32 + rhalf)) * TS + col;
                    const f32x16 tv = T[mb][q][b];
#pragma unroll
                    for (int r = 0; r < 16; ++r) e[((r & 3) + 8 * (r >> 2)) * TS] = tv[r];
                }
            __syncthreads();
            WT_MARK(5);
            // finish: one (tile, channel quad) per thread and round
            float4 yk[4];                                                 // this thread's four outputs (SBC_EPI_MOMENTS_OUT)
#pragma unroll
            for (int i = 0; i < 4; ++i) yk[i] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 1
            for (int task = gtid; task < NTW * 8; task += 256) {
                const int t = task >> 3, c4 = task & 7;
                const int co = nb * 32 + c4 * 4;
                float4 y[2][2];                                       // [a][b]
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    float4 tx[4];
#pragma unroll
                    for (int x = 0; x < 4; ++x)
                        tx[x] = *reinterpret_cast<const float4*>(tl + ((size_t)((x * 2 + b) * NTW + t)) * TS + c4 * 4);
                    y[0][b] = make_float4((tx[0].x + tx[1].x) + tx[2].x, (tx[0].y + tx[1].y) + tx[2].y,
                                          (tx[0].z + tx[1].z) + tx[2].z, (tx[0].w + tx[1].w) + tx[2].w);
                    y[1][b] = make_float4((tx[1].x - tx[2].x) - tx[3].x, (tx[1].y - tx[2].y) - tx[3].y,
                                          (tx[1].z - tx[2].z) - tx[3].z, (tx[1].w - tx[2].w) - tx[3].w);
                }
                const int tr = P2 ? t >> (p.wsh - 1) : t / Wt, tc = t - tr * Wt;
                const int grow = r0 + 2 * tr;
                if (grow >= p.B * H) continue;
                if (MODE == 2) {
                    // descale (an exact power of two) in the same rounding as the bias add
                    const float4 bv = p.bias ? *reinterpret_cast<const float4*>(p.bias + co) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int a = 0; a < 2; ++a)
#pragma unroll
                        for (int b = 0; b < 2; ++b) {
                            y[a][b].x = fmaf(y[a][b].x, descale, bv.x); y[a][b].y = fmaf(y[a][b].y, descale, bv.y);
                            y[a][b].z = fmaf(y[a][b].z, descale, bv.z); y[a][b].w = fmaf(y[a][b].w, descale, bv.w);
                        }
                } else if (p.bias) {
                    const float4 bv = *reinterpret_cast<const float4*>(p.bias + co);
#pragma unroll
                    for (int a = 0; a < 2; ++a)
#pragma unroll
                        for (int b = 0; b < 2; ++b) {
                            y[a][b].x += bv.x; y[a][b].y += bv.y; y[a][b].z += bv.z; y[a][b].w += bv.w;
                        }
                }
                if (p.flags & SBC_EPI_POOL) {
                    // ((((0 + a) + b) + c) + d) / 4 with a=[0::2,0::2] b=[1::2,0::2] c=[0::2,1::2] d=[1::2,1::2]
                    float4 v;
                    v.x = (((y[0][0].x + y[1][0].x) + y[0][1].x) + y[1][1].x) * 0.25f;
                    v.y = (((y[0][0].y + y[1][0].y) + y[0][1].y) + y[1][1].y) * 0.25f;
                    v.z = (((y[0][0].z + y[1][0].z) + y[0][1].z) + y[1][1].z) * 0.25f;
                    v.w = (((y[0][0].w + y[1][0].w) + y[0][1].w) + y[1][1].w) * 0.25f;
                    const int n = dm.div_h(grow), ho = (grow - n * H) >> 1;
                    const size_t o = ((size_t)(n * (H >> 1) + ho) * Wt + tc) * COUT + co;
                    if (p.res1) {
                        const float4 rr = ld_stream(p.res1 + o);
                        v.x = rr.x + v.x; v.y = rr.y + v.y; v.z = rr.z + v.z; v.w = rr.w + v.w;
                    }
                    st_stream(p.out + o, v);
                    continue;
                }
                size_t o[2][2];
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b) o[a][b] = ((size_t)(grow + a) * W + 2 * tc + b) * COUT + co;
                if (p.flags & SBC_EPI_ELUGRAD) {
                    // reverse pass (adjoint convolution): times ELU'(forward input), res2 = that input (conv_epilogue.h)
#pragma unroll
                    for (int a = 0; a < 2; ++a)
#pragma unroll
                        for (int b = 0; b < 2; ++b) {
                            const float4 xv = ld_stream(p.res2 + o[a][b]);
                            y[a][b].x *= elu_grad1(xv.x); y[a][b].y *= elu_grad1(xv.y);
                            y[a][b].z *= elu_grad1(xv.z); y[a][b].w *= elu_grad1(xv.w);
                        }
                }
                if (p.res1) {
                    float4 rr[2][2];
#pragma unroll
                    for (int a = 0; a < 2; ++a)
#pragma unroll
                        for (int b = 0; b < 2; ++b) rr[a][b] = ld_stream(p.res1 + o[a][b]);
                    if (p.flags & SBC_EPI_RES1_ELU) {
#pragma unroll
                        for (int a = 0; a < 2; ++a)
#pragma unroll
                            for (int b = 0; b < 2; ++b) rr[a][b] = elu4_acc(rr[a][b]);
                    }
                    if (p.res2 && !(p.flags & SBC_EPI_ELUGRAD)) {
#pragma unroll
                        for (int a = 0; a < 2; ++a)
#pragma unroll
                            for (int b = 0; b < 2; ++b) {
                                const float4 r2 = ld_stream(p.res2 + o[a][b]);
                                rr[a][b].x = r2.x + rr[a][b].x; rr[a][b].y = r2.y + rr[a][b].y;
                                rr[a][b].z = r2.z + rr[a][b].z; rr[a][b].w = r2.w + rr[a][b].w;
                            }
                    }
#pragma unroll
                    for (int a = 0; a < 2; ++a)
#pragma unroll
                        for (int b = 0; b < 2; ++b) {
                            y[a][b].x += rr[a][b].x; y[a][b].y += rr[a][b].y;
                            y[a][b].z += rr[a][b].z; y[a][b].w += rr[a][b].w;
                        }
                }
                if (p.flags & SBC_EPI_UP) {
                    // F.interpolate(bilinear, align_corners=True) of `up` added on top (MSFBlock, layers.py:182-183)
                    const float sh = H > 1 ? (float)(p.up_h - 1) / (float)(H - 1) : 0.f;
                    const float sw = W > 1 ? (float)(p.up_w - 1) / (float)(W - 1) : 0.f;
                    const int n = dm.div_h(grow), hrow = grow - n * H;
                    const float* u = p.up + (size_t)n * p.up_h * p.up_w * COUT + co;
#pragma unroll
                    for (int a = 0; a < 2; ++a)
#pragma unroll
                        for (int b = 0; b < 2; ++b) {
                            const float fh = sh * (float)(hrow + a), fw = sw * (float)(2 * tc + b);
                            const int h0 = min((int)fh, p.up_h - 1), w0 = min((int)fw, p.up_w - 1);
                            const int h1 = min(h0 + 1, p.up_h - 1), w1 = min(w0 + 1, p.up_w - 1);
                            const float lh1 = fh - (float)h0, lw1 = fw - (float)w0;
                            const float lh0 = 1.f - lh1, lw0 = 1.f - lw1;
                            const float4 v00 = *reinterpret_cast<const float4*>(u + (size_t)(h0 * p.up_w + w0) * COUT);
                            const float4 v01 = *reinterpret_cast<const float4*>(u + (size_t)(h0 * p.up_w + w1) * COUT);
                            const float4 v10 = *reinterpret_cast<const float4*>(u + (size_t)(h1 * p.up_w + w0) * COUT);
                            const float4 v11 = *reinterpret_cast<const float4*>(u + (size_t)(h1 * p.up_w + w1) * COUT);
                            y[a][b].x += lh0 * (lw0 * v00.x + lw1 * v01.x) + lh1 * (lw0 * v10.x + lw1 * v11.x);
                            y[a][b].y += lh0 * (lw0 * v00.y + lw1 * v01.y) + lh1 * (lw0 * v10.y + lw1 * v11.y);
                            y[a][b].z += lh0 * (lw0 * v00.z + lw1 * v01.z) + lh1 * (lw0 * v10.z + lw1 * v11.z);
                            y[a][b].w += lh0 * (lw0 * v00.w + lw1 * v01.w) + lh1 * (lw0 * v10.w + lw1 * v11.w);
                        }
                }
                if ((MOM && (p.flags & SBC_EPI_MOMENTS_OUT))) {
                    // no statistics launch will read this tensor back in before its consumer does: store it cacheable, so
                    // that the consumer finds it in the last-level cache
#pragma unroll
                    for (int a = 0; a < 2; ++a)
#pragma unroll
                        for (int b = 0; b < 2; ++b) *reinterpret_cast<float4*>(p.out + o[a][b]) = y[a][b];
                } else {
#pragma unroll
                    for (int a = 0; a < 2; ++a)
#pragma unroll
                        for (int b = 0; b < 2; ++b) st_stream(p.out + o[a][b], y[a][b]);
                }
                if constexpr (MOM) {
                    yk[0] = y[0][0]; yk[1] = y[0][1]; yk[2] = y[1][0]; yk[3] = y[1][1];
                }
            }
            if constexpr (MOM) {
                if (p.flags & SBC_EPI_MOMENTS_OUT) {
                    // tile moments of the output for the InstanceNorm++ that reads it next (tile.h)
                    __shared__ __attribute__((aligned(16))) float red[8 * 8 * 8];
                    tile_moments_out32(yk, red, p.pm_out + ((size_t)(g.p0 >> 7) * COUT + nb * 32) * 2, gtid);      // [tile][COUT][2]
                }
            }
            // The T planes of this GROUP are rewritten for its next block: every wave of the group must be done reading
            // them.  The condition depends only on (q, ph - grp), so both wave groups execute the same barrier sequence
            // (a workgroup barrier counts all 256 * NG threads).
            if (q + 1 < NBP || ph + NG < PH) __syncthreads();
        }
    }
#ifdef SBC_WX3_TIMING
    WT_MARK(6);
    if (tid == 0 && p.up && !(p.flags & SBC_EPI_UP)) {
        unsigned long long* d = reinterpret_cast<unsigned long long*>(const_cast<float*>(p.up)) + (size_t)blockIdx.x * 8;
        for (int k = 0; k < 7; ++k) d[k] = wt[k];
        d[7] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | __builtin_amdgcn_s_getreg((31 << 11) | 4);
    }
#endif
}

// ------------------------------------------------------------------------------------------------ dispatch
template <int CIN, int COUT, int MB, int F16>
static int launch_wx3(const ConvParams& p, hipStream_t stream, bool dry) {
    SBC_REQUIRE(!(p.flags & SBC_EPI_MOMENTS_OUT) || (MB == 1 && (COUT == 32 || COUT == 64)),
                "conv_wx3: tile moments are written by the 128-pixel variants with 32 / 64 output channels only");
    constexpr int TM = 128 * MB;
    constexpr int S = CIN + 4;
    constexpr int NBLK = COUT / 32;
    const int HW = p.H * p.W;
    const bool multi = TM >= HW;
    const size_t staged = (size_t)(multi ? TM + 1 : TM + 2 * p.W + 1) * S * sizeof(float);
    // two wave groups where a layer has two or four output blocks and the launch cannot fill the chip anyway (each group
    // then walks half of the blocks: one phase of two blocks for 128 output channels, one block for 64)
    constexpr int NGMAX = ((NBLK == 4 || NBLK == 2) && MB == 1) ? 2 : 1;
    const int ntiles = (p.total_px + TM - 1) / TM;
    // (tile moments come out of the one-group variants: a launch that writes them keeps one group even when it is small)
    const int ng = (NGMAX == 2 && ntiles <= (NBLK == 4 ? 512 : 256) && !(p.flags & SBC_EPI_MOMENTS_OUT)) ? 2 : 1;
    const size_t tplanes = (size_t)8 * 32 * MB * WX3_TS * sizeof(float) * ng;
    const size_t lds = NBLK == 1 ? max(staged, tplanes) : staged + tplanes;
    // + the statistics of the samples of a tile: [samples][3][CIN] floats
    const size_t nsamp = multi ? TM / HW : 1;
    const size_t stats_off = lds / sizeof(float);
    SBC_REQUIRE(!(p.flags & SBC_PRO_NORM_SELF) || multi, "conv_wx3: SBC_PRO_NORM_SELF needs tiles of whole samples (H*W <= %d)", TM);
    const size_t lds_all = lds + ((p.flags & SBC_PRO_NORM) ? nsamp * (3 * CIN + 2) * sizeof(float) : 0);
    if (lds_all > 160 * 1024) return 1;
    // 32 -> 32 with 128-pixel tiles: three waves per SIMD (168 registers), filter ring three columns deep.  (Four waves at 128
    // registers with a one-column ring measured 4 % faster per launch, but only with the packed-fp32 instructions the build
    // no longer allows -- see the Makefile; without them that variant spills.)
    // (the two-term fp16 mode needs fewer registers -- no middle term, a two-set filter ring -- and fits four waves: 126 VGPRs)
        constexpr int WPE = (CIN == 32 && COUT == 32 && MB == 1) ? (F16 == 2 ? 4 : 3) : 2;
    constexpr int NBP_BIG = NBLK == 2 ? 2 : 1;
    const bool top = (CIN == 32 && COUT == 32) && p.top;
    auto kern = ng == 2 ? conv_wx3_kernel<CIN, COUT, MB, true, 2, false, (NBLK == 4 ? 2 : 1), NGMAX, F16>
              : top     ? conv_wx3_kernel<CIN, COUT, MB, true, WPE, (CIN == 32 && COUT == 32), NBP_BIG, 1, F16>
                        : conv_wx3_kernel<CIN, COUT, MB, true, WPE, false, NBP_BIG, 1, F16>;
    { const int rc = ensure_dyn_lds(reinterpret_cast<const void*>(kern), lds_all); if (rc) return rc; }
    if (dry) return SBC_OK;
    ConvParams q = p;
    q.stats_off = (int)stats_off;
    hipLaunchKernelGGL(kern, dim3(ntiles), dim3(256 * ng), lds_all, stream, q);
    SBC_CHECK_HIP(hipGetLastError());
    return SBC_OK;
}

template <int CIN, int COUT, int MB>
static int launch_wx3_mode(const ConvParams& p, hipStream_t stream, bool dry) {
    return (p.flags & SBC_CONV_F16W)    ? launch_wx3<CIN, COUT, MB, 1>(p, stream, dry)
           : (p.flags & SBC_CONV_F16X2) ? launch_wx3<CIN, COUT, MB, 2>(p, stream, dry)
                                        : launch_wx3<CIN, COUT, MB, 0>(p, stream, dry);
}

template <int CIN, int COUT>
static int launch_wx3_sized(const ConvParams& p, hipStream_t stream, bool dry) {
    const int HW = p.H * p.W;
    auto fits = [&](int tm) { return tm % (2 * p.W) == 0 && (HW % tm == 0 || tm % HW == 0); };
    // 128-pixel tiles: more resident workgroups beat the smaller halo overhead of 256-pixel tiles (32 -> 32 at 64x16:
    // 205 us with three 128-pixel workgroups per CU against 227 us with two 256-pixel ones); 256 only when 128 does
    // not tile the image
    constexpr bool mb2_ok = (COUT == 32);
    static const bool force2 = getenv("SBC_WX3_MB2") != nullptr;                   // tuning aid
    if constexpr (mb2_ok) {
        if (force2 && fits(256)) return launch_wx3_mode<CIN, COUT, 2>(p, stream, dry);
    }
    if (fits(128)) return launch_wx3_mode<CIN, COUT, 1>(p, stream, dry);
    if constexpr (mb2_ok) {
        if (fits(256)) return launch_wx3_mode<CIN, COUT, 2>(p, stream, dry);
    }
    return 1;
}

int launch_conv_wx3(const ConvParams& p, int cin, int cout, hipStream_t stream, bool dry) {
    // power-of-two images with even sides only (every level the score network produces for Nt, Nr in {16, 64, 256})
    if (p.dil != 1 || p.hsh < 1 || p.wsh < 1) return 1;
#ifdef SBC_WITH_WSP
    { const int rc = launch_conv_wsp(p, cin, cout, stream, dry); if (rc <= 0) return rc; }
#endif
    const int key = cin * 1000 + cout;
    switch (key) {
        case 32 * 1000 + 32: return launch_wx3_sized<32, 32>(p, stream, dry);
        case 32 * 1000 + 64: return launch_wx3_sized<32, 64>(p, stream, dry);
        case 64 * 1000 + 64: return launch_wx3_sized<64, 64>(p, stream, dry);
        case 64 * 1000 + 32: return launch_wx3_sized<64, 32>(p, stream, dry);
        case 64 * 1000 + 128: return launch_wx3_sized<64, 128>(p, stream, dry);
        case 128 * 1000 + 128: return launch_wx3_sized<128, 128>(p, stream, dry);
        case 128 * 1000 + 64: return launch_wx3_sized<128, 64>(p, stream, dry);
        default: return 1;
    }
}

}  // namespace sbc

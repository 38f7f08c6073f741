// 64 -> 64 channel 3x3 layers of the low-resolution levels (images 8, 4 or 2 pixels wide) as a DIRECT persistent convolution:
// the layers of ncsnv2/models/layers.py:126-134 (RCU stages) and :107-117 (CRP stages) at the 32x8, 16x4 and 8x2 levels of a
// 64 x 16 channel matrix.  The Winograd kernel (conv_wx3.hip) spends half of a 64 -> 64 tile waiting for filter fragments (its
// transformed filter, 256 KB, is streamed from L2 for every 128 pixels); here a wave keeps the fragments of its 16 output channels
// -- 9 taps x 2 k-halves x 2 fp16 terms x 4 registers = 144 -- for the life of the workgroup, as conv_pair.hip does for 32
// channels, and the K loop is LDS reads and matrix instructions only: 2.25 x the matrix work of F(2x2, 3x3), none of its waiting.
//
// Two 4-wave workgroups per CU (256 registers a wave); wave w owns output channels 16 w .. + 15 of every 16-pixel unit of a tile:
//   W = 8:  tiles of 8 rows of one sample (+ a halo row above and below), 4 units of 2 rows;
//   W = 4:  one whole 16 x 4 sample, 4 units of 4 rows                                     (no halo: the rows outside are padding)
//   W = 2:  two whole 8 x 2 samples, one unit each.
// Per tile: the raw fp32 rows arrive by LDS-DMA (requested one tile ahead), are converted ([ELU] -> x act_scale -> two fp16 terms ->
// operand planes, as in conv_pair.hip), one barrier, K loop, epilogue (x descale [+ bias] [+ (ELU) res1 [+ res2]]).  The two
// workgroups of a CU drift apart, so one converts while the other multiplies.
//
// Round 6: the 64 -> 64 layers at 8-pixel rows WITH an InstanceNorm++ prologue and a tile-moment output (res2.1's two convolutions and
// res3.0.conv1 of a 64 x 16 array: 115 us each as Winograd launches at 1700 samples, this kernel's plain form 79) run here too: the
// lane's channel quad is the same for all its chunks of a tile, so the norm is three float4 of (mu, scale, shift) per tile; the output's
// (mean, M2) per 128-pixel moment tile = two 64-pixel tiles of this kernel, which a workgroup then takes back to back, merging the two
// partial moments in registers (sums over the 16 pixel lanes of a unit by DPP, over the four units in the lane; equal counts: a fixed
// order that depends on the layer's shape only).
//
// The kernel lives on the LDS: three matrix instructions consume two ds_read_b128 (2 KB of operands), so four SIMDs ask for 64 LDS
// cycles per 48 matrix cycles even without a bank conflict -- with the straightforward layout (a unit = adjacent rows, planes 256 B
// aligned, thread = raw chunk) PMC counted 57 % of the LDS cycles as conflict cycles and the LDS 75 % busy.  Hence:
//   * reads: ds_read_b128 serves lanes {kq 0: pixels 0-3, 12-15; kq 1: pixels 4-11} (and the mirror sets) in one cycle when the 16
//     pixels of a unit sit on 16 different 16-byte slots modulo 256 B.  Plane rows are PITCH slots apart, so a unit takes the rows
//     whose offsets are a bijection onto Z16:  W = 8, pitch 10: rows i and i + 4;  W = 4, pitch 6: rows i, i + 2, i + 4, i + 6;
//     W = 2, pitch 6 (not 4): eight adjacent rows.
//   * writes: ds_write_b64 serves 16 adjacent lanes per cycle, banks modulo 128 B.  Adjacent lanes take the two halves of one
//     k-group of 8 pixels whose slots differ modulo 8 (W = 8: a row; W = 4: rows r, r + 2; W = 2: four rows) -- not the 16 channel
//     quads of one pixel, which land on ONE bank (the planes are 256 B aligned): 1 cycle per 16 lanes instead of 8.  The raw reads
//     pay for it (4 pixels of a 1 KB DMA piece share their banks: 4 cycles per 16 lanes instead of 1), one read against two writes.
#include <stdlib.h>
#include <type_traits>
#include "conv_common.h"
#ifndef DP_RING
#define DP_RING 3      // operand ring depth of the K loop (4 and 5 measured: no gain)
#endif
#ifndef DP_SETPRIO
#define DP_SETPRIO 3   // wave priority inside the K loop (88.2 us against 90.8 without at 1700 x 32 x 8)
#endif

namespace sbc {

typedef float f32x4v __attribute__((ext_vector_type(4)));

// sum over the 16 lanes of a DPP row (= the 16 pixels of a unit for one k-quarter); every lane of the row gets the total
__device__ __forceinline__ float dp_row_sum16(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128 /* row_ror:8 */, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124 /* row_ror:4 */, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122 /* row_ror:2 */, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121 /* row_ror:1 */, 0xf, 0xf, false));
    return v;
}

struct DpParams {
    const float* __restrict__ in;
    float* __restrict__ out;
    const uint4* __restrict__ w;        // sbc_pack_conv_weight_f16x2 layout (64 -> 64, 3x3)
    const float* __restrict__ bias;
    const float* __restrict__ res1;
    const float* __restrict__ res2;
    const float* __restrict__ stats;    // SBC_PRO_NORM: (mu, scale, shift) [B][3][C] of the InstanceNorm++ in front of the ELU (tiles of one sample only)
    float* __restrict__ pm_out;         // SBC_EPI_MOMENTS_OUT: (mean, M2) of the output's 128-pixel tiles [B][HW / 128][C][2] (W = 8: two tiles of this kernel each)
    unsigned* __restrict__ range_flag;
    float* __restrict__ calib;          // sbc_f16x2_calibrate: amax slot of the input, else NULL
    int flags;                          // SBC_PRO_ELU, SBC_EPI_RES1_ELU
    int B, H, ntiles, tiles_per_sample, wgs_per_xcd, tiles_per_xcd;
    unsigned long long* dbg;            // SBC_PAIR_TIMING builds: per-phase cycle sums of wave 0 of every workgroup
};

#ifdef SBC_PAIR_TIMING
#define DP_MARK(k) do { const unsigned long long _t = __builtin_readcyclecounter(); pt[k] += _t - pt_last; pt_last = _t; } while (0)
#else
#define DP_MARK(k) do { } while (0)
#endif

// FULL: a tile is S whole samples (R = H): no halo rows are fetched, the plane rows above and below a sample stay zero.
// C = 32: eight waves (two output-channel groups x four unit groups, 128 registers a wave: the fragments are 72), still two
// workgroups per CU -- four waves per SIMD.
// NM: the instantiation that knows the norm prologue and the tile-moment output (64 channels, 8-pixel rows; its own symbol so that the plain
// one keeps its 248 registers: the extras spill there)
template <int C, int W, int R, int S, bool FULL, int NW, bool NM = false>
__global__ __launch_bounds__(64 * NW, (C == 32 ? 3 : 2) * NW / 4) void conv_dp_kernel(DpParams p) {
    constexpr int NTH = 64 * NW, KGS = C / 8, KH = C / 32, C4 = C / 4, NT = 2;
    constexpr int NHF = C / 16, NSUB = NW / NHF;      // 16-output-channel groups; unit groups (wave = (hf, sub))
    static_assert(C == 32 || C == 64, "32 or 64 channels");
    static_assert(W == 2 || W == 4 || W == 8 || W == 16, "image rows of 2, 4, 8 or 16 pixels");
    static_assert(FULL || S == 1, "tiles with halo rows belong to one sample");
    constexpr int NUT = S * R * W / 16;               // units per tile
    static_assert(NUT % NSUB == 0 && (W != 4 || NSUB == 1), "units must divide over the unit groups");
    constexpr int NU = NUT / NSUB;                    // units per wave: i NSUB + sub
    constexpr int RR = FULL ? R : R + 2;              // raw rows per sample
    constexpr int RP = R + 2;                         // plane rows per sample
    constexpr int WP = W == 16 ? 18 : W == 8 ? 10 : 6;   // slots per plane row (>= W + 2; see the header for the choice)
    static_assert((W == 16 && R == 8 && S == 1) || (W == 8 && R == 8 && S == 1) || (W == 4 && R == 16 && S == 1 && FULL) || (W == 2 && R == 8 && FULL),
                  "tile shapes of the header");
    constexpr int PP = 256 / C;                       // pixels of a 1 KB LDS-DMA piece
    static_assert(FULL || W % PP == 0, "a piece must not straddle clamped rows");
    constexpr int XPS = (S * RP * WP * 16 + 255) / 256 * 256;
    constexpr int RAW_BYTES = S * RR * W * C * 4;
    constexpr int NPIECE = RAW_BYTES / 1024;
    constexpr int X_OFF = RAW_BYTES;
    constexpr int MOM_OFF = X_OFF + NT * KGS * XPS;     // NM: [wave][kq][(mean, M2) x 4 channels] floats
    constexpr int NCOMBO = S * RR * W / 8 * KGS;       // conversion work items: (8 pixels, k-group); 16 lanes each
    static_assert(NCOMBO % 4 == 0 && RAW_BYTES % 1024 == 0, "whole waves of conversion work, whole DMA pieces");
    constexpr int NIT = (NCOMBO + NTH / 16 - 1) / (NTH / 16);
    extern __shared__ __attribute__((aligned(256))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hf = wave % NHF, sub = wave / NHF;       // 16-output-channel group, unit group
    const int kq = lane >> 4, c = lane & 15;
    const int H = p.H;

    // ---- filter fragments, resident for the whole launch (packed layout: conv_pair.hip)
    uint4 wf[9][KH][NT];
    {
        const int lsrc = (16 * (hf & 1) + c) + 32 * (kq & 1), nb = hf >> 1;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int kh = 0; kh < KH; ++kh)
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    wf[tap][kh][t] = p.w[(((tap * (C / 16) + 2 * kh + (kq >> 1)) * (C / 32) + nb) * NT + t) * 64 + lsrc];
    }
    const float4 tr = f16x2_trailer(reinterpret_cast<const float4*>(p.w), 9 * (C / 16) * (C / 32) * NT);
    const float scale = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, tr.x)));
    const float descale = tr.y;
    // the calibration found this layer's input below 2^-4 (fourth trailer word): ELU in its accurate form (common.h)
    const bool elu_acc = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, tr.w)) != 0;
    const bool pro_elu = (p.flags & SBC_PRO_ELU) != 0;
    const int cq = 4 * hf + kq;                        // channel quad of this lane's four outputs
    unsigned rbits = 0;

    // ---- zero the planes once: padding columns (and, for whole-sample tiles, the rows above and below) are never written again
    for (int i = tid; i < NT * KGS * XPS / 16; i += NTH) *reinterpret_cast<uint4*>(smem + X_OFF + i * 16) = make_uint4(0, 0, 0, 0);

    const int xcd = blockIdx.x & 7, jw = blockIdx.x >> 3;
    const int t_begin = xcd * p.tiles_per_xcd;
    const int t_end = min(t_begin + p.tiles_per_xcd, p.ntiles);
    auto issue_dma = [&](int tile) {
        int n, r0;
        if (FULL) { n = tile * S; r0 = 0; } else { n = tile / p.tiles_per_sample; r0 = (tile - n * p.tiles_per_sample) * R; }
#pragma unroll
        for (int k = 0; k < (NPIECE + NW - 1) / NW; ++k) {
            const int j = k * NW + wave;                                      // piece: chunks j * 64 .. + 63 of the raw tile
            if (NPIECE % NW != 0 && j >= NPIECE) continue;
            const char* sbase;
            if (FULL) {
                // S consecutive samples = one contiguous run; a piece never straddles samples: skip those past the batch
                const int s = (j * 1024) / (R * W * C * 4);
                if (S > 1 && n + s >= p.B) continue;
                sbase = reinterpret_cast<const char*>(p.in) + (size_t)n * H * W * C * 4 + (size_t)j * 1024;
            } else {
                const int ri = (j * PP) / W, within = j * PP - ri * W;         // raw row, first pixel of the piece in it
                const int grow = min(max(r0 - 1 + ri, 0), H - 1);             // rows outside the image: any row inside (zeroed below)
                sbase = reinterpret_cast<const char*>(p.in) + ((size_t)(n * H + grow) * W + within) * C * 4;
            }
            const unsigned dst = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem + j * 1024;
            // (the lane offset is made here, by hand: as a loop invariant of an eight-wave workgroup it is spilled, and its reload --
            // a scratch load -- puts an s_waitcnt vmcnt(0) between the pieces)
            unsigned keep, l16;
            asm volatile("v_mbcnt_lo_u32_b32 %1, -1, 0\n\tv_mbcnt_hi_u32_b32 %1, -1, %1\n\tv_lshlrev_b32 %1, 4, %1\n\t"
                         "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep), "=&v"(l16) : "s"(dst), "s"(sbase) : "memory");
        }
    };
    // A workgroup walks groups of GS consecutive tiles, groups wgs_per_xcd apart (GS = 2 when tile moments are written: the two 64-pixel
    // tiles of one 128-pixel moment tile; t_begin and the group bases are even then)
    const int GS = (NM && p.pm_out) ? 2 : 1;
    auto next_tile = [&](int t) {                                             // the tile this workgroup takes after tile t
        const int g = (t - t_begin) % GS;
        return g + 1 < GS ? t + 1 : t - g + p.wgs_per_xcd * GS;
    };
    int tile = t_begin + jw * GS;
    if (tile < t_end) issue_dma(tile);
#ifdef SBC_PAIR_TIMING
    unsigned long long pt[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pt_last = __builtin_readcyclecounter();
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                          // first tile (and the filter fragments) landed

    for (; tile < t_end; tile = next_tile(tile)) {
        int n, r0;
        if (FULL) { n = tile * S; r0 = 0; } else { n = tile / p.tiles_per_sample; r0 = (tile - n * p.tiles_per_sample) * R; }
        // SBC_PRO_NORM: the lane converts the same channel quad in every chunk of the tile: one (mu, scale, shift) per tile, requested here
        float4 nmu = make_float4(0.f, 0.f, 0.f, 0.f), nsc = make_float4(1.f, 1.f, 1.f, 1.f), nsh = nmu;
        const bool pro_norm = NM && p.stats != nullptr;
        if (pro_norm) {
            int tq = tid;
            asm volatile("" : "+v"(tq));
            const float* st = p.stats + (size_t)n * 3 * C + (2 * ((tq >> 4) % KGS) + (tq & 1)) * 4;
            nmu = *reinterpret_cast<const float4*>(st); nsc = *reinterpret_cast<const float4*>(st + C); nsh = *reinterpret_cast<const float4*>(st + 2 * C);
        }
        // (1) raw tile landed, for every wave; and every wave is through the previous tile's K loop (the planes are free)
        DP_MARK(0);
        asm volatile("s_barrier" ::: "memory");
        DP_MARK(1);
        // (2) convert raw -> operand planes: all of the lane's raw chunks first (one LDS round trip), then straight-line arithmetic --
        // one copy of the loop per ELU form, so that nothing branches between the chunks
        float ta = 0.f;
        auto convert = [&](auto eluc) {
            constexpr int ELU = decltype(eluc)::value;                        // 0: none, 1: exp(x) - 1 form, 2: accurate form
            // (32 channels, 168 registers: the lane's read / write addresses are recomputed for every tile from an opaque copy of the thread index --
            // hoisted out of the tile loop they are a dozen registers the 128-register budget does not have)
            int tq = tid;
            if (C == 32) asm volatile("" : "+v"(tq));
            const int half = tq & 1, pj = (tq >> 1) & 7;
            int rrow[NIT], col, kgv[NIT];
#pragma unroll
            for (int k = 0; k < NIT; ++k) {
                const int m = k * (NTH / 16) + (tq >> 4);
                const int G = m / KGS;                                         // group of 8 pixels
                kgv[k] = m % KGS;
                if (W == 16) { rrow[k] = G >> 1; col = 8 * (G & 1) + pj; }                         // half a raw row
                else if (W == 8) { rrow[k] = G; col = pj; }                                        // one raw row
                else if (W == 4) { rrow[k] = (G & 1) + 4 * (G >> 1) + 2 * (pj >> 2); col = pj & 3; }   // rows r, r + 2
                else { rrow[k] = (G >> 1) * RR + 4 * (G & 1) + (pj >> 1); col = pj & 1; }          // four rows of one sample
            }
            float4 v[NIT];
#pragma unroll
            for (int k = 0; k < NIT; ++k)
                if (NCOMBO % (NTH / 16) == 0 || k * (NTH / 16) + 4 * wave < NCOMBO)
                    v[k] = *reinterpret_cast<const float4*>(smem + ((rrow[k] * W + col) * C4 + 2 * kgv[k] + half) * 16);
#pragma unroll
            for (int k = 0; k < NIT; ++k) {
                if (NCOMBO % (NTH / 16) != 0 && k * (NTH / 16) + 4 * wave >= NCOMBO) continue;   // (wave-uniform: a wave is four items)
                const int s = rrow[k] / RR, ri = rrow[k] % RR;                 // sample of the tile, raw row of the sample
                bool inside;
                if (FULL) inside = S == 1 || n + s < p.B;
                else { const int grow = r0 - 1 + ri; inside = grow >= 0 && grow < H; }
                float4 x = v[k];
                if (pro_norm) {            // (uniform) the zero padding is the padding of the NORMALISED tensor: norm first, then the row mask
                    x.x = fmaf(x.x - nmu.x, nsc.x, nsh.x); x.y = fmaf(x.y - nmu.y, nsc.y, nsh.y);
                    x.z = fmaf(x.z - nmu.z, nsc.z, nsh.z); x.w = fmaf(x.w - nmu.w, nsc.w, nsh.w);
                }
                x.x = inside ? x.x : 0.f; x.y = inside ? x.y : 0.f; x.z = inside ? x.z : 0.f; x.w = inside ? x.w : 0.f;
                if (ELU == 1) x = elu4(x);
                if (ELU == 2) x = elu4_acc(x);
                const int prow = s * RP + (FULL ? ri + 1 : ri);
                unsigned char* dst = smem + X_OFF + kgv[k] * XPS + (prow * WP + col + 1) * 16 + half * 8;
                StageScale ss{scale, ta};
                scale_track(x, &ss);
                ta = ss.amax;
                uint2 h, l;
                split_f16x2(x, scale, h, l);
                *reinterpret_cast<uint2*>(dst) = h;
                *reinterpret_cast<uint2*>(dst + KGS * XPS) = l;
            }
        };
        if (!pro_elu) convert(std::integral_constant<int, 0>{});
        else if (!elu_acc) convert(std::integral_constant<int, 1>{});
        else convert(std::integral_constant<int, 2>{});
        pair_range_tile(ta, scale, rbits, p.calib);
        DP_MARK(2);
        lds_barrier();
        DP_MARK(3);
        // the raw copy is consumed: request the next tile of this workgroup; it flies during the K loop
        if (next_tile(tile) < t_end) issue_dma(next_tile(tile));

        // (3) residual operands: requested before the K loop, used after it.  Unit i of the tile, lane pixel c: image row (counted through
        // the tile's samples) UMR(i) + lrow, column lcol; the top-left tap of that pixel is plane row UPR(i) + lrow, slot lcol.
        // (wave (hf, sub) owns units i NSUB + sub; both maps are linear in the unit index unless W = 4, where NSUB = 1)
        auto UMR = [](int u) { return W >= 8 ? u : W == 4 ? (u & 1) + 8 * (u >> 1) : u * R; };
        auto UPR = [](int u) { return W >= 8 ? u : W == 4 ? (u & 1) + 8 * (u >> 1) : u * RP; };
        const int lrow = W == 16 ? 0 : W == 8 ? 4 * (c >> 3) : W == 4 ? 2 * (c >> 2) : c >> 1, lcol = c & (W - 1);
        const unsigned o0 = (unsigned)(((FULL ? n * H : n * H + r0) + lrow + UMR(1) * sub) * W + lcol) * C + cq * 4;
        auto DO = [&](int i) { return UMR(i * NSUB) * W * C; };
        auto valid = [&](int i) { return S == 1 || n + i * NSUB + sub < p.B; };   // (S > 1: a unit is a sample)
        // (eight-wave workgroups have 128 registers a wave and no room to hold them through the K loop: requested after it, the
        // other three waves of the SIMD cover the round trip)
        constexpr bool PRE_RES = NW == 4;
        float4 x1[NU];
        if (PRE_RES && p.res1) {
#pragma unroll
            for (int i = 0; i < NU; ++i)
                if (valid(i)) x1[i] = *reinterpret_cast<const float4*>(p.res1 + o0 + DO(i));
        }
        // (4) K loop: acc[i] = D[16 couts of this wave][16 pixels of unit i]
        f32x4v acc[NU];
        {
            const int ub0 = X_OFF + kq * XPS + ((lrow + UPR(1) * sub) * WP + lcol) * 16;
            constexpr int NS = 9 * KH * NU, D = DP_RING;
            f16x8 ring[D][NT];
            auto ld = [&](int s) {                                            // s is a compile-time constant at every call
                const int tap = s / (KH * NU), kh = (s / NU) % KH, i = s % NU;
                const int off = (UPR(i * NSUB) * WP + (tap / 3) * WP + (tap % 3)) * 16 + kh * 4 * XPS;
#pragma unroll
                for (int t = 0; t < NT; ++t) ring[s % D][t] = *reinterpret_cast<const f16x8*>(smem + ub0 + (off + t * KGS * XPS));
            };
#pragma unroll
            for (int s = 0; s < D - 1; ++s) ld(s);
            __builtin_amdgcn_s_setprio(DP_SETPRIO);
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int tap = s / (KH * NU), kh = (s / NU) % KH, i = s % NU;
#ifndef DP_PROBE_NOLDS      // timing probe (wrong results): the K loop without its LDS reads
                if (s + D - 1 < NS) ld(s + D - 1);
#endif
                const f16x8 xh = ring[s % D][0], xl = ring[s % D][1];
                const f16x8 wh = __builtin_bit_cast(f16x8, wf[tap][kh][0]), wl = __builtin_bit_cast(f16x8, wf[tap][kh][1]);
                const f32x4v c0 = (tap == 0 && kh == 0) ? f32x4v{0.f, 0.f, 0.f, 0.f} : acc[i];
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xl, c0, 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, xh, acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xh, acc[i], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __builtin_amdgcn_s_setprio(0);
        DP_MARK(4);
        float4 bias = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.bias) {
            int cqo = cq;
            asm volatile("" : "+v"(cqo));                                     // (address formed here, not carried through the tile loop)
            bias = *reinterpret_cast<const float4*>(p.bias + cqo * 4);
        }
        if (!PRE_RES && p.res1) {
#pragma unroll
            for (int i = 0; i < NU; ++i)
                if (valid(i)) x1[i] = *reinterpret_cast<const float4*>(p.res1 + o0 + DO(i));
        }
        float4 x2[NU];
        if (!NM && p.res2) {                                                  // (the NM instantiation takes no second residual operand)
#pragma unroll
            for (int i = 0; i < NU; ++i)
                if (valid(i)) x2[i] = *reinterpret_cast<const float4*>(p.res2 + o0 + DO(i));
        }
        // everything this wave has in flight -- the residuals, its pieces of the next tile's DMA -- has landed
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        DP_MARK(5);
        float4 ysum = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < NU; ++i) {
            if (!valid(i)) continue;
            float4 y = make_float4(fmaf(acc[i][0], descale, bias.x), fmaf(acc[i][1], descale, bias.y),
                                   fmaf(acc[i][2], descale, bias.z), fmaf(acc[i][3], descale, bias.w));
            if (p.res1) {
                // r = res1 [ELU];  if res2: r = res2 + r;  y = y + r     (include/sbc_hip.h: the CONV epilogue)
                float4 rr = x1[i];
                if (!NM && (p.flags & SBC_EPI_RES1_ELU)) rr = elu4_acc(rr);
                if (!NM && p.res2) { rr.x = x2[i].x + rr.x; rr.y = x2[i].y + rr.y; rr.z = x2[i].z + rr.z; rr.w = x2[i].w + rr.w; }
                y.x += rr.x; y.y += rr.y; y.z += rr.z; y.w += rr.w;
            }
#ifdef DP_PROBE_NOSTORE   // timing probe (wrong results): everything but the output stores
            if (y.x == 123456.f)
#endif
            st_out(p.out + o0 + DO(i), y);
            if (NM && p.pm_out) { acc[i] = f32x4v{y.x, y.y, y.z, y.w}; ysum.x += y.x; ysum.y += y.y; ysum.z += y.z; ysum.w += y.w; }
        }
        if constexpr (NM) {
            static_assert(!NM || (W == 8 && C == 64 && !FULL && NSUB == 1), "the wave holds a whole 64-pixel tile of its channels");
            if (p.pm_out) {
                // (mean, M2) of the lane's four channels over this 64-pixel tile: the wave holds all of it (NSUB = 1) -- units in the lane, the
                // 16 pixels of a unit across the DPP row; two tiles of a group merge into the 128-pixel moment tile (equal counts)
                float4 mean = make_float4(dp_row_sum16(ysum.x) * (1.f / 64.f), dp_row_sum16(ysum.y) * (1.f / 64.f),
                                          dp_row_sum16(ysum.z) * (1.f / 64.f), dp_row_sum16(ysum.w) * (1.f / 64.f));
                float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int i = 0; i < NU; ++i) {
                    float d;
                    d = acc[i][0] - mean.x; q.x = fmaf(d, d, q.x); d = acc[i][1] - mean.y; q.y = fmaf(d, d, q.y);
                    d = acc[i][2] - mean.z; q.z = fmaf(d, d, q.z); d = acc[i][3] - mean.w; q.w = fmaf(d, d, q.w);
                }
                q = make_float4(dp_row_sum16(q.x), dp_row_sum16(q.y), dp_row_sum16(q.z), dp_row_sum16(q.w));
                // (the first tile's moments wait in a corner of the LDS, not in eight registers across the next K loop)
                float* keep = reinterpret_cast<float*>(smem + MOM_OFF) + (wave * 4 + kq) * 8;
                if (((tile - t_begin) & 1) == 0) {
                    if (c == 0) { *reinterpret_cast<float4*>(keep) = mean; *reinterpret_cast<float4*>(keep + 4) = q; }
                } else if (c == 0) {
                    const float4 mom_mean = *reinterpret_cast<const float4*>(keep), mom_m2 = *reinterpret_cast<const float4*>(keep + 4);
                    const int jt = tile - n * p.tiles_per_sample;                 // (odd) tile of the sample: moment tile jt >> 1
                    float* o = p.pm_out + (((size_t)n * (p.tiles_per_sample >> 1) + (jt >> 1)) * C + cq * 4) * 2;
                    float d;
                    d = mean.x - mom_mean.x; *reinterpret_cast<float2*>(o) = make_float2(mom_mean.x + 0.5f * d, mom_m2.x + q.x + d * d * 32.f);
                    d = mean.y - mom_mean.y; *reinterpret_cast<float2*>(o + 2) = make_float2(mom_mean.y + 0.5f * d, mom_m2.y + q.y + d * d * 32.f);
                    d = mean.z - mom_mean.z; *reinterpret_cast<float2*>(o + 4) = make_float2(mom_mean.z + 0.5f * d, mom_m2.z + q.z + d * d * 32.f);
                    d = mean.w - mom_mean.w; *reinterpret_cast<float2*>(o + 6) = make_float2(mom_mean.w + 0.5f * d, mom_m2.w + q.w + d * d * 32.f);
                }
            }
        }
        DP_MARK(6);
    }
    if (rbits && lane == 0) atomicOr(p.range_flag, rbits);
#ifdef SBC_PAIR_TIMING
    // [wait for the raw tile, barrier, convert, barrier, dma issue + residual requests + K loop, load wait, store]
    if (tid == 0 && p.dbg)
        for (int k = 0; k < 7; ++k) atomicAdd(p.dbg + k, pt[k]);
#endif
}

template <int C, int W, int R, int S, bool FULL, int NW, bool NM = false>
static int launch_dp(const DpParams& p0, hipStream_t stream, bool dry) {
    constexpr int NT = 2, RR = FULL ? R : R + 2, RP = R + 2, WP = W == 16 ? 18 : W == 8 ? 10 : 6;
    constexpr int XPS = (S * RP * WP * 16 + 255) / 256 * 256;
    constexpr size_t lds = (size_t)S * RR * W * C * 4 + (size_t)NT * (C / 8) * XPS + (NM ? NW * 4 * 8 * 4 : 0);
    constexpr int WGPC = (C == 32 ? 3 : 2) * 4 / NW;                       // workgroups per CU (registers: 168 / 256 a wave)
    static_assert(lds * WGPC <= 160 * 1024, "LDS of the resident workgroups");
    auto kern = conv_dp_kernel<C, W, R, S, FULL, NW, NM>;
    { const int rc = ensure_dyn_lds(reinterpret_cast<const void*>(kern), lds); if (rc) return rc; }
    if (dry) return SBC_OK;
    DpParams p = p0;
    p.tiles_per_sample = FULL ? 1 : p.H / R;
    p.ntiles = FULL ? (p.B + S - 1) / S : p.B * p.tiles_per_sample;
    int dev = 0, cus = 256;
    SBC_CHECK_HIP(hipGetDevice(&dev));
    SBC_CHECK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const int gs = p.pm_out ? 2 : 1;                                       // tiles a workgroup takes back to back (conv_dp_kernel: GS)
    p.tiles_per_xcd = ((p.ntiles + 8 * gs - 1) / (8 * gs)) * gs;
    p.wgs_per_xcd = max(1, min(WGPC * persistent_cus(cus) / 8, (p.tiles_per_xcd + gs - 1) / gs));
#ifdef SBC_PAIR_TIMING
    if (getenv("SBC_DP_WGS")) p.wgs_per_xcd = max(1, min(atoi(getenv("SBC_DP_WGS")), p.tiles_per_xcd));   // probe: workgroups per XCD
#endif
    hipLaunchKernelGGL(kern, dim3(8 * p.wgs_per_xcd), dim3(64 * NW), lds, stream, p);
    SBC_CHECK_HIP(hipGetLastError());
    return SBC_OK;
}

// 1: not this kernel's layer (the caller goes on to the Winograd kernel); 0: launched; < 0: error
int launch_conv_dp(const sbc_op& op, unsigned* range_flag, hipStream_t stream, bool dry) {
    static const bool off = getenv("SBC_NO_CONV_DP") != nullptr;             // A/B aid
    static const bool off32 = getenv("SBC_NO_CONV_DP32") != nullptr;         // A/B aid: 32-channel layers only
    if (off || !(op.flags & SBC_CONV_F16X2) || !op.weight_split || op.cin != op.cout || op.ksize != 3 || op.dil != 1) return 1;
    if (op.cin != 64 && (op.cin != 32 || off32)) return 1;
    if (op.flags & (SBC_EPI_POOL | SBC_EPI_UP | SBC_EPI_ELUGRAD | SBC_PRO_NORM_SELF)) return 1;
    if (op.res2 && !op.res1) return 1;
    // a norm prologue / a tile-moment output: the 64-channel kernel at 8-pixel rows only (round 6; moment tiles are 16 rows)
    static const bool no_norm = getenv("SBC_NO_CONV_DP_NORM") != nullptr;    // A/B aid: those layers on the Winograd kernel
    if (op.flags & (SBC_PRO_NORM | SBC_EPI_MOMENTS_OUT)) {
        if (no_norm || op.cin != 64 || op.W != 8 || op.H % 16 != 0 || op.res2 || (op.flags & SBC_EPI_RES1_ELU)) return 1;
        if (((op.flags & SBC_PRO_NORM) && !op.stats) || ((op.flags & SBC_EPI_MOMENTS_OUT) && !op.aux)) return 1;
    }
    const bool w16 = op.W == 16 && op.H % 8 == 0 && op.cin == 32, w8 = op.W == 8 && op.H % 8 == 0;
    const bool w4 = op.W == 4 && op.H == 16 && op.cin == 64, w2 = op.W == 2 && op.H == 8 && op.cin == 64;
    if (!w16 && !w8 && !w4 && !w2) return 1;
    DpParams p{};
    p.in = (const float*)op.in; p.out = (float*)op.out; p.w = (const uint4*)op.weight_split;
    p.bias = (const float*)op.bias; p.res1 = (const float*)op.res1; p.res2 = (const float*)op.res2;
    p.flags = op.flags; p.B = op.B; p.H = op.H;
    p.stats = (op.flags & SBC_PRO_NORM) ? (const float*)op.stats : nullptr;
    p.pm_out = (op.flags & SBC_EPI_MOMENTS_OUT) ? (float*)op.aux : nullptr;
    p.range_flag = range_flag; p.calib = (float*)op.calib;
    p.dbg = (op.flags & SBC_EPI_MOMENTS_OUT) ? nullptr : (unsigned long long*)op.aux;
    if (op.cin == 32) return w16 ? launch_dp<32, 16, 8, 1, false, 4>(p, stream, dry) : launch_dp<32, 8, 8, 1, false, 4>(p, stream, dry);
    if (w8 && (p.stats || p.pm_out)) return launch_dp<64, 8, 8, 1, false, 4, true>(p, stream, dry);
    if (w8) return launch_dp<64, 8, 8, 1, false, 4>(p, stream, dry);
    if (w4) return launch_dp<64, 4, 16, 1, true, 4>(p, stream, dry);
    return launch_dp<64, 2, 8, 2, true, 4>(p, stream, dry);
}

}  // namespace sbc

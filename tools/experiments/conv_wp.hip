// EXPERIMENT (round 4), not in the product build.  Build it into a variant library with
//     tools/build_wp_variant.sh        (adds this file and -DSBC_WITH_WP to conv_mfma.hip; see the script)
// Result on MI355X (1700 x 32x8, ELU prologue, residual): correct in every prologue / epilogue case of tests/test_gpu_ops.py, and
// SLOWER than conv_wx3: 134 us against 122 (16x4: 42 against 33; 8x2: 24 against 11 -- the 256 KB of filter per workgroup are
// loaded for 1.7 blocks there).  Why (per-phase cycle counters, -DSBC_WP_TIMING, and PMC): one workgroup of
// sixteen lock-stepped waves per CU walks its phases -- transform (LDS), matrix instructions, exchange, conversion, finish (LDS +
// stores) -- one after the other with two barriers per 64 pixels, so the matrix pipe (0.16 busy), the LDS (conflict-free: 0.08
// against conv_wx3's 0.59 conflict cycles per access cycle) and the vector ALU (70 instructions per pixel against 45) are each
// mostly idle; conv_wx3's small independent workgroups overlap those phases between workgroups.  128 registers per wave (64 of
// them filter) leave no room to pipeline two blocks inside a wave, 160 KB of LDS none to double-buffer the exchange planes.
//
// 3x3 stride-1 convolution 64 -> 64 (padding 1) by Winograd F(2x2, 3x3) on the fp16 matrix cores of gfx950 with the transformed
// filter RESIDENT IN REGISTERS (conv_mode f16x2; round 4).
//
// conv_wx3.hip gives a wave one transform row (four of the sixteen positions): its slice of U = G g G^T for 64 -> 64 is 64 KB per
// wave, so the fragments stream from L2 through the vector cache for every 128-pixel tile, and the lone matrix wave of a SIMD
// waits for every column of them (7 of the 14 us a tile takes: DESIGN.md section 5).  Here a workgroup has SIXTEEN waves and each
// owns ONE position (xi, nu): its U[xi][nu] is 64 x 64 x two fp16 terms = 16 KB = 64 registers per lane, loaded once per launch;
// the workgroups are persistent (one per CU) and walk blocks of 16 Winograd tiles (64 output pixels):
//   stage     the block's input rows arrive in registers one block ahead (issued at the top of the previous iteration), go through
//             the prologue (InstanceNorm++ affine from a statistics table kept in LDS two blocks ahead, ELU), are multiplied by
//             the layer's act_scale and written to one of two fp32 tiles d[pixel][64 + 4] (conflict-free row padding, below);
//   transform every wave forms ITS position's V = (d[ia][ja] + sr d[ib][ja]) + sc (d[ia][jb] + sr d[ib][jb]) for the 16 tiles x 64
//             channels of the block straight in the B-operand layout of v_mfma_f32_16x16x32_f16 (lane = tile, k-group), splits it
//             into two fp16 terms and issues 4 output groups x 2 k-halves x 3 = 24 matrix instructions: M[pos][64 couts][16 tiles];
//   exchange  the sixteen M planes meet in LDS ([pos][tile][64 + 4] floats, 68 KB);
//   finish    one thread per (tile, output pixel of its 2x2 block, channel quad): A^T M A over the nine positions it needs, bias /
//             descale, residuals / pool / bilinear resize-add, 16-byte stores.
// Two workgroup barriers per block (LDS only: the prefetch in flight is not waited for).  The K loop has no global load at all.
//
// Channel order inside a 32-channel contraction block: lane (tile, kq) of the B operand reads the two 16-byte runs of channels
// kq*4 .. +3 and 16 + kq*4 .. +3 -- with the natural order (8 consecutive channels per lane) the sixteen lanes of a ds_read_b128
// service group reach only even 16-byte slots (the tiles of a row are 2 pixels = 34 slots apart), a 2-way bank conflict on every
// read of the loop that is the kernel's LDS bottleneck.  The filter fragments are permuted to match when they are loaded.
#include <stdlib.h>
#include <type_traits>
#include "../../score_based_channels_amd/csrc/conv_common.h"

namespace sbc {

typedef float f32x4v __attribute__((ext_vector_type(4)));

constexpr int WP_S = 68;                         // floats per staged pixel
constexpr int WP_ES = 68;                        // floats per (position, tile) row of the exchange planes
// floats of padding behind a staged image row, chosen so that the 16 tiles of a block (2 pixels apart along a row, 2 rows apart
// between tile rows) fall on 8 distinct even 16-byte slots per k-group parity (see the header): 2 rows must shift by 8 slots at
// W = 8 (4 tile columns), by 4 at W = 4, by 2 at W = 2
template <int W> constexpr int wp_rowpad() { return W == 8 ? 16 : W == 4 ? 24 : W == 2 ? 28 : W == 16 ? 8 : 0; }

#ifdef SBC_WP_TIMING   // tuning aid (tools/prof_conv.py WP_TIMING=1): per-phase cycle sums of wave 0 of every workgroup through p.up
#define WPT(k) do { const unsigned long long _t = __builtin_readcyclecounter(); pt[k] += _t - pt_last; pt_last = _t; } while (0)
#else
#define WPT(k) do { } while (0)
#endif

struct WpExtra {
    int n_blocks, blocks_per_xcd, wgs_per_xcd;
    int bps_sh;                                  // PARTIAL: log2(blocks per sample)
    int spb;                                     // !PARTIAL: samples per block
};

__device__ __forceinline__ void wp_barrier() {
    // LDS traffic only: the prefetched block in flight (vmcnt) must NOT be waited for here
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// PARTIAL: a block is R rows of ONE sample (H > R): R + 2 rows are staged, rows outside the sample as zeros.
// !PARTIAL: a block is `spb` whole samples (H <= R): no halo rows, taps outside a sample read the zero pixel.
template <int W, bool PARTIAL>
__global__ __launch_bounds__(1024) void conv_wp_kernel(ConvParams p, WpExtra x) {
    constexpr int CIN = 64, COUT = 64, S = WP_S, ES = WP_ES;
    constexpr int WT = W / 2, TR = 16 / WT, R = 2 * TR;       // tile columns, tile rows and output rows of a block
    constexpr int ROWS = PARTIAL ? R + 2 : R;
    constexpr int RS = W * S + wp_rowpad<W>();                // floats per staged row
    constexpr int DSZ = ROWS * RS + S;                        // one staged block + the zero pixel
    constexpr int ZOFF = ROWS * RS;
    constexpr int NCH = ROWS * W * (CIN / 4);                 // 16-byte chunks of a block
    constexpr int NPF = (NCH + 1023) / 1024;
    constexpr int WSH = W == 32 ? 5 : W == 16 ? 4 : W == 8 ? 3 : W == 4 ? 2 : 1;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* const dbuf = lds;                                  // [2][DSZ]
    float* const ex = lds + 2 * DSZ;                          // [16 positions][16 tiles][ES]
    float* const stl = ex + 16 * 16 * ES;                     // [2][spb][3][CIN]: (mu, scale, shift) of the block's samples

    const int tid = threadIdx.x, lane = tid & 63;
    const int pos = __builtin_amdgcn_readfirstlane(tid >> 6);  // transform position of this wave = tile its threads finish
    const int xi = pos >> 2, nu = pos & 3;
    const int kq = lane >> 4, tn = lane & 15;                 // k-group of the lane's operand fragments; tile (B) / output channel (A)
    const int H = p.H, hsh = p.hsh;
    const int spb = PARTIAL ? 1 : x.spb;
    const bool norm = (p.flags & SBC_PRO_NORM) != 0;

    // ---- the transformed filter of this wave's position: [output group][k-half][term], resident for the whole launch
    uint4 u[4][2][2];
    {
        const uint4* wp = reinterpret_cast<const uint4*>(p.wpk);
#pragma unroll
        for (int cg = 0; cg < 4; ++cg)
#pragma unroll
            for (int kh = 0; kh < 2; ++kh)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    // packed layout [pos][cin / 16][cout / 32][term][64 lanes][8]: lane' = cout % 32 + 32 * ((cin % 16) / 8)
                    const int lsrc = 16 * (cg & 1) + tn + 32 * (kq >> 1);
                    const uint4 f0 = wp[(((pos * 4 + 2 * kh) * 2 + (cg >> 1)) * 2 + t) * 64 + lsrc];
                    const uint4 f1 = wp[(((pos * 4 + 2 * kh + 1) * 2 + (cg >> 1)) * 2 + t) * 64 + lsrc];
                    u[cg][kh][t] = (kq & 1) ? make_uint4(f0.z, f0.w, f1.z, f1.w) : make_uint4(f0.x, f0.y, f1.x, f1.y);
                }
    }
    const float4 trl = f16x2_trailer(p.wpk, 16 * 4 * 2 * 2);
    const float act_scale = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, trl.x)));
    const float descale = trl.y;
    const bool elu = (p.flags & SBC_PRO_ELU) != 0;
    const bool elu_acc = elu && ((p.flags & SBC_PRO_ELU_ACC) || __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, trl.w)) != 0);

    // ---- LDS offsets (floats) of the four patch pixels this wave's position combines, for this lane's tile
    // B^T rows: xi=0: d0 - d2, xi=1: d1 + d2, xi=2: d2 - d1, xi=3: d1 - d3   ->  d[ia] + sr * d[ib]; the same over columns with nu
    const int ia = xi == 0 ? 0 : xi == 2 ? 2 : 1, ib = xi == 0 ? 2 : xi == 1 ? 2 : xi == 2 ? 1 : 3;
    const int ja = nu == 0 ? 0 : nu == 2 ? 2 : 1, jb = nu == 0 ? 2 : nu == 1 ? 2 : nu == 2 ? 1 : 3;
    const float sr = xi == 1 ? 1.f : -1.f, sc = nu == 1 ? 1.f : -1.f;
    int o_aa, o_ba, o_ab, o_bb;
    {
        const int tr = tn / WT, tc = tn % WT;
        auto off = [&](int i, int j) {
            const int col = 2 * tc - 1 + j;
            int row;
            bool ok = col >= 0 && col < W;
            if (PARTIAL) {
                row = 2 * tr + i;                              // buffer row 0 = the row above the block
            } else {
                row = 2 * tr - 1 + i;
                const int hs = ((2 * tr) & (H - 1)) - 1 + i;   // row inside its sample
                ok = ok && hs >= 0 && hs < H;
            }
            return (ok ? row * RS + col * S : ZOFF) + kq * 4;
        };
        o_aa = off(ia, ja); o_ba = off(ib, ja); o_ab = off(ia, jb); o_bb = off(ib, jb);
    }

    // ---- block walk: XCD x (= blockIdx % 8) owns a contiguous run of blocks, its workgroups take every wgs_per_xcd-th of them
    const int xcd = blockIdx.x & 7, jw = blockIdx.x >> 3;
    const int b_begin = xcd * x.blocks_per_xcd;
    const int b_end = min(b_begin + x.blocks_per_xcd, x.n_blocks);
    const int first = b_begin + jw;
    const int n_my = first < b_end ? (b_end - first + x.wgs_per_xcd - 1) / x.wgs_per_xcd : 0;
    if (n_my == 0) return;
    auto block_of = [&](int k) { return first + k * x.wgs_per_xcd; };

    // zero pixels of both buffers (nothing writes them afterwards); the bias behind the statistics tables
    if (tid < 2 * S) dbuf[(tid / S) * DSZ + ZOFF + (tid % S)] = 0.f;
    float* const bl = stl + 2 * spb * 3 * CIN;                // [COUT]
    if (tid < COUT) bl[tid] = p.bias ? p.bias[tid] : 0.f;

    // ---- staging.  A block's chunks are ONE contiguous run of NHWC memory (R [+ 2] image rows, or spb whole samples): chunk idx
    // of block blk is element (blk * R [- 1]) * W * 64 + idx * 4.
    constexpr int BLK_ELEMS = R * W * CIN;                    // output (= input without halo) elements of a block
    // (everything per chunk is recomputed from the thread index where it is used, behind an opaque copy of it: hipcc would otherwise
    // keep a dozen loop-invariant offsets and masks in vector registers for the whole launch, and this kernel has none to spare)
    auto opaque_tid = [&]() { int t = tid; asm volatile("" : "+v"(t)); return t; };
    // which chunks of block blk exist: the halo rows inside the sample (PARTIAL), the samples below B (!PARTIAL)
    auto chunk_ok = [&](int blk, int idx) {
        if (idx >= NCH) return false;
        if (PARTIAL) {
            const int brow = idx >> (4 + WSH);
            const int r0 = (blk & ((1 << x.bps_sh) - 1)) * R;  // first output row of the block inside its sample
            return !((brow == 0 && r0 == 0) || (brow == ROWS - 1 && r0 + R >= H));
        }
        return idx < (p.B - blk * spb) * (H * W * (CIN / 4));
    };
    float4 pf[NPF];
    auto issue = [&](int blk) {
        const float* src = p.in + (size_t)blk * BLK_ELEMS - (PARTIAL ? W * CIN : 0);
        const int t = opaque_tid();
#pragma unroll
        for (int q = 0; q < NPF; ++q)
            pf[q] = chunk_ok(blk, q * 1024 + t) ? ld_stream(src + (q * 1024 + t) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    float amax = 0.f;
    auto commit = [&](int blk, int buf) {
        float* d = dbuf + buf * DSZ;
        const float* st = stl + buf * spb * 3 * CIN;          // (the statistics rows of block k live in table k & 1, like its tile)
        const int t = opaque_tid();
#pragma unroll
        for (int q = 0; q < NPF; ++q) {
            const int idx = q * 1024 + t;
            if (idx >= NCH) continue;
            const int c4 = idx & 15, px = idx >> 4, brow = px >> WSH, col = px & (W - 1);
            float4 v = pf[q];
            if (norm) {
                const float* s3 = st + (PARTIAL ? 0 : (brow >> hsh)) * 3 * CIN + c4 * 4;
                const float4 mu = *reinterpret_cast<const float4*>(s3);
                const float4 sc4 = *reinterpret_cast<const float4*>(s3 + CIN);
                const float4 sh = *reinterpret_cast<const float4*>(s3 + 2 * CIN);
                v.x = (v.x - mu.x) * sc4.x + sh.x; v.y = (v.y - mu.y) * sc4.y + sh.y;
                v.z = (v.z - mu.z) * sc4.z + sh.z; v.w = (v.w - mu.w) * sc4.w + sh.w;
            }
            if (elu) v = elu4(v, elu_acc);
            if (!chunk_ok(blk, idx)) v = make_float4(0.f, 0.f, 0.f, 0.f);   // zero padding of the convolution, not the prologue of zeros
            amax = __builtin_fmaxf(__builtin_fmaxf(amax, __builtin_fabsf(v.x)), __builtin_fabsf(v.y));
            amax = __builtin_fmaxf(__builtin_fmaxf(amax, __builtin_fabsf(v.z)), __builtin_fabsf(v.w));
            v.x *= act_scale; v.y *= act_scale; v.z *= act_scale; v.w *= act_scale;
            *reinterpret_cast<float4*>(d + brow * RS + col * S + c4 * 4) = v;
        }
    };
    // statistics rows of a block's samples, [3][CIN] = 768 bytes each: wave s (< spb) copies the row of sample s straight into the
    // LDS table by LDS-DMA (lanes 0 .. 47, 16 bytes each; no registers held while the request flies).  By hand like conv_pair.hip:
    // M0 = LDS byte address, lane l lands at M0 + 16 l; hipcc does not count these requests, the wait below is explicit.
    auto stat_issue = [&](int blk, int tbl) {
        if (norm && pos < spb) {
            const int n = PARTIAL ? (blk >> x.bps_sh) : blk * spb + pos;
            if (n < p.B && (opaque_tid() & 63) < 48) {
                const float* sbase = p.stats + (size_t)n * 3 * CIN;
                const unsigned dst = (unsigned)(size_t)(__attribute__((address_space(3))) float*)(stl + (tbl * spb + pos) * 3 * CIN);
                unsigned keep;
                // (the lane offset from an opaque copy of the thread index: as a loop invariant it was spilled, and the reload in front
                // of this request cost an s_waitcnt vmcnt(0) -- every store of the previous block -- at the top of every iteration)
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"((opaque_tid() & 63) * 16), "s"(dst), "s"(sbase) : "memory");
            }
        }
    };

    // ---- prologue of the pipeline: statistics of blocks 0 and 1, tile of block 0
    stat_issue(block_of(0), 0);
    if (n_my > 1) stat_issue(block_of(1), 1);
    issue(block_of(0));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    commit(block_of(0), 0);
    __syncthreads();

#ifdef SBC_WP_TIMING
    unsigned long long pt[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pt_last = __builtin_readcyclecounter();
    const unsigned long long pt_start = pt_last;
#endif
    for (int k = 0; k < n_my; ++k) {
        const int blk = block_of(k);
        WPT(7);
        // (the statistics request first: the wait in front of the first barrier then leaves only the NPF tile loads in flight)
        if (k + 2 < n_my) stat_issue(block_of(k + 2), k & 1);  // table k & 1: block k's rows were used an iteration ago
        if (k + 1 < n_my) issue(block_of(k + 1));

        WPT(0);
        // ---- transform + matrix instructions on tile k & 1
        const float* d = dbuf + (k & 1) * DSZ;
        f32x4v acc[4];
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
            uint4 vh, vl;
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {                   // the lane's two 4-channel runs of this k-half
                const int co = kh * 32 + hf * 16;
                // (two reads at a time: all four would be hoisted to the top of the k-half and cost 8 more live registers)
                float4 v, r2;
                {
                    const float4 aa = *reinterpret_cast<const float4*>(d + o_aa + co), ba = *reinterpret_cast<const float4*>(d + o_ba + co);
                    v.x = fmaf(sr, ba.x, aa.x); v.y = fmaf(sr, ba.y, aa.y); v.z = fmaf(sr, ba.z, aa.z); v.w = fmaf(sr, ba.w, aa.w);
                }
                __builtin_amdgcn_sched_barrier(0);
                {
                    const float4 ab = *reinterpret_cast<const float4*>(d + o_ab + co), bb = *reinterpret_cast<const float4*>(d + o_bb + co);
                    r2.x = fmaf(sr, bb.x, ab.x); r2.y = fmaf(sr, bb.y, ab.y); r2.z = fmaf(sr, bb.z, ab.z); r2.w = fmaf(sr, bb.w, ab.w);
                }
                v.x = fmaf(sc, r2.x, v.x); v.y = fmaf(sc, r2.y, v.y); v.z = fmaf(sc, r2.z, v.z); v.w = fmaf(sc, r2.w, v.w);
                uint2 h2, l2;
                split_f16x2_unit(v, h2, l2);
                if (hf == 0) { vh.x = h2.x; vh.y = h2.y; vl.x = l2.x; vl.y = l2.y; }
                else         { vh.z = h2.x; vh.w = h2.y; vl.z = l2.x; vl.w = l2.y; }
            }
            split_f16x2_settle(vh, vl);                        // wait states before the matrix instructions read the terms (tile.h)
            const f16x8 xh = __builtin_bit_cast(f16x8, vh), xl = __builtin_bit_cast(f16x8, vl);
#pragma unroll
            for (int cg = 0; cg < 4; ++cg) {
                const f16x8 wh = __builtin_bit_cast(f16x8, u[cg][kh][0]), wl = __builtin_bit_cast(f16x8, u[cg][kh][1]);
                const f32x4v c0 = kh == 0 ? f32x4v{0.f, 0.f, 0.f, 0.f} : acc[cg];
                acc[cg] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xl, c0, 0, 0, 0);
                acc[cg] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, xh, acc[cg], 0, 0, 0);
                acc[cg] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xh, acc[cg], 0, 0, 0);
            }
        }
        WPT(1);
        if (norm && pos < spb) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NPF) : "memory");   // this wave's statistics row has landed
        wp_barrier();                                          // every wave has finished the previous block (its reads of `ex`)
        WPT(2);

        // ---- exchange: M[pos][tile tn][couts cg * 16 + kq * 4 .. + 3]
        {
            const int tl = opaque_tid();
            float* e = ex + (pos * 16 + (tl & 15)) * ES + ((tl >> 4) & 3) * 4;
#pragma unroll
            for (int cg = 0; cg < 4; ++cg) *reinterpret_cast<f32x4v*>(e + cg * 16) = acc[cg];
        }
        WPT(3);
        // ---- finish, part 1: thread = (tile `pos`, output pixel (af, bf) of its 2x2 block, channel quad c4f).  The residual operands
        // are requested HERE, before the next tile's conversion: the accumulators are dead (16 registers free), and the conversion
        // and the barrier hide the round trip that used to sit in front of every block's stores
        const int tf = opaque_tid();
        const int c4f = tf & 15, abf = (tf >> 4) & 3, af = abf >> 1, bf = abf & 1;
        const int ftr = pos / WT, ftc = pos % WT;
        const int fbrow = 2 * ftr + af, fcol = 2 * ftc + bf; // output pixel inside the block
        // blocks tile the flattened (sample, row) space: block blk = global rows blk * R .. + R - 1
        const int grow = blk * R + fbrow;                      // n * H + row
        const bool pool = (p.flags & SBC_EPI_POOL) != 0;
        const bool live = grow < p.B * H && (!pool || abf == 0);
        // element offset of this thread's output: the pixel itself, or (pooled) the tile's pixel of the half-resolution tensor
        const unsigned o = pool ? (unsigned)(((blk * TR + ftr) * WT + ftc) * COUT + c4f * 4)
                                : (unsigned)blk * BLK_ELEMS + (unsigned)((fbrow * W + fcol) * COUT + c4f * 4);
        float4 rr = make_float4(0.f, 0.f, 0.f, 0.f), r2 = rr;
        if (live && p.res1) rr = ld_stream(p.res1 + o);
        if (live && p.res1 && p.res2) r2 = ld_stream(p.res2 + o);
        if (k + 1 < n_my) commit(block_of(k + 1), (k + 1) & 1);
        WPT(4);
        wp_barrier();
        WPT(5);

        // ---- finish, part 2
        const float sa = af ? -1.f : 1.f, sb = bf ? -1.f : 1.f;
        const float* exr = ex + ((af * 4 + bf) * 16 + pos) * ES + c4f * 4;       // first of the nine planes this thread combines
        float4 y;
        {
            float4 t3[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const float4 m0 = *reinterpret_cast<const float4*>(exr + (i * 4 + 0) * 16 * ES);
                const float4 m1 = *reinterpret_cast<const float4*>(exr + (i * 4 + 1) * 16 * ES);
                const float4 m2 = *reinterpret_cast<const float4*>(exr + (i * 4 + 2) * 16 * ES);
                // A^T over nu: b = 0: (M0 + M1) + M2,  b = 1: (M1 - M2) - M3
                t3[i].x = fmaf(sb, m2.x, fmaf(sb, m1.x, m0.x)); t3[i].y = fmaf(sb, m2.y, fmaf(sb, m1.y, m0.y));
                t3[i].z = fmaf(sb, m2.z, fmaf(sb, m1.z, m0.z)); t3[i].w = fmaf(sb, m2.w, fmaf(sb, m1.w, m0.w));
            }
            // ... and over xi
            y.x = fmaf(sa, t3[2].x, fmaf(sa, t3[1].x, t3[0].x)); y.y = fmaf(sa, t3[2].y, fmaf(sa, t3[1].y, t3[0].y));
            y.z = fmaf(sa, t3[2].z, fmaf(sa, t3[1].z, t3[0].z)); y.w = fmaf(sa, t3[2].w, fmaf(sa, t3[1].w, t3[0].w));
        }
        // descale (an exact power of two) in the same rounding as the bias add
        {
            const float4 bias4 = *reinterpret_cast<const float4*>(bl + c4f * 4);
            y.x = fmaf(y.x, descale, bias4.x); y.y = fmaf(y.y, descale, bias4.y);
            y.z = fmaf(y.z, descale, bias4.z); y.w = fmaf(y.w, descale, bias4.w);
        }
        if (pool) {
            // ((((0 + a) + b) + c) + d) / 4 with a=[0::2,0::2] b=[1::2,0::2] c=[0::2,1::2] d=[1::2,1::2]: the four pixels of the tile
            // sit in lanes 16 apart (abf = lane bits 4..5): lane abf = 0 collects them
            float4 y10, y01, y11;
            y10.x = __shfl_xor(y.x, 32); y10.y = __shfl_xor(y.y, 32); y10.z = __shfl_xor(y.z, 32); y10.w = __shfl_xor(y.w, 32);
            y01.x = __shfl_xor(y.x, 16); y01.y = __shfl_xor(y.y, 16); y01.z = __shfl_xor(y.z, 16); y01.w = __shfl_xor(y.w, 16);
            y11.x = __shfl_xor(y.x, 48); y11.y = __shfl_xor(y.y, 48); y11.z = __shfl_xor(y.z, 48); y11.w = __shfl_xor(y.w, 48);
            if (live) {
                float4 v;
                v.x = (((y.x + y10.x) + y01.x) + y11.x) * 0.25f; v.y = (((y.y + y10.y) + y01.y) + y11.y) * 0.25f;
                v.z = (((y.z + y10.z) + y01.z) + y11.z) * 0.25f; v.w = (((y.w + y10.w) + y01.w) + y11.w) * 0.25f;
                if (p.res1) { v.x = rr.x + v.x; v.y = rr.y + v.y; v.z = rr.z + v.z; v.w = rr.w + v.w; }
                st_stream(p.out + o, v);
            }
            continue;
        }
        if (!live) continue;
        if (p.res1) {
            if (p.flags & SBC_EPI_RES1_ELU) rr = elu4_acc(rr);
            if (p.res2) { rr.x = r2.x + rr.x; rr.y = r2.y + rr.y; rr.z = r2.z + rr.z; rr.w = r2.w + rr.w; }
            y.x += rr.x; y.y += rr.y; y.z += rr.z; y.w += rr.w;
        }
        if (p.flags & SBC_EPI_UP) {
            // F.interpolate(bilinear, align_corners=True) of `up` added on top (MSFBlock, layers.py:182-183)
            const int n = grow >> hsh, row = grow & (H - 1);
            const float shh = H > 1 ? (float)(p.up_h - 1) / (float)(H - 1) : 0.f;
            const float sww = W > 1 ? (float)(p.up_w - 1) / (float)(W - 1) : 0.f;
            const float* uu = p.up + (size_t)n * p.up_h * p.up_w * COUT + c4f * 4;
            const float fh = shh * (float)row, fw = sww * (float)fcol;
            const int h0 = min((int)fh, p.up_h - 1), w0 = min((int)fw, p.up_w - 1);
            const int h1 = min(h0 + 1, p.up_h - 1), w1 = min(w0 + 1, p.up_w - 1);
            const float lh1 = fh - (float)h0, lw1 = fw - (float)w0;
            const float lh0 = 1.f - lh1, lw0 = 1.f - lw1;
            const float4 v00 = *reinterpret_cast<const float4*>(uu + (size_t)(h0 * p.up_w + w0) * COUT);
            const float4 v01 = *reinterpret_cast<const float4*>(uu + (size_t)(h0 * p.up_w + w1) * COUT);
            const float4 v10 = *reinterpret_cast<const float4*>(uu + (size_t)(h1 * p.up_w + w0) * COUT);
            const float4 v11 = *reinterpret_cast<const float4*>(uu + (size_t)(h1 * p.up_w + w1) * COUT);
            y.x += lh0 * (lw0 * v00.x + lw1 * v01.x) + lh1 * (lw0 * v10.x + lw1 * v11.x);
            y.y += lh0 * (lw0 * v00.y + lw1 * v01.y) + lh1 * (lw0 * v10.y + lw1 * v11.y);
            y.z += lh0 * (lw0 * v00.z + lw1 * v01.z) + lh1 * (lw0 * v10.z + lw1 * v11.z);
            y.w += lh0 * (lw0 * v00.w + lw1 * v01.w) + lh1 * (lw0 * v10.w + lw1 * v11.w);
        }
        st_stream(p.out + o, y);
    }
#ifdef SBC_WP_TIMING
    WPT(6);
    if (tid == 0 && p.up && !(p.flags & SBC_EPI_UP)) {
        unsigned long long* dd = reinterpret_cast<unsigned long long*>(const_cast<float*>(p.up)) + (size_t)blockIdx.x * 10;
        for (int q = 0; q < 8; ++q) dd[q] = pt[q];
        dd[8] = __builtin_readcyclecounter() - pt_start;
        dd[9] = n_my;
    }
#endif
    f16x2_range_report(amax, act_scale, p.range_flag, p.calib);
}

// ------------------------------------------------------------------------------------------------ dispatch
template <int W, bool PARTIAL>
static int launch_wp(const ConvParams& p, hipStream_t stream, bool dry) {
    constexpr int WT = W / 2, TR = 16 / WT, R = 2 * TR;
    constexpr int ROWS = PARTIAL ? R + 2 : R;
    constexpr int RS = W * WP_S + wp_rowpad<W>(), DSZ = ROWS * RS + WP_S;
    WpExtra x{};
    x.spb = PARTIAL ? 1 : R / p.H;
    x.bps_sh = PARTIAL ? log2_exact(p.H / R) : 0;
    const long units = PARTIAL ? (long)p.B * (p.H / R) : ((long)p.B + x.spb - 1) / x.spb;
    x.n_blocks = (int)units;
    const size_t lds = ((size_t)2 * DSZ + 16 * 16 * WP_ES + (size_t)2 * x.spb * 3 * 64 + 64) * sizeof(float);
    if (lds > 160 * 1024) return 1;
    auto kern = conv_wp_kernel<W, PARTIAL>;
    { const int rc = ensure_dyn_lds(reinterpret_cast<const void*>(kern), lds); if (rc) return rc; }
    if (dry) return SBC_OK;
    int dev = 0, cus = 256;
    SBC_CHECK_HIP(hipGetDevice(&dev));
    SBC_CHECK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    x.blocks_per_xcd = (x.n_blocks + 7) / 8;
    x.wgs_per_xcd = max(1, min(cus / 8, x.blocks_per_xcd));
    hipLaunchKernelGGL(kern, dim3(8 * x.wgs_per_xcd), dim3(1024), lds, stream, p, x);
    SBC_CHECK_HIP(hipGetLastError());
    return SBC_OK;
}

// SBC_OK after launching, 1 when the layer is not this kernel's (the caller then uses conv_wx3), < 0 on errors
int launch_conv_wp(const ConvParams& p, int cin, int cout, hipStream_t stream, bool dry) {
    static const bool off = getenv("SBC_NO_WP") != nullptr;                        // A/B aid
    if (off || cin != 64 || cout != 64 || p.dil != 1 || !(p.flags & SBC_CONV_F16X2)) return 1;
    if (p.flags & (SBC_EPI_MOMENTS_OUT | SBC_EPI_ELUGRAD | SBC_PRO_NORM_SELF)) return 1;
    if (p.hsh < 1 || p.wsh < 1) return 1;                                          // power-of-two images with even sides
    const int H = p.H;
    auto go = [&](auto wc) {
        constexpr int W = decltype(wc)::value;
        constexpr int R = 2 * (16 / (W / 2));
        if (H > R) return launch_wp<W, true>(p, stream, dry);
        if (R / H > 8) return 1;
        return launch_wp<W, false>(p, stream, dry);
    };
    switch (p.W) {
        case 2: return go(std::integral_constant<int, 2>{});
        case 4: return go(std::integral_constant<int, 4>{});
        case 8: return go(std::integral_constant<int, 8>{});
        default: return 1;
    }
}

}  // namespace sbc

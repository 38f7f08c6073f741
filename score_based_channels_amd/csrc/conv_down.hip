// The tail of a DOWNSAMPLING ResidualBlock in one launch (SBC_OP_CONV_DOWN; ncsnv2/models/layers.py:443-456 with
// ConvMeanPool :309-313, the undilated 'down' blocks res2.0 / res3.0 of ncsnv2.py:222-234):
//       out = meanpool2(conv3x3(ELU(norm2(a))) + b2)  +  meanpool2(conv1x1(x) + bs)
// a = conv1's output, x = the block's input, both [B][H][W][CIN]; out [B][H/2][W/2][COUT].
//
// A 2 x 2 mean pool behind a convolution is a STRIDE-2 convolution with the summed filter: pooling the 3 x 3 convolution gives a
// 4 x 4 stride-2 convolution with W4[p][q] = 1/4 sum_{a,b in {0,1}} W3[p - a][q - b], pooling the 1 x 1 shortcut a 2 x 2 stride-2
// convolution with Ws / 4 on every tap (both formed in float64 on the host: `#pool4` / `#pool2` weight forms).  The unfused plan
// ran the 3 x 3 convolution at FULL resolution as Winograd F(2x2,3x3) -- 16 products per pooled pixel too, but behind 35 vector
// instructions per matrix instruction (conv_wx3<32,64>: 254 us) -- then pooled, and ran the shortcut as a second launch that reads
// the block input and writes a pooled tensor the first launch reads back (conv_x3<32,64,1x1>: 99 us): 780 MB moved.  Here: 16 + 4
// taps of a direct implicit GEMM on v_mfma_f32_16x16x32_f16 (three fp16 products per tap, conv_mode f16x2), both inputs read
// once, the output written once: 557 MB at 64 x 16.
//
// One 4-wave workgroup per tile of R pooled rows of one sample (two workgroups per CU: one converts while the other multiplies);
// wave = 16-output-channel block, all units of the tile.  Operand planes [term][8-channel group][column parity][row][RP slots][8
// halves]: de-interleaving the columns by parity makes a stride-2 tap a unit-stride read -- tap column q of pooled pixel xo is
// slot xo + (q >> 1) of parity (q + 1) & 1 (odd columns are stored one slot to the right, so column -1 is slot 0) -- and the row
// pitch RP puts the 16 pooled pixels of a unit on 16 different 16-byte slots (mod 16): conflict-free ds_read_b128 (conv_dp.hip has
// the rule).  The shortcut's input goes through the SAME planes after the main K loop; its loads are in flight during that loop.
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include "conv_common.h"

namespace sbc {

typedef float f32x4v __attribute__((ext_vector_type(4)));

struct DownParams {
    const float* __restrict__ a;          // conv1's output [B][H][W][CIN]
    const float* __restrict__ x;          // the block's input [B][H][W][CIN]
    float* __restrict__ out;              // [B][H/2][W/2][COUT]
    const float* __restrict__ stats;      // normalize2: (mu, scale, shift) [B][3][CIN]
    const uint4* __restrict__ w4;         // sbc_pack_conv_weight_f16x2(ksize 4) of the pooled 3x3 filter
    const uint4* __restrict__ w2;         // ... (ksize 2) of the pooled 1x1 shortcut filter
    const float* __restrict__ bias;       // conv2's bias [COUT]
    const float* __restrict__ bias2;      // the shortcut's bias [COUT]
    unsigned* __restrict__ range_flag;
    float* __restrict__ calib;            // sbc_f16x2_calibrate: two amax slots (the activated a, x), else NULL
    int B, H;
};

template <int B_, int E_, class F>
__device__ __forceinline__ void down_static_for(F&& f) {
    if constexpr (B_ < E_) {
        f(std::integral_constant<int, B_>{});
        down_static_for<B_ + 1, E_>(f);
    }
}

// W = 16: tiles of 8 pooled rows (18 input rows), units of two pooled rows;  W = 8: tiles of 8 pooled rows, units of four.
template <int CIN, int COUT, int W>
__global__ __launch_bounds__(256, 2) void conv_down_kernel(DownParams p) {
    constexpr int WO = W / 2, R = 8, NR = 2 * R + 2;
    constexpr int RP = W == 16 ? 12 : 6;                        // slots per plane row (>= WO + 1; see the header for the choice)
    constexpr int CG = CIN / 8, KH = CIN / 32, C4 = CIN / 4, NCB = COUT / 16;
    constexpr int NU = R * WO / 16;                             // units per tile: 4 (W = 16) or 2 (W = 8)
    constexpr int RPU = 16 / WO;                                // pooled rows per unit
    constexpr int PS = (2 * NR * RP * 16 + 255) / 256 * 256;    // bytes of one (term, channel group) plane
    constexpr int TERM = CG * PS;
    static_assert(NCB == 4, "four waves, one 16-output-channel block each");
    static_assert((W == 16 && CIN == 32) || (W == 8 && CIN == 64), "res2.0 (32 -> 64 at 16-pixel rows) and res3.0 (64 -> 64 at 8)");
    extern __shared__ __attribute__((aligned(256))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int cb = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kq = lane >> 4, n = lane & 15;
    const int H = p.H, HO = H / 2;
    const int tiles = HO / R;
    const int s = blockIdx.x / tiles, y0 = (blockIdx.x - s * tiles) * R;       // sample, first pooled row of the tile
    const int r_in0 = 2 * y0 - 1;                                              // input row of plane row 0

    // ---- filters stream from L2 through a register ring: K step = (tap, 32-channel slice)
    const int wl_base = (((kq >> 1) * (COUT / 32) + (cb >> 1)) * 2) * 64 + (16 * (cb & 1) + n) + 32 * (kq & 1);
    constexpr int WD = 4;
    uint4 wr[WD][2];
    auto ldw = [&](const uint4* __restrict__ w, int tap, int kh, int slot) {    // compile-time constants at every call
        const int idx = wl_base + ((tap * (CIN / 16) + 2 * kh) * (COUT / 32) * 2) * 64;
        wr[slot][0] = w[idx];
        wr[slot][1] = w[idx + 64];
    };
    const float4 tr4 = f16x2_trailer(reinterpret_cast<const float4*>(p.w4), 16 * (CIN / 16) * (COUT / 32) * 2);
    const float4 tr2 = f16x2_trailer(reinterpret_cast<const float4*>(p.w2), 4 * (CIN / 16) * (COUT / 32) * 2);
    const float scale_m = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, tr4.x)));
    const float scale_s = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, tr2.x)));
    const float descale_m = tr4.y, descale_s = tr2.y;
    const bool elu_acc = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, tr4.w)) != 0;
    unsigned rbits = 0;

    // ---- staging: thread t converts 16-byte chunks q = k * 256 + t of the tile's rows (row-major NHWC: channel quad c4 = t % C4
    // for every chunk of a thread); plane row of input row r: r - r_in0; column c: parity c & 1, slot (c >> 1) + (c & 1)
    const int c4 = tid % C4;
    constexpr int CPR = W * C4;                                // chunks per input row
    auto plane_dst = [&](int prow, int col) {
        return smem + (c4 >> 1) * PS + (((col & 1) * NR + prow) * RP + (col >> 1) + (col & 1)) * 16 + (c4 & 1) * 8;
    };
    // zero the planes once: padding slots, and rows outside the image, are never written (a tile's rows are either all inside the
    // image for the whole launch or outside it: a workgroup has one tile)
    for (int i = tid; i < 2 * TERM / 16; i += 256) *reinterpret_cast<uint4*>(smem + i * 16) = make_uint4(0, 0, 0, 0);

    // the activated a: NR rows (two of them halo); requests first, then the norm's (mu, scale, shift) of this thread's channels
    constexpr int NQA = NR * CPR / 256, NQX = 2 * R * CPR / 256;
    static_assert(NR * CPR % 256 == 0 && 2 * R * CPR % 256 == 0, "whole rounds of chunks");
    float4 va[NQA];
    const float* a_base = p.a + (size_t)s * H * W * CIN;
#pragma unroll
    for (int k = 0; k < NQA; ++k) {
        const int q = k * 256 + tid, prow = q / CPR;
        const int r = r_in0 + prow;
        va[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r >= 0 && r < H) va[k] = *reinterpret_cast<const float4*>(a_base + (size_t)r * W * CIN + (q - prow * CPR) * 4);
    }
    const float* st = p.stats + (size_t)s * 3 * CIN + c4 * 4;
    const float4 mu = *reinterpret_cast<const float4*>(st), sc = *reinterpret_cast<const float4*>(st + CIN),
                 sh = *reinterpret_cast<const float4*>(st + 2 * CIN);
    ldw(p.w4, 0, 0, 0);
    ldw(p.w4, 1 / KH, 1 % KH, 1);
    ldw(p.w4, 2 / KH, 2 % KH, 2);
    __syncthreads();                                            // planes zeroed
    float ta = 0.f;
#pragma unroll
    for (int k = 0; k < NQA; ++k) {
        const int q = k * 256 + tid, prow = q / CPR, col = (q - prow * CPR) / C4;
        const int r = r_in0 + prow;
        if (r < 0 || r >= H) continue;                          // zero padding of the convolution (applies to the ACTIVATED tensor)
        float4 v = va[k];
        v.x = fmaf(v.x - mu.x, sc.x, sh.x); v.y = fmaf(v.y - mu.y, sc.y, sh.y);
        v.z = fmaf(v.z - mu.z, sc.z, sh.z); v.w = fmaf(v.w - mu.w, sc.w, sh.w);
        v = elu4(v, elu_acc);
        StageScale ss{scale_m, ta};
        scale_track(v, &ss);
        ta = ss.amax;
        uint2 h, l;
        split_f16x2(v, scale_m, h, l);
        unsigned char* dst = plane_dst(prow, col);
        *reinterpret_cast<uint2*>(dst) = h;
        *reinterpret_cast<uint2*>(dst + TERM) = l;
    }
    pair_range_tile(ta, scale_m, rbits, p.calib);
    // the shortcut's input: requested now, converted behind the main K loop
    float4 vx[NQX];
    const float* x_base = p.x + ((size_t)s * H + 2 * y0) * W * CIN;
#pragma unroll
    for (int k = 0; k < NQX; ++k) vx[k] = *reinterpret_cast<const float4*>(x_base + (size_t)(k * 256 + tid) * 4);
    lds_barrier();

    // ---- K loops.  Lane n = pooled pixel (yl, xo) of unit u: yl = u * RPU + n / WO, xo = n % WO; tap (p, q): plane row 2 yl + p,
    // parity (q + 1) & 1, slot xo + (q >> 1)
    const int yl0 = n / WO, xo = n % WO;
    const int rd_base = kq * PS + ((2 * yl0) * RP + xo) * 16;
    f32x4v accm[NU], accs[NU];
    auto kloop = [&](auto mainc, const uint4* __restrict__ w, f32x4v* acc) {
        constexpr bool MAIN = decltype(mainc)::value;
        constexpr int KT = MAIN ? 4 : 2, NS = KT * KT * KH;     // taps per side; K steps
        down_static_for<0, NS>([&](auto sc_) {
            constexpr int s_ = decltype(sc_)::value;
            constexpr int tap = s_ / KH, kh = s_ % KH, tp = tap / KT, tq = tap % KT;
            if constexpr (s_ + WD - 1 < NS) ldw(w, (s_ + WD - 1) / KH, (s_ + WD - 1) % KH, (s_ + WD - 1) % WD);
            const f16x8 wh = __builtin_bit_cast(f16x8, wr[s_ % WD][0]);
            const f16x8 wl = __builtin_bit_cast(f16x8, wr[s_ % WD][1]);
            // the shortcut's 2 x 2 window starts one row and one column inside the 4 x 4 window of the main convolution
            constexpr int prow = MAIN ? tp : tp + 1, q = MAIN ? tq : tq + 1;
            constexpr int off = (4 * kh) * PS + ((((q + 1) & 1) * NR + prow) * RP + (q >> 1)) * 16;
            f16x8 xh[NU], xl[NU];
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                xh[u] = *reinterpret_cast<const f16x8*>(smem + rd_base + off + (2 * u * RPU * RP) * 16);
                xl[u] = *reinterpret_cast<const f16x8*>(smem + rd_base + off + (2 * u * RPU * RP) * 16 + TERM);
            }
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const f32x4v z = s_ == 0 ? f32x4v{0.f, 0.f, 0.f, 0.f} : acc[u];
                acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xl[u], z, 0, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < NU; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, xh[u], acc[u], 0, 0, 0);
#pragma unroll
            for (int u = 0; u < NU; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xh[u], acc[u], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    kloop(std::true_type{}, p.w4, accm);

    // ---- the shortcut through the same planes: rows 2 y0 .. 2 y0 + 2 R - 1 = plane rows 1 .. 2 R (always inside the image)
    ldw(p.w2, 0, 0, 0);
    ldw(p.w2, 1 / KH, 1 % KH, 1);
    ldw(p.w2, 2 / KH, 2 % KH, 2);
    float tb = 0.f;
    uint2 xh_[NQX], xl_[NQX];
#pragma unroll
    for (int k = 0; k < NQX; ++k) {
        StageScale ss{scale_s, tb};
        scale_track(vx[k], &ss);
        tb = ss.amax;
        split_f16x2(vx[k], scale_s, xh_[k], xl_[k]);
    }
    pair_range_tile(tb, scale_s, rbits, p.calib ? p.calib + 1 : nullptr);
    lds_barrier();                                              // every wave has left the main K loop
#pragma unroll
    for (int k = 0; k < NQX; ++k) {
        const int q = k * 256 + tid, prow = q / CPR, col = (q - prow * CPR) / C4;
        unsigned char* dst = plane_dst(prow + 1, col);
        *reinterpret_cast<uint2*>(dst) = xh_[k];
        *reinterpret_cast<uint2*>(dst + TERM) = xl_[k];
    }
    lds_barrier();
    kloop(std::false_type{}, p.w2, accs);

    // ---- out = (main * descale + b2) + (shortcut * descale + bs)
    const int ch0 = 16 * cb + 4 * kq;
    const f32x4v b2 = *reinterpret_cast<const f32x4v*>(p.bias + ch0), bs = *reinterpret_cast<const f32x4v*>(p.bias2 + ch0);
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        float4 o;
        o.x = fmaf(accs[u][0], descale_s, bs[0]) + fmaf(accm[u][0], descale_m, b2[0]);
        o.y = fmaf(accs[u][1], descale_s, bs[1]) + fmaf(accm[u][1], descale_m, b2[1]);
        o.z = fmaf(accs[u][2], descale_s, bs[2]) + fmaf(accm[u][2], descale_m, b2[2]);
        o.w = fmaf(accs[u][3], descale_s, bs[3]) + fmaf(accm[u][3], descale_m, b2[3]);
        const int y = y0 + u * RPU + yl0;
        st_out(p.out + (((size_t)s * HO + y) * WO + xo) * COUT + ch0, o);
    }
    if (rbits && lane == 0) atomicOr(p.range_flag, rbits);
}

template <int CIN, int COUT, int W>
static int launch_down_t(const DownParams& p, hipStream_t stream, bool dry) {
    constexpr int NR = 18, RP = W == 16 ? 12 : 6;
    constexpr int PS = (2 * NR * RP * 16 + 255) / 256 * 256;
    constexpr int LDS = 2 * (CIN / 8) * PS;
    auto kern = conv_down_kernel<CIN, COUT, W>;
    { const int rc = ensure_dyn_lds(reinterpret_cast<const void*>(kern), LDS); if (rc) return rc; }
    if (dry) return SBC_OK;
    hipLaunchKernelGGL(kern, dim3(p.B * (p.H / 2 / 8)), dim3(256), LDS, stream, p);
    SBC_CHECK_HIP(hipGetLastError());
    return SBC_OK;
}

int launch_conv_down(const sbc_op& op, hipStream_t stream, bool dry) {
    SBC_REQUIRE(op.in && op.out && op.res1 && op.stats && op.weight_split && op.weight2_split && op.bias && op.bias2,
                "conv_down: in / res1 / out / stats / weight_split / weight2_split / bias / bias2 must be set");
    SBC_REQUIRE((op.W == 16 && op.cin == 32 && op.cout == 64) || (op.W == 8 && op.cin == 64 && op.cout == 64),
                "conv_down: 32 -> 64 channels at 16-pixel rows or 64 -> 64 at 8 (got %d -> %d, W = %d)", op.cin, op.cout, op.W);
    SBC_REQUIRE(op.H % 16 == 0 && op.H >= 16, "conv_down: H = %d must be a multiple of 16 (tiles of 8 pooled rows)", op.H);
    SBC_REQUIRE((op.flags & SBC_CONV_F16X2) && !(op.flags & SBC_CONV_F16W), "conv_down: SBC_CONV_F16X2 only (the weight forms it reads)");
    SBC_REQUIRE(op.out != op.in && op.out != op.res1, "conv_down: out must not alias its inputs");
    SBC_REQUIRE(op.B > 0 && (long)op.B * op.H * op.W * op.cin <= 0x7fffffffL, "conv_down: bad batch %d", op.B);
    DownParams p;
    memset(&p, 0, sizeof(p));
    p.a = (const float*)op.in; p.x = (const float*)op.res1; p.out = (float*)op.out; p.stats = (const float*)op.stats;
    p.w4 = (const uint4*)op.weight_split; p.w2 = (const uint4*)op.weight2_split;
    p.bias = (const float*)op.bias; p.bias2 = (const float*)op.bias2;
    p.calib = (float*)op.calib;
    p.B = op.B; p.H = op.H;
    unsigned* flag = nullptr;
    { const int rc = range_flag_ptr(&flag); if (rc) return rc; }
    p.range_flag = flag;
    return op.W == 16 ? launch_down_t<32, 64, 16>(p, stream, dry) : launch_down_t<64, 64, 8>(p, stream, dry);
}

}  // namespace sbc

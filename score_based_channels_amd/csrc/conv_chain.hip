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
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include "conv_common.h"

namespace sbc {

typedef float f32x4v __attribute__((ext_vector_type(4)));

struct ChainParams {
    const float* __restrict__ in;
    float* __restrict__ out;
    const uint4* w[SBC_CHAIN_MAX_BLOCKS][2];
    int type[SBC_CHAIN_MAX_BLOCKS];
    int n_blocks;
    unsigned* __restrict__ range_flag;
    float* __restrict__ calib;          // sbc_f16x2_calibrate: 2 amax slots per block (conv1's input, conv2's input), else NULL
    int B;
};

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

// The K loop's micro-steps.  K step s = (tap, 32-channel slice kh), tap-major; within a K step the units are taken in pairs:
//   dx =  0: NU / 2 micro-steps, (column 0, column 1) of sample pair q;
//   dx = -1: NU / 4 micro-steps, column 1 of sample pairs 2q, 2q + 1 (column 0 would read padding: skipped);  dx = +1: column 0.
struct MicroStep { int s, ua, ub; bool first_of_step; };
template <int NU, int KH>
__host__ __device__ constexpr int micro_steps() { return 3 * (NU / 2 + 2 * (NU / 4)) * KH; }
template <int NU, int KH>
__host__ __device__ constexpr MicroStep micro_step(int m) {
    constexpr int M0 = NU / 2, M1 = NU / 4, ROW = (M0 + 2 * M1) * KH;
    const int row = m / ROW;
    int r = m % ROW, tapc = 0, kh = 0, q = 0;
    if (r < M1 * KH) { tapc = 0; kh = r / M1; q = r % M1; }
    else if (r < (M1 + M0) * KH) { r -= M1 * KH; tapc = 1; kh = r / M0; q = r % M0; }
    else { r -= (M1 + M0) * KH; tapc = 2; kh = r / M1; q = r % M1; }
    MicroStep d{(3 * row + tapc) * KH + kh, 0, 0, q == 0};
    if (tapc == 1) { d.ua = 2 * q; d.ub = 2 * q + 1; }
    else if (tapc == 0) { d.ua = 4 * q + 1; d.ub = 4 * q + 3; }
    else { d.ua = 4 * q; d.ub = 4 * q + 2; }
    return d;
}

// C channels in = out; G = 8 samples of 8 x 2 pixels per workgroup; 8 waves:
//   C = 128: wave = 16-output-channel block cb, all 8 units (4 sample pairs x 2 column parities);
//   C =  64: wave = (cb, half): 4 output-channel blocks x 2 halves of the sample pairs, 4 units each.
template <int C>
__global__ __launch_bounds__(512, 2) void conv_chain_w2_kernel(ChainParams p) {
    constexpr int G = 8, CG = C / 8, KH = C / 32, NCB = C / 16;
    constexpr int NP = C == 128 ? 4 : 2;                  // sample pairs per wave
    constexpr int NU = 2 * NP;                            // units per wave: i = 2 * pair + column
    constexpr int PS = G * 24 * 16;                       // bytes of one (term, channel group) plane: 3072 = 12 bank rows
    constexpr int TERM = CG * PS;                         // low-term planes behind the high-term planes
    static_assert(C == 64 || C == 128, "64 or 128 channels");
    extern __shared__ __attribute__((aligned(256))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cb = wave % NCB, uh = wave / NCB;
    const int j0 = uh * NP;                               // first sample pair of this wave
    const int kq = lane >> 4, n = lane & 15;
    const int sp = n >> 3, y = n & 7;                     // sample of the pair, image row of this lane's pixel
    const int s0 = blockIdx.x * G;

    // ---- zero the planes once: the slots above and below every column are never written
    for (int i = tid; i < 2 * TERM / 16; i += 512) *reinterpret_cast<uint4*>(smem + i * 16) = make_uint4(0, 0, 0, 0);

    // ---- the running tensor, accumulator layout: lane (kq, n) holds channels 16 cb + 4 kq .. + 3 of pixel n of each unit
    f32x4v xs[NU];
    const int ch0 = 16 * cb + 4 * kq;
#pragma unroll
    for (int i = 0; i < NU; ++i) {
        const int s = s0 + 2 * (j0 + (i >> 1)) + sp;
        xs[i] = f32x4v{0.f, 0.f, 0.f, 0.f};
        if (s < p.B) xs[i] = *reinterpret_cast<const f32x4v*>(p.in + ((size_t)(s * 8 + y) * 2 + (i & 1)) * C + ch0);
    }
    // lane parts of the LDS addresses (bytes); the rest are compile-time constants
    const int rd_base = kq * PS + ((2 * j0 + sp) * 24 + y) * 16;
    const int wr_base = (2 * cb + (kq >> 1)) * PS + ((2 * j0 + sp) * 24 + 1 + y) * 16 + (kq & 1) * 8;
    // lane part of the filter-fragment index (packed layout [tap][C/16 input groups][C/32 output blocks][2 terms][64 lanes], conv_pair.hip)
    const int wl_base = (((kq >> 1) * (C / 32) + (cb >> 1)) * 2) * 64 + (16 * (cb & 1) + n) + 32 * (kq & 1);

    f32x4v acc[NU];
#pragma unroll
    for (int i = 0; i < NU; ++i) acc[i] = f32x4v{0.f, 0.f, 0.f, 0.f};
    float prev_descale = 1.f;
    unsigned rbits = 0;
    __syncthreads();

#pragma unroll 1
    for (int blk = 0; blk < p.n_blocks; ++blk) {
        const int type = p.type[blk];
#pragma unroll 1
        for (int cv = 0; cv < 2; ++cv) {
            const uint4* __restrict__ w = p.w[blk][cv];
            const float4 tr = f16x2_trailer(reinterpret_cast<const float4*>(w), 9 * (C / 16) * (C / 32) * 2);
            const float scale = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, tr.x)));
            const float descale = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, tr.y)));
            const bool elu_acc = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, tr.w)) != 0;

            // ---- filter ring: K step s = (tap, 32-channel slice kh); the first three steps are requested before the operand work
            constexpr int NS = 9 * KH, WD = 4;
            uint4 wr[WD][2];
            auto ldw = [&](int s) {                                     // s is a compile-time constant at every call
                const int tap = s / KH, kh = s % KH;
                const int idx = wl_base + ((tap * (C / 16) + 2 * kh) * (C / 32) * 2) * 64;
                wr[s % WD][0] = w[idx];
                wr[s % WD][1] = w[idx + 64];
            };
            ldw(0);
            ldw(1);
            ldw(2);

            // ---- the operand of this convolution, from registers
            f32x4v v[NU];
            if (type == SBC_CHAIN_RCU) {
#pragma unroll
                for (int i = 0; i < NU; ++i) {
                    const f32x4v src = cv == 0 ? xs[i] : vscale4(acc[i], prev_descale);
                    v[i] = to_v4(elu4(to_f4(src), elu_acc));
                }
            } else {
                if (cv == 0) {
                    // x = act(x) (layers.py:77): the activated tensor is both the running sum and the first pooling input.  ELU in
                    // its accurate form: it sits outside a convolution prologue here (common.h)
#pragma unroll
                    for (int i = 0; i < NU; ++i) { xs[i] = to_v4(elu4_acc(to_f4(xs[i]))); v[i] = xs[i]; }
                } else {
                    // p = conv_w1(...), x = p + x (layers.py:82), and p is the second pooling input
#pragma unroll
                    for (int i = 0; i < NU; ++i) { acc[i] = vscale4(acc[i], prev_descale); xs[i] = vadd4(acc[i], xs[i]); v[i] = acc[i]; }
                }
                // nn.MaxPool2d(5, 1, 2) on an 8 x 2 image: both columns (same lane of the pair's two units) x rows y - 2 .. y + 2
                // (lanes n - 2 .. n + 2 of the same sample; -inf outside, as PyTorch pads)
#pragma unroll
                for (int jp = 0; jp < NP; ++jp) {
                    const f32x4v m = vmax4(v[2 * jp], v[2 * jp + 1]);
                    f32x4v r = m;
#pragma unroll
                    for (int d = -2; d <= 2; ++d) {
                        if (d == 0) continue;
                        const bool ok = (unsigned)(y + d) < 8u;
                        f32x4v o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = __shfl(m[e], lane + d);
                        if (ok) r = vmax4(r, o);
                    }
                    v[2 * jp] = r;
                    v[2 * jp + 1] = r;
                }
            }
            float ta = 0.f;
            uint2 vh[NU], vl[NU];
#pragma unroll
            for (int i = 0; i < NU; ++i) {
                StageScale ss{scale, ta};
                const float4 f = to_f4(v[i]);
                scale_track(f, &ss);
                ta = ss.amax;
                split_f16x2(f, scale, vh[i], vl[i]);
            }
            pair_range_tile(ta, scale, rbits, p.calib ? p.calib + 2 * blk + cv : nullptr);
            // every wave has left the previous K loop: the planes may be rewritten
            lds_barrier();
#pragma unroll
            for (int i = 0; i < NU; ++i) {
                unsigned char* dst = smem + wr_base + ((2 * (i >> 1)) * 24 + (i & 1) * 12) * 16;
                *reinterpret_cast<uint2*>(dst) = vh[i];
                *reinterpret_cast<uint2*>(dst + TERM) = vl[i];
            }
            lds_barrier();

            // ---- K loop: D[16 couts][16 pixels] += W[16 couts][32 cin] X[32 cin][16 pixels] per (tap, slice), three fp16 products each.
            // Flat walk over micro-steps m = (K step, pair of units): the four X fragments of a micro-step are requested XD - 1
            // micro-steps ahead of its six matrix instructions through a ring of statically indexed registers (left to itself the
            // scheduler reads each fragment right in front of its first use: one LDS round trip per unit), and the two units' matrix
            // instructions alternate, so no instruction waits for the accumulator of the one before.
            // A K step has NU / 2 micro-steps when dx = 0 (all units), half as many otherwise (the units of one column parity).
            constexpr int XD = 4;
            f16x8 xr[XD][4];
            // micro-step m -> (K step, first unit, second unit); everything folds at compile time (all callers pass constants)
            auto ldx = [&](int m) {
                const MicroStep d = micro_step<NU, KH>(m);
                const int tap = d.s / KH, kh = d.s % KH, dy = tap / 3 - 1, dx = tap % 3 - 1;
                const int offa = (4 * kh) * PS + ((2 * (d.ua >> 1)) * 24 + ((d.ua & 1) + dx) * 12 + 1 + dy) * 16;
                const int offb = (4 * kh) * PS + ((2 * (d.ub >> 1)) * 24 + ((d.ub & 1) + dx) * 12 + 1 + dy) * 16;
                xr[m % XD][0] = *reinterpret_cast<const f16x8*>(smem + rd_base + offa);
                xr[m % XD][1] = *reinterpret_cast<const f16x8*>(smem + rd_base + offa + TERM);
                xr[m % XD][2] = *reinterpret_cast<const f16x8*>(smem + rd_base + offb);
                xr[m % XD][3] = *reinterpret_cast<const f16x8*>(smem + rd_base + offb + TERM);
            };
            constexpr int NM = micro_steps<NU, KH>();
#pragma unroll
            for (int m = 0; m < XD - 1; ++m) ldx(m);
            static_for<0, NM>([&](auto mc) {
                constexpr int m = decltype(mc)::value;
                constexpr MicroStep d = micro_step<NU, KH>(m);
                constexpr int tap = d.s / KH, kh = d.s % KH;
                if constexpr (d.first_of_step && d.s + WD - 1 < NS) ldw(d.s + WD - 1);     // keep the filter ring full
                if constexpr (m + XD - 1 < NM) ldx(m + XD - 1);
                const f16x8 wh = __builtin_bit_cast(f16x8, wr[d.s % WD][0]);
                const f16x8 wl = __builtin_bit_cast(f16x8, wr[d.s % WD][1]);
                const f16x8 ah = xr[m % XD][0], al = xr[m % XD][1], bh = xr[m % XD][2], bl = xr[m % XD][3];
                // a unit's first matrix instruction takes a literal zero addend: tap 0 for column 1, tap 1 for column 0
                constexpr bool fa = kh == 0 && tap == ((d.ua & 1) ? 0 : 1), fb = kh == 0 && tap == ((d.ub & 1) ? 0 : 1);
                const f32x4v za = fa ? f32x4v{0.f, 0.f, 0.f, 0.f} : acc[d.ua], zb = fb ? f32x4v{0.f, 0.f, 0.f, 0.f} : acc[d.ub];
                acc[d.ua] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, al, za, 0, 0, 0);
                acc[d.ub] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, bl, zb, 0, 0, 0);
                acc[d.ua] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, ah, acc[d.ua], 0, 0, 0);
                acc[d.ub] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, bh, acc[d.ub], 0, 0, 0);
                acc[d.ua] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, ah, acc[d.ua], 0, 0, 0);
                acc[d.ub] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, bh, acc[d.ub], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            });
            // ---- second convolution of a block: the running sum takes it (descale is a power of two: one rounding, as an add)
            if (cv == 1) {
#pragma unroll
                for (int i = 0; i < NU; ++i) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) xs[i][e] = fmaf(acc[i][e], descale, xs[i][e]);
                }
            }
            prev_descale = descale;
        }
    }
#pragma unroll
    for (int i = 0; i < NU; ++i) {
        const int s = s0 + 2 * (j0 + (i >> 1)) + sp;
        if (s < p.B) st_out(p.out + ((size_t)(s * 8 + y) * 2 + (i & 1)) * C + ch0, to_f4(xs[i]));
    }
    if (rbits && lane == 0) atomicOr(p.range_flag, rbits);
}

template <int C>
static int launch_chain_w2(const ChainParams& p, hipStream_t stream, bool dry) {
    constexpr int LDS = 2 * (C / 8) * 8 * 24 * 16;
    auto kern = conv_chain_w2_kernel<C>;
    { const int rc = ensure_dyn_lds(reinterpret_cast<const void*>(kern), LDS); if (rc) return rc; }
    if (dry) return SBC_OK;
    hipLaunchKernelGGL(kern, dim3((p.B + 7) / 8), dim3(512), LDS, stream, p);
    SBC_CHECK_HIP(hipGetLastError());
    return SBC_OK;
}

int launch_chain(const sbc_op& op, const sbc_chain& c, hipStream_t stream, bool dry) {
    SBC_REQUIRE(op.in && op.out && op.in != op.out, "chain: in / out must be set and distinct");
    SBC_REQUIRE(op.H == 8 && op.W == 2 && op.cin == op.cout && (op.cin == 64 || op.cin == 128),
                "chain: 8 x 2 samples of 64 or 128 channels (got %d x %d, %d -> %d)", op.H, op.W, op.cin, op.cout);
    SBC_REQUIRE((op.flags & SBC_CONV_F16X2) && !(op.flags & SBC_CONV_F16W), "chain: SBC_CONV_F16X2 only (the weight forms it reads)");
    SBC_REQUIRE(c.n_blocks >= 1 && c.n_blocks <= SBC_CHAIN_MAX_BLOCKS, "chain: %d blocks (1 .. %d)", c.n_blocks, SBC_CHAIN_MAX_BLOCKS);
    SBC_REQUIRE(op.B > 0 && (long)op.B * 16 * op.cin <= 0x7fffffffL, "chain: bad batch %d", op.B);
    ChainParams p;
    memset(&p, 0, sizeof(p));
    p.in = (const float*)op.in;
    p.out = (float*)op.out;
    p.n_blocks = c.n_blocks;
    for (int b = 0; b < c.n_blocks; ++b) {
        SBC_REQUIRE(c.w1[b] && c.w2[b] && (c.type[b] == SBC_CHAIN_RCU || c.type[b] == SBC_CHAIN_CRP), "chain: block %d: weights / type", b);
        p.w[b][0] = (const uint4*)c.w1[b];
        p.w[b][1] = (const uint4*)c.w2[b];
        p.type[b] = c.type[b];
    }
    p.calib = (float*)op.calib;
    p.B = op.B;
    unsigned* flag = nullptr;
    { const int rc = range_flag_ptr(&flag); if (rc) return rc; }
    p.range_flag = flag;
    return op.cin == 128 ? launch_chain_w2<128>(p, stream, dry) : launch_chain_w2<64>(p, stream, dry);
}

}  // namespace sbc

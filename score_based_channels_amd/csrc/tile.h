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

// Streaming (non-temporal) 16-byte accesses for activations: each element is touched once per kernel, so it
// should not displace the L1/L2-resident weight fragments the MFMA loops of co-resident workgroups re-read.
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld_stream(const float* p) {
#ifdef SBC_NO_STREAM
    return *reinterpret_cast<const float4*>(p);
#else
    const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
#endif
}
__device__ __forceinline__ void st_stream(float* p, float4 v) {
#ifdef SBC_NO_STREAM
    *reinterpret_cast<float4*>(p) = v;
#else
    f32x4 t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
    __builtin_nontemporal_store(t, reinterpret_cast<f32x4*>(p));
#endif
}

struct TileGeom {
    int p0;        // first output pixel (flattened n*H*W + h*W + w)
    int rs0;       // first staged global row (n*H + h)
    int nps;       // staged pixels; LDS pixel index nps is the zero pixel
    int n_first;   // sample index of the first staged row
    int multi;     // tile spans whole samples (no halo)
};

__device__ __forceinline__ TileGeom tile_geom(int tile, int TM, int B, int H, int W, int halo_rows) {
    TileGeom g;
    const int HW = H * W;
    g.p0 = tile * TM;
    const int r0 = g.p0 / W;
    int r1 = r0 + TM / W;
    if (r1 > B * H) r1 = B * H;
    g.multi = TM >= HW;
    int rs1;
    if (g.multi) {
        g.rs0 = r0;
        rs1 = r1;
    } else {
        const int n = r0 / H;
        g.rs0 = max(r0 - halo_rows, n * H);
        rs1 = min(r1 + halo_rows, (n + 1) * H);
    }
    g.nps = (rs1 - g.rs0) * W;
    g.n_first = g.rs0 / H;
    return g;
}

// Staging is split in two so a persistent workgroup can request tile i+1 before it computes tile i:
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

template <int CIN>
__device__ __forceinline__ void stage_put(float* lds, float4 x, int idx, const float* __restrict__ stats, int flags,
                                          const TileGeom& g, int HW) {
    constexpr int S = CIN + 4;
    constexpr int C4 = CIN / 4;
    const int pix = idx / C4, c4 = idx % C4;
    if (flags & SBC_PRO_NORM) {
        const int n = g.n_first + (g.multi ? pix / HW : 0);
        const float* st = stats + (size_t)n * 3 * CIN + c4 * 4;
        const float4 mu = *reinterpret_cast<const float4*>(st);
        const float4 sc = *reinterpret_cast<const float4*>(st + CIN);
        const float4 sh = *reinterpret_cast<const float4*>(st + 2 * CIN);
        x.x = (x.x - mu.x) * sc.x + sh.x;
        x.y = (x.y - mu.y) * sc.y + sh.y;
        x.z = (x.z - mu.z) * sc.z + sh.z;
        x.w = (x.w - mu.w) * sc.w + sh.w;
    }
    if (flags & SBC_PRO_ELU) x = elu4(x);
    *reinterpret_cast<float4*>(lds + pix * S + c4 * 4) = x;
}

template <int CIN, int NTHREADS, int NPF>
__device__ __forceinline__ void stage_commit(float* lds, const float4 (&pf)[NPF], const float* __restrict__ in,
                                             const float* __restrict__ stats, int flags, const TileGeom& g, int H,
                                             int W, int tid) {
    constexpr int S = CIN + 4;
    const int HW = H * W;
    const int total = g.nps * (CIN / 4);
#pragma unroll
    for (int u = 0; u < NPF; ++u) {
        const int idx = u * NTHREADS + tid;
        if (idx < total) stage_put<CIN>(lds, pf[u], idx, stats, flags, g, HW);
    }
    const float* src = in + (size_t)g.rs0 * W * CIN;
    for (int idx = NPF * NTHREADS + tid; idx < total; idx += NTHREADS)
        stage_put<CIN>(lds, ld_stream(src + (size_t)idx * 4), idx, stats, flags, g, HW);
    for (int i = tid; i < S; i += NTHREADS) lds[g.nps * S + i] = 0.f;
}

template <int CIN, int NTHREADS, int NPF>
__device__ __forceinline__ void stage_tile(float* lds, const float* __restrict__ in, const float* __restrict__ stats,
                                           int flags, const TileGeom& g, int H, int W, int tid) {
    float4 pf[NPF];
    stage_issue<CIN, NTHREADS, NPF>(pf, in, g, W, tid);
    stage_commit<CIN, NTHREADS, NPF>(lds, pf, in, stats, flags, g, H, W, tid);
}

}  // namespace sbc

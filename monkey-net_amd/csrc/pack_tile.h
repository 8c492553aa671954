// Weight re-layout tile shared by the pack kernels (conv3x3.hip) and the optimiser kernel that emits the packed layouts
// right after updating the parameter (optim.hip).
#pragma once
#include "mnk_common.h"

namespace mnk {

// forward layout and the data-gradient layouts of both sources in one launch (what a training forward needs; the pack
// kernels of conv3x3.hip stay for single uses).  One block per (16 output channels, source, 16 input channels) tile of
// the parameter: the 16 x (16 * ntaps) floats are read as 16 contiguous runs, transposed through LDS and written as
// 16 + 16 contiguous 16 * ntaps-float groups -- forward wf[co][chunk][tap][ci] and, for the same tile,
// data-gradient wd[ci][chunk][ntaps-1-tap][co] -- so the parameter crosses HBM once and every access is coalesced.
// T: LDS [16 co][16 ci][ntaps <= 16] with padded strides (odd: no bank conflicts); cc = forward chunk (source 0
// chunks, then source 1 chunks), cot = co tile (16 rows).
struct PackTileGeom {
    int ntaps, ntp, cos, chunks0, chunks, dchunks, Cs, cstart, ci0, co0, Cin, run;
    bool second;
};

template <int NT>                                 // NT = ntaps when known at compile time (9: constant divisions), else 0
__device__ __forceinline__ PackTileGeom pack_tile_geom(int Cout, int C0, int C1, int C0p, int C1p, int ntaps_rt, int cc,
                                                       int cot) {
    PackTileGeom g;
    g.ntaps = NT ? NT : ntaps_rt;
    g.ntp = g.ntaps | 1;
    g.cos = 16 * g.ntp + 1;
    g.chunks0 = C0p / 16;
    g.chunks = (C0p + C1p) / 16;
    g.dchunks = (Cout + 15) / 16;
    g.second = cc >= g.chunks0;
    g.Cs = g.second ? C1 : C0;
    g.cstart = g.second ? C0 : 0;
    const int lc = g.second ? cc - g.chunks0 : cc;
    g.ci0 = lc * 16;
    g.co0 = cot * 16;
    g.Cin = C0 + C1;
    g.run = 16 * g.ntaps;                         // floats per row of the tile
    return g;
}

// T (LDS tile, already holding the parameter values of the tile; zeros outside the parameter) -> packed layouts
template <int NT>
__device__ __forceinline__ void pack_tile_emit(const float* T, const PackTileGeom& g, float* __restrict__ wf,
                                               float* __restrict__ wd0, float* __restrict__ wd1, int Cout, int cc, int cot) {
    const int ntaps = NT ? NT : g.ntaps;
    const int run = 16 * ntaps;
    const int t = threadIdx.x;
    for (int i = t; i < 16 * run; i += 256) {       // forward: 16 rows (co) of [tap][16 ci]
        const int r = i / run, o = i - r * run;
        const int tap = o >> 4, k16 = o & 15;
        const int co = g.co0 + r;
        if (co < Cout) wf[(((size_t)co * g.chunks + cc) * ntaps) * 16 + o] = T[r * g.cos + k16 * g.ntp + tap];
    }
    float* wd = g.second ? wd1 : wd0;
    if (wd)
        for (int i = t; i < 16 * run; i += 256) {   // data gradient: 16 rows (ci) of [flipped tap][16 co]
            const int r = i / run, o = i - r * run;
            const int tap = o >> 4, k16 = o & 15;
            const int ci = g.ci0 + r;
            if (ci < g.Cs) wd[(((size_t)ci * g.dchunks + cot) * ntaps) * 16 + o] = T[k16 * g.cos + r * g.ntp + (ntaps - 1 - tap)];
        }
}

// ---- the same tile as the packs of the sub-pixel forms of an up-sampled 3x3 convolution (conv3x3.hip: ConvArgs::phases) ----
// forward  wf[phase = 2a + b][co][chunk][tap4 = 2u + v][16 ci] = sum_{ky in S(a,u)} sum_{kx in S(b,v)} w[co][ci][ky][kx]
// dgrad    wd[ci][chunk over co][tap16 = 4 ty + tx][16 co]      = sum_{ky in D(ty)} sum_{kx in D(tx)} w[co][ci][ky][kx]
//   S(0,0) = {0}, S(0,1) = {1,2}, S(1,0) = {0,1}, S(1,1) = {2};  D(0) = {2}, D(1) = {1,2}, D(2) = {0,1}, D(3) = {0}
__device__ __forceinline__ void up_phase_set(int a, int u, int& lo, int& hi) {
    lo = a == 0 ? (u == 0 ? 0 : 1) : (u == 0 ? 0 : 2);
    hi = a == 0 ? (u == 0 ? 0 : 2) : (u == 0 ? 1 : 2);
}
__device__ __forceinline__ void up_dgrad_set(int t, int& lo, int& hi) {
    lo = t == 0 ? 2 : (t == 1 ? 1 : 0);
    hi = t == 0 ? 2 : (t == 1 ? 2 : (t == 2 ? 1 : 0));
}

// ---- the weight gradient of the sub-pixel form (16 pseudo taps) folded into the nine kernel taps ------------------------------
// the pseudo taps of the sub-pixel form that contribute to kernel row (column) k: (a, u) with k in S(a, u) -- two each
__device__ __forceinline__ void up_fold_pairs(int k, int& a0, int& u0, int& a1, int& u1) {
    // S(0,0) = {0}, S(0,1) = {1,2}, S(1,0) = {0,1}, S(1,1) = {2}
    a0 = 0, u0 = k == 0 ? 0 : 1;          // k = 0: (0,0); k = 1, 2: (0,1)
    a1 = 1, u1 = k == 2 ? 1 : 0;          // k = 0, 1: (1,0); k = 2: (1,1)
}
// dW[ky][kx] from the 16 pseudo-tap sums acc[4 * (2a + b) + 2u + v]
__device__ __forceinline__ float up_fold(const float* acc, int ky, int kx, int stride) {
    int ya[2], yu[2], xb[2], xv[2];
    up_fold_pairs(ky, ya[0], yu[0], ya[1], yu[1]);
    up_fold_pairs(kx, xb[0], xv[0], xb[1], xv[1]);
    float v = 0.f;
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int q = 0; q < 2; ++q) v += acc[(4 * (2 * ya[p] + xb[q]) + 2 * yu[p] + xv[q]) * stride];
    return v;
}

__device__ __forceinline__ void pack_tile_emit_up(const float* T, const PackTileGeom& g, float* __restrict__ wf,
                                                  float* __restrict__ wd0, float* __restrict__ wd1, int Cout, int cc, int cot) {
    const int t = threadIdx.x;
    for (int i = t; i < 4096; i += 256) {           // forward: 16 co x 4 phases x 4 taps x 16 ci
        const int k16 = i & 15, tap4 = (i >> 4) & 3, phase = (i >> 6) & 3, r = i >> 8;
        const int co = g.co0 + r;
        if (co >= Cout) continue;
        int y0, y1, x0, x1;
        up_phase_set(phase >> 1, tap4 >> 1, y0, y1);
        up_phase_set(phase & 1, tap4 & 1, x0, x1);
        const float* tp = T + r * g.cos + k16 * g.ntp;
        float v = 0.f;
        for (int ky = y0; ky <= y1; ++ky)
            for (int kx = x0; kx <= x1; ++kx) v += tp[ky * 3 + kx];
        wf[((((size_t)phase * Cout + co) * g.chunks + cc) * 4 + tap4) * 16 + k16] = v;
    }
    float* wd = g.second ? wd1 : wd0;
    if (wd)
        for (int i = t; i < 4096; i += 256) {       // data gradient: 16 ci x 16 taps x 16 co
            const int k16 = i & 15, tap16 = (i >> 4) & 15, r = i >> 8;
            const int ci = g.ci0 + r;
            if (ci >= g.Cs) continue;
            int y0, y1, x0, x1;
            up_dgrad_set(tap16 >> 2, y0, y1);
            up_dgrad_set(tap16 & 3, x0, x1);
            const float* tp = T + k16 * g.cos + r * g.ntp;
            float v = 0.f;
            for (int ky = y0; ky <= y1; ++ky)
                for (int kx = x0; kx <= x1; ++kx) v += tp[ky * 3 + kx];
            wd[(((size_t)ci * g.dchunks + cot) * 16 + tap16) * 16 + k16] = v;
        }
}

template <int NT>
__device__ __forceinline__ void pack_tile(float* T, const float* __restrict__ w, float* __restrict__ wf,
                                          float* __restrict__ wd0, float* __restrict__ wd1, int Cout, int C0, int C1,
                                          int C0p, int C1p, int ntaps_rt, int cc, int cot, int up = 0) {
    const PackTileGeom g = pack_tile_geom<NT>(Cout, C0, C1, C0p, C1p, ntaps_rt, cc, cot);
    const int ntaps = NT ? NT : g.ntaps;
    const int run = 16 * ntaps;
    const int t = threadIdx.x;
    for (int i = t; i < 16 * run; i += 256) {
        const int r = i / run, o = i - r * run;     // row (co), offset inside the row = ci * ntaps + tap
        const int ci = o / ntaps, tap = o - ci * ntaps;
        const int co = g.co0 + r;
        float v = 0.f;
        if (co < Cout && g.ci0 + ci < g.Cs) v = w[((size_t)co * g.Cin + g.cstart + g.ci0) * ntaps + o];
        T[r * g.cos + ci * g.ntp + tap] = v;
    }
    __syncthreads();
    if (up)
        pack_tile_emit_up(T, g, wf, wd0, wd1, Cout, cc, cot);
    else
        pack_tile_emit<NT>(T, g, wf, wd0, wd1, Cout, cc, cot);
}

}  // namespace mnk

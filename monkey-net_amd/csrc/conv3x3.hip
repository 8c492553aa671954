// 3x3 / pad 1 / stride 1 convolution on folded NHWC frames as an implicit GEMM on the gfx950 fp32 matrix
// cores (v_mfma_f32_32x32x2_f32: exact fp32, 64 FLOP/clk/SIMD, 157 TFLOP/s chip peak).
//
// Replaces nn.Conv3d(kernel (1,3,3), padding (0,1,1)) of the reference (modules/util.py:52-55,79,98,176) in
// forward, data-gradient (same kernel, flipped/transposed packed weights) and weight-gradient form.
//
//   forward / dgrad GEMM:  Y[m][r] = sum_{chunk,tap,k} X[pix(m)+off(tap)][16*chunk+k] * W(r, 16*chunk+k, tap)
//       M = N*H*W pixels (rows, NHWC so channels are contiguous), N = output channels, K = 9 * (C0p + C1p).
//       K is walked chunk-major / tap-minor: the nine taps of one 16-channel chunk touch the same (halo of) pixel
//       rows and the same 36-byte (row, k) weight groups back to back, so both operands are re-read from L1.
//       The weights are re-packed per use to Wp[row][chunk][tap][16] (forward: row = co; data gradient: row = ci with
//       flipped taps), so a K step is one contiguous 64-byte read per row.  (Reading the parameter in place with
//       36-byte strides was measured 20 % slower: a 128 x 16 tile then spans 72 KB of cache lines per chunk.)
//       In LDS both operands are K-contiguous, so each lane fetches its MFMA fragment as one ds_read_b128:
//       lane (i = l&31, kk = l>>5) holds k = 8*kh + 4*kk + e (e = 0..3) of row i -- a permutation of K inside the
//       16-wide K step that is applied identically to A and B, hence harmless.
//   weight-gradient GEMM:  dW[co][ci*9+tap] = sum_p dY[p][co] * X[p+off(tap)][ci]   (K = pixels, split over blocks)
//
// Tiling: 256 threads = 4 waves; block tile BM x BN (128x128 default) staged through LDS in 16-deep K steps,
// double buffered with register prefetch (one barrier per step); wave tile = (BM/WM) x (BN/WN) as 32x32 MFMA
// tiles, i.e. 64 accumulator VGPRs for 64x64.  LDS rows are padded to 20 floats (80 B = 5 x 16 B, odd) so the
// 16-lane groups of a ds_read_b128 hit 16 distinct 16-B slots (no bank conflicts).
// Small problems (deep hourglass levels: 4x4 / 2x2 maps with 1024 channels) are split along K across blockIdx.z
// with a deterministic second-pass reduction (no atomics), which also applies bias and the residual add.
#include <stdlib.h>
#include <string.h>
#include <type_traits>

#include "mnk_common.h"
#include "pack_tile.h"

using namespace mnk;

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

__device__ __forceinline__ int round_up16(int v) { return (v + 15) & ~15; }

constexpr int BK = 16;        // K step (floats)
constexpr int LDS_K = 20;     // padded LDS row (floats)

// The BatchNorm (+ activation) layer whose OUTPUT this launch's output is the gradient of (a data-gradient launch: dz of the
// layer in front).  With it the fused column sums of the epilogue are that layer's backward statistics
//     sum g,  sum g * xhat      g = act'((y - mean) * scale + beta) * dz,   xhat = (y - mean) * invstd
// (sync_batchnorm/batchnorm.py's backward through F.batch_norm; what colsum2_partial_kernel<BwdLoader> makes in a pass of its
// own over y and dz) instead of sum v, sum v^2 -- the tile of dz is in registers here, only y is read.  slope: < 0 no
// activation, 0 ReLU, > 0 LeakyReLU.  y == nullptr: plain statistics.
struct BnBwdSrc {
    const float* y;
    const float *mean, *invstd, *scale, *beta;
    int ld;
    float slope;
};

struct ConvArgs {
    const float* x0;
    const float* x1;
    int ld0, ld1, C0, C1, C0p, C1p;
    int ups;
    const float* wp;  // packed weights [rows][chunks][9][16]
    const float* bias;
    const float* residual;
    int ld_res;
    float* y;
    int ld_y;
    int N, H, W, Cout;   // output frames / height / width / channels
    int Hi, Wi;          // input height / width (before the optional x2 up-sampling view): H,W for 3x3 pad 1
    int ntaps, kw, pad;  // kernel taps (kh*kw), kernel width, zero padding (3x3: 9, 3, 1; discriminator 4x4: 16, 4, 0)
    long M;
    int chunks;        // (C0p + C1p) / 16
    int ksteps;        // ntaps * chunks, step s = chunk * ntaps + tap
    int ksteps_per_split;
    int splits;
    float* ws;         // [splits][M][ldw] partial sums when splits > 1
    int ldw;
    float* stats;      // optional [gridDim.x][2][ld_y]: per-block column sums / sums of squares of the written output
    int xcd;           // re-chunk the launch order per XCD (xcd_tile)
    int clean;         // the sources' pad channels [C, ld) hold zeros (finite values): the 3x3 fast loader may be used
    unsigned mulW, shW, mulH, shH;   // division of an output pixel index (< 2^31) by W and H (fast_div)
    // ---- K x K loader generalisations (ActLoaderK) ---------------------------------------------------------------
    int stride;        // input pixel of output (h, w), tap (ky, kx): (h * stride + ky - pad_y, w * stride + kx - pad_x)
    int pad_x;         // left padding (`pad` is the top padding); -1: same as `pad`
    // ---- sub-pixel ("phase") form of [nearest x2 up-sampling -> 3x3 / pad 1] (UpBlock3D, modules/util.py:83-85) -------
    // Output pixel (2i + a, 2j + b) only sees the 2 x 2 low-resolution neighbourhood rows {i + a - 1, i + a}, columns
    // {j + b - 1, j + b}, with weights that are sums of the 3x3 taps falling on the same input pixel: four 2x2
    // convolutions on the low-resolution input (one per phase (a, b)) instead of one 3x3 convolution on the 4x larger
    // up-sampled view -- 4 instead of 9 multiply-adds per output and channel pair, same result up to the rounding of the
    // pre-summed weights.  phases = 4: tile bx belongs to phase bx / tiles_per_phase; H, W, M are the LOW-resolution
    // geometry of one phase; the weights of phase p start at wp + p * phase_wstride; output rows are scattered to
    // (2i + a, 2j + b) of the (N, 2H, 2W) tensor.
    int phases, tiles_per_phase;
    long phase_wstride;
    BnBwdSrc bnb;
};

// buffer resource from values the compiler cannot prove wave-uniform (e.g. derived from a 64-bit division): pin the
// pointer into scalar registers, otherwise every buffer load becomes a readfirstlane "waterfall" loop
__device__ __forceinline__ __amdgpu_buffer_rsrc_t uniform_rsrc(const void* p, unsigned num_records) {
    unsigned long v = (unsigned long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((int)(unsigned)v);
    const unsigned hi = __builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    const unsigned nr = __builtin_amdgcn_readfirstlane((int)num_records);
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long)hi << 32) | lo), 0, (int)nr, 0x00020000);
}

struct TrueTag { static constexpr bool value = true; };
struct FalseTag { static constexpr bool value = false; };

// n / d for n < 2^31 with host-made constants (fast_div_consts): mulhi + shift, or a shift alone when mul == 0
__device__ __forceinline__ unsigned fast_div(unsigned n, unsigned mul, unsigned sh) {
    return mul ? __umulhi(n, mul) >> sh : n >> sh;
}

// Workgroups are handed to the 8 XCDs round-robin in launch order (x fastest), and every XCD has its own L2: with the
// plain mapping each XCD touches every weight tile and every pixel tile of a layer, so both operands cross the fabric
// up to 8 times.  Re-chunking the launch order -- XCD class c = L % 8 works on the contiguous range
// [start(c), start(c) + count(c)) of the logical (x fastest, then y) tile order -- gives each XCD a few complete rows
// of the tile grid: one operand is fetched once per chip, the other once per XCD that needs it.  (Placement is not
// guaranteed by the hardware; this is a locality heuristic only -- any mapping is correct.)
__device__ __forceinline__ void xcd_tile(int enable, int& bx, int& by) {
    bx = blockIdx.x;
    by = blockIdx.y;
    const unsigned gx = gridDim.x, per_z = gx * gridDim.y;
    if (!enable || per_z < 16) return;
    const unsigned L = blockIdx.x + gx * blockIdx.y, c = L & 7u, base = per_z >> 3, rem = per_z & 7u;
    const unsigned logical = c * base + (c < rem ? c : rem) + (L >> 3);
    by = (int)(logical / gx);
    bx = (int)(logical - (unsigned)by * gx);
}

// ---- the activation-side loader of the implicit GEMM (shared by the 32x32 and 16x16 tile kernels) -------------------
// Everything a K step needs per row is precomputed (frame base, row / column, validity bit masks of the kernel rows
// and columns); the per-step gather is branch-free: a tap outside the image reads the clamped pixel and is zeroed on
// its way to LDS, rows beyond M are clamped to the last pixel (their results are never stored).  K-step cursor
// (step = chunk * ntaps + ky * kw + kx) advances without divisions or branches.
template <int RA, int NST = 1>      // NST: register stages (K steps in flight between their global loads and their LDS stores)
struct ActLoader {
    int prow[RA], ph[RA], pw[RA];
    unsigned pmask[RA];            // bits 0..7: kernel rows inside the image, bits 8..15: kernel columns
    float4 ra[NST][RA];
    int tail[NST][RA];             // real channels in ra[st][j] (<= 0: tap outside the image / chunk beyond C)
    int chunk, ky, kx, khh, Hs, Ws, hmax, wmax, lq;

    __device__ __forceinline__ void setup(const ConvArgs& a, long m0, int lrow, int lq_, int s_begin, int, int) {
        lq = lq_;
        Hs = a.ups ? a.Hi >> 1 : a.Hi;
        Ws = a.ups ? a.Wi >> 1 : a.Wi;
        khh = a.ntaps / a.kw;
        hmax = a.Hi - 1;
        wmax = a.Wi - 1;
#pragma unroll
        for (int j = 0; j < RA; ++j) {
            long ml = m0 + lrow + 64 * j;
            if (ml > a.M - 1) ml = a.M - 1;
            const unsigned m = (unsigned)ml, tt = fast_div(m, a.mulW, a.shW), fr = fast_div(tt, a.mulH, a.shH);
            pw[j] = (int)(m - tt * (unsigned)a.W);
            ph[j] = (int)(tt - fr * (unsigned)a.H);
            prow[j] = (int)fr * Hs * Ws;
            unsigned mk = 0;
            for (int y = 0; y < khh; ++y) {
                const int hh = ph[j] + y - a.pad;
                if (hh >= 0 && hh < a.Hi) mk |= 1u << y;
            }
            for (int x = 0; x < a.kw; ++x) {
                const int ww = pw[j] + x - a.pad;
                if (ww >= 0 && ww < a.Wi) mk |= 256u << x;
            }
            pmask[j] = mk;
#pragma unroll
            for (int st = 0; st < NST; ++st) tail[st][j] = 0;
        }
        chunk = s_begin / a.ntaps;
        ky = (s_begin - chunk * a.ntaps) / a.kw;
        kx = (s_begin - chunk * a.ntaps) - ky * a.kw;
    }
    // issue the global loads of the cursor's K step into register stage ST, then advance the cursor
    template <int ST = 0>
    __device__ __forceinline__ void load(const ConvArgs& a) {
        const int c0 = chunk * BK;
        const bool second = c0 >= a.C0p;
        const int cbase = second ? c0 - a.C0p : c0;
        const float* src = second ? a.x1 : a.x0;
        const int ld = second ? a.ld1 : a.ld0, C = second ? a.C1 : a.C0;
        const int ch = cbase + lq * 4;
        const int tl = C - ch;
        const int che = tl > 0 ? ch : 0;
        const int dy = ky - a.pad, dx = kx - a.pad;
#pragma unroll
        for (int j = 0; j < RA; ++j) {
            const bool ok = (pmask[j] >> ky) & (pmask[j] >> (8 + kx)) & 1u;
            int hh = ph[j] + dy, ww = pw[j] + dx;     // clamped into the image: the load is unconditional
            hh = hh < 0 ? 0 : (hh > hmax ? hmax : hh);
            ww = ww < 0 ? 0 : (ww > wmax ? wmax : ww);
            const unsigned off = (unsigned)(prow[j] + (hh >> a.ups) * Ws + (ww >> a.ups)) * (unsigned)ld + (unsigned)che;
            ra[ST][j] = *reinterpret_cast<const float4*>(src + off);
            tail[ST][j] = ok ? tl : 0;
        }
        const int kx1 = kx + 1;
        const bool wx = kx1 == a.kw;
        kx = wx ? 0 : kx1;
        const int ky1 = ky + (wx ? 1 : 0);
        const bool wy = ky1 == khh;
        ky = wy ? 0 : ky1;
        chunk += wy ? 1 : 0;
    }
    // row j on its way to LDS: pad channels of the producer may hold anything, taps outside the image are zero
    template <int ST = 0>
    __device__ __forceinline__ float4 masked(int j) const {
        float4 v = ra[ST][j];
        v.x = tail[ST][j] < 1 ? 0.f : v.x;
        v.y = tail[ST][j] < 2 ? 0.f : v.y;
        v.z = tail[ST][j] < 3 ? 0.f : v.z;
        v.w = tail[ST][j] < 4 ? 0.f : v.w;
        return v;
    }
};

// ---- the same loader for the hot case (3x3, pad 1, sources with clean pad channels) with the fewest instructions per
// K step -- every vector instruction between two MFMAs costs matrix-pipe time on this hardware (profiles/README.md:
// dropping only the data masks of the generic loader was worth +9 %).  Raw buffer loads relative to a per-block base
// (offsets stay far below 2^30; num_records = 2^30): an out-of-image tap or a chunk beyond the channel count gets bit
// 30 added to its offset and the hardware returns zeros -- no clamping, no data masks, no branches.  Per row and step:
// one add, one bit-field extract, one and-or (+ five for the parity shifts of the nearest x2 up-sampling view).
template <int RA, bool UPS, int NST = 1>
struct ActLoader3 {
    unsigned b0[RA], b1[RA];       // byte offset of the row's centre pixel in source 0 / 1 (relative to the block base)
    unsigned inv[RA];              // bit t: tap t lies outside the image
    unsigned par[RA];              // up-sampling: bit 0 h even, 1 h odd, 2 w even, 3 w odd (bit 4 stays 0)
    float4 ra[NST][RA];
    __amdgpu_buffer_rsrc_t r0, r1;
    int chunk, ky, kx, Ws, lq4;

    __device__ __forceinline__ void setup(const ConvArgs& a, long m0, int lrow, int lq_, int s_begin, int, int) {
        lq4 = lq_ * 16;
        const int Hs = UPS ? a.Hi >> 1 : a.Hi;
        Ws = UPS ? a.Wi >> 1 : a.Wi;
        const unsigned mb = (unsigned)(m0 < a.M ? m0 : a.M - 1), tb = fast_div(mb, a.mulW, a.shW),
                       fb = fast_div(tb, a.mulH, a.shH);
        const long rowidx0 = (long)fb * Hs + ((int)(tb - fb * (unsigned)a.H) >> (UPS ? 1 : 0));
        long pbase = rowidx0 * Ws - Ws - 1;
        if (pbase < 0) pbase = 0;
        r0 = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x0 + pbase * a.ld0), 0, 0x40000000, 0x00020000);
        r1 = __builtin_amdgcn_make_buffer_rsrc((void*)((a.x1 ? a.x1 : a.x0) + pbase * (a.x1 ? a.ld1 : a.ld0)), 0, 0x40000000,
                                               0x00020000);
#pragma unroll
        for (int j = 0; j < RA; ++j) {
            long ml = m0 + lrow + 64 * j;
            if (ml > a.M - 1) ml = a.M - 1;
            const unsigned m = (unsigned)ml, tt = fast_div(m, a.mulW, a.shW), fr = fast_div(tt, a.mulH, a.shH);
            const int w = (int)(m - tt * (unsigned)a.W), h = (int)(tt - fr * (unsigned)a.H);
            const long rel = ((long)fr * Hs + (h >> (UPS ? 1 : 0))) * Ws + (w >> (UPS ? 1 : 0)) - pbase;
            b0[j] = (unsigned)(rel * a.ld0 * 4);
            b1[j] = (unsigned)(rel * a.ld1 * 4);
            unsigned mk = 0;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int hh = h + t / 3 - 1, ww = w + t % 3 - 1;
                if (hh < 0 || hh >= a.Hi || ww < 0 || ww >= a.Wi) mk |= 1u << t;
            }
            inv[j] = mk;
            par[j] = ((h & 1) ? 2u : 1u) | ((w & 1) ? 8u : 4u);
        }
        chunk = s_begin / 9;
        ky = (s_begin - chunk * 9) / 3;
        kx = (s_begin - chunk * 9) - ky * 3;
    }
    template <int ST = 0>
    __device__ __forceinline__ void load(const ConvArgs& a) {
        const int c0 = chunk * BK;
        const bool second = c0 >= a.C0p;
        const int cbase = second ? c0 - a.C0p : c0;
        const int ldb = (second ? a.ld1 : a.ld0) * 4, C = second ? a.C1 : a.C0;
        const __amdgpu_buffer_rsrc_t rs = second ? r1 : r0;
        const int tap = ky * 3 + kx;
        // per thread: channel offset, + bit 30 when the whole float4 lies beyond the channel count
        unsigned st = (unsigned)(cbase * 4 + lq4);
        st += (cbase * 4 + lq4 >= C * 4) ? 0x40000000u : 0u;
        int s_dr = 0, s_dc = 0, rbit = 4, cbit = 4;
        if (UPS) {
            s_dr = ky == 0 ? -Ws * ldb : (ky == 2 ? Ws * ldb : 0);
            s_dc = kx == 0 ? -ldb : (kx == 2 ? ldb : 0);
            rbit = ky == 0 ? 0 : (ky == 2 ? 1 : 4);
            cbit = kx == 0 ? 2 : (kx == 2 ? 3 : 4);
        } else {
            st += (unsigned)(((ky - 1) * Ws + (kx - 1)) * ldb);
        }
#pragma unroll
        for (int j = 0; j < RA; ++j) {
            unsigned off = (second ? b1[j] : b0[j]) + st;
            if (UPS) {
                const int nr = __builtin_amdgcn_sbfe(par[j], rbit, 1), nc = __builtin_amdgcn_sbfe(par[j], cbit, 1);
                off += (unsigned)(nr & s_dr) + (unsigned)(nc & s_dc);
            }
            const int bad = __builtin_amdgcn_sbfe(inv[j], tap, 1);            // 0 or -1
            off = ((unsigned)bad & 0x40000000u) | off;
            ra[ST][j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
        }
        const int kx1 = kx + 1;
        const bool wx = kx1 == 3;
        kx = wx ? 0 : kx1;
        const int ky1 = ky + (wx ? 1 : 0);
        const bool wy = ky1 == 3;
        ky = wy ? 0 : ky1;
        chunk += wy ? 1 : 0;
    }
    template <int ST = 0>
    __device__ __forceinline__ float4 masked(int j) const { return ra[ST][j]; }
};

// ---- the same idea for any K x K kernel with stride 1 and padding `pad` (the discriminator's 4x4 / pad 0 forward and its
// pad 3 data gradient), sources with clean pad channels, no up-sampled view: per-row base offset + one validity bit per
// tap (K*K <= 32); an out-of-image tap or a channel chunk beyond C reads zeros through the buffer range check.
template <int RA, int NST = 1>
struct ActLoaderK {
    unsigned b0[RA], b1[RA];       // byte offset of input pixel (h * stride, w * stride) -- tap (pad_y, pad_x) -- relative to the
                                   // block base; may lie outside the image (pad > 0): it is only a base for the tap arithmetic
    unsigned inv[RA];              // bit ky * kw + kx: the tap lies outside the image
    float4 ra[NST][RA];
    __amdgpu_buffer_rsrc_t r0, r1;
    int chunk, ky, kx, khh, lq4, pady, padx;

    __device__ __forceinline__ void setup(const ConvArgs& a, long m0, int lrow, int lq_, int s_begin, int pad_y, int pad_x) {
        lq4 = lq_ * 16;
        khh = a.ntaps / a.kw;
        pady = pad_y;
        padx = pad_x;
        const int sd = a.stride;
        const unsigned mb = (unsigned)(m0 < a.M ? m0 : a.M - 1), tb = fast_div(mb, a.mulW, a.shW),
                       fb = fast_div(tb, a.mulH, a.shH);
        // lowest address a valid tap of this block can have: input pixel (h0 * stride - pad_y, -pad_x) of the first row's frame
        long pbase = ((long)fb * a.Hi + (int)(tb - fb * (unsigned)a.H) * sd - pad_y) * a.Wi - pad_x;
        if (pbase < 0) pbase = 0;
        r0 = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x0 + pbase * a.ld0), 0, 0x40000000, 0x00020000);
        r1 = __builtin_amdgcn_make_buffer_rsrc((void*)((a.x1 ? a.x1 : a.x0) + pbase * (a.x1 ? a.ld1 : a.ld0)), 0, 0x40000000,
                                               0x00020000);
#pragma unroll
        for (int j = 0; j < RA; ++j) {
            long ml = m0 + lrow + 64 * j;
            if (ml > a.M - 1) ml = a.M - 1;
            const unsigned m = (unsigned)ml, tt = fast_div(m, a.mulW, a.shW), fr = fast_div(tt, a.mulH, a.shH);
            const int w = (int)(m - tt * (unsigned)a.W) * sd, h = (int)(tt - fr * (unsigned)a.H) * sd;
            const long rel = ((long)fr * a.Hi + h) * a.Wi + w - pbase;
            b0[j] = (unsigned)(rel * a.ld0 * 4);
            b1[j] = (unsigned)(rel * a.ld1 * 4);
            unsigned mk = 0, bit = 1;
            for (int y = 0; y < khh; ++y) {
                const int hh = h + y - pad_y;
                const bool rowbad = hh < 0 || hh >= a.Hi;
                for (int x = 0; x < a.kw; ++x, bit <<= 1) {
                    const int ww = w + x - pad_x;
                    if (rowbad || ww < 0 || ww >= a.Wi) mk |= bit;
                }
            }
            inv[j] = mk;
        }
        chunk = s_begin / a.ntaps;
        ky = (s_begin - chunk * a.ntaps) / a.kw;
        kx = (s_begin - chunk * a.ntaps) - ky * a.kw;
    }
    template <int ST = 0>
    __device__ __forceinline__ void load(const ConvArgs& a) {
        const int c0 = chunk * BK;
        const bool second = c0 >= a.C0p;
        const int cbase = second ? c0 - a.C0p : c0;
        const int ldb = (second ? a.ld1 : a.ld0) * 4, C = second ? a.C1 : a.C0;
        const __amdgpu_buffer_rsrc_t rs = second ? r1 : r0;
        const int tap = ky * a.kw + kx;
        unsigned st = (unsigned)(cbase * 4 + lq4);
        st += (cbase * 4 + lq4 >= C * 4) ? 0x40000000u : 0u;
        st += (unsigned)(((ky - pady) * a.Wi + (kx - padx)) * ldb);
#pragma unroll
        for (int j = 0; j < RA; ++j) {
            unsigned off = (second ? b1[j] : b0[j]) + st;
            const int bad = __builtin_amdgcn_sbfe(inv[j], tap, 1);            // 0 or -1
            off = ((unsigned)bad & 0x40000000u) | off;
            ra[ST][j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
        }
        const int kx1 = kx + 1;
        const bool wx = kx1 == a.kw;
        kx = wx ? 0 : kx1;
        const int ky1 = ky + (wx ? 1 : 0);
        const bool wy = ky1 == khh;
        ky = wy ? 0 : ky1;
        chunk += wy ? 1 : 0;
    }
    template <int ST = 0>
    __device__ __forceinline__ float4 masked(int j) const { return ra[ST][j]; }
};

template <int RA, int MODE, int NST = 1> struct LoaderSel { typedef ActLoader<RA, NST> type; };
template <int RA, int NST> struct LoaderSel<RA, 3, NST> { typedef ActLoaderK<RA, NST> type; };
template <int RA, int NST> struct LoaderSel<RA, 1, NST> { typedef ActLoader3<RA, false, NST> type; };
template <int RA, int NST> struct LoaderSel<RA, 2, NST> { typedef ActLoader3<RA, true, NST> type; };

#ifndef MNK_IGEMM_OCC
#define MNK_IGEMM_OCC 3                       // waves per SIMD = blocks per CU the register budget is held to
#endif
#ifndef MNK_IGEMM_NACC
#define MNK_IGEMM_NACC 2                      // accumulator sets of a one-tile wave (64x64 / 128x32 block tiles)
#endif
#ifndef MNK_IGEMM_NST
#define MNK_IGEMM_NST 2                       // register stages: K steps between a step's global loads and its LDS stores
#endif

// -DMNK_PHASE_CLOCKS (an experiment build, tools/phase_probe.py): thread 0 of every block of the 32x32-tile kernel stamps the
// 100 MHz wall clock at its entry, in front of its K loop, behind it and at its exit
#ifdef MNK_PHASE_CLOCKS
__device__ unsigned long long mnk_phase_log[4 * 16384];
__device__ unsigned long long mnk_phase_sclk[2 * 16384];      // the shader clock (s_memtime) in front of / behind the K loop
#define MNK_PHASE(i)                                                                                              \
    do {                                                                                                          \
        const unsigned lin__ = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);                    \
        if (threadIdx.x == 0 && lin__ < 16384u) {                                                                 \
            mnk_phase_log[4 * lin__ + (i)] = wall_clock64();                                                      \
            if ((i) == 1 || (i) == 2) mnk_phase_sclk[2 * lin__ + (i) - 1] = clock64();                            \
        }                                                                                                         \
    } while (0)
#else
#define MNK_PHASE(i) ((void)0)
#endif

constexpr int LDS_H = 24;     // padded LDS row of the bf16 planes (16 + 8 halves = 48 bytes: conflict-free b128 fragment reads)

// GM: 0 -- v_mfma_f32_32x32x2_f32 on fp32 tiles; 1 -- the same products on the bf16 matrix cores: the loaders split every fp32
// operand into three bf16 planes on its way to LDS and a K step of 16 is six v_mfma_f32_32x32x16_bf16 per tile (mnk_common.h)
template <int BM, int BN, int WM, int WN, int MODE, int GM = 0>     // MODE: 0 generic loader, 1 / 2 the 3x3 fast loader (plain / x2 up-sampled)
__global__ void __launch_bounds__(256, MNK_IGEMM_OCC) conv3x3_igemm_kernel(ConvArgs a) {
    MNK_PHASE(0);
    constexpr int RA = BM / 64;               // A rows per thread per K step
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    // A wave that owns ONE 32x32 tile would run all its MFMAs as one dependent chain on one accumulator: an MFMA that follows
    // its predecessor on the same accumulator with other instructions in between waits for the write-back (the back-to-back
    // forwarding path only serves adjacent issues), and the loop has loads, LDS traffic and address arithmetic between every
    // pair.  Two accumulator sets fed by alternating K elements make neighbours independent; they are added once at the end
    // (which also halves the length of the fp32 summation chain: K = 9 * Cin <= 18 522 terms).
    constexpr int NACC = (TM * TN == 1) ? MNK_IGEMM_NACC : 1;
    // 8 (16) registers per stage; the 128x128 tile has none to spare, and the bf16x3 form measured 0.05 ms per iteration
    // faster with one stage (profiles/r06_knob_ab_log.txt, v16): its six MFMAs per K step already cover the load latency
    constexpr int NST = (BM * BN <= 64 * 128 && GM == 0) ? MNK_IGEMM_NST : 1;
    static_assert(WM * WN == 4, "4 waves per block");
    static_assert(NST == 1 || NST == 2, "one or two register stages");
    // one LDS image, two views: fp32 rows [2][rows][LDS_K] (GM 0) / three bf16 planes [2][3][rows][LDS_H] (GM 1)
    constexpr int A_BYTES = GM ? 2 * 3 * BM * LDS_H * 2 : 2 * BM * LDS_K * 4, B_BYTES = GM ? 2 * 3 * BN * LDS_H * 2 : 2 * BN * LDS_K * 4;
    __shared__ __attribute__((aligned(16))) unsigned char smem_a[A_BYTES], smem_b[B_BYTES];
    float (*const As)[BM][LDS_K] = reinterpret_cast<float (*)[BM][LDS_K]>(smem_a);
    float (*const Bs)[BN][LDS_K] = reinterpret_cast<float (*)[BN][LDS_K]>(smem_b);
    unsigned short (*const Ah)[3][BM][LDS_H] = reinterpret_cast<unsigned short (*)[3][BM][LDS_H]>(smem_a);
    unsigned short (*const Bh)[3][BN][LDS_H] = reinterpret_cast<unsigned short (*)[3][BN][LDS_H]>(smem_b);

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wm = wave / WN, wn = wave % WN;
    int bx, by;
    xcd_tile(a.xcd, bx, by);
    // sub-pixel form: the M tiles of phase (pa, pb) are [phase * tiles_per_phase, ...); each phase has its own weights
    // and its own top / left padding (rows i + pa - 1, i + pa of the low-resolution input)
    const int phase = a.phases > 1 ? bx / a.tiles_per_phase : 0;
    const int pa = phase >> 1, pb = phase & 1;
    const long m0 = (long)(bx - phase * a.tiles_per_phase) * BM;
    const int n0 = by * BN;
    const int split = blockIdx.z;
    const int s_begin = split * a.ksteps_per_split;
    int s_end = s_begin + a.ksteps_per_split;
    if (s_end > a.ksteps) s_end = a.ksteps;
    const float* const wpb = a.wp + (long)phase * a.phase_wstride;

    // ---- per-thread global->LDS assignment: row inside a 64-row slab, float4 column (4 channels) --------------
    const int lrow = t >> 2, lq = t & 3;
    typename LoaderSel<RA, MODE, NST>::type L;
    L.setup(a, m0, lrow, lq, s_begin, a.pad - pa, (a.pad_x < 0 ? a.pad : a.pad_x) - pb);
    constexpr int RB = (BN + 63) / 64;        // B rows per thread per K step
    const long KT = (long)a.ksteps * BK;     // packed row length
    static_assert(RB <= 2, "at most two weight rows per thread");
    const int wco0 = n0 + lrow, wco1 = n0 + lrow + 64;    // rows beyond Cout: clamped, never stored
    const unsigned woff0 = (unsigned)((wco0 < a.Cout ? wco0 : a.Cout - 1) * KT) + lq * 4;
    const unsigned woff1 = (unsigned)((wco1 < a.Cout ? wco1 : a.Cout - 1) * KT) + lq * 4;
    float4 rb0a, rb0b, rb1a, rb1b;            // weight rows of register stage 0 (a) / 1 (b): scalars, not arrays -- an array
                                              // captured by the lambdas below ends up in scratch memory

    // global loads of K step s into register stage ST (the activation loader keeps its own cursor: steps in order)
    auto load_step = [&](int s, auto st_tag) __attribute__((always_inline)) {
        constexpr int ST = decltype(st_tag)::value;
        L.template load<ST>(a);
        const float* wsrc_ptr = wpb + (long)s * BK;
        if constexpr (ST == 0) {
            rb0a = *reinterpret_cast<const float4*>(wsrc_ptr + woff0);
            if constexpr (RB > 1) rb1a = *reinterpret_cast<const float4*>(wsrc_ptr + woff1);
        } else {
            rb0b = *reinterpret_cast<const float4*>(wsrc_ptr + woff0);
            if constexpr (RB > 1) rb1b = *reinterpret_cast<const float4*>(wsrc_ptr + woff1);
        }
    };
    auto store_step = [&](int buf, auto st_tag) __attribute__((always_inline)) {
        constexpr int ST = decltype(st_tag)::value;
        if constexpr (GM == 0) {
#pragma unroll
            for (int j = 0; j < RA; ++j) *reinterpret_cast<float4*>(&As[buf][lrow + 64 * j][lq * 4]) = L.template masked<ST>(j);
            if (BN >= 64 || lrow < BN) *reinterpret_cast<float4*>(&Bs[buf][lrow][lq * 4]) = ST == 0 ? rb0a : rb0b;
            if constexpr (RB > 1) *reinterpret_cast<float4*>(&Bs[buf][lrow + 64][lq * 4]) = ST == 0 ? rb1a : rb1b;
        } else {
            uint2 p0, p1, p2;
#pragma unroll
            for (int j = 0; j < RA; ++j) {
                mnk_split3(L.template masked<ST>(j), p0, p1, p2);
                *reinterpret_cast<uint2*>(&Ah[buf][0][lrow + 64 * j][lq * 4]) = p0;
                *reinterpret_cast<uint2*>(&Ah[buf][1][lrow + 64 * j][lq * 4]) = p1;
                *reinterpret_cast<uint2*>(&Ah[buf][2][lrow + 64 * j][lq * 4]) = p2;
            }
            if (BN >= 64 || lrow < BN) {
                mnk_split3(ST == 0 ? rb0a : rb0b, p0, p1, p2);
                *reinterpret_cast<uint2*>(&Bh[buf][0][lrow][lq * 4]) = p0;
                *reinterpret_cast<uint2*>(&Bh[buf][1][lrow][lq * 4]) = p1;
                *reinterpret_cast<uint2*>(&Bh[buf][2][lrow][lq * 4]) = p2;
            }
            if constexpr (RB > 1) {
                mnk_split3(ST == 0 ? rb1a : rb1b, p0, p1, p2);
                *reinterpret_cast<uint2*>(&Bh[buf][0][lrow + 64][lq * 4]) = p0;
                *reinterpret_cast<uint2*>(&Bh[buf][1][lrow + 64][lq * 4]) = p1;
                *reinterpret_cast<uint2*>(&Bh[buf][2][lrow + 64][lq * 4]) = p2;
            }
        }
    };

    f32x16 acc[NACC][TM][TN];
#pragma unroll
    for (int q = 0; q < NACC; ++q)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[q][i][j][r] = 0.f;

    const int fi = lane & 31, fk = lane >> 5;
    const int a_row0 = wm * (BM / WM) + fi, b_row0 = wn * (BN / WN) + fi;

    auto mfma_step = [&](int buf) __attribute__((always_inline)) {
        if constexpr (GM == 1) {
            // lane (fi, fk) holds k = 8 fk .. 8 fk + 7 of its row in every plane: one b128 per plane and operand row
            mnk_bf16x8 ha[3][TM], hb[3][TN];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    ha[pl][i] = mnk_as_bf16x8(*reinterpret_cast<const uint4*>(&Ah[buf][pl][a_row0 + 32 * i][fk * 8]));
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    hb[pl][j] = mnk_as_bf16x8(*reinterpret_cast<const uint4*>(&Bh[buf][pl][b_row0 + 32 * j][fk * 8]));
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    constexpr int Q = NACC - 1;       // (two accumulator sets of a one-tile wave: neighbours independent)
                    acc[0][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ha[1][i], hb[1][j], acc[0][i][j], 0, 0, 0);
                    acc[Q][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ha[2][i], hb[0][j], acc[Q][i][j], 0, 0, 0);
                    acc[0][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ha[0][i], hb[2][j], acc[0][i][j], 0, 0, 0);
                    acc[Q][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ha[1][i], hb[0][j], acc[Q][i][j], 0, 0, 0);
                    acc[0][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ha[0][i], hb[1][j], acc[0][i][j], 0, 0, 0);
                    acc[Q][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ha[0][i], hb[0][j], acc[Q][i][j], 0, 0, 0);
                }
            return;
        }
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
            float4 fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                fa[i] = *reinterpret_cast<const float4*>(&As[buf][a_row0 + 32 * i][kh * 8 + fk * 4]);
#pragma unroll
            for (int j = 0; j < TN; ++j)
                fb[j] = *reinterpret_cast<const float4*>(&Bs[buf][b_row0 + 32 * j][kh * 8 + fk * 4]);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    constexpr int Q = NACC - 1;
                    acc[0][i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].x, fb[j].x, acc[0][i][j], 0, 0, 0);
                    acc[Q][i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].y, fb[j].y, acc[Q][i][j], 0, 0, 0);
                    acc[0][i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].z, fb[j].z, acc[0][i][j], 0, 0, 0);
                    acc[Q][i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].w, fb[j].w, acc[Q][i][j], 0, 0, 0);
                }
        }
    };
    using St0 = std::integral_constant<int, 0>;
    using St1 = std::integral_constant<int, NST - 1>;

    // Pipeline.  K step t (relative to s_begin) travels in register stage t % NST and lands in LDS buffer t & 1.
    //   NST = 1: the registers always hold the step after the one in LDS.  In step s the wave parks step s+1 in the other
    //            LDS buffer (loaded a whole step ago), issues the global loads of step s+2 and then runs the MFMAs of step s.
    //   NST = 2: two steps are in flight in registers: step s parks step s+1, issues the loads of step s+3 into the stage
    //            that store just freed, runs the MFMAs of step s -- a load has two whole steps to arrive (one step of eight
    //            MFMAs does not cover an HBM miss when few waves share the SIMD).
    // Loads, address arithmetic and LDS writes sit in the MFMA shadow; one barrier per step; the steady-state loop body is
    // branch-free (two steps per trip: LDS buffer and register stage are compile-time constants); the tail is peeled.
    // (Two steps per barrier -- four LDS buffers, half the barriers -- was built and measured in round 3: the per-layer
    // bench unchanged, the whole step 10.90 vs 10.79 ms with 40 KB of LDS per block: removed.  profiles/r03_knob_ab_log.txt)
    const int n = s_end - s_begin;
    if (n > 0) {
        load_step(s_begin, St0{});
        store_step(0, St0{});
        if (n > 1) load_step(s_begin + 1, St1{});
        if (NST == 2 && n > 2) load_step(s_begin + 2, St0{});
    }
    if (NST == 2) MNK_WAIT_VMEM();            // exact wait counts inside the loop (see the macro)
    __syncthreads();
    MNK_PHASE(1);
    int s = 0;
    if constexpr (NST == 2) {
        for (; s + 4 < n; s += 2) {
            store_step(1, St1{});
            load_step(s_begin + s + 3, St1{});
            mfma_step(0);
            __syncthreads();
            store_step(0, St0{});
            load_step(s_begin + s + 4, St0{});
            mfma_step(1);
            __syncthreads();
        }
        for (; s < n; ++s) {                  // at most four steps (s is even here)
            if ((s & 1) == 0) {
                if (s + 1 < n) store_step(1, St1{});
                if (s + 3 < n) load_step(s_begin + s + 3, St1{});
                mfma_step(0);
            } else {
                if (s + 1 < n) store_step(0, St0{});
                if (s + 3 < n) load_step(s_begin + s + 3, St0{});
                mfma_step(1);
            }
            __syncthreads();
        }
    } else {
        for (; s + 3 < n; s += 2) {
            store_step(1, St0{});
            load_step(s_begin + s + 2, St0{});
            mfma_step(0);
            __syncthreads();
            store_step(0, St0{});
            load_step(s_begin + s + 3, St0{});
            mfma_step(1);
            __syncthreads();
        }
        for (; s < n; ++s) {
            if (s + 1 < n) store_step((s & 1) ^ 1, St0{});
            if (s + 2 < n) load_step(s_begin + s + 2, St0{});
            mfma_step(s & 1);
            __syncthreads();                  // (the last one: the epilogue reuses As for the column sums)
        }
    }
    MNK_PHASE(2);
    if constexpr (NACC == 2) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][0][0][r] += acc[1][0][0][r];
    }

    // ---- epilogue: D[row][col], col = lane&31 (-> co), row = (r&3) + 8*(r>>2) + 4*(lane>>5) (-> pixel) ------
    // bias is fetched once per column, residual values are fetched as a batch before the stores (no per-element
    // load -> wait -> store chains), stores are predicated.
    const bool split_out = a.splits > 1;
    const unsigned ldo = split_out ? (unsigned)a.ldw : (unsigned)a.ld_y;          // 32-bit element offsets (host check)
    // split partials of the sub-pixel form are phase-major: [split][phase][M][ldw]
    float* const obase = split_out ? a.ws + ((long)split * a.phases + phase) * a.M * a.ldw : a.y;
    const int co_lim = split_out ? a.ldw : a.ld_y;
    const bool use_res = !split_out && a.residual;
    const bool full = m0 + BM <= a.M;             // every row of the tile is a real pixel: no per-row guards
    const unsigned mrow0 = (unsigned)m0 + wm * (BM / WM) + 4 * fk, Mu = (unsigned)a.M;
    const bool scatter = a.phases > 1 && !split_out;   // row m = (n, i, j) of the phase -> pixel (2i + pa, 2j + pb) of y
    auto out_row = [&](unsigned m) __attribute__((always_inline)) -> unsigned {
        if (!scatter) return m;
        const unsigned tt = fast_div(m, a.mulW, a.shW), fr = fast_div(tt, a.mulH, a.shH);
        const unsigned j = m - tt * (unsigned)a.W, i = tt - fr * (unsigned)a.H;
        return ((fr * (unsigned)a.H + i) * 2u + (unsigned)pa) * (2u * (unsigned)a.W) + 2u * j + (unsigned)pb;
    };
    int cov[TN];
    float bv[TN], s1[TN], s2[TN];
    const bool bnb = a.bnb.y != nullptr && a.stats && !split_out;     // the column sums are a BatchNorm layer's backward statistics
    float bm[TN], bis[TN], bsc[TN], bbe[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        cov[j] = n0 + wn * (BN / WN) + 32 * j + fi;
        bv[j] = (!split_out && a.bias && cov[j] < a.Cout) ? a.bias[cov[j]] : 0.f;
        s1[j] = s2[j] = 0.f;
        const bool real = bnb && cov[j] < a.Cout;
        bm[j] = real ? a.bnb.mean[cov[j]] : 0.f;
        bis[j] = real ? a.bnb.invstd[cov[j]] : 0.f;
        bsc[j] = real ? a.bnb.scale[cov[j]] : 0.f;
        bbe[j] = real ? a.bnb.beta[cov[j]] : 0.f;
    }
    auto emit = [&](auto full_tag) __attribute__((always_inline)) {
        constexpr bool FULL = decltype(full_tag)::value;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int co = cov[j];
                const bool c_real = co < a.Cout, c_store = co < co_lim;
                const unsigned mb = mrow0 + 32 * i;
                const unsigned off0 = mb * ldo + (unsigned)co, roff0 = mb * (unsigned)a.ld_res + (unsigned)co;
                if (c_store) {
                    float rv[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) rv[r] = 0.f;
                    if (use_res && c_real) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const unsigned ro = (r & 3) + 8 * (r >> 2);
                            if (FULL || mb + ro < Mu) rv[r] = a.residual[roff0 + ro * (unsigned)a.ld_res];
                        }
                    }
                    float yv[16];
                    if (bnb) {
                        const unsigned yoff0 = mb * (unsigned)a.bnb.ld + (unsigned)co;
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const unsigned ro = (r & 3) + 8 * (r >> 2);
                            yv[r] = (c_real && (FULL || mb + ro < Mu)) ? a.bnb.y[yoff0 + ro * (unsigned)a.bnb.ld] : 0.f;
                        }
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const unsigned ro = (r & 3) + 8 * (r >> 2);
                        float v = acc[0][i][j][r];
                        if (!split_out) v = c_real ? (v + bv[j]) + rv[r] : 0.f;
                        if (FULL || mb + ro < Mu) {
                            if (scatter)
                                obase[out_row(mb + ro) * ldo + (unsigned)co] = v;
                            else
                                obase[off0 + ro * ldo] = v;
                            if (bnb) {
                                const float d = yv[r] - bm[j];
                                float g = v;
                                if (a.bnb.slope >= 0.f && !(fmaf(d, bsc[j], bbe[j]) > 0.f)) g *= a.bnb.slope;
                                s1[j] += g;
                                s2[j] = fmaf(g, d * bis[j], s2[j]);
                            } else {
                                s1[j] += v;
                                s2[j] = fmaf(v, v, s2[j]);
                            }
                        }
                    }
                }
            }
    };
    if (full)
        emit(TrueTag{});
    else
        emit(FalseTag{});
    // ---- fused BatchNorm statistics of the tensor just written (sync_batchnorm/batchnorm.py:60-62): per-block
    // column sums -> stats[blockIdx.x][2][ld_y]; the tiny final reduction over blocks is mnk_bn_stats_finish.
    if (a.stats && !split_out) {
        float* red = reinterpret_cast<float*>(smem_a);          // the main loop ended with a barrier: LDS is free
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            s1[j] += __shfl_xor(s1[j], 32);
            s2[j] += __shfl_xor(s2[j], 32);
            if (fk == 0) {
                const int col = wn * (BN / WN) + 32 * j + fi;
                red[(wm * 2 + 0) * BN + col] = s1[j];
                red[(wm * 2 + 1) * BN + col] = s2[j];
            }
        }
        __syncthreads();
        if (t < BN && n0 + t < a.ld_y) {
            float t1 = 0.f, t2 = 0.f;
#pragma unroll
            for (int w = 0; w < WM; ++w) {
                t1 += red[(w * 2 + 0) * BN + t];
                t2 += red[(w * 2 + 1) * BN + t];
            }
            float* sp = a.stats + (long)bx * 2 * a.ld_y;
            sp[n0 + t] = t1;
            sp[a.ld_y + n0 + t] = t2;
        }
    }
#ifdef MNK_PHASE_CLOCKS
    __builtin_amdgcn_s_waitcnt(0);            // the output stores of this wave have left (vmcnt / lgkmcnt / expcnt = 0)
#endif
    MNK_PHASE(3);
}

// ---- narrow-output variant on v_mfma_f32_16x16x4_f32 -----------------------------------------------------------
// For Cout <= 48 (the 45-channel refinement stack, the 10 / 13-channel heads, dgrad into 45 channels) a 32-wide MFMA
// tile wastes up to 3/4 of the matrix pipe.  Here the block tile is 128 pixels x BN (16 / 48) channels built from
// 16x16 tiles: 4 waves x 32 rows, each wave covers all BN columns (2 x BN/16 accumulator tiles of 4 VGPRs).
// Fragment layout of the 16x16x4 form: lane l holds A[i = l&15][k = l>>4], B[k = l>>4][j = l&15]; with K-contiguous
// LDS rows a lane reads the float4 at k = 4*(l>>4) .. +3 and feeds element e to MFMA e (K permuted identically for A
// and B); D: col = l&15, row = 4*(l>>4) + reg.  Loader, K order, split-K and epilogue semantics are those of the
// 32x32 kernel above.
typedef float f32x4 __attribute__((ext_vector_type(4)));

// GM = 1 (round 6): the products on the bf16 matrix cores (mnk_common.h).  v_mfma_f32_16x16x32_bf16 spans K = 32 = TWO K steps:
// the loop works on PAIRS of steps -- step 2p lands in LDS buffer 0, step 2p + 1 in buffer 1 (zeros behind an odd count), lane
// (fi, fk) reads k = 8 fk .. 8 fk + 7 of its row from buffer fk >> 1 -- six MFMAs per 16x16 tile and pair.  Two register stages
// hold the next pair while the MFMAs of this one run; two barriers per pair.
template <int BN, int MODE, int GM = 0>
__global__ void __launch_bounds__(256) conv3x3_igemm16_kernel(ConvArgs a) {
    constexpr int BM = 128, RA = 2, TM = 2, TN = BN / 16;
    constexpr int NSTG = GM ? 2 : 1;
    constexpr int A_BYTES = GM ? 2 * 3 * BM * LDS_H * 2 : 2 * BM * LDS_K * 4, B_BYTES = GM ? 2 * 3 * BN * LDS_H * 2 : 2 * BN * LDS_K * 4;
    __shared__ __attribute__((aligned(16))) unsigned char smem_a[A_BYTES], smem_b[B_BYTES];
    float (*const As)[BM][LDS_K] = reinterpret_cast<float (*)[BM][LDS_K]>(smem_a);
    float (*const Bs)[BN][LDS_K] = reinterpret_cast<float (*)[BN][LDS_K]>(smem_b);
    unsigned short (*const Ah)[3][BM][LDS_H] = reinterpret_cast<unsigned short (*)[3][BM][LDS_H]>(smem_a);
    unsigned short (*const Bh)[3][BN][LDS_H] = reinterpret_cast<unsigned short (*)[3][BN][LDS_H]>(smem_b);
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    int bx, by;
    xcd_tile(a.xcd, bx, by);
    const long m0 = (long)bx * BM;
    const int n0 = by * BN;
    const int split = blockIdx.z;
    const int s_begin = split * a.ksteps_per_split;
    int s_end = s_begin + a.ksteps_per_split;
    if (s_end > a.ksteps) s_end = a.ksteps;
    const int lrow = t >> 2, lq = t & 3;
    typename LoaderSel<RA, MODE, NSTG>::type L;
    L.setup(a, m0, lrow, lq, s_begin, a.pad, a.pad_x < 0 ? a.pad : a.pad_x);
    const long KT = (long)a.ksteps * BK;
    const int wco = n0 + (lrow < BN ? lrow : 0);           // rows beyond BN / Cout: clamped, never stored
    const unsigned woff = (unsigned)((wco < a.Cout ? wco : a.Cout - 1) * KT) + lq * 4;
    float4 rb, rb1;
    auto load_step = [&](int s) __attribute__((always_inline)) {
        L.load(a);
        rb = *reinterpret_cast<const float4*>(a.wp + (long)s * BK + woff);
    };
    auto store_step = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < RA; ++j) *reinterpret_cast<float4*>(&As[buf][lrow + 64 * j][lq * 4]) = L.masked(j);
        if (lrow < BN) *reinterpret_cast<float4*>(&Bs[buf][lrow][lq * 4]) = rb;
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;

    const int fi = lane & 15, fk = lane >> 4;     // row/col inside a 16-tile, k group 0..3
    if constexpr (GM == 1) {
        using S0 = std::integral_constant<int, 0>;
        using S1 = std::integral_constant<int, 1>;
        // loads of K step s into register stage ST (the activation loader keeps its own cursor: steps in order)
        auto load_h = [&](int s, auto st_tag) __attribute__((always_inline)) {
            constexpr int ST = decltype(st_tag)::value;
            L.template load<ST>(a);
            const float4 w = *reinterpret_cast<const float4*>(a.wp + (long)s * BK + woff);
            if constexpr (ST == 0) rb = w; else rb1 = w;
        };
        // stage ST -> the three planes of LDS buffer ST (zero: the missing second half of an odd pair)
        auto store_h = [&](auto st_tag, bool zero) __attribute__((always_inline)) {
            constexpr int ST = decltype(st_tag)::value;
            const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
            uint2 p0, p1, p2;
#pragma unroll
            for (int j = 0; j < RA; ++j) {
                mnk_split3(zero ? z : L.template masked<ST>(j), p0, p1, p2);
                *reinterpret_cast<uint2*>(&Ah[ST][0][lrow + 64 * j][lq * 4]) = p0;
                *reinterpret_cast<uint2*>(&Ah[ST][1][lrow + 64 * j][lq * 4]) = p1;
                *reinterpret_cast<uint2*>(&Ah[ST][2][lrow + 64 * j][lq * 4]) = p2;
            }
            if (lrow < BN) {
                mnk_split3(zero ? z : (ST == 0 ? rb : rb1), p0, p1, p2);
                *reinterpret_cast<uint2*>(&Bh[ST][0][lrow][lq * 4]) = p0;
                *reinterpret_cast<uint2*>(&Bh[ST][1][lrow][lq * 4]) = p1;
                *reinterpret_cast<uint2*>(&Bh[ST][2][lrow][lq * 4]) = p2;
            }
        };
        const int kb = fk >> 1, ko = (fk & 1) * 8;      // this lane's k group: LDS buffer and offset inside the step
        auto mfma_pair = [&]() __attribute__((always_inline)) {
            mnk_bf16x8 ha[3][TM], hb[3][TN];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    ha[pl][i] = mnk_as_bf16x8(*reinterpret_cast<const uint4*>(&Ah[kb][pl][wave * 32 + 16 * i + fi][ko]));
#pragma unroll
                for (int j = 0; j < TN; ++j) hb[pl][j] = mnk_as_bf16x8(*reinterpret_cast<const uint4*>(&Bh[kb][pl][16 * j + fi][ko]));
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ha[1][i], hb[1][j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ha[2][i], hb[0][j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ha[0][i], hb[2][j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ha[1][i], hb[0][j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ha[0][i], hb[1][j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ha[0][i], hb[0][j], acc[i][j], 0, 0, 0);
                }
        };
        const int n = s_end - s_begin;
        if (n > 0) load_h(s_begin, S0{});
        if (n > 1) load_h(s_begin + 1, S1{});
        for (int s = 0; s < n; s += 2) {
            store_h(S0{}, false);
            store_h(S1{}, s + 1 >= n);
            __syncthreads();
            if (s + 2 < n) load_h(s_begin + s + 2, S0{});
            if (s + 3 < n) load_h(s_begin + s + 3, S1{});
            mfma_pair();
            __syncthreads();                      // (also in front of the epilogue's reuse of the LDS image)
        }
    } else {
    auto mfma_step = [&](int buf) __attribute__((always_inline)) {
        float4 fa[TM], fb[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
            fa[i] = *reinterpret_cast<const float4*>(&As[buf][wave * 32 + 16 * i + fi][fk * 4]);
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const float4*>(&Bs[buf][16 * j + fi][fk * 4]);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i].x, fb[j].x, acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i].y, fb[j].y, acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i].z, fb[j].z, acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i].w, fb[j].w, acc[i][j], 0, 0, 0);
            }
    };
    // same pipeline as conv3x3_igemm_kernel: registers hold step s+1, loads of step s+2 precede the MFMAs of step s
    if (s_begin < s_end) {
        load_step(s_begin);
        store_step(0);
        if (s_begin + 1 < s_end) load_step(s_begin + 1);
    }
    __syncthreads();
    int s = s_begin;
    for (; s + 2 < s_end; ++s) {
        const int buf = (s - s_begin) & 1;
        store_step(buf ^ 1);
        load_step(s + 2);
        mfma_step(buf);
        __syncthreads();
    }
    if (s + 1 < s_end) {
        const int buf = (s - s_begin) & 1;
        store_step(buf ^ 1);
        mfma_step(buf);
        __syncthreads();
        ++s;
    }
    if (s < s_end) mfma_step((s - s_begin) & 1);
    __syncthreads();                          // the epilogue reuses As for the column sums
    }
    const bool split_out = a.splits > 1;
    const unsigned ldo = split_out ? (unsigned)a.ldw : (unsigned)a.ld_y;          // 32-bit element offsets (host check)
    float* const obase = split_out ? a.ws + (long)split * a.M * a.ldw : a.y;
    const int co_lim = split_out ? a.ldw : a.ld_y;
    const bool use_res = !split_out && a.residual;
    const bool full = m0 + BM <= a.M;
    const unsigned mrow0 = (unsigned)m0 + wave * 32 + 4 * fk, Mu = (unsigned)a.M;
    float s1[TN], s2[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) s1[j] = s2[j] = 0.f;
    const bool bnb = a.bnb.y != nullptr && a.stats && !split_out;     // (see BnBwdSrc)
    auto emit = [&](auto full_tag) __attribute__((always_inline)) {
        constexpr bool FULL = decltype(full_tag)::value;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int co = n0 + 16 * j + fi;
            const bool c_real = co < a.Cout, c_store = co < co_lim;
            const float bv = (!split_out && a.bias && c_real) ? a.bias[co] : 0.f;
            const bool breal = bnb && c_real;
            const float bm = breal ? a.bnb.mean[co] : 0.f, bis = breal ? a.bnb.invstd[co] : 0.f,
                        bsc = breal ? a.bnb.scale[co] : 0.f, bbe = breal ? a.bnb.beta[co] : 0.f;
            if (c_store) {
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const unsigned mb = mrow0 + 16 * i;
                    const unsigned off0 = mb * ldo + (unsigned)co, roff0 = mb * (unsigned)a.ld_res + (unsigned)co;
                    float rv[4] = {0.f, 0.f, 0.f, 0.f}, yv[4] = {0.f, 0.f, 0.f, 0.f};
                    if (use_res && c_real) {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (FULL || mb + r < Mu) rv[r] = a.residual[roff0 + r * (unsigned)a.ld_res];
                    }
                    if (breal) {
                        const unsigned yoff0 = mb * (unsigned)a.bnb.ld + (unsigned)co;
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (FULL || mb + r < Mu) yv[r] = a.bnb.y[yoff0 + r * (unsigned)a.bnb.ld];
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float v = acc[i][j][r];
                        if (!split_out) v = c_real ? (v + bv) + rv[r] : 0.f;
                        if (FULL || mb + r < Mu) {
                            obase[off0 + r * ldo] = v;
                            if (bnb) {
                                const float d = yv[r] - bm;
                                float g = v;
                                if (a.bnb.slope >= 0.f && !(fmaf(d, bsc, bbe) > 0.f)) g *= a.bnb.slope;
                                s1[j] += g;
                                s2[j] = fmaf(g, d * bis, s2[j]);
                            } else {
                                s1[j] += v;
                                s2[j] = fmaf(v, v, s2[j]);
                            }
                        }
                    }
                }
            }
        }
    };
    if (full)
        emit(TrueTag{});
    else
        emit(FalseTag{});
    if (a.stats && !split_out) {
        float* red = reinterpret_cast<float*>(smem_a);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            s1[j] += __shfl_xor(s1[j], 16);
            s1[j] += __shfl_xor(s1[j], 32);
            s2[j] += __shfl_xor(s2[j], 16);
            s2[j] += __shfl_xor(s2[j], 32);
            if (fk == 0) {
                red[(wave * 2 + 0) * BN + 16 * j + fi] = s1[j];
                red[(wave * 2 + 1) * BN + 16 * j + fi] = s2[j];
            }
        }
        __syncthreads();
        if (t < BN && n0 + t < a.ld_y) {
            float t1 = 0.f, t2 = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                t1 += red[(w * 2 + 0) * BN + t];
                t2 += red[(w * 2 + 1) * BN + t];
            }
            float* sp = a.stats + (long)bx * 2 * a.ld_y;
            sp[n0 + t] = t1;
            sp[a.ld_y + n0 + t] = t2;
        }
    }
}

// Split-K reductions: 64 outputs x 4 split-groups per block -- each thread sums every 4th partial (4x the loads in
// flight of a one-thread-per-output loop), the groups are combined through LDS in a fixed order (deterministic).
__global__ void __launch_bounds__(256) conv3x3_splitk_reduce_kernel(const float* __restrict__ ws, int splits, long M,
                                                                    int ldw, const float* __restrict__ bias,
                                                                    const float* __restrict__ residual, int ld_res,
                                                                    float* __restrict__ y, int ld_y, int Cout, int phases,
                                                                    int H, int W) {
    // phases = 4: partials are [split][phase][M][ldw] over the low-resolution pixels (n, i, j) of each sub-pixel phase;
    // the sum goes to pixel (2i + pa, 2j + pb) of the (N, 2H, 2W) output
    __shared__ float sm[4][64];
    const int o = threadIdx.x & 63, g = threadIdx.x >> 6;
    const long rows = M * phases, total = rows * ld_y;
    for (long base = (long)blockIdx.x * 64; base < total; base += (long)gridDim.x * 64) {
        const long i = base + o;
        long m = 0;
        int co = 0;
        float v = 0.f;
        if (i < total) {
            m = i / ld_y;
            co = (int)(i - m * ld_y);
            if (co < Cout)
                for (int s = g; s < splits; s += 4) v += ws[((long)s * rows + m) * ldw + co];
        }
        sm[g][o] = v;
        __syncthreads();
        if (g == 0 && i < total) {
            float r = 0.f;
            long orow = m;
            if (phases > 1) {
                const long ph = m / M, mm = m - ph * M;
                const long tt = mm / W, fr = tt / H;
                const long jj = mm - tt * W, ii = tt - fr * H;
                orow = ((fr * H + ii) * 2 + (ph >> 1)) * (2L * W) + 2 * jj + (ph & 1);
            }
            if (co < Cout) {
                r = (sm[0][o] + sm[1][o]) + (sm[2][o] + sm[3][o]);
                if (bias) r += bias[co];
                if (residual) r += residual[orow * ld_res + co];
            }
            y[orow * ld_y + co] = r;
        }
        __syncthreads();
    }
}

// The same reduction for a layer whose BatchNorm follows (sync_batchnorm/batchnorm.py:60-62 wants sum and sum of squares of
// what is written here): a block owns `rows_per_block` output rows x tx_n channel quads, every thread sums the splits of its
// float4 in the order of the kernel above (four interleaved groups, (g0 + g1) + (g2 + g3): the same bits), writes it and
// keeps the two column sums; stats[row_block][2][ld_y] are the per-block partials that the conv epilogue leaves for an
// unsplit launch (finished by bn_final_finalize / mnk_bn_stats_finish).  One launch instead of reduction + statistics pass.
struct RSMap {
    int tx, ty, col_tiles, row_blocks;
    long rows_per_block;
};
// rows per thread before a layer is cut into more row blocks (A/B with every split-K reduction on this kernel, visit 48:
// 1 / 2 / 4 -> 10.92 / 10.97 / 11.10 ms; the 64 x 4-group kernel: 10.95)
static int g_rs_rpt = tuning_knob("rs_rpt", &g_rs_rpt, 1);
static RSMap make_rsmap(long rows, int ld) {
    RSMap m;
    const int nv = ld / 4;
    int tx = 1;
    while (tx < nv && tx < 64) tx <<= 1;
    m.tx = tx;
    m.ty = 256 / tx;
    m.col_tiles = (nv + tx - 1) / tx;
    long want = 1024 / m.col_tiles;
    if (want < 1) want = 1;
    const long min_rows = (long)m.ty * (g_rs_rpt > 0 ? g_rs_rpt : 1);
    long rb = (rows + min_rows - 1) / min_rows;
    if (rb > want) rb = want;
    if (rb < 1) rb = 1;
    m.row_blocks = (int)rb;
    m.rows_per_block = (rows + rb - 1) / rb;
    return m;
}

template <bool STATS>
__global__ void __launch_bounds__(256) conv3x3_splitk_reduce_stats_kernel(const float* __restrict__ ws, int splits, long M,
                                                                          int ldw, const float* __restrict__ bias,
                                                                          const float* __restrict__ residual, int ld_res,
                                                                          float* __restrict__ y, int ld_y, int Cout,
                                                                          int phases, int H, int W, int tx_n, int ty_n,
                                                                          long rows_per_block, float* __restrict__ stats,
                                                                          BnBwdSrc bnb = BnBwdSrc{}) {
    __shared__ float4 red[2][256];
    const int tx = threadIdx.x % tx_n, ty = threadIdx.x / tx_n;
    const int q = blockIdx.x * tx_n + tx, nv = ld_y / 4, c = q * 4;
    const long rows = M * phases;
    const long r0 = (long)blockIdx.y * rows_per_block;
    long r1 = r0 + rows_per_block;
    if (r1 > rows) r1 = rows;
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
    if (q < nv) {
        const bool k0 = c < Cout, k1 = c + 1 < Cout, k2 = c + 2 < Cout, k3 = c + 3 < Cout;
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (bias) bv = make_float4(k0 ? bias[c] : 0.f, k1 ? bias[c + 1] : 0.f, k2 ? bias[c + 2] : 0.f, k3 ? bias[c + 3] : 0.f);
        const long sstride = rows * ldw;
        for (long m = r0 + ty; m < r1; m += ty_n) {
            const float* p = ws + m * ldw + c;
            float4 g[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) g[e] = make_float4(0.f, 0.f, 0.f, 0.f);
            int s = c < ldw ? 0 : splits;        // quads beyond the partial rows (ld_y > ldw): zero columns, nothing to read
            for (; s + 4 <= splits; s += 4) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float4 v = *reinterpret_cast<const float4*>(p + (long)(s + e) * sstride);
                    g[e].x += v.x;
                    g[e].y += v.y;
                    g[e].z += v.z;
                    g[e].w += v.w;
                }
            }
#pragma unroll
            for (int e = 0; e < 3; ++e)          // up to three left over: compile-time indices keep g[] in registers
                if (s + e < splits) {
                    const float4 v = *reinterpret_cast<const float4*>(p + (long)(s + e) * sstride);
                    g[e].x += v.x;
                    g[e].y += v.y;
                    g[e].z += v.z;
                    g[e].w += v.w;
                }
            long orow = m;
            if (phases > 1) {
                const long ph = m / M, mm = m - ph * M;
                const long tt = mm / W, fr = tt / H;
                const long jj = mm - tt * W, ii = tt - fr * H;
                orow = ((fr * H + ii) * 2 + (ph >> 1)) * (2L * W) + 2 * jj + (ph & 1);
            }
            float4 r;
            r.x = (g[0].x + g[1].x) + (g[2].x + g[3].x);
            r.y = (g[0].y + g[1].y) + (g[2].y + g[3].y);
            r.z = (g[0].z + g[1].z) + (g[2].z + g[3].z);
            r.w = (g[0].w + g[1].w) + (g[2].w + g[3].w);
            if (bias) {
                r.x += bv.x;
                r.y += bv.y;
                r.z += bv.z;
                r.w += bv.w;
            }
            if (residual) {
                const float* rp = residual + orow * ld_res + c;
                if (k0) r.x += rp[0];
                if (k1) r.y += rp[1];
                if (k2) r.z += rp[2];
                if (k3) r.w += rp[3];
            }
            r.x = k0 ? r.x : 0.f;
            r.y = k1 ? r.y : 0.f;
            r.z = k2 ? r.z : 0.f;
            r.w = k3 ? r.w : 0.f;
            *reinterpret_cast<float4*>(y + orow * ld_y + c) = r;
            if (!STATS) continue;
            if (bnb.y) {         // the sums are a BatchNorm layer's backward statistics (see BnBwdSrc)
                const float* yp = bnb.y + orow * bnb.ld + c;
                const float rr[4] = {r.x, r.y, r.z, r.w};
                float a1[4], a2[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    a1[e] = a2[e] = 0.f;
                    if (c + e < Cout) {
                        const float d = yp[e] - bnb.mean[c + e];
                        float g = rr[e];
                        if (bnb.slope >= 0.f && !(fmaf(d, bnb.scale[c + e], bnb.beta[c + e]) > 0.f)) g *= bnb.slope;
                        a1[e] = g;
                        a2[e] = g * (d * bnb.invstd[c + e]);
                    }
                }
                s1.x += a1[0], s1.y += a1[1], s1.z += a1[2], s1.w += a1[3];
                s2.x += a2[0], s2.y += a2[1], s2.z += a2[2], s2.w += a2[3];
                continue;
            }
            s1.x += r.x;
            s1.y += r.y;
            s1.z += r.z;
            s1.w += r.w;
            s2.x = fmaf(r.x, r.x, s2.x);
            s2.y = fmaf(r.y, r.y, s2.y);
            s2.z = fmaf(r.z, r.z, s2.z);
            s2.w = fmaf(r.w, r.w, s2.w);
        }
    }
    if (!STATS) return;          // STATS = false: the plain reduction as float4 rows (no LDS hop, every thread busy for any split count)
    red[0][threadIdx.x] = s1;
    red[1][threadIdx.x] = s2;
    __syncthreads();
    for (int s = ty_n >> 1; s > 0; s >>= 1) {
        if (ty < s) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const float4 a = red[k][threadIdx.x], b = red[k][threadIdx.x + s * tx_n];
                red[k][threadIdx.x] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
            }
        }
        __syncthreads();
    }
    if (ty == 0 && q < nv) {
        float* o = stats + (long)blockIdx.y * 2 * ld_y;
        *reinterpret_cast<float4*>(o + c) = red[0][tx];
        *reinterpret_cast<float4*>(o + ld_y + c) = red[1][tx];
    }
}

// ---- weight packing: Wp[row][chunk][tap][16] --------------------------------------------------------------------
__global__ void __launch_bounds__(256) pack_fwd_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cout,
                                                       int C0, int C1, int C0p, int C1p, int ntaps) {
    const int Cin = C0 + C1, chunks = (C0p + C1p) / 16;
    const long total = (long)Cout * chunks * ntaps * 16;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int k16 = (int)(i & 15);
        long t = i >> 4;
        const int tap = (int)(t % ntaps);
        t /= ntaps;
        const int chunk = (int)(t % chunks);
        const int co = (int)(t / chunks);
        const int k = chunk * 16 + k16;
        int ci = -1;
        if (k < C0p) {
            if (k < C0) ci = k;
        } else if (k - C0p < C1) {
            ci = C0 + k - C0p;
        }
        wp[i] = ci >= 0 ? w[((long)co * Cin + ci) * ntaps + tap] : 0.f;
    }
}

__global__ void __launch_bounds__(256) pack_dgrad_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cout,
                                                         int Cin_total, int c_start, int c_count, int chunks,
                                                         int ntaps) {
    const long total = (long)c_count * chunks * ntaps * 16;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int k16 = (int)(i & 15);
        long t = i >> 4;
        const int tap = (int)(t % ntaps);
        t /= ntaps;
        const int chunk = (int)(t % chunks);
        const int ci = (int)(t / chunks);
        const int co = chunk * 16 + k16;
        wp[i] = co < Cout ? w[((long)co * Cin_total + c_start + ci) * ntaps + (ntaps - 1 - tap)] : 0.f;
    }
}

// ---- packs of the sub-pixel forms of [nearest x2 up-sampling -> 3x3 / pad 1] (ConvArgs::phases) -----------------------
// S(a, u): the 3x3 kernel rows that fall on low-resolution row i + a - 1 + u for an output row 2i + a:
//   S(0,0) = {0}, S(0,1) = {1,2}, S(1,0) = {0,1}, S(1,1) = {2}                       (same sets for columns)
// D(t): the 3x3 kernel rows whose gradient reaches low-resolution row i from up-sampled-output row 2i - 1 + t:
//   D(0) = {2}, D(1) = {1,2}, D(2) = {0,1}, D(3) = {0}
__device__ __forceinline__ void phase_set(int a, int u, int& lo, int& hi) { up_phase_set(a, u, lo, hi); }
__device__ __forceinline__ void dgrad_set(int t, int& lo, int& hi) { up_dgrad_set(t, lo, hi); }

// forward: wp[phase][co][chunk][tap4 = 2u + v][16] = sum_{ky in S(a,u)} sum_{kx in S(b,v)} w[co][ci][ky][kx], phase = 2a + b
__global__ void __launch_bounds__(256) pack_up_fwd_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cout,
                                                          int C0, int C1, int C0p, int C1p) {
    const int Cin = C0 + C1, chunks = (C0p + C1p) / 16;
    const long per_phase = (long)Cout * chunks * 64, total = 4 * per_phase;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int phase = (int)(i / per_phase);
        long t = i - phase * per_phase;
        const int k16 = (int)(t & 15);
        t >>= 4;
        const int tap = (int)(t & 3);
        t >>= 2;
        const int chunk = (int)(t % chunks), co = (int)(t / chunks);
        const int k = chunk * 16 + k16;
        int ci = -1;
        if (k < C0p) {
            if (k < C0) ci = k;
        } else if (k - C0p < C1) {
            ci = C0 + k - C0p;
        }
        float v = 0.f;
        if (ci >= 0) {
            int y0, y1, x0, x1;
            phase_set(phase >> 1, tap >> 1, y0, y1);
            phase_set(phase & 1, tap & 1, x0, x1);
            const float* wr = w + ((long)co * Cin + ci) * 9;
            for (int ky = y0; ky <= y1; ++ky)
                for (int kx = x0; kx <= x1; ++kx) v += wr[ky * 3 + kx];
        }
        wp[i] = v;
    }
}

// data gradient w.r.t. the low-resolution input, source channels [c_start, c_start + c_count): a 4x4 / stride 2 / pad 1
// convolution over dy:  wp[ci][chunk(co)][tap16 = 4 ty + tx][16] = sum_{ky in D(ty)} sum_{kx in D(tx)} w[co][c_start+ci][ky][kx]
__global__ void __launch_bounds__(256) pack_up_dgrad_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cout,
                                                            int Cin_total, int c_start, int c_count, int chunks) {
    const long total = (long)c_count * chunks * 256;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int k16 = (int)(i & 15);
        long t = i >> 4;
        const int tap = (int)(t & 15);
        t >>= 4;
        const int chunk = (int)(t % chunks), ci = (int)(t / chunks);
        const int co = chunk * 16 + k16;
        float v = 0.f;
        if (co < Cout) {
            int y0, y1, x0, x1;
            dgrad_set(tap >> 2, y0, y1);
            dgrad_set(tap & 3, x0, x1);
            const float* wr = w + ((long)co * Cin_total + c_start + ci) * 9;
            for (int ky = y0; ky <= y1; ++ky)
                for (int kx = x0; kx <= x1; ++kx) v += wr[ky * 3 + kx];
        }
        wp[i] = v;
    }
}

template <int NT>
__global__ void __launch_bounds__(256) pack_all_kernel(const float* __restrict__ w, float* __restrict__ wf,
                                                       float* __restrict__ wd0, float* __restrict__ wd1, int Cout, int C0,
                                                       int C1, int C0p, int C1p, int ntaps_rt) {
    __shared__ float T[16 * (16 * 17 + 1)];
    pack_tile<NT>(T, w, wf, wd0, wd1, Cout, C0, C1, C0p, C1p, ntaps_rt, blockIdx.x, blockIdx.y);
}

// every 3x3 layer of a model in ONE launch: block b works on tile (b - tile_begin) of the layer whose
// [tile_begin, tile_begin + tiles) range contains it (binary search over the descriptor table in device memory)
__global__ void __launch_bounds__(256) pack_multi_kernel(const MnkPackDesc* __restrict__ descs, int n) {
    __shared__ float T[16 * (16 * 17 + 1)];
    int lo = 0, hi = n - 1;
    const int b = blockIdx.x;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (descs[mid].tile_begin <= b)
            lo = mid;
        else
            hi = mid - 1;
    }
    const MnkPackDesc d = descs[lo];
    const int local = b - d.tile_begin;
    const int C0p = round_up16(d.C0), C1p = d.C1 > 0 ? round_up16(d.C1) : 0, tiles_x = (C0p + C1p) / 16;
    const int cot = local / tiles_x, cc = local - cot * tiles_x;
    pack_tile<9>(T, d.w, d.wp_fwd, d.wp_d0, d.wp_d1, d.Cout, d.C0, d.C1, C0p, C1p, 9, cc, cot, d.flags & 1);
}

// ---- weight gradient ----------------------------------------------------------------------------------------
// GEMM  dW[co][n] = sum_p dY[p][co] * X[p + off(tap)][ci]  with n = ci*9 + tap, i.e. exactly the memory order of
// the (Cout, Cin, 1, 3, 3) parameter: a block's result tile is written straight into the gradient (or into a
// split-K partial of the same shape), coalesced along n.  K = pixels, split over blockIdx.z.
struct WgradArgs {
    const float* x;
    int ld_x, C, ups;
    const float* dy;
    int ld_dy, Cout;
    int N, H, W;         // output (dy) geometry
    int Hi, Wi, ntaps, kw, pad;
    long M;              // pixels
    long pix_per_split;  // multiple of 16
    int NT;              // ntaps * C
    float* out;          // unsplit: dw + c_start*9 (row stride ld_out); else partials [splits][Cout][NT]
    long ld_out;
    int splits;
};

template <int BM>
__global__ void __launch_bounds__(256) conv3x3_wgrad_kernel(WgradArgs a) {
    constexpr int BN = 128;
    constexpr int TMW = BM >= 64 ? 2 : 1;          // 32-row MFMA tiles per wave along co
    constexpr int WM = BM / (32 * TMW), WN = 4 / WM;   // waves along co / along n
    constexpr int TN = BN / WN / 32;               // 32-wide MFMA tiles per wave along n
    constexpr int LDA = BM + 4, LDB = BN + 4;
    constexpr int A4 = BM / 4;                     // float4 columns of the dy tile
    constexpr int APASS = 256 / A4;                // dy rows covered by one pass of the block
    constexpr int RA = (BK + APASS - 1) / APASS;   // dy rows per thread per step
    __shared__ __attribute__((aligned(16))) float As[2][BK][LDA];   // dy tile   [pixel][co]
    __shared__ __attribute__((aligned(16))) float Bs[2][BK][LDB];   // x-shifted [pixel][n]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int co0 = blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;
    const int split = blockIdx.z;
    const long p_begin = (long)split * a.pix_per_split;
    long p_end = p_begin + a.pix_per_split;
    if (p_end > a.M) p_end = a.M;
    const int Hs = a.ups ? a.Hi >> 1 : a.Hi, Ws = a.ups ? a.Wi >> 1 : a.Wi;

    // dy loader: float4 along co
    const int akr = t / A4, ac4 = t % A4;
    const int coa = co0 + ac4 * 4;
    // x loader: one fixed column n (-> ci, tap) per thread, 8 pixel rows (bk2, bk2+2, ...)
    const int bn = t & 127, bk2 = t >> 7;
    const int ncol = n0 + bn;
    const bool n_ok = ncol < a.NT;
    const int ci = n_ok ? ncol / a.ntaps : 0;
    const int tap = ncol - ci * a.ntaps;
    const int dyb = tap / a.kw - a.pad, dxb = tap % a.kw - a.pad;

    float4 ra[RA];
    float rb[8];
    auto load_step = [&](long p0) {
#pragma unroll
        for (int j = 0; j < RA; ++j) {
            const long p = p0 + akr + APASS * j;
            float4 va = make_float4(0.f, 0.f, 0.f, 0.f);
            if (akr + APASS * j < BK && p < p_end && coa < a.Cout) {
                va = *reinterpret_cast<const float4*>(a.dy + p * a.ld_dy + coa);
                const int rem = a.Cout - coa;
                if (rem < 4) {
                    if (rem < 2) va.y = 0.f;
                    if (rem < 3) va.z = 0.f;
                    va.w = 0.f;
                }
            }
            ra[j] = va;
        }
        // (n_img, h, w) of the first row, then incremental updates (rows advance by 2 pixels)
        long p = p0 + bk2;
        int w = (int)(p % a.W);
        long tt = p / a.W;
        int h = (int)(tt % a.H);
        long nimg = tt / a.H;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float v = 0.f;
            if (n_ok && p < p_end) {
                const int hh = h + dyb, ww = w + dxb;
                if (hh >= 0 && hh < a.Hi && ww >= 0 && ww < a.Wi) {
                    const int hs = a.ups ? hh >> 1 : hh, wsrc = a.ups ? ww >> 1 : ww;
                    v = a.x[((nimg * Hs + hs) * Ws + wsrc) * a.ld_x + ci];
                }
            }
            rb[j] = v;
            p += 2;
            w += 2;
            while (w >= a.W) {
                w -= a.W;
                if (++h >= a.H) {
                    h = 0;
                    ++nimg;
                }
            }
        }
    };
    auto store_step = [&](int buf) {
#pragma unroll
        for (int j = 0; j < RA; ++j)
            if (akr + APASS * j < BK) *reinterpret_cast<float4*>(&As[buf][akr + APASS * j][ac4 * 4]) = ra[j];
#pragma unroll
        for (int j = 0; j < 8; ++j) Bs[buf][bk2 + 2 * j][bn] = rb[j];
    };

    f32x16 acc[TMW][TN];
#pragma unroll
    for (int i = 0; i < TMW; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int fi = lane & 31, fk = lane >> 5;
    if (p_begin < p_end) {
        load_step(p_begin);
        store_step(0);
    }
    __syncthreads();
    int it = 0;
    for (long p0 = p_begin; p0 < p_end; p0 += BK, ++it) {
        const int buf = it & 1;
        if (p0 + BK < p_end) load_step(p0 + BK);
#pragma unroll
        for (int e = 0; e < BK / 2; ++e) {
            float fa[TMW], fb[TN];
#pragma unroll
            for (int i = 0; i < TMW; ++i) fa[i] = As[buf][2 * e + fk][wm * (32 * TMW) + 32 * i + fi];
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j] = Bs[buf][2 * e + fk][wn * (32 * TN) + 32 * j + fi];
#pragma unroll
            for (int i = 0; i < TMW; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
        if (p0 + BK < p_end) store_step(buf ^ 1);
        __syncthreads();
    }
    // rows = co, cols = n (contiguous in the parameter layout): 32 lanes write 128 consecutive bytes
    const bool partial = a.splits > 1;       // split partials are summed in a fixed order by the reduction (no fp32 atomics)
    float* outp = partial ? a.out + (long)split * a.Cout * a.NT : a.out;
    const long ldo = partial ? (long)a.NT : a.ld_out;
#pragma unroll
    for (int i = 0; i < TMW; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * (32 * TN) + 32 * j + fi;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + wm * (32 * TMW) + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * fk;
                if (co < a.Cout && n < a.NT) outp[(long)co * ldo + n] = acc[i][j][r];
            }
        }
}

// ---- weight gradient, LDS-halo form (layers with W >= 16) -------------------------------------------------------
// One block owns a 64 (co) x 64 (ci) x 3 (one tap ROW: ky fixed, kx = 0..2) slab of dW and walks a range of 64-pixel
// tiles (TR rows x TC columns of one frame, TC = min(W,64)).  Per tile it stages in LDS (a) the dy tile [64 px][64 co]
// and (b) the x rows shifted by ky-1 with one ZERO-bordered column on each side [(TR) x (TC+2) px][64 ci]; the three
// kx taps are then served from LDS: B(k = pixel, n = ci) for kx is the staged row shifted by kx-1 -- no per-tap global
// gathers, no masks.  Each wave owns one 32x32 (co, ci) quadrant = 3 accumulators; per pixel pair it issues 1 + 3
// ds_read_b32 and 3 MFMAs.  Splitting the taps over blockIdx.y triples the block count at the same split-K partial
// volume (different tap rows write different dW elements), which is what keeps the partial traffic small.
struct WgradHaloArgs {
    const float* x;
    int ld_x, C, ups;
    const float* dy;
    int ld_dy, Cout;
    int N, H, W;
    int TR, TC, tiles_w, tiles_per_img, gn;
    long total_tiles, tiles_per_split;
    int NT;
    float* out;
    long ld_out;
    int splits;
};

constexpr int WH_HP = 72;   // max staged x pixels: TR x (TC+2) = 1x66, 2x34, 4x18

__global__ void __launch_bounds__(256) conv3x3_wgrad_halo_kernel(WgradHaloArgs a) {
    __shared__ __attribute__((aligned(16))) float As[64][68];        // dy tile  [pixel][co]
    __shared__ __attribute__((aligned(16))) float Xs[WH_HP][64];     // x rows   [staged pixel][ci]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int cot = wave >> 1, cit = wave & 1;
    const int fi = lane & 31, fk = lane >> 5;
    const int co0 = blockIdx.x * 64;
    const int ci0 = (blockIdx.y % a.gn) * 64;
    const int ky = blockIdx.y / a.gn;            // tap row 0..2  (dy = ky - 1)
    const int split = blockIdx.z;
    const long tile_begin = (long)split * a.tiles_per_split;
    long tile_end = tile_begin + a.tiles_per_split;
    if (tile_end > a.total_tiles) tile_end = a.total_tiles;
    const int TC = a.TC, TR = a.TR, HW2 = TC + 2, HP = TR * HW2;
    const int Hs = a.ups ? a.H >> 1 : a.H, Ws = a.ups ? a.W >> 1 : a.W;

    f32x16 acc[3];
#pragma unroll
    for (int tp = 0; tp < 3; ++tp)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[tp][r] = 0.f;

    for (long tile = tile_begin; tile < tile_end; ++tile) {
        const int n = (int)(tile / a.tiles_per_img);
        const int ti = (int)(tile - (long)n * a.tiles_per_img);
        const int r0 = (ti / a.tiles_w) * TR, c0 = (ti % a.tiles_w) * TC;
        __syncthreads();   // previous tile's LDS reads are done
        // ---- dy tile: pixel q = (q / TC, q % TC), 16 float4 of co per pixel
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int q = (t >> 4) + 16 * j, c4 = t & 15;
            const int r = q / TC, c = q - r * TC;
            const int h = r0 + r, w = c0 + c, co = co0 + c4 * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (h < a.H && w < a.W && co < a.Cout) {
                v = *reinterpret_cast<const float4*>(a.dy + (((long)n * a.H + h) * a.W + w) * a.ld_dy + co);
                const int rem = a.Cout - co;
                if (rem < 4) {
                    if (rem < 2) v.y = 0.f;
                    if (rem < 3) v.z = 0.f;
                    v.w = 0.f;
                }
            }
            *reinterpret_cast<float4*>(&As[q][c4 * 4]) = v;
        }
        // ---- x rows (shifted by ky-1) with zero border columns / rows outside the frame
        for (int idx = t; idx < HP * 16; idx += 256) {
            const int hp = idx >> 4, c4 = idx & 15;
            const int hr = hp / HW2, hc = hp - hr * HW2;
            const int h = r0 + hr + ky - 1, w = c0 + hc - 1, ci = ci0 + c4 * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (h >= 0 && h < a.H && w >= 0 && w < a.W && ci < a.C) {
                const int hs = a.ups ? h >> 1 : h, wsrc = a.ups ? w >> 1 : w;
                const float* px = a.x + (((long)n * Hs + hs) * Ws + wsrc) * a.ld_x + ci;
                const int rem = a.C - ci;
                if (rem >= 4) {
                    v = *reinterpret_cast<const float4*>(px);
                } else {   // ld_x is only guaranteed >= C: read the tail element-wise
                    v.x = px[0];
                    if (rem > 1) v.y = px[1];
                    if (rem > 2) v.z = px[2];
                }
            }
            *reinterpret_cast<float4*>(&Xs[hp][c4 * 4]) = v;
        }
        __syncthreads();
        // ---- 32 pixel pairs x 3 taps
        int r = 0, c = fk;               // pixel q = 2e + fk  ->  (r, c); TC is even
        while (c >= TC) {
            c -= TC;
            ++r;
        }
#pragma unroll 4
        for (int e = 0; e < 32; ++e) {
            const float av = As[2 * e + fk][cot * 32 + fi];
            const float* xb = &Xs[r * HW2 + c][cit * 32 + fi];      // kx = 0 reads column c-1 (+1 border) = index c
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, xb[0], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, xb[64], acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, xb[128], acc[2], 0, 0, 0);
            c += 2;
            if (c >= TC) {
                c -= TC;
                ++r;
            }
        }
    }
    const bool partial = a.splits > 1;
    float* outp = partial ? a.out + (long)split * a.Cout * a.NT : a.out;
    const long ldo = partial ? (long)a.NT : a.ld_out;
    const int ci = ci0 + cit * 32 + fi;
#pragma unroll
    for (int tp = 0; tp < 3; ++tp)
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
            const int co = co0 + cot * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * fk;
            if (co < a.Cout && ci < a.C) outp[(long)co * ldo + ci * 9 + ky * 3 + tp] = acc[tp][rr];
        }
}

// ---- weight gradient, narrow layers (C, Cout <= 64 at high resolution: the 45-channel refinement stack) ------------
// 16x16x4 MFMA tiles on a 48 (co) x 48 (ci) channel tile (45 channels fill 94 % of it; a 64-wide tile only 49 %) with
// ALL nine taps in one block: per 8x8-pixel tile the dy tile [64 px][48 co] and the x tile with a one-pixel halo
// [10 x 10 px][48 ci] are staged in LDS once, and every tap is a shifted LDS read.
// Work split over the 4 wavefronts (81 = 9 taps x 3 co tiles x 3 ci tiles accumulators of 4 registers): wavefront w
// owns taps 2w and 2w+1 completely (18 tiles, sharing the three dy fragments) plus co tile w of tap 8 (3 tiles;
// wavefront 3 repeats one as a dummy) -- 21 MFMAs per 13 ds_read_b32 and K group, 96 % balanced.
// The next pixel tile is prefetched into registers while the MFMAs of the current one run.
// LDS rows are 48 floats (= 16 mod 32 banks): the four 16-lane groups of a fragment read hit disjoint banks.
struct WgradN16Args {
    const float* x;
    int ld_x, C, ups;
    const float* dy;
    int ld_dy, Cout;
    int H, W;
    int tiles_w, tiles_per_img;
    long total_tiles, tiles_per_split;
    int NT;
    float* out;
    long ld_out;
    int splits;
};

template <int NCT, int NCI>     // 16-wide co / ci tiles in use (1..3): narrower layers skip the padded tiles at compile time
__device__ __forceinline__ void wgrad_n16_body(const WgradN16Args& a, int bxi, int byi, int split) {
    constexpr int LD = 48, HW2 = 10, HP = 100;
    constexpr int XP = (HP * 12 + 255) / 256;                       // x loader passes (5)
    __shared__ __attribute__((aligned(16))) float As[64][LD];       // dy tile [pixel][co]
    __shared__ __attribute__((aligned(16))) float Xs[HP][LD];       // x tile with halo [staged pixel][ci]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int co0 = bxi * 48, ci0 = byi * 48;
    const long tile_begin = (long)split * a.tiles_per_split;
    long tile_end = tile_begin + a.tiles_per_split;
    if (tile_end > a.total_tiles) tile_end = a.total_tiles;
    const int Hs = a.ups ? a.H >> 1 : a.H, Ws = a.ups ? a.W >> 1 : a.W;

    f32x4 acc[2][NCT][NCI];       // taps 2*wave + {0, 1}: [tap][co tile][ci tile]
    f32x4 accx[NCI];              // tap 8, co tile xc: [ci tile]
#pragma unroll
    for (int tp = 0; tp < 2; ++tp)
#pragma unroll
        for (int i = 0; i < NCT; ++i)
#pragma unroll
            for (int j = 0; j < NCI; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[tp][i][j][r] = 0.f;
#pragma unroll
    for (int j = 0; j < NCI; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) accx[j][r] = 0.f;
    // staged-pixel offsets of this wavefront's taps (kernel row * 10 + kernel column) and its co tile of tap 8
    const int tap0 = 2 * wave, tap1 = 2 * wave + 1;
    const int toff0 = (tap0 / 3) * HW2 + tap0 % 3, toff1 = (tap1 / 3) * HW2 + tap1 % 3, toffx = 2 * HW2 + 2;
    const int xc = wave < NCT ? wave : 0;

    // loader assignment: dy 64 px x 12 float4 = 3 per thread; x 100 px x 12 float4 = 1200 -> 5 passes of 256
    float4 rd[3], rx[XP];
    const int tail_b0 = a.C - ci0;                // valid channels from the tile start

    auto zero_tail = [&](float4 v, int tl) __attribute__((always_inline)) {
        v.x = tl < 1 ? 0.f : v.x;
        v.y = tl < 2 ? 0.f : v.y;
        v.z = tl < 3 ? 0.f : v.z;
        v.w = tl < 4 ? 0.f : v.w;
        return v;
    };
    auto load_tile = [&](long tile) __attribute__((always_inline)) {
        const unsigned ut = (unsigned)tile;                        // total_tiles < 2^31 (host check)
        const int n = (int)(ut / (unsigned)a.tiles_per_img);
        const unsigned ti = ut - (unsigned)n * (unsigned)a.tiles_per_img, tr = ti / (unsigned)a.tiles_w;
        const int r0 = (int)tr * 8, c0 = (int)(ti - tr * (unsigned)a.tiles_w) * 8;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int idx = t + 256 * j;
            const int q = idx / 12, c4 = idx % 12;       // 256 * 3 = 64 px * 12 float4 exactly
            const int h = r0 + (q >> 3), w = c0 + (q & 7);
            const int tl = a.Cout - (co0 + c4 * 4);
            const int ce = tl > 0 ? co0 + c4 * 4 : 0;
            const float4 v = *reinterpret_cast<const float4*>(a.dy + (((long)n * a.H + h) * a.W + w) * a.ld_dy + ce);
            rd[j] = zero_tail(v, tl);
        }
#pragma unroll
        for (int j = 0; j < XP; ++j) {
            const int idx = t + 256 * j;
            const int hp = idx / 12 < HP ? idx / 12 : HP - 1, c4 = idx % 12;
            const int hr = hp / HW2, hc = hp - hr * HW2;
            int h = r0 + hr - 1, w = c0 + hc - 1;
            const bool ok = h >= 0 && h < a.H && w >= 0 && w < a.W;
            h = h < 0 ? 0 : (h >= a.H ? a.H - 1 : h);
            w = w < 0 ? 0 : (w >= a.W ? a.W - 1 : w);
            const int tl = ok ? tail_b0 - c4 * 4 : 0;
            const int ce = tl > 0 ? ci0 + c4 * 4 : 0;
            const float4 v = *reinterpret_cast<const float4*>(
                a.x + (((long)n * Hs + (h >> a.ups)) * Ws + (w >> a.ups)) * a.ld_x + ce);
            rx[j] = zero_tail(v, tl);
        }
    };
    auto store_tile = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {             // 256 * 3 = 768 = 64 px * 12 exactly; the float4 column is idx % 12
            const int idx = t + 256 * j;
            *reinterpret_cast<float4*>(&As[idx / 12][(idx % 12) * 4]) = rd[j];
        }
#pragma unroll
        for (int j = 0; j < XP; ++j) {
            const int idx = t + 256 * j;
            if (idx < HP * 12) *reinterpret_cast<float4*>(&Xs[idx / 12][(idx % 12) * 4]) = rx[j];
        }
    };

    const int fi = lane & 15, fk = lane >> 4;     // row/col inside a 16-tile, pixel 0..3 of the K group
    if (tile_begin < tile_end) load_tile(tile_begin);
    for (long tile = tile_begin; tile < tile_end; ++tile) {
        __syncthreads();                          // the previous tile's LDS reads are done
        store_tile();
        __syncthreads();
        if (tile + 1 < tile_end) load_tile(tile + 1);     // in flight during the MFMAs below
#pragma unroll 2
        for (int g = 0; g < 16; ++g) {            // K group = pixels 4g..4g+3 = row g>>1, columns 4*(g&1)..+3
            const int q = 4 * g + fk;
            const int xr = (g >> 1) * HW2 + (g & 1) * 4 + fk;        // staged pixel of tap (0, 0)
            float fa[NCT], fb0[NCI], fb1[NCI], fbx[NCI];
#pragma unroll
            for (int i = 0; i < NCT; ++i) fa[i] = As[q][16 * i + fi];
            const float fax = As[q][16 * xc + fi];
#pragma unroll
            for (int j = 0; j < NCI; ++j) {
                fb0[j] = Xs[xr + toff0][16 * j + fi];
                fb1[j] = Xs[xr + toff1][16 * j + fi];
                fbx[j] = Xs[xr + toffx][16 * j + fi];
            }
#pragma unroll
            for (int i = 0; i < NCT; ++i)
#pragma unroll
                for (int j = 0; j < NCI; ++j) {
                    acc[0][i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i], fb0[j], acc[0][i][j], 0, 0, 0);
                    acc[1][i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i], fb1[j], acc[1][i][j], 0, 0, 0);
                }
#pragma unroll
            for (int j = 0; j < NCI; ++j) accx[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fax, fbx[j], accx[j], 0, 0, 0);
        }
    }
    // D: col = lane & 15 (-> ci), row = 4 * (lane >> 4) + r (-> co).  A single split writes the parameter layout
    // (n = ci * 9 + tap) directly; split partials are tap-major [split][tap][co][ci] -- 64-byte runs along ci instead
    // of 36-byte-strided words (measured: 82 MB of HBM writes per launch for 37 MB of partials) -- and are summed and
    // transposed by conv3x3_wgrad_tap_reduce_kernel.
    const bool partial = a.splits > 1;
    float* outp = partial ? a.out + (long)split * a.Cout * a.NT : a.out;
    const long ldo = partial ? (long)a.C : a.ld_out;                 // row stride (co)
    const long tstride = partial ? (long)a.Cout * a.C : 1;             // tap stride
    const int cstride = partial ? 1 : 9;                               // ci stride
    if (wave < 4) {
#pragma unroll
        for (int tp = 0; tp < 2; ++tp) {
            const int tap = 2 * wave + tp;
            if (tap < 8)
#pragma unroll
                for (int i = 0; i < NCT; ++i)
#pragma unroll
                    for (int j = 0; j < NCI; ++j) {
                        const int ci = ci0 + 16 * j + fi;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int co = co0 + 16 * i + 4 * fk + r;
                            if (co < a.Cout && ci < a.C)
                                outp[tap * tstride + (long)co * ldo + ci * cstride] = acc[tp][i][j][r];
                        }
                    }
        }
        if (wave < NCT)
#pragma unroll
            for (int j = 0; j < NCI; ++j) {
                const int ci = ci0 + 16 * j + fi;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = co0 + 16 * xc + 4 * fk + r;
                    if (co < a.Cout && ci < a.C) outp[8 * tstride + (long)co * ldo + ci * cstride] = accx[j][r];
                }
            }
    }
}

template <int NCT, int NCI>
__global__ void __launch_bounds__(256, 2) conv3x3_wgrad_n16_kernel(WgradN16Args a) {
    wgrad_n16_body<NCT, NCI>(a, blockIdx.x, blockIdx.y, blockIdx.z);
}

// index of the last record whose block_begin (an int column with `stride_ints` between rows) is <= b
__device__ __forceinline__ int find_desc(const int* begins_stride_bytes_base, int stride_ints, int n, int b, int* sh) {
    // sh: one LDS int; every thread of the block returns the index of the last descriptor with block_begin <= b
    if (threadIdx.x == 0) *sh = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 256) {
        const int i = base + (int)threadIdx.x;
        if (i < n && begins_stride_bytes_base[(long)i * stride_ints] <= b) atomicAdd(sh, 1);    // LDS atomic, <= n per block
    }
    __syncthreads();
    return *sh - 1;
}


// ---- weight gradient, tap-major form (the default for C >= 16) ---------------------------------------------------
// The same pipeline as the forward kernel with K = pixels: one block owns a BM (co) x BN (ci) tile of ONE tap and a
// range of pixels.  Per 16-pixel K step both operands are plain coalesced float4 reads along channels -- dy rows
// [pixel][co] and x rows of the tap-shifted pixels [pixel + off(tap)][ci] (clamped into the image, zeroed on the way
// to LDS when the tap falls outside) -- so there are no scalar gathers and no branches in the steady-state loop;
// registers hold step s+1, loads of step s+2 are issued before the MFMAs of step s.  The tile is written to a
// tap-major partial [split][tap][co][ci] (coalesced along ci); conv3x3_wgrad_tap_reduce_kernel sums the splits and
// transposes (tap, ci) -> the parameter order ci*ntaps + tap through LDS.
struct WgradTapArgs {
    const float* x;
    int ld_x, C, ups;
    const float* dy;
    int ld_dy, Cout;
    int H, W;            // dy geometry
    int Hi, Wi, ntaps, kw, pad;
    long M, pix_per_split;
    int gn;              // ci tiles per tap
    float* part;         // [splits][ntaps][Cout][C]
    unsigned mulW, shW, mulH, shH;   // division by W / H of a pixel index < 2^31 (mul == 0: shift only)
    int xcd;             // re-chunk the launch order per XCD (xcd_tile)
    int clean;           // pad channels of x and dy hold zeros: the buffer-load fast path may be used
    int sw, sh, sn;      // fast path: one 16-pixel K step = sw columns + sh rows + sn frames
    // small maps (H * W <= wtap_compact, MODE 0 / 3): K runs over the pixels whose tap lies INSIDE the source only -- on a 2 x 2 map
    // a corner tap sees one pixel of four, an edge tap two (56 % of the (pixel, tap) pairs of a 3x3 pad-1 convolution are zeros
    // there, 31 % on 4 x 4, 16 % on 8 x 8; three quarters of the sub-pixel form's pseudo taps on a 1 x 1 source).  The range of a
    // tap is cut into `nsplits` equal pieces.
    int compact, nsplits;
    // MODE 3 -- sub-pixel form of an up-sampled 3x3 convolution: H, W, M are the LOW resolution; the 16 "taps" are
    // t = 4 * (2a + b) + (2u + v): dWeff[t] = sum_{n,i,j} dy[n, 2i+a, 2j+b] (x) x[n, i+a-1+u, j+b-1+v]  (4/9 of the multiply-adds of
    // the nine-tap form; the reduction folds the 16 pseudo taps into the nine kernel taps)
};

// MODE 0: generic loader (any K x K, masks, clamps, magic-number division per row and step).  MODE 1 / 2: 3x3 pad 1 with
// clean pad channels, W >= 16 (plain / x2 up-sampled source): raw buffer loads whose out-of-range lanes read zero --
// dy rows beyond the split's pixel range fall off the end of the buffer, taps outside the image and float4s beyond
// the channel count get bit 30 added to their offset -- and the (h, w) of a row is advanced incrementally (16 pixels
// per step wrap at most once), so a row costs ~10 vector instructions per step instead of ~35.
template <int BM, int BN, int WM, int WN, int MODE>
__device__ __forceinline__ void wgrad_tap_body(const WgradTapArgs& a, const int bx, const int by, const int split) {
    static_assert(WM * WN == 4, "4 waves per block");
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int LDA = BM + 4, LDB = BN + 4;
    constexpr int A4 = BM / 4, B4 = BN / 4;              // float4 columns of a tile row
    constexpr int APASS = 256 / A4, BPASS = 256 / B4;     // pixel rows covered by one pass of the block
    constexpr int RA = (BK + APASS - 1) / APASS, RB = (BK + BPASS - 1) / BPASS;
    static_assert(RA <= 2 && RB <= 2, "at most two rows per thread and operand");
    __shared__ __attribute__((aligned(16))) float As[2][BK][LDA];   // dy tile   [pixel][co]
    __shared__ __attribute__((aligned(16))) float Bs[2][BK][LDB];   // x-shifted [pixel][ci]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int co0 = bx * BM;
    const int tap = by / a.gn;
    const int ci0 = (by - tap * a.gn) * BN;
    long p_begin = (long)split * a.pix_per_split;
    long p_end = p_begin + a.pix_per_split;
    if (p_end > a.M) p_end = a.M;
    const int Hs = a.ups ? a.Hi >> 1 : a.Hi, Ws = a.ups ? a.Wi >> 1 : a.Wi;
    constexpr bool SUBPIX = MODE == 3, FAST = MODE == 1 || MODE == 2;
    const int ph_a = (tap >> 3) & 1, ph_b = (tap >> 2) & 1;                      // SUBPIX: output phase of this pseudo tap
    const int dyt = SUBPIX ? ph_a - 1 + ((tap >> 1) & 1) : tap / a.kw - a.pad;   // SUBPIX: low-resolution row / column offset
    const int dxt = SUBPIX ? ph_b - 1 + (tap & 1) : tap % a.kw - a.pad;
    const int hmax = a.Hi - 1, wmax = a.Wi - 1;
    // compact K (small maps): the rectangle [ch0, ch0 + chh) x [cw0, cw0 + cww) of output pixels whose tap lies inside the source;
    // K index k -> (frame, row, column) of that rectangle, Kt = frames * chh * cww entries
    const bool compact = !FAST && a.compact;
    int ch0 = 0, cw0 = 0, chh = a.H, cww = a.W;
    float inv_chh = 0.f, inv_cww = 0.f;
    long Klast = a.M - 1;
    if (compact) {
        ch0 = dyt < 0 ? -dyt : 0;
        cw0 = dxt < 0 ? -dxt : 0;
        int h1 = a.Hi - dyt, w1 = a.Wi - dxt;
        h1 = h1 > a.H ? a.H : h1;
        w1 = w1 > a.W ? a.W : w1;
        chh = h1 > ch0 ? h1 - ch0 : 0;
        cww = w1 > cw0 ? w1 - cw0 : 0;
        const long frames = a.M / ((long)a.H * a.W);
        const long Kt = frames * chh * cww;
        const long per = ((Kt + a.nsplits - 1) / a.nsplits + BK - 1) / BK * BK;
        p_begin = (long)split * per;
        p_end = p_begin + per;
        if (p_end > Kt) p_end = Kt;
        if (p_begin > p_end) p_begin = p_end;
        Klast = Kt > 0 ? Kt - 1 : 0;
        inv_chh = chh > 0 ? 1.f / (float)chh : 0.f;
        inv_cww = cww > 0 ? 1.f / (float)cww : 0.f;
    }
    const unsigned plast = (unsigned)Klast, pend = (unsigned)p_end;
    // k -> (n, i, j) inside the rectangle; k < 2^20: (k + 0.5) / d is at least 0.5 / d away from an integer, far beyond the rounding
    auto unpack = [&](unsigned k, unsigned& n, unsigned& i, unsigned& j) __attribute__((always_inline)) {
        const unsigned q = (unsigned)(((float)k + 0.5f) * inv_cww);
        j = k - q * (unsigned)cww + (unsigned)cw0;
        n = (unsigned)(((float)q + 0.5f) * inv_chh);
        i = q - n * (unsigned)chh + (unsigned)ch0;
    };

    const int ar = t / A4, ac4 = t % A4, br = t / B4, bc4 = t % B4;
    const int coa = co0 + ac4 * 4, cib = ci0 + bc4 * 4;
    const int tail_a = a.Cout - coa, tail_b = a.C - cib;
    const unsigned coa_e = tail_a > 0 ? coa : 0, cib_e = tail_b > 0 ? cib : 0;

    float4 ra0, ra1, rb0, rb1;
    int ta0 = 0, ta1 = 0, tb0 = 0, tb1 = 0;       // valid channels of the float4s (<= 0: zero the whole vector)

    auto load_a = [&](unsigned p, float4& v, int& tl) __attribute__((always_inline)) {
        tl = p < pend ? tail_a : 0;
        const unsigned pe = p < plast ? p : plast;
        unsigned long row = pe;
        if (compact) {
            unsigned n, i, j;
            unpack(pe, n, i, j);
            row = SUBPIX ? ((unsigned long)(n * (unsigned)a.H + i) * 2u + (unsigned)ph_a) * (2u * (unsigned)a.W) + 2u * j + (unsigned)ph_b
                         : (unsigned long)(n * (unsigned)a.H + i) * (unsigned)a.W + j;
        } else if constexpr (SUBPIX) {   // low-resolution pixel (n, i, j) -> pixel (2i + a, 2j + b) of the up-sampled dy
            const unsigned q = fast_div(pe, a.mulW, a.shW), n = fast_div(q, a.mulH, a.shH);
            const unsigned j = pe - q * (unsigned)a.W, i = q - n * (unsigned)a.H;
            row = ((unsigned long)(n * (unsigned)a.H + i) * 2u + (unsigned)ph_a) * (2u * (unsigned)a.W) + 2u * j + (unsigned)ph_b;
        }
        v = *reinterpret_cast<const float4*>(a.dy + row * (unsigned)a.ld_dy + coa_e);
    };
    auto load_b = [&](unsigned p, float4& v, int& tl) __attribute__((always_inline)) {
        const unsigned pe = p < plast ? p : plast;
        if (compact) {                   // every entry of the range lies inside the source: no clamps, no masks but the range's end
            unsigned n, i, j;
            unpack(pe, n, i, j);
            tl = p < pend ? tail_b : 0;
            const unsigned pix = (n * (unsigned)Hs + (unsigned)((int)i + dyt)) * (unsigned)Ws + (unsigned)((int)j + dxt);
            v = *reinterpret_cast<const float4*>(a.x + (unsigned long)pix * (unsigned)a.ld_x + cib_e);
            return;
        }
        const unsigned q = fast_div(pe, a.mulW, a.shW);
        const int w = (int)(pe - q * (unsigned)a.W);
        const unsigned n = fast_div(q, a.mulH, a.shH);
        const int h = (int)(q - n * (unsigned)a.H);
        int hh = h + dyt, ww = w + dxt;
        const bool ok = hh >= 0 && hh <= hmax && ww >= 0 && ww <= wmax;
        hh = hh < 0 ? 0 : (hh > hmax ? hmax : hh);
        ww = ww < 0 ? 0 : (ww > wmax ? wmax : ww);
        tl = ok ? tail_b : 0;
        const unsigned pix = (n * (unsigned)Hs + (unsigned)(hh >> a.ups)) * (unsigned)Ws + (unsigned)(ww >> a.ups);
        v = *reinterpret_cast<const float4*>(a.x + (unsigned long)pix * (unsigned)a.ld_x + cib_e);
    };
    // ---- fast loader state (MODE 1 / 2) ---------------------------------------------------------------------
    constexpr bool FUPS = MODE == 2;
    __amdgpu_buffer_rsrc_t rsa, rsb;
    unsigned aoff0 = 0, aoff1 = 0, boff0 = 0, boff1 = 0;     // running byte offsets (non-ups B: linear in the pixel)
    int bw0 = 0, bh0 = 0, bn0 = 0, bw1 = 0, bh1 = 0, bn1 = 0;  // (w, h, frame relative to the first) of the B rows
    const int ldy4 = a.ld_dy * 4, ldx4 = a.ld_x * 4;
    const int hbad = dyt < 0 ? 0 : (dyt > 0 ? a.H - 1 : -1), wbad = dxt < 0 ? 0 : (dxt > 0 ? a.W - 1 : -1);
    if constexpr (FAST) {
        rsa = uniform_rsrc(a.dy + p_begin * a.ld_dy, (unsigned)((p_end - p_begin) * ldy4));
        const unsigned tail_flag_a = tail_a > 0 ? 0u : 0x40000000u, tail_flag_b = tail_b > 0 ? 0u : 0x40000000u;
        aoff0 = (unsigned)(ar * ldy4 + (int)coa_e * 4) + tail_flag_a;
        aoff1 = aoff0 + (unsigned)(APASS * ldy4);
        const long frame_px = (long)a.H * a.W;
        const long nb = p_begin / frame_px;                       // first frame of the split
        long pb0 = FUPS ? nb * Hs * Ws : p_begin + dyt * a.W + dxt;   // pixel the B resource starts at
        if (pb0 < 0) pb0 = 0;
        const long total_src = FUPS ? (a.M / frame_px) * Hs * Ws : a.M;
        long nrec = (total_src - pb0) * ldx4;
        if (nrec > 0x40000000L) nrec = 0x40000000L;
        // a split that starts in the last image row has pb0 > total_src for the taps of the row below (every row of it is
        // "bad"): an empty buffer -- a negative length would wrap to ~4 GB of "valid" range and the bit-30 offsets of the
        // bad rows would be dereferenced (seen as a memory fault at 256x256: three image rows per split)
        if (nrec < 0) nrec = 0;
        rsb = uniform_rsrc(a.x + pb0 * a.ld_x, (unsigned)nrec);
        auto init_b = [&](long p, int& w, int& h, int& n, unsigned& off) __attribute__((always_inline)) {
            const unsigned up = (unsigned)p, q = fast_div(up, a.mulW, a.shW), fr = fast_div(q, a.mulH, a.shH);
            w = (int)(up - q * (unsigned)a.W);
            h = (int)(q - fr * (unsigned)a.H);
            n = (int)((long)fr - nb);
            off = (FUPS ? (unsigned)((int)cib_e * 4) : (unsigned)((int)(p + dyt * a.W + dxt - pb0) * ldx4 + (int)cib_e * 4)) +
                  tail_flag_b;
        };
        init_b(p_begin + br, bw0, bh0, bn0, boff0);
        init_b(p_begin + br + BPASS, bw1, bh1, bn1, boff1);
    }
    auto fast_b = [&](int& w, int& h, int& n, unsigned& off, float4& v) __attribute__((always_inline)) {
        const bool bad = (h == hbad) | (w == wbad);
        unsigned o = off;
        if constexpr (FUPS)
            o += (unsigned)(((n * Hs + ((h + dyt) >> 1)) * Ws + ((w + dxt) >> 1)) * ldx4);
        o |= bad ? 0x40000000u : 0u;
        v = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsb, o, 0, 0));
        if constexpr (!FUPS) off += (unsigned)(BK * ldx4);
        // 16 pixels ahead as (sw, sh, sn) columns / rows / frames (host: W >= 16 -> (16,0,0); W | 16 -> rows or whole
        // frames), each with at most one wrap
        w += a.sw;
        const bool ww = w >= a.W;
        w -= ww ? a.W : 0;
        h += a.sh + (ww ? 1 : 0);
        const bool hw = h >= a.H;
        h -= hw ? a.H : 0;
        if constexpr (FUPS) n += a.sn + (hw ? 1 : 0);
    };
    auto load_step = [&](long p0) __attribute__((always_inline)) {
        if constexpr (FAST) {
            ra0 = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsa, aoff0, 0, 0));
            aoff0 += (unsigned)(BK * ldy4);
            if constexpr (RA > 1) {
                ra1 = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsa, aoff1, 0, 0));
                aoff1 += (unsigned)(BK * ldy4);
            }
            fast_b(bw0, bh0, bn0, boff0, rb0);
            if constexpr (RB > 1) fast_b(bw1, bh1, bn1, boff1, rb1);
        } else {
            const unsigned pa = (unsigned)p0 + ar, pb = (unsigned)p0 + br;
            load_a(pa, ra0, ta0);
            if constexpr (RA > 1) load_a(pa + APASS, ra1, ta1);
            load_b(pb, rb0, tb0);
            if constexpr (RB > 1) load_b(pb + BPASS, rb1, tb1);
        }
    };
    auto masked = [&](float4 v, int tl) __attribute__((always_inline)) {
        if constexpr (FAST) return v;
        v.x = tl < 1 ? 0.f : v.x;
        v.y = tl < 2 ? 0.f : v.y;
        v.z = tl < 3 ? 0.f : v.z;
        v.w = tl < 4 ? 0.f : v.w;
        return v;
    };
    auto store_step = [&](int buf) __attribute__((always_inline)) {
        if (APASS >= BK ? ar < BK : true) *reinterpret_cast<float4*>(&As[buf][ar][ac4 * 4]) = masked(ra0, ta0);
        if constexpr (RA > 1) *reinterpret_cast<float4*>(&As[buf][ar + APASS][ac4 * 4]) = masked(ra1, ta1);
        if (BPASS >= BK ? br < BK : true) *reinterpret_cast<float4*>(&Bs[buf][br][bc4 * 4]) = masked(rb0, tb0);
        if constexpr (RB > 1) *reinterpret_cast<float4*>(&Bs[buf][br + BPASS][bc4 * 4]) = masked(rb1, tb1);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int fi = lane & 31, fk = lane >> 5;
    auto mfma_step = [&](int buf) __attribute__((always_inline)) {
        // fragments are read two pixel pairs ahead of the MFMAs that use them (sched_group_barrier pins the order the
        // source states: without it the scheduler reads each pair right before its MFMAs and waits for the LDS every time)
        float fa[BK / 2][TM], fb[BK / 2][TN];
        auto rd = [&](int e) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[e][i] = As[buf][2 * e + fk][wm * (32 * TM) + 32 * i + fi];
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[e][j] = Bs[buf][2 * e + fk][wn * (32 * TN) + 32 * j + fi];
        };
        rd(0);
        rd(1);
#pragma unroll
        for (int e = 0; e < BK / 2; ++e) {
            if (e + 2 < BK / 2) rd(e + 2);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[e][i], fb[e][j], acc[i][j], 0, 0, 0);
            MNK_SCHED_GROUP(0x100, TM + TN);      // DS reads of pair e + 2
            MNK_SCHED_GROUP(0x008, TM * TN);      // MFMAs of pair e
        }
    };

    if (p_begin < p_end) {
        load_step(p_begin);
        store_step(0);
        if (p_begin + BK < p_end) load_step(p_begin + BK);
    }
    __syncthreads();
    long p0 = p_begin;
    int it = 0;
    for (; p0 + 2 * BK < p_end; p0 += BK, ++it) {
        const int buf = it & 1;
        store_step(buf ^ 1);
        load_step(p0 + 2 * BK);
        mfma_step(buf);
        __syncthreads();
    }
    if (p0 + BK < p_end) {
        const int buf = it & 1;
        store_step(buf ^ 1);
        mfma_step(buf);
        __syncthreads();
        p0 += BK;
        ++it;
    }
    if (p0 < p_end) mfma_step(it & 1);

    // rows = co, cols = ci: 32 lanes write 128 consecutive bytes of the tap-major partial (32-bit offsets inside the
    // tap plane, no per-row guards when the whole co tile exists)
    float* outp = a.part + ((long)split * a.ntaps + tap) * a.Cout * a.C;
    const unsigned Cu = (unsigned)a.C, corow0 = (unsigned)co0 + wm * (32 * TM) + 4 * fk;
    auto emit = [&](auto full_tag) __attribute__((always_inline)) {
        constexpr bool FULL = decltype(full_tag)::value;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const unsigned ci = (unsigned)ci0 + wn * (32 * TN) + 32 * j + fi;
                const unsigned cb = corow0 + 32 * i, off0 = cb * Cu + ci;
                if (ci < Cu) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const unsigned ro = (r & 3) + 8 * (r >> 2);
                        if (FULL || cb + ro < (unsigned)a.Cout) outp[off0 + ro * Cu] = acc[i][j][r];
                    }
                }
            }
    };
    if (co0 + BM <= a.Cout)
        emit(TrueTag{});
    else
        emit(FalseTag{});
}

// ---- the tap-major weight gradient on the bf16 matrix cores (round 6; GM = 1 of the forward kernel, mnk_common.h) -------------
// K = pixels here, and both operands are K-STRIDED in memory (dy [pixel][co], x [pixel][ci]) while a lane of
// v_mfma_f32_32x32x16_bf16 wants eight consecutive k of ONE channel.  The loader therefore transposes on its way to LDS: a
// loader thread owns four consecutive pixels x four consecutive channels (four float4 loads, lanes along the channels: coalesced
// rows), regroups them per channel, splits each float4-of-pixels into three bf16 planes and writes four halves (8 bytes) per
// plane and channel.  LDS image per plane: 16-byte chunks [k group of 8][channel], chunk(kg, m) = kg * BMP + m + (m >> 4) (one pad
// chunk per 16 channels: conflict-free b128 fragment reads, 2-way on the writes).  Threads [0, BM) load dy, [BM, BM + BN) load x.
// Generic addressing only (magic-number divisions, clamps, masks, the compact K of small maps): the split dominates the loader.
template <int BM, int BN, int WM, int WN, bool SUBPIX>
__device__ __forceinline__ void wgrad_tap_body_h(const WgradTapArgs& a, const int bx, const int by, const int split) {
    static_assert(WM * WN == 4, "4 waves per block");
    static_assert(BM + BN <= 256, "one loader thread per (pixel group, channel quad) of both operands");
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int BMP = BM + BM / 16, BNP = BN + BN / 16;
    __shared__ __attribute__((aligned(16))) uint4 Ah[2][3][2 * BMP];
    __shared__ __attribute__((aligned(16))) uint4 Bh[2][3][2 * BNP];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int co0 = bx * BM;
    const int tap = by / a.gn;
    const int ci0 = (by - tap * a.gn) * BN;
    long p_begin = (long)split * a.pix_per_split;
    long p_end = p_begin + a.pix_per_split;
    if (p_end > a.M) p_end = a.M;
    const int Hs = a.ups ? a.Hi >> 1 : a.Hi, Ws = a.ups ? a.Wi >> 1 : a.Wi;
    const int ph_a = (tap >> 3) & 1, ph_b = (tap >> 2) & 1;
    const int dyt = SUBPIX ? ph_a - 1 + ((tap >> 1) & 1) : tap / a.kw - a.pad;
    const int dxt = SUBPIX ? ph_b - 1 + (tap & 1) : tap % a.kw - a.pad;
    const int hmax = a.Hi - 1, wmax = a.Wi - 1;
    const bool compact = a.compact;
    int ch0 = 0, cw0 = 0, chh = a.H, cww = a.W;
    float inv_chh = 0.f, inv_cww = 0.f;
    long Klast = a.M - 1;
    if (compact) {          // (see wgrad_tap_body)
        ch0 = dyt < 0 ? -dyt : 0;
        cw0 = dxt < 0 ? -dxt : 0;
        int h1 = a.Hi - dyt, w1 = a.Wi - dxt;
        h1 = h1 > a.H ? a.H : h1;
        w1 = w1 > a.W ? a.W : w1;
        chh = h1 > ch0 ? h1 - ch0 : 0;
        cww = w1 > cw0 ? w1 - cw0 : 0;
        const long frames = a.M / ((long)a.H * a.W);
        const long Kt = frames * chh * cww;
        const long per = ((Kt + a.nsplits - 1) / a.nsplits + BK - 1) / BK * BK;
        p_begin = (long)split * per;
        p_end = p_begin + per;
        if (p_end > Kt) p_end = Kt;
        if (p_begin > p_end) p_begin = p_end;
        Klast = Kt > 0 ? Kt - 1 : 0;
        inv_chh = chh > 0 ? 1.f / (float)chh : 0.f;
        inv_cww = cww > 0 ? 1.f / (float)cww : 0.f;
    }
    const unsigned plast = (unsigned)Klast, pend = (unsigned)p_end;
    auto unpack = [&](unsigned k, unsigned& n, unsigned& i, unsigned& j) __attribute__((always_inline)) {
        const unsigned q = (unsigned)(((float)k + 0.5f) * inv_cww);
        j = k - q * (unsigned)cww + (unsigned)cw0;
        n = (unsigned)(((float)q + 0.5f) * inv_chh);
        i = q - n * (unsigned)chh + (unsigned)ch0;
    };
    // loader role: A (dy) threads [0, BM), B (x) threads [BM, BM + BN)
    const bool is_a = t < BM, is_b = !is_a && t < BM + BN;
    const int u = is_a ? t : t - BM;
    const int quads = is_a ? BM / 4 : BN / 4;
    const int cq = u % quads, pg = u / quads;                 // channel quad, pixel group (4 pixels) of the 16-pixel K step
    const int ch = (is_a ? co0 : ci0) + cq * 4;
    const int tail = (is_a ? a.Cout : a.C) - ch;               // real channels of the quad (<= 0: none)
    const unsigned ch_e = tail > 0 ? (unsigned)ch : 0u;

    float4 rv[4];
    int rt[4];
    auto load_one = [&](unsigned p, float4& v, int& tl) __attribute__((always_inline)) {
        const unsigned pe = p < plast ? p : plast;
        tl = p < pend ? tail : 0;
        unsigned n, i, j;
        if (compact) {
            unpack(pe, n, i, j);
        } else {
            const unsigned q = fast_div(pe, a.mulW, a.shW);
            j = pe - q * (unsigned)a.W;
            n = fast_div(q, a.mulH, a.shH);
            i = q - n * (unsigned)a.H;
        }
        if (is_a) {
            const unsigned long row = SUBPIX ? ((unsigned long)(n * (unsigned)a.H + i) * 2u + (unsigned)ph_a) * (2u * (unsigned)a.W) + 2u * j + (unsigned)ph_b
                                             : (unsigned long)(n * (unsigned)a.H + i) * (unsigned)a.W + j;
            v = *reinterpret_cast<const float4*>(a.dy + row * (unsigned)a.ld_dy + ch_e);
        } else {
            int hh = (int)i + dyt, ww = (int)j + dxt;
            const bool ok = hh >= 0 && hh <= hmax && ww >= 0 && ww <= wmax;
            hh = hh < 0 ? 0 : (hh > hmax ? hmax : hh);
            ww = ww < 0 ? 0 : (ww > wmax ? wmax : ww);
            if (!ok) tl = 0;
            const unsigned pix = (n * (unsigned)Hs + (unsigned)(hh >> a.ups)) * (unsigned)Ws + (unsigned)(ww >> a.ups);
            v = *reinterpret_cast<const float4*>(a.x + (unsigned long)pix * (unsigned)a.ld_x + ch_e);
        }
    };
    auto load_step = [&](long p0) __attribute__((always_inline)) {
        if (is_a || is_b) {
#pragma unroll
            for (int i = 0; i < 4; ++i) load_one((unsigned)p0 + 4 * pg + i, rv[i], rt[i]);
        }
    };
    auto store_step = [&](int buf) __attribute__((always_inline)) {
        if (!(is_a || is_b)) return;
        float m[4][4];          // [pixel][channel of the quad], masked
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            m[i][0] = rt[i] < 1 ? 0.f : rv[i].x;
            m[i][1] = rt[i] < 2 ? 0.f : rv[i].y;
            m[i][2] = rt[i] < 3 ? 0.f : rv[i].z;
            m[i][3] = rt[i] < 4 ? 0.f : rv[i].w;
        }
        const int rowbase = cq * 4, kg = pg >> 1, half = pg & 1;
        uint4* const planes = is_a ? &Ah[buf][0][0] : &Bh[buf][0][0];
        const int pstride = is_a ? 2 * BMP : 2 * BNP, kstride = is_a ? BMP : BNP;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            uint2 p0, p1, p2;
            mnk_split3(make_float4(m[0][e], m[1][e], m[2][e], m[3][e]), p0, p1, p2);
            const int row = rowbase + e;
            const int chunk = kg * kstride + row + (row >> 4);
            reinterpret_cast<uint2*>(planes + chunk)[half] = p0;
            reinterpret_cast<uint2*>(planes + pstride + chunk)[half] = p1;
            reinterpret_cast<uint2*>(planes + 2 * pstride + chunk)[half] = p2;
        }
    };

    constexpr int NACC = (TM * TN == 1) ? 2 : 1;
    f32x16 acc[NACC][TM][TN];
#pragma unroll
    for (int q = 0; q < NACC; ++q)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[q][i][j][r] = 0.f;
    const int fi = lane & 31, fk = lane >> 5;
    auto mfma_step = [&](int buf) __attribute__((always_inline)) {
        mnk_bf16x8 ha[3][TM], hb[3][TN];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row = wm * (32 * TM) + 32 * i + fi;
                ha[pl][i] = mnk_as_bf16x8(Ah[buf][pl][fk * BMP + row + (row >> 4)]);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int row = wn * (32 * TN) + 32 * j + fi;
                hb[pl][j] = mnk_as_bf16x8(Bh[buf][pl][fk * BNP + row + (row >> 4)]);
            }
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                constexpr int Q = NACC - 1;
                acc[0][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ha[1][i], hb[1][j], acc[0][i][j], 0, 0, 0);
                acc[Q][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ha[2][i], hb[0][j], acc[Q][i][j], 0, 0, 0);
                acc[0][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ha[0][i], hb[2][j], acc[0][i][j], 0, 0, 0);
                acc[Q][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ha[1][i], hb[0][j], acc[Q][i][j], 0, 0, 0);
                acc[0][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ha[0][i], hb[1][j], acc[0][i][j], 0, 0, 0);
                acc[Q][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ha[0][i], hb[0][j], acc[Q][i][j], 0, 0, 0);
            }
    };

    if (p_begin < p_end) {
        load_step(p_begin);
        store_step(0);
        if (p_begin + BK < p_end) load_step(p_begin + BK);
    }
    __syncthreads();
    long p0 = p_begin;
    int it = 0;
    for (; p0 + 2 * BK < p_end; p0 += BK, ++it) {
        const int buf = it & 1;
        store_step(buf ^ 1);
        load_step(p0 + 2 * BK);
        mfma_step(buf);
        __syncthreads();
    }
    if (p0 + BK < p_end) {
        const int buf = it & 1;
        store_step(buf ^ 1);
        mfma_step(buf);
        __syncthreads();
        p0 += BK;
        ++it;
    }
    if (p0 < p_end) mfma_step(it & 1);
    if constexpr (NACC == 2) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][0][0][r] += acc[1][0][0][r];
    }

    float* outp = a.part + ((long)split * a.ntaps + tap) * a.Cout * a.C;
    const unsigned Cu = (unsigned)a.C, corow0 = (unsigned)co0 + wm * (32 * TM) + 4 * fk;
    auto emit = [&](auto full_tag) __attribute__((always_inline)) {
        constexpr bool FULL = decltype(full_tag)::value;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const unsigned ci = (unsigned)ci0 + wn * (32 * TN) + 32 * j + fi;
                const unsigned cb = corow0 + 32 * i, off0 = cb * Cu + ci;
                if (ci < Cu) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const unsigned ro = (r & 3) + 8 * (r >> 2);
                        if (FULL || cb + ro < (unsigned)a.Cout) outp[off0 + ro * Cu] = acc[0][i][j][r];
                    }
                }
            }
    };
    if (co0 + BM <= a.Cout)
        emit(TrueTag{});
    else
        emit(FalseTag{});
}

template <int BM, int BN, int WM, int WN, bool SUBPIX>
__global__ void __launch_bounds__(256, 2) conv3x3_wgrad_tap_h_kernel(WgradTapArgs a) {
    int bx, by;
    xcd_tile(a.xcd, bx, by);
    wgrad_tap_body_h<BM, BN, WM, WN, SUBPIX>(a, bx, by, (int)blockIdx.z);
}

template <int BM, int BN, int WM, int WN, int MODE>
__global__ void __launch_bounds__(256, 3) conv3x3_wgrad_tap_kernel(WgradTapArgs a) {
    int bx, by;
    xcd_tile(a.xcd, bx, by);
    wgrad_tap_body<BM, BN, WM, WN, MODE>(a, bx, by, (int)blockIdx.z);
}

// ---- the same GEMM for MANY layers in one launch ("grouped"): a block looks up its job (layer, source), its tile and
// its pixel chunk.  A backward pass has ~50 tap-major weight-gradient GEMMs of 27..650 tiles each; launched one by one,
// every layer has to be cut into 2..86 pixel splits to fill 256 CUs, and the split partials (1.3 GB per iteration on
// BASELINE configs[1]) are written by the GEMMs and read again by the reductions.  Launched together the tiles of all
// layers fill the chip, so a layer is only split where its pixel range is long (chunks of ~1024 pixels): 0.2 GB.
struct TapJobRec {
    WgradTapArgs a;
    int gm, gnt, splits, block_begin;     // tile grid (gnt = ci tiles x taps), pixel splits; blocks = gm * gnt * splits
};

template <int BM, int BN, int WM, int WN, int MODE>
__global__ void __launch_bounds__(256, 3) conv3x3_wgrad_tap_grouped_kernel(const TapJobRec* __restrict__ recs, int n) {
    __shared__ int sh_idx;
    // XCD-aware order (see xcd_tile): workgroup L runs on XCD L % 8; XCD class c takes the contiguous range
    // [c * per, (c + 1) * per) of the logical block order, in which the blocks that share a pixel chunk of a layer
    // (all its co / ci tiles and taps) are neighbours -- so a chunk's activations are fetched into one L2, not eight
    int b = blockIdx.x;
    {
        const unsigned total = gridDim.x, L = blockIdx.x, c = L & 7u, base = total >> 3, rem = total & 7u;
        if (total >= 64) b = (int)(c * base + (c < rem ? c : rem) + (L >> 3));
    }
    const int di = find_desc(&recs[0].block_begin, (int)(sizeof(TapJobRec) / sizeof(int)), n, b, &sh_idx);
    // the record is wave-uniform: keep it in scalar registers (copied field by field through readfirstlane by the compiler
    // when it can prove uniformity; `di` comes from LDS, so say it explicitly)
    const TapJobRec* __restrict__ rp = recs + __builtin_amdgcn_readfirstlane(di);
    const WgradTapArgs a = rp->a;
    const int local = b - rp->block_begin;
    const int gm = rp->gm, gnt = rp->gnt;
    const int bx = local % gm, rest = local / gm;
    const int by = rest % gnt, split = rest / gnt;
    wgrad_tap_body<BM, BN, WM, WN, MODE>(a, bx, by, split);
}

template <int BM, int BN, int WM, int WN, bool SUBPIX>
__global__ void __launch_bounds__(256, 2) conv3x3_wgrad_tap_grouped_h_kernel(const TapJobRec* __restrict__ recs, int n) {
    __shared__ int sh_idx;
    int b = blockIdx.x;
    {
        const unsigned total = gridDim.x, L = blockIdx.x, c = L & 7u, base = total >> 3, rem = total & 7u;
        if (total >= 64) b = (int)(c * base + (c < rem ? c : rem) + (L >> 3));
    }
    const int di = find_desc(&recs[0].block_begin, (int)(sizeof(TapJobRec) / sizeof(int)), n, b, &sh_idx);
    const TapJobRec* __restrict__ rp = recs + __builtin_amdgcn_readfirstlane(di);
    const WgradTapArgs a = rp->a;
    const int local = b - rp->block_begin;
    const int gm = rp->gm, gnt = rp->gnt;
    const int bx = local % gm, rest = local / gm;
    const int by = rest % gnt, split = rest / gnt;
    wgrad_tap_body_h<BM, BN, WM, WN, SUBPIX>(a, bx, by, split);
}

// the nine-tap 16x16 kernel for MANY narrow layers in one launch (the eight 45 -> 45 convolutions of the refinement stack:
// launched one by one each needs 512 pixel splits to fill the chip -- 37 MB of partials per layer; together 128 do)
struct N16JobRec {      // same size and block_begin offset as TapJobRec (one table, one lookup)
    WgradN16Args a;
    char pad[sizeof(WgradTapArgs) - sizeof(WgradN16Args)];
    int gm, gn, splits, block_begin;
};
static_assert(sizeof(N16JobRec) == sizeof(TapJobRec), "grouped job records share one table");

template <int NCT, int NCI>
__global__ void __launch_bounds__(256, 2) conv3x3_wgrad_n16_grouped_kernel(const N16JobRec* __restrict__ recs, int n) {
    __shared__ int sh_idx;
    const int b = blockIdx.x;
    const int di = find_desc(&recs[0].block_begin, (int)(sizeof(N16JobRec) / sizeof(int)), n, b, &sh_idx);
    const N16JobRec* __restrict__ rp = recs + __builtin_amdgcn_readfirstlane(di);
    const WgradN16Args a = rp->a;
    const int local = b - rp->block_begin;
    const int gm = rp->gm, gn = rp->gn;
    const int bx = local % gm, rest = local / gm;
    wgrad_n16_body<NCT, NCI>(a, bx, rest % gn, rest / gn);
}

// dw[co][(c_start + ci) * ntaps + tap] = sum_s part[s][tap][co][ci]: one block per (64-channel ci tile, co row);
// reads are coalesced along ci with four independent split-sum chains per element (loads in flight), the
// (tap, ci) -> (ci, tap) transposition goes through LDS, writes are contiguous runs of 64 * ntaps floats.
// Fixed summation order (deterministic).
// (up_fold / up_fold_pairs: pack_tile.h -- the optimiser kernel folds tap-major partials too)
__global__ void __launch_bounds__(256) conv3x3_wgrad_tap_reduce_kernel(const float* __restrict__ part, int splits,
                                                                       int ntaps, int Cout, int C,
                                                                       float* __restrict__ dw, long ld_out, int up) {
    // up: `ntaps` = 16 pseudo taps of the sub-pixel form in `part`, folded into the 9 kernel taps of dw
    __shared__ float tile[16][65];
    const int t = threadIdx.x;
    const int ci0 = blockIdx.x * 64, co = blockIdx.y;
    const long plane = (long)Cout * C, sstride = (long)ntaps * plane;
    const int c = t & 63;
    const bool c_ok = ci0 + c < C;
    for (int tp = t >> 6; tp < ntaps; tp += 4) {
        const float* src = c_ok ? part + (long)tp * plane + (long)co * C + ci0 + c : part;   // always loadable
        float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
        int sp = 0;
        for (; sp + 3 < splits; sp += 4) {
            const float x0 = src[(long)sp * sstride], x1 = src[(long)(sp + 1) * sstride];
            const float x2 = src[(long)(sp + 2) * sstride], x3 = src[(long)(sp + 3) * sstride];
            v0 += x0;
            v1 += x1;
            v2 += x2;
            v3 += x3;
        }
        for (; sp < splits; ++sp) v0 += src[(long)sp * sstride];
        tile[tp][c] = c_ok ? (v0 + v1) + (v2 + v3) : 0.f;
    }
    __syncthreads();
    const int nout = up ? 9 : ntaps;
    float* dst = dw + (long)co * ld_out + (long)ci0 * nout;
    const int lim = (C - ci0 < 64 ? C - ci0 : 64) * nout;
    for (int idx = t; idx < lim; idx += 256) {
        const int cc = idx / nout, tp = idx - cc * nout;
        dst[idx] = up ? up_fold(&tile[0][cc], tp / 3, tp % 3, 65) : tile[tp][cc];
    }
}

// first stage for many-split layers (large pixel counts, small dW): out[z][i] = sum of the splits of group z, so the
// summation runs over (elements x groups) threads instead of elements only; the transposing kernel above then sums
// the groups.  Fixed order inside a group and over the groups (deterministic).
__global__ void __launch_bounds__(256) conv3x3_wgrad_group_sum_kernel(const float* __restrict__ part, long n, int splits,
                                                                      int per_group, float* __restrict__ out) {
    const int z = blockIdx.y;
    const int s0 = z * per_group;
    int s1 = s0 + per_group;
    if (s1 > splits) s1 = splits;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float* src = part + i;
        float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
        int sp = s0;
        for (; sp + 3 < s1; sp += 4) {
            const float x0 = src[(long)sp * n], x1 = src[(long)(sp + 1) * n];
            const float x2 = src[(long)(sp + 2) * n], x3 = src[(long)(sp + 3) * n];
            v0 += x0;
            v1 += x1;
            v2 += x2;
            v3 += x3;
        }
        for (; sp < s1; ++sp) v0 += src[(long)sp * n];
        out[(long)z * n + i] = (v0 + v1) + (v2 + v3);
    }
}

// dw[co][c_start*9 + n] = sum_splits partial[s][co][n]
__global__ void __launch_bounds__(256) conv3x3_wgrad_reduce_kernel(const float* __restrict__ ws, int splits, int Cout,
                                                                   int NT, float* __restrict__ dw, long ld_out) {
    __shared__ float sm[4][64];
    const int o = threadIdx.x & 63, g = threadIdx.x >> 6;
    const long total = (long)Cout * NT;
    for (long base = (long)blockIdx.x * 64; base < total; base += (long)gridDim.x * 64) {
        const long i = base + o;
        float v = 0.f;
        if (i < total)
            for (int s = g; s < splits; s += 4) v += ws[(long)s * total + i];
        sm[g][o] = v;
        __syncthreads();
        if (g == 0 && i < total) {
            const int n = (int)(i % NT);
            const int co = (int)(i / NT);
            dw[(long)co * ld_out + n] = (sm[0][o] + sm[1][o]) + (sm[2][o] + sm[3][o]);
        }
        __syncthreads();
    }
}

// ---- split reductions of MANY layers in one launch (deferred weight-gradient reductions of a whole backward pass) ----
// Block b belongs to the layer whose [block_begin, block_begin + blocks) range contains it -- found with ONE coalesced
// read of the table's block_begin column and an LDS count (a binary search over device memory costs ~6 dependent loads per
// block, several microseconds on blocks that move 5 KB) -- and owns a tw-channel ci tile of one output row (or of four
// rows): thread group g sums the splits g, g + groups, ... with 2 * ntaps independent loads in flight, the groups are
// combined through LDS in a fixed order (deterministic), and the (tap, ci) -> (ci, tap) transposition of the tap-major
// partials happens on the way into LDS.  blocks = ceil(Cout / rows) * ceil(C / tw) with (tw, rows) = reduce_map(splits).
// thread map of one layer: channel-tile width tw and thread groups = 256 / tw
//   splits <  4 : tw = 64, the 4 groups own 4 different output rows (each sums all its splits);
//   splits < 32 : tw = 64, the 4 groups share one row and split the splits;
//   splits >= 32: tw = 16, 16 groups share one row (the many-split layers have tiny gradients: without this a 512-split
//                 layer leaves ten blocks summing 128 partials in sequence -- the tail of the whole launch).
__host__ __device__ __forceinline__ void reduce_map(int splits, int* tw, int* rows) {
    *tw = splits >= 32 ? 16 : 64;
    *rows = splits < 4 ? 4 : 1;
}

// Few-split layers with C % 4 == 0 -- the deep levels (2x2 ... 8x8 maps, 256 ... 2048 channels), whose "partials" ARE the
// gradient (240 of the 265 MB of BASELINE configs[1]) in tap-major order -- take a flat map instead: a thread owns four
// consecutive input channels of one output row, reads one float4 per tap and split (1 KB contiguous per 64 lanes; the tile
// map above reads 256-byte pieces: ~1.8 TB/s measured) and writes its 4 * ntaps consecutive gradient floats from registers:
// no LDS, no barrier.  blocks = ceil(Cout * C / 4 / 256).
__host__ __device__ __forceinline__ bool reduce_flat(int splits, int C) { return splits < 4 && (C & 3) == 0; }

// out[e * NT + tp] = v[tp].e for the four channels e of a thread: the (tap, ci) -> (ci, tap) transposition in registers
template <int NT>
__device__ __forceinline__ void flat_store(const float4* v, float* __restrict__ dst, bool accumulate) {
    float o[4 * NT];
#pragma unroll
    for (int tp = 0; tp < NT; ++tp) {
        o[0 * NT + tp] = v[tp].x;
        o[1 * NT + tp] = v[tp].y;
        o[2 * NT + tp] = v[tp].z;
        o[3 * NT + tp] = v[tp].w;
    }
    if (((size_t)dst & 15) == 0) {
#pragma unroll
        for (int k = 0; k < NT; ++k) {
            float4 w = make_float4(o[4 * k], o[4 * k + 1], o[4 * k + 2], o[4 * k + 3]);
            if (accumulate) {
                const float4 old = *reinterpret_cast<const float4*>(dst + 4 * k);
                w = make_float4(old.x + w.x, old.y + w.y, old.z + w.z, old.w + w.w);
            }
            *reinterpret_cast<float4*>(dst + 4 * k) = w;
        }
    } else {
#pragma unroll
        for (int k = 0; k < 4 * NT; ++k) dst[k] = accumulate ? dst[k] + o[k] : o[k];
    }
}

template <int NIN, int NOUT, int LAYOUT>      // taps read / written; LAYOUT 0 tap-major, 2 tap-major sub-pixel (16 -> 9), 1 parameter-major
__device__ __forceinline__ void reduce_flat_body(const MnkWgradReduceDesc& d, int co, int ci) {
    float4 acc[NIN];
#pragma unroll
    for (int tp = 0; tp < NIN; ++tp) acc[tp] = make_float4(0.f, 0.f, 0.f, 0.f);
    float* dst = d.dw + ((long)co * d.Cin_total + d.c_start + ci) * NOUT;
    if (LAYOUT == 1) {
        // part[s][co][ci * NOUT + tap]: the thread's 4 * NOUT floats are contiguous and already in gradient order
        const long NT = (long)d.C * NOUT, sstride = (long)d.Cout * NT;
        const float* src = d.part + (long)co * NT + (long)ci * NOUT;
        for (int sp = 0; sp < d.splits; ++sp) {
#pragma unroll
            for (int k = 0; k < NIN; ++k) {
                const float4 v = *reinterpret_cast<const float4*>(src + (long)sp * sstride + 4 * k);
                acc[k] = make_float4(acc[k].x + v.x, acc[k].y + v.y, acc[k].z + v.z, acc[k].w + v.w);
            }
        }
        if (((size_t)dst & 15) == 0) {
#pragma unroll
            for (int k = 0; k < NIN; ++k) {
                float4 w = acc[k];
                if (d.accumulate) {
                    const float4 old = *reinterpret_cast<const float4*>(dst + 4 * k);
                    w = make_float4(old.x + w.x, old.y + w.y, old.z + w.z, old.w + w.w);
                }
                *reinterpret_cast<float4*>(dst + 4 * k) = w;
            }
        } else {
#pragma unroll
            for (int k = 0; k < NIN; ++k) {
                const float e[4] = {acc[k].x, acc[k].y, acc[k].z, acc[k].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) dst[4 * k + j] = d.accumulate ? dst[4 * k + j] + e[j] : e[j];
            }
        }
        return;
    }
    const long plane = (long)d.Cout * d.C, sstride = (long)NIN * plane;
    const float* src = d.part + (long)co * d.C + ci;
    for (int sp = 0; sp < d.splits; ++sp) {                      // NIN independent 16-byte loads in flight per split
#pragma unroll
        for (int tp = 0; tp < NIN; ++tp) {
            const float4 v = *reinterpret_cast<const float4*>(src + (long)sp * sstride + (long)tp * plane);
            acc[tp] = make_float4(acc[tp].x + v.x, acc[tp].y + v.y, acc[tp].z + v.z, acc[tp].w + v.w);
        }
    }
    if (LAYOUT == 2) {                                           // fold the 16 pseudo taps into the nine kernel taps, per channel
        float ax[16], ay[16], az[16], aw[16];
#pragma unroll
        for (int tp = 0; tp < 16; ++tp) ax[tp] = acc[tp].x, ay[tp] = acc[tp].y, az[tp] = acc[tp].z, aw[tp] = acc[tp].w;
        float4 f[9];
#pragma unroll
        for (int tp = 0; tp < 9; ++tp)
            f[tp] = make_float4(up_fold(ax, tp / 3, tp % 3, 1), up_fold(ay, tp / 3, tp % 3, 1), up_fold(az, tp / 3, tp % 3, 1),
                                up_fold(aw, tp / 3, tp % 3, 1));
        flat_store<9>(f, dst, d.accumulate != 0);
    } else {
        flat_store<NOUT>(acc, dst, d.accumulate != 0);
    }
}

// parameter-major partials part[s][co][ci * ntaps + tap]: this thread's elements src[0], src[TW], ... (those below `left`)
// summed over the splits s0, s0 + sstep, ... into out[0], out[TW], ...; TW is a compile-time constant so that the eight
// elements of a pass share one address register pair (immediate offsets)
template <int TW>
__device__ __forceinline__ void reduce_param_major(const float* __restrict__ src, long sstride, int splits, int s0, int sstep,
                                                   int left, bool row_ok, float* __restrict__ out) {
    const int nk = (left + TW - 1) / TW;                         // elements of this thread (<= 0: none)
    for (int k0 = 0; k0 < nk; k0 += 8) {                         // eight elements x two splits in flight
        float a0[8], a1[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) a0[k] = a1[k] = 0.f;
        const float* sk = src + k0 * TW;
        int sp = s0;
        for (; sp + sstep < splits; sp += 2 * sstep) {
            const float* ps = sk + (long)sp * sstride;
            const float* pt = ps + (long)sstep * sstride;
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (k0 + k < nk) {
                    a0[k] += ps[k * TW];
                    a1[k] += pt[k * TW];
                }
        }
        if (sp < splits) {
            const float* ps = sk + (long)sp * sstride;
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (k0 + k < nk) a0[k] += ps[k * TW];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (k0 + k < nk) out[(k0 + k) * TW] = row_ok ? a0[k] + a1[k] : 0.f;
    }
}

// (four waves per SIMD: the kernel needs 132 registers unconstrained, i.e. three waves; capped at 128 it reduces the generator's
// 692 MB of partials in 311 instead of 361 us; five waves -- 96 registers -- spill: 765 us.  tools/reduce_probe.py)
__global__ void __launch_bounds__(256, 4) wgrad_reduce_multi_kernel(const MnkWgradReduceDesc* __restrict__ descs, int n) {
    __shared__ float sm[16 * 16 * 16 + 64];        // [group][channel * ntaps + tap], group stride tw * 16
    __shared__ int sh_idx;
    const int b = blockIdx.x;
    const int di = find_desc(&descs[0].block_begin, (int)(sizeof(MnkWgradReduceDesc) / sizeof(int)), n, b, &sh_idx);
    const MnkWgradReduceDesc d = descs[di];
    const int local = b - d.block_begin;
    if (reduce_flat(d.splits, d.C) && (d.ntaps == 9 || d.ntaps == 16)) {      // (block-uniform: no barrier follows on this path)
        const int q4 = d.C >> 2;
        const long item = (long)local * 256 + threadIdx.x;
        if (item >= (long)d.Cout * q4) return;
        const int co = (int)(item / q4), ci = 4 * (int)(item - (long)co * q4);
        if (d.layout == 2)
            reduce_flat_body<16, 9, 2>(d, co, ci);
        else if (d.layout == 0 && d.ntaps == 9)
            reduce_flat_body<9, 9, 0>(d, co, ci);
        else if (d.layout == 0)
            reduce_flat_body<16, 16, 0>(d, co, ci);
        else if (d.ntaps == 9)
            reduce_flat_body<9, 9, 1>(d, co, ci);
        else
            reduce_flat_body<16, 16, 1>(d, co, ci);
        return;
    }
    int tw, rpb;
    reduce_map(d.splits, &tw, &rpb);
    const int groups = 256 / tw, gstride = tw * 16;
    const int ctiles = (d.C + tw - 1) / tw;
    const int rq = local / ctiles, ci0 = (local - rq * ctiles) * tw;
    const int t = threadIdx.x, g = t / tw, c = t - g * tw;
    const int ntaps = d.ntaps;
    const int cw = d.C - ci0 < tw ? d.C - ci0 : tw;          // channels of this tile
    const int lim = cw * ntaps;                              // gradient floats of this tile (per row)
    const int co = rpb == 1 ? rq : rq * 4 + g;               // the row this thread group reads
    const bool row_ok = co < d.Cout;
    const int s0 = rpb == 1 ? g : 0, sstep = rpb == 1 ? groups : 1;
    float* smg = sm + g * gstride;
    if (d.layout == 0 || d.layout == 2) {
        // part[s][tap][co][ci]; layout 2: 16 pseudo taps of the sub-pixel form, folded into the 9 kernel taps below
        const int nin = d.layout == 2 ? 16 : ntaps;
        const long plane = (long)d.Cout * d.C, sstride = (long)nin * plane;
        const bool ok = c < cw && row_ok;
        const float* src = d.part + (long)(row_ok ? co : 0) * d.C + ci0 + (ok ? c : 0);
        float acc[16], acc2[16];
#pragma unroll
        for (int tp = 0; tp < 16; ++tp) acc[tp] = acc2[tp] = 0.f;
        int sp = s0;
        for (; sp + sstep < d.splits; sp += 2 * sstep) {       // two splits per trip: 2 * ntaps loads in flight
            const float* ps = src + (long)sp * sstride;
            const float* pt = ps + (long)sstep * sstride;
#pragma unroll
            for (int tp = 0; tp < 16; ++tp)
                if (tp < nin) {
                    acc[tp] += ps[(long)tp * plane];
                    acc2[tp] += pt[(long)tp * plane];
                }
        }
        if (sp < d.splits) {
            const float* ps = src + (long)sp * sstride;
#pragma unroll
            for (int tp = 0; tp < 16; ++tp)
                if (tp < nin) acc[tp] += ps[(long)tp * plane];
        }
#pragma unroll
        for (int tp = 0; tp < 16; ++tp) acc[tp] += acc2[tp];
        if (d.layout == 2) {
#pragma unroll
            for (int tp = 0; tp < 9; ++tp) smg[c * 9 + tp] = ok ? up_fold(acc, tp / 3, tp % 3, 1) : 0.f;
        } else {
#pragma unroll
            for (int tp = 0; tp < 16; ++tp)
                if (tp < ntaps) smg[c * ntaps + tp] = ok ? acc[tp] : 0.f;      // already in (ci, tap) order
        }
    } else {
        // part[s][co][ci * ntaps + tap]
        const long NT = (long)d.C * ntaps, sstride = (long)d.Cout * NT;
        // a thread owns the elements c, c + tw, ... of the tile's lim = cw * ntaps contiguous floats (at most ntaps of them)
        // and walks the splits with eight of them in flight (one at a time meant two loads in flight and ntaps passes over
        // the splits: the 512-split 45 -> 45 layers' blocks were the tail of the launch)
        const float* src = d.part + (long)(row_ok ? co : 0) * NT + (long)ci0 * ntaps + c;
        if (tw == 16)
            reduce_param_major<16>(src, sstride, d.splits, s0, sstep, lim - c, row_ok, smg + c);
        else
            reduce_param_major<64>(src, sstride, d.splits, s0, sstep, lim - c, row_ok, smg + c);
    }
    __syncthreads();
    if (rpb == 1) {
        float* dst = d.dw + ((long)rq * d.Cin_total + d.c_start + ci0) * ntaps;
        for (int idx = t; idx < lim; idx += 256) {
            float v = 0.f;
            for (int gg = 0; gg < groups; ++gg) v += sm[gg * gstride + idx];       // fixed order: deterministic
            dst[idx] = d.accumulate ? dst[idx] + v : v;
        }
    } else if (row_ok) {
        float* dst = d.dw + ((long)co * d.Cin_total + d.c_start + ci0) * ntaps;
        for (int idx = c; idx < lim; idx += tw) dst[idx] = d.accumulate ? dst[idx] + smg[idx] : smg[idx];
    }
}

struct Plan {
    int bm, bn, gm, gn, splits, ksteps, ksteps_per_split, ldw;
};

// split-K plan values (defaults from the MI355X sweeps in profiles/README.md; tuning_knob: settable by name through
// mnk_set_tuning / MNK_TUNING for tuning runs, no environment switch of their own)
static int g_split_tiles = tuning_knob("split_tiles", &g_split_tiles, 192), g_split_target = tuning_knob("split_target", &g_split_target, 512),
           g_split_minsteps = tuning_knob("split_minsteps", &g_split_minsteps, 6);
// (the weight-gradient splits are deterministic partials + a reduce kernel; an fp32-atomic form measured equal in round 1 and
// was removed in round 6 together with -munsafe-fp-atomics: no floating-point atomic exists in this library)
// 1 (default): a split-K forward launch that was asked for BatchNorm statistics sums its partials with
// conv3x3_splitk_reduce_stats_kernel (one launch for reduction + statistics pass); 0: no statistics from split launches
static int g_splitk_stats = tuning_knob("splitk_stats", &g_splitk_stats, 1);
// 1: every split-K reduction runs the float4-row kernel (the statistics kernel without its statistics; the same bits);
// 0: conv3x3_splitk_reduce_kernel (64 outputs x 4 split groups per block, combined through LDS)
static int g_reduce_v4 = tuning_knob("reduce_v4", &g_reduce_v4, 1);
static int g_wsplit_tiles = tuning_knob("wsplit_tiles", &g_wsplit_tiles, 512), g_wsplit_target = tuning_knob("wsplit_target", &g_wsplit_target, 1024),
           g_wsplit_minsteps = tuning_knob("wsplit_minsteps", &g_wsplit_minsteps, 8);

// mid-size layers (fewer than ~2 blocks per CU with 128-row tiles) use 64-row tiles: twice the blocks, so every SIMD
// has a second wave to overlap loads with MFMA, and less (or no) split-K
static int g_bm64_tiles = tuning_knob("bm64_tiles", &g_bm64_tiles, 512);
static int g_split64_tiles = tuning_knob("split64_tiles", &g_split64_tiles, 384), g_split64_target = tuning_knob("split64_target", &g_split64_target, 1024),
           g_split64_deep = tuning_knob("split64_deep", &g_split64_deep, 32), g_split64_minsteps = tuning_knob("split64_minsteps", &g_split64_minsteps, 16),
           g_split64_tiny = tuning_knob("split64_tiny", &g_split64_tiny, 4);      // K steps per split of the tiny-problem rule; 0: off
static int g_bn128_kwork = tuning_knob("bn128_kwork", &g_bn128_kwork, 8388);   // 1000 pixels x channels from which 128-wide tiles are used
static int g_xcd_remap = tuning_knob("xcd_remap", &g_xcd_remap, 1);
static int g_fast_loader = tuning_knob("fast_loader", &g_fast_loader, 1);
static int g_kxk_fast = tuning_knob("kxk_fast", &g_kxk_fast, 1);     // buffer-load loader for K x K / any pad (MODE 3)
static int g_mfma16 = tuning_knob("mfma16", &g_mfma16, 1);
// 1: the 32x32-tile implicit-GEMM kernels (forward / data gradient) run their products on the bf16 matrix cores through the exact
// three-way split of both fp32 operands (conv3x3_igemm_kernel<..., GM = 1>, mnk_common.h); 0: v_mfma_f32_32x32x2_f32
static int g_gemm_bf16x3 = tuning_knob("gemm_bf16x3", &g_gemm_bf16x3, 0);
// with gemm_bf16x3, 1: 33 .. 48 output channels (the 45-channel refinement stack) take the 64-wide 32x32-tile kernel -- 45 of 64
// columns at 2.67x the matrix rate -- instead of the 48-wide 16x16x4 fp32 kernel.  Measured SLOWER (10.08 vs 9.92 ms per step,
// profiles/r06_knob_ab_log.txt): 0 keeps the 16x16 kernel
// 1: the 16x16-tile kernels (Cout <= 16, 33 .. 48: the 45-channel refinement stack) too: pairs of K steps on v_mfma_f32_16x16x32_bf16
static int g_gemm16_bf16x3 = tuning_knob("gemm16_bf16x3", &g_gemm16_bf16x3, 0);
// 1: the tap-major weight-gradient kernels on the bf16 matrix cores too (wgrad_tap_body_h: transposing loader)
static int g_wgrad_bf16x3 = tuning_knob("wgrad_bf16x3", &g_wgrad_bf16x3, 0);
static int g_gemm_bf16x3_n48 = tuning_knob("gemm_bf16x3_n48", &g_gemm_bf16x3_n48, 0);
static bool narrow48_on_wide_tiles() { return g_gemm_bf16x3 && g_gemm_bf16x3_n48; }

struct PlanRow {
    long M;
    int Cout, chunks, ntaps, phases, bm, bn, splits;
};
static const PlanRow g_tuned_rows[] = {
#include "plan_table.h"
    {0, 0, 0, 0, 0, 0, 0, 0}};
static const PlanRow g_tuned_rows_bf16x3[] = {
#include "plan_table_bf16x3.h"
    {0, 0, 0, 0, 0, 0, 0, 0}};
static int g_plan_table = tuning_knob("plan_table", &g_plan_table, 1), g_force_bm = tuning_knob("force_bm", &g_force_bm, 0),
           g_force_bn = tuning_knob("force_bn", &g_force_bn, 0), g_force_splits = tuning_knob("force_splits", &g_force_splits, 0);
static long g_last_plan[8];

// block tiles the GEMM kernels are instantiated for (conv2d_fwd_impl's dispatch)
static bool plan_tile_ok(int bm, int bn, int Cout, int phases) {
    if (bn == 48 && narrow48_on_wide_tiles()) return false;
    if (bn == 16 || bn == 48) return bm == 128 && phases == 1 && g_mfma16 && Cout <= bn;
    if (bn == 32) return bm == 128;
    return (bn == 64 || bn == 128) && (bm == 64 || bm == 128);
}

static Plan make_plan(long M, int Cout, int chunks, int ntaps = 9, int phases = 1) {
    // phases > 1 (sub-pixel form): M = pixels of ONE phase, p.gm = tiles of one phase; the launch has phases * gm M tiles
    Plan p;
    p.bn = Cout > 64 ? 128 : (Cout > 32 ? 64 : 32);
    if (g_mfma16 && phases == 1) {   // narrow outputs: 16x16x4 MFMA tiles (BN = 16 / 48), see conv3x3_igemm16_kernel
        if (Cout <= 16) p.bn = 16;
        else if (Cout > 32 && Cout <= 48 && !narrow48_on_wide_tiles()) p.bn = 48;
    }
    // measured on the MI355X over both benchmark configurations' layer shapes (tools/plan_tune.py, profiles/r02_plan_tune_*.txt):
    // the 64x64 tile (56 registers, 20 KB of LDS: 8 blocks per CU) is the fastest instantiation for every layer wider than
    // 48 channels -- by 5..40 % where 128-wide tiles left CUs idle or forced a split-K the 64x64 plan does not need -- except
    // large layers whose width is a multiple of 128 (>= 65536 pixels x 128 channels: level with the 128x128 tile)
    // (bn128_kwork < 0: the previous rule -- 128-wide tiles for every layer wider than 64 channels -- for A/B runs)
    const bool small_tiles = g_bn128_kwork >= 0;
    if (small_tiles && p.bn == 128 && (Cout % 128 != 0 || (double)M * phases * Cout < (double)g_bn128_kwork * 1000.0)) p.bn = 64;
    p.gn = ceil_div(Cout, p.bn);
    p.bm = 128;
    if ((small_tiles && p.bn == 64) || (p.bn >= 64 && (long)ceil_div(M, 128) * p.gn * phases < g_bm64_tiles)) p.bm = 64;
    p.gm = ceil_div(M, p.bm);
    p.ksteps = ntaps * chunks;
    long tiles = (long)p.gm * p.gn * phases;
    int splits = 1;
    if (small_tiles && p.bn == 64 && p.bm == 64) {
        // 64x64 tiles (same sweep): a CU holds eight of these blocks, and the fastest plans put ~1024 blocks on the 256 CUs
        // with >= 16 K steps each; layers that already have a block per CU only gain once a split is >= 32 steps deep (the
        // partials cross HBM twice and, in front of a BatchNorm, a split plan's epilogue cannot produce the statistics)
        if (tiles < g_split64_tiles) {
            splits = (int)((g_split64_target + tiles / 2) / tiles);
            if (splits > p.ksteps / g_split64_minsteps) splits = p.ksteps / g_split64_minsteps;
            if (splits < 1) splits = 1;
            if (tiles >= 192 && p.ksteps / splits < g_split64_deep) splits = 1;
            // (round 6) tiny problems -- the per-frame evaluation loops at batch 1 (reconstruction.py:45-62): 4 ... 32 tiles with
            // 36 ... 150 K steps -- are one serial K loop per block on a mostly idle chip: a launch's time is its loop length, so
            // splits as short as 4 steps pay until ~256 blocks exist (tools/plan_tune.py --eval, profiles/r06_plan_tune_eval_*:
            // 18.4 -> 13.2 us for conv + reduction of a 256-pixel layer).  Plans that already fill half the chip are left alone.
            if (g_split64_tiny && tiles < 64 && tiles * splits < 128) {
                int s2 = (int)(256 / tiles);
                if (s2 > p.ksteps / g_split64_tiny) s2 = p.ksteps / g_split64_tiny;
                if (s2 > splits) splits = s2;
            }
        }
    } else if (tiles < g_split_tiles) {
        splits = (int)((g_split_target + tiles - 1) / tiles);
        int max_splits = p.ksteps / g_split_minsteps;      // keep >= 6 K steps (96 deep) per split
        if (splits > max_splits) splits = max_splits;
        if (splits < 1) splits = 1;
    }
    // measured plans: the benchmark configurations' layer shapes were swept on the MI355X (tools/plan_tune.py -> plan_table.h);
    // a forced plan (mnk_set_tuning MNK_FORCE_BM / _BN / _SPLITS) is what that sweep drives.  Anything the kernels have no
    // instantiation for keeps the rule's choice.
    int want_bm = 0, want_bn = 0, want_splits = 0;
    if (g_plan_table) {
        const PlanRow* tables[2] = {g_gemm_bf16x3 ? g_tuned_rows_bf16x3 : g_tuned_rows, g_gemm_bf16x3 ? g_tuned_rows : nullptr};
        for (int ti = 0; ti < 2 && !want_bm && !want_splits; ++ti)
            for (const PlanRow* r = tables[ti]; r && r->M; ++r)
                if (r->M == M && r->Cout == Cout && r->chunks == chunks && r->ntaps == ntaps && r->phases == phases) {
                    want_bm = r->bm, want_bn = r->bn, want_splits = r->splits;
                    break;
                }
    }
    if (g_force_bm) want_bm = g_force_bm;
    if (g_force_bn) want_bn = g_force_bn;
    if (g_force_splits) want_splits = g_force_splits;
    if (want_bm || want_bn) {
        const int bm = want_bm ? want_bm : p.bm, bn = want_bn ? want_bn : p.bn;
        if (plan_tile_ok(bm, bn, Cout, phases)) {
            p.bm = bm, p.bn = bn;
            p.gn = ceil_div(Cout, p.bn);
            p.gm = ceil_div(M, p.bm);
        }
    }
    if (want_splits > 0) splits = want_splits > p.ksteps ? p.ksteps : want_splits;
    p.ksteps_per_split = (p.ksteps + splits - 1) / splits;
    p.splits = (p.ksteps + p.ksteps_per_split - 1) / p.ksteps_per_split;
    p.ldw = round_up(Cout, 4);
    g_last_plan[0] = M, g_last_plan[1] = Cout, g_last_plan[2] = chunks, g_last_plan[3] = ntaps, g_last_plan[4] = phases;
    g_last_plan[5] = p.bm, g_last_plan[6] = p.bn, g_last_plan[7] = p.splits;
    return p;
}

struct WPlan {
    int bm, gm, gn, splits;
    long pix_per_split;
};

// LDS-halo wgrad plan
struct HPlan {
    bool use;
    int TR, TC, tiles_w, tiles_per_img, gm, gn, splits;
    long total_tiles, tiles_per_split;
};
static int g_wgrad_halo = tuning_knob("wgrad_halo", &g_wgrad_halo, 1), g_whalo_target = tuning_knob("whalo_target", &g_whalo_target, 768),
           g_whalo_mintiles = tuning_knob("whalo_mintiles", &g_whalo_mintiles, 8);

static HPlan make_hplan(int N, int H, int W, int Cout, int C) {
    HPlan p;
    // measured on the MI355X (profiles/README.md): the halo kernel wins when its 64x64 (co, ci) slab is reasonably
    // full; narrow layers (3 input channels, 10/13/32 output channels with ragged ci) stay on the gather kernel
    const double fill = ((double)C / round_up(C, 64)) * ((double)Cout / round_up(Cout, 64));
    p.use = g_wgrad_halo && W >= 16 && (W % 2) == 0 && H >= 2 && fill >= 0.45;
    if (!p.use) return p;
    p.TC = W < 64 ? W : 64;
    if (64 % p.TC != 0) {        // widths that do not divide the 64-pixel tile: keep the gather kernel
        p.use = false;
        return p;
    }
    p.TR = 64 / p.TC;
    p.tiles_w = ceil_div(W, p.TC);
    p.tiles_per_img = ceil_div(H, p.TR) * p.tiles_w;
    p.total_tiles = (long)N * p.tiles_per_img;
    p.gm = ceil_div(Cout, 64);
    p.gn = ceil_div(C, 64);
    long base = (long)p.gm * p.gn * 3;                             // x3: one block per tap row
    long splits = (g_whalo_target + base - 1) / base;
    if (splits > p.total_tiles / g_whalo_mintiles) splits = p.total_tiles / g_whalo_mintiles;   // tiles per block
    if (splits < 1) splits = 1;
    p.tiles_per_split = (p.total_tiles + splits - 1) / splits;
    p.splits = (int)((p.total_tiles + p.tiles_per_split - 1) / p.tiles_per_split);
    return p;
}

static WPlan make_wplan(long M, int Cout, int C, int ntaps = 9) {
    WPlan p;
    p.bm = Cout > 64 ? 128 : (Cout > 32 ? 64 : 32);
    p.gm = ceil_div(Cout, p.bm);
    p.gn = ceil_div(ntaps * C, 128);
    long tiles = (long)p.gm * p.gn;
    long steps = (M + BK - 1) / BK;
    long splits = 1;
    if (tiles < g_wsplit_tiles) {
        splits = (g_wsplit_target + tiles - 1) / tiles;
        long max_splits = steps / g_wsplit_minsteps;       // >= 128 pixels per split
        if (splits > max_splits) splits = max_splits;
        if (splits < 1) splits = 1;
    }
    long steps_per = (steps + splits - 1) / splits;
    p.pix_per_split = steps_per * BK;
    p.splits = (int)((steps + steps_per - 1) / steps_per);
    return p;
}

// narrow-layer (16x16 tiles, nine taps per block) wgrad plan
struct NPlan {
    bool use;
    int gm, gn, tiles_w, tiles_per_img, splits;
    long total_tiles, tiles_per_split;
};
static int g_wgrad_n16 = tuning_knob("wgrad_n16", &g_wgrad_n16, 1), g_wn16_target = tuning_knob("wn16_target", &g_wn16_target, 512),
           g_wn16_mintiles = tuning_knob("wn16_mintiles", &g_wn16_mintiles, 2), g_wn16_minc = tuning_knob("wn16_minc", &g_wn16_minc, 1);

// blocks per layer of a grouped nine-tap launch: layers with all 3 x 3 channel tiles in use come in numbers (the eight
// 45 -> 45 convolutions of the refinement stack), the narrower ones are one or two per launch and need more blocks each
static int g_wn16_group_target = tuning_knob("wn16_group_target", &g_wn16_group_target, 128), g_wn16_group_target_few = tuning_knob("wn16_group_few", &g_wn16_group_target_few, 256);
static int n16_group_target(int Cout, int C) { return (Cout > 32 && C > 32) ? g_wn16_group_target : g_wn16_group_target_few; }

static NPlan make_nplan(int N, int H, int W, int Cout, int C, int ld_x, int target = 0) {
    NPlan p;
    p.use = g_wgrad_n16 && C <= 64 && Cout <= 64 && C >= g_wn16_minc && H % 8 == 0 && W % 8 == 0 && ld_x % 4 == 0 &&
            ld_x >= round_up(C, 4) && (long)N * H * W < (1L << 31);
    if (!p.use) return p;
    p.gm = ceil_div(Cout, 48);
    p.gn = ceil_div(C, 48);
    p.tiles_w = W / 8;
    p.tiles_per_img = (H / 8) * p.tiles_w;
    p.total_tiles = (long)N * p.tiles_per_img;
    const long base = (long)p.gm * p.gn;
    long splits = ((target > 0 ? target : g_wn16_target) + base - 1) / base;
    if (splits > p.total_tiles / g_wn16_mintiles) splits = p.total_tiles / g_wn16_mintiles;
    if (splits < 1) splits = 1;
    p.tiles_per_split = (p.total_tiles + splits - 1) / splits;
    p.splits = (int)((p.total_tiles + p.tiles_per_split - 1) / p.tiles_per_split);
    return p;
}

// tap-major wgrad plan
struct TPlan {
    bool use;
    int bm, bn, gm, gn, splits;
    int groups, per_group;       // two-stage split reduction when splits > 12 (groups of ~8 splits), else groups = 0
    long pix_per_split;
};
static int g_wgrad_tap = tuning_knob("wgrad_tap", &g_wgrad_tap, 1), g_wtap_target = tuning_knob("wtap_target", &g_wtap_target, 768),
           g_wtap_minsteps = tuning_knob("wtap_minsteps", &g_wtap_minsteps, 8), g_wtap_minc = tuning_knob("wtap_minc", &g_wtap_minc, 16),
           g_wtap_bm_max = tuning_knob("wtap_bm_max", &g_wtap_bm_max, 128);       // 64: no 128-row tiles (A/B runs)

static TPlan make_tplan(long M, int Cout, int C, int ntaps, int ld_x) {
    TPlan p;
    // measured on the MI355X (profiles/README.md): the tap-major form wins (+10..35 %) once one of the channel counts
    // exceeds a 64-wide tile; narrow high-resolution layers (45 -> 45, 35 -> 10, 44 -> 64) keep the halo / gather
    // kernels, whose tiles span several taps of the same channels (higher arithmetic intensity per staged byte)
    p.use = g_wgrad_tap && C >= g_wtap_minc && (g_wgrad_tap > 1 || C > 64 || Cout > 64) && ntaps <= 16 &&
            ld_x % 4 == 0 && ld_x >= round_up(C, 4) && M < (1L << 31);
    if (!p.use) return p;
    p.bm = Cout > 64 ? 128 : (Cout > 32 ? 64 : 32);
    if (p.bm > g_wtap_bm_max) p.bm = g_wtap_bm_max;
    p.bn = (C > 64 || p.bm <= 64) ? 128 : 64;   // tiles in use: 128x128, 128x64, 64x128, 32x128
    p.gm = ceil_div(Cout, p.bm);
    p.gn = ceil_div(C, p.bn);
    const long tiles = (long)p.gm * p.gn * ntaps;
    const long steps = (M + BK - 1) / BK;
    long splits = (g_wtap_target + tiles - 1) / tiles;
    const long max_splits = steps / g_wtap_minsteps;
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    const long steps_per = (steps + splits - 1) / splits;
    p.pix_per_split = steps_per * BK;
    p.splits = (int)((steps + steps_per - 1) / steps_per);
    p.groups = p.splits > 12 ? ceil_div(p.splits, 8) : 0;     // == split_groups(splits)
    p.per_group = 8;
    return p;
}

// n / d for n < 2^31 as (mulhi(n, mul) >> sh), or (n >> sh) when mul == 0 (d a power of two)
static void fast_div_consts(unsigned d, unsigned* mul, unsigned* sh) {
    unsigned s = 0;
    while ((1u << s) < d) ++s;
    if ((1u << s) == d) {
        *mul = 0;
        *sh = s;
        return;
    }
    const unsigned long long num = 1ull << (31 + s);
    *mul = (unsigned)((num + d - 1) / d);
    *sh = s - 1;
}

static inline int grid_for(long total, int cap = 4096) {
    long b = (total + 255) / 256;
    if (b < 1) b = 1;
    return (int)(b < cap ? b : cap);
}

// groups of the two-stage split reduction (0: single stage)
static inline int split_groups(int splits) { return splits > 12 ? ceil_div(splits, 8) : 0; }

// sum `splits` partials of n floats each (at ws) into dst rows: optional first stage over groups of 8 splits
static void launch_wgrad_reduce(float* ws, int splits, int Cout, int NT, float* dst, long ld_out, hipStream_t s) {
    const long n = (long)Cout * NT;
    const float* src = ws;
    int nsum = splits;
    const int groups = split_groups(splits);
    if (groups) {
        float* part2 = ws + (size_t)splits * n;
        hipLaunchKernelGGL(conv3x3_wgrad_group_sum_kernel, dim3(grid_for(n, 1024), groups), dim3(256), 0, s, ws, n, splits, 8,
                           part2);
        src = part2;
        nsum = groups;
    }
    hipLaunchKernelGGL(conv3x3_wgrad_reduce_kernel, dim3(grid_for(n * 4, 8192)), dim3(256), 0, s, src, nsum, Cout, NT, dst,
                       ld_out);
}

// ---- grouped tap-major weight gradients: plan / build / launch --------------------------------------------------------
static int g_wgroup_chunk = tuning_knob("wgroup_chunk", &g_wgroup_chunk, 256);
static int g_wgroup_long = tuning_knob("wgroup_long", &g_wgroup_long, 1), g_wgroup_long_from = tuning_knob("wgroup_long_from", &g_wgroup_long_from, 128);
static int g_up_subpixel = tuning_knob("up_subpixel", &g_up_subpixel, 1);     // weight gradients of up-sampled convolutions: sub-pixel form

// the sub-pixel tap-major plan of an up-sampled 3x3 layer (flags: UPSAMPLED | CLEAN_PADS), or use = false
static TPlan make_up_tplan(int N, int Ho, int Wo, int Cout, int C, int kh, int kw, int pad, int ld_x, int flags) {
    TPlan tp;
    tp.use = false;
    if (!(g_up_subpixel && (flags & MNK_CONV_UPSAMPLED) && (flags & MNK_CONV_CLEAN_PADS) && kh == 3 && kw == 3 && pad == 1 &&
          Ho % 2 == 0 && Wo % 2 == 0))
        return tp;
    TPlan base = make_tplan((long)N * Ho * Wo, Cout, C, 9, ld_x);
    if (!base.use) return tp;                       // narrow layers keep their own kernels
    return make_tplan((long)N * (Ho / 2) * (Wo / 2), Cout, C, 16, ld_x);
}     // pixels per block of a grouped launch (multiple of 16)

// pixel splits of one job of a grouped launch: chunks of ~g_wgroup_chunk pixels, at least 8 K steps each
static void grouped_split(long M, int* splits, long* pix_per_split) {
    const long steps = (M + BK - 1) / BK;
    long chunk = g_wgroup_chunk;
    if (g_wgroup_long > 1 && M / chunk >= g_wgroup_long_from) chunk *= g_wgroup_long;     // long layers: longer chunks, fewer partials
    long sp = (M + chunk - 1) / chunk;
    const long max_sp = steps / 8;
    if (sp > max_sp) sp = max_sp;
    if (sp < 1) sp = 1;
    const long steps_per = (steps + sp - 1) / sp;
    *pix_per_split = steps_per * BK;
    *splits = (int)((steps + steps_per - 1) / steps_per);
}

// small maps: the tap-major kernel's K runs over the (pixel, tap) pairs inside the source only (WgradTapArgs::compact);
// wtap_compact = the largest H * W (of the dy geometry; the low resolution for the sub-pixel form) that takes this form, 0: off
static int g_wtap_compact = tuning_knob("wtap_compact", &g_wtap_compact, 64);
static bool tap_compact_ok(int H, int W, long M, int kh, int kw, int pad, int ups, int subpix) {
    if (g_wtap_compact <= 0 || (long)H * W > g_wtap_compact || M >= (1L << 20)) return false;
    return subpix || (!ups && kh == 3 && kw == 3 && pad == 1);
}
// (pixel, tap) pairs a compact job multiplies: 3x3 pad 1: (3H - 2)(3W - 2) per frame; sub-pixel form (offsets {-1, 0} / {0, +1}
// per phase and axis): (4H - 2)(4W - 2) per frame
static double tap_compact_pairs(long frames, int H, int W, int subpix) {
    return subpix ? (double)frames * (4.0 * H - 2.0) * (4.0 * W - 2.0) : (double)frames * (3.0 * H - 2.0) * (3.0 * W - 2.0);
}

// variant id of a tap-major job: 4 * tile + mode; tile 0: 128x128, 1: 128x64, 2: 64x128, 3: 32x128; mode 3: sub-pixel form
static int tap_tile_id(const TPlan& tp) {
    if (tp.bm == 128 && tp.bn == 128) return 0;
    if (tp.bm == 128) return 1;
    if (tp.bm == 64) return 2;
    return 3;
}

// loader mode + walk constants of the tap-major kernel for one pixel range length (shared by the single-layer entry)
static int tap_mode(WgradTapArgs& g, int N, int H, int W, int Hi, int Wi, int kh, int kw, int pad, int ups, int clean, int ld_x,
                    int ld_dy, long pix_per_split, int subpix = 0) {
    if (subpix) {
        g.sw = g.sh = g.sn = 0;
        return 3;
    }
    if (tap_compact_ok(H, W, (long)N * H * W, kh, kw, pad, ups, 0)) {      // compact K lives in the generic loader
        g.sw = g.sh = g.sn = 0;
        return 0;
    }
    const long span_a = pix_per_split * (long)ld_dy * 4, span_b = (pix_per_split + 2L * W + 2 * BK) * ld_x * 4;
    bool walk = true;          // can a 16-pixel step be walked as columns / rows / frames with single wraps?
    g.sw = BK;
    g.sh = g.sn = 0;
    if (W < BK) {
        g.sw = 0;
        const int r = BK / W;
        if (BK % W != 0)
            walk = false;
        else if (r < H)
            g.sh = r;
        else if (r % H == 0)
            g.sn = r / H;
        else
            walk = false;
    }
    return (g_fast_loader && clean && kh == 3 && kw == 3 && pad == 1 && walk && span_a < (1L << 29) && span_b < (1L << 29) &&
            (!ups || (long)N * (Hi / 2) * (Wi / 2) * ld_x * 4 < (1L << 29)))
               ? (ups ? 2 : 1) : 0;
}

struct GroupedHeader {
    int magic, n, nvariants, reserved;
    int first[32], count[32], blocks[32];     // per variant: first record, records, blocks (records are sorted by variant)
};

}  // namespace

extern "C" {

// ---- general K x K entry points (3x3 pad 1 of the hot path; 4x4 pad 0 of the discriminator, its data gradient = 4x4 pad 3)
size_t mnk_conv2d_packed_floats(int Cout, int C0, int C1, int ntaps) {
    if (Cout <= 0 || C0 <= 0 || C1 < 0 || ntaps <= 0) return 0;
    return (size_t)Cout * ntaps * (round_up(C0, 16) + (C1 > 0 ? round_up(C1, 16) : 0));
}

int mnk_conv2d_pack_fwd(const float* w, float* wp, int Cout, int C0, int C1, int ntaps, void* stream) {
    MNK_REQUIRE(w && wp && Cout > 0 && C0 > 0 && C1 >= 0 && ntaps > 0);
    hipStream_t s = (hipStream_t)stream;
    const int C0p = round_up(C0, 16), C1p = C1 > 0 ? round_up(C1, 16) : 0;
    const long total = (long)Cout * ntaps * (C0p + C1p);
    ProfScope prof(K_CONV_REDUCE, s, (double)total * 8);
    hipLaunchKernelGGL(pack_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, s, w, wp, Cout, C0, C1, C0p, C1p, ntaps);
    MNK_LAUNCH_CHECK();
    return MNK_OK;
}

int mnk_conv2d_pack_dgrad(const float* w, float* wp, int Cout, int Cin_total, int c_start, int c_count, int ntaps,
                          void* stream) {
    MNK_REQUIRE(w && wp && Cout > 0 && Cin_total > 0 && c_start >= 0 && c_count > 0 && c_start + c_count <= Cin_total);
    MNK_REQUIRE(ntaps > 0);
    hipStream_t s = (hipStream_t)stream;
    const int chunks = round_up(Cout, 16) / 16;
    const long total = (long)c_count * chunks * ntaps * 16;
    ProfScope prof(K_CONV_REDUCE, s, (double)total * 8);
    hipLaunchKernelGGL(pack_dgrad_kernel, dim3(grid_for(total)), dim3(256), 0, s, w, wp, Cout, Cin_total, c_start,
                       c_count, chunks, ntaps);
    MNK_LAUNCH_CHECK();
    return MNK_OK;
}

int mnk_conv2d_pack_all(const float* w, float* wp_fwd, float* wp_d0, float* wp_d1, int Cout, int C0, int C1, int ntaps,
                        void* stream) {
    MNK_REQUIRE(w && wp_fwd && Cout > 0 && C0 > 0 && C1 >= 0 && ntaps > 0 && (!wp_d1 || C1 > 0));
    hipStream_t s = (hipStream_t)stream;
    const int C0p = round_up(C0, 16), C1p = C1 > 0 ? round_up(C1, 16) : 0;
    const long dper = (long)round_up(Cout, 16) * ntaps;
    const long total = (long)Cout * ntaps * (C0p + C1p) + (wp_d0 ? C0 * dper : 0) + (wp_d1 ? C1 * dper : 0);
    MNK_REQUIRE(total < (1L << 31) && ((size_t)wp_fwd % 16) == 0 && ((size_t)wp_d0 % 16) == 0 && ((size_t)wp_d1 % 16) == 0);
    ProfScope prof(K_CONV_REDUCE, s, (double)total * 8);
    MNK_REQUIRE(ntaps <= 16);
    const dim3 grid((C0p + C1p) / 16, ceil_div(Cout, 16));
    if (ntaps == 9)
        hipLaunchKernelGGL(pack_all_kernel<9>, grid, dim3(256), 0, s, w, wp_fwd, wp_d0, wp_d1, Cout, C0, C1, C0p, C1p, ntaps);
    else
        hipLaunchKernelGGL(pack_all_kernel<0>, grid, dim3(256), 0, s, w, wp_fwd, wp_d0, wp_d1, Cout, C0, C1, C0p, C1p, ntaps);
    MNK_LAUNCH_CHECK();
    return MNK_OK;
}

int mnk_conv3x3_pack_multi(const MnkPackDesc* descs_device, int n, int total_tiles, void* stream) {
    MNK_REQUIRE(descs_device && n > 0 && total_tiles > 0);
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(K_CONV_REDUCE, s, (double)total_tiles * 16 * 16 * 9 * 12);
    hipLaunchKernelGGL(pack_multi_kernel, dim3(total_tiles), dim3(256), 0, s, descs_device, n);
    MNK_LAUNCH_CHECK();
    return MNK_OK;
}

size_t mnk_conv2d_workspace_floats(int N, int Ho, int Wo, int C0, int C1, int Cout, int ntaps) {
    if (N <= 0 || Ho <= 0 || Wo <= 0 || C0 <= 0 || C1 < 0 || Cout <= 0 || ntaps <= 0) return 0;
    const int chunks = (round_up(C0, 16) + (C1 > 0 ? round_up(C1, 16) : 0)) / 16;
    Plan p = make_plan((long)N * Ho * Wo, Cout, chunks, ntaps);
    return p.splits > 1 ? (size_t)p.splits * N * Ho * Wo * p.ldw : 0;
}

size_t mnk_conv2d_stats_floats(int N, int Ho, int Wo, int C0, int C1, int Cout, int ntaps) {
    if (N <= 0 || Ho <= 0 || Wo <= 0 || C0 <= 0 || C1 < 0 || Cout <= 0 || ntaps <= 0) return 0;
    const int chunks = (round_up(C0, 16) + (C1 > 0 ? round_up(C1, 16) : 0)) / 16;
    Plan p = make_plan((long)N * Ho * Wo, Cout, chunks, ntaps);
    if (p.splits > 1)
        return g_splitk_stats ? (size_t)make_rsmap((long)N * Ho * Wo, round_up(Cout, 4)).row_blocks * 2 * round_up(Cout, 4) : 0;
    return (size_t)p.gm * 2 * round_up(Cout, 4);
}

}  // extern "C"

// general form behind mnk_conv2d_fwd / mnk_conv3x3_up_fwd / mnk_conv3x3_up_dgrad:
//   phases == 1: Ho x Wo outputs, input pixel of output (h, w), tap (ky, kx) = (h * stride + ky - pad, w * stride + kx - pad)
//   phases == 4: the sub-pixel form of [nearest x2 -> 3x3 / pad 1] (ConvArgs): kh = kw = 2, pad = 1, (Hi, Wi) = (Ho, Wo) =
//                the LOW resolution; y is the (N, 2 Ho, 2 Wo) tensor; wp = four per-phase packs
static int conv2d_fwd_impl(const float* x0, int ld0, int C0, const float* x1, int ld1, int C1, int flags, int Hi, int Wi, int kh,
                           int kw, int pad, int stride, int phases, const float* wp, const float* bias, const float* residual,
                           int ld_res, float* y, int ld_y, int N, int Ho, int Wo, int Cout, float* ws, size_t ws_floats,
                           float* stats_partial, void* stream, const BnBwdSrc* bnb = nullptr) {
    MNK_REQUIRE(flags >= 0 && flags <= 7);
    const int ups = flags & MNK_CONV_UPSAMPLED, clean = (flags & MNK_CONV_CLEAN_PADS) ? 1 : 0;
    // MNK_CONV_DEFER_SPLITK: a split-K launch leaves its partials in `ws` ([split][phase][M][ldw], bias not added) and the
    // caller sums them (mnk_bn_small_fwd does, together with the normalisation that follows)
    const bool defer_splitk = (flags & MNK_CONV_DEFER_SPLITK) != 0;
    MNK_REQUIRE(x0 && wp && y && N > 0 && Ho > 0 && Wo > 0 && Cout > 0 && C0 > 0 && C1 >= 0);
    MNK_REQUIRE(kh > 0 && kw > 0 && pad >= 0 && Hi > 0 && Wi > 0 && stride >= 1 && (phases == 1 || phases == 4));
    if (phases == 4)
        MNK_REQUIRE(kh == 2 && kw == 2 && pad == 1 && stride == 1 && Ho == Hi && Wo == Wi && !ups && clean && !residual);
    else
        MNK_REQUIRE(Ho == (Hi + 2 * pad - kh) / stride + 1 && Wo == (Wi + 2 * pad - kw) / stride + 1);
    MNK_REQUIRE(stride == 1 || (clean && !ups));
    MNK_REQUIRE(ld0 % 4 == 0 && ld0 >= C0 && ld_y % 4 == 0 && ld_y >= Cout && ld_y <= round_up(Cout, 16));
    MNK_REQUIRE(C1 == 0 || (x1 && ld1 % 4 == 0 && ld1 >= C1));
    MNK_REQUIRE(!ups || (Hi % 2 == 0 && Wi % 2 == 0));
    MNK_REQUIRE(!residual || (ld_res >= Cout));
    const int ntaps = kh * kw;
    ConvArgs a;
    a.x0 = x0;
    a.x1 = x1;
    a.ld0 = ld0;
    a.ld1 = ld1;
    a.C0 = C0;
    a.C1 = C1;
    a.C0p = round_up(C0, 16);
    a.C1p = C1 > 0 ? round_up(C1, 16) : 0;
    a.ups = ups;
    a.clean = clean;
    a.wp = wp;
    a.bias = bias;
    a.residual = residual;
    a.ld_res = ld_res;
    a.y = y;
    a.ld_y = ld_y;
    a.N = N;
    a.H = Ho;
    a.W = Wo;
    a.Hi = Hi;
    a.Wi = Wi;
    a.ntaps = ntaps;
    a.kw = kw;
    a.pad = pad;
    a.Cout = Cout;
    a.M = (long)N * Ho * Wo;
    MNK_REQUIRE(a.M * phases < (1L << 31) && a.M * phases * round_up(Cout, 16) < (1L << 32) &&
                (!residual || a.M * ld_res < (1L << 32)));
    fast_div_consts((unsigned)Wo, &a.mulW, &a.shW);
    fast_div_consts((unsigned)Ho, &a.mulH, &a.shH);
    a.chunks = (a.C0p + a.C1p) / 16;
    a.stride = stride;
    a.pad_x = -1;
    Plan p = make_plan(a.M, Cout, a.chunks, ntaps, phases);
    a.phases = phases;
    a.tiles_per_phase = p.gm;
    a.phase_wstride = (long)Cout * ntaps * (a.C0p + a.C1p);
    a.ksteps = p.ksteps;
    a.ksteps_per_split = p.ksteps_per_split;
    a.splits = p.splits;
    a.ws = ws;
    a.ldw = p.ldw;
    a.stats = stats_partial;
    a.bnb = bnb ? *bnb : BnBwdSrc{};
    MNK_REQUIRE(!bnb || (stats_partial && bnb->y && bnb->mean && bnb->invstd && bnb->scale && bnb->beta && bnb->ld >= Cout &&
                         phases == 1 && !defer_splitk));
    a.xcd = g_xcd_remap;
    MNK_REQUIRE(!stats_partial || (ld_y == round_up(Cout, 4) && (p.splits == 1 || (g_splitk_stats && !defer_splitk))));
    if (p.splits > 1 && (!ws || ws_floats < (size_t)p.splits * phases * a.M * p.ldw)) {
        set_error("mnk_conv2d_fwd: workspace too small (%zu < %zu floats)", ws_floats, (size_t)p.splits * phases * a.M * p.ldw);
        return MNK_EWORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(p.gm * phases, p.gn, p.splits);
    {
        // algorithmic FLOPs of the convolution this launch stands for: the sub-pixel form computes a 3x3 convolution on
        // 4 M output pixels; a stride-2 4x4 data gradient stands for the 3x3 data gradient at 4 M pixels
        const double alg = phases == 4 ? 2.0 * 4.0 * (double)a.M * Cout * 9.0 * (C0 + C1)
                                       : (stride == 2 ? 2.0 * 4.0 * (double)a.M * Cout * 9.0 * (C0 + C1)
                                                      : 2.0 * (double)a.M * Cout * (double)ntaps * (C0 + C1));
        // what the launch issues: the sub-pixel forms run 4 (forward) / 16 at a quarter of the pixels (data gradient) taps
        ProfScope prof(K_CONV_FWD, s, alg, 2.0 * (double)a.M * phases * Cout * (double)ntaps * (C0 + C1));
        // loader: the 3x3 / pad 1 fast form when the caller vouches for clean pad channels and a block's pixel span
        // fits the 2^30-byte buffer window (always, short of ~2 M-float pixel rows)
        const long span = ((long)BK * 8 + 3L * (ups ? Wi / 2 : Wi) + 8) * (ld0 > ld1 ? ld0 : ld1) * 4;
        int mode = (g_fast_loader && a.clean && kh == 3 && kw == 3 && pad == 1 && stride == 1 && phases == 1 &&
                    span < (1L << 29) && (size_t)x0 % 16 == 0 && (!x1 || (size_t)x1 % 16 == 0)) ? (ups ? 2 : 1) : 0;
        // any other K x K / pad (the discriminator's 4x4 convolutions and their pad-3 data gradients): ActLoaderK.  A
        // block's 128 output pixels span at most 128 * kh * kw + (kh + 3) * Wi input pixels (a 1x1 output per frame
        // advances a whole kh x kw input frame per output pixel; plus the rows of the taps)
        const long span_k = (128L * kh * kw * stride * stride + (long)(kh + 3) * Wi) * (ld0 > ld1 ? ld0 : ld1) * 4;
        MNK_REQUIRE((stride == 1 && phases == 1) ||
                    (g_fast_loader && g_kxk_fast && ntaps <= 32 && span_k < (1L << 29) && (size_t)x0 % 16 == 0 &&
                     (!x1 || (size_t)x1 % 16 == 0)));          // strided / sub-pixel forms exist for the K x K buffer loader only
        if (mode == 0 && g_fast_loader && g_kxk_fast && a.clean && !ups && ntaps <= 32 && pad >= 0 && pad < kh && pad < kw &&
            span_k < (1L << 29) && (size_t)x0 % 16 == 0 && (!x1 || (size_t)x1 % 16 == 0))
            mode = 3;
        // the roofline kernel is timed by its own begin / end stamps (bench.py `roofline`, agrees with rocprofv3)
        hipEvent_t ev0, ev1;
        const bool timed = prof.kernel_events(&ev0, &ev1);
        // (padding a block's LDS request so that exactly ceil(blocks / CUs) blocks fit a CU was built and measured in round 4: the
        // dispatcher already puts 1024 blocks on 256 CUs four by four -- tools/microbench/launch_gap.hip (e) -- and the step did
        // not move, 10.33 vs 10.32 ms: removed.  profiles/r04_knob_ab_log.txt)
        const unsigned dyn = 0;
#define MNK_IGEMM_MODE(KERNEL, MODE, ...)                                                                       \
    do {                                                                                                        \
        if (timed) hipExtLaunchKernelGGL((KERNEL<__VA_ARGS__, MODE>), grid, dim3(256), dyn, s, ev0, ev1, 0, a); \
        else hipLaunchKernelGGL((KERNEL<__VA_ARGS__, MODE>), grid, dim3(256), dyn, s, a);                       \
    } while (0)
#define MNK_IGEMM_MODE_H(KERNEL, MODE, ...)                                                                        \
    do {                                                                                                           \
        if (timed) hipExtLaunchKernelGGL((KERNEL<__VA_ARGS__, MODE, 1>), grid, dim3(256), dyn, s, ev0, ev1, 0, a); \
        else hipLaunchKernelGGL((KERNEL<__VA_ARGS__, MODE, 1>), grid, dim3(256), dyn, s, a);                       \
    } while (0)
#define MNK_IGEMM_H(KERNEL, ...)                                      \
    do {                                                              \
        if (mode == 1) MNK_IGEMM_MODE_H(KERNEL, 1, __VA_ARGS__);      \
        else if (mode == 2) MNK_IGEMM_MODE_H(KERNEL, 2, __VA_ARGS__); \
        else if (mode == 3) MNK_IGEMM_MODE_H(KERNEL, 3, __VA_ARGS__); \
        else MNK_IGEMM_MODE_H(KERNEL, 0, __VA_ARGS__);                \
    } while (0)
#define MNK_IGEMM(KERNEL, ...)                                      \
    do {                                                            \
        if (mode == 1) MNK_IGEMM_MODE(KERNEL, 1, __VA_ARGS__);      \
        else if (mode == 2) MNK_IGEMM_MODE(KERNEL, 2, __VA_ARGS__); \
        else if (mode == 3) MNK_IGEMM_MODE(KERNEL, 3, __VA_ARGS__); \
        else MNK_IGEMM_MODE(KERNEL, 0, __VA_ARGS__);                \
    } while (0)
        if (p.bn == 16 && g_gemm16_bf16x3)
            MNK_IGEMM_H(conv3x3_igemm16_kernel, 16);
        else if (p.bn == 48 && g_gemm16_bf16x3)
            MNK_IGEMM_H(conv3x3_igemm16_kernel, 48);
        else if (p.bn == 16)
            MNK_IGEMM(conv3x3_igemm16_kernel, 16);
        else if (p.bn == 48)
            MNK_IGEMM(conv3x3_igemm16_kernel, 48);
        else if (g_gemm_bf16x3 && p.bn == 128 && p.bm == 128)
            MNK_IGEMM_H(conv3x3_igemm_kernel, 128, 128, 2, 2);
        else if (g_gemm_bf16x3 && p.bn == 128)
            MNK_IGEMM_H(conv3x3_igemm_kernel, 64, 128, 1, 4);
        else if (g_gemm_bf16x3 && p.bn == 64 && p.bm == 128)
            MNK_IGEMM_H(conv3x3_igemm_kernel, 128, 64, 2, 2);
        else if (g_gemm_bf16x3 && p.bn == 64)
            MNK_IGEMM_H(conv3x3_igemm_kernel, 64, 64, 2, 2);
        else if (g_gemm_bf16x3)
            MNK_IGEMM_H(conv3x3_igemm_kernel, 128, 32, 4, 1);
        else if (p.bn == 128 && p.bm == 128)
            MNK_IGEMM(conv3x3_igemm_kernel, 128, 128, 2, 2);
        else if (p.bn == 128)
            MNK_IGEMM(conv3x3_igemm_kernel, 64, 128, 1, 4);
        else if (p.bn == 64 && p.bm == 128)
            MNK_IGEMM(conv3x3_igemm_kernel, 128, 64, 2, 2);
        else if (p.bn == 64)
            MNK_IGEMM(conv3x3_igemm_kernel, 64, 64, 2, 2);
        else
            MNK_IGEMM(conv3x3_igemm_kernel, 128, 32, 4, 1);
#undef MNK_IGEMM
#undef MNK_IGEMM_MODE
#undef MNK_IGEMM_H
#undef MNK_IGEMM_MODE_H
    }
    if (p.splits > 1 && !defer_splitk) {
        ProfScope prof(K_CONV_REDUCE, s, (double)p.splits * a.M * p.ldw * 4);
        if (stats_partial) {
            const RSMap m = make_rsmap(a.M * phases, ld_y);
            hipLaunchKernelGGL(conv3x3_splitk_reduce_stats_kernel<true>, dim3(m.col_tiles, m.row_blocks), dim3(256), 0, s, ws,
                               p.splits, a.M, p.ldw, bias, residual, ld_res, y, ld_y, Cout, phases, a.H, a.W, m.tx, m.ty,
                               m.rows_per_block, stats_partial, a.bnb);
        } else if (g_reduce_v4 && (size_t)y % 16 == 0 && (size_t)ws % 16 == 0) {
            const RSMap m = make_rsmap(a.M * phases, ld_y);
            hipLaunchKernelGGL(conv3x3_splitk_reduce_stats_kernel<false>, dim3(m.col_tiles, m.row_blocks), dim3(256), 0, s, ws,
                               p.splits, a.M, p.ldw, bias, residual, ld_res, y, ld_y, Cout, phases, a.H, a.W, m.tx, m.ty,
                               m.rows_per_block, (float*)nullptr);
        } else
            hipLaunchKernelGGL(conv3x3_splitk_reduce_kernel, dim3(grid_for(a.M * phases * ld_y * 4, 8192)), dim3(256), 0, s, ws,
                               p.splits, a.M, p.ldw, bias, residual, ld_res, y, ld_y, Cout, phases, a.H, a.W);
    }
    MNK_LAUNCH_CHECK();
    return MNK_OK;
}

extern "C" {

int mnk_conv2d_fwd(const float* x0, int ld0, int C0, const float* x1, int ld1, int C1, int flags, int Hi, int Wi, int kh,
                   int kw, int pad, const float* wp, const float* bias, const float* residual, int ld_res, float* y,
                   int ld_y, int N, int Ho, int Wo, int Cout, float* ws, size_t ws_floats, float* stats_partial,
                   void* stream) {
    return conv2d_fwd_impl(x0, ld0, C0, x1, ld1, C1, flags, Hi, Wi, kh, kw, pad, 1, 1, wp, bias, residual, ld_res, y, ld_y, N,
                           Ho, Wo, Cout, ws, ws_floats, stats_partial, stream);
}

size_t mnk_conv2d_wgrad_workspace_floats(int N, int Ho, int Wo, int C, int Cout, int kh, int kw, int pad) {
    if (N <= 0 || Ho <= 0 || Wo <= 0 || C <= 0 || Cout <= 0 || kh <= 0 || kw <= 0) return 0;
    const int ntaps = kh * kw;
    {   // the caller's ld_x is not known here: size for the tap-major form whenever the shape allows it
        TPlan tp = make_tplan((long)N * Ho * Wo, Cout, C, ntaps, round_up(C, 4));
        if (tp.use) return (size_t)(tp.splits + tp.groups) * ntaps * Cout * C;
    }
    if (kh == 3 && kw == 3 && pad == 1) {
        NPlan np = make_nplan(N, Ho, Wo, Cout, C, round_up(C, 4));
        HPlan hp = make_hplan(N, Ho, Wo, Cout, C);
        size_t need = 0;        // the caller's ld_x / alignment may still demote the n16 form: size for both
        if (np.use && np.splits > 1) need = (size_t)(np.splits + split_groups(np.splits)) * Cout * 9 * C;
        if (hp.use) {
            const size_t nh = hp.splits > 1 ? (size_t)(hp.splits + split_groups(hp.splits)) * Cout * 9 * C : 0;
            return nh > need ? nh : need;
        }
        if (np.use) {
            WPlan p = make_wplan((long)N * Ho * Wo, Cout, C, ntaps);
            const size_t ng = p.splits > 1 ? (size_t)(p.splits + split_groups(p.splits)) * Cout * ntaps * C : 0;
            return ng > need ? ng : need;
        }
    }
    WPlan p = make_wplan((long)N * Ho * Wo, Cout, C, ntaps);
    return p.splits > 1 ? (size_t)(p.splits + split_groups(p.splits)) * Cout * ntaps * C : 0;
}

size_t mnk_conv3x3_up_wgrad_workspace_floats(int N, int Ho, int Wo, int C, int Cout) {
    if (N <= 0 || Ho <= 0 || Wo <= 0 || C <= 0 || Cout <= 0) return 0;
    TPlan up = make_up_tplan(N, Ho, Wo, Cout, C, 3, 3, 1, round_up(C, 4), MNK_CONV_UPSAMPLED | MNK_CONV_CLEAN_PADS);
    const size_t a = up.use ? (size_t)(up.splits + up.groups) * 16 * Cout * C : 0;
    const size_t b = mnk_conv2d_wgrad_workspace_floats(N, Ho, Wo, C, Cout, 3, 3, 1);
    return a > b ? a : b;
}

int mnk_conv2d_wgrad(const float* x, int ld_x, int C, int flags, int Hi, int Wi, int kh, int kw, int pad, const float* dy,
                     int ld_dy, int Cout, float* dw, int Cin_total, int c_start, int N, int Ho, int Wo, float* ws,
                     size_t ws_floats, void* stream) {
    MNK_REQUIRE(flags >= 0 && flags <= 7);
    const int ups = flags & MNK_CONV_UPSAMPLED, clean = (flags & MNK_CONV_CLEAN_PADS) ? 1 : 0;
    // MNK_WGRAD_DEFER: leave the split partials in `ws` (layout / size: mnk_conv2d_wgrad_plan) and skip the reduction --
    // the caller reduces the partials of many layers in one launch (mnk_wgrad_reduce_multi)
    const bool defer = (flags & MNK_WGRAD_DEFER) != 0;
    MNK_REQUIRE(x && dy && dw && N > 0 && Ho > 0 && Wo > 0 && C > 0 && Cout > 0 && kh > 0 && kw > 0 && pad >= 0);
    MNK_REQUIRE(!defer || ((size_t)x % 16 == 0 && (size_t)dy % 16 == 0));      // the plan query assumes aligned operands
    MNK_REQUIRE(Ho == Hi + 2 * pad - kh + 1 && Wo == Wi + 2 * pad - kw + 1);
    MNK_REQUIRE(ld_x >= C && ld_dy % 4 == 0 && ld_dy >= Cout);
    MNK_REQUIRE(c_start >= 0 && c_start + C <= Cin_total && (!ups || (Hi % 2 == 0 && Wi % 2 == 0)));
    const int ntaps = kh * kw;
    const int H = Ho, W = Wo;
    {   // up-sampled 3x3 layer with clean sources: the sub-pixel form (16 pseudo taps over the LOW-resolution pixels)
        TPlan up = make_up_tplan(N, Ho, Wo, Cout, C, kh, kw, pad, ld_x, flags);
        if (up.use && (size_t)x % 16 == 0 && (size_t)dy % 16 == 0) {
            const int Hl = Ho / 2, Wl = Wo / 2;
            const size_t need = (size_t)(up.splits + (defer ? 0 : up.groups)) * 16 * Cout * C;
            if (!ws || ws_floats < need) {
                set_error("mnk_conv2d_wgrad: workspace too small (%zu < %zu floats)", ws_floats, need);
                return MNK_EWORKSPACE;
            }
            WgradTapArgs g;
            g.x = x, g.ld_x = ld_x, g.C = C, g.ups = 0, g.dy = dy, g.ld_dy = ld_dy, g.Cout = Cout;
            g.H = Hl, g.W = Wl, g.Hi = Hl, g.Wi = Wl, g.ntaps = 16, g.kw = 4, g.pad = 0;
            g.M = (long)N * Hl * Wl;
            g.pix_per_split = up.pix_per_split;
            g.gn = up.gn;
            g.part = ws;
            g.xcd = g_xcd_remap;
            g.clean = 1;
            g.sw = g.sh = g.sn = 0;
            g.compact = tap_compact_ok(Hl, Wl, g.M, 3, 3, 1, 0, 1) ? 1 : 0;
            g.nsplits = up.splits;
            fast_div_consts((unsigned)Wl, &g.mulW, &g.shW);
            fast_div_consts((unsigned)Hl, &g.mulH, &g.shH);
            hipStream_t st = (hipStream_t)stream;
            dim3 grid(up.gm, up.gn * 16, up.splits);
            {
                ProfScope prof(K_CONV_WGRAD, st, 2.0 * (double)N * Ho * Wo * Cout * 9.0 * C,       // 16 pseudo taps at the low resolution
                               g.compact ? 2.0 * tap_compact_pairs(N, Hl, Wl, 1) * Cout * C : 2.0 * (double)g.M * Cout * 16.0 * C);
                if (g_wgrad_bf16x3) {
                    if (up.bm == 128 && up.bn == 128)
                        hipLaunchKernelGGL((conv3x3_wgrad_tap_h_kernel<128, 128, 2, 2, true>), grid, dim3(256), 0, st, g);
                    else if (up.bm == 128)
                        hipLaunchKernelGGL((conv3x3_wgrad_tap_h_kernel<128, 64, 2, 2, true>), grid, dim3(256), 0, st, g);
                    else if (up.bm == 64)
                        hipLaunchKernelGGL((conv3x3_wgrad_tap_h_kernel<64, 128, 1, 4, true>), grid, dim3(256), 0, st, g);
                    else
                        hipLaunchKernelGGL((conv3x3_wgrad_tap_h_kernel<32, 128, 1, 4, true>), grid, dim3(256), 0, st, g);
                } else if (up.bm == 128 && up.bn == 128)
                    hipLaunchKernelGGL((conv3x3_wgrad_tap_kernel<128, 128, 2, 2, 3>), grid, dim3(256), 0, st, g);
                else if (up.bm == 128)
                    hipLaunchKernelGGL((conv3x3_wgrad_tap_kernel<128, 64, 2, 2, 3>), grid, dim3(256), 0, st, g);
                else if (up.bm == 64)
                    hipLaunchKernelGGL((conv3x3_wgrad_tap_kernel<64, 128, 1, 4, 3>), grid, dim3(256), 0, st, g);
                else
                    hipLaunchKernelGGL((conv3x3_wgrad_tap_kernel<32, 128, 1, 4, 3>), grid, dim3(256), 0, st, g);
            }
            if (!defer) {
                ProfScope prof(K_CONV_REDUCE, st, (double)(up.splits + 1) * 16 * Cout * C * 4);
                const long n = (long)16 * Cout * C;
                const float* src = ws;
                int nsum = up.splits;
                if (up.groups) {
                    float* part2 = ws + (size_t)up.splits * n;
                    hipLaunchKernelGGL(conv3x3_wgrad_group_sum_kernel, dim3(grid_for(n, 1024), up.groups), dim3(256), 0, st, ws, n,
                                       up.splits, up.per_group, part2);
                    src = part2;
                    nsum = up.groups;
                }
                hipLaunchKernelGGL(conv3x3_wgrad_tap_reduce_kernel, dim3(ceil_div(C, 64), Cout), dim3(256), 0, st, src, nsum, 16,
                                   Cout, C, dw + (long)c_start * 9, (long)Cin_total * 9, 1);
            }
            MNK_LAUNCH_CHECK();
            return MNK_OK;
        }
    }
    TPlan tp = make_tplan((long)N * H * W, Cout, C, ntaps, ld_x);
    if (tp.use && ((size_t)x % 16 != 0 || (size_t)dy % 16 != 0)) tp.use = false;
    if (tp.use) {
        const size_t need = (size_t)(tp.splits + (defer ? 0 : tp.groups)) * ntaps * Cout * C;
        if (!ws || ws_floats < need) {
            set_error("mnk_conv2d_wgrad: workspace too small (%zu < %zu floats)", ws_floats, need);
            return MNK_EWORKSPACE;
        }
        WgradTapArgs g;
        g.x = x;
        g.ld_x = ld_x;
        g.C = C;
        g.ups = ups;
        g.dy = dy;
        g.ld_dy = ld_dy;
        g.Cout = Cout;
        g.H = H;
        g.W = W;
        g.Hi = Hi;
        g.Wi = Wi;
        g.ntaps = ntaps;
        g.kw = kw;
        g.pad = pad;
        g.M = (long)N * H * W;
        g.pix_per_split = tp.pix_per_split;
        g.gn = tp.gn;
        g.part = ws;
        g.xcd = g_xcd_remap;
        g.clean = clean;
        fast_div_consts((unsigned)W, &g.mulW, &g.shW);
        fast_div_consts((unsigned)H, &g.mulH, &g.shH);
        hipStream_t st = (hipStream_t)stream;
        dim3 grid(tp.gm, tp.gn * ntaps, tp.splits);
        g.compact = tap_compact_ok(H, W, g.M, kh, kw, pad, ups, 0) ? 1 : 0;
        g.nsplits = tp.splits;
        {
            ProfScope prof(K_CONV_WGRAD, st, 2.0 * (double)g.M * Cout * (double)ntaps * C,
                           g.compact ? 2.0 * tap_compact_pairs(N, H, W, 0) * Cout * C : -1.0);
            // fast loader: 3x3 pad 1, clean pads, rows of >= 16 pixels, split ranges inside the 2^30-byte buffer window
            const int mode = tap_mode(g, N, H, W, Hi, Wi, kh, kw, pad, ups, clean, ld_x, ld_dy, tp.pix_per_split);
#define MNK_WTAP(...)                                                                                            \
    do {                                                                                                         \
        if (mode == 1) hipLaunchKernelGGL((conv3x3_wgrad_tap_kernel<__VA_ARGS__, 1>), grid, dim3(256), 0, st, g);      \
        else if (mode == 2) hipLaunchKernelGGL((conv3x3_wgrad_tap_kernel<__VA_ARGS__, 2>), grid, dim3(256), 0, st, g); \
        else hipLaunchKernelGGL((conv3x3_wgrad_tap_kernel<__VA_ARGS__, 0>), grid, dim3(256), 0, st, g);                \
    } while (0)
            if (g_wgrad_bf16x3) {       // (one generic-loader form per tile: `mode` only selects among the fp32 kernels)
                if (tp.bm == 128 && tp.bn == 128)
                    hipLaunchKernelGGL((conv3x3_wgrad_tap_h_kernel<128, 128, 2, 2, false>), grid, dim3(256), 0, st, g);
                else if (tp.bm == 128)
                    hipLaunchKernelGGL((conv3x3_wgrad_tap_h_kernel<128, 64, 2, 2, false>), grid, dim3(256), 0, st, g);
                else if (tp.bm == 64)
                    hipLaunchKernelGGL((conv3x3_wgrad_tap_h_kernel<64, 128, 1, 4, false>), grid, dim3(256), 0, st, g);
                else
                    hipLaunchKernelGGL((conv3x3_wgrad_tap_h_kernel<32, 128, 1, 4, false>), grid, dim3(256), 0, st, g);
            } else if (tp.bm == 128 && tp.bn == 128)
                MNK_WTAP(128, 128, 2, 2);
            else if (tp.bm == 128)
                MNK_WTAP(128, 64, 2, 2);
            else if (tp.bm == 64)
                MNK_WTAP(64, 128, 1, 4);
            else
                MNK_WTAP(32, 128, 1, 4);
#undef MNK_WTAP
        }
        if (!defer) {
            ProfScope prof(K_CONV_REDUCE, st, (double)(tp.splits + 1) * ntaps * Cout * C * 4);
            const long n = (long)ntaps * Cout * C;
            const float* src = ws;
            int nsum = tp.splits;
            if (tp.groups) {
                float* part2 = ws + (size_t)tp.splits * n;
                hipLaunchKernelGGL(conv3x3_wgrad_group_sum_kernel, dim3(grid_for(n, 1024), tp.groups), dim3(256), 0, st, ws, n,
                                   tp.splits, tp.per_group, part2);
                src = part2;
                nsum = tp.groups;
            }
            hipLaunchKernelGGL(conv3x3_wgrad_tap_reduce_kernel, dim3(ceil_div(C, 64), Cout), dim3(256), 0, st, src, nsum,
                               ntaps, Cout, C, dw + (long)c_start * ntaps, (long)Cin_total * ntaps, 0);
        }
        MNK_LAUNCH_CHECK();
        return MNK_OK;
    }
    if (kh == 3 && kw == 3 && pad == 1) {
        NPlan np = make_nplan(N, H, W, Cout, C, ld_x);
        if (np.use && ((size_t)x % 16 != 0 || (size_t)dy % 16 != 0)) np.use = false;
        if (np.use) {
            WgradN16Args g;
            g.x = x;
            g.ld_x = ld_x;
            g.C = C;
            g.ups = ups;
            g.dy = dy;
            g.ld_dy = ld_dy;
            g.Cout = Cout;
            g.H = H;
            g.W = W;
            g.tiles_w = np.tiles_w;
            g.tiles_per_img = np.tiles_per_img;
            g.total_tiles = np.total_tiles;
            g.tiles_per_split = np.tiles_per_split;
            g.NT = 9 * C;
            g.splits = np.splits;
            float* dstn = dw + (long)c_start * 9;
            const long ldn = (long)Cin_total * 9;
            if (np.splits > 1) {
                const size_t need = (size_t)(np.splits + (defer ? 0 : split_groups(np.splits))) * Cout * g.NT;
                if (!ws || ws_floats < need) {
                    set_error("mnk_conv2d_wgrad: workspace too small (%zu < %zu floats)", ws_floats, need);
                    return MNK_EWORKSPACE;
                }
                g.out = ws;
                g.ld_out = g.NT;
            } else {
                g.out = dstn;
                g.ld_out = ldn;
            }
            hipStream_t sn = (hipStream_t)stream;
            {
                ProfScope prof(K_CONV_WGRAD, sn, 2.0 * (double)N * H * W * Cout * 9.0 * C);
                const int nct = Cout > 32 ? 3 : (Cout > 16 ? 2 : 1), nci = C > 32 ? 3 : (C > 16 ? 2 : 1);
                const dim3 gridn(np.gm, np.gn, np.splits);
#define MNK_N16(T, I)                                                                                      \
    if (nct == T && nci == I) hipLaunchKernelGGL((conv3x3_wgrad_n16_kernel<T, I>), gridn, dim3(256), 0, sn, g)
                MNK_N16(3, 3); MNK_N16(3, 2); MNK_N16(3, 1);
                MNK_N16(2, 3); MNK_N16(2, 2); MNK_N16(2, 1);
                MNK_N16(1, 3); MNK_N16(1, 2); MNK_N16(1, 1);
#undef MNK_N16
            }
            if (np.splits > 1 && !defer) {
                ProfScope prof(K_CONV_REDUCE, sn, (double)np.splits * Cout * g.NT * 4);
                const long n = (long)Cout * g.NT;
                const float* src = ws;
                int nsum = np.splits;
                const int groups = split_groups(np.splits);
                if (groups) {
                    float* part2 = ws + (size_t)np.splits * n;
                    hipLaunchKernelGGL(conv3x3_wgrad_group_sum_kernel, dim3(grid_for(n, 1024), groups), dim3(256), 0, sn, ws,
                                       n, np.splits, 8, part2);
                    src = part2;
                    nsum = groups;
                }
                hipLaunchKernelGGL(conv3x3_wgrad_tap_reduce_kernel, dim3(ceil_div(C, 64), Cout), dim3(256), 0, sn, src, nsum, 9,
                                   Cout, C, dstn, ldn, 0);
            }
            MNK_LAUNCH_CHECK();
            return MNK_OK;
        }
    }
    HPlan hp;
    hp.use = false;
    if (kh == 3 && kw == 3 && pad == 1) hp = make_hplan(N, H, W, Cout, C);
    if (hp.use) {
        WgradHaloArgs h;
        h.x = x;
        h.ld_x = ld_x;
        h.C = C;
        h.ups = ups;
        h.dy = dy;
        h.ld_dy = ld_dy;
        h.Cout = Cout;
        h.N = N;
        h.H = H;
        h.W = W;
        h.TR = hp.TR;
        h.TC = hp.TC;
        h.tiles_w = hp.tiles_w;
        h.tiles_per_img = hp.tiles_per_img;
        h.gn = hp.gn;
        h.total_tiles = hp.total_tiles;
        h.tiles_per_split = hp.tiles_per_split;
        h.NT = 9 * C;
        h.splits = hp.splits;
        float* dsth = dw + (long)c_start * 9;
        const long ldh = (long)Cin_total * 9;
        if (hp.splits > 1) {
            if (!ws || ws_floats < (size_t)(hp.splits + (defer ? 0 : split_groups(hp.splits))) * Cout * h.NT) {
                set_error("mnk_conv2d_wgrad: workspace too small");
                return MNK_EWORKSPACE;
            }
            h.out = ws;
            h.ld_out = h.NT;
        } else {
            h.out = dsth;
            h.ld_out = ldh;
        }
        hipStream_t sh = (hipStream_t)stream;
        {
            ProfScope prof(K_CONV_WGRAD, sh, 2.0 * (double)N * H * W * Cout * 9.0 * C);
            hipLaunchKernelGGL(conv3x3_wgrad_halo_kernel, dim3(hp.gm, hp.gn * 3, hp.splits), dim3(256), 0, sh, h);
        }
        if (hp.splits > 1 && !defer) {
            ProfScope prof(K_CONV_REDUCE, sh, (double)hp.splits * Cout * h.NT * 4);
            launch_wgrad_reduce(ws, hp.splits, Cout, h.NT, dsth, ldh, sh);
        }
        MNK_LAUNCH_CHECK();
        return MNK_OK;
    }
    WgradArgs a;
    a.x = x;
    a.ld_x = ld_x;
    a.C = C;
    a.ups = ups;
    a.dy = dy;
    a.ld_dy = ld_dy;
    a.Cout = Cout;
    a.N = N;
    a.H = H;
    a.W = W;
    a.Hi = Hi;
    a.Wi = Wi;
    a.ntaps = ntaps;
    a.kw = kw;
    a.pad = pad;
    a.M = (long)N * H * W;
    a.NT = ntaps * C;
    WPlan p = make_wplan(a.M, Cout, C, ntaps);
    a.pix_per_split = p.pix_per_split;
    a.splits = p.splits;
    float* dst = dw + (long)c_start * ntaps;
    const long ld_out = (long)Cin_total * ntaps;
    hipStream_t s = (hipStream_t)stream;
    if (p.splits > 1) {
        if (!ws || ws_floats < (size_t)(p.splits + (defer ? 0 : split_groups(p.splits))) * Cout * a.NT) {
            set_error("mnk_conv2d_wgrad: workspace too small");
            return MNK_EWORKSPACE;
        }
        a.out = ws;
        a.ld_out = a.NT;
    } else {
        a.out = dst;
        a.ld_out = ld_out;
    }
    {
        ProfScope prof(K_CONV_WGRAD, s, 2.0 * (double)a.M * Cout * (double)ntaps * C);
        if (p.bm == 128)
            hipLaunchKernelGGL((conv3x3_wgrad_kernel<128>), dim3(p.gm, p.gn, p.splits), dim3(256), 0, s, a);
        else if (p.bm == 64)
            hipLaunchKernelGGL((conv3x3_wgrad_kernel<64>), dim3(p.gm, p.gn, p.splits), dim3(256), 0, s, a);
        else
            hipLaunchKernelGGL((conv3x3_wgrad_kernel<32>), dim3(p.gm, p.gn, p.splits), dim3(256), 0, s, a);
    }
    if (p.splits > 1 && !defer) {
        ProfScope prof(K_CONV_REDUCE, s, (double)p.splits * Cout * a.NT * 4);
        launch_wgrad_reduce(ws, p.splits, Cout, a.NT, dst, ld_out, s);
    }
    MNK_LAUNCH_CHECK();
    return MNK_OK;
}

// which weight-gradient form mnk_conv2d_wgrad runs for a shape (16-byte aligned operands assumed) and what it leaves behind
// under MNK_WGRAD_DEFER: layout 0 = tap-major partials [split][tap][Cout][C], 1 = parameter-major [split][Cout][C*ntaps];
// splits == 0: the GEMM writes dw itself (nothing to reduce)
int mnk_conv2d_wgrad_plan(int N, int Ho, int Wo, int C, int Cout, int kh, int kw, int pad, int ld_x, MnkWgradPlan* plan) {
    return mnk_conv2d_wgrad_plan2(N, Ho, Wo, C, Cout, kh, kw, pad, ld_x, 0, plan);
}

int mnk_conv2d_wgrad_plan2(int N, int Ho, int Wo, int C, int Cout, int kh, int kw, int pad, int ld_x, int flags,
                           MnkWgradPlan* plan) {
    MNK_REQUIRE(plan && N > 0 && Ho > 0 && Wo > 0 && C > 0 && Cout > 0 && kh > 0 && kw > 0 && pad >= 0 && ld_x >= C);
    const int ntaps = kh * kw;
    plan->layout = 0;
    plan->splits = 0;
    plan->part_floats = 0;
    {
        TPlan up = make_up_tplan(N, Ho, Wo, Cout, C, kh, kw, pad, ld_x, flags);
        if (up.use) {                       // sub-pixel form: 16 pseudo taps, folded by the reduction (layout 2)
            plan->layout = 2;
            plan->splits = up.splits;
            plan->part_floats = (size_t)up.splits * 16 * Cout * C;
            return MNK_OK;
        }
    }
    TPlan tp = make_tplan((long)N * Ho * Wo, Cout, C, ntaps, ld_x);
    if (tp.use) {
        plan->splits = tp.splits;
        plan->part_floats = (size_t)tp.splits * ntaps * Cout * C;
        return MNK_OK;
    }
    if (kh == 3 && kw == 3 && pad == 1) {
        NPlan np = make_nplan(N, Ho, Wo, Cout, C, ld_x);
        if (np.use) {
            if (np.splits > 1) {
                plan->splits = np.splits;
                plan->part_floats = (size_t)np.splits * Cout * 9 * C;
            }
            return MNK_OK;
        }
        HPlan hp = make_hplan(N, Ho, Wo, Cout, C);
        if (hp.use) {
            if (hp.splits > 1) {
                plan->layout = 1;
                plan->splits = hp.splits;
                plan->part_floats = (size_t)hp.splits * Cout * 9 * C;
            }
            return MNK_OK;
        }
    }
    WPlan p = make_wplan((long)N * Ho * Wo, Cout, C, ntaps);
    if (p.splits > 1) {
        plan->layout = 1;
        plan->splits = p.splits;
        plan->part_floats = (size_t)p.splits * Cout * ntaps * C;
    }
    return MNK_OK;
}

// the tap-major plan of a job: the sub-pixel form for up-sampled layers with clean sources, else the plain one
static TPlan job_plan(const MnkWgradJob& j, int* subpix, long* M, int* ntaps, int* H, int* W) {
    TPlan up = make_up_tplan(j.N, j.Ho, j.Wo, j.Cout, j.C, j.kh, j.kw, j.pad, j.ld_x, j.flags);
    if (up.use) {
        *subpix = 1, *ntaps = 16, *H = j.Ho / 2, *W = j.Wo / 2;
        *M = (long)j.N * *H * *W;
        return up;
    }
    *subpix = 0, *ntaps = j.kh * j.kw, *H = j.Ho, *W = j.Wo;
    *M = (long)j.N * j.Ho * j.Wo;
    return make_tplan(*M, j.Cout, j.C, *ntaps, j.ld_x);
}

int mnk_wgrad_grouped_plan(MnkWgradJob* jobs, int n) {
    MNK_REQUIRE(jobs && n > 0);
    for (int i = 0; i < n; ++i) {
        MnkWgradJob& j = jobs[i];
        MNK_REQUIRE(j.N > 0 && j.Ho > 0 && j.Wo > 0 && j.C > 0 && j.Cout > 0 && j.kh > 0 && j.kw > 0 && j.pad >= 0);
        int subpix, ntaps, H, W;
        long M;
        TPlan tp = job_plan(j, &subpix, &M, &ntaps, &H, &W);
        j.variant = -1;
        j.splits = 0;
        j.part_floats = 0;
        if (!tp.use) {
            // narrow 3x3 layers: the nine-tap 16x16 kernel, grouped (variants 16 + 3 * (co tiles - 1) + (ci tiles - 1));
            // anything else (LDS-halo, gather, K x K) is launched by the caller on its own
            if (g_wn16_group_target > 0 && j.kh == 3 && j.kw == 3 && j.pad == 1) {
                NPlan np = make_nplan(j.N, j.Ho, j.Wo, j.Cout, j.C, j.ld_x, n16_group_target(j.Cout, j.C));
                if (np.use && np.splits > 1) {
                    const int nct = j.Cout > 32 ? 3 : (j.Cout > 16 ? 2 : 1), nci = j.C > 32 ? 3 : (j.C > 16 ? 2 : 1);
                    j.variant = 16 + 3 * (nct - 1) + (nci - 1);
                    j.splits = np.splits;
                    j.part_floats = (size_t)np.splits * j.Cout * 9 * j.C;
                }
            }
            continue;
        }
        long pps;
        grouped_split(M, &j.splits, &pps);
        WgradTapArgs g;
        const int mode = tap_mode(g, j.N, j.Ho, j.Wo, j.Hi, j.Wi, j.kh, j.kw, j.pad, j.flags & MNK_CONV_UPSAMPLED,
                                  (j.flags & MNK_CONV_CLEAN_PADS) ? 1 : 0, j.ld_x, j.ld_dy, pps, subpix);
        j.variant = 4 * tap_tile_id(tp) + mode;
        j.part_floats = (size_t)j.splits * ntaps * j.Cout * j.C;
    }
    return MNK_OK;
}

size_t mnk_wgrad_grouped_table_bytes(int n) { return n > 0 ? sizeof(GroupedHeader) + (size_t)n * sizeof(TapJobRec) : 0; }

int mnk_wgrad_grouped_build(const MnkWgradJob* jobs, int n, void* host_table, size_t table_bytes) {
    MNK_REQUIRE(jobs && n > 0 && host_table && table_bytes >= mnk_wgrad_grouped_table_bytes(n));
    GroupedHeader* hd = (GroupedHeader*)host_table;
    TapJobRec* recs = (TapJobRec*)((char*)host_table + sizeof(GroupedHeader));
    hd->magic = 0x4d4e4b47;
    hd->n = n;
    hd->nvariants = 25;
    hd->reserved = 0;
    int k = 0;
    for (int v = 0; v < 25; ++v) {
        hd->first[v] = k;
        int blocks = 0;
        for (int i = 0; i < n; ++i) {
            const MnkWgradJob& j = jobs[i];
            MNK_REQUIRE(j.variant >= 0 && j.variant < 25);
            if (j.variant != v) continue;
            MNK_REQUIRE(j.x && j.dy && j.part && ((size_t)j.x % 16) == 0 && ((size_t)j.dy % 16) == 0);
            const int ups = j.flags & MNK_CONV_UPSAMPLED, clean = (j.flags & MNK_CONV_CLEAN_PADS) ? 1 : 0;
            if (v >= 16) {          // nine-tap 16x16 job
                NPlan np = make_nplan(j.N, j.Ho, j.Wo, j.Cout, j.C, j.ld_x, n16_group_target(j.Cout, j.C));
                MNK_REQUIRE(np.use && np.splits == j.splits && np.splits > 1);
                N16JobRec& r = reinterpret_cast<N16JobRec*>(recs)[k];
                WgradN16Args& g = r.a;
                g.x = j.x, g.ld_x = j.ld_x, g.C = j.C, g.ups = ups, g.dy = j.dy, g.ld_dy = j.ld_dy, g.Cout = j.Cout;
                g.H = j.Ho, g.W = j.Wo, g.tiles_w = np.tiles_w, g.tiles_per_img = np.tiles_per_img;
                g.total_tiles = np.total_tiles, g.tiles_per_split = np.tiles_per_split;
                g.NT = 9 * j.C, g.splits = np.splits, g.out = j.part, g.ld_out = g.NT;
                r.gm = np.gm, r.gn = np.gn, r.splits = np.splits, r.block_begin = blocks;
                blocks += np.gm * np.gn * np.splits;
                ++k;
                continue;
            }
            int subpix, ntaps, H, W;
            long M;
            TPlan tp = job_plan(j, &subpix, &M, &ntaps, &H, &W);
            MNK_REQUIRE(tp.use && tap_tile_id(tp) == v / 4);
            TapJobRec& r = recs[k];
            WgradTapArgs& g = r.a;
            int splits;
            long pps;
            grouped_split(M, &splits, &pps);
            MNK_REQUIRE(splits == j.splits);
            g.x = j.x;
            g.ld_x = j.ld_x;
            g.C = j.C;
            g.ups = subpix ? 0 : ups;
            g.dy = j.dy;
            g.ld_dy = j.ld_dy;
            g.Cout = j.Cout;
            g.H = H;
            g.W = W;
            g.Hi = subpix ? H : j.Hi;
            g.Wi = subpix ? W : j.Wi;
            g.ntaps = ntaps;
            g.kw = subpix ? 4 : j.kw;
            g.pad = subpix ? 0 : j.pad;
            g.M = M;
            g.pix_per_split = pps;
            g.gn = tp.gn;
            g.part = j.part;
            g.xcd = 0;
            g.clean = clean;
            fast_div_consts((unsigned)W, &g.mulW, &g.shW);
            fast_div_consts((unsigned)H, &g.mulH, &g.shH);
            const int mode = tap_mode(g, j.N, j.Ho, j.Wo, j.Hi, j.Wi, j.kh, j.kw, j.pad, ups, clean, j.ld_x, j.ld_dy, pps, subpix);
            MNK_REQUIRE(mode == v % 4);
            g.compact = tap_compact_ok(H, W, M, j.kh, j.kw, j.pad, subpix ? 0 : ups, subpix) ? 1 : 0;
            g.nsplits = splits;
            r.gm = tp.gm;
            r.gnt = tp.gn * ntaps;
            r.splits = splits;
            r.block_begin = blocks;
            blocks += r.gm * r.gnt * r.splits;
            ++k;
        }
        hd->count[v] = k - hd->first[v];
        hd->blocks[v] = blocks;
    }
    MNK_REQUIRE(k == n);
    return MNK_OK;
}

int mnk_wgrad_grouped_launch(const void* device_table, const void* host_table, void* stream) {
    MNK_REQUIRE(device_table && host_table);
    const GroupedHeader* hd = (const GroupedHeader*)host_table;
    MNK_REQUIRE(hd->magic == 0x4d4e4b47 && hd->n > 0);
    const TapJobRec* hrecs = (const TapJobRec*)((const char*)host_table + sizeof(GroupedHeader));
    const TapJobRec* drecs = (const TapJobRec*)((const char*)device_table + sizeof(GroupedHeader));
    hipStream_t st = (hipStream_t)stream;
    for (int v = 0; v < 16; ++v) {
        const int cnt = hd->count[v], blocks = hd->blocks[v];
        if (!cnt) continue;
        double flop = 0.0, issued = 0.0;
        for (int i = 0; i < cnt; ++i) {
            const WgradTapArgs& g = hrecs[hd->first[v] + i].a;      // algorithmic: the sub-pixel form stands for 9 taps at 4 M pixels
            flop += v % 4 == 3 ? 2.0 * 4.0 * (double)g.M * g.Cout * 9.0 * g.C : 2.0 * (double)g.M * g.Cout * (double)g.ntaps * g.C;
            issued += g.compact ? 2.0 * tap_compact_pairs(g.M / ((long)g.H * g.W), g.H, g.W, v % 4 == 3) * g.Cout * g.C
                                : 2.0 * (double)g.M * g.Cout * (double)g.ntaps * g.C;
        }
        ProfScope prof(K_CONV_WGRAD, st, flop, issued);
        const TapJobRec* rv = drecs + hd->first[v];
        const int mode = v % 4;
#define MNK_WGROUP(...)                                                                                                        \
    do {                                                                                                                       \
        if (mode == 1) hipLaunchKernelGGL((conv3x3_wgrad_tap_grouped_kernel<__VA_ARGS__, 1>), dim3(blocks), dim3(256), 0, st, rv, cnt);      \
        else if (mode == 2) hipLaunchKernelGGL((conv3x3_wgrad_tap_grouped_kernel<__VA_ARGS__, 2>), dim3(blocks), dim3(256), 0, st, rv, cnt); \
        else if (mode == 3) hipLaunchKernelGGL((conv3x3_wgrad_tap_grouped_kernel<__VA_ARGS__, 3>), dim3(blocks), dim3(256), 0, st, rv, cnt); \
        else hipLaunchKernelGGL((conv3x3_wgrad_tap_grouped_kernel<__VA_ARGS__, 0>), dim3(blocks), dim3(256), 0, st, rv, cnt);                \
    } while (0)
#define MNK_WGROUP_H(...)                                                                                                     \
    do {                                                                                                                      \
        if (mode == 3) hipLaunchKernelGGL((conv3x3_wgrad_tap_grouped_h_kernel<__VA_ARGS__, true>), dim3(blocks), dim3(256), 0, st, rv, cnt);  \
        else hipLaunchKernelGGL((conv3x3_wgrad_tap_grouped_h_kernel<__VA_ARGS__, false>), dim3(blocks), dim3(256), 0, st, rv, cnt);           \
    } while (0)
        if (g_wgrad_bf16x3) {
            switch (v / 4) {
                case 0: MNK_WGROUP_H(128, 128, 2, 2); break;
                case 1: MNK_WGROUP_H(128, 64, 2, 2); break;
                case 2: MNK_WGROUP_H(64, 128, 1, 4); break;
                default: MNK_WGROUP_H(32, 128, 1, 4); break;
            }
        } else
        switch (v / 4) {
            case 0: MNK_WGROUP(128, 128, 2, 2); break;
            case 1: MNK_WGROUP(128, 64, 2, 2); break;
            case 2: MNK_WGROUP(64, 128, 1, 4); break;
            default: MNK_WGROUP(32, 128, 1, 4); break;
        }
#undef MNK_WGROUP_H
#undef MNK_WGROUP
    }
    for (int v = 16; v < 25; ++v) {
        const int cnt = hd->count[v], blocks = hd->blocks[v];
        if (!cnt) continue;
        const N16JobRec* hn = reinterpret_cast<const N16JobRec*>(hrecs) + hd->first[v];
        double flop = 0.0;
        for (int i = 0; i < cnt; ++i) {
            const WgradN16Args& g = hn[i].a;
            flop += 2.0 * (double)g.total_tiles * 64.0 * g.Cout * 9.0 * g.C;
        }
        ProfScope prof(K_CONV_WGRAD, st, flop);
        const N16JobRec* rv = reinterpret_cast<const N16JobRec*>(drecs) + hd->first[v];
        const int nct = (v - 16) / 3 + 1, nci = (v - 16) % 3 + 1;
#define MNK_N16G(T, I)                                                                                                  \
    if (nct == T && nci == I) hipLaunchKernelGGL((conv3x3_wgrad_n16_grouped_kernel<T, I>), dim3(blocks), dim3(256), 0, st, rv, cnt)
        MNK_N16G(3, 3); MNK_N16G(3, 2); MNK_N16G(3, 1);
        MNK_N16G(2, 3); MNK_N16G(2, 2); MNK_N16G(2, 1);
        MNK_N16G(1, 3); MNK_N16G(1, 2); MNK_N16G(1, 1);
#undef MNK_N16G
    }
    MNK_LAUNCH_CHECK();
    return MNK_OK;
}

// launch-plan switches are read from the environment when the library is loaded; this sets one afterwards (A/B runs, tests)
// the last forward / data-gradient launch plan that was made: {M, Cout, chunks, taps, phases, bm, bn, splits}
int mnk_last_plan(long* out8) {
    MNK_REQUIRE(out8);
    for (int i = 0; i < 8; ++i) out8[i] = g_last_plan[i];
    return MNK_OK;
}

int mnk_wgrad_reduce_blocks(int splits, int Cout, int C) {
    if (splits <= 0 || Cout <= 0 || C <= 0) return 0;
    // (the flat map is taken for 3x3 / 4x4 kernels only; any other tap count still gets enough blocks from it: the tile map
    // needs ceil(Cout / 4) * ceil(C / 64) <= ceil(Cout * C / 1024))
    if (reduce_flat(splits, C)) {
        const long flat = ((long)Cout * (C >> 2) + 255) / 256, tile = (long)ceil_div(Cout, 4) * ceil_div(C, 64);
        return (int)(flat > tile ? flat : tile);
    }
    int tw, rows;
    reduce_map(splits, &tw, &rows);
    return ceil_div(Cout, rows) * ceil_div(C, tw);
}

int mnk_wgrad_reduce_multi(const MnkWgradReduceDesc* descs_device, int n, int total_blocks, void* stream) {
    MNK_REQUIRE(descs_device && n > 0 && total_blocks > 0);
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(K_CONV_REDUCE, s, 0.0);
    hipLaunchKernelGGL(wgrad_reduce_multi_kernel, dim3(total_blocks), dim3(256), 0, s, descs_device, n);
    MNK_LAUNCH_CHECK();
    return MNK_OK;
}

// ---- sub-pixel forms of UpBlock3D's [nearest x2 -> 3x3 / pad 1] (modules/util.py:83-85): (H, W) = LOW resolution -----------
size_t mnk_conv3x3_up_packed_floats(int Cout, int C0, int C1) {
    if (Cout <= 0 || C0 <= 0 || C1 < 0) return 0;
    return (size_t)16 * Cout * (round_up(C0, 16) + (C1 > 0 ? round_up(C1, 16) : 0));
}
size_t mnk_conv3x3_up_dgrad_packed_floats(int Cout, int c_count) {
    if (Cout <= 0 || c_count <= 0) return 0;
    return (size_t)c_count * 16 * round_up(Cout, 16);
}
int mnk_conv3x3_up_pack_fwd(const float* w, float* wp, int Cout, int C0, int C1, void* stream) {
    MNK_REQUIRE(w && wp && Cout > 0 && C0 > 0 && C1 >= 0);
    hipStream_t s = (hipStream_t)stream;
    const int C0p = round_up(C0, 16), C1p = C1 > 0 ? round_up(C1, 16) : 0;
    const long total = (long)16 * Cout * (C0p + C1p);
    ProfScope prof(K_CONV_REDUCE, s, (double)total * 8);
    hipLaunchKernelGGL(pack_up_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, s, w, wp, Cout, C0, C1, C0p, C1p);
    MNK_LAUNCH_CHECK();
    return MNK_OK;
}
int mnk_conv3x3_up_pack_dgrad(const float* w, float* wp, int Cout, int Cin_total, int c_start, int c_count, void* stream) {
    MNK_REQUIRE(w && wp && Cout > 0 && Cin_total > 0 && c_start >= 0 && c_count > 0 && c_start + c_count <= Cin_total);
    hipStream_t s = (hipStream_t)stream;
    const int chunks = round_up(Cout, 16) / 16;
    const long total = (long)c_count * chunks * 256;
    ProfScope prof(K_CONV_REDUCE, s, (double)total * 8);
    hipLaunchKernelGGL(pack_up_dgrad_kernel, dim3(grid_for(total)), dim3(256), 0, s, w, wp, Cout, Cin_total, c_start, c_count,
                       chunks);
    MNK_LAUNCH_CHECK();
    return MNK_OK;
}
size_t mnk_conv3x3_up_workspace_floats(int N, int H, int W, int C0, int C1, int Cout) {
    if (N <= 0 || H <= 0 || W <= 0 || C0 <= 0 || C1 < 0 || Cout <= 0) return 0;
    const int chunks = (round_up(C0, 16) + (C1 > 0 ? round_up(C1, 16) : 0)) / 16;
    Plan p = make_plan((long)N * H * W, Cout, chunks, 4, 4);
    return p.splits > 1 ? (size_t)p.splits * 4 * N * H * W * p.ldw : 0;
}
size_t mnk_conv3x3_up_stats_floats(int N, int H, int W, int C0, int C1, int Cout) {
    if (N <= 0 || H <= 0 || W <= 0 || C0 <= 0 || C1 < 0 || Cout <= 0) return 0;
    const int chunks = (round_up(C0, 16) + (C1 > 0 ? round_up(C1, 16) : 0)) / 16;
    Plan p = make_plan((long)N * H * W, Cout, chunks, 4, 4);
    if (p.splits > 1)
        return g_splitk_stats ? (size_t)make_rsmap(4L * N * H * W, round_up(Cout, 4)).row_blocks * 2 * round_up(Cout, 4) : 0;
    return (size_t)4 * p.gm * 2 * round_up(Cout, 4);
}
int mnk_conv3x3_up_fwd(const float* x0, int ld0, int C0, const float* x1, int ld1, int C1, int flags, const float* wp_up,
                       const float* bias, float* y, int ld_y, int N, int H, int W, int Cout, float* ws, size_t ws_floats,
                       float* stats_partial, void* stream) {
    MNK_REQUIRE((flags & ~MNK_CONV_DEFER_SPLITK) == 0);
    return conv2d_fwd_impl(x0, ld0, C0, x1, ld1, C1, MNK_CONV_CLEAN_PADS | flags, H, W, 2, 2, 1, 1, 4, wp_up, bias, nullptr, 0, y,
                           ld_y, N, H, W, Cout, ws, ws_floats, stats_partial, stream);
}
// pixel-independent K splits of the launches above (1: no split): what a caller that sums the partials itself must know
int mnk_conv3x3_splits(int N, int H, int W, int C0, int C1, int Cout) {
    if (N <= 0 || H <= 0 || W <= 0 || C0 <= 0 || C1 < 0 || Cout <= 0) return 0;
    return make_plan((long)N * H * W, Cout, (round_up(C0, 16) + (C1 > 0 ? round_up(C1, 16) : 0)) / 16, 9, 1).splits;
}
int mnk_conv3x3_up_splits(int N, int H, int W, int C0, int C1, int Cout) {
    if (N <= 0 || H <= 0 || W <= 0 || C0 <= 0 || C1 < 0 || Cout <= 0) return 0;
    return make_plan((long)N * H * W, Cout, (round_up(C0, 16) + (C1 > 0 ? round_up(C1, 16) : 0)) / 16, 4, 4).splits;
}
// data gradient w.r.t. one low-resolution source of an up-sampled convolution: dy (N, 2H, 2W, Cout) -> dx (N, H, W, C)
size_t mnk_conv3x3_up_dgrad_workspace_floats(int N, int H, int W, int Cout, int C) {
    if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || Cout <= 0) return 0;
    Plan p = make_plan((long)N * H * W, C, round_up(Cout, 16) / 16, 16, 1);
    return p.splits > 1 ? (size_t)p.splits * N * H * W * p.ldw : 0;
}
int mnk_conv3x3_up_dgrad(const float* dy, int ld_dy, int Cout, const float* wp_up_dgrad, float* dx, int ld_dx, int N, int H,
                         int W, int C, float* ws, size_t ws_floats, void* stream) {
    return conv2d_fwd_impl(dy, ld_dy, Cout, nullptr, 0, 0, MNK_CONV_CLEAN_PADS, 2 * H, 2 * W, 4, 4, 1, 2, 1, wp_up_dgrad, nullptr,
                           nullptr, 0, dx, ld_dx, N, H, W, C, ws, ws_floats, nullptr, stream);
}

// ---- data-gradient launches that also leave the backward statistics of the BatchNorm layer in front (round 4) ------------------
static BnBwdSrc bnb_of(const float* bn_y, int ld_bny, const float* mean, const float* invstd, const float* scale, const float* beta,
                       float slope) {
    BnBwdSrc b;
    b.y = bn_y, b.ld = ld_bny, b.mean = mean, b.invstd = invstd, b.scale = scale, b.beta = beta, b.slope = slope;
    return b;
}
int mnk_conv3x3_dgrad_bnstats(const float* dy, int ld_dy, int Cout, const float* wp_dgrad, const float* residual, int ld_res,
                              float* dx, int ld_dx, int N, int H, int W, int C, float* ws, size_t ws_floats, float* stats_partial,
                              const float* bn_y, int ld_bny, const float* bn_mean, const float* bn_invstd, const float* bn_scale,
                              const float* bn_beta, float slope, void* stream) {
    const BnBwdSrc b = bnb_of(bn_y, ld_bny, bn_mean, bn_invstd, bn_scale, bn_beta, slope);
    return conv2d_fwd_impl(dy, ld_dy, Cout, nullptr, 0, 0, MNK_CONV_CLEAN_PADS, H, W, 3, 3, 1, 1, 1, wp_dgrad, nullptr, residual,
                           ld_res, dx, ld_dx, N, H, W, C, ws, ws_floats, stats_partial, stream, &b);
}
size_t mnk_conv3x3_up_dgrad_stats_floats(int N, int H, int W, int Cout, int C) {
    return mnk_conv2d_stats_floats(N, H, W, Cout, 0, C, 16);
}
int mnk_conv3x3_up_dgrad_bnstats(const float* dy, int ld_dy, int Cout, const float* wp_up_dgrad, float* dx, int ld_dx, int N, int H,
                                 int W, int C, float* ws, size_t ws_floats, float* stats_partial, const float* bn_y, int ld_bny,
                                 const float* bn_mean, const float* bn_invstd, const float* bn_scale, const float* bn_beta,
                                 float slope, void* stream) {
    const BnBwdSrc b = bnb_of(bn_y, ld_bny, bn_mean, bn_invstd, bn_scale, bn_beta, slope);
    return conv2d_fwd_impl(dy, ld_dy, Cout, nullptr, 0, 0, MNK_CONV_CLEAN_PADS, 2 * H, 2 * W, 4, 4, 1, 2, 1, wp_up_dgrad, nullptr,
                           nullptr, 0, dx, ld_dx, N, H, W, C, ws, ws_floats, stats_partial, stream, &b);
}

// ---- 3x3 / pad 1 forms (the hot path's nn.Conv3d (1,3,3)) ------------------------------------------------------------
size_t mnk_conv3x3_packed_floats(int Cout, int C0, int C1) { return mnk_conv2d_packed_floats(Cout, C0, C1, 9); }
int mnk_conv3x3_pack_fwd(const float* w, float* wp, int Cout, int C0, int C1, void* stream) {
    return mnk_conv2d_pack_fwd(w, wp, Cout, C0, C1, 9, stream);
}
int mnk_conv3x3_pack_all(const float* w, float* wp_fwd, float* wp_d0, float* wp_d1, int Cout, int C0, int C1,
                         void* stream) {
    return mnk_conv2d_pack_all(w, wp_fwd, wp_d0, wp_d1, Cout, C0, C1, 9, stream);
}
int mnk_conv3x3_pack_dgrad(const float* w, float* wp, int Cout, int Cin_total, int c_start, int c_count, void* stream) {
    return mnk_conv2d_pack_dgrad(w, wp, Cout, Cin_total, c_start, c_count, 9, stream);
}
size_t mnk_conv3x3_workspace_floats(int N, int H, int W, int C0, int C1, int Cout) {
    return mnk_conv2d_workspace_floats(N, H, W, C0, C1, Cout, 9);
}
size_t mnk_conv3x3_stats_floats(int N, int H, int W, int C0, int C1, int Cout) {
    return mnk_conv2d_stats_floats(N, H, W, C0, C1, Cout, 9);
}
int mnk_conv3x3_fwd(const float* x0, int ld0, int C0, const float* x1, int ld1, int C1, int flags, const float* wp,
                    const float* bias, const float* residual, int ld_res, float* y, int ld_y, int N, int H, int W,
                    int Cout, float* ws, size_t ws_floats, float* stats_partial, void* stream) {
    return mnk_conv2d_fwd(x0, ld0, C0, x1, ld1, C1, flags, H, W, 3, 3, 1, wp, bias, residual, ld_res, y, ld_y, N, H, W, Cout,
                          ws, ws_floats, stats_partial, stream);
}
size_t mnk_conv3x3_wgrad_workspace_floats(int N, int H, int W, int C, int Cout) {
    return mnk_conv2d_wgrad_workspace_floats(N, H, W, C, Cout, 3, 3, 1);
}
int mnk_conv3x3_wgrad(const float* x, int ld_x, int C, int flags, const float* dy, int ld_dy, int Cout, float* dw,
                      int Cin_total, int c_start, int N, int H, int W, float* ws, size_t ws_floats, void* stream) {
    return mnk_conv2d_wgrad(x, ld_x, C, flags, H, W, 3, 3, 1, dy, ld_dy, Cout, dw, Cin_total, c_start, N, H, W, ws, ws_floats,
                            stream);
}
}

#ifdef MNK_PHASE_CLOCKS
extern "C" int mnk_phase_sclk_read(void* host, size_t bytes) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(mnk_phase_sclk), bytes) == hipSuccess ? MNK_OK : MNK_ELAUNCH;
}
extern "C" int mnk_phase_log_read(void* host, size_t bytes, int clear) {
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(mnk_phase_log), bytes) != hipSuccess) return MNK_ELAUNCH;
    if (clear) {
        void* p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(mnk_phase_log)) != hipSuccess) return MNK_ELAUNCH;
        if (hipMemset(p, 0, sizeof(unsigned long long) * 4 * 16384) != hipSuccess) return MNK_ELAUNCH;
    }
    return MNK_OK;
}
#endif

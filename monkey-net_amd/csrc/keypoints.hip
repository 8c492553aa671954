// Heat-map <-> key-point kernels.
//   mnk_softmax_kp_*        : KPDetector head, modules/keypoint_detector.py:43-78,103-107
//   mnk_gaussian_sums       : 'sum' normaliser of MovementEmbeddingModule.normalize_heatmap (movement_embedding.py:33-38)
//   mnk_movement_embedding_*: MovementEmbeddingModule.forward (movement_embedding.py:42-92) incl. kp2gaussian
//                             (keypoint_detector.py:7-40) and the translation grid_sample (:76-87)
// All HBM/L2-bound; spatial reductions use wavefront shuffles + one LDS hop across the 4 waves of a block.
#include <stdlib.h>

#include "mnk_common.h"

using namespace mnk;

namespace {

constexpr int MAXK = 16;   // key-points per block-row handled in registers

// align_corners=True grid of modules/util.py:26-42:  x_j = 2*(j/(w-1)) - 1
__device__ __forceinline__ float grid_coord(int j, int n) { return 2.f * ((float)j / (float)(n - 1)) - 1.f; }

// a * d - b * c without the cancellation of the naive form (Kahan): the covariances this file inverts are clipped to
// sigma_min >= 1e-3 at sigma_max up to a few hundred, i.e. det ~ 0.5 from products ~ 3e4
__device__ __forceinline__ float det_2x2(float a, float b, float c, float d) {
    const float w = b * c;
    const float e = fmaf(-b, c, w);        // rounding error of w
    const float f = fmaf(a, d, -w);
    return f + e;
}

// Block-wide sums of small per-thread register arrays.  Loops are fully unrolled with compile-time indices so
// the arrays stay in VGPRs (runtime-indexed arrays would be demoted to scratch).
struct BlockRed {
    float* red;  // [4][NMAX]
    template <int NMAX>
    __device__ __forceinline__ void reduce(float (&v)[NMAX], int n) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < NMAX; ++k)
            if (k < n) {
                float s = wave_sum(v[k]);
                if (lane == 0) red[wave * NMAX + k] = s;
            }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < NMAX; ++k)
            if (k < n) v[k] = red[k] + red[NMAX + k] + red[2 * NMAX + k] + red[3 * NMAX + k];
    }
    template <int NMAX>
    __device__ __forceinline__ void reduce_max(float (&v)[NMAX], int n) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < NMAX; ++k)
            if (k < n) {
                float s = wave_max(v[k]);
                if (lane == 0) red[wave * NMAX + k] = s;
            }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < NMAX; ++k)
            if (k < n) v[k] = fmaxf(fmaxf(red[k], red[NMAX + k]), fmaxf(red[2 * NMAX + k], red[3 * NMAX + k]));
    }
};

// one block per frame and group of kg channels; the channels of a pixel are contiguous (NHWC): a pixel is one 4..64 B read
__global__ void __launch_bounds__(256) softmax_kp_fwd_multipass_kernel(const float* __restrict__ heat, int ld, int H, int W,
                                                             int Kall, int kg, float temperature,
                                                             float* __restrict__ mean, float* __restrict__ var,
                                                             float* __restrict__ stat) {
    __shared__ float red_mem[4 * MAXK];
    BlockRed br{red_mem};
    // blockIdx.y = group of kg key-point channels (all of them: one block per frame): K = channels of this block
    const int n = blockIdx.x, t = threadIdx.x, k0 = blockIdx.y * kg;
    const int K = Kall - k0 < kg ? Kall - k0 : kg;
    const int P = H * W;
    const float* hp = heat + (long)n * P * ld + k0;
    float mx[MAXK];
#pragma unroll
    for (int k = 0; k < MAXK; ++k) mx[k] = -INFINITY;
    for (int p = t; p < P; p += 256) {
#pragma unroll
        for (int k = 0; k < MAXK; ++k)
            if (k < K) mx[k] = fmaxf(mx[k], hp[(long)p * ld + k] / temperature);
    }
    br.reduce_max(mx, K);
    // pass 2: S = sum e, Sx = sum e*gx, Sy = sum e*gy  (+ G = sum of grid coords for the +1e-7 term)
    float S[MAXK], Sx[MAXK], Sy[MAXK];
#pragma unroll
    for (int k = 0; k < MAXK; ++k) S[k] = Sx[k] = Sy[k] = 0.f;
    float gsum[2] = {0.f, 0.f};
    for (int p = t; p < P; p += 256) {
        const float gx = grid_coord(p % W, W), gy = grid_coord(p / W, H);
        gsum[0] += gx;
        gsum[1] += gy;
#pragma unroll
        for (int k = 0; k < MAXK; ++k)
            if (k < K) {
                const float e = expf(hp[(long)p * ld + k] / temperature - mx[k]);
                S[k] += e;
                Sx[k] += e * gx;
                Sy[k] += e * gy;
            }
    }
    br.reduce(S, K);
    br.reduce(Sx, K);
    br.reduce(Sy, K);
    br.reduce(gsum, 2);
    float mux[MAXK], muy[MAXK], invS[MAXK];
#pragma unroll
    for (int k = 0; k < MAXK; ++k)
        if (k < K) {
            invS[k] = 1.f / S[k];
            mux[k] = Sx[k] * invS[k] + 1e-7f * gsum[0];
            muy[k] = Sy[k] * invS[k] + 1e-7f * gsum[1];
        }
    // pass 3: centred second moments with weight p + 1e-7 (keypoint_detector.py:49,57-60)
    float vxx[MAXK], vxy[MAXK], vyy[MAXK];
#pragma unroll
    for (int k = 0; k < MAXK; ++k) vxx[k] = vxy[k] = vyy[k] = 0.f;
    for (int p = t; p < P; p += 256) {
        const float gx = grid_coord(p % W, W), gy = grid_coord(p / W, H);
#pragma unroll
        for (int k = 0; k < MAXK; ++k)
            if (k < K) {
                const float wgt = expf(hp[(long)p * ld + k] / temperature - mx[k]) * invS[k] + 1e-7f;
                const float dx = gx - mux[k], dy = gy - muy[k];
                vxx[k] += wgt * dx * dx;
                vxy[k] += wgt * dx * dy;
                vyy[k] += wgt * dy * dy;
            }
    }
    br.reduce(vxx, K);
    br.reduce(vxy, K);
    br.reduce(vyy, K);
    if (t == 0) {
#pragma unroll
        for (int k = 0; k < MAXK; ++k) {
            if (k >= K) continue;
            const long o = (long)n * Kall + k0 + k;
            mean[o * 2 + 0] = mux[k];
            mean[o * 2 + 1] = muy[k];
            var[o * 4 + 0] = vxx[k];
            var[o * 4 + 1] = vxy[k];
            var[o * 4 + 2] = vxy[k];
            var[o * 4 + 3] = vyy[k];
            stat[o * 2 + 0] = mx[k];
            stat[o * 2 + 1] = S[k];
        }
    }
}

__global__ void __launch_bounds__(256) softmax_kp_bwd_multipass_kernel(const float* __restrict__ heat, int ld, int H, int W,
                                                             int Kall, int kg, float temperature,
                                                             const float* __restrict__ mean,
                                                             const float* __restrict__ stat,
                                                             const float* __restrict__ dmean,
                                                             const float* __restrict__ dvar, float* __restrict__ dheat,
                                                             int ld_d) {
    __shared__ float red_mem[4 * MAXK];
    BlockRed br{red_mem};
    const int n = blockIdx.x, t = threadIdx.x, k0 = blockIdx.y * kg;
    const int K = Kall - k0 < kg ? Kall - k0 : kg;
    const int P = H * W;
    const float* hp = heat + (long)n * P * ld + k0;
    float mux[MAXK], muy[MAXK], mx[MAXK], invS[MAXK], gmx[MAXK], gmy[MAXK], v00[MAXK], v01[MAXK], v11[MAXK];
#pragma unroll
    for (int k = 0; k < MAXK; ++k)
        if (k < K) {
            const long o = (long)n * Kall + k0 + k;
            mux[k] = mean[o * 2];
            muy[k] = mean[o * 2 + 1];
            mx[k] = stat[o * 2];
            invS[k] = 1.f / stat[o * 2 + 1];
            const float d00 = dvar[o * 4], d01 = dvar[o * 4 + 1], d10 = dvar[o * 4 + 2], d11 = dvar[o * 4 + 3];
            // c = sum_i w_i (g_i - mu) = -mu * P * 1e-7 (weights sum to 1 + P*1e-7); the centring of var feeds
            // back into mean:  dmu_total = dmean - (dvar + dvar^T) c
            const float cx = -mux[k] * (float)P * 1e-7f, cy = -muy[k] * (float)P * 1e-7f;
            gmx[k] = dmean[o * 2] - (2.f * d00 * cx + (d01 + d10) * cy);
            gmy[k] = dmean[o * 2 + 1] - ((d01 + d10) * cx + 2.f * d11 * cy);
            v00[k] = d00;
            v01[k] = d01 + d10;
            v11[k] = d11;
        }
    float A[MAXK];
#pragma unroll
    for (int k = 0; k < MAXK; ++k) A[k] = 0.f;
    for (int p = t; p < P; p += 256) {
        const float gx = grid_coord(p % W, W), gy = grid_coord(p / W, H);
#pragma unroll
        for (int k = 0; k < MAXK; ++k)
            if (k < K) {
                const float s = expf(hp[(long)p * ld + k] / temperature - mx[k]) * invS[k];
                const float dx = gx - mux[k], dy = gy - muy[k];
                const float ai = gmx[k] * gx + gmy[k] * gy + v00[k] * dx * dx + v01[k] * dx * dy + v11[k] * dy * dy;
                A[k] += s * ai;
            }
    }
    br.reduce(A, K);
    float* dp = dheat + (long)n * P * ld_d + k0;
    const bool last_group = k0 + K == Kall;     // the block of the last channels also zeroes the pad channels
    for (int p = t; p < P; p += 256) {
        const float gx = grid_coord(p % W, W), gy = grid_coord(p / W, H);
#pragma unroll
        for (int k = 0; k < MAXK; ++k)
            if (k < K) {
                const float s = expf(hp[(long)p * ld + k] / temperature - mx[k]) * invS[k];
                const float dx = gx - mux[k], dy = gy - muy[k];
                const float ai = gmx[k] * gx + gmy[k] * gy + v00[k] * dx * dx + v01[k] * dx * dy + v11[k] * dy * dy;
                dp[(long)p * ld_d + k] = s * (ai - A[k]) / temperature;
            }
        if (last_group)
            for (int k = K; k < ld_d - k0; ++k) dp[(long)p * ld_d + k] = 0.f;
    }
}

// ---- one-pass soft-argmax (round 6) ------------------------------------------------------------------------------------
// One WAVEFRONT (heat-maps up to 32 x 32) or one four-wavefront block (up to 64 x 64) per (frame, key-point channel): the
// channel's whole heat-map is held in registers (<= 16 pixels per lane, pixel p = j * lanes + lane, so a load instruction's
// lanes touch neighbouring pixels), every element is divided by the temperature once and exponentiated once, and the nine sums
// (max; S, Sx, Sy and the two grid sums; the three centred second moments) are wavefront shuffles (+ one LDS hop between the
// four wavefronts of the large form) -- one read of the heat-map.  frames x K wavefronts (640 for the 64 frames x 10 key
// points of a batch-32 iteration) instead of frames x 2 blocks that walked global memory three times.  Same formulas as the
// multi-pass kernels above (keypoint_detector.py:43-78,103-107), which stay for heat-maps beyond 64 x 64.
template <int WAVES>
struct PairRed {      // sums / maxima over the WAVES wavefronts that share a (frame, key point) pair; every lane gets the result
    float* red;       // [WAVES][8] (WAVES > 1 only)
    template <int N>
    __device__ __forceinline__ void sum(float (&v)[N]) {
        static_assert(N <= 8, "eight values per round");
#pragma unroll
        for (int k = 0; k < N; ++k) v[k] = wave_sum(v[k]);
        if constexpr (WAVES > 1) {
            const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
            __syncthreads();
            if (lane == 0)
#pragma unroll
                for (int k = 0; k < N; ++k) red[wave * 8 + k] = v[k];
            __syncthreads();
#pragma unroll
            for (int k = 0; k < N; ++k) {
                float t = red[k];
#pragma unroll
                for (int w = 1; w < WAVES; ++w) t += red[w * 8 + k];
                v[k] = t;
            }
        }
    }
    __device__ __forceinline__ float max(float v) {
        v = wave_max(v);
        if constexpr (WAVES > 1) {
            const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
            __syncthreads();
            if (lane == 0) red[wave * 8] = v;
            __syncthreads();
            v = red[0];
#pragma unroll
            for (int w = 1; w < WAVES; ++w) v = fmaxf(v, red[w * 8]);
        }
        return v;
    }
};

// WAVES = 1: four pairs per 256-thread block (no barrier anywhere); WAVES = 4: one pair per block
template <int PPT, int WAVES>
__global__ void __launch_bounds__(256) softmax_kp_fwd_kernel(const float* __restrict__ heat, int ld, int H, int W, int K,
                                                             int pairs, float temperature, float* __restrict__ mean,
                                                             float* __restrict__ var, float* __restrict__ stat) {
    __shared__ float red_mem[WAVES > 1 ? WAVES * 8 : 1];
    PairRed<WAVES> pr{red_mem};
    constexpr int LANES = 64 * WAVES;
    const int tl = threadIdx.x % LANES;                                  // position inside the pair's thread group
    const int pair = blockIdx.x * (4 / WAVES) + threadIdx.x / LANES;
    if (pair >= pairs) return;                                           // (WAVES = 1 only: wave-uniform, no barrier follows)
    const int n = pair / K, k = pair - n * K;
    const int P = H * W;
    const float* hp = heat + (long)n * P * ld + k;
    const float invW = 1.f / (float)W;
    float v[PPT];
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        const int p = j * LANES + tl;
        v[j] = p < P ? hp[(long)p * ld] : -INFINITY;
    }
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        v[j] = v[j] / temperature;
        mx = fmaxf(mx, v[j]);
    }
    mx = pr.max(mx);
    float s5[5] = {0.f, 0.f, 0.f, 0.f, 0.f};      // S, Sx, Sy, sum gx, sum gy
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        const int p = j * LANES + tl;
        // p / W for p < 4096: (p + 0.5) / W is at least 0.5 / W away from an integer, far beyond the rounding of the product
        const int py = (int)(((float)p + 0.5f) * invW), px = p - py * W;
        const bool in = p < P;
        const float gx = in ? grid_coord(px, W) : 0.f, gy = in ? grid_coord(py, H) : 0.f;
        const float e = in ? expf(v[j] - mx) : 0.f;
        v[j] = e;
        s5[0] += e;
        s5[1] += e * gx;
        s5[2] += e * gy;
        s5[3] += gx;
        s5[4] += gy;
    }
    pr.sum(s5);
    const float S = s5[0], invS = 1.f / S;
    const float mux = s5[1] * invS + 1e-7f * s5[3], muy = s5[2] * invS + 1e-7f * s5[4];
    // centred second moments with weight p + 1e-7 (keypoint_detector.py:49,57-60) from the held exponentials
    float m3[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        const int p = j * LANES + tl;
        const int py = (int)(((float)p + 0.5f) * invW), px = p - py * W;
        const float wgt = p < P ? v[j] * invS + 1e-7f : 0.f;
        const float dx = grid_coord(px, W) - mux, dy = grid_coord(py, H) - muy;
        m3[0] += wgt * dx * dx;
        m3[1] += wgt * dx * dy;
        m3[2] += wgt * dy * dy;
    }
    pr.sum(m3);
    if (tl == 0) {
        const long o = pair;
        mean[o * 2 + 0] = mux;
        mean[o * 2 + 1] = muy;
        var[o * 4 + 0] = m3[0];
        var[o * 4 + 1] = m3[1];
        var[o * 4 + 2] = m3[1];
        var[o * 4 + 3] = m3[2];
        stat[o * 2 + 0] = mx;
        stat[o * 2 + 1] = S;
    }
}

template <int PPT, int WAVES>
__global__ void __launch_bounds__(256) softmax_kp_bwd_kernel(const float* __restrict__ heat, int ld, int H, int W, int K,
                                                             int pairs, float temperature, const float* __restrict__ mean,
                                                             const float* __restrict__ stat, const float* __restrict__ dmean,
                                                             const float* __restrict__ dvar, float* __restrict__ dheat,
                                                             int ld_d) {
    __shared__ float red_mem[WAVES > 1 ? WAVES * 8 : 1];
    PairRed<WAVES> pr{red_mem};
    constexpr int LANES = 64 * WAVES;
    const int tl = threadIdx.x % LANES;
    const int pair = blockIdx.x * (4 / WAVES) + threadIdx.x / LANES;
    if (pair >= pairs) return;
    const int n = pair / K, k = pair - n * K;
    const int P = H * W;
    const float* hp = heat + (long)n * P * ld + k;
    const float invW = 1.f / (float)W;
    float s[PPT];
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        const int p = j * LANES + tl;
        s[j] = p < P ? hp[(long)p * ld] : 0.f;
    }
    const long o = pair;
    const float mux = mean[o * 2], muy = mean[o * 2 + 1], mx = stat[o * 2], invS = 1.f / stat[o * 2 + 1];
    const float d00 = dvar[o * 4], d01 = dvar[o * 4 + 1], d10 = dvar[o * 4 + 2], d11 = dvar[o * 4 + 3];
    // c = sum_i w_i (g_i - mu) = -mu * P * 1e-7 (weights sum to 1 + P*1e-7); the centring of var feeds back into mean:
    // dmu_total = dmean - (dvar + dvar^T) c
    const float cx = -mux * (float)P * 1e-7f, cy = -muy * (float)P * 1e-7f;
    const float gmx = dmean[o * 2] - (2.f * d00 * cx + (d01 + d10) * cy);
    const float gmy = dmean[o * 2 + 1] - ((d01 + d10) * cx + 2.f * d11 * cy);
    const float v00 = d00, v01 = d01 + d10, v11 = d11;
    auto a_of = [&](int p) __attribute__((always_inline)) {
        const int py = (int)(((float)p + 0.5f) * invW), px = p - py * W;
        const float gx = grid_coord(px, W), gy = grid_coord(py, H);
        const float dx = gx - mux, dy = gy - muy;
        return gmx * gx + gmy * gy + v00 * dx * dx + v01 * dx * dy + v11 * dy * dy;
    };
    float A[1] = {0.f};
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        const int p = j * LANES + tl;
        const float sj = p < P ? expf(s[j] / temperature - mx) * invS : 0.f;
        s[j] = sj;
        A[0] += sj * a_of(p);
    }
    pr.sum(A);
    float* dp = dheat + (long)n * P * ld_d + k;
    const int npad = k == K - 1 ? ld_d - K : 0;      // the threads of the last channel also zero the pad channels
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        const int p = j * LANES + tl;
        if (p < P) {
            dp[(long)p * ld_d] = s[j] * (a_of(p) - A[0]) / temperature;
            for (int q = 1; q <= npad; ++q) dp[(long)p * ld_d + q] = 0.f;
        }
    }
}

// ---- gaussians ---------------------------------------------------------------------------------------------
struct Gauss {
    float mx, my, a00, a01, a10, a11;  // mean and inverse covariance
    __device__ __forceinline__ void load(const float* mean, const float* var, float const_var, long idx) {
        mx = mean[idx * 2];
        my = mean[idx * 2 + 1];
        if (var) {
            const float a = var[idx * 4], b = var[idx * 4 + 1], c = var[idx * 4 + 2], d = var[idx * 4 + 3];
            const float det = det_2x2(a, b, c, d);
            a00 = d / det;
            a01 = -b / det;
            a10 = -c / det;
            a11 = a / det;
        } else {
            a00 = a11 = 1.f / const_var;
            a01 = a10 = 0.f;
        }
    }
    // exp(-0.5 * d^T A d), evaluated as ((d^T A) d) like the reference's two matmuls (keypoint_detector.py:32)
    __device__ __forceinline__ float eval(float gx, float gy, float& dx, float& dy) const {
        dx = gx - mx;
        dy = gy - my;
        const float r0 = dx * a00 + dy * a10, r1 = dx * a01 + dy * a11;
        return expf(-0.5f * (r0 * dx + r1 * dy));
    }
};

__global__ void __launch_bounds__(256) gaussian_sums_kernel(const float* __restrict__ mean,
                                                            const float* __restrict__ var, float const_var, int h,
                                                            int w, float* __restrict__ sums) {
    __shared__ float red[4];
    Gauss g;
    g.load(mean, var, const_var, blockIdx.x);
    float acc = 0.f;
    for (int p = threadIdx.x; p < h * w; p += 256) {
        float dx, dy;
        acc += g.eval(grid_coord(p % w, w), grid_coord(p / w, h), dx, dy);
    }
    acc = block_sum_256(acc, red);
    if (threadIdx.x == 0) sums[blockIdx.x] = acc;
}

// bilinear sample of one NHWC pixel vector at normalised (x,y): zeros padding, align_corners=True
// (ATen grid_sampler_2d: ix = ((x+1)/2)*(W-1); weights nw,ne,sw,se)
struct Bilin {
    int x0, y0;
    float wnw, wne, wsw, wse;
    float tx, ty;   // ix - x0, iy - y0  (for the gradient)
    __device__ __forceinline__ void setup(float x, float y, int W, int H) {
        const float ix = ((x + 1.f) / 2.f) * (float)(W - 1), iy = ((y + 1.f) / 2.f) * (float)(H - 1);
        const float fx = floorf(ix), fy = floorf(iy);
        x0 = (int)fx;
        y0 = (int)fy;
        const float ex = fx + 1.f, ey = fy + 1.f;
        wnw = (ex - ix) * (ey - iy);
        wne = (ix - fx) * (ey - iy);
        wsw = (ex - ix) * (iy - fy);
        wse = (ix - fx) * (iy - fy);
        tx = ix - fx;
        ty = iy - fy;
    }
};

struct EmbedArgs {
    const float* img;
    int ld_img, Cimg;
    const float *mean_d, *var_d, *mean_s, *var_s;
    float const_var;
    int Nb, d, h, w, K, add_bg, use_heatmap, use_difference, use_deformed, heatmap_diff;
    float norm_const;
    const float *norm_d, *norm_s;
    int slots, per;
};

__global__ void __launch_bounds__(256) movement_embedding_fwd_kernel(EmbedArgs a, float* __restrict__ out, int ld_out) {
    const long P = (long)a.h * a.w;
    const long total = (long)a.Nb * a.d * P * a.slots;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int s = (int)(i % a.slots);
        const long fp = i / a.slots;
        const int p = (int)(fp % P);
        const int f = (int)(fp / P);
        const int b = f / a.d;
        const int k = s - a.add_bg;
        const int px = p % a.w, py = p / a.w;
        const float gx = grid_coord(px, a.w), gy = grid_coord(py, a.h);
        float* o = out + fp * ld_out + s * a.per;
        int j = 0;
        float ddx = 0.f, ddy = 0.f;
        if (k >= 0) {
            ddx = a.mean_s[((long)b * a.K + k) * 2] - a.mean_d[((long)f * a.K + k) * 2];
            ddy = a.mean_s[((long)b * a.K + k) * 2 + 1] - a.mean_d[((long)f * a.K + k) * 2 + 1];
        }
        if (a.use_heatmap) {
            float hv = 0.f;
            if (k >= 0) {
                Gauss g;
                float dx, dy;
                g.load(a.mean_d, a.var_d, a.const_var, (long)f * a.K + k);
                const float nd = a.norm_const > 0.f ? a.norm_const : a.norm_d[(long)f * a.K + k];
                hv = g.eval(gx, gy, dx, dy) / nd;
                if (a.heatmap_diff) {
                    g.load(a.mean_s, a.var_s, a.const_var, (long)b * a.K + k);
                    const float ns = a.norm_const > 0.f ? a.norm_const : a.norm_s[(long)b * a.K + k];
                    hv -= g.eval(gx, gy, dx, dy) / ns;
                }
            }
            o[j++] = hv;
        }
        if (a.use_difference) {
            o[j++] = ddx;
            o[j++] = ddy;
        }
        if (a.use_deformed) {
            Bilin bl;
            bl.setup(gx + ddx, gy + ddy, a.w, a.h);
            const float* ib = a.img + (long)b * P * a.ld_img;
            const bool x0ok = bl.x0 >= 0 && bl.x0 < a.w, x1ok = bl.x0 + 1 >= 0 && bl.x0 + 1 < a.w;
            const bool y0ok = bl.y0 >= 0 && bl.y0 < a.h, y1ok = bl.y0 + 1 >= 0 && bl.y0 + 1 < a.h;
            for (int c = 0; c < a.Cimg; ++c) {
                float v = 0.f;
                if (y0ok && x0ok) v += ib[((long)bl.y0 * a.w + bl.x0) * a.ld_img + c] * bl.wnw;
                if (y0ok && x1ok) v += ib[((long)bl.y0 * a.w + bl.x0 + 1) * a.ld_img + c] * bl.wne;
                if (y1ok && x0ok) v += ib[((long)(bl.y0 + 1) * a.w + bl.x0) * a.ld_img + c] * bl.wsw;
                if (y1ok && x1ok) v += ib[((long)(bl.y0 + 1) * a.w + bl.x0 + 1) * a.ld_img + c] * bl.wse;
                o[j++] = v;
            }
        }
        if (s == 0)
            for (int c = a.slots * a.per; c < ld_out; ++c) out[fp * ld_out + c] = 0.f;
    }
}

// one block per (frame, key-point): reduces the pixel gradients to d mean / d var of the driving and the
// source key-point (per frame; the caller sums the source gradients over the d frames of a batch entry)
__global__ void __launch_bounds__(256) movement_embedding_bwd_kernel(EmbedArgs a, const float* __restrict__ dout,
                                                                     int ld_out, float* __restrict__ dmean_d,
                                                                     float* __restrict__ dvar_d,
                                                                     float* __restrict__ dmean_s,
                                                                     float* __restrict__ dvar_s) {
    __shared__ float red_mem[4 * 14];
    BlockRed br{red_mem};
    const int f = blockIdx.x / a.K, k = blockIdx.x % a.K;
    const int b = f / a.d;
    const int s = k + a.add_bg;
    const int P = a.h * a.w;
    const float* gp = dout + (long)f * P * ld_out + s * a.per;
    Gauss gd, gs;
    gd.load(a.mean_d, a.var_d, a.const_var, (long)f * a.K + k);
    gs.load(a.mean_s, a.var_s, a.const_var, (long)b * a.K + k);
    const bool sum_norm = a.use_heatmap && !(a.norm_const > 0.f);
    const float nd = sum_norm ? a.norm_d[(long)f * a.K + k] : a.norm_const;
    const float ns = sum_norm ? (a.heatmap_diff ? a.norm_s[(long)b * a.K + k] : 1.f) : a.norm_const;
    const float ddx = a.mean_s[((long)b * a.K + k) * 2] - a.mean_d[((long)f * a.K + k) * 2];
    const float ddy = a.mean_s[((long)b * a.K + k) * 2 + 1] - a.mean_d[((long)f * a.K + k) * 2 + 1];
    // 'sum' normalisation: h = e/S  =>  dL/de_p = g_p/S - (sum_r g_r e_r)/S^2
    float T[2] = {0.f, 0.f};
    if (sum_norm) {
        for (int p = threadIdx.x; p < P; p += 256) {
            const float gx = grid_coord(p % a.w, a.w), gy = grid_coord(p / a.w, a.h);
            float dx, dy;
            const float g = gp[(long)p * ld_out];
            T[0] += g * gd.eval(gx, gy, dx, dy);
            if (a.heatmap_diff) T[1] += g * gs.eval(gx, gy, dx, dy);
        }
        br.reduce(T, 2);
    }
    // acc: 0-1 dmu_d, 2-5 GA_d, 6-7 dmu_s, 8-11 GA_s, 12-13 d(delta)
    float acc[14];
#pragma unroll
    for (int i = 0; i < 14; ++i) acc[i] = 0.f;
    const int off_diff = a.use_heatmap;
    const int off_img = a.use_heatmap + 2 * a.use_difference;
    for (int p = threadIdx.x; p < P; p += 256) {
        const int px = p % a.w, py = p / a.w;
        const float gx = grid_coord(px, a.w), gy = grid_coord(py, a.h);
        const float* g = gp + (long)p * ld_out;
        if (a.use_heatmap) {
            float dx, dy;
            float e = gd.eval(gx, gy, dx, dy);
            float ge = sum_norm ? g[0] / nd - T[0] / (nd * nd) : g[0] / nd;   // dL/de
            float c = -0.5f * e * ge;                                          // dL/dq
            acc[0] += -c * ((gd.a00 + gd.a00) * dx + (gd.a01 + gd.a10) * dy);
            acc[1] += -c * ((gd.a10 + gd.a01) * dx + (gd.a11 + gd.a11) * dy);
            acc[2] += c * dx * dx;
            acc[3] += c * dx * dy;
            acc[4] += c * dy * dx;
            acc[5] += c * dy * dy;
            if (a.heatmap_diff) {
                e = gs.eval(gx, gy, dx, dy);
                ge = -(sum_norm ? g[0] / ns - T[1] / (ns * ns) : g[0] / ns);
                c = -0.5f * e * ge;
                acc[6] += -c * ((gs.a00 + gs.a00) * dx + (gs.a01 + gs.a10) * dy);
                acc[7] += -c * ((gs.a10 + gs.a01) * dx + (gs.a11 + gs.a11) * dy);
                acc[8] += c * dx * dx;
                acc[9] += c * dx * dy;
                acc[10] += c * dy * dx;
                acc[11] += c * dy * dy;
            }
        }
        if (a.use_difference) {
            acc[12] += g[off_diff];
            acc[13] += g[off_diff + 1];
        }
        if (a.use_deformed) {
            Bilin bl;
            bl.setup(gx + ddx, gy + ddy, a.w, a.h);
            const float* ib = a.img + (long)b * P * a.ld_img;
            const bool x0ok = bl.x0 >= 0 && bl.x0 < a.w, x1ok = bl.x0 + 1 >= 0 && bl.x0 + 1 < a.w;
            const bool y0ok = bl.y0 >= 0 && bl.y0 < a.h, y1ok = bl.y0 + 1 >= 0 && bl.y0 + 1 < a.h;
            float gix = 0.f, giy = 0.f;
            for (int c = 0; c < a.Cimg; ++c) {
                const float go = g[off_img + c];
                const float nw = (y0ok && x0ok) ? ib[((long)bl.y0 * a.w + bl.x0) * a.ld_img + c] : 0.f;
                const float ne = (y0ok && x1ok) ? ib[((long)bl.y0 * a.w + bl.x0 + 1) * a.ld_img + c] : 0.f;
                const float sw = (y1ok && x0ok) ? ib[((long)(bl.y0 + 1) * a.w + bl.x0) * a.ld_img + c] : 0.f;
                const float se = (y1ok && x1ok) ? ib[((long)(bl.y0 + 1) * a.w + bl.x0 + 1) * a.ld_img + c] : 0.f;
                gix += go * ((ne - nw) * (1.f - bl.ty) + (se - sw) * bl.ty);
                giy += go * ((sw - nw) * (1.f - bl.tx) + (se - ne) * bl.tx);
            }
            acc[12] += gix * (float)(a.w - 1) * 0.5f;
            acc[13] += giy * (float)(a.h - 1) * 0.5f;
        }
    }
    br.reduce(acc, 14);
    if (threadIdx.x == 0) {
        const long od = (long)f * a.K + k;
        // delta = mean_s - mean_d
        dmean_d[od * 2] = acc[0] - acc[12];
        dmean_d[od * 2 + 1] = acc[1] - acc[13];
        dmean_s[od * 2] = acc[6] + acc[12];
        dmean_s[od * 2 + 1] = acc[7] + acc[13];
        // dSigma = -A^T G A^T
        if (dvar_d) {
            const Gauss* gg[2] = {&gd, &gs};
            float* outp[2] = {dvar_d + od * 4, dvar_s + od * 4};
            for (int w = 0; w < 2; ++w) {
                const float* G = acc + (w == 0 ? 2 : 8);
                const float t00 = gg[w]->a00, t01 = gg[w]->a10, t10 = gg[w]->a01, t11 = gg[w]->a11;  // A^T
                // M = A^T G
                const float m00 = t00 * G[0] + t01 * G[2], m01 = t00 * G[1] + t01 * G[3];
                const float m10 = t10 * G[0] + t11 * G[2], m11 = t10 * G[1] + t11 * G[3];
                outp[w][0] = -(m00 * t00 + m01 * t10);
                outp[w][1] = -(m00 * t01 + m01 * t11);
                outp[w][2] = -(m10 * t00 + m11 * t10);
                outp[w][3] = -(m10 * t01 + m11 * t11);
            }
        }
    }
}

static inline int grid_for(long total, int cap = 4096) {
    long b = (total + 255) / 256;
    if (b < 1) b = 1;
    return (int)(b < cap ? b : cap);
}

static int fill_embed_args(EmbedArgs& a, const float* img, int ld_img, int Cimg, const float* mean_d,
                           const float* var_d, const float* mean_s, const float* var_s, float const_var, int Nb, int d,
                           int h, int w, int K, int add_bg, int use_heatmap, int use_difference, int use_deformed,
                           int heatmap_diff, float norm_const, const float* norm_d, const float* norm_s) {
    a.img = img;
    a.ld_img = ld_img;
    a.Cimg = Cimg;
    a.mean_d = mean_d;
    a.var_d = var_d;
    a.mean_s = mean_s;
    a.var_s = var_s;
    a.const_var = const_var;
    a.Nb = Nb;
    a.d = d;
    a.h = h;
    a.w = w;
    a.K = K;
    a.add_bg = add_bg ? 1 : 0;
    a.use_heatmap = use_heatmap ? 1 : 0;
    a.use_difference = use_difference ? 1 : 0;
    a.use_deformed = use_deformed ? 1 : 0;
    a.heatmap_diff = heatmap_diff ? 1 : 0;
    a.norm_const = norm_const;
    a.norm_d = norm_d;
    a.norm_s = norm_s;
    a.slots = K + a.add_bg;
    a.per = a.use_heatmap + 2 * a.use_difference + (a.use_deformed ? Cimg : 0);
    return a.per;
}

// ---- clip_variance (keypoint_detector.py:62-65): var * max(clip, sigma_min(var)) / sigma_min(var), with the closed
// form of the singular values of a 2x2 matrix (modules/util.py:244-255): sigma^2 = (s1 -+ s2) / 2.  The reference takes
// sigma_min from the difference; for a nearly singular covariance (a heat-map stretched along a line: det ~ 1e-6 at entries
// ~ 0.4) sigma_min^2 is far below the rounding of s1, the difference is pure rounding noise -- zero or negative as often as
// not -- and the clipped variance comes out inf / NaN (measured: one of 24 initialisation seeds of a fixed-batch run went
// non-finite within four iterations).  Here sigma_max comes from the SUM (no cancellation) and sigma_min = |det| / sigma_max
// (the product of the singular values is |det|; det by Kahan's fma scheme): identical in exact arithmetic, equal to rounding
// where the reference's form is accurate, and equal to the fp64 reference where it is not.
// reference_mode: sigma_min by the reference's own closed form sqrt((s1 - s2) / 2) in its fp32 operation order
// (modules/util.py:244-255) -- the switch that keeps the reference's exact numbers (its NaNs on nearly singular covariances
// included) reachable for parity work; the default is the stable form.
__device__ __forceinline__ void sigma_2x2(float a, float b, float c, float d, float& s2, float& smax, float& sg,
                                          int reference_mode = 0) {
    const float s1 = a * a + b * b + c * c + d * d;
    const float t = a * a + b * b - c * c - d * d;
    const float u = a * c + b * d;
    s2 = sqrtf(t * t + 4.f * (u * u));
    smax = sqrtf((s1 + s2) * 0.5f);
    sg = reference_mode ? sqrtf((s1 - s2) / 2.f) : fabsf(det_2x2(a, b, c, d)) / smax;
}

__global__ void __launch_bounds__(256) kp_clip_var_fwd_kernel(const float* __restrict__ var, float clip, long M,
                                                              float* __restrict__ out, int reference_mode) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    const float4 v = *reinterpret_cast<const float4*>(var + i * 4);
    float s2, smax, sg;
    sigma_2x2(v.x, v.y, v.z, v.w, s2, smax, sg, reference_mode);
    const float mx = fmaxf(clip, sg);
    *reinterpret_cast<float4*>(out + i * 4) = make_float4((mx * v.x) / sg, (mx * v.y) / sg, (mx * v.z) / sg, (mx * v.w) / sg);
}

// dvar = dout * mx / sg + g_sg * d sigma_min / d var,  g_sg = sum_ij dout_ij v_ij ([sg > clip] / sg - mx / sg^2);
// sigma_min = |det| / sigma_max:  d sigma_min = sign(det) d det / sigma_max - sigma_min d sigma_max / sigma_max,
// d det = (d, -c, -b, a),  sigma_max = sqrt((s1 + s2) / 2): d sigma_max = (d s1 + d s2) / (4 sigma_max),
// d s2 = (t dt + 4 u du) / s2  -- sums only, no cancellation
__global__ void __launch_bounds__(256) kp_clip_var_bwd_kernel(const float* __restrict__ var, float clip, long M,
                                                              const float* __restrict__ dout, float* __restrict__ dvar,
                                                              int reference_mode) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    const float4 v = *reinterpret_cast<const float4*>(var + i * 4);
    const float4 g = *reinterpret_cast<const float4*>(dout + i * 4);
    const float a = v.x, b = v.y, c = v.z, d = v.w;
    float s2, smax, sg;
    sigma_2x2(a, b, c, d, s2, smax, sg, reference_mode);
    const float mx = fmaxf(clip, sg);
    const float dmx = sg > clip ? 1.f : (sg == clip ? 0.5f : 0.f);      // torch.max splits the gradient on ties
    const float gv = g.x * a + g.y * b + g.z * c + g.w * d;
    const float g_sg = gv * (dmx / sg - mx / (sg * sg));
    const float t = a * a + b * b - c * c - d * d, u = a * c + b * d;
    const float sgn = det_2x2(a, b, c, d) < 0.f ? -1.f : 1.f;
    // d sigma_max / d(a, b, c, d); an isotropic matrix has s2 = 0 and d s2 = 0 (the two singular values coincide)
    const float is2 = s2 > 0.f ? 1.f / s2 : 0.f, q = 1.f / (4.f * smax);
    const float ma = q * (2.f * a + (t * 2.f * a + 4.f * u * c) * is2);
    const float mb = q * (2.f * b + (t * 2.f * b + 4.f * u * d) * is2);
    const float mc = q * (2.f * c + (-t * 2.f * c + 4.f * u * a) * is2);
    const float md = q * (2.f * d + (-t * 2.f * d + 4.f * u * b) * is2);
    const float r = sg / smax, ism = sgn / smax;
    float da = ism * d - r * ma, db = -ism * c - r * mb, dc = -ism * b - r * mc, dd = ism * a - r * md;
    if (reference_mode) {
        // what autograd makes of sqrt((s1 - s2) / 2): d sigma_min = (d s1 - d s2) / (4 sigma_min),
        // d s1 = 2 (a, b, c, d), d s2 = (t dt + 4 u du) / s2
        const float qq = 1.f / (4.f * sg);
        da = qq * (2.f * a - (t * 2.f * a + 4.f * u * c) * is2);
        db = qq * (2.f * b - (t * 2.f * b + 4.f * u * d) * is2);
        dc = qq * (2.f * c - (-t * 2.f * c + 4.f * u * a) * is2);
        dd = qq * (2.f * d - (-t * 2.f * d + 4.f * u * b) * is2);
    }
    const float f = mx / sg;
    *reinterpret_cast<float4*>(dvar + i * 4) =
        make_float4(fmaf(g.x, f, g_sg * da), fmaf(g.y, f, g_sg * db), fmaf(g.z, f, g_sg * dc), fmaf(g.w, f, g_sg * dd));
}


// ---- transfer-time key-point normalisation (transfer.py:31-62 normalize_kp) on the device --------------------------------
// area of the convex hull of K <= 32 points (scipy.spatial.ConvexHull(points).volume in 2-D): Andrew's monotone chain and
// the shoelace formula, one thread (transfer.py:34-36 uses the hulls of the first frame's key-points only)
__global__ void kp_hull_area_kernel(const float* __restrict__ pts, int K, float* __restrict__ area) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float x[32], y[32];
    for (int i = 0; i < K; ++i) {
        x[i] = pts[2 * i];
        y[i] = pts[2 * i + 1];
    }
    for (int i = 1; i < K; ++i) {          // insertion sort by (x, y)
        const float xi = x[i], yi = y[i];
        int j = i - 1;
        while (j >= 0 && (x[j] > xi || (x[j] == xi && y[j] > yi))) {
            x[j + 1] = x[j];
            y[j + 1] = y[j];
            --j;
        }
        x[j + 1] = xi;
        y[j + 1] = yi;
    }
    int hull[66];
    int n = 0;
    auto cross = [&](int o, int a, int b) { return (x[a] - x[o]) * (y[b] - y[o]) - (y[a] - y[o]) * (x[b] - x[o]); };
    for (int i = 0; i < K; ++i) {          // lower hull
        while (n >= 2 && cross(hull[n - 2], hull[n - 1], i) <= 0.f) --n;
        hull[n++] = i;
    }
    const int lower = n + 1;
    for (int i = K - 2; i >= 0; --i) {     // upper hull
        while (n >= lower && cross(hull[n - 2], hull[n - 1], i) <= 0.f) --n;
        hull[n++] = i;
    }
    --n;                                   // the last point repeats the first
    float a2 = 0.f;
    for (int i = 0; i < n; ++i) {
        const int p = hull[i], q = hull[(i + 1) % n];
        a2 += x[p] * y[q] - x[q] * y[p];
    }
    *area = 0.5f * fabsf(a2);
}

// one thread per (batch entry b, driving frame f, key-point k):
//   mean' = (mean_v[b,f,k] - mean_v[b,0,k]) * sqrt(area_a / area_v) + mean_a[b,0,k]      (move_location; then clamp to [-1,1])
//   var'  = sym_posdef( var_v[b,f,k] * inverse(var_v[b,0,k]) * var_a[b,0,k] )            (adapt_variance)
// sym_posdef = make_symetric_matrix (transfer.py:17-28): (A + A^T) / 2 with eigenvalues <= 0 replaced by 1e-6 (closed form
// for 2x2: the matrix itself when both eigenvalues are positive)
__global__ void __launch_bounds__(256) kp_normalize_kernel(const float* __restrict__ mean_v, const float* __restrict__ var_v,
                                                           const float* __restrict__ mean_a, const float* __restrict__ var_a,
                                                           int B, int D, int K, const float* __restrict__ area_a,
                                                           const float* __restrict__ area_v, int move_location, int clip_mean,
                                                           int adapt_variance, float* __restrict__ mean_out,
                                                           float* __restrict__ var_out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * D * K) return;
    const int k = (int)(i % K);
    const int b = (int)(i / ((long)D * K));
    const long first = ((long)b * D) * K + k, app = (long)b * K + k;
    float mx = mean_v[2 * i], my = mean_v[2 * i + 1];
    if (move_location) {
        const float mult = (area_a && area_v) ? sqrtf(*area_a) / sqrtf(*area_v) : 1.f;
        mx = (mx - mean_v[2 * first]) * mult + mean_a[2 * app];
        my = (my - mean_v[2 * first + 1]) * mult + mean_a[2 * app + 1];
    }
    if (clip_mean) {
        mx = fminf(fmaxf(mx, -1.f), 1.f);
        my = fminf(fmaxf(my, -1.f), 1.f);
    }
    mean_out[2 * i] = mx;
    mean_out[2 * i + 1] = my;
    if (!var_out) return;
    float v00 = var_v[4 * i], v01 = var_v[4 * i + 1], v10 = var_v[4 * i + 2], v11 = var_v[4 * i + 3];
    if (adapt_variance) {
        const float f00 = var_v[4 * first], f01 = var_v[4 * first + 1], f10 = var_v[4 * first + 2], f11 = var_v[4 * first + 3];
        const float det = det_2x2(f00, f01, f10, f11);
        const float i00 = f11 / det, i01 = -f01 / det, i10 = -f10 / det, i11 = f00 / det;     // matrix_inverse, eps = 0
        const float t00 = v00 * i00 + v01 * i10, t01 = v00 * i01 + v01 * i11;
        const float t10 = v10 * i00 + v11 * i10, t11 = v10 * i01 + v11 * i11;
        const float a00 = var_a[4 * app], a01 = var_a[4 * app + 1], a10 = var_a[4 * app + 2], a11 = var_a[4 * app + 3];
        float r00 = t00 * a00 + t01 * a10, r01 = t00 * a01 + t01 * a11;
        float r10 = t10 * a00 + t11 * a10, r11 = t10 * a01 + t11 * a11;
        // symmetrise, then lift non-positive eigenvalues to 1e-6
        const float p = r00, q = 0.5f * (r01 + r10), r = r11;
        const float half = 0.5f * (p + r), dif = 0.5f * (p - r), rad = sqrtf(dif * dif + q * q);
        float l1 = half + rad, l2 = half - rad;
        r00 = p, r01 = r10 = q, r11 = r;
        if (l1 <= 0.f || l2 <= 0.f) {
            // unit eigenvector of l1: (q, l1 - p) or (l1 - r, q), whichever is better conditioned; l2's is orthogonal
            float ex = q, ey = l1 - p;
            if (fabsf(l1 - r) > fabsf(l1 - p)) ex = l1 - r, ey = q;
            const float nrm = sqrtf(ex * ex + ey * ey);
            if (nrm > 0.f) ex /= nrm, ey /= nrm; else ex = 1.f, ey = 0.f;
            const float d1 = l1 <= 0.f ? 1e-6f : l1, d2 = l2 <= 0.f ? 1e-6f : l2;
            r00 = d1 * ex * ex + d2 * ey * ey;
            r01 = r10 = (d1 - d2) * ex * ey;
            r11 = d1 * ey * ey + d2 * ex * ex;
        }
        v00 = r00, v01 = r01, v10 = r10, v11 = r11;
    }
    var_out[4 * i] = v00;
    var_out[4 * i + 1] = v01;
    var_out[4 * i + 2] = v10;
    var_out[4 * i + 3] = v11;
}

// key-point channels per block of the soft-argmax kernels (MNK_KP_GROUP).  16 = all channels of a frame in one block: 64
// blocks of one wave per SIMD for the 64 frames of an iteration, each doing 3 passes x 10 expf / divisions per pixel; 5 / 2 / 1
// channels per block measured -0.05 / -0.04 / -0.05 ms per iteration against that (visit 45) -> 5
static int g_kp_group = tuning_knob("kp_group", &g_kp_group, 5);
// 1: heat-maps up to 64 x 64 take the one-pass wavefront-per-(frame, key point) kernels; 0: the multi-pass block kernels
static int g_kp_onepass = tuning_knob("kp_onepass", &g_kp_onepass, 1);
static int kp_group(int K) {
    int kg = g_kp_group < 1 ? 1 : g_kp_group;
    return kg > K ? K : kg;
}


// ---- the integers of the key-point path (north star: "bit-exact keypoint indices") --------------------------------------
// heatmap_argmax: linear pixel index h * W + w of the largest heat-map logit per (frame, key point), first occurrence --
// the integer form of the soft-argmax of keypoint_detector.py:103-104 (soft-max and the division by the temperature are
// monotone, so this is the arg-max of the soft-max heat-map too).  One block per frame, key points in registers.
__global__ void __launch_bounds__(256) heatmap_argmax_kernel(const float* __restrict__ heat, int ld, int H, int W, int K,
                                                             int* __restrict__ index) {
    __shared__ float rv[4][MAXK];
    __shared__ int ri[4][MAXK];
    const int n = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int P = H * W;
    const float* hp = heat + (long)n * P * ld;
    float bv[MAXK];
    int bi[MAXK];
#pragma unroll
    for (int k = 0; k < MAXK; ++k) {
        bv[k] = -INFINITY;
        bi[k] = 0x7fffffff;
    }
    for (int p = t; p < P; p += 256) {            // ascending p per thread: `>` keeps the first occurrence
#pragma unroll
        for (int k = 0; k < MAXK; ++k)
            if (k < K) {
                const float v = hp[(long)p * ld + k];
                if (v > bv[k]) {
                    bv[k] = v;
                    bi[k] = p;
                }
            }
    }
#pragma unroll
    for (int k = 0; k < MAXK; ++k)
        if (k < K) {
            float v = bv[k];
            int i = bi[k];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const float ov = __shfl_xor(v, o);
                const int oi = __shfl_xor(i, o);
                if (ov > v || (ov == v && oi < i)) {
                    v = ov;
                    i = oi;
                }
            }
            if (lane == 0) {
                rv[wave][k] = v;
                ri[wave][k] = i;
            }
        }
    __syncthreads();
    if (t < K) {
        float v = rv[0][t];
        int i = ri[0][t];
#pragma unroll
        for (int w = 1; w < 4; ++w)
            if (rv[w][t] > v || (rv[w][t] == v && ri[w][t] < i)) {
                v = rv[w][t];
                i = ri[w][t];
            }
        index[(long)n * K + t] = i;
    }
}

// kp_pixel_index: the pixel a key point is drawn at -- floor(size * (mean + 1) / 2) per axis, the mapping of
// Visualizer.draw_video_with_kp (logger.py:99-100: `spatial_size * (kp_array + 1) / 2`, then rasterised), in numpy's
// arithmetic: `kp_array + 1` stays float32 (a float32 array plus a Python int), the int64 `spatial_size` array times that
// float32 array is promoted to float64, and so is the division.  (For frame sizes that are powers of two the fp32 product is
// exact and both orders agree; for e.g. 96 x 80 frames an fp32 product can round across a cell boundary.)
__global__ void __launch_bounds__(256) kp_pixel_index_kernel(const float* __restrict__ mean, long n, int W, int H,
                                                             int* __restrict__ pixel) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= 2 * n) return;
    const double size = (i & 1) ? (double)H : (double)W;
    const float t = mean[i] + 1.f;
    pixel[i] = (int)floor(size * (double)t / 2.0);
}

}  // namespace

extern "C" {

int mnk_softmax_kp_fwd(const float* heat, int ld, int N, int H, int W, int K, float temperature, float* mean,
                       float* var, float* stat, void* stream) {
    MNK_REQUIRE(heat && mean && var && stat && N > 0 && H > 1 && W > 1 && K > 0 && K <= MAXK && ld >= K);
    MNK_REQUIRE(temperature > 0.f);
    hipStream_t s = (hipStream_t)stream;
    // algorithmic bytes: one read of the K heat-map channels + the 8 output floats per key point
    ProfScope prof(K_KEYPOINT, s, (double)N * H * W * K * 4 + (double)N * K * 32);
    const int P = H * W, pairs = N * K;
    const dim3 grid((unsigned)((pairs + 3) / 4));
    if (g_kp_onepass && P <= 256)
        hipLaunchKernelGGL((softmax_kp_fwd_kernel<4, 1>), grid, dim3(256), 0, s, heat, ld, H, W, K, pairs, temperature, mean, var, stat);
    else if (g_kp_onepass && P <= 1024)
        hipLaunchKernelGGL((softmax_kp_fwd_kernel<16, 1>), grid, dim3(256), 0, s, heat, ld, H, W, K, pairs, temperature, mean, var, stat);
    else if (g_kp_onepass && P <= 4096)
        hipLaunchKernelGGL((softmax_kp_fwd_kernel<16, 4>), dim3((unsigned)pairs), dim3(256), 0, s, heat, ld, H, W, K, pairs, temperature,
                           mean, var, stat);
    else {
        const int kg = kp_group(K);
        hipLaunchKernelGGL(softmax_kp_fwd_multipass_kernel, dim3(N, (K + kg - 1) / kg), dim3(256), 0, s, heat, ld, H, W, K, kg,
                           temperature, mean, var, stat);
    }
    MNK_LAUNCH_CHECK();
    return MNK_OK;
}

int mnk_heatmap_argmax(const float* heat, int ld, int N, int H, int W, int K, int* index, void* stream) {
    MNK_REQUIRE(heat && index && N > 0 && H > 0 && W > 0 && K > 0 && K <= MAXK && ld >= K);
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(K_KEYPOINT, s, (double)N * H * W * K * 4);
    hipLaunchKernelGGL(heatmap_argmax_kernel, dim3(N), dim3(256), 0, s, heat, ld, H, W, K, index);
    MNK_LAUNCH_CHECK();
    return MNK_OK;
}

int mnk_kp_pixel_index(const float* mean, long n, int W, int H, int* pixel, void* stream) {
    MNK_REQUIRE(mean && pixel && n > 0 && W > 0 && H > 0);
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(K_KEYPOINT, s, (double)n * 16);
    hipLaunchKernelGGL(kp_pixel_index_kernel, dim3((unsigned)((2 * n + 255) / 256)), dim3(256), 0, s, mean, n, W, H, pixel);
    MNK_LAUNCH_CHECK();
    return MNK_OK;
}

int mnk_softmax_kp_bwd(const float* heat, int ld, int N, int H, int W, int K, float temperature, const float* mean,
                       const float* stat, const float* dmean, const float* dvar, float* dheat, int ld_d, void* stream) {
    MNK_REQUIRE(heat && mean && stat && dmean && dvar && dheat && N > 0 && H > 1 && W > 1 && K > 0 && K <= MAXK);
    MNK_REQUIRE(ld >= K && ld_d >= K && temperature > 0.f);
    hipStream_t s = (hipStream_t)stream;
    // algorithmic bytes: one read of the K heat-map channels, one write of the ld_d gradient channels
    ProfScope prof(K_KEYPOINT, s, (double)N * H * W * (K + ld_d) * 4);
    const int P = H * W, pairs = N * K;
    const dim3 grid((unsigned)((pairs + 3) / 4));
    if (g_kp_onepass && P <= 256)
        hipLaunchKernelGGL((softmax_kp_bwd_kernel<4, 1>), grid, dim3(256), 0, s, heat, ld, H, W, K, pairs, temperature, mean, stat, dmean,
                           dvar, dheat, ld_d);
    else if (g_kp_onepass && P <= 1024)
        hipLaunchKernelGGL((softmax_kp_bwd_kernel<16, 1>), grid, dim3(256), 0, s, heat, ld, H, W, K, pairs, temperature, mean, stat,
                           dmean, dvar, dheat, ld_d);
    else if (g_kp_onepass && P <= 4096)
        hipLaunchKernelGGL((softmax_kp_bwd_kernel<16, 4>), dim3((unsigned)pairs), dim3(256), 0, s, heat, ld, H, W, K, pairs, temperature,
                           mean, stat, dmean, dvar, dheat, ld_d);
    else {
        const int kg = kp_group(K);
        hipLaunchKernelGGL(softmax_kp_bwd_multipass_kernel, dim3(N, (K + kg - 1) / kg), dim3(256), 0, s, heat, ld, H, W, K, kg,
                           temperature, mean, stat, dmean, dvar, dheat, ld_d);
    }
    MNK_LAUNCH_CHECK();
    return MNK_OK;
}

int mnk_gaussian_sums(const float* mean, const float* var, float const_var, int Nkp, int h, int w, float* sums,
                      void* stream) {
    MNK_REQUIRE(mean && sums && Nkp > 0 && h > 1 && w > 1 && (var || const_var > 0.f));
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(K_EMBED, s, (double)Nkp * 4);
    hipLaunchKernelGGL(gaussian_sums_kernel, dim3(Nkp), dim3(256), 0, s, mean, var, const_var, h, w, sums);
    MNK_LAUNCH_CHECK();
    return MNK_OK;
}

int mnk_movement_embedding_fwd(const float* img, int ld_img, int Cimg, const float* mean_d, const float* var_d,
                               const float* mean_s, const float* var_s, float const_var, int Nb, int d, int h, int w,
                               int K, int add_bg, int use_heatmap, int use_difference, int use_deformed,
                               int heatmap_diff, float norm_const, const float* norm_d, const float* norm_s,
                               float* out, int ld_out, void* stream) {
    MNK_REQUIRE(mean_d && mean_s && out && Nb > 0 && d > 0 && h > 1 && w > 1 && K > 0);
    MNK_REQUIRE(!use_deformed || (img && ld_img >= Cimg && Cimg > 0));
    MNK_REQUIRE(!use_heatmap || ((var_d && var_s) || const_var > 0.f));
    MNK_REQUIRE(!use_heatmap || norm_const > 0.f || (norm_d && (!heatmap_diff || norm_s)));
    EmbedArgs a;
    int per = fill_embed_args(a, img, ld_img, Cimg, mean_d, var_d, mean_s, var_s, const_var, Nb, d, h, w, K, add_bg,
                              use_heatmap, use_difference, use_deformed, heatmap_diff, norm_const, norm_d, norm_s);
    MNK_REQUIRE(per > 0 && ld_out >= a.slots * per);
    hipStream_t s = (hipStream_t)stream;
    const long total = (long)Nb * d * h * w * a.slots;
    ProfScope prof(K_EMBED, s, (double)total * per * 4);
    hipLaunchKernelGGL(movement_embedding_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, s, a, out, ld_out);
    MNK_LAUNCH_CHECK();
    return MNK_OK;
}

int mnk_movement_embedding_bwd(const float* img, int ld_img, int Cimg, const float* mean_d, const float* var_d,
                               const float* mean_s, const float* var_s, float const_var, int Nb, int d, int h, int w,
                               int K, int add_bg, int use_heatmap, int use_difference, int use_deformed,
                               int heatmap_diff, float norm_const, const float* norm_d, const float* norm_s,
                               const float* dout, int ld_out, float* dmean_d, float* dvar_d, float* dmean_s,
                               float* dvar_s, void* stream) {
    MNK_REQUIRE(mean_d && mean_s && dout && dmean_d && dmean_s && Nb > 0 && d > 0 && h > 1 && w > 1 && K > 0);
    MNK_REQUIRE(!use_deformed || (img && ld_img >= Cimg && Cimg > 0));
    MNK_REQUIRE((dvar_d == nullptr) == (dvar_s == nullptr));
    MNK_REQUIRE(!use_heatmap || ((var_d && var_s) || const_var > 0.f));
    MNK_REQUIRE(!use_heatmap || norm_const > 0.f || (norm_d && (!heatmap_diff || norm_s)));
    EmbedArgs a;
    int per = fill_embed_args(a, img, ld_img, Cimg, mean_d, var_d, mean_s, var_s, const_var, Nb, d, h, w, K, add_bg,
                              use_heatmap, use_difference, use_deformed, heatmap_diff, norm_const, norm_d, norm_s);
    MNK_REQUIRE(per > 0 && ld_out >= a.slots * per);
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(K_EMBED, s, (double)Nb * d * h * w * K * per * 4);
    hipLaunchKernelGGL(movement_embedding_bwd_kernel, dim3(Nb * d * K), dim3(256), 0, s, a, dout, ld_out, dmean_d, dvar_d,
                       dmean_s, dvar_s);
    MNK_LAUNCH_CHECK();
    return MNK_OK;
}
int mnk_kp_clip_variance_fwd(const float* var, float clip, long M, float* out, int reference_mode, void* stream) {
    MNK_REQUIRE(var && out && M > 0 && clip > 0.f && ((size_t)var % 16) == 0 && ((size_t)out % 16) == 0);
    MNK_REQUIRE(reference_mode == 0 || reference_mode == 1);
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(K_KEYPOINT, s, (double)M * 32);
    hipLaunchKernelGGL(kp_clip_var_fwd_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, s, var, clip, M, out, reference_mode);
    MNK_LAUNCH_CHECK();
    return MNK_OK;
}

int mnk_kp_clip_variance_bwd(const float* var, float clip, long M, const float* dout, float* dvar, int reference_mode,
                             void* stream) {
    MNK_REQUIRE(var && dout && dvar && M > 0 && clip > 0.f && (reference_mode == 0 || reference_mode == 1));
    MNK_REQUIRE(((size_t)var % 16) == 0 && ((size_t)dout % 16) == 0 && ((size_t)dvar % 16) == 0);
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(K_KEYPOINT, s, (double)M * 48);
    hipLaunchKernelGGL(kp_clip_var_bwd_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, s, var, clip, M, dout, dvar, reference_mode);
    MNK_LAUNCH_CHECK();
    return MNK_OK;
}

int mnk_kp_hull_area(const float* points, int K, float* area, void* stream) {
    MNK_REQUIRE(points && area && K >= 3 && K <= 32);
    hipLaunchKernelGGL(kp_hull_area_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, points, K, area);
    MNK_LAUNCH_CHECK();
    return MNK_OK;
}

int mnk_kp_normalize(const float* mean_v, const float* var_v, const float* mean_a, const float* var_a, int B, int D, int K,
                     const float* area_a, const float* area_v, int move_location, int clip_mean, int adapt_variance,
                     float* mean_out, float* var_out, void* stream) {
    MNK_REQUIRE(mean_v && mean_a && mean_out && B > 0 && D > 0 && K > 0 && (!var_out || var_v));
    MNK_REQUIRE(!adapt_variance || (var_v && var_a && var_out));
    MNK_REQUIRE((area_a == nullptr) == (area_v == nullptr));
    hipStream_t s = (hipStream_t)stream;
    const long n = (long)B * D * K;
    ProfScope prof(K_KEYPOINT, s, (double)n * 48);
    hipLaunchKernelGGL(kp_normalize_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, mean_v, var_v, mean_a, var_a, B, D,
                       K, area_a, area_v, move_location, clip_mean, adapt_variance, mean_out, var_out);
    MNK_LAUNCH_CHECK();
    return MNK_OK;
}
}

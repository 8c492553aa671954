// Dense-motion head, bilinear warps and the small 1x1 convolutions of the generator.
//   mnk_gconv1x1_*          : SameBlock3D grouped 1x1 conv (modules/util.py:118; dense_motion_module.py:24-28)
//   mnk_conv1x1_sigmoid_*   : refinement 'conv-last' + torch.sigmoid (modules/generator.py:48,79-80)
//   mnk_motion_field_*      : mask softmax, sum_k m_k*delta_k + correction + identity grid (dense_motion_module.py:52-76)
//   mnk_deform_*            : MotionTransferGenerator.deform_input (generator.py:51-58): field resize + grid_sample
// All HBM-bound element-wise / small-reduction kernels.
#include <algorithm>

#include "mnk_common.h"

using namespace mnk;

namespace {

constexpr int MAXG = 8;    // channels per group of the grouped 1x1 conv
constexpr int MAXS = 17;   // mask slots (num_kp + 1)
constexpr int MAXCO = 4;   // output channels of the 1x1 + sigmoid conv

__device__ __forceinline__ float grid_coord(int j, int n) { return 2.f * ((float)j / (float)(n - 1)) - 1.f; }

static inline int grid_for(long total, int cap = 4096) {
    long b = (total + 255) / 256;
    if (b < 1) b = 1;
    return (int)(b < cap ? b : cap);
}

// ---- grouped 1x1 ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) gconv1x1_fwd_kernel(const float* __restrict__ x, int ld_x,
                                                           const float* __restrict__ w, const float* __restrict__ bias,
                                                           float* __restrict__ y, int ld_y, long rows, int G, int S,
                                                           int transpose) {
    const long total = rows * G;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int g = (int)(i % G);
        const long r = i / G;
        float xin[MAXG];
#pragma unroll
        for (int k = 0; k < MAXG; ++k) xin[k] = k < S ? x[r * ld_x + g * S + k] : 0.f;
#pragma unroll
        for (int o = 0; o < MAXG; ++o) {
            if (o >= S) continue;
            float acc = bias ? bias[g * S + o] : 0.f;
#pragma unroll
            for (int k = 0; k < MAXG; ++k)
                if (k < S) acc += xin[k] * (transpose ? w[(g * S + k) * S + o] : w[(g * S + o) * S + k]);
            y[r * ld_y + g * S + o] = acc;
        }
        if (g == 0)
            for (int c = G * S; c < ld_y; ++c) y[r * ld_y + c] = 0.f;
    }
}

// partial[rb][g][S*S + S]: dw then dbias of one group, reduced over the block's row range
__global__ void __launch_bounds__(256) gconv1x1_wgrad_partial_kernel(const float* __restrict__ x, int ld_x,
                                                                     const float* __restrict__ dy, int ld_dy, long rows,
                                                                     int G, int S, long rows_per_block,
                                                                     float* __restrict__ partial) {
    __shared__ float red[4 * (MAXG * MAXG + MAXG)];
    constexpr int NV = MAXG * MAXG + MAXG;
    const int g = blockIdx.y;
    const long r0 = (long)blockIdx.x * rows_per_block;
    long r1 = r0 + rows_per_block;
    if (r1 > rows) r1 = rows;
    float acc[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) acc[k] = 0.f;
    for (long r = r0 + threadIdx.x; r < r1; r += 256) {
        float xi[MAXG], go[MAXG];
#pragma unroll
        for (int k = 0; k < MAXG; ++k) {
            xi[k] = k < S ? x[r * ld_x + g * S + k] : 0.f;
            go[k] = k < S ? dy[r * ld_dy + g * S + k] : 0.f;
        }
#pragma unroll
        for (int o = 0; o < MAXG; ++o) {
#pragma unroll
            for (int k = 0; k < MAXG; ++k) acc[o * MAXG + k] += go[o] * xi[k];
            acc[MAXG * MAXG + o] += go[o];
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        float s = wave_sum(acc[k]);
        if (lane == 0) red[wave * NV + k] = s;
    }
    __syncthreads();
    if (threadIdx.x < NV) {
        const int k = threadIdx.x;
        partial[((long)blockIdx.x * G + g) * NV + k] = red[k] + red[NV + k] + red[2 * NV + k] + red[3 * NV + k];
    }
}

__global__ void __launch_bounds__(256) gconv1x1_wgrad_final_kernel(const float* __restrict__ partial, int row_blocks,
                                                                   int G, int S, float* __restrict__ dw,
                                                                   float* __restrict__ dbias) {
    constexpr int NV = MAXG * MAXG + MAXG;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int n_w = G * S * S, n_b = G * S;
    if (i >= n_w + n_b) return;
    int g, slot;
    if (i < n_w) {
        g = i / (S * S);
        const int rem = i - g * S * S;
        slot = (rem / S) * MAXG + rem % S;
    } else {
        const int j = i - n_w;
        g = j / S;
        slot = MAXG * MAXG + j % S;
    }
    float acc = 0.f;
    for (int rb = 0; rb < row_blocks; ++rb) acc += partial[((long)rb * G + g) * NV + slot];
    if (i < n_w)
        dw[i] = acc;
    else if (dbias)
        dbias[i - n_w] = acc;
}

// ---- 1x1 + sigmoid ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) conv1x1_sigmoid_fwd_kernel(const float* __restrict__ x, int ld_x, int Cin,
                                                                  const float* __restrict__ w,
                                                                  const float* __restrict__ bias,
                                                                  float* __restrict__ out, int B, int D, int H, int W,
                                                                  int Cout, int act) {
    const long HW = (long)H * W;
    const long rows = (long)B * D * HW;
    for (long r = (long)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += (long)gridDim.x * blockDim.x) {
        float acc[MAXCO];
#pragma unroll
        for (int c = 0; c < MAXCO; ++c) acc[c] = (bias && c < Cout) ? bias[c] : 0.f;
        const float* xr = x + r * ld_x;
        for (int k = 0; k < Cin; ++k) {
            const float v = xr[k];
#pragma unroll
            for (int c = 0; c < MAXCO; ++c)
                if (c < Cout) acc[c] = fmaf(v, w[c * Cin + k], acc[c]);
        }
        const long hw = r % HW;
        const long f = r / HW;
        const int d = (int)(f % D);
        const long b = f / D;
#pragma unroll
        for (int c = 0; c < MAXCO; ++c)
            if (c < Cout) out[((b * Cout + c) * D + d) * HW + hw] = act ? 1.f / (1.f + expf(-acc[c])) : acc[c];
    }
}

// dpre = dout * o * (1 - o); dx[r][k] = sum_c w[c][k] dpre_c
__global__ void __launch_bounds__(256) conv1x1_sigmoid_bwd_dx_kernel(const float* __restrict__ w,
                                                                     const float* __restrict__ out,
                                                                     const float* __restrict__ dout,
                                                                     float* __restrict__ dx, int ld_dx, int Cin, int B,
                                                                     int D, int H, int W, int Cout, int act) {
    const long HW = (long)H * W;
    const long rows = (long)B * D * HW;
    for (long r = (long)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += (long)gridDim.x * blockDim.x) {
        const long hw = r % HW;
        const long f = r / HW;
        const int d = (int)(f % D);
        const long b = f / D;
        float dp[MAXCO];
#pragma unroll
        for (int c = 0; c < MAXCO; ++c) {
            dp[c] = 0.f;
            if (c < Cout) {
                const long o = ((b * Cout + c) * D + d) * HW + hw;
                const float s = out[o];
                dp[c] = act ? dout[o] * s * (1.f - s) : dout[o];
            }
        }
        float* xr = dx + r * ld_dx;
        for (int k = 0; k < ld_dx; ++k) {
            float v = 0.f;
            if (k < Cin) {
#pragma unroll
                for (int c = 0; c < MAXCO; ++c)
                    if (c < Cout) v = fmaf(dp[c], w[c * Cin + k], v);
            }
            xr[k] = v;
        }
    }
}

// partial[rb][c][Cin + 1]: dw row c then dbias, threads (tx = input channel (+1 bias column), ty = row lane)
__global__ void __launch_bounds__(256) conv1x1_sigmoid_wgrad_partial_kernel(const float* __restrict__ x, int ld_x,
                                                                            int Cin, const float* __restrict__ out,
                                                                            const float* __restrict__ dout, int B, int D,
                                                                            int H, int W, int Cout, long rows_per_block,
                                                                            float* __restrict__ partial, int act) {
    __shared__ float red[MAXCO][256];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;   // 64 columns x 4 row lanes
    const long HW = (long)H * W;
    const long rows = (long)B * D * HW;
    const long r0 = (long)blockIdx.x * rows_per_block;
    long r1 = r0 + rows_per_block;
    if (r1 > rows) r1 = rows;
    for (int k0 = 64 * blockIdx.y; k0 <= Cin; k0 += 64 * gridDim.y) {       // wide layers: the 64-column chunks are blocks
        const int k = k0 + tx;
        float acc[MAXCO];
#pragma unroll
        for (int c = 0; c < MAXCO; ++c) acc[c] = 0.f;
        if (k <= Cin) {
            for (long r = r0 + ty; r < r1; r += 4) {
                const float xv = k < Cin ? x[r * ld_x + k] : 1.f;
                const long hw = r % HW;
                const long f = r / HW;
                const int d = (int)(f % D);
                const long b = f / D;
#pragma unroll
                for (int c = 0; c < MAXCO; ++c)
                    if (c < Cout) {
                        const long o = ((b * Cout + c) * D + d) * HW + hw;
                        const float s = out[o];
                        acc[c] = fmaf(act ? dout[o] * s * (1.f - s) : dout[o], xv, acc[c]);
                    }
            }
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < MAXCO; ++c) red[c][threadIdx.x] = acc[c];
        __syncthreads();
        if (ty == 0 && k <= Cin) {
#pragma unroll
            for (int c = 0; c < MAXCO; ++c)
                if (c < Cout)
                    partial[((long)blockIdx.x * Cout + c) * (Cin + 1) + k] =
                        red[c][tx] + red[c][tx + 64] + red[c][tx + 128] + red[c][tx + 192];
        }
    }
}

// one wavefront per output element: lanes stride over the row blocks, fixed-order wave reduction (deterministic)
__global__ void __launch_bounds__(256) conv1x1_sigmoid_wgrad_final_kernel(const float* __restrict__ partial,
                                                                          int row_blocks, int Cin, int Cout,
                                                                          float* __restrict__ dw,
                                                                          float* __restrict__ dbias) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= Cout * (Cin + 1)) return;
    const int c = i / (Cin + 1), k = i - c * (Cin + 1);
    float acc = 0.f;
    for (int rb = lane; rb < row_blocks; rb += 64) acc += partial[((long)rb * Cout + c) * (Cin + 1) + k];
    acc = wave_sum(acc);
    if (lane == 0) {
        if (k < Cin)
            dw[c * Cin + k] = acc;
        else if (dbias)
            dbias[c] = acc;
    }
}

// ---- 1x1, few rows x many channels (the discriminator's score head: 64 rows x 512 channels) ---------------------------
// one wavefront per pixel row (a thread per row walks 512 channels alone: 38 us for 128 KB of input)
__global__ void __launch_bounds__(256) conv1x1_wave_fwd_kernel(const float* __restrict__ x, int ld_x, int Cin,
                                                               const float* __restrict__ w, const float* __restrict__ bias,
                                                               float* __restrict__ out, int B, int D, int H, int W, int Cout,
                                                               int act) {
    const long HW = (long)H * W, rows = (long)B * D * HW;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (r >= rows) return;
    float acc[MAXCO];
#pragma unroll
    for (int c = 0; c < MAXCO; ++c) acc[c] = 0.f;
    const float* xr = x + r * ld_x;
    for (int k = lane; k < Cin; k += 64) {
        const float v = xr[k];
#pragma unroll
        for (int c = 0; c < MAXCO; ++c)
            if (c < Cout) acc[c] = fmaf(v, w[c * Cin + k], acc[c]);
    }
    const long hw = r % HW, f = r / HW;
    const int d = (int)(f % D);
    const long b = f / D;
#pragma unroll
    for (int c = 0; c < MAXCO; ++c) {
        if (c >= Cout) continue;
        float v = wave_sum(acc[c]);
        if (lane == 0) {
            v += bias ? bias[c] : 0.f;
            out[((b * Cout + c) * D + d) * HW + hw] = act ? 1.f / (1.f + expf(-v)) : v;
        }
    }
}

// dx, a thread per element (coalesced along the channels)
__global__ void __launch_bounds__(256) conv1x1_elem_bwd_dx_kernel(const float* __restrict__ w, const float* __restrict__ out,
                                                                  const float* __restrict__ dout, float* __restrict__ dx,
                                                                  int ld_dx, int Cin, int B, int D, int H, int W, int Cout,
                                                                  int act) {
    const long HW = (long)H * W, total = (long)B * D * HW * ld_dx;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / ld_dx;
        const int k = (int)(i - r * ld_dx);
        const long hw = r % HW, f = r / HW;
        const int d = (int)(f % D);
        const long b = f / D;
        float v = 0.f;
        if (k < Cin) {
#pragma unroll
            for (int c = 0; c < MAXCO; ++c)
                if (c < Cout) {
                    const long o = ((b * Cout + c) * D + d) * HW + hw;
                    const float sg = out[o];
                    v = fmaf(act ? dout[o] * sg * (1.f - sg) : dout[o], w[c * Cin + k], v);
                }
        }
        dx[i] = v;
    }
}

// ---- 1x1 (+ sigmoid), row-tile forms ---------------------------------------------------------------------------------
// The thread-per-pixel kernels above read a pixel's channel vector with scalar loads at a row stride across the wavefront
// (measured 33 us for a 25 MB input: 0.75 TB/s).  Here a block stages C11_TR whole pixel rows -- one contiguous piece of
// the NHWC tensor -- in LDS with coalesced 16-byte loads and works on the LDS copy; the backward does the data gradient and
// the weight-gradient partial of its rows from one staged tile (the thread-per-pixel backward needed two kernels that both
// re-read out / dout).  Used when ld_x is a multiple of 4, <= C11_MAXLD, and the tensors are 16-byte aligned.
constexpr int C11_TR = 128;                  // pixel rows per tile
constexpr int C11_MAXLD = 64;                // widest staged row (floats)
constexpr int C11_LDT = C11_MAXLD + 4;       // LDS row pitch bound

__device__ __forceinline__ void c11_stage(const float* __restrict__ x, int ld_x, long r0, int nrows, float* tile, int ldp) {
    const int qpr = ld_x >> 2, nq = nrows * qpr;
    const float4* src = reinterpret_cast<const float4*>(x + r0 * ld_x);
    for (int i = threadIdx.x; i < nq; i += 256) {
        const float4 v = src[i];
        const int row = i / qpr, q = i - row * qpr;
        *reinterpret_cast<float4*>(&tile[row * ldp + 4 * q]) = v;
    }
}

__global__ void __launch_bounds__(256) conv1x1_rows_fwd_kernel(const float* __restrict__ x, int ld_x, int Cin,
                                                               const float* __restrict__ w, const float* __restrict__ bias,
                                                               float* __restrict__ out, int B, int D, int H, int W, int Cout,
                                                               int act) {
    __shared__ __attribute__((aligned(16))) float tile[C11_TR * C11_LDT];
    const int ldp = ld_x + 4;
    const long HW = (long)H * W, rows = (long)B * D * HW;
    const long r0 = (long)blockIdx.x * C11_TR;
    const int nrows = rows - r0 < C11_TR ? (int)(rows - r0) : C11_TR;
    c11_stage(x, ld_x, r0, nrows, tile, ldp);
    __syncthreads();
    // thread = (row, half): the two halves of the block own output channels {0, 1} and {2, 3}
    const int row = threadIdx.x & (C11_TR - 1), c0 = (threadIdx.x >> 7) * 2;
    if (row >= nrows || c0 >= Cout) return;
    const bool two = c0 + 1 < Cout;
    const float* w0 = w + c0 * Cin;
    const float* w1 = w + (two ? c0 + 1 : c0) * Cin;
    float a0 = bias ? bias[c0] : 0.f, a1 = (bias && two) ? bias[c0 + 1] : 0.f;
    const float* xr = tile + row * ldp;
    for (int k = 0; k < Cin; ++k) {
        const float v = xr[k];
        a0 = fmaf(v, w0[k], a0);
        a1 = fmaf(v, w1[k], a1);
    }
    const long r = r0 + row, hw = r % HW, f = r / HW;
    const int d = (int)(f % D);
    const long b = f / D;
    out[((b * Cout + c0) * D + d) * HW + hw] = act ? 1.f / (1.f + expf(-a0)) : a0;
    if (two) out[((b * Cout + c0 + 1) * D + d) * HW + hw] = act ? 1.f / (1.f + expf(-a1)) : a1;
}

// dx rows + the weight-gradient partial of the block's row range [blockIdx.x * rows_per_block, ...), tile by tile.
// partial[block][c][Cin + 1] as conv1x1_sigmoid_wgrad_partial_kernel writes it (finished by ..._wgrad_final_kernel).
__global__ void __launch_bounds__(256) conv1x1_rows_bwd_kernel(const float* __restrict__ x, int ld_x, int Cin,
                                                               const float* __restrict__ w, const float* __restrict__ out,
                                                               const float* __restrict__ dout, float* __restrict__ dx, int B,
                                                               int D, int H, int W, int Cout, int act, long rows_per_block,
                                                               float* __restrict__ partial) {
    __shared__ __attribute__((aligned(16))) float tile[C11_TR * C11_LDT];
    __shared__ float dps[MAXCO][C11_TR];
    __shared__ float red[MAXCO][256];
    const int ldp = ld_x + 4;
    const long HW = (long)H * W, rows = (long)B * D * HW;
    const long rb0 = (long)blockIdx.x * rows_per_block;
    long rb1 = rb0 + rows_per_block;
    if (rb1 > rows) rb1 = rows;
    const int t = threadIdx.x;
    const int row = t & (C11_TR - 1), half = t >> 7;          // data-gradient map: (row, half of the channels)
    const int kx = t & 63, rl = t >> 6;                       // weight-gradient map: column (input channel / bias), row lane
    const int kq = (ld_x >> 2), kq0 = half * ((kq + 1) >> 1), kq1 = half ? kq : ((kq + 1) >> 1);   // this half's channel quads
    float acc[2][MAXCO];                                       // columns kx and kx + 64 (Cin + 1 <= 128)
#pragma unroll
    for (int c = 0; c < MAXCO; ++c) acc[0][c] = acc[1][c] = 0.f;
    for (long r0 = rb0; r0 < rb1; r0 += C11_TR) {
        const int nrows = rb1 - r0 < C11_TR ? (int)(rb1 - r0) : C11_TR;
        c11_stage(x, ld_x, r0, nrows, tile, ldp);
        if (half == 0) {                                       // dpre = dout * o * (1 - o) of this row, every channel
            const long r = r0 + row, hw = r % HW, f = r / HW;
            const int d = (int)(f % D);
            const long b = f / D;
#pragma unroll
            for (int c = 0; c < MAXCO; ++c) {
                float v = 0.f;
                if (c < Cout && row < nrows) {
                    const long o = ((b * Cout + c) * D + d) * HW + hw;
                    const float sg = out[o];
                    v = act ? dout[o] * sg * (1.f - sg) : dout[o];
                }
                dps[c][row] = v;
            }
        }
        __syncthreads();
        // weight gradient: acc[c] += sum over the tile's rows of dpre[c][r] * x[r][k]  (k == Cin: the bias column)
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            const int k = kx + 64 * pass;
            if (k <= Cin) {
                for (int r = rl; r < nrows; r += 4) {
                    const float xv = k < Cin ? tile[r * ldp + k] : 1.f;
#pragma unroll
                    for (int c = 0; c < MAXCO; ++c)
                        if (c < Cout) acc[pass][c] = fmaf(dps[c][r], xv, acc[pass][c]);
                }
            }
        }
        __syncthreads();
        // data gradient into the tile (the x copy is dead): dx[r][k] = sum_c w[c][k] * dpre[c][r], pad channels zero
        if (row < nrows) {
            float dp[MAXCO];
#pragma unroll
            for (int c = 0; c < MAXCO; ++c) dp[c] = dps[c][row];
            for (int q = kq0; q < kq1; ++q) {
                float v4[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int k = 4 * q + e;
                    float v = 0.f;
                    if (k < Cin) {
#pragma unroll
                        for (int c = 0; c < MAXCO; ++c)
                            if (c < Cout) v = fmaf(dp[c], w[c * Cin + k], v);
                    }
                    v4[e] = v;
                }
                *reinterpret_cast<float4*>(&tile[row * ldp + 4 * q]) = make_float4(v4[0], v4[1], v4[2], v4[3]);
            }
        }
        __syncthreads();
        {
            const int qpr = ld_x >> 2, nq = nrows * qpr;
            float4* dst = reinterpret_cast<float4*>(dx + r0 * ld_x);
            for (int i = t; i < nq; i += 256) {
                const int rr = i / qpr, q = i - rr * qpr;
                dst[i] = *reinterpret_cast<const float4*>(&tile[rr * ldp + 4 * q]);
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int k = kx + 64 * pass;
#pragma unroll
        for (int c = 0; c < MAXCO; ++c) red[c][t] = acc[pass][c];
        __syncthreads();
        if (rl == 0 && k <= Cin) {
#pragma unroll
            for (int c = 0; c < MAXCO; ++c)
                if (c < Cout)
                    partial[((long)blockIdx.x * Cout + c) * (Cin + 1) + k] =
                        (red[c][kx] + red[c][kx + 64]) + (red[c][kx + 128] + red[c][kx + 192]);
        }
        __syncthreads();
    }
}

// ---- motion field ------------------------------------------------------------------------------------------------
// delta[n][k][c] either stored (slot 0 = background = 0) or taken from the key points: kp_s.mean - kp_d.mean ([N][K][2] each)
struct DeltaSrc {
    const float *delta, *kp_s, *kp_d;
    __device__ __forceinline__ float operator()(long n, int S, int k, int c) const {
        if (delta) return delta[(n * S + k) * 2 + c];
        if (k == 0) return 0.f;
        const long o = (n * (S - 1) + (k - 1)) * 2 + c;
        return kp_s[o] - kp_d[o];
    }
};

__global__ void __launch_bounds__(256) motion_field_fwd_kernel(const float* __restrict__ pred, int ld,
                                                               DeltaSrc delta, int N, int h, int w,
                                                               int K, int use_mask, int use_corr,
                                                               float* __restrict__ field) {
    const long P = (long)h * w;
    const long total = (long)N * P;
    const int S = K + 1;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int p = (int)(i % P);
        const long n = i / P;
        const float* pr = pred + i * ld;
        float fx = 0.f, fy = 0.f;
        if (use_mask) {
            float mx = -INFINITY;
            for (int k = 0; k < S; ++k) mx = fmaxf(mx, pr[k]);
            float den = 0.f;
            for (int k = 0; k < S; ++k) den += expf(pr[k] - mx);
            for (int k = 0; k < S; ++k) {
                const float m = expf(pr[k] - mx) / den;
                fx += delta(n, S, k, 0) * m;
                fy += delta(n, S, k, 1) * m;
            }
        }
        if (use_corr) {
            const int o = use_mask ? S : 0;
            fx += pr[o];
            fy += pr[o + 1];
        }
        field[i * 2] = fx + grid_coord(p % w, w);
        field[i * 2 + 1] = fy + grid_coord(p / w, h);
    }
}

// one block per frame: writes dpred for its pixels and reduces ddelta[n][k][2] = sum_p m_k(p) * dfield(p)
// one block per frame (the mask's delta gradient is a per-frame sum): 1024 threads, because a batch is only 32 blocks and a
// pixel costs S exponentials (256 threads, three exponentials per slot: 40 us)
constexpr int MF_THREADS = 1024;
// (dkp_s / dkp_d given: the gradient goes to the key points instead, [N][K][2] each: +sum and -sum of slots 1..K)
__global__ void __launch_bounds__(MF_THREADS) motion_field_bwd_kernel(const float* __restrict__ pred, int ld,
                                                                      DeltaSrc delta,
                                                                      const float* __restrict__ dfield, int h, int w, int K,
                                                                      int use_mask, int use_corr, float* __restrict__ dpred,
                                                                      int ld_d, float* __restrict__ ddelta,
                                                                      float* __restrict__ dkp_s, float* __restrict__ dkp_d) {
    __shared__ float red[(MF_THREADS / 64) * 2 * MAXS];
    const int n = blockIdx.x;
    const int P = h * w, S = K + 1;
    float dd[2 * MAXS];
#pragma unroll
    for (int k = 0; k < 2 * MAXS; ++k) dd[k] = 0.f;
    for (int p = threadIdx.x; p < P; p += MF_THREADS) {
        const long i = (long)n * P + p;
        const float* pr = pred + i * ld;
        float* dp = dpred + i * ld_d;
        const float gx = dfield[i * 2], gy = dfield[i * 2 + 1];
        int o = 0;
        if (use_mask) {
            float e[MAXS];
            float mx = -INFINITY;
#pragma unroll
            for (int k = 0; k < MAXS; ++k)
                if (k < S) {
                    e[k] = pr[k];
                    mx = fmaxf(mx, e[k]);
                }
            float den = 0.f;
#pragma unroll
            for (int k = 0; k < MAXS; ++k)
                if (k < S) {
                    e[k] = expf(e[k] - mx);
                    den += e[k];
                }
            float dot = 0.f;
#pragma unroll
            for (int k = 0; k < MAXS; ++k)
                if (k < S) {
                    const float m = e[k] / den;
                    dot += m * (delta(n, S, k, 0) * gx + delta(n, S, k, 1) * gy);
                }
#pragma unroll
            for (int k = 0; k < MAXS; ++k)
                if (k < S) {
                    const float m = e[k] / den;
                    const float dm = delta(n, S, k, 0) * gx + delta(n, S, k, 1) * gy;
                    dp[k] = m * (dm - dot);
                    dd[2 * k] += m * gx;
                    dd[2 * k + 1] += m * gy;
                }
            o = S;
        }
        if (use_corr) {
            dp[o] = gx;
            dp[o + 1] = gy;
            o += 2;
        }
        for (int c = o; c < ld_d; ++c) dp[c] = 0.f;
    }
    if (use_mask) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
        for (int k = 0; k < 2 * MAXS; ++k)
            if (k < 2 * S) {
                float s = wave_sum(dd[k]);
                if (lane == 0) red[wave * 2 * MAXS + k] = s;
            }
        __syncthreads();
        if (threadIdx.x < 2 * S) {
            const int k = threadIdx.x;
            float s = 0.f;
            for (int wv = 0; wv < MF_THREADS / 64; ++wv) s += red[wv * 2 * MAXS + k];
            if (ddelta) ddelta[(long)n * S * 2 + k] = s;
            if (dkp_s && k >= 2) {
                dkp_s[(long)n * K * 2 + k - 2] = s;
                dkp_d[(long)n * K * 2 + k - 2] = -s;
            }
        }
    } else if (threadIdx.x < 2 * S && ddelta) {
        ddelta[(long)n * S * 2 + threadIdx.x] = 0.f;
    }
}

// ---- deform (field resize + bilinear grid_sample) ------------------------------------------------------------------
__device__ __forceinline__ int nearest_src(int dst, int in_size, int out_size) {
    float scale = (float)in_size / (float)out_size;
    int s = (int)floorf((float)dst * scale);
    return s < in_size - 1 ? s : in_size - 1;
}

// ATen area_pixel_compute_source_index, align_corners=False: max(0, scale*(dst+0.5)-0.5)
struct Lin1D {
    int i0, i1;
    float l0, l1;
    __device__ __forceinline__ void setup(int dst, int in_size, int out_size) {
        const float scale = (float)in_size / (float)out_size;
        float src = scale * ((float)dst + 0.5f) - 0.5f;
        if (src < 0.f) src = 0.f;
        i0 = (int)src;
        if (i0 > in_size - 1) i0 = in_size - 1;
        i1 = i0 < in_size - 1 ? i0 + 1 : i0;
        l1 = src - (float)i0;
        l0 = 1.f - l1;
    }
};

struct FieldAt {
    float x, y;
    // contributions of the resized field value to the original field texels (for the backward)
    int idx[4];
    float wgt[4];
    int n;
    Lin1D ly, lx;
    // which field texels pixel (py, px) of an (h, w) map reads, and with which weights
    __device__ __forceinline__ void setup(int hf, int wf, int py, int px, int h, int w, int mode) {
        if (mode == 0) {
            const int ys = nearest_src(py, hf, h), xs = nearest_src(px, wf, w);
            idx[0] = ys * wf + xs;
            wgt[0] = 1.f;
            n = 1;
        } else {
            ly.setup(py, hf, h);
            lx.setup(px, wf, w);
            idx[0] = ly.i0 * wf + lx.i0;
            wgt[0] = ly.l0 * lx.l0;
            idx[1] = ly.i0 * wf + lx.i1;
            wgt[1] = ly.l0 * lx.l1;
            idx[2] = ly.i1 * wf + lx.i0;
            wgt[2] = ly.l1 * lx.l0;
            idx[3] = ly.i1 * wf + lx.i1;
            wgt[3] = ly.l1 * lx.l1;
            n = 4;
        }
    }
    __device__ __forceinline__ void eval(const float* __restrict__ field, long nimg, int hf, int wf, int py, int px,
                                         int h, int w, int mode) {
        const float* fb = field + nimg * hf * wf * 2;
        setup(hf, wf, py, px, h, w, mode);
        if (mode == 0) {
            x = fb[idx[0] * 2];
            y = fb[idx[0] * 2 + 1];
        } else {
            // ATen upsample_bilinear2d: l0y*(l0x*v00 + l1x*v01) + l1y*(l0x*v10 + l1x*v11)
            x = ly.l0 * (lx.l0 * fb[idx[0] * 2] + lx.l1 * fb[idx[1] * 2]) +
                ly.l1 * (lx.l0 * fb[idx[2] * 2] + lx.l1 * fb[idx[3] * 2]);
            y = ly.l0 * (lx.l0 * fb[idx[0] * 2 + 1] + lx.l1 * fb[idx[1] * 2 + 1]) +
                ly.l1 * (lx.l0 * fb[idx[2] * 2 + 1] + lx.l1 * fb[idx[3] * 2 + 1]);
        }
    }
};

struct Bilin {
    int x0, y0;
    float wnw, wne, wsw, wse, tx, ty, ix, iy;
    bool x0ok, x1ok, y0ok, y1ok;
    __device__ __forceinline__ void setup(float x, float y, int W, int H) {
        ix = ((x + 1.f) / 2.f) * (float)(W - 1), iy = ((y + 1.f) / 2.f) * (float)(H - 1);
        const float fx = floorf(ix), fy = floorf(iy);
        // clamp before the int conversion so that far-away / non-finite coordinates stay out of range
        x0 = (fx >= -2.f && fx <= (float)W) ? (int)fx : -2;
        y0 = (fy >= -2.f && fy <= (float)H) ? (int)fy : -2;
        const float ex = fx + 1.f, ey = fy + 1.f;
        wnw = (ex - ix) * (ey - iy);
        wne = (ix - fx) * (ey - iy);
        wsw = (ex - ix) * (iy - fy);
        wse = (ix - fx) * (iy - fy);
        tx = ix - fx;
        ty = iy - fy;
        x0ok = x0 >= 0 && x0 < W;
        x1ok = x0 + 1 >= 0 && x0 + 1 < W;
        y0ok = y0 >= 0 && y0 < H;
        y1ok = y0 + 1 >= 0 && y0 + 1 < H;
    }
};

// (vb, vgrid): this block's index / the block count of the work item -- blockIdx.x / gridDim.x of a launch of its own, a
// sub-range of the blocks of a multi-level launch (warp_levels_*_kernel)
__device__ __forceinline__ void deform_fwd_body(const float* __restrict__ inp, int ld_in, int C, int h, int w,
                                                const float* __restrict__ field, int hf, int wf, int mode,
                                                float* __restrict__ out, int ld_out, int out_off, int N, int vb, int vgrid,
                                                bool zero_tail = false) {
    // zero_tail: nothing follows the C channels in the row (no embedding): the last quad's pad channels are written 0
    const int nq = (C + 3) / 4;
    const long P = (long)h * w;
    const long total = (long)N * P * nq;
    const bool vec_store = ((out_off & 3) == 0) && ((ld_out & 3) == 0);
    for (long i = (long)vb * blockDim.x + threadIdx.x; i < total; i += (long)vgrid * blockDim.x) {
        const int q = (int)(i % nq);
        const long np = i / nq;
        const int p = (int)(np % P);
        const long n = np / P;
        const int py = p / w, px = p % w;
        FieldAt fa;
        fa.eval(field, n, hf, wf, py, px, h, w, mode);
        Bilin bl;
        bl.setup(fa.x, fa.y, w, h);
        const float* ib = inp + n * P * ld_in + q * 4;
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        if (bl.y0ok && bl.x0ok) {
            const float4 v = *reinterpret_cast<const float4*>(ib + ((long)bl.y0 * w + bl.x0) * ld_in);
            o.x += v.x * bl.wnw; o.y += v.y * bl.wnw; o.z += v.z * bl.wnw; o.w += v.w * bl.wnw;
        }
        if (bl.y0ok && bl.x1ok) {
            const float4 v = *reinterpret_cast<const float4*>(ib + ((long)bl.y0 * w + bl.x0 + 1) * ld_in);
            o.x += v.x * bl.wne; o.y += v.y * bl.wne; o.z += v.z * bl.wne; o.w += v.w * bl.wne;
        }
        if (bl.y1ok && bl.x0ok) {
            const float4 v = *reinterpret_cast<const float4*>(ib + ((long)(bl.y0 + 1) * w + bl.x0) * ld_in);
            o.x += v.x * bl.wsw; o.y += v.y * bl.wsw; o.z += v.z * bl.wsw; o.w += v.w * bl.wsw;
        }
        if (bl.y1ok && bl.x1ok) {
            const float4 v = *reinterpret_cast<const float4*>(ib + ((long)(bl.y0 + 1) * w + bl.x0 + 1) * ld_in);
            o.x += v.x * bl.wse; o.y += v.y * bl.wse; o.z += v.z * bl.wse; o.w += v.w * bl.wse;
        }
        float* op = out + np * ld_out + out_off + q * 4;
        const int rem = C - q * 4;
        if (vec_store && rem >= 4) {
            *reinterpret_cast<float4*>(op) = o;
        } else {
            if (rem > 0) op[0] = o.x;
            if (rem > 1) op[1] = o.y;
            if (rem > 2) op[2] = o.z;
            if (rem > 3) op[3] = o.w;
            if (zero_tail)
                for (int k = rem > 0 ? rem : 0; k < 4; ++k)
                    if (out_off + q * 4 + k < ld_out) op[k] = 0.f;
        }
    }
}

__global__ void __launch_bounds__(256) deform_fwd_kernel(const float* __restrict__ inp, int ld_in, int C, int h, int w,
                                                         const float* __restrict__ field, int hf, int wf, int mode,
                                                         float* __restrict__ out, int ld_out, int out_off, int N) {
    deform_fwd_body(inp, ld_in, C, h, w, field, hf, wf, mode, out, ld_out, out_off, N, blockIdx.x, gridDim.x);
}

// ---- backward of the warps: deterministic, no floating-point atomics ---------------------------------------------------
// grid_sample's adjoint scatters every output pixel's gradient onto the four source texels around its sampling point, and the
// sampling points are data (the predicted field): any number of pixels may name the same texel.  The reference's CPU
// backward is a loop over output pixels, i.e. a fixed-order sum per texel (generator.py:51-58 -> F.grid_sample backward);
// fp32 atomics give a different order -- and different bits -- on every run.  Here the adjoint is two passes:
//   pass A, per OUTPUT PIXEL (warp_bwd_pixel_body): the sampling point (ix, iy) in texel units is stored (samp), and the
//     pixel's share of the field gradient -- sum over channels of dout * d(sample)/d(ix, iy), CL lanes per pixel, channel
//     slices over blockIdx.y, finished with wavefront shuffles -- goes to gpart[slice][pixel][2].  Nothing is added to
//     anything shared.
//   pass B, per SOURCE TEXEL (warp_bwd_gather_body): a block owns a T x T tile of one frame's texels.  It scans the frame's
//     sampling points in pixel order, WG_BATCH at a time, and compacts those whose 2 x 2 footprint touches the tile into an
//     LDS list IN PIXEL ORDER (wavefront ballots + popcounts of the lower lanes: no atomics either); then every
//     (texel, channel quad) thread walks the list and accumulates weight * dout[pixel] of the entries that touch its texel.
//     Per texel that is the reference's own summation order.  d input is WRITTEN (pad channels 0): no zero fill.
//   The field gradient of a field texel is a gather as well (warp_bwd_field_gather): over the levels in order, the pixels
//     of each level that read the texel (nearest pick or the bilinear footprint) in pixel order, the channel slices in order.
constexpr int WG_BATCH = 1024;   // sampling points scanned per round (4 per thread)
constexpr int WG_MAXACC = 4;     // channel quads a thread accumulates (float4 each)
constexpr int WG_PEND = 4;       // hits a thread collects before it fetches their gradients (all loads of a flush are in flight together)

__device__ __forceinline__ int lane_prefix_count(unsigned long long m) {     // set bits of m below this lane
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}

// CL lanes (power of two <= 64) cooperate on one pixel: each walks channels c = cl, cl + CL, ... of the block's channel
// slice (blockIdx.y: few-pixel maps with many channels -- 2 x 2 ... 8 x 8 with 512 ... 1024 -- would otherwise be a few
// dozen blocks whose lanes walk 16 channels one after another); gpart == NULL: only the sampling points are wanted.
__device__ __forceinline__ void warp_bwd_pixel_body(const float* __restrict__ inp, int ld_in, int C, int h, int w,
                                                    const float* __restrict__ field, int hf, int wf, int mode,
                                                    const float* __restrict__ dout, int ld_out, int out_off,
                                                    float* __restrict__ samp, float* __restrict__ gpart, int N, int CL,
                                                    int cslice, int vbx, int vgx, int vby) {
    const long P = (long)h * w;
    const long npix = (long)N * P;
    const int ppb = 256 / CL;   // pixels per block iteration
    const int cl = threadIdx.x % CL, pl = threadIdx.x / CL;
    const long iters = (npix + ppb - 1) / ppb;
    for (long it = vbx; it < iters; it += vgx) {
        const long np = it * ppb + pl;
        const bool live = np < npix;
        float gix = 0.f, giy = 0.f;
        if (live) {
            const int p = (int)(np % P);
            const long n = np / P;
            FieldAt fa;
            fa.eval(field, n, hf, wf, p / w, p % w, h, w, mode);
            Bilin bl;
            bl.setup(fa.x, fa.y, w, h);
            if (vby == 0 && cl == 0) {
                samp[np * 2] = bl.ix;
                samp[np * 2 + 1] = bl.iy;
            }
            if (gpart) {
                const float* ib = inp + n * P * ld_in;
                const long o_nw = ((long)bl.y0 * w + bl.x0) * ld_in, o_ne = o_nw + ld_in;
                const long o_sw = o_nw + (long)w * ld_in, o_se = o_sw + ld_in;
                const bool k_nw = bl.y0ok && bl.x0ok, k_ne = bl.y0ok && bl.x1ok, k_sw = bl.y1ok && bl.x0ok, k_se = bl.y1ok && bl.x1ok;
                const float* gp = dout + np * ld_out + out_off;
                const int c_begin = vby * cslice, c_end = c_begin + cslice < C ? c_begin + cslice : C;
                for (int c = c_begin + cl; c < c_end; c += CL) {
                    const float go = gp[c];
                    const float vnw = k_nw ? ib[o_nw + c] : 0.f, vne = k_ne ? ib[o_ne + c] : 0.f;
                    const float vsw = k_sw ? ib[o_sw + c] : 0.f, vse = k_se ? ib[o_se + c] : 0.f;
                    gix += go * ((vne - vnw) * (1.f - bl.ty) + (vse - vsw) * bl.ty);
                    giy += go * ((vsw - vnw) * (1.f - bl.tx) + (vse - vne) * bl.tx);
                }
            }
        }
        if (gpart) {
            for (int o = CL >> 1; o > 0; o >>= 1) {
                gix += __shfl_xor(gix, o);
                giy += __shfl_xor(giy, o);
            }
            if (live && cl == 0) {
                gpart[((long)vby * npix + np) * 2] = gix * ((float)(w - 1) * 0.5f);
                gpart[((long)vby * npix + np) * 2 + 1] = giy * ((float)(h - 1) * 0.5f);
            }
        }
    }
}

__device__ __forceinline__ float4 load4_channels(const float* __restrict__ row, int c, int C, bool vec) {
    if (vec && c + 4 <= C) return *reinterpret_cast<const float4*>(row + c);
    float4 v;
    v.x = c < C ? row[c] : 0.f;
    v.y = c + 1 < C ? row[c + 1] : 0.f;
    v.z = c + 2 < C ? row[c + 2] : 0.f;
    v.w = c + 3 < C ? row[c + 3] : 0.f;
    return v;
}

// vb = ((frame * tiles + tile) * qslices + quad slice); 256 threads = T*T texels x QL = 256 / (T*T) lanes per texel; a thread
// owns channel quads qbase + a * QL + ql, a < nacc <= WG_MAXACC
__device__ __forceinline__ void warp_bwd_gather_body(const float* __restrict__ dout, int ld_out, int out_off, int C, int h,
                                                     int w, const float* __restrict__ samp, float* __restrict__ dinp,
                                                     int ld_in, int T, int nacc, int qslices, int vb) {
    __shared__ float s_ix[WG_BATCH], s_iy[WG_BATCH];
    __shared__ int s_p[WG_BATCH];
    __shared__ int s_cnt[2 * 16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_x = (w + T - 1) / T, tiles = tiles_x * ((h + T - 1) / T);
    const int slice = vb % qslices;
    const int tile = (vb / qslices) % tiles, n = (vb / qslices) / tiles;
    const int tx0 = (tile % tiles_x) * T, ty0 = (tile / tiles_x) * T;
    const int QL = 256 / (T * T);
    const int tex = tid / QL, ql = tid % QL;
    const int tx = tx0 + tex % T, ty = ty0 + tex / T;
    const bool active = tx < w && ty < h;
    const int nq = ld_in >> 2;
    const int qbase = slice * nacc * QL;
    const int P = h * w;
    const float* sb = samp + (long)n * P * 2;
    const float* db = dout + (long)n * P * ld_out + out_off;
    const bool vec = ((out_off & 3) == 0) && ((ld_out & 3) == 0);
    float4 acc[WG_MAXACC];
#pragma unroll
    for (int a = 0; a < WG_MAXACC; ++a) acc[a] = make_float4(0.f, 0.f, 0.f, 0.f);
    int pend_p[WG_PEND], npend = 0;
    float pend_w[WG_PEND];
#pragma unroll
    for (int k = 0; k < WG_PEND; ++k) pend_p[k] = 0, pend_w[k] = 0.f;
    auto flush = [&]() {
        float4 v[WG_PEND][WG_MAXACC];
#pragma unroll
        for (int k = 0; k < WG_PEND; ++k) {
            const float* gp = db + (long)pend_p[k] * ld_out;
#pragma unroll
            for (int a = 0; a < WG_MAXACC; ++a) {
                const int q = qbase + a * QL + ql;
                v[k][a] = (k < npend && a < nacc && q < nq) ? load4_channels(gp, q * 4, C, vec) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
#pragma unroll
        for (int k = 0; k < WG_PEND; ++k)
            if (k < npend) {
#pragma unroll
                for (int a = 0; a < WG_MAXACC; ++a) {
                    acc[a].x += v[k][a].x * pend_w[k];
                    acc[a].y += v[k][a].y * pend_w[k];
                    acc[a].z += v[k][a].z * pend_w[k];
                    acc[a].w += v[k][a].w * pend_w[k];
                }
            }
        npend = 0;
    };
    // the sampling points of the next round are fetched while this round is worked on; an empty round (no point of the 1024
    // touches the tile -- most rounds of a large map) costs one barrier: the per-wave counts are double-buffered, and the
    // barrier in front of a list write also is the one behind the previous list's readers
    float2 nxt[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int p = j * 256 + tid;
        nxt[j] = p < P ? *reinterpret_cast<const float2*>(sb + 2 * (long)p) : make_float2(0.f, 0.f);
    }
    const float x_lo = (float)(tx0 - 1), x_hi = (float)(tx0 + T - 1), y_lo = (float)(ty0 - 1), y_hi = (float)(ty0 + T - 1);
    int round = 0;
    for (int base = 0; base < P; base += WG_BATCH, ++round) {
        float2 cur[4];
        bool hit[4];
        int pre[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) cur[j] = nxt[j];
        if (base + WG_BATCH < P) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int p = base + WG_BATCH + j * 256 + tid;
                nxt[j] = p < P ? *reinterpret_cast<const float2*>(sb + 2 * (long)p) : make_float2(0.f, 0.f);
            }
        }
        int* cnt = s_cnt + (round & 1) * 16;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int p = base + j * 256 + tid;
            const float fx = floorf(cur[j].x), fy = floorf(cur[j].y);
            // (non-finite and far-away points fail the comparisons: Bilin::setup's clamp)
            hit[j] = p < P && fx >= x_lo && fx <= x_hi && fy >= y_lo && fy <= y_hi;
            const unsigned long long m = __ballot(hit[j]);
            pre[j] = lane_prefix_count(m);
            if (lane == 0) cnt[j * 4 + wave] = __popcll(m);
        }
        __syncthreads();
        int off[4], total = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if ((k & 3) == wave) off[k >> 2] = total;
            total += cnt[k];
        }
        if (total == 0) continue;              // (uniform: every thread read the same sixteen counts)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (hit[j]) {
                const int pos = off[j] + pre[j];
                s_p[pos] = base + j * 256 + tid;
                s_ix[pos] = cur[j].x;
                s_iy[pos] = cur[j].y;
            }
        __syncthreads();
        for (int e = 0; e < total; ++e) {
            const float eix = s_ix[e], eiy = s_iy[e];
            const float fx = floorf(eix), fy = floorf(eiy);
            const int dx = tx - (int)fx, dy = ty - (int)fy;
            if (active && (unsigned)dx < 2u && (unsigned)dy < 2u) {
                // the forward's corner weights: (ex - ix | ix - fx) * (ey - iy | iy - fy), ex = fx + 1
                const float wx = dx ? eix - fx : (fx + 1.f) - eix;
                const float wy = dy ? eiy - fy : (fy + 1.f) - eiy;
                const float wgt = wx * wy;
                const int pe = s_p[e];
#pragma unroll
                for (int k = 0; k < WG_PEND; ++k)
                    if (npend == k) pend_p[k] = pe, pend_w[k] = wgt;
                ++npend;
            }
            // a hit is not fetched at once: one exposed memory latency per hit was the whole kernel (a wavefront meets a hit of
            // SOME lane at almost every entry).  When any lane of the wavefront holds WG_PEND hits, every lane fetches all of
            // its pending ones together and adds them in entry order.
            if (__ballot(npend == WG_PEND)) flush();
        }
    }
    flush();
    if (active) {
        float* op = dinp + (((long)n * h + ty) * w + tx) * ld_in;
#pragma unroll
        for (int a = 0; a < WG_MAXACC; ++a) {
            const int q = qbase + a * QL + ql;
            if (a < nacc && q < nq) *reinterpret_cast<float4*>(op + q * 4) = acc[a];
        }
    }
}

// destination pixels (of a size-`out` axis) that may read field texel s of a size-`in` axis; one pixel of slack either side
__device__ __forceinline__ void field_window(int s, int in_size, int out_size, int mode, int& lo, int& hi) {
    if (mode == 0) {
        lo = (int)((long)s * out_size / in_size) - 1;
        hi = (int)(((long)s + 1) * out_size / in_size) + 1;
    } else {     // Lin1D: i0 in {s-1, s} <=> in/out*(dst+0.5)-0.5 in [s-1, s+1) <=> dst in [((2s-1)out-in)/(2in), ((2s+3)out-in)/(2in))
        const long a = (2l * s - 1) * out_size - in_size, b = (2l * s + 3) * out_size - in_size, d = 2l * in_size;
        lo = (int)(a >= 0 ? a / d : -((-a + d - 1) / d)) - 1;
        hi = (int)(b >= 0 ? (b + d - 1) / d : -((-b) / d)) + 1;
    }
    if (lo < 0) lo = 0;
    if (hi > out_size - 1) hi = out_size - 1;
}

// ---- all warps of a generator pass in ONE launch (generator.py:60-78: the appearance skips of every decoder level are
// warped by the same field, and the key-point embedding is resized into each of them) -----------------------------------
constexpr int MAX_WARP_LEVELS = 12;    // (the vox generator warps nine tensors; the struct travels as kernel arguments: < 4 KB)
struct WarpSeg {          // one level's share of the launch
    const float* inp;
    float* out;           // forward: [N][h][w][ld_out]
    const float* dout;    // backward: its gradient
    float* dinp;          // backward: written by the gather pass (or NULL)
    float* samp;          // backward workspace: sampling point of every output pixel, texel units [N*h*w][2]
    float* gpart;         // backward workspace: field-gradient share of every pixel [slices][N*h*w][2]
    int ld_in, C, h, w, ld_out, out_off, ke, emb_off;
    int warp_begin, warp_blocks;      // forward: blocks of the warp / of the embedding copy; backward pass A: gx * slices
    int emb_begin, emb_blocks;
    int CL, cslice, gx, slices;
    int gat_begin, gat_blocks, T, nacc, qslices;       // backward pass B: the texel-tile gathers of d input
};
struct WarpSegs {
    WarpSeg lv[MAX_WARP_LEVELS];
    int n, N, hf, wf, mode, ld_emb, He, We;
    const float* field;
    const float* emb;
    float* dfield;
    float* demb;
    int dfield_begin, dfield_blocks, dfield_accumulate, dfield_serial;
    int demb_begin, demb_blocks;
};

__global__ void __launch_bounds__(256) warp_levels_fwd_kernel(WarpSegs a) {
    const int b = blockIdx.x;
    for (int l = 0; l < a.n; ++l) {
        const WarpSeg& L = a.lv[l];
        if (b >= L.warp_begin && b < L.warp_begin + L.warp_blocks) {
            deform_fwd_body(L.inp, L.ld_in, L.C, L.h, L.w, a.field, a.hf, a.wf, a.mode, L.out, L.ld_out, 0, a.N, b - L.warp_begin,
                            L.warp_blocks, L.ke == 0);
            return;
        }
        if (b >= L.emb_begin && b < L.emb_begin + L.emb_blocks) {
            // resize of the embedding into channels [emb_off, emb_off + ke) (resize_nearest_kernel / resize_bilinear_kernel of layout.hip)
            const int kz = L.ld_out - L.emb_off;              // the embedding's ke channels and the row's pad channels behind them
            const long total = (long)a.N * L.h * L.w * kz;
            for (long i = (long)(b - L.emb_begin) * 256 + threadIdx.x; i < total; i += (long)L.emb_blocks * 256) {
                const int c = (int)(i % kz);
                const long p = i / kz;
                if (c >= L.ke) {
                    L.out[p * L.ld_out + L.emb_off + c] = 0.f;
                    continue;
                }
                const int x = (int)(p % L.w);
                const long t = p / L.w;
                const int y = (int)(t % L.h);
                const int n = (int)(t / L.h);
                if (a.mode == 0) {
                    const int ys = nearest_src(y, a.He, L.h), xs = nearest_src(x, a.We, L.w);
                    L.out[p * L.ld_out + L.emb_off + c] = a.emb[(((long)n * a.He + ys) * a.We + xs) * a.ld_emb + c];
                } else {          // 'trilinear' with unchanged depth (vox configs): resize_bilinear_kernel of layout.hip
                    Lin1D ly, lx;
                    ly.setup(y, a.He, L.h);
                    lx.setup(x, a.We, L.w);
                    const float* sb = a.emb + (long)n * a.He * a.We * a.ld_emb + c;
                    const float v00 = sb[((long)ly.i0 * a.We + lx.i0) * a.ld_emb], v01 = sb[((long)ly.i0 * a.We + lx.i1) * a.ld_emb];
                    const float v10 = sb[((long)ly.i1 * a.We + lx.i0) * a.ld_emb], v11 = sb[((long)ly.i1 * a.We + lx.i1) * a.ld_emb];
                    L.out[p * L.ld_out + L.emb_off + c] = ly.l0 * (lx.l0 * v00 + lx.l1 * v01) + ly.l1 * (lx.l0 * v10 + lx.l1 * v11);
                }
            }
            return;
        }
    }
}

// pass A of the backward, every level in one launch
__global__ void __launch_bounds__(256) warp_levels_bwd_pixel_kernel(WarpSegs a) {
    const int b = blockIdx.x;
    for (int l = 0; l < a.n; ++l) {
        const WarpSeg& L = a.lv[l];
        if (b >= L.warp_begin && b < L.warp_begin + L.warp_blocks) {
            const int vb = b - L.warp_begin;
            warp_bwd_pixel_body(L.inp, L.ld_in, L.C, L.h, L.w, a.field, a.hf, a.wf, a.mode, L.dout, L.ld_out, L.out_off, L.samp,
                                a.dfield ? L.gpart : nullptr, a.N, L.CL, L.cslice, vb % L.gx, L.gx, vb / L.gx);
            return;
        }
    }
}

// pass B: d input of every level (texel tiles), the field gradient (one thread per field texel) and the gradient of the
// embedding (one thread per element)
// one level's share of the gradient of field texel (n, ys, xs): the pixels of the level that read the texel (nearest pick or
// the bilinear footprint, with the forward's weights in the forward's order) in pixel order, the channel slices in order
__device__ __forceinline__ void field_grad_of_level(const WarpSegs& a, const WarpSeg& L, long n, int ys, int xs, float& sx, float& sy) {
    const long npix = (long)a.N * L.h * L.w;
    int h_lo, h_hi, w_lo, w_hi;
    field_window(ys, a.hf, L.h, a.mode, h_lo, h_hi);
    field_window(xs, a.wf, L.w, a.mode, w_lo, w_hi);
    for (int y = h_lo; y <= h_hi; ++y) {
        Lin1D ly;
        bool y0, y1;
        if (a.mode == 0) {
            y0 = nearest_src(y, a.hf, L.h) == ys, y1 = false;
            ly.l0 = 1.f, ly.l1 = 0.f;
        } else {
            ly.setup(y, a.hf, L.h);
            y0 = ly.i0 == ys, y1 = ly.i1 == ys;
        }
        if (!y0 && !y1) continue;
        for (int x = w_lo; x <= w_hi; ++x) {
            Lin1D lx;
            bool x0, x1;
            if (a.mode == 0) {
                x0 = nearest_src(x, a.wf, L.w) == xs, x1 = false;
                lx.l0 = 1.f, lx.l1 = 0.f;
            } else {
                lx.setup(x, a.wf, L.w);
                x0 = lx.i0 == xs, x1 = lx.i1 == xs;
            }
            if (!x0 && !x1) continue;
            // FieldAt::setup's (index, weight) pairs in its order: (i0y,i0x) (i0y,i1x) (i1y,i0x) (i1y,i1x)
            float wsum = 0.f;
            bool any = false;
            if (y0 && x0) wsum = ly.l0 * lx.l0, any = true;
            if (y0 && x1) wsum = any ? wsum + ly.l0 * lx.l1 : ly.l0 * lx.l1, any = true;
            if (y1 && x0) wsum = any ? wsum + ly.l1 * lx.l0 : ly.l1 * lx.l0, any = true;
            if (y1 && x1) wsum = any ? wsum + ly.l1 * lx.l1 : ly.l1 * lx.l1, any = true;
            const long np = (n * L.h + y) * L.w + x;
            float gx = 0.f, gy = 0.f;
            for (int sl = 0; sl < L.slices; ++sl) {
                gx += L.gpart[((long)sl * npix + np) * 2];
                gy += L.gpart[((long)sl * npix + np) * 2 + 1];
            }
            sx += gx * wsum;
            sy += gy * wsum;
        }
    }
}

// ---- the embedding's gradient under a BILINEAR resize (mode 1: the vox configurations) ----------------------------------------
// A 64 x 64 embedding under a 256 x 256 map: every texel is read by ~64 pixels of that level, by 16 of the next ...
// One lane per level (the nearest form below) then walks 64 scattered pixels alone -- every load instruction of the wavefront
// touches as many cache lines as it has lanes.  Here a WAVEFRONT owns the texel and walks (level, window row) in order; the
// lanes of a row step are (x candidate 0..15) x (4-float column q of the pixel's record): one coalesced read of the row
// segment, weight = the forward's own products for the texel (zero for a pixel that does not read it), a lane accumulates its
// (x, q) column over the rows in order, the sixteen x lanes are summed by a fixed shuffle tree, the levels in order.  All
// control flow is uniform over the wavefront.
__device__ __forceinline__ float xlanes_sum(float v) {       // the sixteen x lanes of a column (lane = x * 4 + q), fixed tree
    v += __shfl_xor(v, 4);
    v += __shfl_xor(v, 8);
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}

__device__ __forceinline__ void emb_grad_bilinear(const WarpSegs& a, long i, int lane) {
    const int xs = (int)(i % a.We);
    const long t = i / a.We;
    const int ys = (int)(t % a.He), n = (int)(t / a.He);
    const int xi = lane >> 2, q = lane & 3;
    for (int c0 = 0; c0 < a.ld_emb; c0 += 16) {
        float4 tot = make_float4(0.f, 0.f, 0.f, 0.f);
        bool first = true;
        for (int l = 0; l < a.n; ++l) {
            const WarpSeg& L = a.lv[l];
            if (L.ke <= c0) continue;
            const int cn = L.ke - c0;                         // channels [c0, c0 + cn) of this level exist
            const bool vec = ((L.emb_off + c0) & 3) == 0 && (L.ld_out & 3) == 0;
            int h_lo, h_hi, w_lo, w_hi;
            field_window(ys, a.He, L.h, 1, h_lo, h_hi);
            field_window(xs, a.We, L.w, 1, w_lo, w_hi);
            float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int xb = w_lo; xb <= w_hi; xb += 16) {
                const int x = xb + xi;
                Lin1D lx;                                   // this lane's column: the same for every row
                lx.setup(x <= w_hi ? x : w_hi, a.We, L.w);
                const bool x0 = lx.i0 == xs, x1 = lx.i1 == xs;
                for (int y = h_lo; y <= h_hi; ++y) {
                    Lin1D ly;
                    ly.setup(y, a.He, L.h);
                    const bool y0 = ly.i0 == ys, y1 = ly.i1 == ys;
                    if (x <= w_hi && 4 * q < cn && (x0 || x1) && (y0 || y1)) {
                        // FieldAt::setup's (index, weight) pairs in its order: (i0y,i0x) (i0y,i1x) (i1y,i0x) (i1y,i1x)
                        float wgt = 0.f;
                        bool any = false;
                        if (y0 && x0) wgt = ly.l0 * lx.l0, any = true;
                        if (y0 && x1) wgt = any ? wgt + ly.l0 * lx.l1 : ly.l0 * lx.l1, any = true;
                        if (y1 && x0) wgt = any ? wgt + ly.l1 * lx.l0 : ly.l1 * lx.l0, any = true;
                        if (y1 && x1) wgt = any ? wgt + ly.l1 * lx.l1 : ly.l1 * lx.l1, any = true;
                        const float* gp = L.dout + (((long)n * L.h + y) * L.w + x) * L.ld_out + L.emb_off + c0 + 4 * q;
                        // (the row's pad channels behind the embedding are zeros and lie inside the row)
                        const float4 g = load4_channels(gp, 0, cn - 4 * q, vec);
                        s.x += g.x * wgt;
                        s.y += g.y * wgt;
                        s.z += g.z * wgt;
                        s.w += g.w * wgt;
                    }
                }
            }
            s.x = xlanes_sum(s.x), s.y = xlanes_sum(s.y), s.z = xlanes_sum(s.z), s.w = xlanes_sum(s.w);
            // (a channel beyond this level's ke got zeros from it: adding them changes nothing)
            tot = first ? s : make_float4(tot.x + s.x, tot.y + s.y, tot.z + s.z, tot.w + s.w);
            first = false;
        }
        if (xi == 0) {
            const float tv[4] = {tot.x, tot.y, tot.z, tot.w};
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (c0 + 4 * q + e < a.ld_emb) a.demb[i * a.ld_emb + c0 + 4 * q + e] = tv[e];
        }
    }
}

// pass B: the field gradient and the gradient of the embedding -- SIXTEEN lanes per texel, lane l works on level l (the levels'
// footprints differ by orders of magnitude), the lanes' sums are added in level order; under a bilinear resize the embedding
// takes the wavefront-per-texel form above -- and d input of every level (texel tiles).  The long-running texel blocks come
// FIRST in the grid so that they run under the tile gathers instead of behind them.
__global__ void __launch_bounds__(256) warp_levels_bwd_gather_kernel(WarpSegs a) {
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63;
    if (a.mode == 1 && b >= a.demb_begin && b < a.demb_begin + a.demb_blocks) {    // bilinear resize of the embedding: a
        const long total = (long)a.N * a.He * a.We;                                    // wavefront per texel (see above)
        for (long i = (long)(b - a.demb_begin) * 4 + (threadIdx.x >> 6); i < total; i += (long)a.demb_blocks * 4)
            emb_grad_bilinear(a, i, lane);
        return;
    }
    // the field gradient (both resize modes) and the embedding under a nearest resize: sixteen lanes per texel, one per level
    const int G = 16, per_block = 256 / G;
    const int lvl = threadIdx.x & (G - 1), group = threadIdx.x / G, gbase = lane & ~(G - 1);
    // a lane picks ITS level: the level table moves from the kernel arguments (uniform access only -- a per-lane index makes
    // the compiler walk the distinct values one after the other) into LDS
    __shared__ WarpSeg s_lv[MAX_WARP_LEVELS];
    const bool texel_block = b < a.demb_begin + a.demb_blocks;      // (the two texel gathers are the first blocks of the grid)
    if (texel_block) {
        for (int l = 0; l < a.n; ++l)
            if (threadIdx.x == 0) s_lv[l] = a.lv[l];
        __syncthreads();
    }
    if (a.dfield_serial && b >= a.dfield_begin && b < a.dfield_begin + a.dfield_blocks) {
        // one thread per field texel, the levels one after the other (tuning value "warp_field_lanes" = 1)
        const long total = (long)a.N * a.hf * a.wf;
        for (long i = (long)(b - a.dfield_begin) * 256 + threadIdx.x; i < total; i += (long)a.dfield_blocks * 256) {
            const int xs = (int)(i % a.wf);
            const long t = i / a.wf;
            float tx = 0.f, ty = 0.f;
            for (int l = 0; l < a.n; ++l) {
                float sx = 0.f, sy = 0.f;
                field_grad_of_level(a, a.lv[l], t / a.hf, (int)(t % a.hf), xs, sx, sy);
                tx = l ? tx + sx : sx;
                ty = l ? ty + sy : sy;
            }
            if (a.dfield_accumulate) {
                a.dfield[i * 2] += tx;
                a.dfield[i * 2 + 1] += ty;
            } else {
                a.dfield[i * 2] = tx;
                a.dfield[i * 2 + 1] = ty;
            }
        }
        return;
    }
    if (b >= a.dfield_begin && b < a.dfield_begin + a.dfield_blocks) {
        const long total = (long)a.N * a.hf * a.wf;
        const long iters = (total + per_block - 1) / per_block;
        for (long it = b - a.dfield_begin; it < iters; it += a.dfield_blocks) {
            const long i = it * per_block + group;
            float sx = 0.f, sy = 0.f;
            if (i < total && lvl < a.n) {
                const int xs = (int)(i % a.wf);
                const long t = i / a.wf;
                field_grad_of_level(a, s_lv[lvl], t / a.hf, (int)(t % a.hf), xs, sx, sy);
            }
            float tx = 0.f, ty = 0.f;
            for (int l = 0; l < a.n; ++l) {                       // the levels' sums in level order
                const float vx = __shfl(sx, gbase + l), vy = __shfl(sy, gbase + l);
                tx = l ? tx + vx : vx;
                ty = l ? ty + vy : vy;
            }
            if (i < total && lvl == 0) {
                if (a.dfield_accumulate) {
                    a.dfield[i * 2] += tx;
                    a.dfield[i * 2 + 1] += ty;
                } else {
                    a.dfield[i * 2] = tx;
                    a.dfield[i * 2 + 1] = ty;
                }
            }
        }
        return;
    }
    if (b >= a.demb_begin && b < a.demb_begin + a.demb_blocks) {
        // nearest resize: every embedding ELEMENT gathers, level after level, the few pixels whose nearest source it is
        // (resize_nearest_bwd_kernel of layout.hip; pad channels are written 0).  One thread per element: the windows are one to
        // four pixels per level -- the lane-per-level form that pays under a bilinear resize cost 3.7x the instructions here
        // (taichi: 0.15 -> 0.27 ms for the pass, profiles/r05_knob_ab_log.txt)
        const long total = (long)a.N * a.He * a.We * a.ld_emb;
        for (long i = (long)(b - a.demb_begin) * 256 + threadIdx.x; i < total; i += (long)a.demb_blocks * 256) {
            const int c = (int)(i % a.ld_emb);
            const long p = i / a.ld_emb;
            const int xs = (int)(p % a.We);
            const long t = p / a.We;
            const int ys = (int)(t % a.He);
            const int n = (int)(t / a.He);
            float tot = 0.f;
            bool first = true;
            for (int l = 0; l < a.n; ++l) {
                const WarpSeg& L = a.lv[l];
                if (L.ke <= 0 || c >= L.ke) continue;
                int h_lo, h_hi, w_lo, w_hi;
                field_window(ys, a.He, L.h, 0, h_lo, h_hi);
                field_window(xs, a.We, L.w, 0, w_lo, w_hi);
                float acc = 0.f;
                for (int y = h_lo; y <= h_hi; ++y) {
                    if (nearest_src(y, a.He, L.h) != ys) continue;
                    for (int x = w_lo; x <= w_hi; ++x) {
                        if (nearest_src(x, a.We, L.w) != xs) continue;
                        acc += L.dout[(((long)n * L.h + y) * L.w + x) * L.ld_out + L.emb_off + c];
                    }
                }
                tot = first ? acc : tot + acc;
                first = false;
            }
            a.demb[i] = tot;
        }
        return;
    }
    for (int l = 0; l < a.n; ++l) {
        const WarpSeg& L = a.lv[l];
        if (b >= L.gat_begin && b < L.gat_begin + L.gat_blocks) {
            warp_bwd_gather_body(L.dout, L.ld_out, L.out_off, L.C, L.h, L.w, L.samp, L.dinp, L.ld_in, L.T, L.nacc, L.qslices,
                                 b - L.gat_begin);
            return;
        }
    }
}

}  // namespace

extern "C" {

int mnk_gconv1x1_fwd(const float* x, int ld_x, const float* w, const float* bias, float* y, int ld_y, long rows,
                     int groups, int gsize, void* stream) {
    MNK_REQUIRE(x && w && y && rows > 0 && groups > 0 && gsize > 0 && gsize <= MAXG);
    MNK_REQUIRE(ld_x >= groups * gsize && ld_y >= groups * gsize);
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(K_CONV1X1, s, (double)rows * groups * gsize * 8);
    hipLaunchKernelGGL(gconv1x1_fwd_kernel, dim3(grid_for(rows * groups)), dim3(256), 0, s, x, ld_x, w, bias, y, ld_y,
                       rows, groups, gsize, 0);
    MNK_LAUNCH_CHECK();
    return MNK_OK;
}

int mnk_gconv1x1_bwd_data(const float* dy, int ld_dy, const float* w, float* dx, int ld_dx, long rows, int groups,
                          int gsize, void* stream) {
    MNK_REQUIRE(dy && w && dx && rows > 0 && groups > 0 && gsize > 0 && gsize <= MAXG);
    MNK_REQUIRE(ld_dy >= groups * gsize && ld_dx >= groups * gsize);
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(K_CONV1X1, s, (double)rows * groups * gsize * 8);
    hipLaunchKernelGGL(gconv1x1_fwd_kernel, dim3(grid_for(rows * groups)), dim3(256), 0, s, dy, ld_dy, w,
                       (const float*)nullptr, dx, ld_dx, rows, groups, gsize, 1);
    MNK_LAUNCH_CHECK();
    return MNK_OK;
}

static long gconv_rows_per_block(long rows) {
    long rb = (rows + 2047) / 2048;
    if (rb > 256) rb = 256;
    if (rb < 1) rb = 1;
    return (rows + rb - 1) / rb;
}

size_t mnk_gconv1x1_workspace_floats(long rows, int groups, int gsize) {
    if (rows <= 0 || groups <= 0) return 0;
    long rpb = gconv_rows_per_block(rows);
    long rb = (rows + rpb - 1) / rpb;
    return (size_t)rb * groups * (MAXG * MAXG + MAXG);
}

int mnk_gconv1x1_bwd_weight(const float* x, int ld_x, const float* dy, int ld_dy, float* dw, float* dbias, long rows,
                            int groups, int gsize, float* ws, size_t ws_floats, void* stream) {
    MNK_REQUIRE(x && dy && dw && ws && rows > 0 && groups > 0 && gsize > 0 && gsize <= MAXG);
    MNK_REQUIRE(ld_x >= groups * gsize && ld_dy >= groups * gsize);
    long rpb = gconv_rows_per_block(rows);
    int rb = (int)((rows + rpb - 1) / rpb);
    if (ws_floats < (size_t)rb * groups * (MAXG * MAXG + MAXG)) {
        set_error("mnk_gconv1x1_bwd_weight: workspace too small");
        return MNK_EWORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(K_CONV1X1, s, (double)rows * groups * gsize * 8);
    hipLaunchKernelGGL(gconv1x1_wgrad_partial_kernel, dim3(rb, groups), dim3(256), 0, s, x, ld_x, dy, ld_dy, rows, groups,
                       gsize, rpb, ws);
    hipLaunchKernelGGL(gconv1x1_wgrad_final_kernel, dim3(ceil_div(groups * gsize * (gsize + 1), 256)), dim3(256), 0, s, ws,
                       rb, groups, gsize, dw, dbias);
    MNK_LAUNCH_CHECK();
    return MNK_OK;
}

// the row-tile forms apply (see conv1x1_rows_fwd_kernel); MNK_CONV1X1_ROWS=0: the thread-per-pixel kernels (A/B runs)
static int g_c11_rows = tuning_knob("conv1x1_rows", &g_c11_rows, 1);
static int g_deform_bwd_blocks = tuning_knob("deform_bwd_blocks", &g_deform_bwd_blocks, 2048);   // 0: no channel slices; 0 / 512 / 2048: 11.03 / 10.99 / 10.95 ms (visit 46)
static bool c11_rows_form(const float* x, int ld_x, int Cin) {
    return g_c11_rows && ld_x % 4 == 0 && ld_x <= C11_MAXLD && Cin + 1 <= 128 && (size_t)x % 16 == 0;
}

int mnk_conv1x1_fwd(const float* x, int ld_x, int Cin, const float* w, const float* bias, float* out, int B, int D, int H,
                    int W, int Cout, int act, void* stream) {
    MNK_REQUIRE(x && w && out && B > 0 && D > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && Cout <= MAXCO && ld_x >= Cin);
    hipStream_t s = (hipStream_t)stream;
    const long rows = (long)B * D * H * W;
    ProfScope prof(K_CONV1X1, s, (double)rows * (Cin + Cout) * 4);
    if (c11_rows_form(x, ld_x, Cin))
        hipLaunchKernelGGL(conv1x1_rows_fwd_kernel, dim3((unsigned)((rows + C11_TR - 1) / C11_TR)), dim3(256), 0, s, x, ld_x,
                           Cin, w, bias, out, B, D, H, W, Cout, act);
    else if (g_c11_rows && Cin >= 64)
        hipLaunchKernelGGL(conv1x1_wave_fwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, x, ld_x, Cin, w, bias,
                           out, B, D, H, W, Cout, act);
    else
        hipLaunchKernelGGL(conv1x1_sigmoid_fwd_kernel, dim3(grid_for(rows)), dim3(256), 0, s, x, ld_x, Cin, w, bias, out, B,
                           D, H, W, Cout, act);
    MNK_LAUNCH_CHECK();
    return MNK_OK;
}

int mnk_conv1x1_sigmoid_fwd(const float* x, int ld_x, int Cin, const float* w, const float* bias, float* out, int B,
                            int D, int H, int W, int Cout, void* stream) {
    return mnk_conv1x1_fwd(x, ld_x, Cin, w, bias, out, B, D, H, W, Cout, 1, stream);
}

static long c11_rows_per_block(long rows) {
    long rb = (rows + 127) / 128;      // ~128 rows per block: 1024 blocks at 64x64 x 32 frames (was 128 blocks: 190 us)
    if (rb > 2048) rb = 2048;
    if (rb < 1) rb = 1;
    return (rows + rb - 1) / rb;
}

size_t mnk_conv1x1_workspace_floats(long rows, int Cin, int Cout) {
    if (rows <= 0 || Cin <= 0 || Cout <= 0) return 0;
    long rpb = c11_rows_per_block(rows);
    long rb = (rows + rpb - 1) / rpb;
    return (size_t)rb * Cout * (Cin + 1);
}

int mnk_conv1x1_bwd(const float* x, int ld_x, int Cin, const float* w, const float* out, const float* dout, float* dx,
                    int ld_dx, float* dw, float* dbias, int B, int D, int H, int W, int Cout, int act, float* ws,
                    size_t ws_floats, void* stream) {
    // dw == NULL: data gradient only (a backward pass that was asked for input gradients); dx == NULL: parameters only
    MNK_REQUIRE(x && w && out && dout && (dx || dw) && B > 0 && D > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0);
    MNK_REQUIRE(Cout <= MAXCO && ld_x >= Cin && ld_dx >= Cin && (!dw || ws));
    const long rows = (long)B * D * H * W;
    long rpb = c11_rows_per_block(rows);
    int rb = (int)((rows + rpb - 1) / rpb);
    if (dw && ws_floats < (size_t)rb * Cout * (Cin + 1)) {
        set_error("mnk_conv1x1_bwd: workspace too small");
        return MNK_EWORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(K_CONV1X1, s, (double)rows * (2 * Cin + 2 * Cout) * 4);
    if (dx && dw && c11_rows_form(x, ld_x, Cin) && ld_dx == ld_x && (size_t)dx % 16 == 0) {
        hipLaunchKernelGGL(conv1x1_rows_bwd_kernel, dim3(rb), dim3(256), 0, s, x, ld_x, Cin, w, out, dout, dx, B, D, H, W,
                           Cout, act, rpb, ws);
    } else {
        if (dx && g_c11_rows && Cin >= 64)
            hipLaunchKernelGGL(conv1x1_elem_bwd_dx_kernel, dim3(grid_for(rows * ld_dx)), dim3(256), 0, s, w, out, dout, dx,
                               ld_dx, Cin, B, D, H, W, Cout, act);
        else if (dx)
            hipLaunchKernelGGL(conv1x1_sigmoid_bwd_dx_kernel, dim3(grid_for(rows)), dim3(256), 0, s, w, out, dout, dx, ld_dx,
                               Cin, B, D, H, W, Cout, act);
        if (dw) {
            const int kchunks = g_c11_rows ? ceil_div(Cin + 1, 64) : 1;
            hipLaunchKernelGGL(conv1x1_sigmoid_wgrad_partial_kernel, dim3(rb, kchunks), dim3(256), 0, s, x, ld_x, Cin, out, dout,
                               B, D, H, W, Cout, rpb, ws, act);
        }
    }
    if (dw)
        hipLaunchKernelGGL(conv1x1_sigmoid_wgrad_final_kernel, dim3(ceil_div(Cout * (Cin + 1), 4)), dim3(256), 0, s, ws, rb,
                           Cin, Cout, dw, dbias);
    MNK_LAUNCH_CHECK();
    return MNK_OK;
}

int mnk_conv1x1_sigmoid_bwd(const float* x, int ld_x, int Cin, const float* w, const float* out, const float* dout,
                            float* dx, int ld_dx, float* dw, float* dbias, int B, int D, int H, int W, int Cout,
                            float* ws, size_t ws_floats, void* stream) {
    return mnk_conv1x1_bwd(x, ld_x, Cin, w, out, dout, dx, ld_dx, dw, dbias, B, D, H, W, Cout, 1, ws, ws_floats, stream);
}

int mnk_motion_field_fwd(const float* pred, int ld, const float* delta, int N, int h, int w, int K, int use_mask,
                         int use_correction, float* field, void* stream) {
    MNK_REQUIRE(pred && field && N > 0 && h > 1 && w > 1 && K >= 0 && K + 1 <= MAXS);
    MNK_REQUIRE(!use_mask || delta);
    MNK_REQUIRE(ld >= (K + 1) * (use_mask ? 1 : 0) + 2 * (use_correction ? 1 : 0));
    hipStream_t s = (hipStream_t)stream;
    const long total = (long)N * h * w;
    ProfScope prof(K_FIELD, s, (double)total * (ld + 2) * 4);
    hipLaunchKernelGGL(motion_field_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, s, pred, ld, DeltaSrc{delta, nullptr, nullptr},
                       N, h, w, K, use_mask, use_correction, field);
    MNK_LAUNCH_CHECK();
    return MNK_OK;
}

int mnk_motion_field_kp_fwd(const float* pred, int ld, const float* mean_s, const float* mean_d, int N, int h, int w, int K,
                            int use_correction, float* field, void* stream) {
    MNK_REQUIRE(pred && field && mean_s && mean_d && N > 0 && h > 1 && w > 1 && K >= 1 && K + 1 <= MAXS);
    MNK_REQUIRE(ld >= K + 1 + 2 * (use_correction ? 1 : 0));
    hipStream_t s = (hipStream_t)stream;
    const long total = (long)N * h * w;
    ProfScope prof(K_FIELD, s, (double)total * (ld + 2) * 4);
    hipLaunchKernelGGL(motion_field_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, s, pred, ld, DeltaSrc{nullptr, mean_s, mean_d},
                       N, h, w, K, 1, use_correction, field);
    MNK_LAUNCH_CHECK();
    return MNK_OK;
}

int mnk_motion_field_bwd(const float* pred, int ld, const float* delta, const float* dfield, int N, int h, int w,
                         int K, int use_mask, int use_correction, float* dpred, int ld_d, float* ddelta,
                         void* stream) {
    MNK_REQUIRE(pred && dfield && dpred && ddelta && N > 0 && h > 1 && w > 1 && K >= 0 && K + 1 <= MAXS);
    MNK_REQUIRE(!use_mask || delta);
    MNK_REQUIRE(ld >= (K + 1) * (use_mask ? 1 : 0) + 2 * (use_correction ? 1 : 0) && ld_d >= ld - 3);
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(K_FIELD, s, (double)N * h * w * (ld + ld_d + 2) * 4);
    hipLaunchKernelGGL(motion_field_bwd_kernel, dim3(N), dim3(MF_THREADS), 0, s, pred, ld, DeltaSrc{delta, nullptr, nullptr}, dfield,
                       h, w, K, use_mask, use_correction, dpred, ld_d, ddelta, (float*)nullptr, (float*)nullptr);
    MNK_LAUNCH_CHECK();
    return MNK_OK;
}

int mnk_motion_field_kp_bwd(const float* pred, int ld, const float* mean_s, const float* mean_d, const float* dfield, int N, int h,
                            int w, int K, int use_correction, float* dpred, int ld_d, float* dmean_s, float* dmean_d,
                            void* stream) {
    MNK_REQUIRE(pred && dfield && dpred && mean_s && mean_d && dmean_s && dmean_d && N > 0 && h > 1 && w > 1 && K >= 1 &&
                K + 1 <= MAXS);
    MNK_REQUIRE(ld >= K + 1 + 2 * (use_correction ? 1 : 0) && ld_d >= ld - 3);
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(K_FIELD, s, (double)N * h * w * (ld + ld_d + 2) * 4);
    hipLaunchKernelGGL(motion_field_bwd_kernel, dim3(N), dim3(MF_THREADS), 0, s, pred, ld, DeltaSrc{nullptr, mean_s, mean_d}, dfield,
                       h, w, K, 1, use_correction, dpred, ld_d, (float*)nullptr, dmean_s, dmean_d);
    MNK_LAUNCH_CHECK();
    return MNK_OK;
}

int mnk_deform_fwd(const float* inp, int ld_in, int C, int h, int w, const float* field, int hf, int wf, int mode,
                   float* out, int ld_out, int out_off, int N, void* stream) {
    MNK_REQUIRE(inp && field && out && N > 0 && C > 0 && h > 0 && w > 0 && hf > 0 && wf > 0 && (mode == 0 || mode == 1));
    MNK_REQUIRE(ld_in % 4 == 0 && ld_in >= round_up(C, 4) && out_off >= 0 && out_off + C <= ld_out);
    hipStream_t s = (hipStream_t)stream;
    const long total = (long)N * h * w * ((C + 3) / 4);
    ProfScope prof(K_DEFORM, s, (double)N * h * w * C * 8);
    hipLaunchKernelGGL(deform_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, s, inp, ld_in, C, h, w, field, hf, wf, mode,
                       out, ld_out, out_off, N);
    MNK_LAUNCH_CHECK();
    return MNK_OK;
}

static void warp_bwd_plan(int C, long npix, int& CL, int& cslice, int& gx, int& slices) {
    CL = 1;
    while (CL < C && CL < 64) CL <<= 1;
    const long iters = (npix + (256 / CL) - 1) / (256 / CL);
    // channel slices (multiples of CL) until the launch has ~2048 blocks; every slice leaves its share of the field gradient
    slices = 1;
    const int max_slices = (C + CL - 1) / CL;
    while (slices < max_slices && iters * slices < g_deform_bwd_blocks) slices <<= 1;
    if (slices > max_slices) slices = max_slices;
    cslice = ((C + slices - 1) / slices + CL - 1) / CL * CL;
    slices = (C + cslice - 1) / cslice;
    gx = (int)(iters < 16384 ? iters : 16384);
}

// tile edge of the gather pass: every block scans its frame's sampling points, so big maps take big tiles (a 256 x 256 map
// in 8 x 8 tiles would read its 512 KB of sampling points 1024 times per frame); small maps take small tiles so that the
// 256 threads split the channels instead of idling
static int g_warp_gather_tile = tuning_knob("warp_gather_tile", &g_warp_gather_tile, 0);    // 0: by map size; 1 / 2 / 4 / 8 / 16 forces the edge (A/B)
static int g_warp_field_lanes = tuning_knob("warp_field_lanes", &g_warp_field_lanes, 1);      // nearest resize: 1 = a thread per field texel (taichi levels 223 -> 182 us per pass, moving-gif 158 -> 146), 16 = a lane per level (the bilinear form)
static int g_warp_gather_rule = tuning_knob("warp_gather_rule", &g_warp_gather_rule, 1);    // 1: lanes per texel by channel count (vox 1243 -> 1040 us per pass, moving-gif 223 -> 214); 0: tile edge by map size
static void warp_gather_plan(int ld_in, int h, int w, int& T, int& nacc, int& qslices, int& tiles) {
    const long P = (long)h * w;
    T = P >= 16384 ? 16 : P >= 1024 ? 8 : P >= 64 ? 4 : P >= 16 ? 2 : 1;
    if (g_warp_gather_rule == 1) {
        // as many lanes per texel as the level has channel quads (every lane busy), the tile as large as that leaves
        int ql = 1;
        while (ql * 2 <= ld_in / 4 && ql < 256) ql <<= 1;
        T = 16;
        while (T > 1 && T * T * ql > 256) T >>= 1;
        while (T > 1 && (long)(T / 2) * (T / 2) >= P) T >>= 1;
    }
    if (g_warp_gather_tile == 1 || g_warp_gather_tile == 2 || g_warp_gather_tile == 4 || g_warp_gather_tile == 8 || g_warp_gather_tile == 16) {
        T = g_warp_gather_tile;
        while (T > 1 && (long)(T / 2) * (T / 2) >= P) T >>= 1;      // (never a tile of mostly idle texel threads)
    }
    const int QL = 256 / (T * T), nq = ld_in / 4;
    nacc = ceil_div(nq, QL);
    if (nacc > WG_MAXACC) nacc = WG_MAXACC;
    qslices = ceil_div(nq, nacc * QL);
    tiles = ceil_div(w, T) * ceil_div(h, T);
}

static size_t warp_level_ws_floats(int C, int h, int w, int N) {
    int CL, cslice, gx, slices;
    const long npix = (long)N * h * w;
    warp_bwd_plan(C, npix, CL, cslice, gx, slices);
    return (size_t)npix * 2 * (1 + slices);
}

// fills the plans and workspace pointers of a.lv[0 .. a.n) (inp, dout, dinp, ld_in, C, h, w, ld_out, out_off, ke, emb_off set by
// the caller) and launches the two passes
static int warp_bwd_launch(WarpSegs& a, float* ws, size_t ws_floats, double bytes, hipStream_t s) {
    size_t need = 0;
    for (int l = 0; l < a.n; ++l) need += warp_level_ws_floats(a.lv[l].C, a.lv[l].h, a.lv[l].w, a.N);
    if (!ws || ws_floats < need) {
        set_error("warp backward: workspace too small");
        return MNK_EWORKSPACE;
    }
    int blocks_a = 0, blocks_b = 0;
    float* wp = ws;
    // pass B's grid: the texel gathers of the field and the embedding first (few, long-running), then the tiles of d input
    a.dfield_begin = blocks_b;
    a.dfield_serial = (a.mode == 0 && g_warp_field_lanes == 1) ? 1 : 0;
    a.dfield_blocks = a.dfield ? (int)std::min<long>(((long)a.N * a.hf * a.wf + (a.dfield_serial ? 255 : 15)) / (a.dfield_serial ? 256 : 16), 32768) : 0;
    blocks_b += a.dfield_blocks;
    a.demb_begin = blocks_b;
    // embedding gather: a thread per element (nearest) / a wavefront per texel (bilinear)
    a.demb_blocks = !a.demb ? 0 : a.mode ? (int)std::min<long>(((long)a.N * a.He * a.We + 3) / 4, 32768)
                                         : grid_for((long)a.N * a.He * a.We * a.ld_emb);
    blocks_b += a.demb_blocks;
    const int small_blocks = blocks_b;
    for (int l = 0; l < a.n; ++l) {
        WarpSeg& L = a.lv[l];
        const long npix = (long)a.N * L.h * L.w;
        warp_bwd_plan(L.C, npix, L.CL, L.cslice, L.gx, L.slices);
        L.samp = wp;
        L.gpart = wp + npix * 2;
        wp += npix * 2 * (1 + L.slices);
        L.warp_begin = blocks_a;
        L.warp_blocks = L.gx * (a.dfield ? L.slices : 1);
        blocks_a += L.warp_blocks;
        L.gat_begin = blocks_b;
        L.gat_blocks = 0;
        if (L.dinp) {
            int tiles;
            warp_gather_plan(L.ld_in, L.h, L.w, L.T, L.nacc, L.qslices, tiles);
            const long nb = (long)a.N * tiles * L.qslices;
            MNK_REQUIRE(nb < (1l << 30));
            L.gat_blocks = (int)nb;
        }
        blocks_b += L.gat_blocks;
    }
    MNK_REQUIRE(blocks_b > 0);
    ProfScope prof(K_DEFORM, s, bytes);
    if (blocks_a > 0 && (a.dfield || blocks_b > small_blocks))
        hipLaunchKernelGGL(warp_levels_bwd_pixel_kernel, dim3(blocks_a), dim3(256), 0, s, a);
    hipLaunchKernelGGL(warp_levels_bwd_gather_kernel, dim3(blocks_b), dim3(256), 0, s, a);
    MNK_LAUNCH_CHECK();
    return MNK_OK;
}

size_t mnk_deform_bwd_workspace_floats(int C, int h, int w, int N) {
    if (C <= 0 || h <= 0 || w <= 0 || N <= 0) return 0;
    return warp_level_ws_floats(C, h, w, N);
}

int mnk_deform_bwd(const float* inp, int ld_in, int C, int h, int w, const float* field, int hf, int wf, int mode,
                   const float* dout, int ld_out, int out_off, float* dinp, float* dfield, int N, float* ws, size_t ws_floats,
                   void* stream) {
    MNK_REQUIRE(inp && field && dout && N > 0 && C > 0 && h > 0 && w > 0 && hf > 0 && wf > 0 && (mode == 0 || mode == 1));
    MNK_REQUIRE(ld_in % 4 == 0 && ld_in >= round_up(C, 4) && out_off >= 0 && out_off + C <= ld_out);
    MNK_REQUIRE(dinp || dfield);
    WarpSegs a = {};
    a.n = 1, a.N = N, a.hf = hf, a.wf = wf, a.mode = mode;
    a.field = field, a.dfield = dfield, a.dfield_accumulate = 1;
    WarpSeg& L = a.lv[0];
    L.inp = inp, L.dout = dout, L.dinp = dinp, L.ld_in = ld_in, L.C = C, L.h = h, L.w = w, L.ld_out = ld_out, L.out_off = out_off;
    return warp_bwd_launch(a, ws, ws_floats, (double)N * h * w * C * 12, (hipStream_t)stream);
}

static int warp_levels_check(const MnkWarpLevel* lv, int n, const float* field, int hf, int wf, int mode, int He, int We,
                             int ld_emb, int N) {
    MNK_REQUIRE(lv && n > 0 && n <= MAX_WARP_LEVELS && field && hf > 0 && wf > 0 && (mode == 0 || mode == 1) && N > 0);
    for (int l = 0; l < n; ++l) {
        MNK_REQUIRE(lv[l].inp && lv[l].C > 0 && lv[l].h > 0 && lv[l].w > 0 && lv[l].ld_in % 4 == 0 &&
                    lv[l].ld_in >= round_up(lv[l].C, 4) && lv[l].C <= lv[l].ld_out && lv[l].ke >= 0);
        MNK_REQUIRE(lv[l].ke == 0 || (He > 0 && We > 0 && lv[l].ke <= ld_emb && lv[l].emb_off >= lv[l].C &&
                                      lv[l].emb_off + lv[l].ke <= lv[l].ld_out));
    }
    return MNK_OK;
}

int mnk_warp_levels_fwd(const MnkWarpLevel* levels, int nlevels, const float* field, int hf, int wf, int mode, const float* emb,
                        int ld_emb, int He, int We, int N, void* stream) {
    if (int rc = warp_levels_check(levels, nlevels, field, hf, wf, mode, He, We, ld_emb, N)) return rc;
    WarpSegs a = {};
    a.n = nlevels, a.N = N, a.hf = hf, a.wf = wf, a.mode = mode, a.ld_emb = ld_emb, a.He = He, a.We = We;
    a.field = field, a.emb = emb;
    int blocks = 0;
    double bytes = 0;
    for (int l = 0; l < nlevels; ++l) {
        const MnkWarpLevel& m = levels[l];
        MNK_REQUIRE(m.out && (m.ke == 0 || emb));
        MNK_REQUIRE(m.ke > 0 || m.ld_out == round_up(m.C, 4));       // (ke = 0: the warp itself writes the row's pad channels)
        WarpSeg& L = a.lv[l];
        L.inp = m.inp, L.out = m.out, L.ld_in = m.ld_in, L.C = m.C, L.h = m.h, L.w = m.w, L.ld_out = m.ld_out, L.ke = m.ke,
        L.emb_off = m.emb_off;
        const long px = (long)N * m.h * m.w;
        L.warp_begin = blocks;
        L.warp_blocks = grid_for(px * ((m.C + 3) / 4));
        blocks += L.warp_blocks;
        L.emb_begin = blocks;
        L.emb_blocks = m.ke > 0 ? grid_for(px * (m.ld_out - m.emb_off)) : 0;
        blocks += L.emb_blocks;
        bytes += (double)px * (m.C + m.ke) * 8;
    }
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(K_DEFORM, s, bytes);
    hipLaunchKernelGGL(warp_levels_fwd_kernel, dim3(blocks), dim3(256), 0, s, a);
    MNK_LAUNCH_CHECK();
    return MNK_OK;
}

size_t mnk_warp_levels_bwd_workspace_floats(const MnkWarpLevel* levels, int nlevels, int N) {
    if (!levels || nlevels <= 0 || nlevels > MAX_WARP_LEVELS || N <= 0) return 0;
    size_t need = 0;
    for (int l = 0; l < nlevels; ++l) {
        if (levels[l].C <= 0 || levels[l].h <= 0 || levels[l].w <= 0) return 0;
        need += warp_level_ws_floats(levels[l].C, levels[l].h, levels[l].w, N);
    }
    return need;
}

int mnk_warp_levels_bwd(const MnkWarpLevel* levels, int nlevels, const float* field, int hf, int wf, int mode, float* dfield,
                        float* demb, int ld_emb, int He, int We, int N, float* ws, size_t ws_floats, void* stream) {
    if (int rc = warp_levels_check(levels, nlevels, field, hf, wf, mode, He, We, ld_emb, N)) return rc;
    WarpSegs a = {};
    a.n = nlevels, a.N = N, a.hf = hf, a.wf = wf, a.mode = mode, a.ld_emb = ld_emb, a.He = He, a.We = We;
    a.field = field, a.dfield = dfield, a.demb = demb, a.dfield_accumulate = 0;
    double bytes = 0;
    for (int l = 0; l < nlevels; ++l) {
        const MnkWarpLevel& m = levels[l];
        MNK_REQUIRE(m.dout);
        WarpSeg& L = a.lv[l];
        L.inp = m.inp, L.dout = m.dout, L.dinp = m.dinp, L.ld_in = m.ld_in, L.C = m.C, L.h = m.h, L.w = m.w,
        L.ld_out = m.ld_out, L.out_off = 0, L.ke = m.ke, L.emb_off = m.emb_off;
        bytes += (double)N * m.h * m.w * m.C * 12;
    }
    return warp_bwd_launch(a, ws, ws_floats, bytes, (hipStream_t)stream);
}
}

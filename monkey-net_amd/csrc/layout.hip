// Layout kernels: (B,C,D,H,W) <-> folded NHWC, channel-slice copies, 2x2 sum-pool, nearest resize.
// All HBM-bound; one thread per pixel (or per pixel x channel-quad), coalesced along the fastest axis.
#include "mnk_common.h"

using namespace mnk;

// PyTorch 'nearest' source index (upsample_nearest: floor(dst * in/out), clamped)
__device__ __forceinline__ int nearest_src(int dst, int in_size, int out_size) {
    float scale = (float)in_size / (float)out_size;
    int s = (int)floorf((float)dst * scale);
    return s < in_size - 1 ? s : in_size - 1;
}

__global__ void __launch_bounds__(256) ncdhw_to_nhwc_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                            int B, int C, int D, int H, int W, int step, int ld) {
    const int Ho = H / step, Wo = W / step;
    const long total = (long)B * D * Ho * Wo;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int wo = (int)(i % Wo);
        long t = i / Wo;
        int ho = (int)(t % Ho);
        t /= Ho;
        int d = (int)(t % D);
        int b = (int)(t / D);
        const float* s = src + (((long)b * C * D + d) * H + (long)ho * step) * W + (long)wo * step;
        float* o = dst + i * ld;
        for (int c = 0; c < C; ++c) o[c] = s[(long)c * D * H * W];
        for (int c = C; c < ld; ++c) o[c] = 0.f;
    }
}

__global__ void __launch_bounds__(256) nhwc_to_ncdhw_kernel(const float* __restrict__ src, int ld,
                                                            float* __restrict__ dst, int B, int C, int D, int H,
                                                            int W) {
    const long total = (long)B * D * H * W;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int w = (int)(i % W);
        long t = i / W;
        int h = (int)(t % H);
        t /= H;
        int d = (int)(t % D);
        int b = (int)(t / D);
        const float* s = src + i * ld;
        float* o = dst + (((long)b * C * D + d) * H + h) * W + w;
        for (int c = 0; c < C; ++c) o[(long)c * D * H * W] = s[c];
    }
}

__global__ void __launch_bounds__(256) copy_channels_kernel(const float* __restrict__ src, int ld_src, int src_off,
                                                            float* __restrict__ dst, int ld_dst, int dst_off, int C,
                                                            long rows, int accumulate) {
    const long total = rows * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long r = i / C;
        int c = (int)(i - r * C);
        float v = src[r * ld_src + src_off + c];
        float* o = dst + r * ld_dst + dst_off + c;
        *o = accumulate ? *o + v : v;
    }
}

// torch.cat([a, b], channel) on acts in ONE launch, pad channels included: out[n][p] = [a[n][p][0..ca) | b[n mod Nb][p][0..cb) | 0...]
// (Nb < N: the batched discriminator pass [generated | real] embeds the same key points for both halves)
__global__ void __launch_bounds__(256) concat2_fwd_kernel(const float* __restrict__ a, int ld_a, int ca,
                                                          const float* __restrict__ b, int ld_b, int cb, int Nb,
                                                          float* __restrict__ out, int ld_out, int N, long rpf) {
    const long total = (long)N * rpf * ld_out;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % ld_out);
        const long r = i / ld_out;
        float v = 0.f;
        if (c < ca) {
            v = a[r * ld_a + c];
        } else if (c < ca + cb) {
            const long n = r / rpf, p = r - n * rpf;
            v = b[((n % Nb) * rpf + p) * ld_b + c - ca];
        }
        out[i] = v;
    }
}

// its adjoint, one launch: ga[n][p] = [g[n][p][0..ca) | 0...], gb[m][p] = [sum over the frames n = m, m + Nb, ... in order of
// g[n][p][ca..ca+cb) | 0...]
__global__ void __launch_bounds__(256) concat2_bwd_kernel(const float* __restrict__ g, int ld_g, int ca, int cb, int Nb,
                                                          float* __restrict__ ga, int ld_a, float* __restrict__ gb, int ld_b,
                                                          int N, long rpf) {
    const long na = (long)N * rpf * ld_a, nb = gb ? (long)Nb * rpf * ld_b : 0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < na + nb; i += (long)gridDim.x * blockDim.x) {
        if (i < na) {
            const int c = (int)(i % ld_a);
            const long r = i / ld_a;
            ga[i] = c < ca ? g[r * ld_g + c] : 0.f;
        } else {
            const long j = i - na;
            const int c = (int)(j % ld_b);
            const long r = j / ld_b;              // (m, p)
            float v = 0.f;
            if (c < cb) {
                v = g[r * ld_g + ca + c];
                for (long n = Nb; n < N; n += Nb) v += g[(n * rpf + r) * ld_g + ca + c];
            }
            gb[j] = v;
        }
    }
}

__global__ void __launch_bounds__(256) sumpool2x2_kernel(const float* __restrict__ src, int ld_src,
                                                         float* __restrict__ dst, int ld_dst, int N, int Hs, int Ws,
                                                         int C) {
    const int Hd = Hs / 2, Wd = Ws / 2;
    const int cq = ld_dst / 4;  // ld_dst is a multiple of 4 and pad channels of src are zero
    const long total = (long)N * Hd * Wd * cq;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int q = (int)(i % cq);
        long p = i / cq;
        int w = (int)(p % Wd);
        long t = p / Wd;
        int h = (int)(t % Hd);
        int n = (int)(t / Hd);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (q * 4 < C) {
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    const float4 v = *reinterpret_cast<const float4*>(
                        src + (((long)n * Hs + 2 * h + dy) * Ws + 2 * w + dx) * ld_src + q * 4);
                    acc.x += v.x;
                    acc.y += v.y;
                    acc.z += v.z;
                    acc.w += v.w;
                }
        }
        *reinterpret_cast<float4*>(dst + p * ld_dst + q * 4) = acc;
    }
}

__global__ void __launch_bounds__(256) resize_nearest_kernel(const float* __restrict__ src, int ld_src, int Hs, int Ws,
                                                             float* __restrict__ dst, int ld_dst, int dst_off, int Hd,
                                                             int Wd, int N, int C) {
    const long total = (long)N * Hd * Wd * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int c = (int)(i % C);
        long p = i / C;
        int w = (int)(p % Wd);
        long t = p / Wd;
        int h = (int)(t % Hd);
        int n = (int)(t / Hd);
        int hs = nearest_src(h, Hs, Hd), ws = nearest_src(w, Ws, Wd);
        dst[p * ld_dst + dst_off + c] = src[(((long)n * Hs + hs) * Ws + ws) * ld_src + c];
    }
}

// adjoint of the gather above, written as a gather over source pixels (deterministic, no atomics):
// each source pixel sums the destination pixels whose nearest source it is.
__global__ void __launch_bounds__(256) resize_nearest_bwd_kernel(const float* __restrict__ ddst, int ld_dst,
                                                                 int dst_off, int Hd, int Wd,
                                                                 float* __restrict__ dsrc, int ld_src, int Hs, int Ws,
                                                                 int N, int C, int accumulate) {
    const long total = (long)N * Hs * Ws * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int c = (int)(i % C);
        long p = i / C;
        int ws = (int)(p % Ws);
        long t = p / Ws;
        int hs = (int)(t % Hs);
        int n = (int)(t / Hs);
        // candidate destination range: dst with floor(dst*Hs/Hd) == hs  (searched in a small window)
        int h_lo = (int)((long)hs * Hd / Hs) - 1, h_hi = (int)(((long)hs + 1) * Hd / Hs) + 1;
        int w_lo = (int)((long)ws * Wd / Ws) - 1, w_hi = (int)(((long)ws + 1) * Wd / Ws) + 1;
        if (h_lo < 0) h_lo = 0;
        if (w_lo < 0) w_lo = 0;
        if (h_hi > Hd - 1) h_hi = Hd - 1;
        if (w_hi > Wd - 1) w_hi = Wd - 1;
        float acc = 0.f;
        for (int h = h_lo; h <= h_hi; ++h) {
            if (nearest_src(h, Hs, Hd) != hs) continue;
            for (int w = w_lo; w <= w_hi; ++w) {
                if (nearest_src(w, Ws, Wd) != ws) continue;
                acc += ddst[(((long)n * Hd + h) * Wd + w) * ld_dst + dst_off + c];
            }
        }
        dsrc[p * ld_src + c] = accumulate ? dsrc[p * ld_src + c] + acc : acc;
    }
}

// bilinear resize with align_corners=False (ATen upsample_bilinear2d / 'trilinear' with unchanged depth):
// source index = max(0, scale*(dst+0.5)-0.5), used for the key-point embedding when interpolation_mode='trilinear'
// (generator.py:72, vox configs)
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

__global__ void __launch_bounds__(256) resize_bilinear_kernel(const float* __restrict__ src, int ld_src, int Hs, int Ws,
                                                              float* __restrict__ dst, int ld_dst, int dst_off, int Hd,
                                                              int Wd, int N, int C) {
    const long total = (long)N * Hd * Wd * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int c = (int)(i % C);
        long p = i / C;
        int w = (int)(p % Wd);
        long t = p / Wd;
        int h = (int)(t % Hd);
        int n = (int)(t / Hd);
        Lin1D ly, lx;
        ly.setup(h, Hs, Hd);
        lx.setup(w, Ws, Wd);
        const float* sb = src + (long)n * Hs * Ws * ld_src + c;
        const float v00 = sb[((long)ly.i0 * Ws + lx.i0) * ld_src], v01 = sb[((long)ly.i0 * Ws + lx.i1) * ld_src];
        const float v10 = sb[((long)ly.i1 * Ws + lx.i0) * ld_src], v11 = sb[((long)ly.i1 * Ws + lx.i1) * ld_src];
        dst[p * ld_dst + dst_off + c] = ly.l0 * (lx.l0 * v00 + lx.l1 * v01) + ly.l1 * (lx.l0 * v10 + lx.l1 * v11);
    }
}

// adjoint, written as a gather over source texels (deterministic, no atomics): a source texel walks the destination
// pixels of a small window in order, forms each pixel's four (index, weight) pairs exactly as the forward does and adds
// the pairs that name it, in the forward's order; the result is ADDED to dsrc by the one thread that owns the texel.
__device__ __forceinline__ void lin1d_window(int s, int in_size, int out_size, int& lo, int& hi) {
    // dst with i0 in {s-1, s} (or clamped onto s): scale*(dst+0.5)-0.5 in [s-1, s+1); one pixel of slack either side
    lo = (int)(((long)s - 1) * out_size / in_size) - 2;
    hi = (int)((((long)s + 2) * out_size + in_size - 1) / in_size) + 1;
    if (lo < 0) lo = 0;
    if (hi > out_size - 1) hi = out_size - 1;
}

__global__ void __launch_bounds__(256) resize_bilinear_bwd_kernel(const float* __restrict__ ddst, int ld_dst,
                                                                  int dst_off, int Hd, int Wd,
                                                                  float* __restrict__ dsrc, int ld_src, int Hs, int Ws,
                                                                  int N, int C) {
    const long total = (long)N * Hs * Ws * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const long p = i / C;
        const int ws = (int)(p % Ws);
        const long t = p / Ws;
        const int hs = (int)(t % Hs);
        const int n = (int)(t / Hs);
        int h_lo, h_hi, w_lo, w_hi;
        lin1d_window(hs, Hs, Hd, h_lo, h_hi);
        lin1d_window(ws, Ws, Wd, w_lo, w_hi);
        float acc = 0.f;
        for (int h = h_lo; h <= h_hi; ++h) {
            Lin1D ly;
            ly.setup(h, Hs, Hd);
            if (ly.i0 != hs && ly.i1 != hs) continue;
            for (int w = w_lo; w <= w_hi; ++w) {
                Lin1D lx;
                lx.setup(w, Ws, Wd);
                if (lx.i0 != ws && lx.i1 != ws) continue;
                const float g = ddst[(((long)n * Hd + h) * Wd + w) * ld_dst + dst_off + c];
                if (ly.i0 == hs && lx.i0 == ws) acc += g * ly.l0 * lx.l0;
                if (ly.i0 == hs && lx.i1 == ws) acc += g * ly.l0 * lx.l1;
                if (ly.i1 == hs && lx.i0 == ws) acc += g * ly.l1 * lx.l0;
                if (ly.i1 == hs && lx.i1 == ws) acc += g * ly.l1 * lx.l1;
            }
        }
        dsrc[p * ld_src + c] += acc;
    }
}

static inline int grid_for(long total, int cap = 2048) {
    long b = (total + 255) / 256;
    if (b < 1) b = 1;
    return (int)(b < cap ? b : cap);
}

extern "C" {

int mnk_ncdhw_to_nhwc(const float* src, float* dst, int B, int C, int D, int H, int W, int step, int ld_dst,
                      void* stream) {
    MNK_REQUIRE(src && dst && B > 0 && C > 0 && D > 0 && H > 0 && W > 0 && step >= 1 && ld_dst >= C);
    hipStream_t s = (hipStream_t)stream;
    long total = (long)B * D * (H / step) * (W / step);
    ProfScope prof(K_LAYOUT, s, (double)total * (C + ld_dst) * 4);
    hipLaunchKernelGGL(ncdhw_to_nhwc_kernel, dim3(grid_for(total)), dim3(256), 0, s, src, dst, B, C, D, H, W, step,
                       ld_dst);
    MNK_LAUNCH_CHECK();
    return MNK_OK;
}

int mnk_nhwc_to_ncdhw(const float* src, int ld_src, float* dst, int B, int C, int D, int H, int W, void* stream) {
    MNK_REQUIRE(src && dst && B > 0 && C > 0 && D > 0 && H > 0 && W > 0 && ld_src >= C);
    hipStream_t s = (hipStream_t)stream;
    long total = (long)B * D * H * W;
    ProfScope prof(K_LAYOUT, s, (double)total * (C + ld_src) * 4);
    hipLaunchKernelGGL(nhwc_to_ncdhw_kernel, dim3(grid_for(total)), dim3(256), 0, s, src, ld_src, dst, B, C, D, H, W);
    MNK_LAUNCH_CHECK();
    return MNK_OK;
}

int mnk_copy_channels(const float* src, int ld_src, int src_off, float* dst, int ld_dst, int dst_off, int C,
                      long rows, int accumulate, void* stream) {
    MNK_REQUIRE(src && dst && C > 0 && rows > 0 && src_off >= 0 && dst_off >= 0 && src_off + C <= ld_src &&
                dst_off + C <= ld_dst);
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(K_LAYOUT, s, (double)rows * C * 8);
    hipLaunchKernelGGL(copy_channels_kernel, dim3(grid_for(rows * C)), dim3(256), 0, s, src, ld_src, src_off, dst,
                       ld_dst, dst_off, C, rows, accumulate);
    MNK_LAUNCH_CHECK();
    return MNK_OK;
}

int mnk_concat2_fwd(const float* a, int ld_a, int ca, const float* b, int ld_b, int cb, int Nb, float* out, int ld_out, int N,
                    long rows_per_frame, void* stream) {
    MNK_REQUIRE(a && b && out && ca > 0 && cb > 0 && ca <= ld_a && cb <= ld_b && ca + cb <= ld_out && N > 0 && Nb > 0 &&
                N % Nb == 0 && rows_per_frame > 0);
    hipStream_t s = (hipStream_t)stream;
    const long total = (long)N * rows_per_frame * ld_out;
    ProfScope prof(K_LAYOUT, s, (double)total * 8);
    hipLaunchKernelGGL(concat2_fwd_kernel, dim3(grid_for(total, 8192)), dim3(256), 0, s, a, ld_a, ca, b, ld_b, cb, Nb, out, ld_out, N,
                       rows_per_frame);
    MNK_LAUNCH_CHECK();
    return MNK_OK;
}

int mnk_concat2_bwd(const float* g, int ld_g, int ca, int cb, int Nb, float* ga, int ld_a, float* gb, int ld_b, int N,
                    long rows_per_frame, void* stream) {
    MNK_REQUIRE(g && ga && ca > 0 && cb > 0 && ca <= ld_a && ca + cb <= ld_g && N > 0 && Nb > 0 && N % Nb == 0 &&
                rows_per_frame > 0 && (!gb || cb <= ld_b));
    hipStream_t s = (hipStream_t)stream;
    const long total = (long)N * rows_per_frame * ld_a + (gb ? (long)Nb * rows_per_frame * ld_b : 0);
    ProfScope prof(K_LAYOUT, s, (double)total * 8);
    hipLaunchKernelGGL(concat2_bwd_kernel, dim3(grid_for(total, 8192)), dim3(256), 0, s, g, ld_g, ca, cb, Nb, ga, ld_a, gb, ld_b, N,
                       rows_per_frame);
    MNK_LAUNCH_CHECK();
    return MNK_OK;
}

int mnk_sumpool2x2(const float* src, int ld_src, float* dst, int ld_dst, int N, int Hs, int Ws, int C, void* stream) {
    MNK_REQUIRE(src && dst && N > 0 && Hs > 0 && Ws > 0 && (Hs % 2) == 0 && (Ws % 2) == 0 && C > 0);
    MNK_REQUIRE(ld_src % 4 == 0 && ld_dst % 4 == 0 && ld_src >= C && ld_dst >= C && ld_src >= mnk::round_up(C, 4));
    hipStream_t s = (hipStream_t)stream;
    long total = (long)N * (Hs / 2) * (Ws / 2) * (ld_dst / 4);
    ProfScope prof(K_LAYOUT, s, (double)N * Hs * Ws * C * 5);
    hipLaunchKernelGGL(sumpool2x2_kernel, dim3(grid_for(total)), dim3(256), 0, s, src, ld_src, dst, ld_dst, N, Hs, Ws, C);
    MNK_LAUNCH_CHECK();
    return MNK_OK;
}

int mnk_resize_nearest(const float* src, int ld_src, int Hs, int Ws, float* dst, int ld_dst, int dst_off, int Hd,
                       int Wd, int N, int C, void* stream) {
    MNK_REQUIRE(src && dst && N > 0 && C > 0 && Hs > 0 && Ws > 0 && Hd > 0 && Wd > 0 && ld_src >= C &&
                dst_off >= 0 && dst_off + C <= ld_dst);
    hipStream_t s = (hipStream_t)stream;
    long total = (long)N * Hd * Wd * C;
    ProfScope prof(K_LAYOUT, s, (double)total * 8);
    hipLaunchKernelGGL(resize_nearest_kernel, dim3(grid_for(total)), dim3(256), 0, s, src, ld_src, Hs, Ws, dst, ld_dst,
                       dst_off, Hd, Wd, N, C);
    MNK_LAUNCH_CHECK();
    return MNK_OK;
}

static int resize_nearest_bwd_impl(const float* ddst, int ld_dst, int dst_off, int Hd, int Wd, float* dsrc, int ld_src,
                                   int Hs, int Ws, int N, int C, int accumulate, void* stream) {
    MNK_REQUIRE(ddst && dsrc && N > 0 && C > 0 && Hs > 0 && Ws > 0 && Hd > 0 && Wd > 0 && ld_src >= C &&
                dst_off >= 0 && dst_off + C <= ld_dst);
    hipStream_t s = (hipStream_t)stream;
    long total = (long)N * Hs * Ws * C;
    ProfScope prof(K_LAYOUT, s, (double)total * 8);
    hipLaunchKernelGGL(resize_nearest_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, s, ddst, ld_dst, dst_off, Hd, Wd,
                       dsrc, ld_src, Hs, Ws, N, C, accumulate);
    MNK_LAUNCH_CHECK();
    return MNK_OK;
}

int mnk_resize_nearest_bwd(const float* ddst, int ld_dst, int dst_off, int Hd, int Wd, float* dsrc, int ld_src,
                           int Hs, int Ws, int N, int C, void* stream) {
    return resize_nearest_bwd_impl(ddst, ld_dst, dst_off, Hd, Wd, dsrc, ld_src, Hs, Ws, N, C, 0, stream);
}

int mnk_resize_nearest_bwd_accumulate(const float* ddst, int ld_dst, int dst_off, int Hd, int Wd, float* dsrc, int ld_src,
                                      int Hs, int Ws, int N, int C, void* stream) {
    return resize_nearest_bwd_impl(ddst, ld_dst, dst_off, Hd, Wd, dsrc, ld_src, Hs, Ws, N, C, 1, stream);
}

int mnk_resize_bilinear(const float* src, int ld_src, int Hs, int Ws, float* dst, int ld_dst, int dst_off, int Hd,
                        int Wd, int N, int C, void* stream) {
    MNK_REQUIRE(src && dst && N > 0 && C > 0 && Hs > 0 && Ws > 0 && Hd > 0 && Wd > 0 && ld_src >= C &&
                dst_off >= 0 && dst_off + C <= ld_dst);
    hipStream_t s = (hipStream_t)stream;
    long total = (long)N * Hd * Wd * C;
    ProfScope prof(K_LAYOUT, s, (double)total * 8);
    hipLaunchKernelGGL(resize_bilinear_kernel, dim3(grid_for(total)), dim3(256), 0, s, src, ld_src, Hs, Ws, dst, ld_dst,
                       dst_off, Hd, Wd, N, C);
    MNK_LAUNCH_CHECK();
    return MNK_OK;
}

int mnk_resize_bilinear_bwd(const float* ddst, int ld_dst, int dst_off, int Hd, int Wd, float* dsrc, int ld_src,
                            int Hs, int Ws, int N, int C, void* stream) {
    MNK_REQUIRE(ddst && dsrc && N > 0 && C > 0 && Hs > 0 && Ws > 0 && Hd > 0 && Wd > 0 && ld_src >= C &&
                dst_off >= 0 && dst_off + C <= ld_dst);
    hipStream_t s = (hipStream_t)stream;
    long total = (long)N * Hs * Ws * C;
    ProfScope prof(K_LAYOUT, s, (double)N * Hd * Wd * C * 8);
    hipLaunchKernelGGL(resize_bilinear_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, s, ddst, ld_dst, dst_off, Hd,
                       Wd, dsrc, ld_src, Hs, Ws, N, C);
    MNK_LAUNCH_CHECK();
    return MNK_OK;
}
}

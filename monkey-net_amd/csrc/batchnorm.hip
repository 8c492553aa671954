// Normalisation statistics / apply / backward for folded NHWC activations: BatchNorm (sync_batchnorm/batchnorm.py:
// 48-78,113-125; statistics over all frames) and, with per-frame statistics, the discriminator's InstanceNorm3d
// (modules/discriminator.py:19-22,29-30); fused with ReLU / LeakyReLU(slope) and the (1,2,2) average pool.
// HBM-bound: float4 along channels, a 2-D thread map (channel-quad x row) so that every thread keeps its
// channel quad in registers while it walks rows; column sums are finished in LDS and by a tiny second pass.
#include "mnk_common.h"

using namespace mnk;

namespace {

struct Map2D {
    int tx, ty, col_tiles, row_blocks;
    long rows_per_block;
};

static Map2D make_map(long rows, int ld) {
    Map2D m;
    int nv = ld / 4;
    int tx = 1;
    while (tx < nv && tx < 64) tx <<= 1;
    m.tx = tx;
    m.ty = 256 / tx;
    m.col_tiles = (nv + tx - 1) / tx;
    long want = 1024 / m.col_tiles;   // ~4 blocks per CU in total
    if (want < 1) want = 1;
    long min_rows = (long)m.ty * 8;  // at least 8 rows per thread before splitting further
    long rb = (rows + min_rows - 1) / min_rows;
    if (rb > want) rb = want;
    if (rb < 1) rb = 1;
    m.row_blocks = (int)rb;
    m.rows_per_block = (rows + rb - 1) / rb;
    return m;
}

__device__ __forceinline__ float4 f4_add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 f4_fma(float4 a, float4 b, float4 c) {
    return make_float4(fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y), fmaf(a.z, b.z, c.z), fmaf(a.w, b.w, c.w));
}

// per-channel parameter quad with a bounds guard (parameter vectors are exactly C long)
__device__ __forceinline__ float4 ld4_guard(const float* p, int q, int C) {
    const int c = q * 4;
    float4 v;
    v.x = c < C ? p[c] : 0.f;
    v.y = c + 1 < C ? p[c + 1] : 0.f;
    v.z = c + 2 < C ? p[c + 2] : 0.f;
    v.w = c + 3 < C ? p[c + 3] : 0.f;
    return v;
}

// partial[rb][which][ld]
template <class F>
__global__ void __launch_bounds__(256) colsum2_partial_kernel(F f, long rows, int nv, int ld, int tx_n, int ty_n,
                                                              long rows_per_block, float* __restrict__ partial) {
    __shared__ float4 red[2][256];
    const int tx = threadIdx.x % tx_n, ty = threadIdx.x / tx_n;
    const int q = blockIdx.x * tx_n + tx;
    // blockIdx.z = frame when the statistics are per frame (`rows` = rows of one frame); 0 otherwise
    const long fbase = (long)blockIdx.z * rows;
    const long r0 = fbase + (long)blockIdx.y * rows_per_block;
    long r1 = r0 + rows_per_block;
    if (r1 > fbase + rows) r1 = fbase + rows;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
    if (q < nv) {
        for (long r = r0 + ty; r < r1; r += ty_n) {
            float4 va, vb;
            f(r, q, va, vb);
            a = f4_add(a, va);
            b = f4_add(b, vb);
        }
    }
    red[0][threadIdx.x] = a;
    red[1][threadIdx.x] = b;
    __syncthreads();
    for (int s = ty_n >> 1; s > 0; s >>= 1) {
        if (ty < s) {
            red[0][threadIdx.x] = f4_add(red[0][threadIdx.x], red[0][threadIdx.x + s * tx_n]);
            red[1][threadIdx.x] = f4_add(red[1][threadIdx.x], red[1][threadIdx.x + s * tx_n]);
        }
        __syncthreads();
    }
    if (ty == 0 && q < nv) {
        float* o = partial + ((long)blockIdx.z * gridDim.y + blockIdx.y) * 2 * ld;
        *reinterpret_cast<float4*>(o + q * 4) = red[0][tx];
        *reinterpret_cast<float4*>(o + ld + q * 4) = red[1][tx];
    }
}

// one wavefront per output column: lanes stride over the row-block partials (fp64 accumulation), the 64 lane sums are
// combined through LDS.  (A serial loop per column was 44 % of the step time in the first MI355X profile.)
__global__ void __launch_bounds__(256) colsum2_final_kernel(const float* __restrict__ partial, int row_blocks, int ld,
                                                            int C, int frames, float* __restrict__ sums) {
    // sums[which][frame][c] = sum_rb partial[frame][rb][which][c]
    __shared__ double sm[256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = blockIdx.x * 4 + wave;
    const int FC = frames * C;
    double acc = 0.0;
    if (i < 2 * FC) {
        const int which = i / FC, rem = i - which * FC;
        const int f = rem / C, c = rem - f * C;
        const float* pb = partial + (long)f * row_blocks * 2 * ld;
        for (int rb = lane; rb < row_blocks; rb += 64) acc += (double)pb[((long)rb * 2 + which) * ld + c];
    }
    sm[threadIdx.x] = acc;
    __syncthreads();
    if (lane < 8) {
        double t = 0.0;
        for (int j = 0; j < 8; ++j) t += sm[wave * 64 + lane * 8 + j];
        sm[wave * 64 + lane * 8] = t;
    }
    __syncthreads();
    if (lane == 0 && i < 2 * FC) {
        double t = 0.0;
        for (int j = 0; j < 8; ++j) t += sm[wave * 64 + j * 8];
        sums[i] = (float)t;
    }
}

struct StatsLoader {
    const float* x;
    int ld;
    __device__ __forceinline__ void operator()(long r, int q, float4& a, float4& b) const {
        a = *reinterpret_cast<const float4*>(x + r * ld + q * 4);
        b = make_float4(a.x * a.x, a.y * a.y, a.z * a.z, a.w * a.w);
    }
};

// g = act'((y-mean)*scale+beta) * dz (dz at half resolution, /4, when pooled); xhat = (y-mean)*invstd
// act: slope < 0 none, 0 ReLU, > 0 LeakyReLU(slope).  pstride = C selects per-frame mean / invstd / scale (InstanceNorm).
struct BwdLoader {
    const float *y, *dz, *mean, *invstd, *scale, *beta;
    int ld_y, ld_dz, dz_off, H, W, C, pool, pstride;
    float slope;
    __device__ __forceinline__ long poff(long r) const { return pstride ? (r / ((long)H * W)) * pstride : 0; }
    __device__ __forceinline__ void load(long r, int q, float4& g, float4& xhat) const {
        const float4 v = *reinterpret_cast<const float4*>(y + r * ld_y + q * 4);
        const long po = poff(r);
        const float4 m = ld4_guard(mean + po, q, C);
        const float4 is = ld4_guard(invstd + po, q, C);
        long rz = r;
        float k = 1.f;
        if (pool) {
            int w = (int)(r % W);
            long t = r / W;
            int h = (int)(t % H);
            long n = t / H;
            rz = (n * (H / 2) + (h >> 1)) * (W / 2) + (w >> 1);
            k = ((h >> 1) < H / 2 && (w >> 1) < W / 2) ? 0.25f : 0.f;   // odd H / W: the last row / column is not pooled
            if (k == 0.f) rz = 0;
        }
        const float* dp = dz + rz * ld_dz + dz_off + q * 4;
        const int rem = C - q * 4;
        g = make_float4(rem > 0 ? dp[0] * k : 0.f, rem > 1 ? dp[1] * k : 0.f, rem > 2 ? dp[2] * k : 0.f,
                        rem > 3 ? dp[3] * k : 0.f);
        const float4 d = make_float4(v.x - m.x, v.y - m.y, v.z - m.z, v.w - m.w);
        if (slope >= 0.f) {
            const float4 sc = ld4_guard(scale + po, q, C);
            const float4 be = ld4_guard(beta, q, C);
            if (!(fmaf(d.x, sc.x, be.x) > 0.f)) g.x *= slope;
            if (!(fmaf(d.y, sc.y, be.y) > 0.f)) g.y *= slope;
            if (!(fmaf(d.z, sc.z, be.z) > 0.f)) g.z *= slope;
            if (!(fmaf(d.w, sc.w, be.w) > 0.f)) g.w *= slope;
        }
        xhat = make_float4(d.x * is.x, d.y * is.y, d.z * is.z, d.w * is.w);
    }
    __device__ __forceinline__ void operator()(long r, int q, float4& a, float4& b) const {
        float4 g, xh;
        load(r, q, g, xh);
        a = g;
        b = make_float4(g.x * xh.x, g.y * xh.y, g.z * xh.z, g.w * xh.w);
    }
};

// C = number of statistics entries (channels, or frames*channels for per-frame statistics); gamma has gamma_mod entries
__global__ void __launch_bounds__(256) bn_finalize_kernel(const float* __restrict__ sums, double count,
                                                          const float* __restrict__ gamma, int gamma_mod,
                                                          float* running_mean, float* running_var, float momentum,
                                                          float eps, int C, int update_running, float* mean,
                                                          float* invstd, float* scale) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double m = (double)sums[c] / count;
    double v = (double)sums[C + c] / count - m * m;
    if (v < 0.0) v = 0.0;
    float mf = (float)m, vf = (float)v;
    float is = 1.0f / sqrtf(vf + eps);
    mean[c] = mf;
    invstd[c] = is;
    scale[c] = gamma[c % gamma_mod] * is;
    if (update_running) {
        float unbiased = (float)(v * count / (count - 1.0));
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mf;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
    }
}

__global__ void __launch_bounds__(256) bn_eval_coeffs_kernel(const float* __restrict__ gamma,
                                                             const float* __restrict__ rm,
                                                             const float* __restrict__ rv, float eps, int C,
                                                             float* mean, float* invstd, float* scale) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float is = 1.0f / sqrtf(rv[c] + eps);
    mean[c] = rm[c];
    invstd[c] = is;
    scale[c] = gamma[c] * is;
}

__device__ __forceinline__ float act_apply(float v, float slope) {   // slope < 0: identity; 0: ReLU; > 0: LeakyReLU
    return (slope >= 0.f && !(v > 0.f)) ? v * slope : v;
}

template <int POOL>
__global__ void __launch_bounds__(256) bn_act_fwd_kernel(const float* __restrict__ y, int ld_y,
                                                         const float* __restrict__ mean,
                                                         const float* __restrict__ scale,
                                                         const float* __restrict__ beta, int pstride,
                                                         float* __restrict__ z, int ld_z, int z_off, int N, int H,
                                                         int W, int C, float slope) {
    const int nv = (C + 3) / 4;
    const int Ho = POOL ? H / 2 : H, Wo = POOL ? W / 2 : W;
    const long total = (long)N * Ho * Wo * nv;
    const long HWo = (long)Ho * Wo;
    const bool vec_store = ((z_off & 3) == 0) && ((ld_z & 3) == 0);
    const bool owns_pads = z_off == 0 && ld_z == nv * 4;   // z is a plain act: its pad channels are written (as zero) here
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int q = (int)(i % nv);
        long p = i / nv;
        const long po = pstride ? (p / HWo) * pstride : 0;     // per-frame statistics (InstanceNorm)
        const float4 m = ld4_guard(mean + po, q, C);
        const float4 sc = ld4_guard(scale + po, q, C);
        const float4 be = ld4_guard(beta, q, C);
        float4 o;
        if (POOL) {
            int wo = (int)(p % Wo);
            long t = p / Wo;
            int ho = (int)(t % Ho);
            long n = t / Ho;
            o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    const float4 v = *reinterpret_cast<const float4*>(
                        y + ((n * H + 2 * ho + dy) * W + 2 * wo + dx) * ld_y + q * 4);
                    o.x += act_apply(fmaf(v.x - m.x, sc.x, be.x), slope);
                    o.y += act_apply(fmaf(v.y - m.y, sc.y, be.y), slope);
                    o.z += act_apply(fmaf(v.z - m.z, sc.z, be.z), slope);
                    o.w += act_apply(fmaf(v.w - m.w, sc.w, be.w), slope);
                }
            o.x *= 0.25f;
            o.y *= 0.25f;
            o.z *= 0.25f;
            o.w *= 0.25f;
        } else {
            const float4 v = *reinterpret_cast<const float4*>(y + p * ld_y + q * 4);
            o.x = act_apply(fmaf(v.x - m.x, sc.x, be.x), slope);
            o.y = act_apply(fmaf(v.y - m.y, sc.y, be.y), slope);
            o.z = act_apply(fmaf(v.z - m.z, sc.z, be.z), slope);
            o.w = act_apply(fmaf(v.w - m.w, sc.w, be.w), slope);
        }
        float* zp = z + p * ld_z + z_off + q * 4;
        const int rem = C - q * 4;
        if (vec_store && (rem >= 4 || owns_pads)) {   // guarded parameter loads make the pad lanes of `o` zero
            *reinterpret_cast<float4*>(zp) = o;
        } else {
            if (rem > 0) zp[0] = o.x;
            if (rem > 1) zp[1] = o.y;
            if (rem > 2) zp[2] = o.z;
            if (rem > 3) zp[3] = o.w;
        }
    }
}

// sums = [sum g][sum g*xhat], each frames*C long when the statistics are per frame (L.pstride = C)
__global__ void __launch_bounds__(256) bn_act_bwd_apply_kernel(BwdLoader L, const float* __restrict__ sums,
                                                               double count, int training, int FC,
                                                               float* __restrict__ dy, int ld_dy, long rows, int C,
                                                               int nv, int tx_n, int ty_n, long rows_per_block) {
    const int tx = threadIdx.x % tx_n, ty = threadIdx.x / tx_n;
    const int q = blockIdx.x * tx_n + tx;
    if (q >= nv) return;
    const long r0 = (long)blockIdx.y * rows_per_block;
    long r1 = r0 + rows_per_block;
    if (r1 > rows) r1 = rows;
    const float inv = training ? (float)(1.0 / count) : 0.f;
    long cur = -1;
    float4 sc = make_float4(0.f, 0.f, 0.f, 0.f), k1 = sc, k2 = sc;
    for (long r = r0 + ty; r < r1; r += ty_n) {
        const long po = L.poff(r);
        if (po != cur) {            // (re)load the per-channel constants: once for BatchNorm, once per frame otherwise
            cur = po;
            sc = ld4_guard(L.scale + po, q, C);
            if (training) {
                k1 = ld4_guard(sums + po, q, C);
                k2 = ld4_guard(sums + FC + po, q, C);
                k1 = make_float4(k1.x * inv, k1.y * inv, k1.z * inv, k1.w * inv);
                k2 = make_float4(k2.x * inv, k2.y * inv, k2.z * inv, k2.w * inv);
            }
        }
        float4 g, xh;
        L.load(r, q, g, xh);
        float4 o;
        o.x = sc.x * (g.x - k1.x - xh.x * k2.x);
        o.y = sc.y * (g.y - k1.y - xh.y * k2.y);
        o.z = sc.z * (g.z - k1.z - xh.z * k2.z);
        o.w = sc.w * (g.w - k1.w - xh.w * k2.w);
        const int rem = C - q * 4;
        if (rem < 4) {  // keep pad channels of dy at zero
            if (rem < 2) o.y = 0.f;
            if (rem < 3) o.z = 0.f;
            o.w = 0.f;
        }
        *reinterpret_cast<float4*>(dy + r * ld_dy + q * 4) = o;
    }
}

static inline int grid_for(long total, int cap = 2048) {
    long b = (total + 255) / 256;
    if (b < 1) b = 1;
    return (int)(b < cap ? b : cap);
}

}  // namespace

extern "C" {

// ---- general forms: `frames` > 1 = per-frame statistics (InstanceNorm), slope = activation (see act_apply) ----------
size_t mnk_norm_workspace_floats(long rows_per_frame, int frames, int ld) {
    if (rows_per_frame <= 0 || ld <= 0 || frames <= 0) return 0;
    Map2D m = make_map(rows_per_frame, ld);
    return (size_t)frames * m.row_blocks * 2 * ld;
}

int mnk_norm_stats(const float* x, int ld, long rows_per_frame, int frames, int C, float* sums, float* ws,
                   size_t ws_floats, void* stream) {
    MNK_REQUIRE(x && sums && ws && rows_per_frame > 0 && frames > 0 && C > 0 && ld % 4 == 0 && ld >= C);
    Map2D m = make_map(rows_per_frame, ld);
    if (ws_floats < (size_t)frames * m.row_blocks * 2 * ld) {
        set_error("mnk_norm_stats: workspace too small");
        return MNK_EWORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(K_BN_STATS, s, (double)rows_per_frame * frames * C * 4);
    StatsLoader L{x, ld};
    hipLaunchKernelGGL(colsum2_partial_kernel<StatsLoader>, dim3(m.col_tiles, m.row_blocks, frames), dim3(256), 0, s, L,
                       rows_per_frame, ld / 4, ld, m.tx, m.ty, m.rows_per_block, ws);
    hipLaunchKernelGGL(colsum2_final_kernel, dim3(ceil_div(2 * frames * C, 4)), dim3(256), 0, s, ws, m.row_blocks, ld, C,
                       frames, sums);
    MNK_LAUNCH_CHECK();
    return MNK_OK;
}

int mnk_norm_finalize(const float* sums, double count, const float* gamma, float* running_mean, float* running_var,
                      float momentum, float eps, int C, int frames, int update_running, float* mean, float* invstd,
                      float* scale, void* stream) {
    MNK_REQUIRE(sums && gamma && mean && invstd && scale && C > 0 && frames > 0 && count > 0);
    MNK_REQUIRE(!update_running || (frames == 1 && running_mean && running_var && count > 1));
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(ceil_div(frames * C, 256)), dim3(256), 0, s, sums, count, gamma, C,
                       running_mean, running_var, momentum, eps, frames * C, update_running, mean, invstd, scale);
    MNK_LAUNCH_CHECK();
    return MNK_OK;
}

int mnk_norm_act_fwd(const float* y, int ld_y, const float* mean, const float* scale, const float* beta, int per_frame,
                     float* z, int ld_z, int z_off, int N, int H, int W, int C, float slope, int pool, void* stream) {
    MNK_REQUIRE(y && mean && scale && beta && z && N > 0 && H > 0 && W > 0 && C > 0);
    MNK_REQUIRE(ld_y % 4 == 0 && ld_y >= round_up(C, 4) && z_off >= 0 && z_off + C <= ld_z);
    MNK_REQUIRE(!pool || (H >= 2 && W >= 2));
    hipStream_t s = (hipStream_t)stream;
    long total = (long)N * (pool ? H / 2 : H) * (pool ? W / 2 : W) * ((C + 3) / 4);
    ProfScope prof(K_BN_APPLY, s, (double)N * H * W * C * 4 * (pool ? 1.25 : 2.0));
    const int pstride = per_frame ? C : 0;
    if (pool)
        hipLaunchKernelGGL(bn_act_fwd_kernel<1>, dim3(grid_for(total, 4096)), dim3(256), 0, s, y, ld_y, mean, scale, beta,
                           pstride, z, ld_z, z_off, N, H, W, C, slope);
    else
        hipLaunchKernelGGL(bn_act_fwd_kernel<0>, dim3(grid_for(total, 4096)), dim3(256), 0, s, y, ld_y, mean, scale, beta,
                           pstride, z, ld_z, z_off, N, H, W, C, slope);
    MNK_LAUNCH_CHECK();
    return MNK_OK;
}

int mnk_norm_act_bwd_stats(const float* y, int ld_y, const float* dz, int ld_dz, int dz_off, const float* mean,
                           const float* invstd, const float* scale, const float* beta, int per_frame, int N, int H,
                           int W, int C, float slope, int pool, float* sums, float* ws, size_t ws_floats,
                           void* stream) {
    MNK_REQUIRE(y && dz && mean && invstd && scale && beta && sums && ws && N > 0 && H > 0 && W > 0 && C > 0);
    MNK_REQUIRE(ld_y % 4 == 0 && ld_y >= round_up(C, 4) && dz_off >= 0 && dz_off + C <= ld_dz);
    MNK_REQUIRE(!pool || (H >= 2 && W >= 2));
    const int frames = per_frame ? N : 1;
    const long rows = per_frame ? (long)H * W : (long)N * H * W;
    const int ldc = round_up(C, 4);
    Map2D m = make_map(rows, ldc);
    if (ws_floats < (size_t)frames * m.row_blocks * 2 * ldc) {
        set_error("mnk_norm_act_bwd_stats: workspace too small");
        return MNK_EWORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(K_BN_BWD, s, (double)N * H * W * C * 4 * (pool ? 1.25 : 2.0));
    BwdLoader L{y, dz, mean, invstd, scale, beta, ld_y, ld_dz, dz_off, H, W, C, pool, per_frame ? C : 0, slope};
    hipLaunchKernelGGL(colsum2_partial_kernel<BwdLoader>, dim3(m.col_tiles, m.row_blocks, frames), dim3(256), 0, s, L, rows,
                       ldc / 4, ldc, m.tx, m.ty, m.rows_per_block, ws);
    hipLaunchKernelGGL(colsum2_final_kernel, dim3(ceil_div(2 * frames * C, 4)), dim3(256), 0, s, ws, m.row_blocks, ldc, C,
                       frames, sums);
    MNK_LAUNCH_CHECK();
    return MNK_OK;
}

int mnk_norm_act_bwd_apply(const float* y, int ld_y, const float* dz, int ld_dz, int dz_off, const float* mean,
                           const float* invstd, const float* scale, const float* beta, int per_frame, const float* sums,
                           double count, int training, float* dy, int ld_dy, int N, int H, int W, int C, float slope,
                           int pool, void* stream) {
    MNK_REQUIRE(y && dz && mean && invstd && scale && beta && dy && N > 0 && H > 0 && W > 0 && C > 0);
    MNK_REQUIRE(!training || (sums && count > 0));
    MNK_REQUIRE(ld_y % 4 == 0 && ld_y >= round_up(C, 4) && ld_dy % 4 == 0 && ld_dy >= round_up(C, 4));
    MNK_REQUIRE(dz_off >= 0 && dz_off + C <= ld_dz && (!pool || (H >= 2 && W >= 2)));
    const long rows = (long)N * H * W;
    const int ldc = round_up(C, 4);
    Map2D m = make_map(rows, ldc);
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(K_BN_BWD, s, (double)rows * C * 4 * (pool ? 2.25 : 3.0));
    BwdLoader L{y, dz, mean, invstd, scale, beta, ld_y, ld_dz, dz_off, H, W, C, pool, per_frame ? C : 0, slope};
    hipLaunchKernelGGL(bn_act_bwd_apply_kernel, dim3(m.col_tiles, m.row_blocks), dim3(256), 0, s, L, sums, count, training,
                       (per_frame ? N : 1) * C, dy, ld_dy, rows, C, ldc / 4, m.tx, m.ty, m.rows_per_block);
    MNK_LAUNCH_CHECK();
    return MNK_OK;
}

// ---- BatchNorm forms (statistics over all frames; relu flag) -----------------------------------------------------
size_t mnk_bn_workspace_floats(long rows, int ld) { return mnk_norm_workspace_floats(rows, 1, ld); }

int mnk_bn_stats(const float* x, int ld, long rows, int C, float* sums, float* ws, size_t ws_floats, void* stream) {
    return mnk_norm_stats(x, ld, rows, 1, C, sums, ws, ws_floats, stream);
}

int mnk_bn_stats_finish(const float* partial, int row_blocks, int ld, int C, float* sums, void* stream) {
    MNK_REQUIRE(partial && sums && row_blocks > 0 && C > 0 && ld >= C);
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(K_BN_STATS, s, (double)row_blocks * 2 * C * 4);
    hipLaunchKernelGGL(colsum2_final_kernel, dim3(ceil_div(2 * C, 4)), dim3(256), 0, s, partial, row_blocks, ld, C, 1, sums);
    MNK_LAUNCH_CHECK();
    return MNK_OK;
}

int mnk_bn_finalize(const float* sums, double count, const float* gamma, float* running_mean, float* running_var,
                    float momentum, float eps, int C, int update_running, float* mean, float* invstd, float* scale,
                    void* stream) {
    return mnk_norm_finalize(sums, count, gamma, running_mean, running_var, momentum, eps, C, 1, update_running, mean,
                             invstd, scale, stream);
}

int mnk_bn_eval_coeffs(const float* gamma, const float* running_mean, const float* running_var, float eps, int C,
                       float* mean, float* invstd, float* scale, void* stream) {
    MNK_REQUIRE(gamma && running_mean && running_var && mean && invstd && scale && C > 0);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(bn_eval_coeffs_kernel, dim3(ceil_div(C, 256)), dim3(256), 0, s, gamma, running_mean, running_var,
                       eps, C, mean, invstd, scale);
    MNK_LAUNCH_CHECK();
    return MNK_OK;
}

int mnk_bn_act_fwd(const float* y, int ld_y, const float* mean, const float* scale, const float* beta, float* z,
                   int ld_z, int z_off, int N, int H, int W, int C, int relu, int pool, void* stream) {
    MNK_REQUIRE(!pool || (H % 2 == 0 && W % 2 == 0));
    return mnk_norm_act_fwd(y, ld_y, mean, scale, beta, 0, z, ld_z, z_off, N, H, W, C, relu ? 0.f : -1.f, pool, stream);
}

int mnk_bn_act_bwd_stats(const float* y, int ld_y, const float* dz, int ld_dz, int dz_off, const float* mean,
                         const float* invstd, const float* scale, const float* beta, int N, int H, int W, int C,
                         int relu, int pool, float* sums, float* ws, size_t ws_floats, void* stream) {
    return mnk_norm_act_bwd_stats(y, ld_y, dz, ld_dz, dz_off, mean, invstd, scale, beta, 0, N, H, W, C,
                                  relu ? 0.f : -1.f, pool, sums, ws, ws_floats, stream);
}

int mnk_bn_act_bwd_apply(const float* y, int ld_y, const float* dz, int ld_dz, int dz_off, const float* mean,
                         const float* invstd, const float* scale, const float* beta, const float* sums, double count,
                         int training, float* dy, int ld_dy, int N, int H, int W, int C, int relu, int pool,
                         void* stream) {
    return mnk_norm_act_bwd_apply(y, ld_y, dz, ld_dz, dz_off, mean, invstd, scale, beta, 0, sums, count, training, dy,
                                  ld_dy, N, H, W, C, relu ? 0.f : -1.f, pool, stream);
}
}

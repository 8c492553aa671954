// BatchNorm statistics / apply / backward for folded NHWC activations (sync_batchnorm/batchnorm.py:48-78,113-125).
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
    const long r0 = (long)blockIdx.y * rows_per_block;
    long r1 = r0 + rows_per_block;
    if (r1 > rows) r1 = rows;
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
        float* o = partial + (long)blockIdx.y * 2 * ld;
        *reinterpret_cast<float4*>(o + q * 4) = red[0][tx];
        *reinterpret_cast<float4*>(o + ld + q * 4) = red[1][tx];
    }
}

// one wavefront per output column: lanes stride over the row-block partials (fp64 accumulation), the 64 lane sums are
// combined through LDS.  (A serial loop per column was 44 % of the step time in the first MI355X profile.)
__global__ void __launch_bounds__(256) colsum2_final_kernel(const float* __restrict__ partial, int row_blocks, int ld,
                                                            int C, float* __restrict__ sums) {
    __shared__ double sm[256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = blockIdx.x * 4 + wave;
    double acc = 0.0;
    if (i < 2 * C) {
        const int which = i / C, c = i - which * C;
        for (int rb = lane; rb < row_blocks; rb += 64) acc += (double)partial[((long)rb * 2 + which) * ld + c];
    }
    sm[threadIdx.x] = acc;
    __syncthreads();
    if (lane < 8) {
        double t = 0.0;
        for (int j = 0; j < 8; ++j) t += sm[wave * 64 + lane * 8 + j];
        sm[wave * 64 + lane * 8] = t;
    }
    __syncthreads();
    if (lane == 0 && i < 2 * C) {
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

// g = relu'((y-mean)*scale+beta) * dz (dz at half resolution, /4, when pooled); xhat = (y-mean)*invstd
struct BwdLoader {
    const float *y, *dz, *mean, *invstd, *scale, *beta;
    int ld_y, ld_dz, dz_off, H, W, C, relu, pool;
    __device__ __forceinline__ void load(long r, int q, float4& g, float4& xhat) const {
        const float4 v = *reinterpret_cast<const float4*>(y + r * ld_y + q * 4);
        const float4 m = ld4_guard(mean, q, C);
        const float4 is = ld4_guard(invstd, q, C);
        long rz = r;
        float k = 1.f;
        if (pool) {
            int w = (int)(r % W);
            long t = r / W;
            int h = (int)(t % H);
            long n = t / H;
            rz = (n * (H / 2) + (h >> 1)) * (W / 2) + (w >> 1);
            k = 0.25f;
        }
        const float* dp = dz + rz * ld_dz + dz_off + q * 4;
        const int rem = C - q * 4;
        g = make_float4(rem > 0 ? dp[0] * k : 0.f, rem > 1 ? dp[1] * k : 0.f, rem > 2 ? dp[2] * k : 0.f,
                        rem > 3 ? dp[3] * k : 0.f);
        const float4 d = make_float4(v.x - m.x, v.y - m.y, v.z - m.z, v.w - m.w);
        if (relu) {
            const float4 sc = ld4_guard(scale, q, C);
            const float4 be = ld4_guard(beta, q, C);
            if (!(fmaf(d.x, sc.x, be.x) > 0.f)) g.x = 0.f;
            if (!(fmaf(d.y, sc.y, be.y) > 0.f)) g.y = 0.f;
            if (!(fmaf(d.z, sc.z, be.z) > 0.f)) g.z = 0.f;
            if (!(fmaf(d.w, sc.w, be.w) > 0.f)) g.w = 0.f;
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

__global__ void __launch_bounds__(256) bn_finalize_kernel(const float* __restrict__ sums, double count,
                                                          const float* __restrict__ gamma, float* running_mean,
                                                          float* running_var, float momentum, float eps, int C,
                                                          int update_running, float* mean, float* invstd, float* scale) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double m = (double)sums[c] / count;
    double v = (double)sums[C + c] / count - m * m;
    if (v < 0.0) v = 0.0;
    float mf = (float)m, vf = (float)v;
    float is = 1.0f / sqrtf(vf + eps);
    mean[c] = mf;
    invstd[c] = is;
    scale[c] = gamma[c] * is;
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

template <int POOL>
__global__ void __launch_bounds__(256) bn_act_fwd_kernel(const float* __restrict__ y, int ld_y,
                                                         const float* __restrict__ mean,
                                                         const float* __restrict__ scale,
                                                         const float* __restrict__ beta, float* __restrict__ z,
                                                         int ld_z, int z_off, int N, int H, int W, int C, int relu) {
    const int nv = (C + 3) / 4;
    const int Ho = POOL ? H / 2 : H, Wo = POOL ? W / 2 : W;
    const long total = (long)N * Ho * Wo * nv;
    const bool vec_store = ((z_off & 3) == 0) && ((ld_z & 3) == 0);
    const bool owns_pads = z_off == 0 && ld_z == nv * 4;   // z is a plain act: its pad channels are written (as zero) here
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int q = (int)(i % nv);
        long p = i / nv;
        const float4 m = ld4_guard(mean, q, C);
        const float4 sc = ld4_guard(scale, q, C);
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
                    float a = fmaf(v.x - m.x, sc.x, be.x), b = fmaf(v.y - m.y, sc.y, be.y);
                    float c = fmaf(v.z - m.z, sc.z, be.z), d = fmaf(v.w - m.w, sc.w, be.w);
                    if (relu) {
                        a = fmaxf(a, 0.f);
                        b = fmaxf(b, 0.f);
                        c = fmaxf(c, 0.f);
                        d = fmaxf(d, 0.f);
                    }
                    o.x += a;
                    o.y += b;
                    o.z += c;
                    o.w += d;
                }
            o.x *= 0.25f;
            o.y *= 0.25f;
            o.z *= 0.25f;
            o.w *= 0.25f;
        } else {
            const float4 v = *reinterpret_cast<const float4*>(y + p * ld_y + q * 4);
            o.x = fmaf(v.x - m.x, sc.x, be.x);
            o.y = fmaf(v.y - m.y, sc.y, be.y);
            o.z = fmaf(v.z - m.z, sc.z, be.z);
            o.w = fmaf(v.w - m.w, sc.w, be.w);
            if (relu) {
                o.x = fmaxf(o.x, 0.f);
                o.y = fmaxf(o.y, 0.f);
                o.z = fmaxf(o.z, 0.f);
                o.w = fmaxf(o.w, 0.f);
            }
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

__global__ void __launch_bounds__(256) bn_act_bwd_apply_kernel(BwdLoader L, const float* __restrict__ sums,
                                                               double count, int training, float* __restrict__ dy,
                                                               int ld_dy, long rows, int C, int nv, int tx_n, int ty_n,
                                                               long rows_per_block) {
    const int tx = threadIdx.x % tx_n, ty = threadIdx.x / tx_n;
    const int q = blockIdx.x * tx_n + tx;
    if (q >= nv) return;
    const long r0 = (long)blockIdx.y * rows_per_block;
    long r1 = r0 + rows_per_block;
    if (r1 > rows) r1 = rows;
    const float4 sc = ld4_guard(L.scale, q, C);
    float4 k1 = make_float4(0.f, 0.f, 0.f, 0.f), k2 = k1;
    if (training) {
        const float inv = (float)(1.0 / count);
        float s1[4], s2[4];
        for (int j = 0; j < 4; ++j) {
            int c = q * 4 + j;
            s1[j] = c < C ? sums[c] * inv : 0.f;
            s2[j] = c < C ? sums[C + c] * inv : 0.f;
        }
        k1 = make_float4(s1[0], s1[1], s1[2], s1[3]);
        k2 = make_float4(s2[0], s2[1], s2[2], s2[3]);
    }
    for (long r = r0 + ty; r < r1; r += ty_n) {
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

size_t mnk_bn_workspace_floats(long rows, int ld) {
    if (rows <= 0 || ld <= 0) return 0;
    Map2D m = make_map(rows, ld);
    return (size_t)m.row_blocks * 2 * ld;
}

int mnk_bn_stats(const float* x, int ld, long rows, int C, float* sums, float* ws, size_t ws_floats, void* stream) {
    MNK_REQUIRE(x && sums && ws && rows > 0 && C > 0 && ld % 4 == 0 && ld >= C);
    Map2D m = make_map(rows, ld);
    if (ws_floats < (size_t)m.row_blocks * 2 * ld) {
        set_error("mnk_bn_stats: workspace too small");
        return MNK_EWORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(K_BN_STATS, s, (double)rows * C * 4);
    StatsLoader L{x, ld};
    hipLaunchKernelGGL(colsum2_partial_kernel<StatsLoader>, dim3(m.col_tiles, m.row_blocks), dim3(256), 0, s, L, rows,
                       ld / 4, ld, m.tx, m.ty, m.rows_per_block, ws);
    hipLaunchKernelGGL(colsum2_final_kernel, dim3(ceil_div(2 * C, 4)), dim3(256), 0, s, ws, m.row_blocks, ld, C, sums);
    MNK_LAUNCH_CHECK();
    return MNK_OK;
}

int mnk_bn_stats_finish(const float* partial, int row_blocks, int ld, int C, float* sums, void* stream) {
    MNK_REQUIRE(partial && sums && row_blocks > 0 && C > 0 && ld >= C);
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(K_BN_STATS, s, (double)row_blocks * 2 * C * 4);
    hipLaunchKernelGGL(colsum2_final_kernel, dim3(ceil_div(2 * C, 4)), dim3(256), 0, s, partial, row_blocks, ld, C, sums);
    MNK_LAUNCH_CHECK();
    return MNK_OK;
}

int mnk_bn_finalize(const float* sums, double count, const float* gamma, float* running_mean, float* running_var,
                    float momentum, float eps, int C, int update_running, float* mean, float* invstd, float* scale,
                    void* stream) {
    MNK_REQUIRE(sums && gamma && mean && invstd && scale && C > 0 && count > 0);
    MNK_REQUIRE(!update_running || (running_mean && running_var && count > 1));
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(ceil_div(C, 256)), dim3(256), 0, s, sums, count, gamma, running_mean,
                       running_var, momentum, eps, C, update_running, mean, invstd, scale);
    MNK_LAUNCH_CHECK();
    return MNK_OK;
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
    MNK_REQUIRE(y && mean && scale && beta && z && N > 0 && H > 0 && W > 0 && C > 0);
    MNK_REQUIRE(ld_y % 4 == 0 && ld_y >= round_up(C, 4) && z_off >= 0 && z_off + C <= ld_z);
    MNK_REQUIRE(!pool || (H % 2 == 0 && W % 2 == 0));
    hipStream_t s = (hipStream_t)stream;
    long total = (long)N * (pool ? H / 2 : H) * (pool ? W / 2 : W) * ((C + 3) / 4);
    ProfScope prof(K_BN_APPLY, s, (double)N * H * W * C * 4 * (pool ? 1.25 : 2.0));
    if (pool)
        hipLaunchKernelGGL(bn_act_fwd_kernel<1>, dim3(grid_for(total, 4096)), dim3(256), 0, s, y, ld_y, mean, scale, beta,
                           z, ld_z, z_off, N, H, W, C, relu);
    else
        hipLaunchKernelGGL(bn_act_fwd_kernel<0>, dim3(grid_for(total, 4096)), dim3(256), 0, s, y, ld_y, mean, scale, beta,
                           z, ld_z, z_off, N, H, W, C, relu);
    MNK_LAUNCH_CHECK();
    return MNK_OK;
}

int mnk_bn_act_bwd_stats(const float* y, int ld_y, const float* dz, int ld_dz, int dz_off, const float* mean,
                         const float* invstd, const float* scale, const float* beta, int N, int H, int W, int C,
                         int relu, int pool, float* sums, float* ws, size_t ws_floats, void* stream) {
    MNK_REQUIRE(y && dz && mean && invstd && scale && beta && sums && ws && N > 0 && H > 0 && W > 0 && C > 0);
    MNK_REQUIRE(ld_y % 4 == 0 && ld_y >= round_up(C, 4) && dz_off >= 0 && dz_off + C <= ld_dz);
    MNK_REQUIRE(!pool || (H % 2 == 0 && W % 2 == 0));
    const long rows = (long)N * H * W;
    const int ldc = round_up(C, 4);
    Map2D m = make_map(rows, ldc);
    if (ws_floats < (size_t)m.row_blocks * 2 * ldc) {
        set_error("mnk_bn_act_bwd_stats: workspace too small");
        return MNK_EWORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(K_BN_BWD, s, (double)rows * C * 4 * (pool ? 1.25 : 2.0));
    BwdLoader L{y, dz, mean, invstd, scale, beta, ld_y, ld_dz, dz_off, H, W, C, relu, pool};
    hipLaunchKernelGGL(colsum2_partial_kernel<BwdLoader>, dim3(m.col_tiles, m.row_blocks), dim3(256), 0, s, L, rows,
                       ldc / 4, ldc, m.tx, m.ty, m.rows_per_block, ws);
    hipLaunchKernelGGL(colsum2_final_kernel, dim3(ceil_div(2 * C, 4)), dim3(256), 0, s, ws, m.row_blocks, ldc, C, sums);
    MNK_LAUNCH_CHECK();
    return MNK_OK;
}

int mnk_bn_act_bwd_apply(const float* y, int ld_y, const float* dz, int ld_dz, int dz_off, const float* mean,
                         const float* invstd, const float* scale, const float* beta, const float* sums, double count,
                         int training, float* dy, int ld_dy, int N, int H, int W, int C, int relu, int pool,
                         void* stream) {
    MNK_REQUIRE(y && dz && mean && invstd && scale && beta && dy && N > 0 && H > 0 && W > 0 && C > 0);
    MNK_REQUIRE(!training || (sums && count > 0));
    MNK_REQUIRE(ld_y % 4 == 0 && ld_y >= round_up(C, 4) && ld_dy % 4 == 0 && ld_dy >= round_up(C, 4));
    MNK_REQUIRE(dz_off >= 0 && dz_off + C <= ld_dz && (!pool || (H % 2 == 0 && W % 2 == 0)));
    const long rows = (long)N * H * W;
    const int ldc = round_up(C, 4);
    Map2D m = make_map(rows, ldc);
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(K_BN_BWD, s, (double)rows * C * 4 * (pool ? 2.25 : 3.0));
    BwdLoader L{y, dz, mean, invstd, scale, beta, ld_y, ld_dz, dz_off, H, W, C, relu, pool};
    hipLaunchKernelGGL(bn_act_bwd_apply_kernel, dim3(m.col_tiles, m.row_blocks), dim3(256), 0, s, L, sums, count,
                       training, dy, ld_dy, rows, C, ldc / 4, m.tx, m.ty, m.rows_per_block);
    MNK_LAUNCH_CHECK();
    return MNK_OK;
}
}

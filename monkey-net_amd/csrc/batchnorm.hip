// Normalisation statistics / apply / backward for folded NHWC activations: BatchNorm (sync_batchnorm/batchnorm.py:
// 48-78,113-125; statistics over all frames) and, with per-frame statistics, the discriminator's InstanceNorm3d
// (modules/discriminator.py:19-22,29-30); fused with ReLU / LeakyReLU(slope) and the (1,2,2) average pool.
// HBM-bound: float4 along channels, a 2-D thread map (channel-quad x row) so that every thread keeps its
// channel quad in registers while it walks rows; column sums are finished in LDS and by a tiny second pass.
#include <stdlib.h>

#include "mnk_common.h"
#include "p2p.h"
namespace mnk {
bool p2p_launch_info(void* handle, PeerTable* peers, int* rank, int* world, unsigned** state);      // p2p.hip
}

using namespace mnk;

namespace {

struct Map2D {
    int tx, ty, col_tiles, row_blocks;
    long rows_per_block;
};

static int g_bn_rpt = tuning_knob("bn_rpt", &g_bn_rpt, 4);            // rows per thread before a layer is cut (A/B: 8 -> 4: 11.51 -> 11.44 ms per step)
static int g_bn_blocks = tuning_knob("bn_blocks", &g_bn_blocks, 1024);   // into more row blocks; block cap
static int g_bn_exact_tx = tuning_knob("bn_exact_tx", &g_bn_exact_tx, 1);   // 0: power-of-two quad lanes per block (rounds 1-5)

static Map2D make_map(long rows, int ld, int rows_per_thread = 0, int want_blocks = 0) {
    if (rows_per_thread <= 0) rows_per_thread = g_bn_rpt;
    if (want_blocks <= 0) want_blocks = g_bn_blocks;
    Map2D m;
    int nv = ld / 4;
    int tx = 1;
    while (tx < nv && tx < 64) tx <<= 1;
    // (round 6) a channel count whose quads are no power of two -- 45 -> 12 quads (the refinement stack: the largest tensors of a
    // step), 44 -> 11, 66 -> 17 -- left a quarter to a half of every block idle with the power-of-two map: tx = the quad count
    // itself, ty = floor(256 / tx) row lanes, the few left-over threads idle (252 of 256 work on 12 quads instead of 192)
    if (g_bn_exact_tx && nv <= 64) tx = nv;
    m.tx = tx;
    m.ty = 256 / tx;
    m.col_tiles = (nv + tx - 1) / tx;
    long want = want_blocks / m.col_tiles;   // default ~4 blocks per CU in total
    if (want < 1) want = 1;
    long min_rows = (long)m.ty * rows_per_thread;  // at least this many rows per thread before splitting further
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

// first step of a halving tree over n entries: the largest power of two below n (n >= 2), 0 for n = 1
__device__ __forceinline__ int tree_half(int n) {
    int s = 1;
    while (s * 2 < n) s <<= 1;
    return n > 1 ? s : 0;
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
    if (q < nv && ty < ty_n) {        // (ty >= ty_n: the left-over threads of a map whose tx_n does not divide 256)
        typename F::State st;         // per-thread channel constants (the column quad q is fixed per thread)
        f.init(st);
#pragma unroll 2
        for (long r = r0 + ty; r < r1; r += ty_n) {
            float4 va, vb;
            f(r, q, st, va, vb);
            a = f4_add(a, va);
            b = f4_add(b, vb);
        }
    }
    red[0][threadIdx.x] = a;
    red[1][threadIdx.x] = b;
    __syncthreads();
    for (int s = tree_half(ty_n); s > 0; s >>= 1) {       // any ty_n; the halving tree of rounds 1-5 when it is a power of two
        if (ty < s && ty + s < ty_n) {
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

// A lane's share of one column of the row-block partials: every 64th entry from `lane` on, fp64.  Four entries in flight
// (a load -> add chain per entry made the second stage of a 1024 ... 2048-partial layer a 16 ... 32-deep latency chain); the
// order is fixed: four interleaved sub-sums, (s0 + s1) + (s2 + s3).  `stride` = floats between consecutive row blocks.
__device__ __forceinline__ double lane_colsum(const float* __restrict__ p, int lane, int row_blocks, long stride) {
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int rb = lane;
    for (; rb + 192 < row_blocks; rb += 256) {
        const float v0 = p[(long)rb * stride], v1 = p[(long)(rb + 64) * stride], v2 = p[(long)(rb + 128) * stride],
                    v3 = p[(long)(rb + 192) * stride];
        s0 += (double)v0;
        s1 += (double)v1;
        s2 += (double)v2;
        s3 += (double)v3;
    }
    for (; rb < row_blocks; rb += 64) s0 += (double)p[(long)rb * stride];
    return (s0 + s1) + (s2 + s3);
}

// the same for the two sums of one channel at once (eight entries in flight); `q` = p + ld
__device__ __forceinline__ void lane_colsum_pair(const float* __restrict__ p, int ld, int lane, int row_blocks, long stride,
                                                 double& a1, double& a2) {
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0, t0 = 0.0, t1 = 0.0, t2 = 0.0, t3 = 0.0;
    const float* q = p + ld;
    int rb = lane;
    for (; rb + 192 < row_blocks; rb += 256) {
        const float v0 = p[(long)rb * stride], v1 = p[(long)(rb + 64) * stride], v2 = p[(long)(rb + 128) * stride],
                    v3 = p[(long)(rb + 192) * stride];
        const float w0 = q[(long)rb * stride], w1 = q[(long)(rb + 64) * stride], w2 = q[(long)(rb + 128) * stride],
                    w3 = q[(long)(rb + 192) * stride];
        s0 += (double)v0;
        s1 += (double)v1;
        s2 += (double)v2;
        s3 += (double)v3;
        t0 += (double)w0;
        t1 += (double)w1;
        t2 += (double)w2;
        t3 += (double)w3;
    }
    for (; rb < row_blocks; rb += 64) {
        s0 += (double)p[(long)rb * stride];
        t0 += (double)q[(long)rb * stride];
    }
    a1 = (s0 + s1) + (s2 + s3);
    a2 = (t0 + t1) + (t2 + t3);
}

// one wavefront per output column: lanes stride over the row-block partials (fp64 accumulation), the 64 lane sums are
// combined through LDS.  (A serial loop per column was 44 % of the step time in the first MI355X profile.)
__global__ void __launch_bounds__(256) colsum2_final_kernel(const float* __restrict__ partial, int row_blocks, int ld,
                                                            int C, int frames, float* __restrict__ sums, int nwhich) {
    // sums[which][frame][c] = sum_rb partial[frame][rb][which][c], which < nwhich (1: first sum only)
    __shared__ double sm[256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = blockIdx.x * 4 + wave;
    const int FC = frames * C;
    double acc = 0.0;
    if (i < nwhich * FC) {
        const int which = i / FC, rem = i - which * FC;
        const int f = rem / C, c = rem - f * C;
        const float* pb = partial + (long)f * row_blocks * 2 * ld;
        acc = lane_colsum(pb + (long)which * ld + c, lane, row_blocks, 2L * ld);
    }
    sm[threadIdx.x] = acc;
    __syncthreads();
    if (lane < 8) {
        double t = 0.0;
        for (int j = 0; j < 8; ++j) t += sm[wave * 64 + lane * 8 + j];
        sm[wave * 64 + lane * 8] = t;
    }
    __syncthreads();
    if (lane == 0 && i < nwhich * FC) {
        double t = 0.0;
        for (int j = 0; j < 8; ++j) t += sm[wave * 64 + j * 8];
        sums[i] = (float)t;
    }
}

// colsum2_final_kernel with the SyncBN exchange of one node inside (csrc/p2p.hip's protocol): the wavefront that finishes the
// sum of column i pushes it into every rank's mailbox and adds the `world` contributions in rank order -- the second stage of
// the statistics and their all-reduce are ONE launch, so a norm layer of a data-parallel run costs as many launches as on one
// GPU (it was: second stage, then a collective: 84 extra launches per iteration).  local (optional): this rank's own sums (the
// scale / shift gradients of the backward pass are local sums; they are averaged with all other gradients later).
__global__ void __launch_bounds__(256) colsum2_final_sync_kernel(const float* __restrict__ partial, int row_blocks, int ld, int C,
                                                                 float* __restrict__ local, float* __restrict__ global_sums,
                                                                 PeerTable peers, int rank, int world, unsigned* state,
                                                                 unsigned long long timeout_ticks) {
    __shared__ double sm[256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = blockIdx.x * 4 + wave;
    const unsigned seq = state[0] + 1;
    const int slot = (int)(seq % P2P_SLOTS);
    double acc = 0.0;
    if (i < 2 * C) {
        const int which = i / C, c = i - which * C;
        acc = lane_colsum(partial + (long)which * ld + c, lane, row_blocks, 2L * ld);
    }
    sm[threadIdx.x] = acc;
    __syncthreads();
    if (lane < 8) {
        double t = 0.0;
        for (int j = 0; j < 8; ++j) t += sm[wave * 64 + lane * 8 + j];
        sm[wave * 64 + lane * 8] = t;
    }
    __syncthreads();
    if (i < 2 * C) {                       // (wave-uniform)
        double t = 0.0;
        for (int j = 0; j < 8; ++j) t += sm[wave * 64 + j * 8];       // every lane: the bits colsum2_final_kernel's lane 0 makes
        const float mine = (float)t;
        const float all = p2p_exchange_value(peers, rank, world, slot, seq, i, mine, state, timeout_ticks);
        if (lane == 0) {
            if (local) local[i] = mine;
            global_sums[i] = all;
        }
    }
    p2p_finish_launch(state, seq);
}

static int launch_final_sync(void* p2p, const float* partial, int row_blocks, int ld, int C, float* local, float* global_sums,
                             int timeout_ms, hipStream_t s) {
    PeerTable peers;
    int rank = 0, world = 0;
    unsigned* state = nullptr;
    if (!p2p_launch_info(p2p, &peers, &rank, &world, &state) || 2 * C > P2P_MAXF || timeout_ms <= 0) {
        set_error("synchronised BatchNorm statistics: the peer-to-peer exchange is not connected, or more than %d channels",
                  P2P_MAXF / 2);
        return MNK_ECOMM;
    }
    hipLaunchKernelGGL(colsum2_final_sync_kernel, dim3(ceil_div(2 * C, 4)), dim3(256), 0, s, partial, row_blocks, ld, C, local,
                       global_sums, peers, rank, world, state, (unsigned long long)timeout_ms * 100000ull);
    return MNK_OK;
}

struct StatsLoader {
    const float* x;
    int ld;
    struct State {};
    __device__ __forceinline__ void init(State&) const {}
    __device__ __forceinline__ void operator()(long r, int q, State&, float4& a, float4& b) const {
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
    // per-thread channel constants: loaded once for BatchNorm, once per frame for per-frame statistics
    struct State {
        long cur;
        float4 m, is, sc, be;
    };
    __device__ __forceinline__ void init(State& st) const { st.cur = -1; }
    __device__ __forceinline__ long poff(long r) const {
        return pstride ? (long)((unsigned)r / (unsigned)(H * W)) * pstride : 0;     // rows < 2^31 (host check)
    }
    __device__ __forceinline__ void load(long r, int q, State& st, float4& g, float4& xhat) const {
        const long po = poff(r);
        if (po != st.cur) {
            st.cur = po;
            st.m = ld4_guard(mean + po, q, C);
            st.is = ld4_guard(invstd + po, q, C);
            st.sc = ld4_guard(scale + po, q, C);
            st.be = ld4_guard(beta, q, C);
        }
        const float4 v = *reinterpret_cast<const float4*>(y + r * ld_y + q * 4);
        long rz = r;
        float k = 1.f;
        if (pool) {
            const unsigned ur = (unsigned)r, t = ur / (unsigned)W;
            const int w = (int)(ur - t * (unsigned)W);
            const unsigned n = t / (unsigned)H;
            const int h = (int)(t - n * (unsigned)H);
            rz = ((long)n * (H / 2) + (h >> 1)) * (W / 2) + (w >> 1);
            k = ((h >> 1) < H / 2 && (w >> 1) < W / 2) ? 0.25f : 0.f;   // odd H / W: the last row / column is not pooled
            if (k == 0.f) rz = 0;
        }
        const float* dp = dz + rz * ld_dz + dz_off + q * 4;
        const int rem = C - q * 4;
        if (rem >= 4 && ((ld_dz | dz_off) & 3) == 0) {
            const float4 d4 = *reinterpret_cast<const float4*>(dp);
            g = make_float4(d4.x * k, d4.y * k, d4.z * k, d4.w * k);
        } else {
            g = make_float4(rem > 0 ? dp[0] * k : 0.f, rem > 1 ? dp[1] * k : 0.f, rem > 2 ? dp[2] * k : 0.f,
                            rem > 3 ? dp[3] * k : 0.f);
        }
        const float4 d = make_float4(v.x - st.m.x, v.y - st.m.y, v.z - st.m.z, v.w - st.m.w);
        if (slope >= 0.f) {
            if (!(fmaf(d.x, st.sc.x, st.be.x) > 0.f)) g.x *= slope;
            if (!(fmaf(d.y, st.sc.y, st.be.y) > 0.f)) g.y *= slope;
            if (!(fmaf(d.z, st.sc.z, st.be.z) > 0.f)) g.z *= slope;
            if (!(fmaf(d.w, st.sc.w, st.be.w) > 0.f)) g.w *= slope;
        }
        xhat = make_float4(d.x * st.is.x, d.y * st.is.y, d.z * st.is.z, d.w * st.is.w);
    }
    __device__ __forceinline__ void operator()(long r, int q, State& st, float4& a, float4& b) const {
        float4 g, xh;
        load(r, q, st, g, xh);
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

// colsum2_final + bn_finalize in one launch (single-process BatchNorm: nothing sits between the two): one wavefront
// per channel sums both statistics over the row-block partials in fp64 and lane 0 derives mean / invstd / scale and
// the running-statistics update.
__global__ void __launch_bounds__(256) bn_final_finalize_kernel(const float* __restrict__ partial, int row_blocks, int ld,
                                                                int C, double count, const float* __restrict__ gamma,
                                                                float* running_mean, float* running_var, float momentum,
                                                                float eps, int update_running, float* sums, float* mean,
                                                                float* invstd, float* scale) {
    __shared__ double sm[2][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 4 + wave;
    double a1 = 0.0, a2 = 0.0;
    if (c < C) {
        lane_colsum_pair(partial + c, ld, lane, row_blocks, 2L * ld, a1, a2);      // the bits of lane_colsum per sum
    }
    sm[0][threadIdx.x] = a1;
    sm[1][threadIdx.x] = a2;
    __syncthreads();
    if (lane < 8) {
        double t1 = 0.0, t2 = 0.0;
        for (int j = 0; j < 8; ++j) {
            t1 += sm[0][wave * 64 + lane * 8 + j];
            t2 += sm[1][wave * 64 + lane * 8 + j];
        }
        sm[0][wave * 64 + lane * 8] = t1;
        sm[1][wave * 64 + lane * 8] = t2;
    }
    __syncthreads();
    if (lane == 0 && c < C) {
        double t1 = 0.0, t2 = 0.0;
        for (int j = 0; j < 8; ++j) {
            t1 += sm[0][wave * 64 + j * 8];
            t2 += sm[1][wave * 64 + j * 8];
        }
        // the same arithmetic as colsum2_final_kernel -> bn_finalize_kernel (sums round-trip through fp32)
        const float s1 = (float)t1, s2 = (float)t2;
        if (sums) {
            sums[c] = s1;
            sums[C + c] = s2;
        }
        const double m = (double)s1 / count;
        double v = (double)s2 / count - m * m;
        if (v < 0.0) v = 0.0;
        const float mf = (float)m, vf = (float)v;
        const float is = 1.0f / sqrtf(vf + eps);
        mean[c] = mf;
        invstd[c] = is;
        scale[c] = gamma[c] * is;
        if (update_running) {
            const float unbiased = (float)(v * count / (count - 1.0));
            running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mf;
            running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
        }
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

// thread (tx = channel quad, ty = output pixel lane): the per-channel constants stay in registers
// the statistics of the FIN form: finished column sums [sum x][sum x^2] of the whole (cross-rank) batch; every block derives the
// constants of its channels from them with bn_finalize_kernel's arithmetic, the first row block also leaves mean / inv-std /
// scale for the backward pass and updates the running statistics -- bn_finalize as no launch of its own (SyncBN path: the sums
// come out of an all-reduce, so the fused second stage of the single-process path does not apply)
struct FwdFinalize {
    const float* sums;
    const float* gamma;
    float *running_mean, *running_var, *mean, *invstd, *scale;
    double count;
    float momentum, eps;
    int update_running;
};

template <int POOL, bool FIN = false>
__global__ void __launch_bounds__(256) bn_act_fwd_kernel(const float* __restrict__ y, int ld_y,
                                                         const float* __restrict__ mean,
                                                         const float* __restrict__ scale,
                                                         const float* __restrict__ beta, int pstride,
                                                         float* __restrict__ z, int ld_z, int z_off, int N, int H,
                                                         int W, int C, float slope, int tx_n, int ty_n,
                                                         long rows_per_block, FwdFinalize fin = FwdFinalize()) {
    const int nv = (C + 3) / 4;
    const int tx = threadIdx.x % tx_n, ty = threadIdx.x / tx_n;
    const int q = blockIdx.x * tx_n + tx;
    if (q >= nv || ty >= ty_n) return;
    const int Ho = POOL ? H / 2 : H, Wo = POOL ? W / 2 : W;
    const long rows = (long)N * Ho * Wo;              // output pixels, < 2^31 (host check)
    const unsigned HWo = (unsigned)(Ho * Wo);
    const long r0 = (long)blockIdx.y * rows_per_block;
    long r1 = r0 + rows_per_block;
    if (r1 > rows) r1 = rows;
    const bool vec_store = ((z_off & 3) == 0) && ((ld_z & 3) == 0);
    const bool owns_pads = z_off == 0 && ld_z == nv * 4;   // z is a plain act: its pad channels are written (as zero) here
    const int rem = C - q * 4;
    long cur = -1;
    float4 m = make_float4(0.f, 0.f, 0.f, 0.f), sc = m;
    const float4 be = ld4_guard(beta, q, C);
    if (FIN) {
        float mm[4] = {0.f, 0.f, 0.f, 0.f}, ss[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = q * 4 + e;
            if (c >= C) continue;
            const double mu = (double)fin.sums[c] / fin.count;
            double var = (double)fin.sums[C + c] / fin.count - mu * mu;
            if (var < 0.0) var = 0.0;
            const float mf = (float)mu, vf = (float)var;
            const float is = 1.0f / sqrtf(vf + fin.eps);
            mm[e] = mf;
            ss[e] = fin.gamma[c] * is;
            if (blockIdx.y == 0 && ty == 0) {
                fin.mean[c] = mf;
                fin.invstd[c] = is;
                fin.scale[c] = ss[e];
                if (fin.update_running) {
                    const float unbiased = (float)(var * fin.count / (fin.count - 1.0));
                    fin.running_mean[c] = (1.f - fin.momentum) * fin.running_mean[c] + fin.momentum * mf;
                    fin.running_var[c] = (1.f - fin.momentum) * fin.running_var[c] + fin.momentum * unbiased;
                }
            }
        }
        m = make_float4(mm[0], mm[1], mm[2], mm[3]);
        sc = make_float4(ss[0], ss[1], ss[2], ss[3]);
        cur = 0;
    }
#pragma unroll 2
    for (long p = r0 + ty; p < r1; p += ty_n) {
        const long po = pstride ? (long)((unsigned)p / HWo) * pstride : 0;     // per-frame statistics (InstanceNorm)
        if (!FIN && po != cur) {
            cur = po;
            m = ld4_guard(mean + po, q, C);
            sc = ld4_guard(scale + po, q, C);
        }
        float4 o;
        if (POOL) {
            const unsigned up = (unsigned)p, t = up / (unsigned)Wo;
            const int wo = (int)(up - t * (unsigned)Wo);
            const unsigned n = t / (unsigned)Ho;
            const int ho = (int)(t - n * (unsigned)Ho);
            o = make_float4(0.f, 0.f, 0.f, 0.f);
            const float* yb = y + (((long)n * H + 2 * ho) * W + 2 * wo) * ld_y + q * 4;
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    const float4 v = *reinterpret_cast<const float4*>(yb + ((long)dy * W + dx) * ld_y);
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
                                                               int nv, int tx_n, int ty_n, long rows_per_block,
                                                               float* __restrict__ dy_partial,
                                                               const float* __restrict__ addend, int ld_add) {
    // dy_partial (optional): [gridDim.y][2][4 * nv] with the column sums of the written dy in slot 0 -- the bias
    // gradient of the convolution in front of this norm layer (Conv3x3Fn.backward), saving its pass over dy
    // addend (optional): a second gradient of the same tensor, added here -- the skip path of a residual block whose
    // first norm layer this is (util.py:58-67: out += x); saves autograd's accumulation pass over both gradients
    __shared__ float4 red[256];
    const int tx = threadIdx.x % tx_n, ty = threadIdx.x / tx_n;
    const int q = blockIdx.x * tx_n + tx;
    float4 csum = make_float4(0.f, 0.f, 0.f, 0.f);
    if (q < nv && ty < ty_n) {
    const long r0 = (long)blockIdx.y * rows_per_block;
    long r1 = r0 + rows_per_block;
    if (r1 > rows) r1 = rows;
    const float inv = training ? (float)(1.0 / count) : 0.f;
    long cur = -1;
    float4 sc = make_float4(0.f, 0.f, 0.f, 0.f), k1 = sc, k2 = sc;
    BwdLoader::State st;
    L.init(st);
#pragma unroll 2
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
        L.load(r, q, st, g, xh);
        float4 o;
        o.x = sc.x * (g.x - k1.x - xh.x * k2.x);
        o.y = sc.y * (g.y - k1.y - xh.y * k2.y);
        o.z = sc.z * (g.z - k1.z - xh.z * k2.z);
        o.w = sc.w * (g.w - k1.w - xh.w * k2.w);
        if (addend) o = f4_add(o, *reinterpret_cast<const float4*>(addend + r * ld_add + q * 4));
        const int rem = C - q * 4;
        if (rem < 4) {  // keep pad channels of dy at zero
            if (rem < 2) o.y = 0.f;
            if (rem < 3) o.z = 0.f;
            o.w = 0.f;
        }
        *reinterpret_cast<float4*>(dy + r * ld_dy + q * 4) = o;
        csum = f4_add(csum, o);
    }
    }
    if (dy_partial) {
        red[threadIdx.x] = csum;
        __syncthreads();
        for (int s = tree_half(ty_n); s > 0; s >>= 1) {
            if (ty < s && ty + s < ty_n) red[threadIdx.x] = f4_add(red[threadIdx.x], red[threadIdx.x + s * tx_n]);
            __syncthreads();
        }
        if (ty == 0 && q < nv)
            *reinterpret_cast<float4*>(dy_partial + (long)blockIdx.y * 2 * (4 * nv) + q * 4) = red[tx];
    }
}


// ---- small layers: the whole training-mode BatchNorm (+ReLU, +2x2 pool) of one layer in ONE launch ---------------------
// The deep levels of the hourglasses (4x4 ... 16x16 maps, 256 ... 1024 channels) have a few thousand pixel rows: the
// general path spends four launches of ~5 us each on them per direction (split-K reduction of the convolution in front,
// statistics partials, second stage + finalisation, apply).  Here one block owns a tile of TXQ channel quads over ALL
// rows, so the column sums never leave the block:
//   forward : [sum of the convolution's split-K partials + bias -> y] -> sum / sum of squares -> mean, inv-std, scale,
//             running statistics -> z = [pool] relu((y - mean) * scale + beta)
//   backward: sum g, sum g*xhat (= dbeta, dgamma) -> dy = scale * (g - sum_g / n - xhat * sum_gx / n)
// Single rank only (SyncBN has an all-reduce between the two halves).  rows <= mnk_bn_small_rows().
// defaults from the MI355X A/B (profiles/README.md): 512 rows with one channel quad per block is the only setting that beats
// the general four-launch path (12.39 vs 12.42 ms per iteration); 2048 rows / 4 quads per block LOSES 0.3 ms -- a handful of
// blocks walking 32+ rows each is slower than four well-filled launches
static int g_small_rows = tuning_knob("bn_small_rows", &g_small_rows, 512);
static int g_small_txn = tuning_knob("bn_small_txn", &g_small_txn, 1);           // channel quads per block: 0 = by channel count
// forward kernel: threads per block (256 / 512 / 1024) and channel quads per block.  It sums the split-K partials of the
// convolution in front (8 ... 32 x rows x C floats): with one quad per block every lane reads 16 of the 128 bytes of its
// line and a block of 256 threads is one wave per SIMD.  1024 threads over up to 8 quads (a full line per row, at least
// 32 blocks) is the measured setting: 11.25 -> 11.19 ms per iteration (profiles/r02_knob_ab_log.txt, visit 42);
// MNK_BN_SMALL_FWD_TXN = 0: by channel count, -1: as MNK_BN_SMALL_TXN
static int g_small_fwd_threads = tuning_knob("bn_small_fwd_threads", &g_small_fwd_threads, 1024);
static int g_small_fwd_txn = tuning_knob("bn_small_fwd_txn", &g_small_fwd_txn, 0);
// 1: the backward kernel takes the same shape; 0 (default): 256 threads x MNK_BN_SMALL_TXN quads -- it reads y and dz once,
// no split partials, and was 0.03 ms per iteration faster that way (visit 43)
static int g_small_bwd_shape = tuning_knob("bn_small_bwd_shape", &g_small_bwd_shape, 0);

__device__ __forceinline__ void small_tree_sum2(float4* red0, float4* red1, float4& a, float4& b, int tx_n, int ty_n, int tx,
                                                int ty) {
    red0[threadIdx.x] = a;
    red1[threadIdx.x] = b;
    __syncthreads();
    for (int s = ty_n >> 1; s > 0; s >>= 1) {
        if (ty < s) {
            red0[threadIdx.x] = f4_add(red0[threadIdx.x], red0[threadIdx.x + s * tx_n]);
            red1[threadIdx.x] = f4_add(red1[threadIdx.x], red1[threadIdx.x + s * tx_n]);
        }
        __syncthreads();
    }
    a = red0[tx];          // every thread gets the totals of its quad (thread layout: index = ty * tx_n + tx)
    b = red1[tx];
    __syncthreads();
}

struct SmallFwdArgs {
    const float* ws;       // split-K partials [split][phase][M][ldw] of the convolution in front, or NULL (y is complete)
    int splits, ldw, phases;
    const float* bias;
    float* y;
    int ld_y, N, H, W, C;  // y geometry (the up-sampled size for the sub-pixel form)
    const float *gamma, *beta;
    float *running_mean, *running_var;
    float momentum, eps;
    float *mean, *invstd, *scale, *z;
    int ld_z, relu, pool, tx_n;
};

// Exchange: what happens to the block's column sums before they become statistics -- nothing on one GPU (SmallNoExchange), the
// peer-to-peer exchange over the ranks of the node (SmallP2PExchange: one channel quad per block, tx_n = 1)
struct SmallNoExchange {
    __device__ __forceinline__ int world_size() const { return 1; }
    __device__ __forceinline__ void all_ranks(float4&, float4&, int, int) const {}
    __device__ __forceinline__ void finish() const {}
};

template <class Exchange>
__device__ __forceinline__ void bn_small_fwd_body(const SmallFwdArgs& a, const Exchange& xch) {
    __shared__ float4 red0[1024], red1[1024];
    const int tx_n = a.tx_n, ty_n = blockDim.x / tx_n;
    const int tx = threadIdx.x % tx_n, ty = threadIdx.x / tx_n;
    const int nv = a.ld_y / 4, q = blockIdx.x * tx_n + tx;
    const bool qok = q < nv;
    const int rows = a.N * a.H * a.W, rem = a.C - q * 4;
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
    if (qok) {
        if (a.ws) {
            const float4 bv = a.bias ? ld4_guard(a.bias, q, a.C) : make_float4(0.f, 0.f, 0.f, 0.f);
            const int Hl = a.H >> 1, Wl = a.W >> 1;
            const long Mlow = (long)a.N * Hl * Wl, prows = a.phases > 1 ? 4 * Mlow : (long)rows;
            for (int r = ty; r < rows; r += ty_n) {
                long pr = r;
                if (a.phases > 1) {          // output pixel (n, Y, X) <- row (n, Y>>1, X>>1) of phase 2 (Y&1) + (X&1)
                    const int X = r % a.W, t = r / a.W, Y = t % a.H, n = t / a.H;
                    pr = (long)((Y & 1) * 2 + (X & 1)) * Mlow + ((long)n * Hl + (Y >> 1)) * Wl + (X >> 1);
                }
                // the splits are added in order, but eight (then four) partials are fetched before the first add: with one
                // block per channel quad there is a single wave per SIMD, and a load -> add chain over 8 ... 32 splits was
                // the whole kernel (23 us for 512 rows)
                float4 v = bv;
                const float* wp = a.ws + pr * a.ldw + q * 4;
                const long sstride = prows * a.ldw;
                int sp = 0;
                for (; sp + 8 <= a.splits; sp += 8) {
                    float4 l[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) l[e] = *reinterpret_cast<const float4*>(wp + (long)(sp + e) * sstride);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v = f4_add(v, l[e]);
                }
                if (sp + 4 <= a.splits) {
                    float4 l[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) l[e] = *reinterpret_cast<const float4*>(wp + (long)(sp + e) * sstride);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v = f4_add(v, l[e]);
                    sp += 4;
                }
                for (; sp < a.splits; ++sp) v = f4_add(v, *reinterpret_cast<const float4*>(wp + (long)sp * sstride));
                if (rem < 4) {               // columns beyond Cout hold the results of clamped weight rows
                    if (rem < 2) v.y = 0.f;
                    if (rem < 3) v.z = 0.f;
                    v.w = 0.f;
                    if (rem < 1) v.x = 0.f;
                }
                *reinterpret_cast<float4*>(a.y + (long)r * a.ld_y + q * 4) = v;
                s1 = f4_add(s1, v);
                s2 = f4_fma(v, v, s2);
            }
        } else {
            for (int r = ty; r < rows; r += ty_n) {
                const float4 v = *reinterpret_cast<const float4*>(a.y + (long)r * a.ld_y + q * 4);
                s1 = f4_add(s1, v);
                s2 = f4_fma(v, v, s2);
            }
        }
    }
    small_tree_sum2(red0, red1, s1, s2, tx_n, ty_n, tx, ty);       // also orders the y writes of the block before the reads below
    xch.all_ranks(s1, s2, q, a.C);                                 // (several ranks: the sums over all of them, in rank order)
    const double count = (double)rows * xch.world_size();
    float mq[4], sq[4], iq[4];
    const float t1[4] = {s1.x, s1.y, s1.z, s1.w}, t2[4] = {s2.x, s2.y, s2.z, s2.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int c = q * 4 + e;
        const double m = (double)t1[e] / count;
        double v = (double)t2[e] / count - m * m;
        if (v < 0.0) v = 0.0;
        const float mf = (float)m, is = 1.0f / sqrtf((float)v + a.eps);
        mq[e] = mf;
        iq[e] = is;
        sq[e] = (qok && c < a.C) ? a.gamma[c] * is : 0.f;
        if (ty == 0 && qok && c < a.C) {
            a.mean[c] = mf;
            a.invstd[c] = is;
            a.scale[c] = sq[e];
            const float unbiased = (float)(v * count / (count - 1.0));
            a.running_mean[c] = (1.f - a.momentum) * a.running_mean[c] + a.momentum * mf;
            a.running_var[c] = (1.f - a.momentum) * a.running_var[c] + a.momentum * unbiased;
        }
    }
    if (!qok) {
        xch.finish();
        return;
    }
    const float4 be = ld4_guard(a.beta, q, a.C);
    const float slope = a.relu ? 0.f : -1.f;
    const float4 m = make_float4(mq[0], mq[1], mq[2], mq[3]), sc = make_float4(sq[0], sq[1], sq[2], sq[3]);
    if (a.pool) {
        const int Ho = a.H / 2, Wo = a.W / 2, orows = a.N * Ho * Wo;
        for (int p = ty; p < orows; p += ty_n) {
            const int wo = p % Wo, t = p / Wo, ho = t % Ho, n = t / Ho;
            const float* yb = a.y + (((long)n * a.H + 2 * ho) * a.W + 2 * wo) * a.ld_y + q * 4;
            float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    const float4 v = *reinterpret_cast<const float4*>(yb + ((long)dy * a.W + dx) * a.ld_y);
                    o.x += act_apply(fmaf(v.x - m.x, sc.x, be.x), slope);
                    o.y += act_apply(fmaf(v.y - m.y, sc.y, be.y), slope);
                    o.z += act_apply(fmaf(v.z - m.z, sc.z, be.z), slope);
                    o.w += act_apply(fmaf(v.w - m.w, sc.w, be.w), slope);
                }
            *reinterpret_cast<float4*>(a.z + (long)p * a.ld_z + q * 4) = make_float4(o.x * 0.25f, o.y * 0.25f, o.z * 0.25f, o.w * 0.25f);
        }
    } else {
        for (int r = ty; r < rows; r += ty_n) {
            const float4 v = *reinterpret_cast<const float4*>(a.y + (long)r * a.ld_y + q * 4);
            float4 o;
            o.x = act_apply(fmaf(v.x - m.x, sc.x, be.x), slope);
            o.y = act_apply(fmaf(v.y - m.y, sc.y, be.y), slope);
            o.z = act_apply(fmaf(v.z - m.z, sc.z, be.z), slope);
            o.w = act_apply(fmaf(v.w - m.w, sc.w, be.w), slope);
            *reinterpret_cast<float4*>(a.z + (long)r * a.ld_z + q * 4) = o;
        }
    }
    xch.finish();
}

__global__ void __launch_bounds__(1024) bn_small_fwd_kernel(SmallFwdArgs a) { bn_small_fwd_body(a, SmallNoExchange()); }

// Evaluation-mode norm layer straight from the split-K partials of the convolution in front (MNK_CONV_DEFER_SPLITK): sums the
// partials, adds the bias, applies the running-statistics affine + ReLU (+ 2x2 average pool) and writes z -- one launch instead of
// split reduction + apply, and y never exists.  The reference's per-frame evaluation loops (reconstruction.py:45-62, batch 1) run
// every convolution split along K, so this is one launch less per norm layer of a frame.  Arithmetic = what the two launches do,
// bit for bit: the splits in conv3x3_splitk_reduce_stats_kernel's order (four interleaved groups, (g0 + g1) + (g2 + g3), then
// the bias), then bn_act_fwd_kernel's fmaf / activation / pooling order.
struct EvalSplitArgs {
    const float* ws;       // [split][phase][M][ldw]
    int splits, ldw, phases;
    const float *bias, *mean, *scale, *beta;
    float* z;
    int ld_z, N, H, W, C;  // (H, W): the size of the convolution's output (the up-sampled size for the sub-pixel form)
    int relu, pool, tx_n, ty_n;
};

__device__ __forceinline__ float4 eval_split_pixel(const EvalSplitArgs& a, int n, int Y, int X, int q, float4 bv, int rem) {
    long pr, prows;
    if (a.phases > 1) {          // output pixel (n, Y, X) <- row (n, Y>>1, X>>1) of phase 2 (Y&1) + (X&1)
        const int Hl = a.H >> 1, Wl = a.W >> 1;
        const long Mlow = (long)a.N * Hl * Wl;
        prows = 4 * Mlow;
        pr = (long)((Y & 1) * 2 + (X & 1)) * Mlow + ((long)n * Hl + (Y >> 1)) * Wl + (X >> 1);
    } else {
        prows = (long)a.N * a.H * a.W;
        pr = ((long)n * a.H + Y) * a.W + X;
    }
    const float* p = a.ws + pr * a.ldw + q * 4;
    const long sstride = prows * a.ldw;
    float4 g[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) g[e] = make_float4(0.f, 0.f, 0.f, 0.f);
    int s = 0;
    for (; s + 4 <= a.splits; s += 4) {
#pragma unroll
        for (int e = 0; e < 4; ++e) g[e] = f4_add(g[e], *reinterpret_cast<const float4*>(p + (long)(s + e) * sstride));
    }
#pragma unroll
    for (int e = 0; e < 3; ++e)
        if (s + e < a.splits) g[e] = f4_add(g[e], *reinterpret_cast<const float4*>(p + (long)(s + e) * sstride));
    float4 r;
    r.x = (g[0].x + g[1].x) + (g[2].x + g[3].x);
    r.y = (g[0].y + g[1].y) + (g[2].y + g[3].y);
    r.z = (g[0].z + g[1].z) + (g[2].z + g[3].z);
    r.w = (g[0].w + g[1].w) + (g[2].w + g[3].w);
    if (a.bias) r = f4_add(r, bv);
    r.x = rem > 0 ? r.x : 0.f;   // columns beyond Cout hold the results of clamped weight rows
    r.y = rem > 1 ? r.y : 0.f;
    r.z = rem > 2 ? r.z : 0.f;
    r.w = rem > 3 ? r.w : 0.f;
    return r;
}

__global__ void __launch_bounds__(256) bn_eval_split_fwd_kernel(EvalSplitArgs a) {
    const int tx = threadIdx.x % a.tx_n, ty = threadIdx.x / a.tx_n;
    const int nv = a.ld_z / 4, q = blockIdx.x * a.tx_n + tx;
    if (q >= nv || ty >= a.ty_n) return;
    const int Ho = a.pool ? a.H / 2 : a.H, Wo = a.pool ? a.W / 2 : a.W;
    const int orows = a.N * Ho * Wo;
    const int p = blockIdx.y * a.ty_n + ty;
    if (p >= orows) return;
    const int rem = a.C - q * 4;
    const float4 bv = a.bias ? ld4_guard(a.bias, q, a.C) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 m = ld4_guard(a.mean, q, a.C), sc = ld4_guard(a.scale, q, a.C), be = ld4_guard(a.beta, q, a.C);
    const float slope = a.relu ? 0.f : -1.f;
    const int wo = p % Wo, t = p / Wo, ho = t % Ho, n = t / Ho;
    float4 o;
    if (a.pool) {
        float4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = eval_split_pixel(a, n, 2 * ho + (k >> 1), 2 * wo + (k & 1), q, bv, rem);
        o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            o.x += act_apply(fmaf(v[k].x - m.x, sc.x, be.x), slope);
            o.y += act_apply(fmaf(v[k].y - m.y, sc.y, be.y), slope);
            o.z += act_apply(fmaf(v[k].z - m.z, sc.z, be.z), slope);
            o.w += act_apply(fmaf(v[k].w - m.w, sc.w, be.w), slope);
        }
        o.x *= 0.25f;
        o.y *= 0.25f;
        o.z *= 0.25f;
        o.w *= 0.25f;
    } else {
        const float4 v = eval_split_pixel(a, n, ho, wo, q, bv, rem);
        o.x = act_apply(fmaf(v.x - m.x, sc.x, be.x), slope);
        o.y = act_apply(fmaf(v.y - m.y, sc.y, be.y), slope);
        o.z = act_apply(fmaf(v.z - m.z, sc.z, be.z), slope);
        o.w = act_apply(fmaf(v.w - m.w, sc.w, be.w), slope);
    }
    *reinterpret_cast<float4*>(a.z + (long)p * a.ld_z + q * 4) = o;      // guarded parameter loads make the pad lanes zero
}

__global__ void __launch_bounds__(1024) bn_small_bwd_kernel(BwdLoader L, double count, int rows, int nv, int tx_n,
                                                            float* __restrict__ sums, float* __restrict__ dy, int ld_dy) {
    __shared__ float4 red0[1024], red1[1024];
    const int ty_n = blockDim.x / tx_n;
    const int tx = threadIdx.x % tx_n, ty = threadIdx.x / tx_n;
    const int q = blockIdx.x * tx_n + tx;
    const bool qok = q < nv;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
    BwdLoader::State st;
    L.init(st);
    if (qok)
        for (int r = ty; r < rows; r += ty_n) {
            float4 g, xh;
            L.load(r, q, st, g, xh);
            a = f4_add(a, g);
            b = f4_fma(g, xh, b);
        }
    small_tree_sum2(red0, red1, a, b, tx_n, ty_n, tx, ty);
    if (!qok) return;
    const int C = L.C, rem = C - q * 4;
    if (ty == 0) {                         // dbeta = sum g, dgamma = sum g * xhat
        const float av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (q * 4 + e < C) {
                sums[q * 4 + e] = av[e];
                sums[C + q * 4 + e] = bv[e];
            }
    }
    const float inv = (float)(1.0 / count);
    const float4 k1 = make_float4(a.x * inv, a.y * inv, a.z * inv, a.w * inv), k2 = make_float4(b.x * inv, b.y * inv, b.z * inv, b.w * inv);
    const float4 sc = ld4_guard(L.scale, q, C);
    for (int r = ty; r < rows; r += ty_n) {
        float4 g, xh;
        L.load(r, q, st, g, xh);
        float4 o;
        o.x = sc.x * (g.x - k1.x - xh.x * k2.x);
        o.y = sc.y * (g.y - k1.y - xh.y * k2.y);
        o.z = sc.z * (g.z - k1.z - xh.z * k2.z);
        o.w = sc.w * (g.w - k1.w - xh.w * k2.w);
        if (rem < 4) {  // keep pad channels of dy at zero
            if (rem < 2) o.y = 0.f;
            if (rem < 3) o.z = 0.f;
            o.w = 0.f;
        }
        *reinterpret_cast<float4*>(dy + (long)r * ld_dy + q * 4) = o;
    }
}

// bn_small_bwd_kernel of one rank of a data-parallel run: ONE channel quad per block (256 threads over the rows); after the
// block's own sums (= this rank's dbeta / dgamma contributions, written to `sums`) wave 0 exchanges the eight of them with every
// rank of the node in one round trip (p2p_exchange_values) and the apply pass runs with the sums over all ranks -- statistics,
// exchange and apply of a small layer in one launch, as on a single GPU (count = rows of ALL ranks).
__global__ void __launch_bounds__(256) bn_small_bwd_sync_kernel(BwdLoader L, double count, int rows, int nv, float* __restrict__ sums,
                                                                float* __restrict__ dy, int ld_dy, PeerTable peers, int rank,
                                                                int world, unsigned* state, unsigned long long timeout_ticks) {
    __shared__ float4 red0[256], red1[256];
    __shared__ float glob[8];
    const int ty = threadIdx.x, q = blockIdx.x;          // (tx_n = 1: q < nv for every block)
    const unsigned seq = state[0] + 1;
    const int slot = (int)(seq % P2P_SLOTS);
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
    BwdLoader::State st;
    L.init(st);
    for (int r = ty; r < rows; r += 256) {
        float4 g, xh;
        L.load(r, q, st, g, xh);
        a = f4_add(a, g);
        b = f4_fma(g, xh, b);
    }
    small_tree_sum2(red0, red1, a, b, 1, 256, 0, ty);
    const int C = L.C, rem = C - q * 4;
    const float av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w};
    if (ty == 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (q * 4 + e < C) {
                sums[q * 4 + e] = av[e];
                sums[C + q * 4 + e] = bv[e];
            }
    }
    if (ty < 64) {                                         // wave 0: value j = lane / W of {a.xyzw, b.xyzw}
        const int W = world <= 8 ? 8 : 16, per = 64 / W;
        for (int j0 = 0; j0 < 8; j0 += per) {
            const int j = j0 + ty / W, e = j & 3, c = q * 4 + e;
            float v = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (e == k) v = j < 4 ? av[k] : bv[k];
            const int index = (j < 8 && c < C) ? (j < 4 ? c : C + c) : -1;
            const float all = p2p_exchange_values(peers, rank, world, slot, seq, index, v, state, timeout_ticks);
            if ((ty & (W - 1)) == 0 && j < 8) glob[j] = all;
        }
    }
    __syncthreads();
    const float inv = (float)(1.0 / count);
    const float4 k1 = make_float4(glob[0] * inv, glob[1] * inv, glob[2] * inv, glob[3] * inv),
                 k2 = make_float4(glob[4] * inv, glob[5] * inv, glob[6] * inv, glob[7] * inv);
    const float4 sc = ld4_guard(L.scale, q, C);
    for (int r = ty; r < rows; r += 256) {
        float4 g, xh;
        L.load(r, q, st, g, xh);
        float4 o;
        o.x = sc.x * (g.x - k1.x - xh.x * k2.x);
        o.y = sc.y * (g.y - k1.y - xh.y * k2.y);
        o.z = sc.z * (g.z - k1.z - xh.z * k2.z);
        o.w = sc.w * (g.w - k1.w - xh.w * k2.w);
        if (rem < 4) {  // keep pad channels of dy at zero
            if (rem < 2) o.y = 0.f;
            if (rem < 3) o.z = 0.f;
            o.w = 0.f;
        }
        *reinterpret_cast<float4*>(dy + (long)r * ld_dy + q * 4) = o;
    }
    p2p_finish_launch(state, seq);
}

// bn_small_fwd_kernel of one rank of a data-parallel run (the forward twin of the kernel above, round 5): ONE channel quad per
// block; after the block's tree sum wave 0 exchanges the quad's eight sums (sum, sum of squares of four channels) with every rank
// of the node in one round trip, and the statistics, the running statistics and the apply pass are made from the sums over ALL
// ranks (count = rows * world) -- split-K reduction, statistics, exchange, finalisation and apply of a small layer in one
// launch, as on a single GPU.  Replaces sync_batchnorm/batchnorm.py:55-78 (+ the reduce / broadcast of :95-111) for these layers.
struct SmallP2PExchange {
    PeerTable peers;
    int rank, world;
    unsigned* state;
    unsigned long long timeout_ticks;
    unsigned seq;
    __device__ __forceinline__ int world_size() const { return world; }
    __device__ __forceinline__ void all_ranks(float4& s1, float4& s2, int q, int C) const {
        __shared__ float glob[8];
        const int t = threadIdx.x, slot = (int)(seq % P2P_SLOTS);
        const float av[4] = {s1.x, s1.y, s1.z, s1.w}, bv[4] = {s2.x, s2.y, s2.z, s2.w};
        if (t < 64) {                                          // wave 0: value j = lane / W of {s1.xyzw, s2.xyzw}
            const int W = world <= 8 ? 8 : 16, per = 64 / W;
            for (int j0 = 0; j0 < 8; j0 += per) {
                const int j = j0 + t / W, e = j & 3, c = q * 4 + e;
                float v = 0.f;
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (e == k) v = j < 4 ? av[k] : bv[k];
                const int index = (j < 8 && c < C) ? (j < 4 ? c : C + c) : -1;
                const float all = p2p_exchange_values(peers, rank, world, slot, seq, index, v, state, timeout_ticks);
                if ((t & (W - 1)) == 0 && j < 8) glob[j] = all;
            }
        }
        __syncthreads();
        s1 = make_float4(glob[0], glob[1], glob[2], glob[3]);
        s2 = make_float4(glob[4], glob[5], glob[6], glob[7]);
    }
    __device__ __forceinline__ void finish() const { p2p_finish_launch(state, seq); }
};

__global__ void __launch_bounds__(256) bn_small_fwd_sync_kernel(SmallFwdArgs a, PeerTable peers, int rank, int world, unsigned* state,
                                                                unsigned long long timeout_ticks) {
    SmallP2PExchange x{peers, rank, world, state, timeout_ticks, state[0] + 1};
    bn_small_fwd_body(a, x);
}

static inline int small_txn(int nv) { return g_small_txn ? g_small_txn : (nv >= 128 ? 4 : (nv >= 64 ? 2 : 1)); }

// launch shape of the two one-launch kernels: threads per block and channel quads per block (see g_small_fwd_threads)
static inline void small_shape(int nv, int* threads, int* tx_n) {
    *tx_n = small_txn(nv);
    if (g_small_fwd_txn > 0)
        *tx_n = g_small_fwd_txn;
    else if (g_small_fwd_txn == 0)
        for (*tx_n = 8; *tx_n > 1 && nv / *tx_n < 32; *tx_n >>= 1) {}
    *threads = (g_small_fwd_threads == 512 || g_small_fwd_threads == 1024) ? g_small_fwd_threads : 256;
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
                       frames, sums, 2);
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
    const long rows = (long)N * (pool ? H / 2 : H) * (pool ? W / 2 : W);
    MNK_REQUIRE((long)N * H * W < (1L << 31));
    ProfScope prof(K_BN_APPLY, s, (double)N * H * W * C * 4 * (pool ? 1.25 : 2.0));
    const int pstride = per_frame ? C : 0;
    Map2D m = make_map(rows, round_up(C, 4), 2, 2048);     // no reduction here: small layers want blocks, not rows per thread
    const dim3 grid(m.col_tiles, m.row_blocks);
    if (pool)
        hipLaunchKernelGGL(bn_act_fwd_kernel<1>, grid, dim3(256), 0, s, y, ld_y, mean, scale, beta, pstride, z, ld_z, z_off,
                           N, H, W, C, slope, m.tx, m.ty, m.rows_per_block, FwdFinalize());
    else
        hipLaunchKernelGGL(bn_act_fwd_kernel<0>, grid, dim3(256), 0, s, y, ld_y, mean, scale, beta, pstride, z, ld_z, z_off,
                           N, H, W, C, slope, m.tx, m.ty, m.rows_per_block, FwdFinalize());
    MNK_LAUNCH_CHECK();
    return MNK_OK;
}

int mnk_norm_act_bwd_stats(const float* y, int ld_y, const float* dz, int ld_dz, int dz_off, const float* mean,
                           const float* invstd, const float* scale, const float* beta, int per_frame, int N, int H,
                           int W, int C, float slope, int pool, float* sums, float* ws, size_t ws_floats,
                           void* stream) {
    MNK_REQUIRE(y && dz && mean && invstd && scale && beta && sums && ws && N > 0 && H > 0 && W > 0 && C > 0);
    MNK_REQUIRE(ld_y % 4 == 0 && ld_y >= round_up(C, 4) && dz_off >= 0 && dz_off + C <= ld_dz);
    MNK_REQUIRE((!pool || (H >= 2 && W >= 2)) && (long)N * H * W < (1L << 31));
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
                       frames, sums, 2);
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
    MNK_REQUIRE((long)N * H * W < (1L << 31));
    const long rows = (long)N * H * W;
    const int ldc = round_up(C, 4);
    Map2D m = make_map(rows, ldc);
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(K_BN_BWD, s, (double)rows * C * 4 * (pool ? 2.25 : 3.0));
    BwdLoader L{y, dz, mean, invstd, scale, beta, ld_y, ld_dz, dz_off, H, W, C, pool, per_frame ? C : 0, slope};
    hipLaunchKernelGGL(bn_act_bwd_apply_kernel, dim3(m.col_tiles, m.row_blocks), dim3(256), 0, s, L, sums, count, training,
                       (per_frame ? N : 1) * C, dy, ld_dy, rows, C, ldc / 4, m.tx, m.ty, m.rows_per_block, (float*)nullptr,
                       (const float*)nullptr, 0);
    MNK_LAUNCH_CHECK();
    return MNK_OK;
}

int mnk_bn_act_bwd_apply_add_colsum(const float* y, int ld_y, const float* dz, int ld_dz, int dz_off, const float* mean,
                                    const float* invstd, const float* scale, const float* beta, const float* sums,
                                    double count, int training, const float* addend, int ld_add, float* dy, int ld_dy, int N,
                                    int H, int W, int C, int relu, int pool, float* dy_sums, float* ws, size_t ws_floats,
                                    void* stream) {
    MNK_REQUIRE(y && dz && mean && invstd && scale && beta && dy && dy_sums && ws && N > 0 && H > 0 && W > 0 && C > 0);
    MNK_REQUIRE(!addend || (ld_add % 4 == 0 && ld_add >= round_up(C, 4) && (size_t)addend % 16 == 0));
    MNK_REQUIRE(!training || (sums && count > 0));
    MNK_REQUIRE(ld_y % 4 == 0 && ld_y >= round_up(C, 4) && ld_dy % 4 == 0 && ld_dy >= round_up(C, 4));
    MNK_REQUIRE(dz_off >= 0 && dz_off + C <= ld_dz && (!pool || (H >= 2 && W >= 2)));
    MNK_REQUIRE((long)N * H * W < (1L << 31));
    const long rows = (long)N * H * W;
    const int ldc = round_up(C, 4);
    Map2D m = make_map(rows, ldc);
    if (ws_floats < (size_t)m.row_blocks * 2 * ldc) {
        set_error("mnk_bn_act_bwd_apply_colsum: workspace too small");
        return MNK_EWORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(K_BN_BWD, s, (double)rows * C * 4 * (pool ? 2.25 : 3.0));
    BwdLoader L{y, dz, mean, invstd, scale, beta, ld_y, ld_dz, dz_off, H, W, C, pool, 0, relu ? 0.f : -1.f};
    hipLaunchKernelGGL(bn_act_bwd_apply_kernel, dim3(m.col_tiles, m.row_blocks), dim3(256), 0, s, L, sums, count, training,
                       C, dy, ld_dy, rows, C, ldc / 4, m.tx, m.ty, m.rows_per_block, ws, addend, ld_add);
    hipLaunchKernelGGL(colsum2_final_kernel, dim3(ceil_div(C, 4)), dim3(256), 0, s, ws, m.row_blocks, ldc, C, 1, dy_sums, 1);
    MNK_LAUNCH_CHECK();
    return MNK_OK;
}

int mnk_bn_act_bwd_apply_colsum(const float* y, int ld_y, const float* dz, int ld_dz, int dz_off, const float* mean,
                                const float* invstd, const float* scale, const float* beta, const float* sums,
                                double count, int training, float* dy, int ld_dy, int N, int H, int W, int C, int relu,
                                int pool, float* dy_sums, float* ws, size_t ws_floats, void* stream) {
    return mnk_bn_act_bwd_apply_add_colsum(y, ld_y, dz, ld_dz, dz_off, mean, invstd, scale, beta, sums, count, training, nullptr,
                                           0, dy, ld_dy, N, H, W, C, relu, pool, dy_sums, ws, ws_floats, stream);
}

int mnk_bn_stats_finalize(const float* x, int ld, long rows, int C, const float* pre_partial, int pre_row_blocks,
                          double count, const float* gamma, float* running_mean, float* running_var, float momentum,
                          float eps, int update_running, float* sums, float* mean, float* invstd, float* scale, float* ws,
                          size_t ws_floats, void* stream) {
    MNK_REQUIRE(gamma && mean && invstd && scale && C > 0 && count > 0 && ld % 4 == 0 && ld >= C);
    MNK_REQUIRE(!update_running || (running_mean && running_var && count > 1));
    MNK_REQUIRE(pre_partial ? pre_row_blocks > 0 : (x && rows > 0 && ws));
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(K_BN_STATS, s, pre_partial ? (double)pre_row_blocks * 2 * C * 4 : (double)rows * C * 4);
    const float* partial = pre_partial;
    int row_blocks = pre_row_blocks;
    if (!pre_partial) {
        Map2D m = make_map(rows, ld);
        if (ws_floats < (size_t)m.row_blocks * 2 * ld) {
            set_error("mnk_bn_stats_finalize: workspace too small");
            return MNK_EWORKSPACE;
        }
        StatsLoader L{x, ld};
        hipLaunchKernelGGL(colsum2_partial_kernel<StatsLoader>, dim3(m.col_tiles, m.row_blocks, 1), dim3(256), 0, s, L, rows,
                           ld / 4, ld, m.tx, m.ty, m.rows_per_block, ws);
        partial = ws;
        row_blocks = m.row_blocks;
    }
    hipLaunchKernelGGL(bn_final_finalize_kernel, dim3(ceil_div(C, 4)), dim3(256), 0, s, partial, row_blocks, ld, C, count, gamma,
                       running_mean, running_var, momentum, eps, update_running, sums, mean, invstd, scale);
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
    hipLaunchKernelGGL(colsum2_final_kernel, dim3(ceil_div(2 * C, 4)), dim3(256), 0, s, partial, row_blocks, ld, C, 1, sums, 2);
    MNK_LAUNCH_CHECK();
    return MNK_OK;
}

// ---- the same statistics with the SyncBN exchange of one node inside their second stage (colsum2_final_sync_kernel) -----------
int mnk_bn_stats_finish_sync(void* p2p, const float* partial, int row_blocks, int ld, int C, float* sums_local, float* sums_global,
                             int timeout_ms, void* stream) {
    MNK_REQUIRE(p2p && partial && sums_global && row_blocks > 0 && C > 0 && ld >= C);
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(K_BN_STATS, s, (double)row_blocks * 2 * C * 4);
    const int rc = launch_final_sync(p2p, partial, row_blocks, ld, C, sums_local, sums_global, timeout_ms, s);
    if (rc != MNK_OK) return rc;
    MNK_LAUNCH_CHECK();
    return MNK_OK;
}
int mnk_bn_stats_sync(void* p2p, const float* x, int ld, long rows, int C, float* sums_local, float* sums_global, float* ws,
                      size_t ws_floats, int timeout_ms, void* stream) {
    MNK_REQUIRE(p2p && x && sums_global && ws && rows > 0 && C > 0 && ld % 4 == 0 && ld >= C);
    Map2D m = make_map(rows, ld);
    if (ws_floats < (size_t)m.row_blocks * 2 * ld) {
        set_error("mnk_bn_stats_sync: workspace too small");
        return MNK_EWORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(K_BN_STATS, s, (double)rows * C * 4);
    StatsLoader L{x, ld};
    hipLaunchKernelGGL(colsum2_partial_kernel<StatsLoader>, dim3(m.col_tiles, m.row_blocks, 1), dim3(256), 0, s, L, rows, ld / 4, ld,
                       m.tx, m.ty, m.rows_per_block, ws);
    const int rc = launch_final_sync(p2p, ws, m.row_blocks, ld, C, sums_local, sums_global, timeout_ms, s);
    if (rc != MNK_OK) return rc;
    MNK_LAUNCH_CHECK();
    return MNK_OK;
}
int mnk_bn_act_bwd_stats_sync(void* p2p, const float* y, int ld_y, const float* dz, int ld_dz, int dz_off, const float* mean,
                              const float* invstd, const float* scale, const float* beta, int N, int H, int W, int C, int relu,
                              int pool, float* sums_local, float* sums_global, float* ws, size_t ws_floats, int timeout_ms,
                              void* stream) {
    MNK_REQUIRE(p2p && y && dz && mean && invstd && scale && beta && sums_global && ws && N > 0 && H > 0 && W > 0 && C > 0);
    MNK_REQUIRE(ld_y % 4 == 0 && ld_y >= round_up(C, 4) && dz_off >= 0 && dz_off + C <= ld_dz);
    MNK_REQUIRE((!pool || (H >= 2 && W >= 2)) && (long)N * H * W < (1L << 31));
    const long rows = (long)N * H * W;
    const int ldc = round_up(C, 4);
    Map2D m = make_map(rows, ldc);
    if (ws_floats < (size_t)m.row_blocks * 2 * ldc) {
        set_error("mnk_bn_act_bwd_stats_sync: workspace too small");
        return MNK_EWORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(K_BN_BWD, s, (double)N * H * W * C * 4 * (pool ? 1.25 : 2.0));
    BwdLoader L{y, dz, mean, invstd, scale, beta, ld_y, ld_dz, dz_off, H, W, C, pool, 0, relu ? 0.f : -1.f};
    hipLaunchKernelGGL(colsum2_partial_kernel<BwdLoader>, dim3(m.col_tiles, m.row_blocks, 1), dim3(256), 0, s, L, rows, ldc / 4, ldc,
                       m.tx, m.ty, m.rows_per_block, ws);
    const int rc = launch_final_sync(p2p, ws, m.row_blocks, ldc, C, sums_local, sums_global, timeout_ms, s);
    if (rc != MNK_OK) return rc;
    MNK_LAUNCH_CHECK();
    return MNK_OK;
}

int mnk_bn_finalize(const float* sums, double count, const float* gamma, float* running_mean, float* running_var,
                    float momentum, float eps, int C, int update_running, float* mean, float* invstd, float* scale,
                    void* stream) {
    return mnk_norm_finalize(sums, count, gamma, running_mean, running_var, momentum, eps, C, 1, update_running, mean,
                             invstd, scale, stream);
}

int mnk_bn_act_fwd_sums(const float* y, int ld_y, const float* sums, double count, const float* gamma, const float* beta,
                        float* running_mean, float* running_var, float momentum, float eps, int update_running, float* mean,
                        float* invstd, float* scale, float* z, int ld_z, int z_off, int N, int H, int W, int C, int relu, int pool,
                        void* stream) {
    MNK_REQUIRE(y && sums && gamma && beta && mean && invstd && scale && z && N > 0 && H > 0 && W > 0 && C > 0 && count > 0);
    MNK_REQUIRE(!update_running || (running_mean && running_var && count > 1));
    MNK_REQUIRE(ld_y % 4 == 0 && ld_y >= round_up(C, 4) && z_off >= 0 && z_off + C <= ld_z);
    MNK_REQUIRE(!pool || (H % 2 == 0 && W % 2 == 0));
    MNK_REQUIRE((long)N * H * W < (1L << 31));
    hipStream_t s = (hipStream_t)stream;
    const long rows = (long)N * (pool ? H / 2 : H) * (pool ? W / 2 : W);
    ProfScope prof(K_BN_APPLY, s, (double)N * H * W * C * 4 * (pool ? 1.25 : 2.0));
    Map2D m = make_map(rows, round_up(C, 4), 2, 2048);
    const dim3 grid(m.col_tiles, m.row_blocks);
    const FwdFinalize fin{sums, gamma, running_mean, running_var, mean, invstd, scale, count, momentum, eps, update_running};
    const float slope = relu ? 0.f : -1.f;
    if (pool)
        hipLaunchKernelGGL((bn_act_fwd_kernel<1, true>), grid, dim3(256), 0, s, y, ld_y, (const float*)nullptr,
                           (const float*)nullptr, beta, 0, z, ld_z, z_off, N, H, W, C, slope, m.tx, m.ty, m.rows_per_block, fin);
    else
        hipLaunchKernelGGL((bn_act_fwd_kernel<0, true>), grid, dim3(256), 0, s, y, ld_y, (const float*)nullptr,
                           (const float*)nullptr, beta, 0, z, ld_z, z_off, N, H, W, C, slope, m.tx, m.ty, m.rows_per_block, fin);
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

// ---- small layers: one launch per direction (single-rank training-mode BatchNorm) ------------------------------------------
int mnk_bn_small_rows(void) { return g_small_rows; }

int mnk_bn_small_fwd(const float* ws, int splits, int ldw, int phases, const float* bias, float* y, int ld_y, int N, int H, int W,
                     int C, const float* gamma, const float* beta, float* running_mean, float* running_var, float momentum,
                     float eps, float* mean, float* invstd, float* scale, float* z, int ld_z, int relu, int pool, void* stream) {
    MNK_REQUIRE(y && gamma && beta && running_mean && running_var && mean && invstd && scale && z && N > 0 && H > 0 && W > 0);
    MNK_REQUIRE(C > 0 && ld_y % 4 == 0 && ld_y == round_up(C, 4) && ld_z == ld_y && (long)N * H * W <= 4096 && (long)N * H * W > 1);
    MNK_REQUIRE(!ws || (splits >= 1 && ldw == ld_y && (phases == 1 || (phases == 4 && H % 2 == 0 && W % 2 == 0))));
    MNK_REQUIRE(!pool || (H % 2 == 0 && W % 2 == 0));
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(K_BN_APPLY, s, (double)N * H * W * C * 4 * 3.0);
    SmallFwdArgs a{ws, splits, ldw, phases, bias, y, ld_y, N, H, W, C, gamma, beta, running_mean, running_var, momentum, eps,
                   mean, invstd, scale, z, ld_z, relu, pool, small_txn(ld_y / 4)};
    int threads;
    small_shape(ld_y / 4, &threads, &a.tx_n);
    MNK_REQUIRE(a.tx_n >= 1 && a.tx_n <= 64 && (a.tx_n & (a.tx_n - 1)) == 0);
    hipLaunchKernelGGL(bn_small_fwd_kernel, dim3(ceil_div(ld_y / 4, a.tx_n)), dim3(threads), 0, s, a);
    MNK_LAUNCH_CHECK();
    return MNK_OK;
}

int mnk_bn_eval_split_fwd(const float* ws, int splits, int ldw, int phases, const float* bias, const float* mean, const float* scale,
                          const float* beta, float* z, int ld_z, int N, int H, int W, int C, int relu, int pool, void* stream) {
    MNK_REQUIRE(ws && mean && scale && beta && z && splits >= 1 && (phases == 1 || phases == 4) && N > 0 && H > 0 && W > 0 && C > 0);
    MNK_REQUIRE(ldw % 4 == 0 && ldw >= C && ld_z == ldw && (size_t)ws % 16 == 0 && (size_t)z % 16 == 0);
    MNK_REQUIRE((phases == 1 || (H % 2 == 0 && W % 2 == 0)) && (!pool || (H % 2 == 0 && W % 2 == 0)));
    MNK_REQUIRE((long)N * H * W * phases < (1L << 31));
    hipStream_t s = (hipStream_t)stream;
    EvalSplitArgs a;
    a.ws = ws, a.splits = splits, a.ldw = ldw, a.phases = phases;
    a.bias = bias, a.mean = mean, a.scale = scale, a.beta = beta;
    a.z = z, a.ld_z = ld_z, a.N = N, a.H = H, a.W = W, a.C = C, a.relu = relu, a.pool = pool;
    const int nv = ld_z / 4;
    int tx = 1;
    while (tx < nv && tx < 64) tx <<= 1;
    if (nv <= 64) tx = nv;               // exact quad lanes (see make_map)
    a.tx_n = tx;
    a.ty_n = 256 / tx;
    const long orows = (long)N * (pool ? H / 2 : H) * (pool ? W / 2 : W);
    ProfScope prof(K_BN_APPLY, s, ((double)splits * N * H * W * ldw + (double)orows * ld_z) * 4);
    hipLaunchKernelGGL(bn_eval_split_fwd_kernel, dim3(ceil_div(nv, tx), ceil_div(orows, a.ty_n)), dim3(256), 0, s, a);
    MNK_LAUNCH_CHECK();
    return MNK_OK;
}

int mnk_bn_small_fwd_sync(void* p2p, const float* ws, int splits, int ldw, int phases, const float* bias, float* y, int ld_y, int N,
                          int H, int W, int C, const float* gamma, const float* beta, float* running_mean, float* running_var,
                          float momentum, float eps, float* mean, float* invstd, float* scale, float* z, int ld_z, int relu, int pool,
                          int timeout_ms, void* stream) {
    MNK_REQUIRE(p2p && y && gamma && beta && running_mean && running_var && mean && invstd && scale && z && N > 0 && H > 0 && W > 0);
    MNK_REQUIRE(C > 0 && ld_y % 4 == 0 && ld_y == round_up(C, 4) && ld_z == ld_y && (long)N * H * W <= 4096 && timeout_ms > 0);
    MNK_REQUIRE(!ws || (splits >= 1 && ldw == ld_y && (phases == 1 || (phases == 4 && H % 2 == 0 && W % 2 == 0))));
    MNK_REQUIRE(!pool || (H % 2 == 0 && W % 2 == 0));
    PeerTable peers;
    int rank = 0, world = 0;
    unsigned* state = nullptr;
    if (!p2p_launch_info(p2p, &peers, &rank, &world, &state) || 2 * C > P2P_MAXF) {
        set_error("mnk_bn_small_fwd_sync: the peer-to-peer exchange is not connected, or more than %d channels", P2P_MAXF / 2);
        return MNK_ECOMM;
    }
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(K_BN_APPLY, s, (double)N * H * W * C * 4 * 3.0);
    SmallFwdArgs a{ws, splits, ldw, phases, bias, y, ld_y, N, H, W, C, gamma, beta, running_mean, running_var, momentum, eps,
                   mean, invstd, scale, z, ld_z, relu, pool, 1};
    hipLaunchKernelGGL(bn_small_fwd_sync_kernel, dim3(ld_y / 4), dim3(256), 0, s, a, peers, rank, world, state,
                       (unsigned long long)timeout_ms * 100000ull);
    MNK_LAUNCH_CHECK();
    return MNK_OK;
}

int mnk_bn_small_bwd(const float* y, int ld_y, const float* dz, int ld_dz, const float* mean, const float* invstd,
                     const float* scale, const float* beta, double count, int N, int H, int W, int C, int relu, int pool,
                     float* sums, float* dy, int ld_dy, void* stream) {
    MNK_REQUIRE(y && dz && mean && invstd && scale && beta && sums && dy && N > 0 && H > 0 && W > 0 && C > 0 && count > 1);
    MNK_REQUIRE(ld_y % 4 == 0 && ld_y >= round_up(C, 4) && ld_dy % 4 == 0 && ld_dy >= round_up(C, 4) && ld_dz >= C);
    MNK_REQUIRE((long)N * H * W <= 4096 && (!pool || (H % 2 == 0 && W % 2 == 0)));
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(K_BN_BWD, s, (double)N * H * W * C * 4 * 5.0);
    const int nv = round_up(C, 4) / 4;
    int threads = 256, txn = small_txn(nv);
    if (g_small_bwd_shape) small_shape(nv, &threads, &txn);
    MNK_REQUIRE(txn >= 1 && txn <= 64 && (txn & (txn - 1)) == 0);
    BwdLoader L{y, dz, mean, invstd, scale, beta, ld_y, ld_dz, 0, H, W, C, pool, 0, relu ? 0.f : -1.f};
    hipLaunchKernelGGL(bn_small_bwd_kernel, dim3(ceil_div(nv, txn)), dim3(threads), 0, s, L, count, N * H * W, nv, txn, sums, dy, ld_dy);
    MNK_LAUNCH_CHECK();
    return MNK_OK;
}

int mnk_bn_small_bwd_sync(void* p2p, const float* y, int ld_y, const float* dz, int ld_dz, const float* mean, const float* invstd,
                          const float* scale, const float* beta, double count_all_ranks, int N, int H, int W, int C, int relu,
                          int pool, float* sums_local, float* dy, int ld_dy, int timeout_ms, void* stream) {
    MNK_REQUIRE(p2p && y && dz && mean && invstd && scale && beta && sums_local && dy && N > 0 && H > 0 && W > 0 && C > 0);
    MNK_REQUIRE(count_all_ranks > 1 && timeout_ms > 0);
    MNK_REQUIRE(ld_y % 4 == 0 && ld_y >= round_up(C, 4) && ld_dy % 4 == 0 && ld_dy >= round_up(C, 4) && ld_dz >= C);
    MNK_REQUIRE((long)N * H * W <= 4096 && (!pool || (H % 2 == 0 && W % 2 == 0)));
    PeerTable peers;
    int rank = 0, world = 0;
    unsigned* state = nullptr;
    if (!p2p_launch_info(p2p, &peers, &rank, &world, &state) || 2 * C > P2P_MAXF) {
        set_error("mnk_bn_small_bwd_sync: the peer-to-peer exchange is not connected, or more than %d channels", P2P_MAXF / 2);
        return MNK_ECOMM;
    }
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(K_BN_BWD, s, (double)N * H * W * C * 4 * 5.0);
    const int nv = round_up(C, 4) / 4;
    BwdLoader L{y, dz, mean, invstd, scale, beta, ld_y, ld_dz, 0, H, W, C, pool, 0, relu ? 0.f : -1.f};
    hipLaunchKernelGGL(bn_small_bwd_sync_kernel, dim3(nv), dim3(256), 0, s, L, count_all_ranks, N * H * W, nv, sums_local, dy, ld_dy,
                       peers, rank, world, state, (unsigned long long)timeout_ms * 100000ull);
    MNK_LAUNCH_CHECK();
    return MNK_OK;
}
}

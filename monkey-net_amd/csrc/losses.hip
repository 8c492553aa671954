// Feature-matching L1 of the training step on the discriminator's NHWC activations (modules/losses.py:8-12
// reconstruction_loss over discriminator maps, called from train.py:47-51): per-sample mean of |generated - real| over
// channels and pixels, straight from the act of the batched discriminator pass [generated | real] -- no NCDHW copies
// of the feature maps, one launch forward and one backward per map.  HBM-bound (reads the act once).
#include "mnk_common.h"

using namespace mnk;

namespace {

// one block per sample: threads stride over the float4 quads of frame i and of frame B + i
// one block per sample (its sum is one output): 1024 threads -- a batch is only 32 blocks (256 threads: 12 us per map)
constexpr int L1_THREADS = 1024;
__global__ void __launch_bounds__(L1_THREADS) pair_l1_fwd_kernel(const float* __restrict__ a, int ld, long rows, int C, int B,
                                                                 float scale, float* __restrict__ out) {
    __shared__ float red[L1_THREADS / 64];
    const int i = blockIdx.x;
    const int nv = ld / 4;
    const long quads = rows * nv;
    const float* pa = a + (long)i * rows * ld;
    const float* pb = a + (long)(B + i) * rows * ld;
    float acc = 0.f;
    for (long q = threadIdx.x; q < quads; q += L1_THREADS) {
        const int c = (int)(q % nv) * 4;
        const float4 u = *reinterpret_cast<const float4*>(pa + q * 4);
        const float4 v = *reinterpret_cast<const float4*>(pb + q * 4);
        float s = 0.f;
        if (c < C) s += fabsf(u.x - v.x);
        if (c + 1 < C) s += fabsf(u.y - v.y);
        if (c + 2 < C) s += fabsf(u.z - v.z);
        if (c + 3 < C) s += fabsf(u.w - v.w);
        acc += s;
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float tot = 0.f;
        for (int w = 0; w < L1_THREADS / 64; ++w) tot += red[w];      // fixed order: deterministic
        out[i] = tot * scale;
    }
}

__device__ __forceinline__ float sgn(float d) { return d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f); }

// da[i] = g[i] * scale * sign(a[i] - a[B + i]) on channels < C (0 on pad channels), da[B + i] = -da[i]
__global__ void __launch_bounds__(256) pair_l1_bwd_kernel(const float* __restrict__ a, int ld, long rows, int C, int B,
                                                          float scale, const float* __restrict__ g,
                                                          const float* __restrict__ addend, float* __restrict__ da) {
    const int nv = ld / 4;
    const long per = rows * nv, total = per * B;
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < total; q += (long)gridDim.x * 256) {
        const int i = (int)(q / per);
        const long r = q - (long)i * per;
        const int c = (int)(r % nv) * 4;
        const long oa = ((long)i * per + r) * 4, ob = ((long)(B + i) * per + r) * 4;
        const float4 u = *reinterpret_cast<const float4*>(a + oa);
        const float4 v = *reinterpret_cast<const float4*>(a + ob);
        const float k = g[i] * scale;
        float4 d;
        d.x = c < C ? k * sgn(u.x - v.x) : 0.f;
        d.y = c + 1 < C ? k * sgn(u.y - v.y) : 0.f;
        d.z = c + 2 < C ? k * sgn(u.z - v.z) : 0.f;
        d.w = c + 3 < C ? k * sgn(u.w - v.w) : 0.f;
        float4 e = make_float4(-d.x, -d.y, -d.z, -d.w);
        if (addend) {            // the gradient of the map's other consumer (the next block), added here instead of by autograd
            const float4 p = *reinterpret_cast<const float4*>(addend + oa), q2 = *reinterpret_cast<const float4*>(addend + ob);
            d = make_float4(d.x + p.x, d.y + p.y, d.z + p.z, d.w + p.w);
            e = make_float4(e.x + q2.x, e.y + q2.y, e.z + q2.z, e.w + q2.w);
        }
        *reinterpret_cast<float4*>(da + oa) = d;
        *reinterpret_cast<float4*>(da + ob) = e;
    }
}

// ---- image-level L1 term and the LSGAN terms (modules/losses.py:8-21; train.py:36-53,66-75) ------------------------
// out[i] = scale * sum_j |a[i][j] - b[i][j]| over n contiguous floats per sample (scale = weight / n)
__global__ void __launch_bounds__(L1_THREADS) l1_mean_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                                 long n, float scale, float* __restrict__ out) {
    __shared__ float red[L1_THREADS / 64];
    const float* pa = a + (long)blockIdx.x * n;
    const float* pb = b + (long)blockIdx.x * n;
    float acc = 0.f;
    for (long j = threadIdx.x; j < n; j += L1_THREADS) acc += fabsf(pa[j] - pb[j]);
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float tot = 0.f;
        for (int w = 0; w < L1_THREADS / 64; ++w) tot += red[w];
        out[blockIdx.x] = tot * scale;
    }
}

// da[i][j] = g[i] * scale * sign(a - b), db = -da (either may be NULL)
__global__ void __launch_bounds__(256) l1_mean_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b, long n,
                                                          long total, float scale, const float* __restrict__ g,
                                                          float* __restrict__ da, float* __restrict__ db) {
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < total; q += (long)gridDim.x * 256) {
        const float k = g[q / n] * scale * sgn(a[q] - b[q]);
        if (da) da[q] = k;
        if (db) db[q] = -k;
    }
}

// score[2B][n]: rows [0, B) of the generated frames, [B, 2B) of the real ones (the batched discriminator pass)
//   gen[i]  = wg * mean_j (1 - sf[i][j])^2                       (generator_gan_loss)
//   disc[i] = wd * mean_j ((1 - sr[i][j])^2 + sf[i][j]^2)         (discriminator_gan_loss)
__global__ void __launch_bounds__(64) gan_terms_fwd_kernel(const float* __restrict__ score, int n, int B, float wg, float wd,
                                                           float* __restrict__ gen, float* __restrict__ disc) {
    const int i = blockIdx.x;
    const float* sf = score + (long)i * n;
    const float* sr = score + (long)(B + i) * n;
    float ag = 0.f, ad = 0.f;
    for (int j = threadIdx.x; j < n; j += 64) {
        const float f = sf[j], r = sr[j];
        ag += (1.f - f) * (1.f - f);
        ad += (1.f - r) * (1.f - r) + f * f;
    }
    ag = wave_sum(ag);
    ad = wave_sum(ad);
    if (threadIdx.x == 0) {
        gen[i] = wg * (ag / (float)n);
        disc[i] = wd * (ad / (float)n);
    }
}

// dscore from the upstream gradients of both vectors (either may be NULL = zero)
__global__ void __launch_bounds__(256) gan_terms_bwd_kernel(const float* __restrict__ score, int n, int B, float wg, float wd,
                                                            const float* __restrict__ ggen, const float* __restrict__ gdisc,
                                                            float* __restrict__ dscore) {
    const long half = (long)B * n;
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < 2 * half; q += (long)gridDim.x * 256) {
        const bool real = q >= half;
        const long i = (real ? q - half : q) / n;
        const float v = score[q];
        const float gg = ggen ? ggen[i] * wg / (float)n : 0.f, gd = gdisc ? gdisc[i] * wd / (float)n : 0.f;
        dscore[q] = real ? gd * (-2.f) * (1.f - v) : gg * (-2.f) * (1.f - v) + gd * 2.f * v;
    }
}

// ---- batch means of the loss vectors (train.py:114 `[val.mean() for val in losses]` and their sum) -----------------------
constexpr int MAX_LOSS_VECS = 16;
struct LossVecs {
    const float* v[MAX_LOSS_VECS];
};

// one wavefront per vector: means[i] = mean_j v[i][j]; thread 0 then adds the means in order -> means[nvec]
__global__ void __launch_bounds__(64 * MAX_LOSS_VECS) vec_means_fwd_kernel(LossVecs vecs, int nvec, int len,
                                                                          float* __restrict__ means) {
    __shared__ float sm[MAX_LOSS_VECS];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave < nvec) {
        const float* v = vecs.v[wave];
        float a = 0.f;
        for (int j = lane; j < len; j += 64) a += v[j];
        a = wave_sum(a);
        if (lane == 0) {
            const float m = a / (float)len;
            sm[wave] = m;
            means[wave] = m;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int i = 0; i < nvec; ++i) t += sm[i];
        means[nvec] = t;
    }
}

// gvecs[i][j] = (gmeans[i] + gtotal[0]) / len   (either upstream gradient may be NULL = zero)
__global__ void __launch_bounds__(256) vec_means_bwd_kernel(const float* __restrict__ gmeans, const float* __restrict__ gtotal,
                                                            int nvec, int len, float* __restrict__ gvecs) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= nvec * len) return;
    const int i = q / len;
    gvecs[q] = ((gmeans ? gmeans[i] : 0.f) + (gtotal ? gtotal[0] : 0.f)) / (float)len;
}

}  // namespace

extern "C" {

int mnk_vec_means_fwd(const float* const* vecs, int nvec, int len, float* means, void* stream) {
    MNK_REQUIRE(vecs && means && nvec > 0 && nvec <= MAX_LOSS_VECS && len > 0);
    LossVecs lv;
    for (int i = 0; i < MAX_LOSS_VECS; ++i) lv.v[i] = i < nvec ? vecs[i] : nullptr;
    for (int i = 0; i < nvec; ++i) MNK_REQUIRE(lv.v[i]);
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(K_LOSS, s, (double)nvec * len * 4);
    hipLaunchKernelGGL(vec_means_fwd_kernel, dim3(1), dim3(64 * nvec), 0, s, lv, nvec, len, means);
    MNK_LAUNCH_CHECK();
    return MNK_OK;
}

int mnk_vec_means_bwd(const float* gmeans, const float* gtotal, int nvec, int len, float* gvecs, void* stream) {
    MNK_REQUIRE((gmeans || gtotal) && gvecs && nvec > 0 && nvec <= MAX_LOSS_VECS && len > 0);
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(K_LOSS, s, (double)nvec * len * 4);
    hipLaunchKernelGGL(vec_means_bwd_kernel, dim3(ceil_div(nvec * len, 256)), dim3(256), 0, s, gmeans, gtotal, nvec, len, gvecs);
    MNK_LAUNCH_CHECK();
    return MNK_OK;
}

int mnk_pair_l1_fwd(const float* a, int ld, long rows, int C, int B, float weight, float* out, void* stream) {
    MNK_REQUIRE(a && out && ld > 0 && ld % 4 == 0 && rows > 0 && C > 0 && C <= ld && B > 0);
    MNK_REQUIRE((size_t)a % 16 == 0);
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(K_LAYOUT, s, 2.0 * B * rows * ld * 4);
    hipLaunchKernelGGL(pair_l1_fwd_kernel, dim3(B), dim3(L1_THREADS), 0, s, a, ld, rows, C, B,
                       weight / (float)((double)rows * C), out);
    MNK_LAUNCH_CHECK();
    return MNK_OK;
}

int mnk_pair_l1_bwd_add(const float* a, int ld, long rows, int C, int B, float weight, const float* g, const float* addend,
                        float* da, void* stream) {
    MNK_REQUIRE(a && g && da && ld > 0 && ld % 4 == 0 && rows > 0 && C > 0 && C <= ld && B > 0);
    MNK_REQUIRE((size_t)a % 16 == 0 && (size_t)da % 16 == 0 && (size_t)addend % 16 == 0);
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(K_LAYOUT, s, (addend ? 6.0 : 4.0) * B * rows * ld * 4);
    const long total = (long)B * rows * (ld / 4);
    int blocks = ceil_div(total, 256 * 4);
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(pair_l1_bwd_kernel, dim3(blocks), dim3(256), 0, s, a, ld, rows, C, B,
                       weight / (float)((double)rows * C), g, addend, da);
    MNK_LAUNCH_CHECK();
    return MNK_OK;
}

int mnk_pair_l1_bwd(const float* a, int ld, long rows, int C, int B, float weight, const float* g, float* da,
                    void* stream) {
    return mnk_pair_l1_bwd_add(a, ld, rows, C, B, weight, g, nullptr, da, stream);
}

int mnk_l1_mean_fwd(const float* a, const float* b, long n, int B, float weight, float* out, void* stream) {
    MNK_REQUIRE(a && b && out && n > 0 && B > 0);
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(K_LOSS, s, 2.0 * B * n * 4);
    hipLaunchKernelGGL(l1_mean_fwd_kernel, dim3(B), dim3(L1_THREADS), 0, s, a, b, n, weight / (float)n, out);
    MNK_LAUNCH_CHECK();
    return MNK_OK;
}

int mnk_l1_mean_bwd(const float* a, const float* b, long n, int B, float weight, const float* g, float* da, float* db,
                    void* stream) {
    MNK_REQUIRE(a && b && g && (da || db) && n > 0 && B > 0);
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(K_LOSS, s, 4.0 * B * n * 4);
    const long total = (long)B * n;
    int blocks = ceil_div(total, 256 * 4);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(l1_mean_bwd_kernel, dim3(blocks < 1 ? 1 : blocks), dim3(256), 0, s, a, b, n, total, weight / (float)n, g,
                       da, db);
    MNK_LAUNCH_CHECK();
    return MNK_OK;
}

int mnk_gan_terms_fwd(const float* score, int n, int B, float w_gen, float w_disc, float* gen, float* disc, void* stream) {
    MNK_REQUIRE(score && gen && disc && n > 0 && B > 0);
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(K_LOSS, s, 2.0 * B * n * 4);
    hipLaunchKernelGGL(gan_terms_fwd_kernel, dim3(B), dim3(64), 0, s, score, n, B, w_gen, w_disc, gen, disc);
    MNK_LAUNCH_CHECK();
    return MNK_OK;
}

int mnk_gan_terms_bwd(const float* score, int n, int B, float w_gen, float w_disc, const float* ggen, const float* gdisc,
                      float* dscore, void* stream) {
    MNK_REQUIRE(score && dscore && (ggen || gdisc) && n > 0 && B > 0);
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(K_LOSS, s, 4.0 * B * n * 4);
    int blocks = ceil_div(2L * B * n, 256);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(gan_terms_bwd_kernel, dim3(blocks), dim3(256), 0, s, score, n, B, w_gen, w_disc, ggen, gdisc, dscore);
    MNK_LAUNCH_CHECK();
    return MNK_OK;
}
}

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
                                                          float* __restrict__ da) {
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
        *reinterpret_cast<float4*>(da + oa) = d;
        *reinterpret_cast<float4*>(da + ob) = make_float4(-d.x, -d.y, -d.z, -d.w);
    }
}

}  // namespace

extern "C" {

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

int mnk_pair_l1_bwd(const float* a, int ld, long rows, int C, int B, float weight, const float* g, float* da,
                    void* stream) {
    MNK_REQUIRE(a && g && da && ld > 0 && ld % 4 == 0 && rows > 0 && C > 0 && C <= ld && B > 0);
    MNK_REQUIRE((size_t)a % 16 == 0 && (size_t)da % 16 == 0);
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(K_LAYOUT, s, 4.0 * B * rows * ld * 4);
    const long total = (long)B * rows * (ld / 4);
    int blocks = ceil_div(total, 256 * 4);
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(pair_l1_bwd_kernel, dim3(blocks), dim3(256), 0, s, a, ld, rows, C, B,
                       weight / (float)((double)rows * C), g, da);
    MNK_LAUNCH_CHECK();
    return MNK_OK;
}
}

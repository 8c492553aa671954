// Adam for every tensor of a model in ONE launch, emitting the packed convolution-weight layouts of the NEXT iteration in
// the same pass.  Replaces torch.optim.Adam(lr, betas=(0.5, 0.999)) of train.py:81-83,118-136 (three optimisers over
// 166+ tensors: 8 multi-tensor launches per iteration + a separate re-pack launch reading every weight again).
//
// HBM-bound: per element it reads p, g, m, v and writes p, m, v (28 B); convolution weights additionally leave as their
// forward / data-gradient GEMM layouts (mnk_conv3x3_pack_multi's tile, straight from the LDS tile the update was made
// in), which saves re-reading the parameter in a separate pack launch.
//
// Update formula = torch/optim/adam.py::_single_tensor_adam (no amsgrad, no weight decay), evaluated in fp32:
//   m += (g - m) * (1 - b1);  v = b2 * v + (1 - b2) * g * g;
//   p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps)         with bc1 = 1 - b1^t, bc2 = 1 - b2^t
// The scalars live in device memory (`hyper`, 10 floats: lr, b1, b2, eps, lr / bc1, sqrt(bc2), grad scale, t, 1 - b1,
// 1 - b2), so a captured hipGraph picks up a new learning rate (MultiStepLR, train.py:91-96) and the step count without
// re-capture; mnk_adam_tick advances t and refreshes the two derived scalars.
#include "mnk_common.h"
#include "pack_tile.h"

using namespace mnk;

namespace {

struct Hyper {
    float b2, eps, step_size, bc2_sqrt, gscale, omb1, omb2;
};

__device__ __forceinline__ void adam_update(float& p, float g, float& m, float& v, const Hyper& h) {
    g *= h.gscale;
    m = m + (g - m) * h.omb1;
    v = h.b2 * v + h.omb2 * g * g;
    const float denom = sqrtf(v) / h.bc2_sqrt + h.eps;
    p = p - h.step_size * (m / denom);
}

// torch evaluates the scalars of a step in Python doubles (step_size = lr / (1 - b1^t), sqrt(1 - b2^t)) and hands
// their fp32 roundings to the tensor ops: the same here (1 - b1 and 1 - b2 come from the host's doubles: 1.f - 0.999f is
// 1.3e-5 away from 0.001f)
__global__ void adam_tick_kernel(float* hyper) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const float t = hyper[7] + 1.f;
        hyper[7] = t;
        const double b1 = 1.0 - (double)hyper[8], b2 = 1.0 - (double)hyper[9];
        hyper[4] = (float)((double)hyper[0] / (1.0 - pow(b1, (double)t)));
        hyper[5] = (float)sqrt(1.0 - pow(b2, (double)t));
    }
}

constexpr int PLAIN_PER_BLOCK = 4096;       // floats of a plain range handled by one block (256 threads x 4 float4)

__global__ void __launch_bounds__(256) adam_multi_kernel(const MnkAdamDesc* __restrict__ descs, int n,
                                                         const float* __restrict__ hyper) {
    __shared__ float T[16 * (16 * 17 + 1)];       // the tile [16 co][16 ci][taps] (padded strides); also the 16-tap staging area
    __shared__ int sh_idx;
    const int b = blockIdx.x;
    // the block's descriptor: one coalesced read of the block_begin column + an LDS count (a binary search over device
    // memory is ~8 dependent loads per block)
    if (threadIdx.x == 0) sh_idx = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 256) {
        const int i = base + (int)threadIdx.x;
        if (i < n && descs[i].block_begin <= b) atomicAdd(&sh_idx, 1);      // LDS atomic, <= n per block
    }
    __syncthreads();
    const MnkAdamDesc d = descs[sh_idx - 1];
    const int local = b - d.block_begin;
    Hyper h;
    h.b2 = hyper[2], h.eps = hyper[3], h.step_size = hyper[4], h.bc2_sqrt = hyper[5], h.gscale = hyper[6];
    h.omb1 = hyper[8], h.omb2 = hyper[9];
    const int t = threadIdx.x;
    if (!d.wp_fwd) {
        // ---- plain range ----------------------------------------------------------------------------------------
        const long base = (long)local * PLAIN_PER_BLOCK;
        const bool vec = (((size_t)d.p | (size_t)d.g | (size_t)d.m | (size_t)d.v) & 15) == 0;
#pragma unroll
        for (int j = 0; j < PLAIN_PER_BLOCK / 1024; ++j) {
            const long i = base + (long)(j * 256 + t) * 4;
            if (i >= d.n) break;
            if (vec && i + 4 <= d.n) {
                float4 p4 = *reinterpret_cast<const float4*>(d.p + i), g4 = *reinterpret_cast<const float4*>(d.g + i);
                float4 m4 = *reinterpret_cast<const float4*>(d.m + i), v4 = *reinterpret_cast<const float4*>(d.v + i);
                adam_update(p4.x, g4.x, m4.x, v4.x, h);
                adam_update(p4.y, g4.y, m4.y, v4.y, h);
                adam_update(p4.z, g4.z, m4.z, v4.z, h);
                adam_update(p4.w, g4.w, m4.w, v4.w, h);
                *reinterpret_cast<float4*>(d.p + i) = p4;
                *reinterpret_cast<float4*>(d.m + i) = m4;
                *reinterpret_cast<float4*>(d.v + i) = v4;
            } else {
                for (long k = i; k < i + 4 && k < d.n; ++k) {
                    float p = d.p[k], m = d.m[k], v = d.v[k];
                    adam_update(p, d.g[k], m, v, h);
                    d.p[k] = p, d.m[k] = m, d.v[k] = v;
                }
            }
        }
        return;
    }
    // ---- 3x3 convolution weight: one 16 (co) x 16 (ci) x 9 tile, updated and re-packed ---------------------------
    const int C0p = (d.C0 + 15) & ~15, C1p = d.C1 > 0 ? (d.C1 + 15) & ~15 : 0, tiles_x = (C0p + C1p) / 16;
    const int cot = local / tiles_x, cc = local - cot * tiles_x;
    const PackTileGeom g = pack_tile_geom<9>(d.Cout, d.C0, d.C1, C0p, C1p, 9, cc, cot);
    constexpr int run = 16 * 9, NI = 16 * run / 256;        // nine elements per thread
    static_assert(16 * run % 256 == 0, "a tile is a whole number of passes of the block");
    // the gradient of this tile's source straight from the tap-major partials of the grouped weight-gradient GEMMs (few-split
    // layers: the deep levels, whose partials ARE the gradient): staged through T with 64-byte runs along ci, summed over the
    // splits in order, the 16 pseudo taps of a sub-pixel form folded -- what mnk_wgrad_reduce_multi would have written to `g`
    const float* gt = g.second ? d.gt1 : d.gt0;
    float gv[NI];
    if (gt) {
        const int nin = (d.flags & (g.second ? 4 : 2)) ? 16 : 9;
        const int splits = g.second ? d.gt_splits1 : d.gt_splits0;
        const long plane = (long)d.Cout * g.Cs, sstride = (long)nin * plane;
        for (int i = t; i < nin * 256; i += 256) {
            const int ci = i & 15, r = (i >> 4) & 15, tp = i >> 8;
            const int co = g.co0 + r, c = g.ci0 + ci;
            float v = 0.f;
            if (co < d.Cout && c < g.Cs) {
                const float* src = gt + ((long)tp * d.Cout + co) * g.Cs + c;
                for (int sp = 0; sp < splits; ++sp) v += src[(long)sp * sstride];
            }
            T[(r * 16 + ci) * 17 + tp] = v;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < NI; ++k) {
            const int i = t + k * 256;
            const int r = i / run, o = i - r * run;
            const int ci = o / 9, tap = o - ci * 9;
            const float* a = T + (r * 16 + ci) * 17;
            gv[k] = nin == 16 ? up_fold(a, tap / 3, tap - (tap / 3) * 3, 1) : a[tap];
        }
        __syncthreads();          // T is re-used for the updated parameters below
    }
#pragma unroll
    for (int k = 0; k < NI; ++k) {
        const int i = t + k * 256;
        const int r = i / run, o = i - r * run;     // row (co), offset inside the row = ci * 9 + tap
        const int ci = o / 9, tap = o - ci * 9;
        const int co = g.co0 + r;
        float p = 0.f;
        if (co < d.Cout && g.ci0 + ci < g.Cs) {
            const size_t idx = ((size_t)co * g.Cin + g.cstart + g.ci0) * 9 + o;
            p = d.p[idx];
            float m = d.m[idx], v = d.v[idx];
            adam_update(p, gt ? gv[k] : d.g[idx], m, v, h);
            d.p[idx] = p;
            d.m[idx] = m;
            d.v[idx] = v;
        }
        T[r * g.cos + ci * g.ntp + tap] = p;
    }
    __syncthreads();
    if (d.flags & 1)       // an up-sampled convolution (UpBlock3D): the packs of its sub-pixel forms
        pack_tile_emit_up(T, g, d.wp_fwd, d.wp_d0, d.wp_d1, d.Cout, cc, cot);
    else
        pack_tile_emit<9>(T, g, d.wp_fwd, d.wp_d0, d.wp_d1, d.Cout, cc, cot);
}

}  // namespace

extern "C" {

int mnk_adam_blocks(long n, int Cout, int C0, int C1, int packed) {
    if (packed) {
        if (Cout <= 0 || C0 <= 0 || C1 < 0) return 0;
        return ((round_up(C0, 16) + (C1 > 0 ? round_up(C1, 16) : 0)) / 16) * ceil_div(Cout, 16);
    }
    return n > 0 ? (int)((n + PLAIN_PER_BLOCK - 1) / PLAIN_PER_BLOCK) : 0;
}

int mnk_adam_tick(float* hyper, void* stream) {
    MNK_REQUIRE(hyper);
    hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, hyper);
    MNK_LAUNCH_CHECK();
    return MNK_OK;
}

int mnk_adam_multi(const MnkAdamDesc* descs_device, int n, int total_blocks, const float* hyper, void* stream) {
    MNK_REQUIRE(descs_device && hyper && n > 0 && total_blocks > 0);
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(K_OPTIM, s, 0.0);
    hipLaunchKernelGGL(adam_multi_kernel, dim3(total_blocks), dim3(256), 0, s, descs_device, n, hyper);
    MNK_LAUNCH_CHECK();
    return MNK_OK;
}
}

// Column sums of a [rows][C] fp32 matrix, two ways: (a) per-block partial rows + a second-stage launch (what the BatchNorm
// statistics do today), (b) one launch whose blocks add their partial sums to per-column fp64 accumulators with device-scope
// atomics (no second launch; the consumer reads 2 * C doubles); (c) (round 4) the same with one accumulator row PER XCD and
// workgroup-scope atomics -- a block adds to the row of the XCD it runs on (s_getreg XCC_ID), so the read-modify-writes stay in
// that XCD's L2 and never meet another XCD's; the kernel boundary publishes them, the consumer adds 8 rows.  Measures a dependent chain of the two forms back to back,
// the way they sit in a training iteration: (a) = 2 launches per layer, (b) = 1.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/colsum tools/microbench/colsum_atomics.hip && /tmp/colsum
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

__global__ void __launch_bounds__(256) partial_kernel(const float* __restrict__ x, long rows, int C, long rows_per_block,
                                                      float* __restrict__ partial, double* __restrict__ acc, int per_xcd = 0) {
    __shared__ float4 red[2][256];
    const int nv = C / 4, tx_n = nv < 16 ? nv : 16, ty_n = 256 / tx_n;
    const int tx = threadIdx.x % tx_n, ty = threadIdx.x / tx_n;
    const int q = blockIdx.x * tx_n + tx;
    const long r0 = (long)blockIdx.y * rows_per_block;
    long r1 = r0 + rows_per_block;
    if (r1 > rows) r1 = rows;
    float4 a = make_float4(0, 0, 0, 0), b = a;
    if (q < nv)
        for (long r = r0 + ty; r < r1; r += ty_n) {
            const float4 v = *reinterpret_cast<const float4*>(x + r * C + q * 4);
            a.x += v.x, a.y += v.y, a.z += v.z, a.w += v.w;
            b.x += v.x * v.x, b.y += v.y * v.y, b.z += v.z * v.z, b.w += v.w * v.w;
        }
    red[0][threadIdx.x] = a;
    red[1][threadIdx.x] = b;
    __syncthreads();
    for (int s = ty_n >> 1; s > 0; s >>= 1) {
        if (ty < s) {
            float4 u = red[0][threadIdx.x + s * tx_n], w = red[1][threadIdx.x + s * tx_n];
            float4& p = red[0][threadIdx.x];
            float4& p2 = red[1][threadIdx.x];
            p.x += u.x, p.y += u.y, p.z += u.z, p.w += u.w;
            p2.x += w.x, p2.y += w.y, p2.z += w.z, p2.w += w.w;
        }
        __syncthreads();
    }
    if (ty == 0 && q < nv) {
        if (acc && per_xcd) {
            unsigned xcc = 0;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            const float4 s1 = red[0][tx], s2 = red[1][tx];
            double* o = acc + (size_t)(xcc & 7) * 2 * C + q * 4;
            const double v[8] = {s1.x, s1.y, s1.z, s1.w, s2.x, s2.y, s2.z, s2.w};
            for (int e = 0; e < 4; ++e) {
                __hip_atomic_fetch_add(o + e, v[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(o + C + e, v[4 + e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        } else if (acc) {
            const float4 s1 = red[0][tx], s2 = red[1][tx];
            double* o = acc + q * 4;
            atomicAdd(o, (double)s1.x), atomicAdd(o + 1, (double)s1.y), atomicAdd(o + 2, (double)s1.z), atomicAdd(o + 3, (double)s1.w);
            o += C;
            atomicAdd(o, (double)s2.x), atomicAdd(o + 1, (double)s2.y), atomicAdd(o + 2, (double)s2.z), atomicAdd(o + 3, (double)s2.w);
        } else {
            float* o = partial + (long)blockIdx.y * 2 * C;
            *reinterpret_cast<float4*>(o + q * 4) = red[0][tx];
            *reinterpret_cast<float4*>(o + C + q * 4) = red[1][tx];
        }
    }
}

__global__ void __launch_bounds__(256) final_kernel(const float* __restrict__ partial, int row_blocks, int C, float* __restrict__ sums) {
    __shared__ double sm[256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = blockIdx.x * 4 + wave;
    double acc = 0;
    if (i < 2 * C)
        for (int rb = lane; rb < row_blocks; rb += 64) acc += (double)partial[(long)rb * 2 * C + i];
    sm[threadIdx.x] = acc;
    __syncthreads();
    if (lane == 0 && i < 2 * C) {
        double t = 0;
        for (int j = 0; j < 64; ++j) t += sm[wave * 64 + j];
        sums[i] = (float)t;
    }
}

// the consumer: y = (x - mean) * invstd; statistics from `sums` (float) or from the fp64 accumulators
__global__ void __launch_bounds__(256) apply_kernel(const float* __restrict__ x, long rows, int C, const float* __restrict__ sums,
                                                    const double* __restrict__ acc, float* __restrict__ y, int per_xcd = 0) {
    const long total = rows * (C / 4);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int q = (int)(i % (C / 4));
        float m[4], is[4];
        for (int e = 0; e < 4; ++e) {
            const int c = q * 4 + e;
            double s1 = acc ? acc[c] : (double)sums[c], s2 = acc ? acc[C + c] : (double)sums[C + c];
            if (per_xcd)
                for (int x8 = 1; x8 < 8; ++x8) s1 += acc[(size_t)x8 * 2 * C + c], s2 += acc[(size_t)x8 * 2 * C + C + c];
            const double mean = s1 / (double)rows;
            double var = s2 / (double)rows - mean * mean;
            m[e] = (float)mean;
            is[e] = 1.f / sqrtf((float)(var > 0 ? var : 0) + 1e-5f);
        }
        const float4 v = *reinterpret_cast<const float4*>(x + i * 4);
        *reinterpret_cast<float4*>(y + i * 4) = make_float4((v.x - m[0]) * is[0], (v.y - m[1]) * is[1], (v.z - m[2]) * is[2], (v.w - m[3]) * is[3]);
    }
}

int main() {
    struct Case { long rows; int C; } cases[] = {{131072, 48}, {131072, 64}, {32768, 128}, {8192, 256}, {2048, 512}, {32768, 64}};
    for (auto cs : cases) {
        const long rows = cs.rows;
        const int C = cs.C;
        const int nv = C / 4, tx_n = nv < 16 ? nv : 16, ty_n = 256 / tx_n;
        const long rpb = 4 * ty_n;
        const int row_blocks = (int)((rows + rpb - 1) / rpb), col_tiles = (nv + tx_n - 1) / tx_n;
        float *x, *y, *partial, *sums;
        double* acc;
        hipMalloc(&x, rows * C * 4), hipMalloc(&y, rows * C * 4), hipMalloc(&partial, (size_t)row_blocks * 2 * C * 4);
        hipMalloc(&sums, 2 * C * 4), hipMalloc(&acc, (size_t)2 * C * 8 * 64 * 8);
        std::vector<float> h(rows * C);
        for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u >> 8) & 0xffff) / 65536.f - 0.5f;
        hipMemcpy(x, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        hipMemset(acc, 0, (size_t)2 * C * 8 * 64 * 8);
        hipEvent_t e0, e1;
        hipEventCreate(&e0), hipEventCreate(&e1);
        const int reps = 50, ablocks = 2048;
        float ms[3];
        for (int form = 0; form < 3; ++form) {
            for (int warm = 0; warm < 2; ++warm) {
                hipEventRecord(e0, 0);
                for (int r = 0; r < reps; ++r) {
                    // a chain of layers: statistics of y_prev -> apply -> ... (x and y alternate)
                    const float* src = (r & 1) ? y : x;
                    float* dst = (r & 1) ? x : y;
                    if (form == 0) {
                        hipLaunchKernelGGL(partial_kernel, dim3(col_tiles, row_blocks), dim3(256), 0, 0, src, rows, C, rpb, partial, (double*)nullptr);
                        hipLaunchKernelGGL(final_kernel, dim3((2 * C + 3) / 4), dim3(256), 0, 0, partial, row_blocks, C, sums);
                        hipLaunchKernelGGL(apply_kernel, dim3(ablocks), dim3(256), 0, 0, src, rows, C, sums, (const double*)nullptr, dst);
                    } else if (form == 2) {
                        double* a = acc + (size_t)(r % 64) * 2 * C * 8;      // eight rows (one per XCD) per layer
                        hipLaunchKernelGGL(partial_kernel, dim3(col_tiles, row_blocks), dim3(256), 0, 0, src, rows, C, rpb, (float*)nullptr, a, 1);
                        hipLaunchKernelGGL(apply_kernel, dim3(ablocks), dim3(256), 0, 0, src, rows, C, (const float*)nullptr, a, dst, 1);
                    } else {
                        double* a = acc + (size_t)(r % 64) * 2 * C;          // a fresh (zeroed) accumulator per layer
                        hipLaunchKernelGGL(partial_kernel, dim3(col_tiles, row_blocks), dim3(256), 0, 0, src, rows, C, rpb, (float*)nullptr, a);
                        hipLaunchKernelGGL(apply_kernel, dim3(ablocks), dim3(256), 0, 0, src, rows, C, (const float*)nullptr, a, dst);
                    }
                }
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
                hipEventElapsedTime(&ms[form], e0, e1);
                hipMemset(acc, 0, (size_t)2 * C * 8 * 64 * 8);
                hipDeviceSynchronize();
            }
        }
        printf("rows %7ld C %4d row_blocks %5d: partial+final+apply %.2f us per layer | device-scope atomics+apply %.2f | per-XCD rows, "
               "workgroup-scope atomics+apply %.2f\n", rows, C, row_blocks, ms[0] * 1e3 / reps, ms[1] * 1e3 / reps, ms[2] * 1e3 / reps);
        hipFree(x), hipFree(y), hipFree(partial), hipFree(sums), hipFree(acc);
    }
    return 0;
}

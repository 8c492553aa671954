#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
template <int WHICH>
__global__ void __launch_bounds__(256) rate(float* out, int iters) {
    f32x16 a16[4]; f32x4 a4[4];
    for (int q = 0; q < 4; ++q) { for (int r = 0; r < 16; ++r) a16[q][r] = 0.f; for (int r = 0; r < 4; ++r) a4[q][r] = 0.f; }
    bf16x8 x, y; s16x4 xs, ys;
    for (int e = 0; e < 8; ++e) { x[e] = (__bf16)(0.001f * (threadIdx.x + e)); y[e] = (__bf16)(0.002f * (threadIdx.x - e)); }
    for (int e = 0; e < 4; ++e) { xs[e] = (short)(threadIdx.x + e); ys[e] = (short)(threadIdx.x * 3 + e); }
    float fx = 0.001f * threadIdx.x, fy = 0.002f * threadIdx.x;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (WHICH == 0) a16[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a16[q], 0, 0, 0);
            if (WHICH == 1) a4[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, a4[q], 0, 0, 0);
            if (WHICH == 2) a4[q] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(xs, ys, a4[q], 0, 0, 0);
            if (WHICH == 3) a16[q] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(xs, ys, a16[q], 0, 0, 0);
            if (WHICH == 4) a4[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(fx, fy, a4[q], 0, 0, 0);
            if (WHICH == 5) a16[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(fx, fy, a16[q], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int q = 0; q < 4; ++q) { for (int r = 0; r < 16; ++r) s += a16[q][r]; for (int r = 0; r < 4; ++r) s += a4[q][r]; }
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int W> void run(const char* name, double macs) {
    float* d; hipMalloc(&d, 1024 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000;
    hipLaunchKernelGGL(rate<W>, dim3(1024), dim3(256), 0, 0, d, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(rate<W>, dim3(1024), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double n = 1024.0 * 4 * iters * 4;   // waves * mfma
    printf("%-28s %8.1f TFLOP/s  (%.1f cycles per MFMA per SIMD at 2.4 GHz, 4 waves/SIMD)\n", name, 2 * macs * n / (ms * 1e-3) / 1e12,
           ms * 1e-3 * 2.4e9 / (n / (256.0 * 4)));
    hipFree(d);
}
int main() {
    run<0>("32x32x16 bf16", 32. * 32 * 16);
    run<1>("16x16x32 bf16", 16. * 16 * 32);
    run<2>("16x16x16 bf16 (1k)", 16. * 16 * 16);
    run<3>("32x32x8 bf16 (1k)", 32. * 32 * 8);
    run<4>("16x16x4 f32", 16. * 16 * 4);
    run<5>("32x32x2 f32", 32. * 32 * 2);
    return 0;
}

// What does the fp32 matrix pipe of THIS chip sustain on real data?  bench.py prices the GEMM kernels against the 157.3 TFLOP/s
// of MI355X_MICROARCH.md (64 FLOP/clk/SIMD x 1024 SIMDs x 2.4 GHz); the chip clocks to its power budget (same guide, "DVFS
// give-back": zero-filled inputs ran +19 % over random ones on a bf16 kernel).  This micro-benchmark runs nothing but
// v_mfma_f32_32x32x2_f32 -- ACC independent accumulator chains per wave, WAVES waves per SIMD, operands in registers -- on
// zeros and on random values, and prints TFLOP/s and the clock that rate implies.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_peak tools/microbench/mfma_f32_peak.hip && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int ACC>
__global__ void __launch_bounds__(256) mfma_loop(const float* __restrict__ in, float* __restrict__ out, int iters) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    float a[4], b[4];
    for (int i = 0; i < 4; ++i) {
        a[i] = in[(t * 8 + i) & 0xfffff];
        b[i] = in[(t * 8 + 4 + i) & 0xfffff];
    }
    f32x16 acc[ACC];
    for (int q = 0; q < ACC; ++q)
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int q = 0; q < ACC; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k], b[(k + q) & 3], acc[q], 0, 0, 0);
    }
    float s = 0.f;
    for (int q = 0; q < ACC; ++q)
        for (int r = 0; r < 16; ++r) s += acc[q][r];
    if (s == 123.456f) out[t] = s;       // keeps the chains alive
}

template <int ACC>
double run(const float* in, float* out, int blocks, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(mfma_loop<ACC>, dim3(blocks), dim3(256), 0, 0, in, out, iters / 8);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(mfma_loop<ACC>, dim3(blocks), dim3(256), 0, 0, in, out, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)blocks * 4 /* waves */ * iters * 4 * ACC * (32.0 * 32 * 2 * 2);
    return flop / (ms * 1e-3) / 1e12;
}

int main() {
    const size_t n = 1 << 20;
    std::vector<float> h(n);
    float *zero, *rnd, *out;
    hipMalloc(&zero, n * 4);
    hipMalloc(&rnd, n * 4);
    hipMalloc(&out, (size_t)256 * 8 * 256 * 4);
    hipMemset(zero, 0, n * 4);
    srand(1);
    for (size_t i = 0; i < n; ++i) h[i] = (float)rand() / RAND_MAX * 2.f - 1.f;
    hipMemcpy(rnd, h.data(), n * 4, hipMemcpyHostToDevice);
    const double spec = 157.3;
    printf("%-8s %-6s %-10s %10s %8s %12s\n", "data", "acc", "waves/SIMD", "TFLOP/s", "of spec", "implied GHz");
    for (int waves = 1; waves <= 4; waves *= 2) {
        const int blocks = 256 * waves;          // 256 CUs x `waves` blocks of 4 waves = `waves` waves per SIMD
        for (int pass = 0; pass < 2; ++pass) {
            const float* src = pass ? rnd : zero;
            const double t1 = run<1>(src, out, blocks, 200000 / waves), t2 = run<2>(src, out, blocks, 100000 / waves),
                         t4 = run<4>(src, out, blocks, 50000 / waves);
            const double ts[3] = {t1, t2, t4};
            for (int k = 0; k < 3; ++k)
                printf("%-8s %-6d %-10d %10.1f %8.3f %12.2f\n", pass ? "random" : "zeros", 1 << k, waves, ts[k], ts[k] / spec,
                       ts[k] * 1e12 / (64.0 * 1024) / 1e9);
        }
    }
    return 0;
}

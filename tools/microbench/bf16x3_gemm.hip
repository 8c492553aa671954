// What would an fp32-ACCURATE GEMM on the bf16 matrix cores cost on gfx950?  (round 6; DESIGN.md section 8, "what comes next")
//
// v_mfma_f32_32x32x16_bf16 issues 16x the multiply-adds of v_mfma_f32_32x32x2_f32 per cycle.  Split both fp32 operands exactly
// into three bf16 terms (a = a1 + a2 + a3: a1 = bf16(a), a2 = bf16(a - a1), a3 = bf16(a - a1 - a2); the subtractions are exact)
// and keep the six cross products whose weight is >= 2^-16 of the leading one -- a1b1, a1b2, a2b1, a1b3, a3b1, a2b2 -- all
// accumulated in fp32 inside the MFMA: every bf16 x bf16 product is exact in fp32, the dropped terms are <= 3 * 2^-24 relative,
// i.e. the result is within ~2 ulp of the fp32 product chain.  Six MFMAs at 16x the rate = 2.67x the fp32 matrix peak.
//
// This file measures, on the tile the library's convolutions use most (64 x 64 block, four wavefronts of 32 x 32, both
// operands K-contiguous, C = A[M][K] * B[N][K]^T):
//   f32        the fp32 MFMA (32x32x2), b128 fragment reads -- the structure of csrc/conv3x3.hip's K loop
//   split-in   bf16x3, fp32 operands split by the LOADER (7 vector instructions per element between global load and LDS store)
//   presplit   bf16x3, operands already stored as three bf16 planes (what a producer epilogue / the optimiser kernel would leave)
// and the error of each against an fp64 product on the host.
// Build: hipcc --offload-arch=gfx950 -O3 -o bf16x3_gemm tools/microbench/bf16x3_gemm.hip ; run: ./bf16x3_gemm
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) unsigned short u16x8;

#define CHECK(x)                                                                    \
    do {                                                                            \
        hipError_t e = (x);                                                         \
        if (e != hipSuccess) {                                                      \
            printf("%s: %s\n", #x, hipGetErrorString(e));                           \
            exit(1);                                                                \
        }                                                                           \
    } while (0)

__device__ __forceinline__ unsigned short bf16_rn(float x) {       // round to nearest even (finite inputs)
    unsigned u = __float_as_uint(x);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf16_f(unsigned short h) { return __uint_as_float((unsigned)h << 16); }
__device__ __forceinline__ void split3(float x, unsigned short& h, unsigned short& m, unsigned short& l) {
    h = bf16_rn(x);
    const float r = x - bf16_f(h);
    m = bf16_rn(r);
    l = bf16_rn(r - bf16_f(m));
}

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
// float4 -> three planes of four bf16 (two packed dwords each), a = h + m + l exactly up to the last plane's rounding
__device__ __forceinline__ void split3_x4(float4 v, uint2 (&out)[3]) {
    f32x2 lo = {v.x, v.y}, hi = {v.z, v.w};
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
        const bf16x2 a = __builtin_convertvector(lo, bf16x2), b = __builtin_convertvector(hi, bf16x2);
        out[pl] = make_uint2(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b));
        if (pl < 2) {
            lo = lo - __builtin_convertvector(a, f32x2);
            hi = hi - __builtin_convertvector(b, f32x2);
        }
    }
}

constexpr int BM = 64, BN = 64;

// ---- fp32 MFMA ---------------------------------------------------------------------------------------------------------------
template <int BK>
__global__ void __launch_bounds__(256) gemm_f32(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
                                                int M, int N, int K) {
    constexpr int LD = BK + 4;
    __shared__ __attribute__((aligned(16))) float As[2][BM][LD], Bs[2][BN][LD];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    constexpr int V = BK / 4, ROWS = 256 / V, PASS = BM / ROWS;       // float4 per row, rows per pass
    const int lr = t / V, lc = t % V;
    float4 ra[PASS], rb[PASS];
    auto gload = [&](int k0) {
#pragma unroll
        for (int p = 0; p < PASS; ++p) {
            ra[p] = *reinterpret_cast<const float4*>(A + (long)(m0 + lr + p * ROWS) * K + k0 + lc * 4);
            rb[p] = *reinterpret_cast<const float4*>(B + (long)(n0 + lr + p * ROWS) * K + k0 + lc * 4);
        }
    };
    auto sstore = [&](int buf) {
#pragma unroll
        for (int p = 0; p < PASS; ++p) {
            *reinterpret_cast<float4*>(&As[buf][lr + p * ROWS][lc * 4]) = ra[p];
            *reinterpret_cast<float4*>(&Bs[buf][lr + p * ROWS][lc * 4]) = rb[p];
        }
    };
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int fi = lane & 31, fk = lane >> 5;
    gload(0);
    sstore(0);
    __syncthreads();
    int buf = 0;
    for (int k0 = 0; k0 < K; k0 += BK) {
        if (k0 + BK < K) gload(k0 + BK);
#pragma unroll
        for (int kk = 0; kk < BK / 8; ++kk) {
            const float4 a = *reinterpret_cast<const float4*>(&As[buf][wm * 32 + fi][kk * 8 + fk * 4]);
            const float4 b = *reinterpret_cast<const float4*>(&Bs[buf][wn * 32 + fi][kk * 8 + fk * 4]);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
        }
        if (k0 + BK < K) sstore(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk, col = n0 + wn * 32 + fi;
        C[(long)row * N + col] = acc[r];
    }
}

// ---- bf16 x 3 ----------------------------------------------------------------------------------------------------------------
// PRE: the operands are three bf16 planes [3][rows][K] in global memory; else fp32 [rows][K], split by the loader
template <int BK, bool PRE>
__global__ void __launch_bounds__(256) gemm_bf16x3(const void* __restrict__ Av, const void* __restrict__ Bv, float* __restrict__ C,
                                                   int M, int N, int K) {
    constexpr int LD = BK + 8;            // bf16 per LDS row (16-byte multiple; 8 extra: conflict-free b128 fragment reads)
    __shared__ __attribute__((aligned(16))) unsigned short As[2][3][BM][LD], Bs[2][3][BN][LD];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int fi = lane & 31, fk = lane >> 5;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    // loader state
    constexpr int V4 = BK / 4, ROWS4 = 256 / V4, PASS4 = BM / ROWS4;            // fp32 source: float4 per row
    constexpr int V8 = BK / 8, ROWS8 = 256 / V8, PASS8 = (BM + ROWS8 - 1) / ROWS8;   // bf16 planes: 8 bf16 (16 B) per vector
    float4 ra[PASS4], rb[PASS4];
    u16x8 pa[3][PASS8], pb[3][PASS8];
    const int lr4 = t / V4, lc4 = t % V4, lr8 = t / V8, lc8 = t % V8;
    auto gload = [&](int k0) {
        if constexpr (PRE) {
            const unsigned short* A = (const unsigned short*)Av;
            const unsigned short* B = (const unsigned short*)Bv;
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                for (int p = 0; p < PASS8; ++p) {
                    const int row = lr8 + p * ROWS8;
                    if (ROWS8 * PASS8 == BM || row < BM) {
                        pa[pl][p] = *reinterpret_cast<const u16x8*>(A + ((long)pl * M + m0 + row) * K + k0 + lc8 * 8);
                        pb[pl][p] = *reinterpret_cast<const u16x8*>(B + ((long)pl * N + n0 + row) * K + k0 + lc8 * 8);
                    }
                }
        } else {
            const float* A = (const float*)Av;
            const float* B = (const float*)Bv;
#pragma unroll
            for (int p = 0; p < PASS4; ++p) {
                ra[p] = *reinterpret_cast<const float4*>(A + (long)(m0 + lr4 + p * ROWS4) * K + k0 + lc4 * 4);
                rb[p] = *reinterpret_cast<const float4*>(B + (long)(n0 + lr4 + p * ROWS4) * K + k0 + lc4 * 4);
            }
        }
    };
    auto sstore = [&](int buf) {
        if constexpr (PRE) {
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                for (int p = 0; p < PASS8; ++p) {
                    const int row = lr8 + p * ROWS8;
                    if (ROWS8 * PASS8 == BM || row < BM) {
                        *reinterpret_cast<u16x8*>(&As[buf][pl][row][lc8 * 8]) = pa[pl][p];
                        *reinterpret_cast<u16x8*>(&Bs[buf][pl][row][lc8 * 8]) = pb[pl][p];
                    }
                }
        } else {
#pragma unroll
            for (int p = 0; p < PASS4; ++p) {
                // the hardware conversion (v_cvt_pk_bf16_f32, round to nearest even) on float pairs: 3 conversions, 4 and / shift
                // and 2 packed subtractions per PAIR -- 4.5 vector instructions per element
                uint2 ha[3], hb[3];
                split3_x4(ra[p], ha);
                split3_x4(rb[p], hb);
                const int row = lr4 + p * ROWS4;
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    *reinterpret_cast<uint2*>(&As[buf][pl][row][lc4 * 4]) = ha[pl];
                    *reinterpret_cast<uint2*>(&Bs[buf][pl][row][lc4 * 4]) = hb[pl];
                }
            }
        }
    };
    gload(0);
    sstore(0);
    __syncthreads();
    int buf = 0;
    for (int k0 = 0; k0 < K; k0 += BK) {
        if (k0 + BK < K) gload(k0 + BK);
#pragma unroll
        for (int kk = 0; kk < BK / 16; ++kk) {
            bf16x8 a[3], b[3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                a[pl] = *reinterpret_cast<const bf16x8*>(&As[buf][pl][wm * 32 + fi][kk * 16 + fk * 8]);
                b[pl] = *reinterpret_cast<const bf16x8*>(&Bs[buf][pl][wn * 32 + fi][kk * 16 + fk * 8]);
            }
            // smallest terms first
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
        }
        if (k0 + BK < K) sstore(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk, col = n0 + wn * 32 + fi;
        C[(long)row * N + col] = acc[r];
    }
}

__global__ void presplit_kernel(const float* __restrict__ x, unsigned short* __restrict__ planes, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned short h, m, l;
    split3(x[i], h, m, l);
    planes[i] = h;
    planes[n + i] = m;
    planes[2 * n + i] = l;
}

template <class F>
static float time_ms(F launch, int iters) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) launch();
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) launch();
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms / iters;
}

static void run(int M, int N, int K, bool gaussian) {
    std::vector<float> hA((size_t)M * K), hB((size_t)N * K);
    srand(7);
    auto rnd = [&]() {
        if (!gaussian) return (float)rand() / (float)RAND_MAX - 0.5f;
        float u1 = ((float)rand() + 1.f) / ((float)RAND_MAX + 2.f), u2 = (float)rand() / (float)RAND_MAX;
        return sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2);
    };
    for (auto& v : hA) v = rnd();
    for (auto& v : hB) v = rnd();
    float *dA, *dB, *dC;
    unsigned short *pA, *pB;
    CHECK(hipMalloc(&dA, hA.size() * 4));
    CHECK(hipMalloc(&dB, hB.size() * 4));
    CHECK(hipMalloc(&dC, (size_t)M * N * 4));
    CHECK(hipMalloc(&pA, hA.size() * 6));
    CHECK(hipMalloc(&pB, hB.size() * 6));
    CHECK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(presplit_kernel, dim3((unsigned)((hA.size() + 255) / 256)), dim3(256), 0, 0, dA, pA, (long)hA.size());
    hipLaunchKernelGGL(presplit_kernel, dim3((unsigned)((hB.size() + 255) / 256)), dim3(256), 0, 0, dB, pB, (long)hB.size());
    CHECK(hipDeviceSynchronize());
    // fp64 reference of a 64 x 64 corner and of the last block
    const int RM = 64, RN = 64;
    std::vector<double> ref((size_t)RM * RN);
    double scale = 0.0;
    for (int i = 0; i < RM; ++i)
        for (int j = 0; j < RN; ++j) {
            double s = 0.0, sa = 0.0;
            for (int k = 0; k < K; ++k) {
                s += (double)hA[(size_t)i * K + k] * (double)hB[(size_t)j * K + k];
                sa += fabs((double)hA[(size_t)i * K + k] * (double)hB[(size_t)j * K + k]);
            }
            ref[(size_t)i * RN + j] = s;
            scale = sa > scale ? sa : scale;
        }
    const dim3 grid(M / BM, N / BN);
    const double flop = 2.0 * M * N * K;
    std::vector<float> hC((size_t)RM * N);
    auto report = [&](const char* name, float ms) {
        CHECK(hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost));
        double err = 0.0;
        for (int i = 0; i < RM; ++i)
            for (int j = 0; j < RN; ++j) {
                const double d = fabs((double)hC[(size_t)i * N + j] - ref[(size_t)i * RN + j]);
                err = d > err ? d : err;
            }
        printf("  %-28s %8.1f us  %7.1f TFLOP/s  (%.2f of the 157.3 fp32-MFMA peak)   max |err| / sum|a b| = %.2e\n", name, ms * 1e3,
               flop / (ms * 1e-3) / 1e12, flop / (ms * 1e-3) / 1e12 / 157.3, err / scale);
    };
    printf("M %d  N %d  K %d  (%s operands), 64x64 block tile, %d blocks\n", M, N, K, gaussian ? "gaussian" : "uniform", grid.x * grid.y);
    report("f32 MFMA  BK 16", time_ms([&] { hipLaunchKernelGGL((gemm_f32<16>), grid, dim3(256), 0, 0, dA, dB, dC, M, N, K); }, 20));
    report("f32 MFMA  BK 32", time_ms([&] { hipLaunchKernelGGL((gemm_f32<32>), grid, dim3(256), 0, 0, dA, dB, dC, M, N, K); }, 20));
    report("bf16x3 split in loader BK 32", time_ms([&] { hipLaunchKernelGGL((gemm_bf16x3<32, false>), grid, dim3(256), 0, 0, dA, dB, dC, M, N, K); }, 20));
    report("bf16x3 split in loader BK 64", time_ms([&] { hipLaunchKernelGGL((gemm_bf16x3<64, false>), grid, dim3(256), 0, 0, dA, dB, dC, M, N, K); }, 20));
    report("bf16x3 pre-split planes BK 32", time_ms([&] { hipLaunchKernelGGL((gemm_bf16x3<32, true>), grid, dim3(256), 0, 0, pA, pB, dC, M, N, K); }, 20));
    report("bf16x3 pre-split planes BK 64", time_ms([&] { hipLaunchKernelGGL((gemm_bf16x3<64, true>), grid, dim3(256), 0, 0, pA, pB, dC, M, N, K); }, 20));
    CHECK(hipFree(dA));
    CHECK(hipFree(dB));
    CHECK(hipFree(dC));
    CHECK(hipFree(pA));
    CHECK(hipFree(pB));
}

int main(int argc, char** argv) {
    if (argc == 4) {
        run(atoi(argv[1]), atoi(argv[2]), atoi(argv[3]), true);
        return 0;
    }
    // the shapes of a batch-32 iteration: a 64 -> 128 @ 32^2 encoder level, the 45 -> 45 @ 64^2 refinement stack (N padded to 64),
    // a deep level (1024 channels on 128 pixels), and a large square for the ceiling
    run(32768, 128, 576, true);
    run(131072, 64, 448, true);
    run(128, 1024, 9216, true);
    run(8192, 1024, 4608, true);
    run(8192, 1024, 4608, false);
    return 0;
}

// What an IN-KERNEL grid-wide column reduction costs on the MI355X when all blocks of the launch are resident (1 024 blocks of
// 256 threads = 4 per CU): every block publishes its 2 x C partial sums as {tag | float} words with device-scope relaxed atomic
// stores (no fence), 2 x C "reducer" blocks each poll the `blocks` words of one column (device-scope atomic loads), sum them in a
// fixed order and publish the total the same way, and every block then polls the 2 x C totals.  That is the two-stage
// BatchNorm statistics (conv epilogue partials -> second-stage launch -> consumer launch) inside ONE launch.  Timed against the
// same kernel without the exchange; K "layers" back to back as in a replayed iteration.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/grid_reduce tools/microbench/grid_reduce.hip && /tmp/grid_reduce
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ unsigned long long wall() { return __builtin_readcyclecounter(); }

template <bool EXCHANGE>
__global__ void __launch_bounds__(256) layer(unsigned long long* part, unsigned long long* tot, float* out, int C2, unsigned tag,
                                             int spin_work, unsigned* err) {
    const int b = blockIdx.x, nb = gridDim.x, t = threadIdx.x;
    // stand-in for the K loop: some register work so that blocks do not all arrive in the same cycle
    float acc = (float)(b * 256 + t);
    for (int i = 0; i < spin_work; ++i) acc = __builtin_fmaf(acc, 1.0000001f, 0.5f);
    float total = 0.f;
    if (EXCHANGE) {
        if (t < C2)
            __hip_atomic_store(part + (size_t)b * C2 + t, ((unsigned long long)tag << 32) | __float_as_uint(acc * 1e-9f + (float)t),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (b < C2) {                   // reducer of column b
            __shared__ float sm[256];
            float s = 0.f;
            for (int r = t; r < nb; r += 256) {
                const unsigned long long* w = part + (size_t)r * C2 + b;
                unsigned long long v = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                unsigned long long t0 = wall();
                while ((unsigned)(v >> 32) != tag) {
                    if (wall() - t0 > 2000000000ull) { *err = 1; break; }
                    __builtin_amdgcn_s_sleep(1);
                    v = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                s += __uint_as_float((unsigned)v);
            }
            sm[t] = s;
            __syncthreads();
            for (int k = 128; k > 0; k >>= 1) {
                if (t < k) sm[t] += sm[t + k];
                __syncthreads();
            }
            if (t == 0)
                __hip_atomic_store(tot + b, ((unsigned long long)tag << 32) | __float_as_uint(sm[0]), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
        }
        if (t < C2) {
            const unsigned long long* w = tot + t;
            unsigned long long v = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned long long t0 = wall();
            while ((unsigned)(v >> 32) != tag) {
                if (wall() - t0 > 2000000000ull) { *err = 2; break; }
                __builtin_amdgcn_s_sleep(1);
                v = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            total = __uint_as_float((unsigned)v);
        }
    }
    out[(size_t)b * 256 + t] = acc + total;
}

// (b) the second stage alone inside the producer: the LAST `C2 / 2 / 4` blocks in dispatch order are the reducers (a wave per
// channel: both of its columns); nobody else waits, a reducer waits only for blocks that were dispatched before it (no deadlock
// whatever the grid size), the tag is a constant and a reducer clears the words it consumed (a captured launch replays with
// the same arguments).  out2[c] = totals, checked on the host.
__global__ void __launch_bounds__(256) layer_second_stage(unsigned long long* part, float* totals, float* out, int C2, int spin_work,
                                                          unsigned* err) {
    const int b = blockIdx.x, nb = gridDim.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    float acc = (float)(b * 256 + t);
    for (int i = 0; i < spin_work; ++i) acc = __builtin_fmaf(acc, 1.0000001f, 0.5f);
    if (t < C2)
        __hip_atomic_store(part + (size_t)b * C2 + t, (1ull << 32) | __float_as_uint((float)((b + t) & 7)), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    out[(size_t)b * 256 + t] = acc;
    // one reducer BLOCK per channel (both of its columns): 256 threads x 2 columns cover the rows, every thread issues all of its
    // loads before it looks at the first one (device-scope loads go to memory: ~1-2 us each, so they must be in flight together)
    const int C = C2 / 2, nred = C, first = nb - nred;
    if (b >= first) {
        __shared__ float sm[2][256];
        const int c = b - first;
        constexpr int U = 8;                       // rows per thread and trip
        float s[2] = {0.f, 0.f};
        for (int r0 = 0; r0 < nb; r0 += 256 * U) {
            unsigned long long v[2][U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int r = r0 + u * 256 + t;
#pragma unroll
                for (int h = 0; h < 2; ++h)
                    v[h][u] = r < nb ? __hip_atomic_load(part + (size_t)r * C2 + h * C + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                     : (1ull << 32);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int r = r0 + u * 256 + t;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    unsigned long long* w = part + (size_t)(r < nb ? r : 0) * C2 + h * C + c;
                    unsigned long long t0 = wall();
                    while ((unsigned)(v[h][u] >> 32) != 1u) {
                        if (wall() - t0 > 2000000000ull) { *err = 3; break; }
                        __builtin_amdgcn_s_sleep(1);
                        v[h][u] = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    s[h] += __uint_as_float((unsigned)v[h][u]);
                    if (r < nb) __hip_atomic_store(w, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
        sm[0][t] = s[0], sm[1][t] = s[1];
        __syncthreads();
        for (int k = 128; k > 0; k >>= 1) {
            if (t < k) sm[0][t] += sm[0][t + k], sm[1][t] += sm[1][t + k];
            __syncthreads();
        }
        if (t == 0) totals[c] = sm[0][0], totals[C + c] = sm[1][0];
    }
}

int main() {
    const int NB = 1024, K = 40;
    unsigned long long *part, *tot;
    float* out;
    unsigned* err;
    CHECK(hipMalloc(&part, (size_t)NB * 256 * 8));
    CHECK(hipMalloc(&tot, 256 * 8));
    CHECK(hipMalloc(&out, (size_t)NB * 256 * 4));
    CHECK(hipMalloc(&err, 4));
    CHECK(hipMemset(part, 0, (size_t)NB * 256 * 8));
    CHECK(hipMemset(tot, 0, 256 * 8));
    CHECK(hipMemset(err, 0, 4));
    hipStream_t s;
    CHECK(hipStreamCreate(&s));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    unsigned tag = 1;
    printf("%d blocks x 256 threads, %d launches back to back, us per launch\n", NB, K);
    printf("%8s %10s | %10s %10s %10s\n", "work", "columns", "plain", "exchange", "difference");
    for (int work : {0, 2000, 20000}) {
        for (int C2 : {96, 128, 256}) {
            float ms[2];
            for (int ex = 0; ex < 2; ++ex) {
                for (int rep = 0; rep < 2; ++rep) {
                    CHECK(hipEventRecord(a, s));
                    for (int k = 0; k < K; ++k) {
                        if (ex) hipLaunchKernelGGL(layer<true>, dim3(NB), dim3(256), 0, s, part, tot, out, C2, tag++, work, err);
                        else hipLaunchKernelGGL(layer<false>, dim3(NB), dim3(256), 0, s, part, tot, out, C2, tag++, work, err);
                    }
                    CHECK(hipEventRecord(b, s));
                    CHECK(hipEventSynchronize(b));
                    CHECK(hipEventElapsedTime(&ms[ex], a, b));
                }
            }
            printf("%8d %10d | %10.2f %10.2f %10.2f\n", work, C2, ms[0] * 1e3 / K, ms[1] * 1e3 / K, (ms[1] - ms[0]) * 1e3 / K);
        }
    }
    printf("(b) second stage inside the producer (reducers = the last blocks; constant tag, words cleared by the reducer)\n");
    printf("%8s %8s %10s | %10s %10s %10s\n", "blocks", "work", "columns", "plain", "in-kernel", "difference");
    float* totals;
    CHECK(hipMalloc(&totals, 1024 * 4));
    CHECK(hipMemset(part, 0, (size_t)NB * 256 * 8));
    for (int nb : {512, 1024, 2048, 4096}) {
        unsigned long long* part2;
        CHECK(hipMalloc(&part2, (size_t)nb * 256 * 8));
        CHECK(hipMemset(part2, 0, (size_t)nb * 256 * 8));
        float* out2;
        CHECK(hipMalloc(&out2, (size_t)nb * 256 * 4));
        for (int work : {2000}) {
            for (int C2 : {96, 256}) {
                float ms[2];
                for (int ex = 0; ex < 2; ++ex)
                    for (int rep = 0; rep < 2; ++rep) {
                        CHECK(hipEventRecord(a, s));
                        for (int k = 0; k < K; ++k) {
                            if (ex) hipLaunchKernelGGL(layer_second_stage, dim3(nb), dim3(256), 0, s, part2, totals, out2, C2, work, err);
                            else hipLaunchKernelGGL(layer<false>, dim3(nb), dim3(256), 0, s, part2, tot, out2, C2, 0u, work, err);
                        }
                        CHECK(hipEventRecord(b, s));
                        CHECK(hipEventSynchronize(b));
                        CHECK(hipEventElapsedTime(&ms[ex], a, b));
                    }
                float ht[256];
                CHECK(hipMemcpy(ht, totals, C2 * 4, hipMemcpyDeviceToHost));
                int bad = 0;
                for (int col = 0; col < C2; ++col) {
                    double want = 0;
                    for (int r = 0; r < nb; ++r) want += (double)((r + col) & 7);
                    if ((double)ht[col] != want) ++bad;
                }
                printf("%8d %8d %10d | %10.2f %10.2f %10.2f   wrong totals: %d\n", nb, work, C2, ms[0] * 1e3 / K, ms[1] * 1e3 / K,
                       (ms[1] - ms[0]) * 1e3 / K, bad);
            }
        }
        CHECK(hipFree(part2));
        CHECK(hipFree(out2));
    }
    unsigned h = 0;
    CHECK(hipMemcpy(&h, err, 4, hipMemcpyDeviceToHost));
    printf("timeouts: %u\n", h);
    return 0;
}

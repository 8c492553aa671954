// HBM read bandwidth of the access patterns of the two big streaming kernels (weight-gradient reduction, Adam): every thread
// issues K independent 16-byte loads, one from each of K planes that lie `plane` bytes apart (K = 1: a plain streaming read),
// and adds them up.  Total bytes per launch are held at ~512 MB (larger than the 256 MB Infinity Cache).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/stream_read tools/microbench/stream_read.hip && /tmp/stream_read
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int K>
__global__ void __launch_bounds__(256) read_k(const float* __restrict__ src, long plane_floats, long per_plane_vec4, float* out,
                                              int iters_per_thread) {
    float4 acc = make_float4(0, 0, 0, 0);
    long i = (long)blockIdx.x * 256 * iters_per_thread + threadIdx.x;
    for (int it = 0; it < iters_per_thread; ++it, i += 256) {
        if (i >= per_plane_vec4) break;
        float4 v[K];
#pragma unroll
        for (int k = 0; k < K; ++k) v[k] = *reinterpret_cast<const float4*>(src + k * plane_floats + i * 4);
#pragma unroll
        for (int k = 0; k < K; ++k) acc.x += v[k].x, acc.y += v[k].y, acc.z += v[k].z, acc.w += v[k].w;
    }
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = acc.x;
}

template <int K>
float run(const float* src, long total_bytes, long pad_bytes, float* out, int ipt) {
    const long plane_bytes = total_bytes / K / 16 * 16 + pad_bytes;
    const long per_plane_vec4 = (total_bytes / K) / 16;
    const int blocks = (int)((per_plane_vec4 + 256L * ipt - 1) / (256L * ipt));
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(read_k<K>, dim3(blocks), dim3(256), 0, 0, src, plane_bytes / 4, per_plane_vec4, out, ipt);
    hipEventRecord(e0, 0);
    const int reps = 10;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(read_k<K>, dim3(blocks), dim3(256), 0, 0, src, plane_bytes / 4, per_plane_vec4, out, ipt);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return (float)((double)per_plane_vec4 * 16 * K * reps / (ms * 1e-3) / 1e12);
}

int main() {
    const long total = 512L << 20;
    float *src, *out;
    hipMalloc(&src, total + (64 << 20));
    hipMalloc(&out, 64);
    hipMemset(src, 0, total + (64 << 20));
    printf("TB/s read, 512 MB per launch; rows: loads per thread before it exits (1 = the reduction's flat map)\n");
    printf("%-28s %8s %8s %8s %8s\n", "planes (stride)", "ipt=1", "ipt=4", "ipt=16", "ipt=64");
    const int ipts[4] = {1, 4, 16, 64};
    for (int pad = 0; pad < 2; ++pad) {
        const long pb = pad ? 4096 + 256 : 0;
        float r[4][4];
        for (int j = 0; j < 4; ++j) {
            r[0][j] = run<1>(src, total, pb, out, ipts[j]);
            r[1][j] = run<4>(src, total, pb, out, ipts[j]);
            r[2][j] = run<9>(src, total, pb, out, ipts[j]);
            r[3][j] = run<16>(src, total, pb, out, ipts[j]);
        }
        const char* names[4] = {"1", "4", "9", "16"};
        for (int k = 0; k < 4; ++k)
            printf("%-3s planes, stride %-12s %8.2f %8.2f %8.2f %8.2f\n", names[k], pad ? "odd (+4352 B)" : "512MB/K", r[k][0], r[k][1],
                   r[k][2], r[k][3]);
    }
    return 0;
}

// What a kernel boundary costs on this chip, and whether two kernels of ONE stream can overlap.
//   spin_kernel: `blocks` workgroups that stamp the 100 MHz wall clock at entry and exit and spin `us` microseconds in between.
//   (a) A -> B as plain stream launches (barrier bit): gap = first block start of B - last block end of A
//   (b) the same through a captured hipGraph (what the training iteration replays)
//   (c) B launched with hipExtAnyOrderLaunch: does B start before A ends?  (the header says "not supported on GFX9xx")
//   (d) A and B on two streams: overlap, and what the cross-stream event costs (C on stream 1 waits for an event behind A)
//   (e) the LDS occupancy shaping: 1024 blocks of 20 KB vs 40 KB LDS -- blocks per CU as seen from the XCC / CU id of each block
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/launch_gap tools/microbench/launch_gap.hip && /tmp/launch_gap
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

struct Stamp {
    unsigned long long t0, t1;
    unsigned hw;       // HW_ID register: CU / SE / XCC position of the block
    unsigned pad;
};

__global__ void __launch_bounds__(256) spin_kernel(Stamp* log, int us) {
    extern __shared__ float dyn[];
    const unsigned long long t0 = wall_clock64();
    unsigned long long t = t0;
    while (t - t0 < (unsigned long long)us * 100ull) t = wall_clock64();
    if (threadIdx.x == 0) {
        unsigned hw = 0, xcc = 0;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        log[blockIdx.x] = Stamp{t0, (unsigned long long)wall_clock64(), hw, xcc};
    }
    if (us < 0) dyn[threadIdx.x] = 0.f;
}

static void stats(const std::vector<Stamp>& a, int n, unsigned long long* first, unsigned long long* last_end,
                  unsigned long long* last_start) {
    *first = ~0ull, *last_end = 0, *last_start = 0;
    for (int i = 0; i < n; ++i) {
        *first = std::min(*first, a[i].t0);
        *last_end = std::max(*last_end, a[i].t1);
        *last_start = std::max(*last_start, a[i].t0);
    }
}

int main() {
    const int NB = 1024;
    Stamp *la, *lb, *lc;
    hipMalloc(&la, sizeof(Stamp) * NB), hipMalloc(&lb, sizeof(Stamp) * NB), hipMalloc(&lc, sizeof(Stamp) * NB);
    std::vector<Stamp> ha(NB), hb(NB), hc(NB);
    hipStream_t s0, s1;
    hipStreamCreateWithFlags(&s0, hipStreamNonBlocking), hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
    auto fetch = [&]() {
        hipDeviceSynchronize();
        hipMemcpy(ha.data(), la, sizeof(Stamp) * NB, hipMemcpyDeviceToHost);
        hipMemcpy(hb.data(), lb, sizeof(Stamp) * NB, hipMemcpyDeviceToHost);
        hipMemcpy(hc.data(), lc, sizeof(Stamp) * NB, hipMemcpyDeviceToHost);
    };
    unsigned long long f, e, ls, f2, e2, ls2, f3, e3, ls3;
    // warm-up
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(spin_kernel, dim3(NB), dim3(256), 0, s0, la, 5);
    hipDeviceSynchronize();

    for (int blocks : {64, 1024}) {
        for (int rep = 0; rep < 3; ++rep) {
            hipLaunchKernelGGL(spin_kernel, dim3(blocks), dim3(256), 0, s0, la, 20);
            hipLaunchKernelGGL(spin_kernel, dim3(blocks), dim3(256), 0, s0, lb, 20);
            hipLaunchKernelGGL(spin_kernel, dim3(blocks), dim3(256), 0, s0, lc, 20);
            fetch();
            stats(ha, blocks, &f, &e, &ls), stats(hb, blocks, &f2, &e2, &ls2), stats(hc, blocks, &f3, &e3, &ls3);
            printf("(a) stream launches, %4d blocks x 20 us: A spans %.2f us (starts within %.2f); gap A end -> B first start %.2f us, "
                   "B -> C %.2f us\n", blocks, (e - f) * 0.01, (ls - f) * 0.01, ((double)f2 - (double)e) * 0.01,
                   ((double)f3 - (double)e2) * 0.01);
        }
    }
    // (b) hipGraph
    {
        hipGraph_t g;
        hipGraphExec_t ge;
        hipStreamBeginCapture(s0, hipStreamCaptureModeGlobal);
        hipLaunchKernelGGL(spin_kernel, dim3(NB), dim3(256), 0, s0, la, 20);
        hipLaunchKernelGGL(spin_kernel, dim3(NB), dim3(256), 0, s0, lb, 20);
        hipLaunchKernelGGL(spin_kernel, dim3(NB), dim3(256), 0, s0, lc, 20);
        hipStreamEndCapture(s0, &g);
        hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        for (int rep = 0; rep < 3; ++rep) {
            hipGraphLaunch(ge, s0);
            fetch();
            stats(ha, NB, &f, &e, &ls), stats(hb, NB, &f2, &e2, &ls2), stats(hc, NB, &f3, &e3, &ls3);
            printf("(b) hipGraph replay,  1024 blocks x 20 us: gap A end -> B first start %.2f us, B -> C %.2f us\n",
                   ((double)f2 - (double)e) * 0.01, ((double)f3 - (double)e2) * 0.01);
        }
    }
    // (c) any-order launch of B
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(spin_kernel, dim3(256), dim3(256), 0, s0, la, 40);
        hipExtLaunchKernelGGL(spin_kernel, dim3(256), dim3(256), 0, s0, nullptr, nullptr, hipExtAnyOrderLaunch, lb, 40);
        hipError_t err = hipGetLastError();
        hipLaunchKernelGGL(spin_kernel, dim3(256), dim3(256), 0, s0, lc, 5);
        fetch();
        stats(ha, 256, &f, &e, &ls), stats(hb, 256, &f2, &e2, &ls2), stats(hc, 256, &f3, &e3, &ls3);
        printf("(c) hipExtAnyOrderLaunch (%s): B first start - A first start = %.2f us (A runs 40 us: < 40 = overlap); C starts %.2f us "
               "after max(A, B) end\n", hipGetErrorString(err), ((double)f2 - (double)f) * 0.01,
               ((double)f3 - (double)std::max(e, e2)) * 0.01);
    }
    // (d) two streams + event
    {
        hipEvent_t ev;
        hipEventCreateWithFlags(&ev, hipEventDisableTiming);
        for (int rep = 0; rep < 3; ++rep) {
            hipLaunchKernelGGL(spin_kernel, dim3(256), dim3(256), 0, s0, la, 40);
            hipEventRecord(ev, s0);
            hipLaunchKernelGGL(spin_kernel, dim3(256), dim3(256), 0, s1, lb, 40);
            hipStreamWaitEvent(s1, ev, 0);
            hipLaunchKernelGGL(spin_kernel, dim3(256), dim3(256), 0, s1, lc, 5);
            fetch();
            stats(ha, 256, &f, &e, &ls), stats(hb, 256, &f2, &e2, &ls2), stats(hc, 256, &f3, &e3, &ls3);
            printf("(d) two streams: B (stream 1) first start - A (stream 0) first start = %.2f us; C (stream 1, waits for the event "
                   "behind A and for B) starts %.2f us after max(A, B) end\n", ((double)f2 - (double)f) * 0.01,
                   ((double)f3 - (double)std::max(e, e2)) * 0.01);
        }
    }
    // (e) blocks per CU with and without LDS padding
    for (int lds : {0, 20480, 39936, 53248}) {
        hipLaunchKernelGGL(spin_kernel, dim3(1024), dim3(256), lds, s0, la, 20);
        fetch();
        // (xcc, se, cu) -> count; HW_ID gfx9: CU_ID bits 11:8, SH_ID bit 12, SE_ID bits 15:13
        std::vector<int> count(8 * 8 * 2 * 16, 0);
        for (int i = 0; i < 1024; ++i) {
            const unsigned hw = ha[i].hw, xcc = ha[i].pad & 0xf;
            const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
            count[((xcc * 8 + se) * 2 + sh) * 16 + cu]++;
        }
        int hist[16] = {0}, used = 0, mx = 0;
        for (int c : count)
            if (c) {
                hist[c < 15 ? c : 15]++;
                used++;
                mx = std::max(mx, c);
            }
        stats(ha, 1024, &f, &e, &ls);
        printf("(e) 1024 blocks, %5d B of dynamic LDS: %d CU positions used, max %d blocks on one; histogram (blocks/CU: CUs):", lds,
               used, mx);
        for (int k = 1; k < 16; ++k)
            if (hist[k]) printf(" %d:%d", k, hist[k]);
        printf("; last start %.2f us after the first, span %.2f us\n", (ls - f) * 0.01, (e - f) * 0.01);
    }
    return 0;
}

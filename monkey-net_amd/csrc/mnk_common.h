// Shared host/device helpers for the gfx950 kernels of libmonkeynet_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include "monkeynet_hip.h"

// keep a wave-uniform pointer in scalar registers across a loop (opaque to rematerialisation); no-op on the emulator
#ifdef HIPEMU
#define MNK_KEEP_SGPR(p) ((void)0)
#else
#define MNK_KEEP_SGPR(p) asm volatile("" : "+s"(p))
#endif

// instruction-scheduling hint (no-op on the emulator): the next `n` instructions of class `mask` form a group, in order
#ifdef HIPEMU
#define MNK_SCHED_GROUP(mask, n) ((void)0)
#else
#define MNK_SCHED_GROUP(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
#endif

// wait until every vector-memory load of this wave has returned (s_waitcnt vmcnt(0); no-op on the emulator).  Used in front of
// a software-pipelined loop: with nothing pending at the loop entry the compiler's wait-count bookkeeping at the loop header
// is the back edge's alone, so the waits inside the loop are exact (merged with a prologue that issued its loads in another
// order they come out pessimistic: `vmcnt(1)` where `vmcnt(3)` would do -- the loop then waits for loads it just issued)
#ifdef HIPEMU
#define MNK_WAIT_VMEM() ((void)0)
#else
#define MNK_WAIT_VMEM() __builtin_amdgcn_s_waitcnt(0x0F70)
#endif

namespace mnk {

void set_error(const char* fmt, ...);
// a named tuning value with its measured default (runtime.hip): settable through mnk_set_tuning / MNK_TUNING, no env switch of its own
int tuning_knob(const char* name, int* slot, int dflt);

// kernel ids of the profiling recorder (mnk_prof_*)
enum KernelId {
    K_CONV_FWD = 0,   // conv3x3 implicit GEMM (forward and dgrad share the kernel)
    K_CONV_WGRAD,
    K_CONV_REDUCE,    // split-K reductions + weight packing
    K_BN_STATS,
    K_BN_APPLY,
    K_BN_BWD,
    K_LAYOUT,
    K_KEYPOINT,
    K_EMBED,
    K_FIELD,
    K_DEFORM,
    K_CONV1X1,
    K_OPTIM,          // multi-tensor Adam (+ weight re-pack)
    K_LOSS,           // loss reductions
    K_NUM
};

// RAII scope: when profiling is on, brackets the launches issued inside it with two HIP events on `stream`.
struct ProfScope {
    // work: ALGORITHMIC FLOPs (MFMA kernels: those of the reference's convolution, whatever form computes it) or bytes;
    // executed: the multiply-adds the launch actually issues x 2 (sub-pixel forms of an up-sampled 3x3 convolution: 4/9 of the
    // algorithmic figure); < 0: the same as `work`
    ProfScope(int kid, hipStream_t stream, double work, double executed = -1.0);
    ~ProfScope();
    // For a scope that holds exactly ONE launch: the two events to hand to hipExtLaunchKernelGGL, which stamps them with
    // the kernel's own begin / end (what rocprofv3 reports) instead of the stream positions around the launch, which
    // also span the dispatch gap.  Returns false (and leaves the events null) when profiling is off.
    bool kernel_events(hipEvent_t* start, hipEvent_t* stop);
    int kid;
    hipStream_t stream;
    int slot;
    bool ext;
};

static inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }
static inline int round_up(int a, int b) { return (a + b - 1) / b * b; }

}  // namespace mnk

#define MNK_REQUIRE(cond)                                                        \
    do {                                                                         \
        if (!(cond)) {                                                           \
            mnk::set_error("%s: invalid argument: %s", __func__, #cond);         \
            return MNK_EINVAL;                                                   \
        }                                                                        \
    } while (0)

#define MNK_LAUNCH_CHECK()                                                       \
    do {                                                                         \
        hipError_t e__ = hipGetLastError();                                      \
        if (e__ != hipSuccess) {                                                 \
            mnk::set_error("%s: launch failed: %s", __func__, hipGetErrorString(e__)); \
            return MNK_ELAUNCH;                                                  \
        }                                                                        \
    } while (0)

// ---- device helpers -------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
// sum over the 256 threads of a block; every thread gets the result.  `red` = 4 floats of LDS.
__device__ __forceinline__ float block_sum_256(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}
__device__ __forceinline__ float block_max_256(float v, float* red) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

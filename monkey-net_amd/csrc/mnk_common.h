// Shared host/device helpers for the gfx950 kernels of libmonkeynet_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <string.h>

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

// ---- fp32-accurate products on the bf16 matrix cores (round 6) ---------------------------------------------------------------
// v_mfma_f32_32x32x16_bf16 issues sixteen times the multiply-adds of v_mfma_f32_32x32x2_f32 per cycle.  An fp32 value splits
// EXACTLY into three bf16 terms, x = x1 + x2 + x3 (x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2): the subtractions are
// exact, 3 x 8 mantissa bits cover the 24 of an fp32), every bf16 x bf16 product is exact in fp32, and of the nine cross products
// of two split operands the three with weight <= 2^-24 of the leading one are below the rounding of an fp32 product: six MFMAs
// (a1b1, a1b2, a2b1, a1b3, a3b1, a2b2; fp32 accumulation inside the matrix core) give the fp32 GEMM's result to ~2 ulp per
// product -- measured: the same error against fp64 as the fp32 MFMA chain (tools/microbench/bf16x3_gemm.hip,
// tests/test_kernels_conv.py) -- at 2.67 x the matrix rate.  The split is made by the loaders between the global load and the LDS
// store with the hardware conversion (v_cvt_pk_bf16_f32, round to nearest even): 4.5 vector instructions per element.
#ifdef HIPEMU
typedef unsigned short mnk_bf16x8 __attribute__((vector_size(16)));
static inline unsigned mnk_bf16_rn_bits(float x) {       // round to nearest even, NaN stays NaN (what v_cvt_pk_bf16_f32 does)
    unsigned u;
    memcpy(&u, &x, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
    u += 0x7fffu + ((u >> 16) & 1u);
    return u >> 16;
}
static inline float mnk_bf16_bits_f(unsigned h) {
    const unsigned u = h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
// float4 -> three planes of four bf16 (two packed dwords each)
static inline void mnk_split3(float4 v, uint2& p0, uint2& p1, uint2& p2) {
    float x[4] = {v.x, v.y, v.z, v.w};
    unsigned h[3][4];
    for (int e = 0; e < 4; ++e) {
        h[0][e] = mnk_bf16_rn_bits(x[e]);
        const float r = x[e] - mnk_bf16_bits_f(h[0][e]);
        h[1][e] = mnk_bf16_rn_bits(r);
        h[2][e] = mnk_bf16_rn_bits(r - mnk_bf16_bits_f(h[1][e]));
    }
    p0 = make_uint2(h[0][0] | (h[0][1] << 16), h[0][2] | (h[0][3] << 16));
    p1 = make_uint2(h[1][0] | (h[1][1] << 16), h[1][2] | (h[1][3] << 16));
    p2 = make_uint2(h[2][0] | (h[2][1] << 16), h[2][2] | (h[2][3] << 16));
}
static inline mnk_bf16x8 mnk_as_bf16x8(uint4 v) {
    mnk_bf16x8 r;
    memcpy(&r, &v, 16);
    return r;
}
#else
typedef __attribute__((ext_vector_type(8))) __bf16 mnk_bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 mnk_bf16x2;
typedef __attribute__((ext_vector_type(2))) float mnk_f32x2;
__device__ __forceinline__ void mnk_split3(float4 v, uint2& p0, uint2& p1, uint2& p2) {
    mnk_f32x2 lo = {v.x, v.y}, hi = {v.z, v.w};
    mnk_bf16x2 a = __builtin_convertvector(lo, mnk_bf16x2), b = __builtin_convertvector(hi, mnk_bf16x2);
    p0 = make_uint2(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b));
    lo = lo - __builtin_convertvector(a, mnk_f32x2);
    hi = hi - __builtin_convertvector(b, mnk_f32x2);
    a = __builtin_convertvector(lo, mnk_bf16x2), b = __builtin_convertvector(hi, mnk_bf16x2);
    p1 = make_uint2(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b));
    lo = lo - __builtin_convertvector(a, mnk_f32x2);
    hi = hi - __builtin_convertvector(b, mnk_f32x2);
    a = __builtin_convertvector(lo, mnk_bf16x2), b = __builtin_convertvector(hi, mnk_bf16x2);
    p2 = make_uint2(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b));
}
__device__ __forceinline__ mnk_bf16x8 mnk_as_bf16x8(uint4 v) { return __builtin_bit_cast(mnk_bf16x8, v); }
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

// hipemu: a tiny single-threaded, fiber-based stand-in for <hip/hip_runtime.h>.  TEST INFRASTRUCTURE ONLY.
//
// The authoring container has no GPU, and GPU minutes are scarce, so the *unmodified* kernel sources under
// monkey-net_amd/csrc are also compiled with g++ against this header (tests/hipemu/build.sh puts
// tests/hipemu/include in front of the include path) into tests/hipemu/build/libmnk_emu.so.  The CPU-side
// unit tests (-m "not gpu") drive that library through the same C-ABI to check indexing, tiling, MFMA
// fragment layouts, LDS staging and barrier placement before a kernel ever reaches an MI355X.
//
// Execution model: the blocks of a launch run one after another; the threads of a block are ucontext fibers
// scheduled round-robin on one OS thread.  __syncthreads() and every wave-level primitive (__shfl*, MFMA)
// are rendez-vous points between fibers; MFMA results follow the documented gfx950 fragment layouts
// (cdna_hip_programming.md section 3).  Nothing in the product (monkey-net_amd/) loads the emulator library:
// the shipped path links the real HIP runtime only and fails loudly without it.
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <functional>

#define HIPEMU 1
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static
#define ext_vector_type(n) vector_size(4 * (n))

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
struct int2 { int x, y; };
struct int4 { int x, y, z, w; };
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline int4 make_int4(int x, int y, int z, int w) { return int4{x, y, z, w}; }
struct uint2 { unsigned x, y; };
struct alignas(16) uint4 { unsigned x, y, z, w; };
static inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }

typedef void* hipStream_t;
typedef struct hipemu_event* hipEvent_t;
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorLaunchFailure = 719 };

namespace hipemu {
extern dim3 g_threadIdx, g_blockIdx, g_blockDim, g_gridDim;
void launch(dim3 grid, dim3 block, const std::function<void()>& body);
void sync_block();
void sync_wave();
unsigned lane_id();
// exchange one 32-bit value per lane inside the calling wave; returns pointer to the 64 published values
const uint32_t* wave_publish(uint32_t v, int slot);
}  // namespace hipemu

#define threadIdx (hipemu::g_threadIdx)
#define blockIdx (hipemu::g_blockIdx)
#define blockDim (hipemu::g_blockDim)
#define gridDim (hipemu::g_gridDim)
#define warpSize 64

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    hipemu::launch((grid), (block), [&]() { kernel(__VA_ARGS__); })
#define HIP_KERNEL_NAME(...) __VA_ARGS__

static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipPeekAtLastError() { return hipSuccess; }
static inline const char* hipGetErrorString(hipError_t) { return "hipemu"; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
static inline hipError_t hipMemset2DAsync(void* p, size_t pitch, int v, size_t w, size_t h, hipStream_t) { for (size_t r = 0; r < h; ++r) memset((char*)p + r * pitch, v, w); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
enum { hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
// hipExtLaunchKernelGGL (hip_ext.h): the launch with per-kernel start / stop events -- events are stubs here
#define hipExtLaunchKernelGGL(kernel, grid, block, shmem, stream, start_ev, stop_ev, flags, ...) \
    hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__)
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = nullptr; return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }

// ---- what the peer-to-peer SyncBN exchange (csrc/p2p.hip, the *_sync kernels of batchnorm.hip) uses -----------------------------
// "Device memory" is host memory here.  Memory from hipExtMallocWithFlags (the exchange's mailboxes) lives in a named POSIX
// shared-memory object, hipIpcGetMemHandle hands its name out and hipIpcOpenMemHandle maps it: the PROCESSES of a gloo test on
// the CPU exchange through each other's mailboxes exactly as the processes of a data-parallel run do through IPC-mapped HBM
// (hipemu.cpp).  The atomic stores / loads of the protocol are real atomics (the polling loops read memory another process
// writes); the kernels that carry an exchange therefore run -- protocol, slots, sequence numbers, rank-ordered sums -- in the
// CPU suite, with one rank and with several.
enum { hipDeviceMallocUncached = 3, hipDeviceMallocFinegrained = 1, hipMemcpyDeviceToHost = 2, hipMemcpyHostToDevice = 1,
       hipIpcMemLazyEnablePeerAccess = 1 };
struct hipIpcMemHandle_t { char reserved[64]; };
hipError_t hipExtMallocWithFlags(void** p, size_t n, unsigned flags);
hipError_t hipFree(void* p);
hipError_t hipIpcGetMemHandle(hipIpcMemHandle_t* h, void* p);
hipError_t hipIpcOpenMemHandle(void** p, hipIpcMemHandle_t h, unsigned flags);
hipError_t hipIpcCloseMemHandle(void* p);
static inline hipError_t hipMalloc(void** p, size_t n) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorInvalidValue; }
static inline hipError_t hipMemset(void* p, int v, size_t n) { memset(p, v, n); return hipSuccess; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, int) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __HIP_MEMORY_SCOPE_SYSTEM 5
template <typename T, typename V>
static inline void __hip_atomic_store(T* p, V v, int, int) { __atomic_store_n(p, (T)v, __ATOMIC_SEQ_CST); }
template <typename T>
static inline T __hip_atomic_load(const T* p, int, int) { return __atomic_load_n(p, __ATOMIC_SEQ_CST); }
void hipemu_yield();                              // a polling wavefront lets the peer PROCESS run
static inline void __builtin_amdgcn_s_sleep(int) { hipemu_yield(); }
static inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
static inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
static inline long long wall_clock64() {          // 100 MHz, like the device's constant-rate counter
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (long long)ts.tv_sec * 100000000ll + ts.tv_nsec / 10;
}

static inline void __syncthreads() { hipemu::sync_block(); }

static inline uint32_t hipemu_shfl32(uint32_t u, int src, int width) {
    unsigned l = hipemu::lane_id();
    const uint32_t* all = hipemu::wave_publish(u, 0);
    unsigned base = l & ~(unsigned)(width - 1);
    return all[base + ((unsigned)src & (unsigned)(width - 1))];
}
template <typename T>
static inline T __shfl(T v, int src, int width = 64) {
    static_assert(sizeof(T) == 4 || sizeof(T) == 8, "hipemu shuffles are 32- or 64-bit (two halves, as the hardware does)");
    uint32_t u[sizeof(T) / 4], r[sizeof(T) / 4];
    memcpy(u, &v, sizeof(T));
    for (unsigned i = 0; i < sizeof(T) / 4; ++i) r[i] = hipemu_shfl32(u[i], src, width);
    T out;
    memcpy(&out, r, sizeof(T));
    return out;
}
template <typename T>
static inline T __shfl_xor(T v, int mask, int width = 64) {
    return __shfl(v, (int)((hipemu::lane_id() & (unsigned)(width - 1)) ^ (unsigned)mask), width);
}
template <typename T>
static inline T __shfl_down(T v, unsigned delta, int width = 64) {
    unsigned l = hipemu::lane_id() & (unsigned)(width - 1);
    unsigned s = l + delta;
    return __shfl(v, (int)(s < (unsigned)width ? s : l), width);
}

// wave64 vote + the "set bits below my lane" pair (v_mbcnt_lo / v_mbcnt_hi)
static inline unsigned long long __ballot(int pred) {
    const uint32_t* all = hipemu::wave_publish(pred ? 1u : 0u, 0);
    unsigned long long m = 0;
    for (int i = 0; i < 64; ++i) m |= (unsigned long long)(all[i] & 1u) << i;
    return m;
}
static inline int __popcll(unsigned long long m) { return __builtin_popcountll(m); }
static inline unsigned __builtin_amdgcn_mbcnt_lo(unsigned mask, unsigned add) {
    const unsigned l = hipemu::lane_id();
    return add + (unsigned)__builtin_popcount(l >= 32 ? mask : (mask & ((1u << l) - 1u)));
}
static inline unsigned __builtin_amdgcn_mbcnt_hi(unsigned mask, unsigned add) {
    const unsigned l = hipemu::lane_id();
    return add + (l > 32 ? (unsigned)__builtin_popcount(mask & ((1u << (l - 32)) - 1u)) : 0u);
}

static inline float atomicAdd(float* p, float v) { float o = *p; *p = o + v; return o; }
static inline int atomicAdd(int* p, int v) { int o = *p; *p = o + v; return o; }
static inline unsigned atomicAdd(unsigned* p, unsigned v) { unsigned o = *p; *p = o + v; return o; }

#define __expf(x) expf(x)
static inline float __fdividef(float a, float b) { return a / b; }
static inline float __frcp_rn(float a) { return 1.0f / a; }
static inline float rsqrtf(float a) { return 1.0f / sqrtf(a); }
static inline float __fmaf_rn(float a, float b, float c) { return fmaf(a, b, c); }
static inline float __fmul_rn(float a, float b) { volatile float r = a * b; return r; }
static inline float __fadd_rn(float a, float b) { volatile float r = a + b; return r; }
static inline int __float2int_rd(float a) { return (int)floorf(a); }
static inline int __builtin_amdgcn_sbfe(int v, unsigned off, unsigned width) {   // signed bit-field extract
    off &= 31u;
    if (width == 0) return 0;
    if (off + width > 32) width = 32 - off;
    return (int)((unsigned)v << (32 - off - width)) >> (32 - width);
}
static inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }
static inline void __builtin_amdgcn_s_setprio(int) {}
static inline void __builtin_amdgcn_sched_barrier(int) {}
static inline void __builtin_amdgcn_s_barrier() { hipemu::sync_block(); }
static inline int __builtin_amdgcn_readfirstlane(int v) { return __shfl(v, 0); }

// raw buffer loads: resource = (base, num_records in bytes); an access that does not lie inside [0, num_records) reads 0
struct __amdgpu_buffer_rsrc_t {
    const char* base;
    unsigned num_records;
};
static inline __amdgpu_buffer_rsrc_t __builtin_amdgcn_make_buffer_rsrc(void* p, short, int num_records, int) {
    return __amdgpu_buffer_rsrc_t{(const char*)p, (unsigned)num_records};
}
typedef int hipemu_i32x4 __attribute__((vector_size(16)));
static inline hipemu_i32x4 __builtin_amdgcn_raw_buffer_load_b128(__amdgpu_buffer_rsrc_t r, unsigned voffset, unsigned soffset,
                                                                 int) {
    hipemu_i32x4 v = {0, 0, 0, 0};
    const unsigned long long off = (unsigned long long)voffset + soffset;
    if (off + 16 <= r.num_records) memcpy(&v, r.base + off, 16);
    return v;
}

typedef float hipemu_f32x16 __attribute__((vector_size(64)));
typedef float hipemu_f32x4 __attribute__((vector_size(16)));

// v_mfma_f32_32x32x2_f32: A[i][k] from lane i+32k, B[k][j] from lane j+32k,
// D[row][col]: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5); k-ordered fmaf chain.
static inline hipemu_f32x16 __builtin_amdgcn_mfma_f32_32x32x2f32(float a, float b, hipemu_f32x16 c, int, int, int) {
    uint32_t ua, ub;
    memcpy(&ua, &a, 4);
    memcpy(&ub, &b, 4);
    unsigned l = hipemu::lane_id();
    const uint32_t* pa = hipemu::wave_publish(ua, 1);
    const uint32_t* pb = hipemu::wave_publish(ub, 2);
    const float* fa = reinterpret_cast<const float*>(pa);
    const float* fb = reinterpret_cast<const float*>(pb);
    // wave_publish for slot 2 synchronised the wave after both values were written
    for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * (int)(l >> 5);
        int col = (int)(l & 31);
        float acc = c[r];
        for (int k = 0; k < 2; ++k) acc = fmaf(fa[row + 32 * k], fb[col + 32 * k], acc);
        c[r] = acc;
    }
    return c;
}
// v_mfma_f32_32x32x16_bf16 (gfx950): lane l holds A[i = l&31][k = 8*(l>>5) .. +7] and B[k = 8*(l>>5) .. +7][j = l&31] as eight
// bf16 (four dwords); D as the 32x32x2 form.  Every bf16 x bf16 product is exact in fp32; the matrix core adds the sixteen
// products of an output element and the accumulator in fp32 -- modelled as a k-ordered fmaf chain (the hardware's internal
// order is not documented; the kernels' tests compare against fp64 with tolerances, not bit for bit).
typedef unsigned short hipemu_u16x8 __attribute__((vector_size(16)));
static inline hipemu_f32x16 __builtin_amdgcn_mfma_f32_32x32x16_bf16(hipemu_u16x8 a, hipemu_u16x8 b, hipemu_f32x16 c, int, int, int) {
    uint32_t wa[4], wb[4];
    memcpy(wa, &a, 16);
    memcpy(wb, &b, 16);
    unsigned l = hipemu::lane_id();
    const uint32_t* pa[4];
    const uint32_t* pb[4];
    for (int d = 0; d < 4; ++d) pa[d] = hipemu::wave_publish(wa[d], 3 + d);
    for (int d = 0; d < 4; ++d) pb[d] = hipemu::wave_publish(wb[d], 7 + d);
    auto elem = [](const uint32_t* const* p, int lane, int e) {      // bf16 e (0..7) of a lane as float
        const uint32_t w = p[e >> 1][lane];
        const uint32_t u = (e & 1) ? (w & 0xffff0000u) : (w << 16);
        float f;
        memcpy(&f, &u, 4);
        return f;
    };
    for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * (int)(l >> 5);
        int col = (int)(l & 31);
        float acc = c[r];
        for (int k = 0; k < 16; ++k) acc = fmaf(elem(pa, row + 32 * (k >> 3), k & 7), elem(pb, col + 32 * (k >> 3), k & 7), acc);
        c[r] = acc;
    }
    return c;
}
// v_mfma_f32_16x16x32_bf16 (gfx950): lane l holds A[i = l&15][k = 8*(l>>4) .. +7], B[k = 8*(l>>4) .. +7][j = l&15]; D as 16x16x4
static inline hipemu_f32x4 __builtin_amdgcn_mfma_f32_16x16x32_bf16(hipemu_u16x8 a, hipemu_u16x8 b, hipemu_f32x4 c, int, int, int) {
    uint32_t wa[4], wb[4];
    memcpy(wa, &a, 16);
    memcpy(wb, &b, 16);
    unsigned l = hipemu::lane_id();
    const uint32_t* pa[4];
    const uint32_t* pb[4];
    for (int d = 0; d < 4; ++d) pa[d] = hipemu::wave_publish(wa[d], 3 + d);
    for (int d = 0; d < 4; ++d) pb[d] = hipemu::wave_publish(wb[d], 7 + d);
    auto elem = [](const uint32_t* const* p, int lane, int e) {
        const uint32_t w = p[e >> 1][lane];
        const uint32_t u = (e & 1) ? (w & 0xffff0000u) : (w << 16);
        float f;
        memcpy(&f, &u, 4);
        return f;
    };
    for (int r = 0; r < 4; ++r) {
        int row = 4 * (int)(l >> 4) + r;
        int col = (int)(l & 15);
        float acc = c[r];
        for (int k = 0; k < 32; ++k) acc = fmaf(elem(pa, row + 16 * (k >> 3), k & 7), elem(pb, col + 16 * (k >> 3), k & 7), acc);
        c[r] = acc;
    }
    return c;
}
// v_mfma_f32_16x16x4_f32: A[i][k] from lane i+16k, B[k][j] from lane j+16k,
// D: col = lane&15, row = 4*(lane>>4) + reg.
static inline hipemu_f32x4 __builtin_amdgcn_mfma_f32_16x16x4f32(float a, float b, hipemu_f32x4 c, int, int, int) {
    uint32_t ua, ub;
    memcpy(&ua, &a, 4);
    memcpy(&ub, &b, 4);
    unsigned l = hipemu::lane_id();
    const uint32_t* pa = hipemu::wave_publish(ua, 1);
    const uint32_t* pb = hipemu::wave_publish(ub, 2);
    const float* fa = reinterpret_cast<const float*>(pa);
    const float* fb = reinterpret_cast<const float*>(pb);
    for (int r = 0; r < 4; ++r) {
        int row = 4 * (int)(l >> 4) + r;
        int col = (int)(l & 15);
        float acc = c[r];
        for (int k = 0; k < 4; ++k) acc = fmaf(fa[row + 16 * k], fb[col + 16 * k], acc);
        c[r] = acc;
    }
    return c;
}

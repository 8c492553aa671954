// The SyncBN sufficient-statistics exchange of one node as ONE small kernel per norm layer and direction (SURVEY.md section 5 /
// section 7 hard part 6, verdict r3 item 3): replaces the reference's reduce-to-master + broadcast through SyncMaster queues
// (sync_batchnorm/batchnorm.py:95-111, comm.py:102-133) -- and the 84 RCCL all-reduces per iteration that csrc/comm.hip issued
// for it: <= 8 KB messages, for which a collective library's launch + protocol latency (15 - 30 us on 8 ranks) is the cost.
//
// Every rank owns a mailbox in its own HBM, exported with hipIpcGetMemHandle and mapped by every peer of the node
// (hipIpcOpenMemHandle; one process per GPU, the handles travel through torch.distributed's rendezvous):
//     mailbox = SLOTS x world x { flag word | MAXF floats }
// An exchange with sequence number q (a DEVICE-resident counter that the kernel advances itself, so a captured hipGraph
// replays it without a host value) uses slot q mod SLOTS:
//   push   rank r stores its n floats into row r of slot s of EVERY rank's mailbox -- its own included; peers are written
//          over xGMI with system-scope stores -- then, behind a system-scope release fence, stores q into that row's flag;
//   wait   spins (system-scope acquire loads) until the `world` flags of ITS OWN mailbox's slot s hold q;
//   sum    adds the `world` rows in rank order -- every rank forms the same sum in the same order: bit-identical statistics
//          on all ranks, no broadcast.
// One block of 256 threads on the kernels' stream: in order with the producing and the consuming kernel, capturable.
// Slot reuse: a rank finishes exchange q + 1 only after every peer pushed q + 1, which a peer does only after it finished
// READING q -- so when a rank pushes q + 2 nobody reads slot q any more: two slots suffice, four are used.
// A peer that never arrives would hang the GPU: the wait gives up after `timeout_ms` (wall clock), raises the handle's
// device error word (mnk_p2p_error) and lets the kernel finish with whatever it has.
#include <string.h>

#include <vector>

#include "mnk_common.h"

using namespace mnk;

#ifndef HIPEMU
namespace {

constexpr int P2P_SLOTS = 4;
constexpr int P2P_MAXF = 2048 + 64;       // floats per message: [sum, sum of squares] of <= 1024 channels (+ slack)
constexpr int P2P_ROW = P2P_MAXF + 16;    // a row: 16 words of header (word 0 = flag), then the payload
constexpr int P2P_MAX_WORLD = 16;

struct PeerTable {
    unsigned* box[P2P_MAX_WORLD];         // every rank's mailbox as mapped into THIS process (box[rank] = the local allocation)
};

struct P2P {
    int rank, world;
    unsigned* local;                      // this rank's mailbox
    unsigned* state;                      // device words: [0] sequence counter, [1] error flag
    PeerTable peers;
    bool opened[P2P_MAX_WORLD];
    size_t bytes;
};

__device__ __forceinline__ unsigned* row_of(unsigned* box, int world, int slot, int r) {
    return box + ((size_t)slot * world + r) * P2P_ROW;
}

__global__ void __launch_bounds__(256) p2p_allreduce_kernel(PeerTable peers, int rank, int world, unsigned* __restrict__ state,
                                                            const float* __restrict__ in, float* __restrict__ out, int n,
                                                            unsigned long long timeout_ticks) {
    __shared__ unsigned seq_s;
    __shared__ int ok_s;
    const int t = threadIdx.x;
    if (t == 0) {
        seq_s = state[0] + 1;             // this exchange's number (the first one is 1: a zeroed mailbox never matches)
        ok_s = 1;
    }
    __syncthreads();
    const unsigned seq = seq_s;
    const int slot = (int)(seq % P2P_SLOTS);
    // ---- push: this rank's row of the slot in every mailbox (system scope: the peers' kernels are running)
    for (int q = 0; q < world; ++q) {
        unsigned* row = row_of(peers.box[q], world, slot, rank);
        for (int i = t; i < n; i += 256)
            __hip_atomic_store(reinterpret_cast<float*>(row + 16) + i, in[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __threadfence_system();
    __syncthreads();
    if (t < world) __hip_atomic_store(row_of(peers.box[t], world, slot, rank), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    // ---- wait: every rank's row of the slot in THIS rank's mailbox
    if (t < world) {
        const unsigned* flag = row_of(peers.box[rank], world, slot, t);
        const unsigned long long t0 = wall_clock64();
        while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != seq) {
            if ((unsigned long long)wall_clock64() - t0 > timeout_ticks) {
                ok_s = 0;
                __hip_atomic_store(state + 1, 1u + (unsigned)t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
    }
    __syncthreads();
    __threadfence_system();
    // ---- sum in rank order (every rank: the same order, the same bits)
    for (int i = t; i < n; i += 256) {
        float s = 0.f;
        for (int q = 0; q < world; ++q)
            s += __hip_atomic_load(reinterpret_cast<const float*>(row_of(peers.box[rank], world, slot, q) + 16) + i,
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        out[i] = s;
    }
    __syncthreads();
    if (t == 0) state[0] = seq;
}

}  // namespace
#endif

extern "C" {

int mnk_p2p_max_floats(void) {
#ifdef HIPEMU
    return 0;
#else
    return P2P_MAXF;
#endif
}

int mnk_p2p_create(int rank, int world, void** handle_out) {
    MNK_REQUIRE(handle_out && world >= 1 && rank >= 0 && rank < world);
#ifdef HIPEMU
    set_error("mnk_p2p_create: the peer-to-peer exchange needs the HIP runtime (not available in the CPU emulation)");
    return MNK_ECOMM;
#else
    MNK_REQUIRE(world <= P2P_MAX_WORLD);
    P2P* p = new P2P();
    memset(p, 0, sizeof(*p));
    p->rank = rank;
    p->world = world;
    p->bytes = (size_t)P2P_SLOTS * world * P2P_ROW * sizeof(unsigned);
    // (plain device memory: the kernel's own accesses are system-scope atomics, which go to memory on every access)
    if (hipMalloc((void**)&p->local, p->bytes) != hipSuccess || hipMalloc((void**)&p->state, 64) != hipSuccess) {
        set_error("mnk_p2p_create: hipMalloc failed");
        delete p;
        return MNK_ECOMM;
    }
    (void)hipMemset(p->local, 0, p->bytes);
    (void)hipMemset(p->state, 0, 64);
    (void)hipDeviceSynchronize();
    p->peers.box[rank] = p->local;
    *handle_out = p;
    return MNK_OK;
#endif
}

int mnk_p2p_export(void* handle, void* ipc_handle64) {
    MNK_REQUIRE(handle && ipc_handle64);
#ifdef HIPEMU
    return MNK_ECOMM;
#else
    P2P* p = (P2P*)handle;
    hipIpcMemHandle_t h;
    static_assert(sizeof(hipIpcMemHandle_t) <= 64, "the exported handle travels as 64 bytes");
    const hipError_t e = hipIpcGetMemHandle(&h, p->local);
    if (e != hipSuccess) {
        set_error("mnk_p2p_export: hipIpcGetMemHandle failed: %s (HSA_ENABLE_IPC_MODE_LEGACY=0 in the environment?)",
                  hipGetErrorString(e));
        return MNK_ECOMM;
    }
    memset(ipc_handle64, 0, 64);
    memcpy(ipc_handle64, &h, sizeof(h));
    return MNK_OK;
#endif
}

int mnk_p2p_connect(void* handle, const void* all_handles) {
    MNK_REQUIRE(handle && all_handles);
#ifdef HIPEMU
    return MNK_ECOMM;
#else
    P2P* p = (P2P*)handle;
    for (int q = 0; q < p->world; ++q) {
        if (q == p->rank) continue;
        hipIpcMemHandle_t h;
        memcpy(&h, (const char*)all_handles + 64 * q, sizeof(h));
        void* ptr = nullptr;
        const hipError_t e = hipIpcOpenMemHandle(&ptr, h, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) {
            set_error("mnk_p2p_connect: hipIpcOpenMemHandle of rank %d's mailbox failed: %s", q, hipGetErrorString(e));
            return MNK_ECOMM;
        }
        p->peers.box[q] = (unsigned*)ptr;
        p->opened[q] = true;
    }
    return MNK_OK;
#endif
}

int mnk_p2p_allreduce(void* handle, const float* in, float* out, int n, int timeout_ms, void* stream) {
    MNK_REQUIRE(handle && in && out && n > 0 && timeout_ms > 0);
#ifdef HIPEMU
    return MNK_ECOMM;
#else
    P2P* p = (P2P*)handle;
    MNK_REQUIRE(n <= P2P_MAXF);
    for (int q = 0; q < p->world; ++q) MNK_REQUIRE(p->peers.box[q] != nullptr);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(p2p_allreduce_kernel, dim3(1), dim3(256), 0, s, p->peers, p->rank, p->world, p->state, in, out, n,
                       (unsigned long long)timeout_ms * 100000ull);      // wall_clock64: 100 MHz
    MNK_LAUNCH_CHECK();
    return MNK_OK;
#endif
}

int mnk_p2p_error(void* handle, int* flag_out) {
    MNK_REQUIRE(handle && flag_out);
#ifdef HIPEMU
    return MNK_ECOMM;
#else
    P2P* p = (P2P*)handle;
    unsigned st[2] = {0, 0};
    if (hipMemcpy(st, p->state, sizeof(st), hipMemcpyDeviceToHost) != hipSuccess) return MNK_ELAUNCH;
    *flag_out = (int)st[1];
    return MNK_OK;
#endif
}

int mnk_p2p_destroy(void* handle) {
    MNK_REQUIRE(handle);
#ifdef HIPEMU
    return MNK_ECOMM;
#else
    P2P* p = (P2P*)handle;
    (void)hipDeviceSynchronize();
    for (int q = 0; q < p->world; ++q)
        if (p->opened[q]) (void)hipIpcCloseMemHandle(p->peers.box[q]);
    if (p->local) (void)hipFree(p->local);
    if (p->state) (void)hipFree(p->state);
    delete p;
    return MNK_OK;
#endif
}
}

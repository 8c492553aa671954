// The SyncBN sufficient-statistics exchange of one node as ONE small kernel per norm layer and direction (SURVEY.md section 5 /
// section 7 hard part 6, verdict r3 item 3): replaces the reference's reduce-to-master + broadcast through SyncMaster queues
// (sync_batchnorm/batchnorm.py:95-111, comm.py:102-133) -- and the 84 RCCL all-reduces per iteration that csrc/comm.hip issued
// for it: <= 8 KB messages, for which a collective library's launch + protocol latency (15 - 30 us on 8 ranks) is the cost.
//
// Every rank owns a mailbox in its own HBM, exported with hipIpcGetMemHandle and mapped by every peer of the node
// (hipIpcOpenMemHandle; one process per GPU, the handles travel through torch.distributed's rendezvous):
//     mailbox = SLOTS x world x MAXF words of 8 bytes, a word = { sequence number | one float }
// An exchange with sequence number q (a DEVICE-resident counter that the kernel advances itself, so a captured hipGraph
// replays it without a host value) uses slot q mod SLOTS:
//   push   rank r stores its n floats, each PACKED WITH q into one 8-byte word, into row r of slot s of EVERY rank's mailbox --
//          its own included; peers are written over xGMI -- with system-scope relaxed atomic stores (write-through, single
//          copy atomic: a reader sees the old word or the new one, never half);
//   sum    every thread polls, for its elements, the `world` rows of ITS OWN mailbox's slot s (system-scope relaxed atomic
//          loads: they bypass the caches) until a word carries q, and adds the floats in rank order -- every rank forms the same
//          sum in the same order: bit-identical statistics on all ranks, no broadcast.
// Flag and data travel in ONE word, so there is no ordering to enforce between them: no release / acquire fence -- on this chip
// a system-scope fence writes back and invalidates the XCD's whole L2, which holds the 16+ MB of the convolution output in front
// of the norm layer (the first form of this kernel, data rows + a flag per row behind __threadfence_system(), cost 7.6 us per
// exchange on one rank against 3.8 for RCCL's one-rank copy: profiles/r04_knob_ab_log.txt).
// One block of 256 threads on the kernels' stream: in order with the producing and the consuming kernel, capturable.
// Slot reuse: a rank finishes exchange q + 1 only after every peer pushed q + 1, which a peer does only after it finished
// READING q -- so when a rank pushes q + 2 nobody reads slot q any more: two slots suffice, four are used.
// A peer that never arrives would hang the GPU: the polling gives up after `timeout_ms` (wall clock), raises the handle's
// device error word (mnk_p2p_error), puts NaN in place of the missing values (never a stale word) and lets the kernel finish;
// once the error word is set no later exchange waits again.
#include <string.h>

#include <vector>

#include "mnk_common.h"
#include "p2p.h"

using namespace mnk;

namespace {

__global__ void __launch_bounds__(256) p2p_allreduce_kernel(PeerTable peers, int rank, int world, unsigned* __restrict__ state,
                                                            const float* __restrict__ in, float* __restrict__ out, int n,
                                                            unsigned long long timeout_ticks) {
    const int t = threadIdx.x;
    const unsigned seq = state[0] + 1;    // this exchange's number (the first one is 1: a zeroed mailbox never matches)
    const int slot = (int)(seq % P2P_SLOTS);
    // ---- push: this rank's row of the slot in every mailbox, {seq | value} words
    float mine[(P2P_MAXF + 255) / 256];
#pragma unroll
    for (int k = 0; k < (P2P_MAXF + 255) / 256; ++k) {
        const int i = t + 256 * k;
        mine[k] = i < n ? in[i] : 0.f;
    }
    for (int q = 0; q < world; ++q) {
        unsigned long long* row = row_of(peers.box[q], world, slot, rank);
#pragma unroll
        for (int k = 0; k < (P2P_MAXF + 255) / 256; ++k) {
            const int i = t + 256 * k;
            if (i < n)
                __hip_atomic_store(row + i, ((unsigned long long)seq << 32) | (unsigned long long)__float_as_uint(mine[k]),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    // ---- poll + sum in rank order (every rank: the same order, the same bits)
    const unsigned long long t0 = wall_clock64();
    bool gave_up = false;

#pragma unroll
    for (int k = 0; k < (P2P_MAXF + 255) / 256; ++k) {
        const int i = t + 256 * k;
        if (i >= n) break;
        float s = 0.f;
        for (int q = 0; q < world; ++q) {
            const unsigned long long* w = row_of(peers.box[rank], world, slot, q) + i;
            unsigned long long v = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            while ((unsigned)(v >> 32) != seq && !gave_up) {
                const unsigned long long waited = (unsigned long long)wall_clock64() - t0;
                if (waited > timeout_ticks ||          // (see p2p.h: no second full wait after a give-up)
                    (waited > P2P_RECHECK_TICKS && __hip_atomic_load(state + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                    gave_up = true;
                    __hip_atomic_store(state + 1, 1u + (unsigned)q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
                v = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            if ((unsigned)(v >> 32) != seq) v = P2P_POISON;     // gave up: NaN, never a stale word
            s += __uint_as_float((unsigned)v);
        }
        out[i] = s;
    }
    __syncthreads();                      // (every thread has read state[0] long before this)
    if (t == 0) state[0] = seq;
}

}  // namespace

namespace mnk {
// what a kernel of another file needs to carry an exchange (batchnorm.hip's synchronised second stage)
bool p2p_launch_info(void* handle, PeerTable* peers, int* rank, int* world, unsigned** state) {
    if (!handle) return false;
    P2P* p = (P2P*)handle;
    for (int q = 0; q < p->world; ++q)
        if (!p->peers.box[q]) return false;
    *peers = p->peers, *rank = p->rank, *world = p->world, *state = p->state;
    return true;
}
}  // namespace mnk

extern "C" {

int mnk_p2p_max_floats(void) {
    return P2P_MAXF;
}

int mnk_p2p_create(int rank, int world, void** handle_out) {
    MNK_REQUIRE(handle_out && world >= 1 && rank >= 0 && rank < world);
    MNK_REQUIRE(world <= P2P_MAX_WORLD);
    P2P* p = new P2P();
    memset(p, 0, sizeof(*p));
    p->rank = rank;
    p->world = world;
    p->bytes = (size_t)P2P_SLOTS * world * P2P_MAXF * sizeof(unsigned long long);
    // The mailbox is written by OTHER devices over xGMI: such writes reach this device's memory without passing through its
    // L2, and ordinary (coarse-grained) device memory may be held in that L2 -- a polling load could then hit a stale line for
    // ever.  Uncached device memory (MTYPE UC: what RCCL allocates for its own flags on this architecture) takes the L2 out of
    // the picture for every accessor; fine-grained memory is the second choice, plain memory the last one (enough for
    // processes that share one device and therefore one L2: tests/test_p2p_gpu.py).  The start-up self-test of
    // mnk.dist.p2p_comm() decides whether what came out is used at all.
    p->memory_kind = 3;
    hipError_t e = hipExtMallocWithFlags((void**)&p->local, p->bytes, hipDeviceMallocUncached);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        p->memory_kind = 1;
        e = hipExtMallocWithFlags((void**)&p->local, p->bytes, hipDeviceMallocFinegrained);
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        p->memory_kind = 0;
        e = hipMalloc((void**)&p->local, p->bytes);
    }
    if (e != hipSuccess || hipMalloc((void**)&p->state, 64) != hipSuccess) {
        set_error("mnk_p2p_create: device allocation failed: %s", hipGetErrorString(e));
        if (p->local) (void)hipFree(p->local);
        delete p;
        return MNK_ECOMM;
    }
    (void)hipMemset(p->local, 0, p->bytes);
    (void)hipMemset(p->state, 0, 64);
    (void)hipDeviceSynchronize();
    p->peers.box[rank] = p->local;
    *handle_out = p;
    return MNK_OK;
}

int mnk_p2p_export(void* handle, void* ipc_handle64) {
    MNK_REQUIRE(handle && ipc_handle64);
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
}

int mnk_p2p_connect(void* handle, const void* all_handles) {
    MNK_REQUIRE(handle && all_handles);
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
        p->peers.box[q] = (unsigned long long*)ptr;
        p->opened[q] = true;
    }
    return MNK_OK;
}

int mnk_p2p_allreduce(void* handle, const float* in, float* out, int n, int timeout_ms, void* stream) {
    MNK_REQUIRE(handle && in && out && n > 0 && timeout_ms > 0);
    P2P* p = (P2P*)handle;
    MNK_REQUIRE(n <= P2P_MAXF);
    for (int q = 0; q < p->world; ++q) MNK_REQUIRE(p->peers.box[q] != nullptr);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(p2p_allreduce_kernel, dim3(1), dim3(256), 0, s, p->peers, p->rank, p->world, p->state, in, out, n,
                       (unsigned long long)timeout_ms * 100000ull);      // wall_clock64: 100 MHz
    MNK_LAUNCH_CHECK();
    return MNK_OK;
}

int mnk_p2p_error(void* handle, int* flag_out) {
    MNK_REQUIRE(handle && flag_out);
    P2P* p = (P2P*)handle;
    unsigned st[2] = {0, 0};
    if (hipMemcpy(st, p->state, sizeof(st), hipMemcpyDeviceToHost) != hipSuccess) return MNK_ELAUNCH;
    *flag_out = (int)st[1];
    return MNK_OK;
}

int mnk_p2p_memory_kind(void* handle) {
    return handle ? ((P2P*)handle)->memory_kind : -1;
}

int mnk_p2p_destroy(void* handle) {
    MNK_REQUIRE(handle);
    P2P* p = (P2P*)handle;
    (void)hipDeviceSynchronize();
    for (int q = 0; q < p->world; ++q)
        if (p->opened[q]) (void)hipIpcCloseMemHandle(p->peers.box[q]);
    if (p->local) (void)hipFree(p->local);
    if (p->state) (void)hipFree(p->state);
    delete p;
    return MNK_OK;
}
}

// Shared definitions of the peer-to-peer SyncBN exchange (csrc/p2p.hip): the mailbox layout and the per-word protocol, for the
// kernels that carry an exchange inside them (p2p.hip: the stand-alone all-reduce; batchnorm.hip: the second stage of the
// BatchNorm statistics, which pushes each column sum as it finishes it).
#pragma once
#include "mnk_common.h"

namespace mnk {

constexpr int P2P_SLOTS = 4;
constexpr int P2P_MAXF = 2048 + 64;       // floats per message: [sum, sum of squares] of <= 1024 channels (+ slack)
constexpr int P2P_MAX_WORLD = 16;
// what an exchange that gave up on a peer (timeout) reads in place of that peer's value: a quiet NaN.  The statistics -- and
// with them the losses -- of every rank that waited become NaN instead of silently wrong; the handle's error word says which
// rank was missing (mnk_p2p_error), and mnk.engine.TrainStep polls it.
constexpr unsigned long long P2P_POISON = 0x7fc00000ull;
constexpr unsigned long long P2P_RECHECK_TICKS = 100000ull;      // 1 ms of the 100 MHz wall clock

struct PeerTable {
    unsigned long long* box[P2P_MAX_WORLD];      // every rank's mailbox as mapped into THIS process (box[rank] = the local one)
};

struct P2P {
    int rank, world;
    unsigned long long* local;            // this rank's mailbox
    unsigned* state;                      // device words: [0] sequence counter, [1] error flag
    PeerTable peers;
    bool opened[P2P_MAX_WORLD];
    size_t bytes;
    int memory_kind;                      // 3 uncached, 1 fine-grained, 0 ordinary device memory
};
__device__ __forceinline__ unsigned long long* row_of(unsigned long long* box, int world, int slot, int r) {
    return box + ((size_t)slot * world + r) * P2P_MAXF;
}


// one value of exchange `seq`: push it into row `rank` of every mailbox (lane q of the calling wave serves peer q), then poll the
// own mailbox's row q (lane q) and add the `world` values in rank order.  Called by one whole wavefront; returns the sum in every
// lane.  state[1] is set -- and the peer's value replaced by NaN -- when its word did not arrive within timeout_ticks of the 100 MHz
// wall clock.
__device__ __forceinline__ float p2p_exchange_value(const PeerTable& peers, int rank, int world, int slot, unsigned seq, int index,
                                                    float value, unsigned* state, unsigned long long timeout_ticks) {
    const int lane = threadIdx.x & 63;
    float got = 0.f;
    if (lane < world) {
        __hip_atomic_store(row_of(peers.box[lane], world, slot, rank) + index,
                           ((unsigned long long)seq << 32) | (unsigned long long)__float_as_uint(value), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_SYSTEM);
        const unsigned long long* w = row_of(peers.box[rank], world, slot, lane) + index;
        const unsigned long long t0 = wall_clock64();
        unsigned long long v = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        while ((unsigned)(v >> 32) != seq) {
            // a handle that already gave a peer up does not wait the whole timeout again -- but the error word is looked at only
            // after a millisecond of waiting: read in front of the poll it put one more memory round trip into EVERY exchange
            // (84 per iteration: +0.1 ms on the forced-rank step)
            const unsigned long long waited = (unsigned long long)wall_clock64() - t0;
            if (waited > timeout_ticks ||
                (waited > P2P_RECHECK_TICKS && __hip_atomic_load(state + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                __hip_atomic_store(state + 1, 1u + (unsigned)lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                v = P2P_POISON;       // a stale word must never pass for the peer's value: the sum becomes NaN
                break;
            }
            __builtin_amdgcn_s_sleep(1);
            v = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        got = __uint_as_float((unsigned)v);
    }
    float s = 0.f;
    for (int q = 0; q < world; ++q) s += __shfl(got, q);
    return s;
}

// up to 64 / W values of exchange `seq` at once (W = 8 for up to eight ranks, else 16): lane l of the calling wave serves value
// l / W and peer l % W -- one round trip for all of them (a kernel that owns a handful of columns, e.g. the one-launch small-layer
// BatchNorm backward: 4 channels x 2 sums per block).  value / index: this lane's value j = l / W and its position in the
// mailbox row (negative: nothing to exchange for this lane).  Returns, in every lane, the rank-ordered sum of ITS value j.
__device__ __forceinline__ float p2p_exchange_values(const PeerTable& peers, int rank, int world, int slot, unsigned seq, int index,
                                                     float value, unsigned* state, unsigned long long timeout_ticks) {
    const int lane = threadIdx.x & 63;
    const int W = world <= 8 ? 8 : 16, q = lane & (W - 1);
    float got = 0.f;
    if (q < world && index >= 0) {
        __hip_atomic_store(row_of(peers.box[q], world, slot, rank) + index,
                           ((unsigned long long)seq << 32) | (unsigned long long)__float_as_uint(value), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_SYSTEM);
        const unsigned long long* w = row_of(peers.box[rank], world, slot, q) + index;
        const unsigned long long t0 = wall_clock64();
        unsigned long long v = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        while ((unsigned)(v >> 32) != seq) {
            // a handle that already gave a peer up does not wait the whole timeout again -- but the error word is looked at only
            // after a millisecond of waiting: read in front of the poll it put one more memory round trip into EVERY exchange
            // (84 per iteration: +0.1 ms on the forced-rank step)
            const unsigned long long waited = (unsigned long long)wall_clock64() - t0;
            if (waited > timeout_ticks ||
                (waited > P2P_RECHECK_TICKS && __hip_atomic_load(state + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                __hip_atomic_store(state + 1, 1u + (unsigned)q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                v = P2P_POISON;       // a stale word must never pass for the peer's value: the sum becomes NaN
                break;
            }
            __builtin_amdgcn_s_sleep(1);
            v = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        got = __uint_as_float((unsigned)v);
    }
    float s = 0.f;
    const int base = lane & ~(W - 1);
    for (int r = 0; r < world; ++r) s += __shfl(got, base + r);
    return s;
}

// the exchange's sequence number: every block reads state[0] + 1 when it starts; the LAST block to finish (ticket in state[2])
// advances state[0] -- no block of the launch can still be waiting to read it then
__device__ __forceinline__ void p2p_finish_launch(unsigned* state, unsigned seq) {
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned done = atomicAdd(state + 2, 1u);
        if (done == gridDim.x * gridDim.y * gridDim.z - 1) {
            state[2] = 0;
            state[0] = seq;
        }
    }
}

}  // namespace mnk

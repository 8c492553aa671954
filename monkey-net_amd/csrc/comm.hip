// The two collectives of the data-parallel hot path behind the C-ABI: RCCL all-reduce over xGMI of
//   * BatchNorm sufficient statistics  [sum x, sum x^2] / [sum g, sum g*xhat]  (<= 8 KB per layer, one per norm layer and
//     direction) -- replaces the reference's reduce-to-master + broadcast through SyncMaster queues
//     (sync_batchnorm/batchnorm.py:95-111, comm.py), and
//   * the flat gradient buffer of an optimiser (hundreds of MB) -- replaces DataParallel's implicit reduce-add of
//     replica gradients and its per-forward parameter broadcast (train.py:104-105).
// The collectives are issued on the CALLER's stream (the stream the kernels run on): in order with the producing and
// consuming kernels, no event hand-over to a communication stream, and capturable into the iteration's hipGraph as they
// are.  One process per GPU; the communicator is created from a 128-byte id that rank 0 makes and the host layer hands
// to the other ranks (mnk.dist broadcasts it through torch.distributed's store).
// RCCL is bound at first use with dlopen (librccl.so.1 -- the copy PyTorch-ROCm already has in the process when the
// host is Python), so the library loads on machines without RCCL and the CPU emulator build has no such dependency.
#include <dlfcn.h>
#include <string.h>

#include "mnk_common.h"

#ifndef HIPEMU
#include <rccl/rccl.h>
#endif

using namespace mnk;

#ifndef HIPEMU
namespace {

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    bool ok = false;
};

Rccl& rccl() {
    static Rccl r;
    if (r.handle || r.ok) return r;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
        r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (r.handle) break;
    }
    if (!r.handle) return r;
#define MNK_SYM(field, name) *(void**)(&r.field) = dlsym(r.handle, name)
    MNK_SYM(GetUniqueId, "ncclGetUniqueId");
    MNK_SYM(CommInitRank, "ncclCommInitRank");
    MNK_SYM(CommDestroy, "ncclCommDestroy");
    MNK_SYM(AllReduce, "ncclAllReduce");
    MNK_SYM(GetErrorString, "ncclGetErrorString");
    MNK_SYM(GroupStart, "ncclGroupStart");
    MNK_SYM(GroupEnd, "ncclGroupEnd");
#undef MNK_SYM
    r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllReduce && r.GetErrorString && r.GroupStart && r.GroupEnd;
    return r;
}

struct Comm {
    ncclComm_t comm;
    int rank, world;
};

int fail(const char* what, ncclResult_t rc) {
    set_error("%s: %s", what, rccl().GetErrorString ? rccl().GetErrorString(rc) : "RCCL error");
    return MNK_ECOMM;
}

}  // namespace
#endif

extern "C" {

int mnk_comm_available(void) {
#ifdef HIPEMU
    return 0;
#else
    return rccl().ok ? 1 : 0;
#endif
}

int mnk_comm_unique_id(void* id128) {
    MNK_REQUIRE(id128);
#ifdef HIPEMU
    set_error("mnk_comm_unique_id: no RCCL in the CPU emulator build");
    return MNK_ECOMM;
#else
    if (!rccl().ok) {
        set_error("mnk_comm_unique_id: librccl.so.1 could not be loaded");
        return MNK_ECOMM;
    }
    static_assert(sizeof(ncclUniqueId) == 128, "RCCL unique id is 128 bytes");
    ncclUniqueId id;
    ncclResult_t rc = rccl().GetUniqueId(&id);
    if (rc != ncclSuccess) return fail("ncclGetUniqueId", rc);
    memcpy(id128, &id, 128);
    return MNK_OK;
#endif
}

int mnk_comm_init(const void* id128, int rank, int world, void** comm_out) {
    MNK_REQUIRE(id128 && comm_out && world > 0 && rank >= 0 && rank < world);
#ifdef HIPEMU
    set_error("mnk_comm_init: no RCCL in the CPU emulator build");
    return MNK_ECOMM;
#else
    if (!rccl().ok) {
        set_error("mnk_comm_init: librccl.so.1 could not be loaded");
        return MNK_ECOMM;
    }
    ncclUniqueId id;
    memcpy(&id, id128, 128);
    Comm* c = new Comm{nullptr, rank, world};
    ncclResult_t rc = rccl().CommInitRank(&c->comm, world, id, rank);     // uses the calling thread's current device
    if (rc != ncclSuccess) {
        delete c;
        return fail("ncclCommInitRank", rc);
    }
    *comm_out = c;
    return MNK_OK;
#endif
}

int mnk_comm_destroy(void* comm) {
#ifdef HIPEMU
    return MNK_OK;
#else
    if (!comm) return MNK_OK;
    Comm* c = (Comm*)comm;
    ncclResult_t rc = rccl().CommDestroy(c->comm);
    delete c;
    return rc == ncclSuccess ? MNK_OK : fail("ncclCommDestroy", rc);
#endif
}

// out[0..n) <- sum over ranks of sums[0..n) on `stream` (out == sums: in place)
int mnk_allreduce_bnstats_to(void* comm, const float* sums, float* out, long n, void* stream) {
    MNK_REQUIRE(comm && sums && out && n > 0);
#ifdef HIPEMU
    set_error("mnk_allreduce_bnstats: no RCCL in the CPU emulator build");
    return MNK_ECOMM;
#else
    Comm* c = (Comm*)comm;
    ncclResult_t rc = rccl().AllReduce(sums, out, (size_t)n, ncclFloat32, ncclSum, c->comm, (hipStream_t)stream);
    return rc == ncclSuccess ? MNK_OK : fail("ncclAllReduce (BatchNorm statistics)", rc);
#endif
}

int mnk_allreduce_bnstats(void* comm, float* sums, long n, void* stream) { return mnk_allreduce_bnstats_to(comm, sums, sums, n, stream); }

// grads[0..n) <- sum (average = 0) or mean (average = 1) over ranks, in place, on `stream`, as ceil(n / chunk) collectives
// grouped into one RCCL launch (xGMI rings are per-link bound: few large messages); chunk_floats <= 0: one collective
int mnk_allreduce_grads(void* comm, float* grads, long n, int average, long chunk_floats, void* stream) {
    MNK_REQUIRE(comm && grads && n > 0);
#ifdef HIPEMU
    set_error("mnk_allreduce_grads: no RCCL in the CPU emulator build");
    return MNK_ECOMM;
#else
    Comm* c = (Comm*)comm;
    const ncclRedOp_t op = average ? ncclAvg : ncclSum;
    if (chunk_floats <= 0 || chunk_floats >= n) {
        ncclResult_t rc = rccl().AllReduce(grads, grads, (size_t)n, ncclFloat32, op, c->comm, (hipStream_t)stream);
        return rc == ncclSuccess ? MNK_OK : fail("ncclAllReduce (gradients)", rc);
    }
    ncclResult_t rc = rccl().GroupStart();
    if (rc != ncclSuccess) return fail("ncclGroupStart", rc);
    for (long o = 0; o < n; o += chunk_floats) {
        const long k = n - o < chunk_floats ? n - o : chunk_floats;
        rc = rccl().AllReduce(grads + o, grads + o, (size_t)k, ncclFloat32, op, c->comm, (hipStream_t)stream);
        if (rc != ncclSuccess) {
            (void)rccl().GroupEnd();
            return fail("ncclAllReduce (gradients)", rc);
        }
    }
    rc = rccl().GroupEnd();
    return rc == ncclSuccess ? MNK_OK : fail("ncclGroupEnd", rc);
#endif
}
}

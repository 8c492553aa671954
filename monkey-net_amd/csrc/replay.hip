// A stream executor for a captured iteration ("hipGraph capture of the whole training step").
//
// mnk.engine.TrainStep captures one iteration of train.py:110-136 as a hipGraph.  hipGraphLaunch replays a LINEAR graph as one
// back-to-back kernel chain, but a graph with a second branch replays 0.85 - 1.0 ms slower on this runtime whatever the branch
// holds (profiles/r03_knob_ab_log.txt), so the iteration could not use what its dependency graph offers: the weight-gradient
// GEMMs only need dy and x of their layer, the appearance encoder does not depend on the key-point detector, the
// discriminator-loss backward does not depend on the generator's update -- independent work that could fill the ramp and
// drain of the one-generation GEMM launches and the matrix-pipe-idle normalisation passes of the critical chain.
//
// This file replays the captured graph itself: the nodes (kernel / memset / memcpy) and edges are read back from the
// hipGraph_t, put in a topological order that follows the capture order, split into chains that are bound to HIP streams
// (chain 0 = the caller's stream), and launched with plain hipLaunchKernel calls; an edge that crosses streams is an event
// record behind the producer + a hipStreamWaitEvent in front of the consumer.  No hipGraphExec is involved: every launch is an
// ordinary stream launch, so kernels of different streams overlap as the hardware queues allow.  The kernel argument blocks
// are the captured graph's own (owned by the hipGraph_t, which the caller keeps alive), i.e. exactly the launches a
// hipGraphLaunch would issue.
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <queue>
#include <vector>

#include "mnk_common.h"

using namespace mnk;

#ifndef HIPEMU
namespace {

constexpr int MAX_STREAMS = 8;
// 1: side streams are created with the lowest stream priority (the caller's chain -- the critical path -- gets the free
// workgroup slots first)
int g_side_priority = tuning_knob("replay_side_priority", &g_side_priority, 0);

struct RNode {
    hipGraphNodeType type;
    hipKernelNodeParams k;
    hipMemsetParams ms;
    hipMemcpy3DParms mc;
    int stream = 0;
    int module_launch = 0;            // 0: not decided, 1: hipLaunchKernel(host function), 2: hipModuleLaunchKernel(hipFunction_t)
    int record = -1;                  // event recorded behind this node (index into Program::events), -1: none
    std::vector<int> waits;           // events this node's stream waits for in front of it
};

struct Program {
    std::vector<RNode> nodes;         // in launch order
    std::vector<hipEvent_t> events;
    hipStream_t side[MAX_STREAMS];    // side[0] unused: stream 0 is the caller's
    int nstreams = 1;
    hipEvent_t begin = nullptr;
    std::vector<int> tails;           // per stream > 0: event recorded at its last node (the caller's stream joins on them)
    long info[8];
};

const char* type_name(hipGraphNodeType t) {
    switch (t) {
        case hipGraphNodeTypeKernel: return "kernel";
        case hipGraphNodeTypeMemcpy: return "memcpy";
        case hipGraphNodeTypeMemset: return "memset";
        case hipGraphNodeTypeHost: return "host";
        case hipGraphNodeTypeGraph: return "child graph";
        case hipGraphNodeTypeEmpty: return "empty";
        case hipGraphNodeTypeWaitEvent: return "event wait";
        case hipGraphNodeTypeEventRecord: return "event record";
        default: return "other";
    }
}

int launch_node(RNode& n, hipStream_t s) {
    hipError_t e = hipSuccess;
    switch (n.type) {
        case hipGraphNodeTypeKernel:
            if (n.module_launch != 2 && n.k.kernelParams) {
                e = hipLaunchKernel(n.k.func, n.k.gridDim, n.k.blockDim, n.k.kernelParams, n.k.sharedMemBytes, s);
                if (e == hipSuccess) {
                    n.module_launch = 1;
                    break;
                }
                if (n.module_launch == 1) break;
                (void)hipGetLastError();
            }
            e = hipModuleLaunchKernel((hipFunction_t)n.k.func, n.k.gridDim.x, n.k.gridDim.y, n.k.gridDim.z, n.k.blockDim.x,
                                      n.k.blockDim.y, n.k.blockDim.z, n.k.sharedMemBytes, s, n.k.kernelParams, n.k.extra);
            if (e == hipSuccess) n.module_launch = 2;
            break;
        case hipGraphNodeTypeMemset:
            if (n.ms.height <= 1) {
                const size_t bytes = n.ms.width * n.ms.elementSize;
                if (n.ms.elementSize == 1)
                    e = hipMemsetAsync(n.ms.dst, (int)n.ms.value, bytes, s);
                else if (n.ms.elementSize == 4)
                    e = hipMemsetD32Async((hipDeviceptr_t)n.ms.dst, (int)n.ms.value, n.ms.width, s);
                else if (n.ms.elementSize == 2)
                    e = hipMemsetD16Async((hipDeviceptr_t)n.ms.dst, (unsigned short)n.ms.value, n.ms.width, s);
                else
                    e = hipErrorNotSupported;
            } else {
                e = n.ms.elementSize == 1 ? hipMemset2DAsync(n.ms.dst, n.ms.pitch, (int)n.ms.value, n.ms.width, n.ms.height, s)
                                          : hipErrorNotSupported;
            }
            break;
        case hipGraphNodeTypeMemcpy:
            e = hipMemcpy3DAsync(&n.mc, s);
            break;
        case hipGraphNodeTypeEmpty:
            break;
        default:
            e = hipErrorNotSupported;
    }
    if (e != hipSuccess) {
        set_error("mnk_replay_launch: %s node failed: %s", type_name(n.type), hipGetErrorString(e));
        return MNK_ELAUNCH;
    }
    return MNK_OK;
}

}  // namespace
#endif

extern "C" {

int mnk_replay_create(void* hip_graph, int max_streams, void** handle_out) {
    MNK_REQUIRE(hip_graph && handle_out && max_streams >= 1);
#ifdef HIPEMU
    set_error("mnk_replay_create: the stream executor needs the HIP runtime (not available in the CPU emulation)");
    return MNK_EINVAL;
#else
    if (max_streams > MAX_STREAMS) max_streams = MAX_STREAMS;
    hipGraph_t g = (hipGraph_t)hip_graph;
    size_t nn = 0, ne = 0;
    if (hipGraphGetNodes(g, nullptr, &nn) != hipSuccess || nn == 0) {
        set_error("mnk_replay_create: hipGraphGetNodes failed or the graph is empty");
        return MNK_ELAUNCH;
    }
    std::vector<hipGraphNode_t> gn(nn);
    if (hipGraphGetNodes(g, gn.data(), &nn) != hipSuccess) {
        set_error("mnk_replay_create: hipGraphGetNodes failed");
        return MNK_ELAUNCH;
    }
    (void)hipGraphGetEdges(g, nullptr, nullptr, &ne);
    std::vector<hipGraphNode_t> ef(ne), et(ne);
    if (ne && hipGraphGetEdges(g, ef.data(), et.data(), &ne) != hipSuccess) {
        set_error("mnk_replay_create: hipGraphGetEdges failed");
        return MNK_ELAUNCH;
    }
    // node handle -> index in capture order
    std::vector<std::pair<hipGraphNode_t, int>> idx(nn);
    for (size_t i = 0; i < nn; ++i) idx[i] = {gn[i], (int)i};
    std::sort(idx.begin(), idx.end());
    auto find = [&](hipGraphNode_t h) -> int {
        auto it = std::lower_bound(idx.begin(), idx.end(), std::make_pair(h, -1));
        return (it != idx.end() && it->first == h) ? it->second : -1;
    };
    std::vector<std::vector<int>> succ(nn), pred(nn);
    for (size_t e = 0; e < ne; ++e) {
        const int a = find(ef[e]), b = find(et[e]);
        if (a < 0 || b < 0) {
            set_error("mnk_replay_create: an edge names a node that is not in the graph");
            return MNK_ELAUNCH;
        }
        succ[a].push_back(b);
        pred[b].push_back(a);
    }
    // topological order that follows the capture order wherever the edges allow it
    std::vector<int> indeg(nn), order;
    std::priority_queue<int, std::vector<int>, std::greater<int>> ready;
    for (size_t i = 0; i < nn; ++i) {
        indeg[i] = (int)pred[i].size();
        if (!indeg[i]) ready.push((int)i);
    }
    while (!ready.empty()) {
        const int u = ready.top();
        ready.pop();
        order.push_back(u);
        for (int v : succ[u])
            if (--indeg[v] == 0) ready.push(v);
    }
    if (order.size() != nn) {
        set_error("mnk_replay_create: the captured graph has a cycle");
        return MNK_ELAUNCH;
    }
    Program* P = new Program();
    memset(P->info, 0, sizeof(P->info));
    memset(P->side, 0, sizeof(P->side));
    P->nodes.resize(nn);
    std::vector<int> pos(nn);          // capture index -> launch position
    for (size_t i = 0; i < nn; ++i) pos[order[i]] = (int)i;
    for (size_t i = 0; i < nn; ++i) {
        RNode& n = P->nodes[i];
        hipGraphNode_t h = gn[order[i]];
        if (hipGraphNodeGetType(h, &n.type) != hipSuccess) {
            set_error("mnk_replay_create: hipGraphNodeGetType failed");
            delete P;
            return MNK_ELAUNCH;
        }
        hipError_t e = hipSuccess;
        bool ok = true;
        if (n.type == hipGraphNodeTypeKernel) {
            e = hipGraphKernelNodeGetParams(h, &n.k);
            ok = e == hipSuccess && n.k.func && (n.k.kernelParams || n.k.extra);
            P->info[1]++;
        } else if (n.type == hipGraphNodeTypeMemset) {
            e = hipGraphMemsetNodeGetParams(h, &n.ms);
            ok = e == hipSuccess && n.ms.dst;
            P->info[2]++;
        } else if (n.type == hipGraphNodeTypeMemcpy) {
            memset(&n.mc, 0, sizeof(n.mc));
            e = hipGraphMemcpyNodeGetParams(h, &n.mc);
            // (the node of a plain 1-D hipMemcpyAsync does not hand its operands out through this call on ROCm 7: zeros)
            ok = e == hipSuccess && n.mc.srcPtr.ptr && n.mc.dstPtr.ptr && n.mc.extent.width;
            P->info[3]++;
        } else if (n.type != hipGraphNodeTypeEmpty) {
            ok = false;
        }
        if (!ok) {
            // say where: the kernels around the node tell the caller which host statement made it
            const char* before = "";
            const char* after = "";
            for (int p : pred[order[i]]) {
                hipGraphNodeType t;
                hipKernelNodeParams kp;
                if (hipGraphNodeGetType(gn[p], &t) == hipSuccess && t == hipGraphNodeTypeKernel &&
                    hipGraphKernelNodeGetParams(gn[p], &kp) == hipSuccess) {
                    const char* nm = hipKernelNameRefByPtr(kp.func, nullptr);
                    if (nm) before = nm;
                }
            }
            for (int s2 : succ[order[i]]) {
                hipGraphNodeType t;
                hipKernelNodeParams kp;
                if (hipGraphNodeGetType(gn[s2], &t) == hipSuccess && t == hipGraphNodeTypeKernel &&
                    hipGraphKernelNodeGetParams(gn[s2], &kp) == hipSuccess) {
                    const char* nm = hipKernelNameRefByPtr(kp.func, nullptr);
                    if (nm) after = nm;
                }
            }
            set_error("mnk_replay_create: node %d of %d (%s) cannot be replayed by stream launches (%s); it follows kernel "
                      "[%.120s] and precedes [%.120s]", (int)i, (int)nn, type_name(n.type),
                      e == hipSuccess ? "its operands are not readable through the graph API" : hipGetErrorString(e), before,
                      after);
            delete P;
            return MNK_EINVAL;
        }
    }
    // ---- chains -> streams.  A node continues the stream of a predecessor that is still the tail of its stream (the first
    // such predecessor in launch order, stream 0 preferred); a node with no such predecessor starts on the side stream with the
    // fewest nodes so far (one stream: everything is one chain in launch order).
    std::vector<int> tail(max_streams, -1), count(max_streams, 0);
    for (size_t i = 0; i < nn; ++i) {
        RNode& n = P->nodes[i];
        int s = -1;
        if (max_streams == 1) {
            s = 0;
        } else {
            for (int p : pred[order[i]]) {
                const int ps = P->nodes[pos[p]].stream;
                if (tail[ps] == pos[p] && (s < 0 || ps < s)) s = ps;
            }
            if (s < 0) {
                if (pred[order[i]].empty() && tail[0] < 0) {
                    s = 0;
                } else {
                    s = 1;
                    for (int c = 2; c < max_streams; ++c)
                        if (count[c] < count[s]) s = c;
                }
            }
        }
        n.stream = s;
        tail[s] = (int)i;
        count[s]++;
        if (s + 1 > P->nstreams) P->nstreams = s + 1;
    }
    // ---- cross-stream edges -> events.  waited[a][b]: stream a has already waited for the event of launch position
    // waited[a][b] of stream b (a later event of the same stream covers every earlier one)
    std::vector<std::vector<int>> waited(max_streams, std::vector<int>(max_streams, -1));
    auto event_of = [&](int node_pos) -> int {
        RNode& p = P->nodes[node_pos];
        if (p.record < 0) {
            hipEvent_t ev;
            if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return -1;
            P->events.push_back(ev);
            p.record = (int)P->events.size() - 1;
        }
        return p.record;
    };
    for (size_t i = 0; i < nn; ++i) {
        RNode& n = P->nodes[i];
        std::vector<int> ps;
        for (int p : pred[order[i]]) ps.push_back(pos[p]);
        std::sort(ps.begin(), ps.end(), std::greater<int>());          // latest producers first: they cover the earlier ones
        for (int pp : ps) {
            const int sb = P->nodes[pp].stream;
            if (sb == n.stream || waited[n.stream][sb] >= pp) continue;
            const int ev = event_of(pp);
            if (ev < 0) {
                set_error("mnk_replay_create: hipEventCreate failed");
                delete P;
                return MNK_ELAUNCH;
            }
            n.waits.push_back(ev);
            waited[n.stream][sb] = pp;
            P->info[5]++;
        }
    }
    for (int s = 1; s < P->nstreams; ++s) {
        int least = 0, greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
        const hipError_t se = g_side_priority ? hipStreamCreateWithPriority(&P->side[s], hipStreamNonBlocking, least)
                                              : hipStreamCreateWithFlags(&P->side[s], hipStreamNonBlocking);
        if (se != hipSuccess) {
            set_error("mnk_replay_create: hipStreamCreate failed");
            delete P;
            return MNK_ELAUNCH;
        }
        if (tail[s] >= 0) P->tails.push_back(event_of(tail[s]));
    }
    if (P->nstreams > 1 && hipEventCreateWithFlags(&P->begin, hipEventDisableTiming) != hipSuccess) {
        set_error("mnk_replay_create: hipEventCreate failed");
        delete P;
        return MNK_ELAUNCH;
    }
    P->info[0] = (long)nn;
    P->info[4] = P->nstreams;
    P->info[6] = (long)ne;
    for (int s = 1; s < P->nstreams; ++s) P->info[7] += count[s];
    *handle_out = P;
    return MNK_OK;
#endif
}

int mnk_replay_info(void* handle, long* out8) {
    MNK_REQUIRE(handle && out8);
#ifdef HIPEMU
    return MNK_EINVAL;
#else
    memcpy(out8, ((Program*)handle)->info, sizeof(long) * 8);
    return MNK_OK;
#endif
}

int mnk_replay_launch(void* handle, void* stream) {
    MNK_REQUIRE(handle);
#ifdef HIPEMU
    return MNK_EINVAL;
#else
    Program* P = (Program*)handle;
    hipStream_t s0 = (hipStream_t)stream;
    if (P->nstreams > 1) {
        // the side streams start behind everything the caller's stream holds (the previous replay joined them)
        if (hipEventRecord(P->begin, s0) != hipSuccess) {
            set_error("mnk_replay_launch: hipEventRecord failed");
            return MNK_ELAUNCH;
        }
        for (int s = 1; s < P->nstreams; ++s) (void)hipStreamWaitEvent(P->side[s], P->begin, 0);
    }
    for (RNode& n : P->nodes) {
        hipStream_t s = n.stream == 0 ? s0 : P->side[n.stream];
        for (int ev : n.waits)
            if (hipStreamWaitEvent(s, P->events[ev], 0) != hipSuccess) {
                set_error("mnk_replay_launch: hipStreamWaitEvent failed");
                return MNK_ELAUNCH;
            }
        const int rc = launch_node(n, s);
        if (rc != MNK_OK) return rc;
        if (n.record >= 0 && hipEventRecord(P->events[n.record], s) != hipSuccess) {
            set_error("mnk_replay_launch: hipEventRecord failed");
            return MNK_ELAUNCH;
        }
    }
    for (int ev : P->tails) (void)hipStreamWaitEvent(s0, P->events[ev], 0);
    return MNK_OK;
#endif
}

int mnk_replay_destroy(void* handle) {
    MNK_REQUIRE(handle);
#ifdef HIPEMU
    return MNK_EINVAL;
#else
    Program* P = (Program*)handle;
    for (hipEvent_t e : P->events) (void)hipEventDestroy(e);
    if (P->begin) (void)hipEventDestroy(P->begin);
    for (int s = 1; s < MAX_STREAMS; ++s)
        if (P->side[s]) (void)hipStreamDestroy(P->side[s]);
    delete P;
    return MNK_OK;
#endif
}
}

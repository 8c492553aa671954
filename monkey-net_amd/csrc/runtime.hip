// Error reporting + per-kernel event timing for libmonkeynet_hip.so.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "mnk_common.h"

namespace mnk {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---- tuning values of the launch plans ------------------------------------------------------------------------------------
// Every measured default (tile counts, split targets, rows per thread ...) is a `static int g_x = tuning_knob("x", &g_x, d)`
// in the file that uses it.  They are NOT environment switches: tuning scripts set them by name through the C-ABI
// (mnk_set_tuning; tools/plan_tune.py), and A/B visits pass ONE environment variable, MNK_TUNING="name=value,name=value",
// that is read here when a knob registers itself.
struct Knob {
    const char* name;
    int* slot;
};
static std::vector<Knob>& knob_registry() {
    static std::vector<Knob> v;
    return v;
}
int tuning_knob(const char* name, int* slot, int dflt) {
    knob_registry().push_back({name, slot});
    const char* env = getenv("MNK_TUNING");
    const size_t len = strlen(name);
    for (const char* p = env; p && *p;) {
        if (strncmp(p, name, len) == 0 && p[len] == '=') return atoi(p + len + 1);
        p = strchr(p, ',');
        if (p) ++p;
    }
    return dflt;
}

static const char* kNames[K_NUM] = {"conv3x3_igemm", "conv3x3_wgrad", "conv3x3_reduce_pack", "bn_stats",
                                    "bn_act_apply", "bn_act_bwd", "layout", "softmax_kp", "movement_embedding",
                                    "motion_field", "deform", "conv1x1", "adam_pack", "losses"};

struct ProfRec {
    int kid;
    hipEvent_t a, b;
    double work, executed;
};
static bool g_prof_on = false;
static std::vector<ProfRec> g_recs;
static std::vector<hipEvent_t> g_pool;
static uint64_t g_launches[K_NUM];
static double g_ms[K_NUM], g_work[K_NUM], g_executed[K_NUM];

static hipEvent_t get_event() {
    if (!g_pool.empty()) {
        hipEvent_t e = g_pool.back();
        g_pool.pop_back();
        return e;
    }
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

ProfScope::ProfScope(int kid_, hipStream_t stream_, double work, double executed)
    : kid(kid_), stream(stream_), slot(-1), ext(false) {
    if (!g_prof_on) return;
    ProfRec r{kid, get_event(), get_event(), work, executed < 0.0 ? work : executed};
    (void)hipEventRecord(r.a, stream);
    g_recs.push_back(r);
    slot = (int)g_recs.size() - 1;
}
ProfScope::~ProfScope() {
    if (slot >= 0 && !ext) (void)hipEventRecord(g_recs[slot].b, stream);
}
bool ProfScope::kernel_events(hipEvent_t* start, hipEvent_t* stop) {
    *start = *stop = nullptr;
    if (slot < 0) return false;
    ext = true;                       // the launch re-records `a` at the kernel's begin and `b` at its end
    *start = g_recs[slot].a;
    *stop = g_recs[slot].b;
    return true;
}

static void drain() {
    for (ProfRec& r : g_recs) {
        float ms = 0.f;
        (void)hipEventSynchronize(r.b);
        (void)hipEventElapsedTime(&ms, r.a, r.b);
        g_launches[r.kid] += 1;
        g_ms[r.kid] += ms;
        g_work[r.kid] += r.work;
        g_executed[r.kid] += r.executed;
        g_pool.push_back(r.a);
        g_pool.push_back(r.b);
    }
    g_recs.clear();
}

}  // namespace mnk

// ---- small host -> device table upload through kernel arguments -------------------------------------------------------
// Launch tables (descriptor arrays of the multi-layer kernels) change whenever operand addresses change.  A hipMemcpyAsync
// needs page-locked staging to be capturable and becomes a memcpy node of the graph; a kernel that carries the bytes in
// its argument block is an ordinary kernel node, needs no staging buffer, and its bytes are frozen at capture time.
namespace {
struct Blob {
    float4 v[224];            // 3584 bytes of payload per launch (kernel argument blocks are limited to 4 KB)
};
__global__ void __launch_bounds__(256) table_write_kernel(Blob blob, float4* __restrict__ dst, int n16) {
    const int i = threadIdx.x;
    if (i < n16) dst[i] = blob.v[i];
}
}  // namespace

extern "C" {

int mnk_table_upload(const void* host, void* device, size_t bytes, void* stream) {
    MNK_REQUIRE(host && device && bytes > 0 && ((size_t)device % 16) == 0);
    hipStream_t s = (hipStream_t)stream;
    const char* src = (const char*)host;
    char* dst = (char*)device;
    for (size_t off = 0; off < bytes; off += sizeof(Blob)) {
        Blob b;
        const size_t k = bytes - off < sizeof(Blob) ? bytes - off : sizeof(Blob);
        memset(&b, 0, sizeof(b));
        memcpy(&b, src + off, k);
        hipLaunchKernelGGL(table_write_kernel, dim3(1), dim3(256), 0, s, b, (float4*)(dst + off), (int)((k + 15) / 16));
    }
    MNK_LAUNCH_CHECK();
    return MNK_OK;
}

int mnk_set_tuning(const char* name, int value) {
    MNK_REQUIRE(name);
    for (const mnk::Knob& k : mnk::knob_registry())
        if (strcmp(k.name, name) == 0) {
            *k.slot = value;
            return MNK_OK;
        }
    mnk::set_error("mnk_set_tuning: unknown tuning value %s", name);
    return MNK_EINVAL;
}

int mnk_get_tuning(const char* name, int* value) {
    MNK_REQUIRE(name && value);
    for (const mnk::Knob& k : mnk::knob_registry())
        if (strcmp(k.name, name) == 0) {
            *value = *k.slot;
            return MNK_OK;
        }
    mnk::set_error("mnk_get_tuning: unknown tuning value %s", name);
    return MNK_EINVAL;
}

int mnk_version(void) { return 100; }
const char* mnk_last_error(void) { return mnk::g_err; }
int mnk_is_device_build(void) {
#ifdef HIPEMU
    return 0;
#else
    return 1;
#endif
}
int mnk_prof_enable(int on) {
    mnk::g_prof_on = on != 0;
    return MNK_OK;
}
int mnk_prof_reset(void) {
    mnk::drain();
    for (int i = 0; i < mnk::K_NUM; ++i) {
        mnk::g_launches[i] = 0;
        mnk::g_ms[i] = 0;
        mnk::g_work[i] = 0;
        mnk::g_executed[i] = 0;
    }
    return MNK_OK;
}
int mnk_prof_num_kernels(void) { return mnk::K_NUM; }
const char* mnk_prof_kernel_name(int k) { return (k >= 0 && k < mnk::K_NUM) ? mnk::kNames[k] : ""; }
int mnk_prof_query(int k, uint64_t* launches, double* total_ms, double* total_work) {
    MNK_REQUIRE(k >= 0 && k < mnk::K_NUM);
    mnk::drain();
    if (launches) *launches = mnk::g_launches[k];
    if (total_ms) *total_ms = mnk::g_ms[k];
    if (total_work) *total_work = mnk::g_work[k];
    return MNK_OK;
}
int mnk_prof_query_executed(int k, double* executed_work) {
    MNK_REQUIRE(k >= 0 && k < mnk::K_NUM && executed_work);
    mnk::drain();
    *executed_work = mnk::g_executed[k];
    return MNK_OK;
}
}

// hipemu runtime: round-robin fibers, one per emulated GPU thread.  TEST INFRASTRUCTURE ONLY.
// See tests/hipemu/include/hip/hip_runtime.h for the rationale.
// Fiber switch: on x86-64 a dozen instructions (callee-saved registers + stack pointer); glibc's swapcontext makes a
// sigprocmask system call per switch, which dominated kernels with a barrier per K step.  ucontext elsewhere.
#include <hip/hip_runtime.h>
#include <string.h>
#if !defined(__x86_64__)
#include <ucontext.h>
#endif

#include <vector>

namespace hipemu {

dim3 g_threadIdx, g_blockIdx, g_blockDim, g_gridDim;

namespace {
constexpr size_t kStack = 96 * 1024;
constexpr int kMaxWaves = 16;
constexpr int kSlots = 11;      // 0: shuffles / ballots, 1-2: fp32 MFMA operands, 3-10: the eight dwords of a bf16 MFMA

#if defined(__x86_64__)
extern "C" void hipemu_switch(void** save_sp, void* load_sp);
asm(R"(
    .text
    .globl hipemu_switch
    .type hipemu_switch,@function
hipemu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    subq $8, %rsp
    stmxcsr (%rsp)
    fnstcw 4(%rsp)
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    ldmxcsr (%rsp)
    fldcw 4(%rsp)
    addq $8, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
    .size hipemu_switch,.-hipemu_switch
)");
struct Context {
    void* sp = nullptr;
};
void trampoline();
// a fresh fiber: the first switch into it "returns" into trampoline() with a correctly aligned stack
static void make_fiber(Context& c, char* stack, size_t size) {
    uintptr_t top = ((uintptr_t)stack + size) & ~(uintptr_t)15;
    uint64_t* sp = (uint64_t*)(top - 72);          // [csr][r15 r14 r13 r12 rbx rbp][return address][pad]
    memset(sp, 0, 72);
    uint32_t csr;
    uint16_t cw;
    asm volatile("stmxcsr %0" : "=m"(csr));
    asm volatile("fnstcw %0" : "=m"(cw));
    ((uint32_t*)sp)[0] = csr;
    ((uint16_t*)sp)[2] = cw;
    sp[7] = (uint64_t)(uintptr_t)&trampoline;
    c.sp = sp;
}
static inline void switch_ctx(Context& from, Context& to) { hipemu_switch(&from.sp, to.sp); }
[[noreturn]] static inline void jump_ctx(Context& to) {
    void* dummy;
    hipemu_switch(&dummy, to.sp);
    abort();
}
#else
struct Context {
    ucontext_t uc;
};
void trampoline();
static void make_fiber(Context& c, char* stack, size_t size) {
    getcontext(&c.uc);
    c.uc.uc_stack.ss_sp = stack;
    c.uc.uc_stack.ss_size = size;
    c.uc.uc_link = nullptr;
    makecontext(&c.uc, trampoline, 0);
}
static inline void switch_ctx(Context& from, Context& to) { swapcontext(&from.uc, &to.uc); }
[[noreturn]] static inline void jump_ctx(Context& to) {
    setcontext(&to.uc);
    abort();
}
#endif

struct Fiber {
    Context ctx;
    bool done = false;
    unsigned lin = 0;
    dim3 tid;
    unsigned slot_ctr[kSlots] = {};
};

struct BlockState {
    std::vector<Fiber> fibers;
    int nthreads = 0, cur = 0, done_count = 0;
    int bar_count = 0;
    unsigned long bar_gen = 0;
    int wave_size[kMaxWaves];
    int wave_count[kMaxWaves];
    int wave_done[kMaxWaves];
    unsigned long wave_gen[kMaxWaves];
    uint32_t xchg[kMaxWaves][kSlots][2][64];
    Context main_ctx;
    const std::function<void()>* body = nullptr;
};

BlockState* B = nullptr;
std::vector<char> g_stacks;

void switch_to(int next) {
    int prev = B->cur;
    B->cur = next;
    g_threadIdx = B->fibers[next].tid;
    switch_ctx(B->fibers[prev].ctx, B->fibers[next].ctx);
}

void yield() {
    int n = B->nthreads;
    int i = B->cur;
    for (int step = 1; step <= n; ++step) {
        int j = (i + step) % n;
        if (!B->fibers[j].done) {
            if (j != i) switch_to(j);
            return;
        }
    }
}

void release_if_complete_block() {
    if (B->bar_count > 0 && B->bar_count + B->done_count >= B->nthreads) {
        B->bar_count = 0;
        B->bar_gen++;
    }
}
void release_if_complete_wave(int w) {
    if (B->wave_count[w] > 0 && B->wave_count[w] + B->wave_done[w] >= B->wave_size[w]) {
        B->wave_count[w] = 0;
        B->wave_gen[w]++;
    }
}

void trampoline() {
    (*B->body)();
    Fiber& f = B->fibers[B->cur];
    f.done = true;
    B->done_count++;
    int w = f.lin / 64;
    B->wave_done[w]++;
    release_if_complete_block();
    release_if_complete_wave(w);
    // hand over to any live fiber, or back to the launcher
    int n = B->nthreads;
    for (int step = 1; step <= n; ++step) {
        int j = (B->cur + step) % n;
        if (!B->fibers[j].done) {
            int prev = B->cur;
            B->cur = j;
            g_threadIdx = B->fibers[j].tid;
            switch_ctx(B->fibers[prev].ctx, B->fibers[j].ctx);
            abort();  // a finished fiber is never resumed
        }
    }
    jump_ctx(B->main_ctx);
}
}  // namespace

void sync_block() {
    unsigned long g = B->bar_gen;
    B->bar_count++;
    release_if_complete_block();
    while (B->bar_gen == g) yield();
}

void sync_wave() {
    int w = B->fibers[B->cur].lin / 64;
    unsigned long g = B->wave_gen[w];
    B->wave_count[w]++;
    release_if_complete_wave(w);
    while (B->wave_gen[w] == g) yield();
}

unsigned lane_id() { return B->fibers[B->cur].lin & 63u; }

const uint32_t* wave_publish(uint32_t v, int slot) {
    Fiber& f = B->fibers[B->cur];
    int w = f.lin / 64;
    unsigned buf = f.slot_ctr[slot]++ & 1u;
    B->xchg[w][slot][buf][f.lin & 63u] = v;
    sync_wave();
    return B->xchg[w][slot][buf];
}

void launch(dim3 grid, dim3 block, const std::function<void()>& body) {
    int nthreads = (int)(block.x * block.y * block.z);
    if (nthreads <= 0 || nthreads > 64 * kMaxWaves) {
        fprintf(stderr, "hipemu: unsupported block size %d\n", nthreads);
        abort();
    }
    if (g_stacks.size() < (size_t)nthreads * kStack) g_stacks.resize((size_t)nthreads * kStack);
    BlockState st;
    st.body = &body;
    st.nthreads = nthreads;
    st.fibers.resize(nthreads);
    g_blockDim = block;
    g_gridDim = grid;
    BlockState* saved = B;
    B = &st;
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx) {
                g_blockIdx = dim3(bx, by, bz);
                st.cur = 0;
                st.done_count = 0;
                st.bar_count = 0;
                for (int w = 0; w < kMaxWaves; ++w) {
                    int lo = w * 64;
                    st.wave_size[w] = nthreads > lo ? (nthreads - lo < 64 ? nthreads - lo : 64) : 0;
                    st.wave_count[w] = 0;
                    st.wave_done[w] = 0;
                }
                for (int t = 0; t < nthreads; ++t) {
                    Fiber& f = st.fibers[t];
                    f.done = false;
                    f.lin = (unsigned)t;
                    f.tid = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
                    f.slot_ctr[0] = f.slot_ctr[1] = f.slot_ctr[2] = 0;
                    make_fiber(f.ctx, g_stacks.data() + (size_t)t * kStack, kStack);
                }
                g_threadIdx = st.fibers[0].tid;
                switch_ctx(st.main_ctx, st.fibers[0].ctx);
            }
    B = saved;
}

}  // namespace hipemu

// ---- "device" allocations that another process can map (the peer-to-peer mailboxes): named POSIX shared memory ---------------------
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <unistd.h>

#include <map>
#include <string>

namespace {
struct ShmBlock {
    std::string name;      // empty: mapped from another process (not ours to unlink)
    size_t bytes;
};
struct ShmRegistry : std::map<void*, ShmBlock> {
    ~ShmRegistry() {           // a process that exits without mnk_p2p_destroy must not leave its objects in /dev/shm
        for (auto& kv : *this)
            if (!kv.second.name.empty()) shm_unlink(kv.second.name.c_str());
    }
};
ShmRegistry& shm_blocks() {
    static ShmRegistry m;
    return m;
}
int g_shm_counter = 0;
}  // namespace

hipError_t hipExtMallocWithFlags(void** p, size_t n, unsigned) {
    if (!p || !n) return hipErrorInvalidValue;
    char name[48];
    snprintf(name, sizeof(name), "/hipemu_%d_%d", (int)getpid(), g_shm_counter++);
    const int fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0) return hipErrorInvalidValue;
    if (ftruncate(fd, (off_t)n) != 0) {
        close(fd);
        shm_unlink(name);
        return hipErrorInvalidValue;
    }
    void* m = mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) {
        shm_unlink(name);
        return hipErrorInvalidValue;
    }
    shm_blocks()[m] = ShmBlock{name, n};
    *p = m;
    return hipSuccess;
}

hipError_t hipFree(void* p) {
    auto it = shm_blocks().find(p);
    if (it == shm_blocks().end()) {
        free(p);
        return hipSuccess;
    }
    munmap(p, it->second.bytes);
    if (!it->second.name.empty()) shm_unlink(it->second.name.c_str());
    shm_blocks().erase(it);
    return hipSuccess;
}

hipError_t hipIpcGetMemHandle(hipIpcMemHandle_t* h, void* p) {
    auto it = shm_blocks().find(p);
    if (!h || it == shm_blocks().end() || it->second.name.empty()) return hipErrorInvalidValue;
    memset(h, 0, sizeof(*h));
    snprintf(h->reserved, 48, "%s", it->second.name.c_str());
    unsigned long long bytes = it->second.bytes;
    memcpy(h->reserved + 48, &bytes, 8);
    return hipSuccess;
}

hipError_t hipIpcOpenMemHandle(void** p, hipIpcMemHandle_t h, unsigned) {
    if (!p || h.reserved[0] != '/') return hipErrorInvalidValue;
    unsigned long long bytes = 0;
    memcpy(&bytes, h.reserved + 48, 8);
    h.reserved[47] = 0;
    const int fd = shm_open(h.reserved, O_RDWR, 0600);
    if (fd < 0 || !bytes) return hipErrorInvalidValue;
    void* m = mmap(nullptr, (size_t)bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) return hipErrorInvalidValue;
    shm_blocks()[m] = ShmBlock{std::string(), (size_t)bytes};
    *p = m;
    return hipSuccess;
}

hipError_t hipIpcCloseMemHandle(void* p) {
    auto it = shm_blocks().find(p);
    if (it == shm_blocks().end() || !it->second.name.empty()) return hipErrorInvalidValue;
    munmap(p, it->second.bytes);
    shm_blocks().erase(it);
    return hipSuccess;
}

void hipemu_yield() { sched_yield(); }

// hipemu runtime: round-robin ucontext fibers, one per emulated GPU thread.  TEST INFRASTRUCTURE ONLY.
// See tests/hipemu/include/hip/hip_runtime.h for the rationale.
#include <hip/hip_runtime.h>
#include <ucontext.h>

#include <vector>

namespace hipemu {

dim3 g_threadIdx, g_blockIdx, g_blockDim, g_gridDim;

namespace {
constexpr size_t kStack = 96 * 1024;
constexpr int kMaxWaves = 16;
constexpr int kSlots = 3;

struct Fiber {
    ucontext_t ctx;
    bool done = false;
    unsigned lin = 0;
    dim3 tid;
    unsigned slot_ctr[kSlots] = {0, 0, 0};
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
    ucontext_t main_ctx;
    const std::function<void()>* body = nullptr;
};

BlockState* B = nullptr;
std::vector<char> g_stacks;

void switch_to(int next) {
    int prev = B->cur;
    B->cur = next;
    g_threadIdx = B->fibers[next].tid;
    swapcontext(&B->fibers[prev].ctx, &B->fibers[next].ctx);
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
            swapcontext(&B->fibers[prev].ctx, &B->fibers[j].ctx);
            abort();  // a finished fiber is never resumed
        }
    }
    setcontext(&B->main_ctx);
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
                    getcontext(&f.ctx);
                    f.ctx.uc_stack.ss_sp = g_stacks.data() + (size_t)t * kStack;
                    f.ctx.uc_stack.ss_size = kStack;
                    f.ctx.uc_link = nullptr;
                    makecontext(&f.ctx, trampoline, 0);
                }
                g_threadIdx = st.fibers[0].tid;
                swapcontext(&st.main_ctx, &st.fibers[0].ctx);
            }
    B = saved;
}

}  // namespace hipemu

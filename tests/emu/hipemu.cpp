// tests/emu/hipemu.cpp -- fiber scheduler behind tests/emu/hip/hip_runtime.h (TEST INFRASTRUCTURE ONLY).
#include "hip/hip_runtime.h"

#include <sys/mman.h>

namespace hipemu {

static State g_host_state;          // used outside kernels (blockIdx etc. are meaningless there)
State* cur = &g_host_state;

namespace {
constexpr size_t STACK = 256 * 1024;
struct Fiber {
    ucontext_t ctx;
    char* stack = nullptr;
    State st;
    YieldKind kind = Y_NONE;
    bool done = false;
    unsigned long long seq = 0;   // wave-op sequence number
};
std::vector<Fiber*> g_pool;
ucontext_t g_sched;
Fiber* g_running = nullptr;
const std::function<void()>* g_body = nullptr;

void trampoline()
{
    (*g_body)();
    g_running->done = true;
    g_running->kind = Y_DONE;
    swapcontext(&g_running->ctx, &g_sched);
}

Fiber* get_fiber(size_t i)
{
    while (g_pool.size() <= i) {
        Fiber* f = new Fiber();
        f->stack = (char*)mmap(nullptr, STACK, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_STACK, -1, 0);
        if (f->stack == MAP_FAILED) { perror("hipemu: mmap"); abort(); }
        g_pool.push_back(f);
    }
    return g_pool[i];
}

void resume(Fiber* f)
{
    g_running = f;
    cur = &f->st;
    swapcontext(&g_sched, &f->ctx);
    cur = &g_host_state;
    g_running = nullptr;
}
}  // namespace

void yield(YieldKind k)
{
    Fiber* f = g_running;
    if (!f) return;     // called from host code: no-op
    f->kind = k;
    swapcontext(&f->ctx, &g_sched);
}

// Publishes v for this lane, yields until every live lane of the wave has published, then returns
// all 64 values plus the mask of lanes that took part.
unsigned long long wave_exchange(unsigned long long v, unsigned long long* all, unsigned long long* active)
{
    Fiber* f = g_running;
    if (!f) { fprintf(stderr, "hipemu: wave op outside a kernel\n"); abort(); }
    unsigned long long seq = ++f->seq;
    int par = (int)(seq & 1);
    // scratch lives in lane 0's State of this wave: find it through the pool layout
    Fiber* w0 = g_pool[(size_t)(f->st.linear - f->st.lane)];
    w0->st.scratch[par][f->st.lane] = v;
    w0->st.stamp[par][f->st.lane] = seq;
    yield(Y_WAVE);
    unsigned long long act = 0;
    for (int i = 0; i < 64; i++) {
        all[i] = w0->st.scratch[par][i];
        if (w0->st.stamp[par][i] == seq) act |= 1ull << i;
    }
    *active = act;
    return v;
}

void run_grid(dim3 grid, dim3 block, const std::function<void()>& body)
{
    if (g_running) { fprintf(stderr, "hipemu: nested launch\n"); abort(); }
    const size_t nthreads = (size_t)block.x * block.y * block.z;
    const size_t nwaves = (nthreads + 63) / 64;
    g_body = &body;
    for (unsigned bz = 0; bz < grid.z; bz++)
    for (unsigned by = 0; by < grid.y; by++)
    for (unsigned bx = 0; bx < grid.x; bx++) {
        for (size_t t = 0; t < nthreads; t++) {
            Fiber* f = get_fiber(t);
            f->done = false; f->kind = Y_NONE; f->seq = 0;
            f->st.tIdx = uint3{(unsigned)(t % block.x), (unsigned)((t / block.x) % block.y), (unsigned)(t / ((size_t)block.x * block.y))};
            f->st.bIdx = uint3{bx, by, bz};
            f->st.bDim = block; f->st.gDim = grid;
            f->st.linear = (int)t; f->st.lane = (int)(t % 64); f->st.wave = (int)(t / 64);
            memset(f->st.stamp, 0, sizeof(f->st.stamp));
            getcontext(&f->ctx);
            f->ctx.uc_stack.ss_sp = f->stack;
            f->ctx.uc_stack.ss_size = STACK;
            f->ctx.uc_link = nullptr;
            makecontext(&f->ctx, trampoline, 0);
        }
        std::vector<char> blocked(nwaves, 0), finished(nwaves, 0);
        size_t nfinished = 0;
        while (nfinished < nwaves) {
            bool progressed = false;
            for (size_t w = 0; w < nwaves; w++) {
                if (finished[w] || blocked[w]) continue;
                progressed = true;
                int kinds[4] = {0, 0, 0, 0};
                size_t lo = w * 64, hi = std::min(nthreads, lo + 64);
                for (size_t t = lo; t < hi; t++) {
                    Fiber* f = g_pool[t];
                    if (f->done) continue;
                    resume(f);
                    kinds[f->kind]++;
                }
                if (kinds[Y_WAVE] && kinds[Y_BLOCK]) {
                    fprintf(stderr, "hipemu: divergent wave (some lanes at a wave op, others at __syncthreads) in block (%u,%u,%u)\n", bx, by, bz);
                    abort();
                }
                if (kinds[Y_BLOCK]) blocked[w] = 1;
                else if (!kinds[Y_WAVE]) { finished[w] = 1; nfinished++; }
            }
            bool all_blocked = true;
            size_t nblocked = 0;
            for (size_t w = 0; w < nwaves; w++) { if (!finished[w] && !blocked[w]) all_blocked = false; nblocked += blocked[w]; }
            if (all_blocked && nblocked) { std::fill(blocked.begin(), blocked.end(), 0); progressed = true; }
            if (!progressed) { fprintf(stderr, "hipemu: deadlock\n"); abort(); }
        }
    }
    g_body = nullptr;
}

}  // namespace hipemu

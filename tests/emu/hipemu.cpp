// tests/emu/hipemu.cpp -- fiber scheduler behind tests/emu/hip/hip_runtime.h (TEST INFRASTRUCTURE ONLY).
#include "hip/hip_runtime.h"

#include <sys/mman.h>

#include <pthread.h>

#include <mutex>

namespace hipemu {

static State g_host_state;          // used outside kernels (blockIdx etc. are meaningless there)
State* cur = &g_host_state;

namespace {
constexpr size_t STACK = 256 * 1024;
struct Fiber {
    void* sp = nullptr;           // saved stack pointer while the fiber is not running
    char* stack = nullptr;
    State st;
    YieldKind kind = Y_NONE;
    bool done = false;
    unsigned long long seq = 0;   // wave-op sequence number
};
std::vector<Fiber*> g_pool;
void* g_sched_sp = nullptr;      // the scheduler's saved stack pointer while a fiber runs
Fiber* g_running = nullptr;
pthread_t g_owner;               // the host thread whose run_grid() is executing fibers (valid while g_running != nullptr)
thread_local bool t_in_launch = false;
// `cur` and `g_running` are process globals: a host thread that is NOT the launching one (the JPEG host_job threads run beside
// kernels of the launching thread) must never switch onto a fiber
static inline bool on_owner_thread() { return g_running && pthread_equal(g_owner, pthread_self()); }
const std::function<void()>* g_body = nullptr;

// Context switch in user space: callee-saved registers and the stack pointer only.  (glibc's swapcontext also saves and restores
// the signal mask -- two system calls per switch, and a workgroup of 1024 fibers switches thousands of times per barrier: a third
// of the emulated suite's time went to the kernel.)  x86-64 System V only, like the container.
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
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
    .size hipemu_switch, .-hipemu_switch
)");
#if !defined(__x86_64__)
#error "tests/emu/hipemu.cpp: the hand-written context switch is x86-64 System V only"
#endif

extern "C" void hipemu_trampoline()
{
    (*g_body)();
    g_running->done = true;
    g_running->kind = Y_DONE;
    hipemu_switch(&g_running->sp, g_sched_sp);
    abort();                       // a finished fiber is never resumed
}

// A fresh fiber: its first hipemu_switch pops six zeroed registers and "returns" into the trampoline with the stack aligned as at a call.
void arm_fiber(Fiber* f)
{
    void** sp = reinterpret_cast<void**>(f->stack + STACK);          // 16-byte aligned (mmap)
    *--sp = nullptr;                                                  // the trampoline's (never used) return address: rsp = 8 mod 16 at entry
    *--sp = reinterpret_cast<void*>(&hipemu_trampoline);
    for (int i = 0; i < 6; i++) *--sp = nullptr;
    f->sp = sp;
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
    hipemu_switch(&g_sched_sp, f->sp);
    cur = &g_host_state;
    g_running = nullptr;
}
}  // namespace

void yield(YieldKind k)
{
    if (!on_owner_thread()) return;     // called from host code (this or another thread): no-op
    Fiber* f = g_running;
    f->kind = k;
    hipemu_switch(&f->sp, g_sched_sp);
}

// Publishes v for this lane, yields until every live lane of the wave has published, then returns
// all 64 values plus the mask of lanes that took part.
unsigned long long wave_exchange(unsigned long long v, unsigned long long* all, unsigned long long* active)
{
    if (!on_owner_thread()) { fprintf(stderr, "hipemu: wave op outside a kernel\n"); abort(); }
    Fiber* f = g_running;
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

// Kernels of several host threads (StreamedDetector: one context per thread) run ONE AT A TIME: the fibers, the scheduler and -- above
// all -- the kernels' __shared__ arrays (function-local statics here) exist once per process.
static std::mutex g_launch_mutex;

void run_grid(dim3 grid, dim3 block, const std::function<void()>& body)
{
    if (t_in_launch) { fprintf(stderr, "hipemu: nested launch\n"); abort(); }      // before the (non-recursive) lock: it would deadlock there
    std::lock_guard<std::mutex> one_kernel_at_a_time(g_launch_mutex);
    struct InLaunch { InLaunch() { t_in_launch = true; } ~InLaunch() { t_in_launch = false; } } in_launch;
    g_owner = pthread_self();
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
            arm_fiber(f);
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

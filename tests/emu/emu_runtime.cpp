// TEST INFRASTRUCTURE ONLY -- fiber scheduler behind tests/emu/hip/hip_runtime.h.
#include <hip/hip_runtime.h>

#include <sys/mman.h>
#if !defined(__x86_64__)
#include <ucontext.h>
#endif

#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

// AddressSanitizer build (tests/emu/build_emu.py build(asan=True), tools/emu_asan.py): the work-item switches are announced
// to the sanitizer, and a stack is unpoisoned before it is reused for the next workgroup's work-item.
#if defined(__has_feature)
#if __has_feature(address_sanitizer)
#include <sanitizer/asan_interface.h>
#include <sanitizer/common_interface_defs.h>
#define LECO_EMU_ASAN 1
#endif
#endif
#ifndef LECO_EMU_ASAN
#define LECO_EMU_ASAN 0
#endif
// ThreadSanitizer build (build(tsan=True), tools/emu_asan.py --tsan): every work-item slot is a sanitizer "fiber" (created once
// per OS thread, reused by the workgroups that thread runs); switches synchronise, so accesses inside a workgroup are ordered
// (they are sequential here) and what is checked is workgroup against workgroup -- the pool's OS threads.
#if defined(__has_feature)
#if __has_feature(thread_sanitizer)
#include <sanitizer/tsan_interface.h>
#define LECO_EMU_TSAN 1
#endif
#endif
#ifndef LECO_EMU_TSAN
#define LECO_EMU_TSAN 0
#endif

namespace emu {
thread_local emu_uint3 t_idx, b_idx;
thread_local dim3 b_dim, g_dim;

namespace {
constexpr size_t kStack = 256 * 1024;
constexpr int kMaxThreads = 1024;

// Work-item switch.  glibc's swapcontext saves and restores the signal mask with a system call on every switch,
// and a kernel with a few hundred barriers / cross-lane gathers per workgroup switches millions of times per launch:
// on x86-64 the switch is the six callee-saved registers and the stack pointer, nothing else (no work-item changes
// the signal mask or the floating-point control words).  Other hosts keep ucontext.
#if defined(__x86_64__)
struct Ctx { void* sp; };
extern "C" void leco_emu_switch(Ctx* from, Ctx* to);
asm(R"(
    .text
    .globl leco_emu_switch
    .type leco_emu_switch,@function
leco_emu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq (%rsi), %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
    .size leco_emu_switch, .-leco_emu_switch
)");
#else
struct Ctx { ucontext_t uc; };
static inline void leco_emu_switch(Ctx* from, Ctx* to) { swapcontext(&from->uc, &to->uc); }
#endif

struct Fiber {
    Ctx ctx;
    bool done;
    void* fake = nullptr;           // sanitizer's fake-stack handle while the work-item is switched out
    void* tsan = nullptr;           // ThreadSanitizer fiber of this work-item slot
};

struct Runner {
    Ctx sched;
    void* sched_fake = nullptr;     // sanitizer: the scheduler's fake stack / real stack bounds (learnt by the first work-item)
    const void* sched_bottom = nullptr;
    size_t sched_size = 0;
    void* sched_tsan = nullptr;
    Fiber fibers[kMaxThreads];
    char* stacks = nullptr;
    int n = 0, cur = 0;
    unsigned long events = 0;        // completed rendezvous + finished work-items (progress, for the greedy schedules)
    unsigned long rng = 0x9E3779B97F4A7C15ul;
    int block_arrived = 0;
    int finished = 0;                // work-items of this block that have returned: a terminated wave no longer counts at s_barrier
    unsigned block_gen = 0;
    int wave_arrived[kMaxThreads / 64] = {};
    unsigned wave_gen[kMaxThreads / 64] = {};
    unsigned char* slots = nullptr;  // [waves][2][64*kSlot]
    const std::function<void()>* body = nullptr;
    dim3 bdim;

    void ensure() {
        if (!stacks) {
            stacks = (char*)mmap(nullptr, kStack * kMaxThreads, PROT_READ | PROT_WRITE,
                                 MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
            slots = (unsigned char*)calloc((kMaxThreads / 64) * 2 * 64 * kSlot, 1);
            if (stacks == MAP_FAILED || !slots) { perror("emu alloc"); abort(); }
        }
    }
};
thread_local Runner* tl_runner = nullptr;

// Work-item schedules.  Any interleaving of the waves of a workgroup between its barriers is legal on the hardware, so
// every kernel must give the same result under each of them: round-robin (default: the waves advance in near lockstep, one
// rendezvous per sweep), reverse, random (random wave order, a wave may run a few rendezvous ahead) and greedy /
// greedy_reverse (one wave runs as far as it can -- to the next block barrier -- before the next one moves at all: the
// extreme skew that exposes an LDS buffer reused without a barrier).  LECO_EMU_SCHED selects.
enum { kRoundRobin = 0, kReverse, kRandom, kGreedy, kGreedyReverse };
int sched_mode() {
    static const int m = [] {
        const char* e = getenv("LECO_EMU_SCHED");
        if (!e || !*e || !strcmp(e, "rr")) return (int)kRoundRobin;
        if (!strcmp(e, "reverse")) return (int)kReverse;
        if (!strcmp(e, "random")) return (int)kRandom;
        if (!strcmp(e, "greedy")) return (int)kGreedy;
        if (!strcmp(e, "greedy_reverse")) return (int)kGreedyReverse;
        fprintf(stderr, "emu: unknown LECO_EMU_SCHED=%s\n", e);
        abort();
    }();
    return m;
}

void trampoline() {
    Runner* r = tl_runner;
#if LECO_EMU_ASAN
    __sanitizer_finish_switch_fiber(nullptr, &r->sched_bottom, &r->sched_size);
#endif
    (*r->body)();
    r->fibers[r->cur].done = true;
    ++r->events;
    ++r->finished;
    if (r->block_arrived > 0 && r->block_arrived == r->n - r->finished) {     // the others were only waiting for this one
        r->block_arrived = 0;
        ++r->block_gen;
    }
#if LECO_EMU_ASAN
    __sanitizer_start_switch_fiber(nullptr, r->sched_bottom, r->sched_size);      // nullptr: this work-item's stack dies
#endif
#if LECO_EMU_TSAN
    __tsan_switch_to_fiber(r->sched_tsan, 0);
#endif
#if defined(__x86_64__)
    leco_emu_switch(&r->fibers[r->cur].ctx, &r->sched);     // a finished work-item is never resumed
    __builtin_trap();
#endif
    // ucontext: returns to uc_link (scheduler)
}

void make_fiber(Runner* r, int i) {
    Fiber& f = r->fibers[i];
    f.done = false;
    char* top = r->stacks + (size_t)(i + 1) * kStack;       // 16-byte aligned (kStack is, the mapping is)
#if LECO_EMU_ASAN
    __asan_unpoison_memory_region(top - kStack, kStack);
#endif
#if defined(__x86_64__)
    // What leco_emu_switch pops on the first switch in: six zeroed callee-saved registers, then `ret` into the
    // trampoline with the stack as a `call` would have left it (rsp % 16 == 8; the slot above is a null return address).
    void** sp = (void**)top - 8;
    for (int k = 0; k < 6; ++k) sp[k] = nullptr;
    sp[6] = (void*)&trampoline;
    sp[7] = nullptr;
    f.ctx.sp = sp;
#else
    getcontext(&f.ctx.uc);
    f.ctx.uc.uc_stack.ss_sp = top - kStack;
    f.ctx.uc.uc_stack.ss_size = kStack;
    f.ctx.uc.uc_link = &r->sched.uc;
    makecontext(&f.ctx.uc, trampoline, 0);
#endif
}

void set_tid(Runner* r, int i) {
    t_idx.x = i % r->bdim.x;
    t_idx.y = (i / r->bdim.x) % r->bdim.y;
    t_idx.z = i / (r->bdim.x * r->bdim.y);
}

void yield_fiber() {
    Runner* r = tl_runner;
    int me = r->cur;
#if LECO_EMU_ASAN
    __sanitizer_start_switch_fiber(&r->fibers[me].fake, r->sched_bottom, r->sched_size);
#endif
#if LECO_EMU_TSAN
    __tsan_switch_to_fiber(r->sched_tsan, 0);
#endif
    leco_emu_switch(&r->fibers[me].ctx, &r->sched);
#if LECO_EMU_ASAN
    __sanitizer_finish_switch_fiber(r->fibers[me].fake, nullptr, nullptr);
#endif
}

thread_local unsigned long tl_block_serial = 0;

void run_block(Runner* r, dim3 block, const std::function<void()>& body) {
    r->ensure();
    ++tl_block_serial;
    r->n = block.x * block.y * block.z;
    if (r->n > kMaxThreads) { fprintf(stderr, "emu: block too large\n"); abort(); }
    r->bdim = block;
    r->body = &body;
    r->block_arrived = 0;
    r->finished = 0;
    for (int w = 0; w < kMaxThreads / 64; ++w) r->wave_arrived[w] = 0;
    for (int i = 0; i < r->n; ++i) make_fiber(r, i);
#if LECO_EMU_TSAN
    r->sched_tsan = __tsan_get_current_fiber();
    for (int i = 0; i < r->n; ++i)
        if (!r->fibers[i].tsan) r->fibers[i].tsan = __tsan_create_fiber(0);
#endif
    int live = r->n;
    long spins = 0;
    auto resume = [&](int i) {
        if (r->fibers[i].done) return;
        r->cur = i;
        set_tid(r, i);
#if LECO_EMU_ASAN
        __sanitizer_start_switch_fiber(&r->sched_fake, r->stacks + (size_t)i * kStack, kStack);
#endif
#if LECO_EMU_TSAN
        __tsan_switch_to_fiber(r->fibers[i].tsan, 0);
#endif
        leco_emu_switch(&r->sched, &r->fibers[i].ctx);
#if LECO_EMU_ASAN
        __sanitizer_finish_switch_fiber(r->sched_fake, nullptr, nullptr);
#endif
        if (r->fibers[i].done) --live;
    };
    const int mode = sched_mode();
    const int nw = (r->n + 63) / 64;
    auto sweep_wave = [&](int w, bool down) {
        int lo = w * 64, hi = lo + 64 < r->n ? lo + 64 : r->n;
        if (down) for (int i = hi - 1; i >= lo; --i) resume(i);
        else for (int i = lo; i < hi; ++i) resume(i);
    };
    while (live > 0) {
        if (mode == kRoundRobin) {
            for (int i = 0; i < r->n; ++i) resume(i);
        } else if (mode == kReverse) {
            for (int i = r->n - 1; i >= 0; --i) resume(i);
        } else if (mode == kRandom) {
            int order[kMaxThreads / 64];
            for (int w = 0; w < nw; ++w) order[w] = w;
            for (int w = nw - 1; w > 0; --w) {
                r->rng ^= r->rng << 13; r->rng ^= r->rng >> 7; r->rng ^= r->rng << 17;
                int j = (int)(r->rng % (unsigned long)(w + 1));
                int t = order[w]; order[w] = order[j]; order[j] = t;
            }
            for (int w = 0; w < nw; ++w) {
                r->rng ^= r->rng << 13; r->rng ^= r->rng >> 7; r->rng ^= r->rng << 17;
                int reps = 1 + (int)((r->rng >> 20) % 4);       // a wave may get a few rendezvous ahead of the others
                for (int k = 0; k < reps; ++k) sweep_wave(order[w], (r->rng >> (8 + k)) & 1);
            }
        } else {    // greedy: one wave runs until it can only wait for the others (a block barrier), then the next
            for (int k = 0; k < nw; ++k) {
                int w = mode == kGreedy ? k : nw - 1 - k;
                unsigned long before;
                do { before = r->events; sweep_wave(w, false); } while (r->events != before);
            }
        }
        if (++spins > 200000000L) { fprintf(stderr, "emu: deadlock (divergent barrier?)\n"); abort(); }
    }
}

// Workgroup dispatch order (LECO_EMU_BLOCKS=reverse|random; default: ascending linear index, as the hardware dispatches).
// Completion order is arbitrary on the hardware, and kernels that pass data between workgroups (split-K "last arriver
// reduces", statistics summed with atomics) must not depend on it: `reverse` starts with the last workgroup, `random` walks
// a fixed pseudo-random permutation (b * p + 7 mod total, p coprime to total).
long block_order(long b, long total) {
    static const int mode = [] {
        const char* e = getenv("LECO_EMU_BLOCKS");
        if (!e || !*e || !strcmp(e, "linear")) return 0;
        if (!strcmp(e, "reverse")) return 1;
        if (!strcmp(e, "random")) return 2;
        fprintf(stderr, "emu: unknown LECO_EMU_BLOCKS=%s\n", e);
        abort();
    }();
    if (mode == 0 || total < 2) return b;
    if (mode == 1) return total - 1 - b;
    auto gcd = [](long a, long c) { while (c) { long t = a % c; a = c; c = t; } return a; };
    long p = (long)(total * 0.6180339887) | 1;
    while (gcd(p, total) != 1) p += 2;
    return (long)(((unsigned long)b * (unsigned long)p + 7ul) % (unsigned long)total);
}

// ---- persistent worker pool ------------------------------------------------
struct Pool {
    std::vector<std::thread> th;
    std::mutex mu;
    std::condition_variable cv, cv_done;
    unsigned long job_id = 0;
    dim3 grid, block;
    const std::function<void()>* body = nullptr;
    std::atomic<long> next{0};
    long total = 0;
    int active = 0;
    bool stop = false;

    Pool() {
        int n = (int)std::thread::hardware_concurrency();
        const char* e = getenv("LECO_EMU_THREADS");
        if (e) n = atoi(e);
        if (n < 1) n = 1;
        for (int i = 0; i < n; ++i) th.emplace_back([this] { worker(); });
    }
    ~Pool() {
        { std::lock_guard<std::mutex> l(mu); stop = true; }
        cv.notify_all();
        for (auto& t : th) t.join();
    }
    void worker() {
        Runner* r = new Runner();
        tl_runner = r;
        unsigned long seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> l(mu);
                cv.wait(l, [&] { return stop || job_id != seen; });
                if (stop) return;
                seen = job_id;
            }
            for (;;) {
                long b = next.fetch_add(1);
                if (b >= total) break;
                b = block_order(b, total);
                b_idx.x = b % grid.x;
                b_idx.y = (b / grid.x) % grid.y;
                b_idx.z = b / ((long)grid.x * grid.y);
                b_dim = block;
                g_dim = grid;
                run_block(r, block, *body);
            }
            {
                std::lock_guard<std::mutex> l(mu);
                if (--active == 0) cv_done.notify_all();
            }
        }
    }
    void run(dim3 g, dim3 b, const std::function<void()>& fn) {
        std::unique_lock<std::mutex> l(mu);
        grid = g; block = b; body = &fn;
        total = (long)g.x * g.y * g.z;
        next = 0;
        active = (int)th.size();
        ++job_id;
        cv.notify_all();
        cv_done.wait(l, [&] { return active == 0; });
    }
};
Pool& pool() { static Pool p; return p; }
}  // namespace

void sync_block() {
    Runner* r = tl_runner;
    unsigned g = r->block_gen;
    if (++r->block_arrived == r->n - r->finished) {
        r->block_arrived = 0;
        r->block_gen = g + 1;
        ++r->events;
    } else {
        while (r->block_gen == g) yield_fiber();
    }
}

int lane() { return tl_runner->cur & 63; }

unsigned long block_serial() { return tl_block_serial; }

bool lds_poison() {
    static const bool on = [] { const char* e = getenv("LECO_EMU_LDS"); return e && !strcmp(e, "poison"); }();
    return on;
}

const unsigned char* wave_gather(const void* in, int bytes) {
    Runner* r = tl_runner;
    int w = r->cur >> 6, ln = r->cur & 63;
    int wave_n = r->n - w * 64 < 64 ? r->n - w * 64 : 64;
    unsigned g = r->wave_gen[w];
    unsigned char* buf = r->slots + ((size_t)(w * 2 + (g & 1)) * 64) * kSlot;
    if (bytes > kSlot) { fprintf(stderr, "emu: slot overflow\n"); abort(); }
    memcpy(buf + (size_t)ln * kSlot, in, bytes);
    if (++r->wave_arrived[w] == wave_n) {
        r->wave_arrived[w] = 0;
        r->wave_gen[w] = g + 1;
        ++r->events;
    } else {
        while (r->wave_gen[w] == g) yield_fiber();
    }
    return buf;
}

void launch(dim3 grid, dim3 block, const std::function<void()>& body) { pool().run(grid, block, body); }
}  // namespace emu

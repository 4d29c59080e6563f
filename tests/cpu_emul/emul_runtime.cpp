// TEST-ONLY runtime behind tests/cpu_emul/hip/hip_runtime.h (see the header comment there).
//
// Execution model: a workgroup's GPU threads are cooperative fibers (ucontext) on ONE OS
// thread, scheduled round-robin; a fiber yields only at a barrier (__syncthreads or the
// 64-lane wave barrier used by the MFMA / shuffle emulation).  Workgroups are distributed over
// a few OS worker threads; LDS (`alsep_smem`) and threadIdx/blockIdx are thread_local, so each
// worker has its own.  A round in which no fiber makes progress is reported as a barrier
// deadlock (divergent barrier / partial-wave MFMA) and aborts.
#include <hip/hip_runtime.h>

#include <ucontext.h>
#include <unistd.h>

#include <atomic>
#include <condition_variable>
#include <chrono>
#include <mutex>
#include <thread>
#include <vector>

#if defined(__has_feature)
#if __has_feature(address_sanitizer)
#define EMUL_ASAN 1
#include <sanitizer/common_interface_defs.h>
#endif
#endif

thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;
alignas(256) thread_local char alsep_smem[160 * 1024];

#if !defined(EMUL_ASAN) && defined(__x86_64__)
// Plain builds switch fibers with a hand-rolled callee-saved-register swap: glibc's swapcontext makes two rt_sigprocmask system calls
// per switch, and a 512-thread workgroup switches ~10^5 times per barrier-heavy kernel (the CPU suite spent more time in the kernel than
// in user code).  The AddressSanitizer build keeps ucontext (its fiber annotations are written against it).
#define EMUL_FAST_SWITCH 1
extern "C" void emul_swap(void** save_sp, void* load_sp);
asm(R"(
.text
.globl emul_swap
.type emul_swap,@function
emul_swap:
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
.size emul_swap, .-emul_swap
)");
#endif

namespace {
constexpr size_t kStack = 256 * 1024;

struct Fiber {
    ucontext_t ctx;
    void* sp = nullptr;                                      // EMUL_FAST_SWITCH: saved stack pointer
    char* stack = nullptr;
    bool done = false;
    dim3 tidx;
    void* fake = nullptr;
};

struct Worker {
    ucontext_t sched;
    void* sched_sp = nullptr;
    std::vector<Fiber> fibers;
    char* stacks = nullptr;                                  // nt * kStack bytes, NOT zero-filled (a vector<char> of 128 MB per worker and
    size_t stacks_cap = 0;                                   // launch cost more than the kernels), kept by the pool's thread across launches
    const std::function<void()>* body = nullptr;
    int nt = 0, cur = 0, alive = 0;
    int block_arrived = 0;
    unsigned block_gen = 0;
    std::vector<int> wave_arrived, wave_alive;
    std::vector<unsigned> wave_gen;
    std::vector<char> slabs;
    bool progress = false;
    void* sched_fake = nullptr;
    const void* sched_bottom = nullptr;
    size_t sched_size = 0;
};
thread_local Worker* tw = nullptr;
std::mutex g_atomic_mu;
std::atomic<int> g_last_error{hipSuccess};

void switch_to_sched(Worker& w, Fiber& f, bool dying) {
#ifdef EMUL_ASAN
    __sanitizer_start_switch_fiber(dying ? nullptr : &f.fake, w.sched_bottom, w.sched_size);
#endif
    (void)dying;
#ifdef EMUL_FAST_SWITCH
    emul_swap(&f.sp, w.sched_sp);
#else
    swapcontext(&f.ctx, &w.sched);
#endif
#ifdef EMUL_ASAN
    __sanitizer_finish_switch_fiber(f.fake, &w.sched_bottom, &w.sched_size);
#endif
}

void yield() {
    Worker& w = *tw;
    switch_to_sched(w, w.fibers[w.cur], false);
}

void fiber_main() {
    Worker& w = *tw;
#ifdef EMUL_ASAN
    __sanitizer_finish_switch_fiber(nullptr, &w.sched_bottom, &w.sched_size);
#endif
    (*w.body)();
    Fiber& f = w.fibers[w.cur];
    f.done = true;
    w.progress = true;
    // a finished thread no longer takes part in barriers (as exited waves on the hardware)
    w.alive--;
    const int wv = w.cur >> 6;
    w.wave_alive[wv]--;
    if (w.alive > 0 && w.block_arrived == w.alive) { w.block_arrived = 0; w.block_gen++; }
    if (w.wave_alive[wv] > 0 && w.wave_arrived[wv] == w.wave_alive[wv]) { w.wave_arrived[wv] = 0; w.wave_gen[wv]++; }
    switch_to_sched(w, f, true);
    std::abort();   // never resumed
}

void run_block(Worker& w) {
    const int nt = w.nt;
    w.alive = nt;
    w.block_arrived = 0;
    for (size_t i = 0; i < w.wave_arrived.size(); ++i) { w.wave_arrived[i] = 0; w.wave_alive[i] = 64; }
    for (int t = 0; t < nt; ++t) {
        Fiber& f = w.fibers[t];
        f.done = false;
#ifdef EMUL_FAST_SWITCH
        // initial frame for emul_swap: six callee-saved registers, then fiber_main as the return address; fiber_main must find
        // rsp = 8 mod 16, as after a call
        uintptr_t top = (((uintptr_t)f.stack + kStack) & ~(uintptr_t)15) - 8;
        void** sp = reinterpret_cast<void**>(top) - 7;
        for (int i = 0; i < 6; ++i) sp[i] = nullptr;
        sp[6] = reinterpret_cast<void*>(&fiber_main);
        f.sp = sp;
#else
        getcontext(&f.ctx);
        f.ctx.uc_stack.ss_sp = f.stack;
        f.ctx.uc_stack.ss_size = kStack;
        f.ctx.uc_link = nullptr;
        makecontext(&f.ctx, (void (*)())fiber_main, 0);
#endif
    }
    int remaining = nt;
    while (remaining > 0) {
        w.progress = false;
        for (int t = 0; t < nt; ++t) {
            Fiber& f = w.fibers[t];
            if (f.done) continue;
            w.cur = t;
            threadIdx = f.tidx;
#ifdef EMUL_ASAN
            __sanitizer_start_switch_fiber(&w.sched_fake, f.stack, kStack);
#endif
#ifdef EMUL_FAST_SWITCH
            emul_swap(&w.sched_sp, f.sp);
#else
            swapcontext(&w.sched, &f.ctx);
#endif
#ifdef EMUL_ASAN
            __sanitizer_finish_switch_fiber(w.sched_fake, nullptr, nullptr);
#endif
            if (f.done) remaining--;
        }
        if (!w.progress && remaining > 0) {
            std::fprintf(stderr,
                         "emul: barrier deadlock in block (%u,%u,%u): %d threads alive, %d at __syncthreads -- "
                         "divergent barrier or MFMA/shuffle executed by a partial wave\n",
                         blockIdx.x, blockIdx.y, blockIdx.z, w.alive, w.block_arrived);
            std::abort();
        }
    }
}
}  // namespace

namespace {
// Persistent worker threads: one set per process (re-created after a fork), each keeps its fiber stacks across launches.
struct Pool {
    std::mutex mu;
    std::condition_variable cv_job, cv_done;
    std::vector<std::thread> threads;
    pid_t pid = 0;
    unsigned long gen = 0;
    int active = 0;
    // the current job
    dim3 grid, block;
    int nt = 0;
    size_t nblocks = 0;
    const std::function<void()>* body = nullptr;
    std::atomic<size_t> next{0};
};
Pool* g_pool = nullptr;
std::mutex g_launch_mu;                                      // one launch at a time (streams are synchronous here)

void pool_thread(Pool* P) {
    Worker w;
    unsigned long seen = 0;
    for (;;) {
        {
            std::unique_lock<std::mutex> lk(P->mu);
            P->cv_job.wait(lk, [&] { return P->gen != seen; });
            seen = P->gen;
        }
        const int nt = P->nt;
        const dim3 grid = P->grid, block = P->block;
        w.nt = nt;
        w.body = P->body;
        w.fibers.resize(nt);
        if ((size_t)nt * kStack > w.stacks_cap) {
            std::free(w.stacks);
            w.stacks_cap = (size_t)nt * kStack;
            w.stacks = static_cast<char*>(std::malloc(w.stacks_cap));
            if (!w.stacks) { std::fprintf(stderr, "emul: out of memory for fiber stacks\n"); std::abort(); }
        }
        const int nw = nt / 64;
        w.wave_arrived.assign(nw, 0);
        w.wave_alive.assign(nw, 64);
        w.wave_gen.assign(nw, 0);
        w.block_gen = 0;
        w.slabs.assign((size_t)nw * 64 * 256, 0);
        for (int t = 0; t < nt; ++t) {
            w.fibers[t].stack = w.stacks + (size_t)t * kStack;
            w.fibers[t].tidx = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
        }
        tw = &w;
        blockDim = block;
        gridDim = grid;
        for (;;) {
            const size_t b = P->next.fetch_add(1);
            if (b >= P->nblocks) break;
            blockIdx = dim3((unsigned)(b % grid.x), (unsigned)((b / grid.x) % grid.y), (unsigned)(b / ((size_t)grid.x * grid.y)));
            run_block(w);
        }
        tw = nullptr;
        {
            std::lock_guard<std::mutex> lk(P->mu);
            if (--P->active == 0) P->cv_done.notify_all();
        }
    }
}

void pool_run(dim3 grid, dim3 block, int nt, size_t nblocks, const std::function<void()>& body) {
    std::lock_guard<std::mutex> launch_lock(g_launch_mu);
    if (!g_pool || g_pool->pid != getpid()) {                // first launch, or a forked child (the parent's threads do not exist here)
        g_pool = new Pool();
        g_pool->pid = getpid();
        unsigned hw = std::thread::hardware_concurrency();
        const unsigned n = hw ? hw : 4;
        for (unsigned i = 0; i < n; ++i) {
            g_pool->threads.emplace_back(pool_thread, g_pool);
            g_pool->threads.back().detach();
        }
    }
    Pool* P = g_pool;
    std::unique_lock<std::mutex> lk(P->mu);
    P->grid = grid;
    P->block = block;
    P->nt = nt;
    P->nblocks = nblocks;
    P->body = &body;
    P->next.store(0);
    P->active = (int)P->threads.size();
    P->gen++;
    P->cv_job.notify_all();
    P->cv_done.wait(lk, [&] { return P->active == 0; });
}
}  // namespace

namespace emul {
void sync_block() {
    Worker& w = *tw;
    const unsigned my = w.block_gen;
    if (++w.block_arrived == w.alive) {
        w.block_arrived = 0;
        w.block_gen++;
        w.progress = true;
        return;
    }
    while (w.block_gen == my) yield();
}
int lane_id() { return tw->cur & 63; }
char* wave_slab() { return tw->slabs.data() + (size_t)(tw->cur >> 6) * 64 * 256; }
void wave_sync() {
    Worker& w = *tw;
    const int wv = w.cur >> 6;
    const unsigned my = w.wave_gen[wv];
    if (++w.wave_arrived[wv] == w.wave_alive[wv]) {
        w.wave_arrived[wv] = 0;
        w.wave_gen[wv]++;
        w.progress = true;
        return;
    }
    while (w.wave_gen[wv] == my) yield();
}

void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body) {
    const int nt = (int)(block.x * block.y * block.z);
    if (nt <= 0 || nt > 1024 || nt % 64 != 0 || shmem > sizeof(alsep_smem)) {
        std::fprintf(stderr, "emul::launch: bad config threads=%d shmem=%zu\n", nt, shmem);
        g_last_error = hipErrorInvalidValue;
        return;
    }
    const size_t nblocks = (size_t)grid.x * grid.y * grid.z;
    if (nblocks == 0) return;
    pool_run(grid, block, nt, nblocks, body);
}
}  // namespace emul

hipError_t hipMalloc(void** p, size_t n) {
    void* q = nullptr;
    if (posix_memalign(&q, 256, n ? n : 256) != 0) return hipErrorOutOfMemory;
    *p = q;
    return hipSuccess;
}
hipError_t hipFree(void* p) { std::free(p); return hipSuccess; }
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { std::memcpy(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { std::memcpy(d, s, n); return hipSuccess; }
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { std::memset(d, v, n); return hipSuccess; }
hipError_t hipMemset(void* d, int v, size_t n) { std::memset(d, v, n); return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipDeviceSynchronize() { return hipSuccess; }
hipError_t hipGetLastError() { return g_last_error.exchange(hipSuccess); }
hipError_t hipPeekAtLastError() { return g_last_error.load(); }
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "emulated HIP error"; }

struct emul_event { std::chrono::steady_clock::time_point t; };
hipError_t hipEventCreate(hipEvent_t* e) { *e = new emul_event(); return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
    *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
    return hipSuccess;
}
hipError_t hipFuncSetAttribute(const void*, int, int) { return hipSuccess; }

float atomicAdd(float* p, float v) { std::lock_guard<std::mutex> g(g_atomic_mu); float o = *p; *p = o + v; return o; }
int atomicAdd(int* p, int v) { std::lock_guard<std::mutex> g(g_atomic_mu); int o = *p; *p = o + v; return o; }
unsigned atomicAdd(unsigned* p, unsigned v) { std::lock_guard<std::mutex> g(g_atomic_mu); unsigned o = *p; *p = o + v; return o; }
unsigned atomicMax(unsigned* p, unsigned v) { std::lock_guard<std::mutex> g(g_atomic_mu); unsigned o = *p; *p = std::max(o, v); return o; }
int atomicMax(int* p, int v) { std::lock_guard<std::mutex> g(g_atomic_mu); int o = *p; *p = std::max(o, v); return o; }

// hipcpu_runtime.cpp -- the execution model and the runtime API behind tests/hipcpu/hipcpu.h.  TEST TOOLING (see hipcpu.h).
#include "hipcpu.h"

#include <execinfo.h>
#include <sched.h>
#include <signal.h>
#include <time.h>
#include <unistd.h>

#include <string>
#include <thread>
#include <vector>

HIPCPU_TLS uint3 threadIdx, blockIdx;
HIPCPU_TLS dim3 blockDim, gridDim;

#if !defined(__x86_64__)
#error "hipcpu's fiber switch is written for x86-64"
#endif
extern "C" void hipcpu_switch(void** save_sp, void* load_sp);
asm(".text\n"
    ".hidden hipcpu_switch\n"
    ".globl hipcpu_switch\n"
    ".type hipcpu_switch,@function\n"
    "hipcpu_switch:\n"
    "    pushq %rbp\n    pushq %rbx\n    pushq %r12\n    pushq %r13\n    pushq %r14\n    pushq %r15\n"
    "    movq %rsp, (%rdi)\n"
    "    movq %rsi, %rsp\n"
    "    popq %r15\n    popq %r14\n    popq %r13\n    popq %r12\n    popq %rbx\n    popq %rbp\n"
    "    ret\n"
    ".size hipcpu_switch,.-hipcpu_switch\n");

namespace {

constexpr size_t kStack = 256 * 1024, kStackCoop = 64 * 1024;

struct Fiber {
    void* sp;
    uint3 tid;
    int lin;
    int parity;
    int xgen, xparity;           // generation / buffer half of this thread's latest wave exchange
    bool done;
};

struct Block {
    std::vector<Fiber> f;
    char* stacks = nullptr;
    std::vector<uint64_t> slots;                 // [2][n]
    std::vector<int> slot_gen;                   // [2][n]: wave-exchange generation in which the slot was written
    std::vector<int> wave_arrived, wave_gen, wave_live;
    int n = 0, live = 0, arrived = 0, gen = 0;
    void* sched = nullptr;
    Fiber* cur = nullptr;
    const std::function<void()>* body = nullptr;
    std::vector<char> dyn;
};

HIPCPU_TLS Block* g_blk = nullptr;
HIPCPU_TLS unsigned long long g_clock = 0;
HIPCPU_TLS bool g_coop = false;   // this OS thread runs one workgroup of a cooperative launch
HIPCPU_TLS unsigned g_yields = 0;

// HIPCPU_SCHEDULE=reverse: threads of a workgroup are resumed in descending order and workgroups run last to first.  Kernels without
// data races (and without order-dependent float atomics) must produce the same bits under both schedules: a cheap race detector.
static int g_reverse = -1;
inline bool reverse_schedule() {
    if (g_reverse < 0) { const char* e = getenv("HIPCPU_SCHEDULE"); g_reverse = (e && !strcmp(e, "reverse")) ? 1 : 0; }
    return g_reverse == 1;
}
inline Fiber* next_live(Block* b, Fiber* me) {
    int i = me->lin;
    if (reverse_schedule()) { do { i = (i == 0) ? b->n - 1 : i - 1; } while (b->f[i].done && i != me->lin); }
    else { do { i = (i + 1 == b->n) ? 0 : i + 1; } while (b->f[i].done && i != me->lin); }
    return &b->f[i];
}
inline void resume(Block* b, Fiber* from, Fiber* to) {
    b->cur = to;
    threadIdx = to->tid;
    hipcpu_switch(&from->sp, to->sp);
}
void fiber_entry() {
    Block* b = g_blk;
    Fiber* me = b->cur;
    (*b->body)();
    me->done = true;
    b->live--;
    b->wave_live[me->lin / 64]--;
    if (b->live > 0) resume(b, me, next_live(b, me));
    else hipcpu_switch(&me->sp, b->sched);
    abort();
}
void make_fiber(Fiber* f, char* stack, size_t size) {
    uintptr_t top = ((uintptr_t)stack + size) & ~(uintptr_t)15;
    void** sp = (void**)(top - 64);
    for (int i = 0; i < 8; i++) sp[i] = nullptr;
    sp[6] = (void*)fiber_entry;
    f->sp = sp;
}
void wave_barrier() {
    Block* b = g_blk;
    const int w = b->cur->lin / 64;
    const int my = b->wave_gen[w];
    b->wave_arrived[w]++;
    while (b->wave_gen[w] == my) {
        if (b->wave_arrived[w] >= b->wave_live[w]) { b->wave_arrived[w] = 0; b->wave_gen[w]++; break; }
        hipcpu::yield();
    }
}

}  // namespace

namespace hipcpu {

void yield() {
    Block* b = g_blk;
    Fiber* me = b->cur;
    Fiber* nx = next_live(b, me);
    g_clock += 16;
    if (g_coop && (++g_yields % (unsigned)b->n) == 0) sched_yield();   // a spinning workgroup lets the other workgroups' threads run
    if (nx != me) resume(b, me, nx);
}
void syncthreads() {
    Block* b = g_blk;
    const int my = b->gen;
    b->arrived++;
    while (b->gen == my) {
        if (b->arrived >= b->live) { b->arrived = 0; b->gen++; break; }
        yield();
    }
}
int lane() { return g_blk->cur->lin & 63; }
// did lane l of the calling thread's wavefront take part in the calling thread's latest exchange?  (Not "is it alive now": a lane may
// return from the kernel right after the exchange, before a slower lane has read its slot.)
bool lane_alive(int l) {
    Block* b = g_blk;
    Fiber* me = b->cur;
    const int i = (me->lin & ~63) + l;
    return l >= 0 && l < 64 && i < b->n && b->slot_gen[(size_t)me->xparity * b->n + i] == me->xgen;
}
const uint64_t* wave_publish(uint64_t v) {
    Block* b = g_blk;
    Fiber* me = b->cur;
    uint64_t* s = b->slots.data() + (size_t)me->parity * b->n;
    me->xparity = me->parity;
    me->xgen = b->wave_gen[me->lin / 64] + 1;      // exchanges of a wavefront are numbered from 1 (slot_gen starts at 0 = never)
    b->slot_gen[(size_t)me->parity * b->n + me->lin] = me->xgen;
    me->parity ^= 1;
    s[me->lin] = v;
    wave_barrier();   // one barrier is enough: this half is only written again two exchanges later (see oracle/ref_shim/mfref_runtime.cpp)
    return s + (me->lin & ~63);
}
void* dyn_shared() { return g_blk->dyn.data(); }
unsigned long long clock() { return g_clock += 4; }

// one workgroup on the calling OS thread
static void run_block(Block& blk, dim3 grid, dim3 block, unsigned bx, unsigned by, unsigned bz, size_t stack) {
    const int n = blk.n, nw = (n + 63) / 64;
    g_blk = &blk;
    gridDim = grid;
    blockDim = block;
    blk.live = n; blk.arrived = 0; blk.gen = 0;
    blk.wave_arrived.assign(nw, 0); blk.wave_gen.assign(nw, 0); blk.wave_live.assign(nw, 0);
    std::fill(blk.slot_gen.begin(), blk.slot_gen.end(), 0);
    for (int i = 0; i < n; ++i) {
        Fiber& f = blk.f[i];
        f.lin = i;
        f.tid = uint3{(unsigned)i % block.x, ((unsigned)i / block.x) % block.y, (unsigned)i / (block.x * block.y)};
        f.done = false;
        f.parity = 0; f.xgen = -1; f.xparity = 0;
        blk.wave_live[i / 64]++;
        make_fiber(&f, blk.stacks + (size_t)i * stack, stack - (size_t)((i * 37) % 64) * 64);
    }
    blockIdx = uint3{bx, by, bz};
    Fiber* first = &blk.f[reverse_schedule() ? n - 1 : 0];
    blk.cur = first;
    threadIdx = first->tid;
    hipcpu_switch(&blk.sched, first->sp);
    if (blk.live != 0) { fprintf(stderr, "hipcpu: workgroup returned with live threads\n"); abort(); }
}

static void init_block(Block& blk, int n, size_t dyn_shared_bytes, const std::function<void()>& body) {
    blk.n = n;
    blk.f.resize(n);
    blk.slots.assign(2 * (size_t)n + 64, 0);
    blk.slot_gen.assign(2 * (size_t)n + 64, 0);
    blk.dyn.assign(dyn_shared_bytes + 16, 0);
    blk.body = &body;
}

// HIPCPU_BACKTRACE=1: print the native stack when a kernel faults (the Python fault handler only knows Python frames)
static void segv_handler(int sig) {
    void* frames[64];
    const int n = backtrace(frames, 64);
    const char msg[] = "hipcpu: fault in emulated device code; native stack:\n";
    (void)!write(2, msg, sizeof(msg) - 1);
    backtrace_symbols_fd(frames, n, 2);
    _exit(139);
}
static void install_backtrace_once() {
    static bool done = false;
    if (done) return;
    done = true;
    if (getenv("HIPCPU_BACKTRACE")) { signal(SIGSEGV, segv_handler); signal(SIGBUS, segv_handler); }
}

static bool is_cooperative(const char* name) {
    const char* env = getenv("HIPCPU_COOPERATIVE");
    if (!env || !*env) return false;
    std::string list(env), kn(name);
    size_t at = 0;
    while (at <= list.size()) {
        size_t comma = list.find(',', at);
        if (comma == std::string::npos) comma = list.size();
        const std::string item = list.substr(at, comma - at);
        if (!item.empty() && kn.compare(0, item.size(), item) == 0) return true;
        at = comma + 1;
    }
    return false;
}

// stream capture (hipGraph): while a capture is open, launches and asynchronous copies are recorded instead of executed; hipGraphLaunch
// replays them.  One capture at a time (the runtime is single-threaded outside cooperative launches).
struct Graph { std::vector<std::function<void()>> ops; };
static Graph* g_capture = nullptr;

void launch(const char* name, dim3 grid, dim3 block, size_t dyn_shared_bytes, const std::function<void()>& body) {
    if (g_capture) {
        const std::string kn(name);
        const std::function<void()> copy = body;
        Graph* keep = g_capture;
        (void)keep;
        g_capture->ops.push_back([=]() { launch(kn.c_str(), grid, block, dyn_shared_bytes, copy); });
        return;
    }
    const int n = (int)(block.x * block.y * block.z);
    const size_t nblocks = (size_t)grid.x * grid.y * grid.z;
    if (n <= 0 || nblocks == 0) return;
    install_backtrace_once();
    if (is_cooperative(name)) {
#ifndef HIPCPU_COOP
        fprintf(stderr, "hipcpu: HIPCPU_COOPERATIVE names %s but this is not the HIPCPU_COOP build\n", name);
        abort();
#endif
        // every workgroup resident at once: one OS thread per workgroup, each with its own fibers, LDS (thread_local) and stacks
        std::vector<std::thread> th;
        for (unsigned bz = 0; bz < grid.z; ++bz)
            for (unsigned by = 0; by < grid.y; ++by)
                for (unsigned bx = 0; bx < grid.x; ++bx)
                    th.emplace_back([=, &body]() {
                        Block blk;
                        init_block(blk, n, dyn_shared_bytes, body);
                        blk.stacks = (char*)malloc((size_t)n * kStackCoop);
                        if (!blk.stacks) { fprintf(stderr, "hipcpu: out of memory for fiber stacks\n"); abort(); }
                        g_coop = true;
                        run_block(blk, grid, block, bx, by, bz, kStackCoop);
                        free(blk.stacks);
                    });
        for (auto& t : th) t.join();
        return;
    }
    Block blk;
    init_block(blk, n, dyn_shared_bytes, body);
    static HIPCPU_TLS char* pool = nullptr;
    static HIPCPU_TLS size_t pool_size = 0;
    if (pool_size < (size_t)n * kStack) {
        free(pool);
        pool_size = (size_t)n * kStack;
        pool = (char*)malloc(pool_size);
        if (!pool) { fprintf(stderr, "hipcpu: out of memory for fiber stacks\n"); abort(); }
    }
    blk.stacks = pool;
    Block* prev = g_blk;
    const dim3 prevGrid = gridDim, prevBlock = blockDim;
    const bool rev = reverse_schedule();
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx)
                run_block(blk, grid, block, rev ? grid.x - 1 - bx : bx, rev ? grid.y - 1 - by : by, rev ? grid.z - 1 - bz : bz, kStack);
    g_blk = prev;
    gridDim = prevGrid; blockDim = prevBlock;
}

}  // namespace hipcpu

// real time in microseconds, divided by HIPCPU_CLOCK_DIV (default 1): device code written against a 100 MHz counter sees time pass 100x
// (or 100 x DIV) more slowly, so a time-out meant for the GPU leaves room for 240 OS threads on a few cores (and for a sanitizer)
long long hipcpu_wall_clock() {
    static long long div = 0;
    if (div == 0) { const char* e = getenv("HIPCPU_CLOCK_DIV"); div = (e && atoll(e) > 0) ? atoll(e) : 1; }
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ((long long)ts.tv_sec * 1000000ll + ts.tv_nsec / 1000) / div;
}

// ---- runtime API: one address space, everything completes before it returns ------------------------------------------------------
static int g_dummy_handles = 0;
hipError_t hipMalloc(void** p, size_t n) { *p = malloc(n ? n : 1); if (*p) memset(*p, 0xCD, n); return *p ? hipSuccess : hipErrorOutOfMemory; }   // device memory is not zero
hipError_t hipFree(void* p) { free(p); return hipSuccess; }
hipError_t hipHostMalloc(void** p, size_t n, unsigned) { *p = calloc(n ? n : 1, 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) {
    if (hipcpu::g_capture) { hipcpu::g_capture->ops.push_back([=]() { memmove(d, s, n); }); return hipSuccess; }
    memmove(d, s, n);
    return hipSuccess;
}
hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) {
    if (hipcpu::g_capture) { hipcpu::g_capture->ops.push_back([=]() { memset(d, v, n); }); return hipSuccess; }
    memset(d, v, n);
    return hipSuccess;
}
hipError_t hipStreamBeginCapture(hipStream_t, hipStreamCaptureMode) {
    if (hipcpu::g_capture) return hipErrorIllegalState;
    hipcpu::g_capture = new hipcpu::Graph;
    return hipSuccess;
}
hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t* g) {
    if (!hipcpu::g_capture) return hipErrorIllegalState;
    *g = reinterpret_cast<hipGraph_t>(hipcpu::g_capture);
    hipcpu::g_capture = nullptr;
    return hipSuccess;
}
hipError_t hipGraphInstantiate(hipGraphExec_t* e, hipGraph_t g, hipGraphNode_t*, char*, size_t) {
    *e = reinterpret_cast<hipGraphExec_t>(new hipcpu::Graph(*reinterpret_cast<hipcpu::Graph*>(g)));
    return hipSuccess;
}
hipError_t hipGraphDestroy(hipGraph_t g) { delete reinterpret_cast<hipcpu::Graph*>(g); return hipSuccess; }
hipError_t hipGraphExecDestroy(hipGraphExec_t e) { delete reinterpret_cast<hipcpu::Graph*>(e); return hipSuccess; }
hipError_t hipGraphLaunch(hipGraphExec_t e, hipStream_t) {
    for (auto& op : reinterpret_cast<hipcpu::Graph*>(e)->ops) op();
    return hipSuccess;
}
hipError_t hipStreamCreate(hipStream_t* s) { *s = (hipStream_t)(uintptr_t)(0x1000 + ++g_dummy_handles); return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { return hipStreamCreate(s); }
hipError_t hipExtStreamCreateWithCUMask(hipStream_t* s, uint32_t, const uint32_t*) { return hipStreamCreate(s); }
hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipDeviceSynchronize() { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) { *e = (hipEvent_t)(uintptr_t)(0x2000 + ++g_dummy_handles); return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
hipError_t hipEventRecordWithFlags(hipEvent_t, hipStream_t, unsigned) { return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }
hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) { *v = 1; return hipSuccess; }   // one "compute unit": workgroups run one after the other
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipGetLastError() { return hipSuccess; }
const char* hipGetErrorString(hipError_t) { return "hipcpu"; }

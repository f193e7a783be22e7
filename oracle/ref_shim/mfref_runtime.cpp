// mfref_runtime.cpp -- the grid/block/thread execution model behind MFREF_LAUNCH (see mfref_cuda.h).
// TEST INFRASTRUCTURE ONLY.
//
// A block's threads are fibers resumed round-robin by the launching OS thread; threadIdx is rewritten before every
// resume.  On x86-64 a fiber switch is a dozen instructions (callee-saved registers + stack pointer, mfref_switch below); elsewhere
// it is ucontext's swapcontext, which costs a signal-mask system call per switch -- the tracking-loop pin (oracle/build_track.py)
// makes ~1e8 switches per call and would take minutes with it.  __syncthreads() = block barrier over the fibers that have not returned; __shfl_down() = exchange through
// a per-block slot array bracketed by two warp barriers (warps are 32 consecutive linear thread ids, as in CUDA).
#include "mfref_cuda.h"

#include <vector>

#if defined(__x86_64__)
// void mfref_switch(void** save_sp, void* load_sp): System V callee-saved integer registers; the MXCSR / x87 control words are
// never changed by this library and are left alone.
extern "C" void mfref_switch(void** save_sp, void* load_sp);
asm(".text\n"
    ".hidden mfref_switch\n"
    ".globl mfref_switch\n"
    ".type mfref_switch,@function\n"
    "mfref_switch:\n"
    "    pushq %rbp\n    pushq %rbx\n    pushq %r12\n    pushq %r13\n    pushq %r14\n    pushq %r15\n"
    "    movq %rsp, (%rdi)\n"
    "    movq %rsi, %rsp\n"
    "    popq %r15\n    popq %r14\n    popq %r13\n    popq %r12\n    popq %rbx\n    popq %rbp\n"
    "    ret\n"
    ".size mfref_switch,.-mfref_switch\n");
struct mfref_ctx { void* sp; };
static inline void ctx_switch(mfref_ctx* from, mfref_ctx* to) { mfref_switch(&from->sp, to->sp); }
static inline void ctx_make(mfref_ctx* c, char* stack, size_t size, void (*entry)()) {
    uintptr_t top = ((uintptr_t)stack + size) & ~(uintptr_t)15;
    void** sp = (void**)(top - 64);          // r15 r14 r13 r12 rbx rbp | return address = entry | a null "caller" (rsp = 8 mod 16 at entry)
    for (int i = 0; i < 8; i++) sp[i] = nullptr;
    sp[6] = (void*)entry;
    c->sp = sp;
}
#else
#include <ucontext.h>
struct mfref_ctx { ucontext_t uc; };
static inline void ctx_switch(mfref_ctx* from, mfref_ctx* to) { swapcontext(&from->uc, &to->uc); }
static inline void ctx_make(mfref_ctx* c, char* stack, size_t size, void (*entry)()) {
    getcontext(&c->uc);
    c->uc.uc_stack.ss_sp = stack;
    c->uc.uc_stack.ss_size = size;
    c->uc.uc_link = nullptr;
    makecontext(&c->uc, entry, 0);
}
#endif

uint3 threadIdx, blockIdx;
dim3 blockDim, gridDim;

namespace {

constexpr size_t kStack = 256 * 1024;

struct Fiber {
    mfref_ctx ctx;
    uint3 tid;
    int lin;
    int parity;               // which half of Block::slots this fiber's next __shfl_down writes
    bool done;
};

struct Block {
    std::vector<Fiber> f;
    char* stacks = nullptr;   // malloc'ed, never initialised: only the pages a fiber touches become resident
    std::vector<unsigned> slots;   // [2][n]: double-buffered, so that one warp barrier per shuffle is enough
    std::vector<int> warp_arrived, warp_gen, warp_live;
    int live = 0, arrived = 0, gen = 0;
    mfref_ctx sched;
    Fiber* cur = nullptr;
    const std::function<void()>* body = nullptr;
};

Block* g_blk = nullptr;

// round-robin: the next fiber after `me` that has not returned (me itself when it is the only one left)
inline Fiber* next_live(Block* b, Fiber* me) {
    const int n = (int)b->f.size();
    int i = me->lin;
    do { i = (i + 1 == n) ? 0 : i + 1; } while (b->f[i].done && i != me->lin);
    return &b->f[i];
}

inline void resume(Block* b, Fiber* from, Fiber* to) {
    b->cur = to;
    threadIdx = to->tid;
    ctx_switch(&from->ctx, &to->ctx);
}

void fiber_entry() {
    Block* b = g_blk;
    Fiber* me = b->cur;
    (*b->body)();
    me->done = true;
    b->live--;
    b->warp_live[me->lin / 32]--;
    if (b->live > 0) resume(b, me, next_live(b, me));
    else ctx_switch(&me->ctx, &b->sched);
    abort();   // a finished fiber is never resumed
}

// fibers hand over to one another directly; the launching thread is only returned to when the whole block has finished
inline void yield_fiber() {
    Block* b = g_blk;
    Fiber* me = b->cur;
    Fiber* nx = next_live(b, me);
    if (nx != me) resume(b, me, nx);
}

void warp_barrier() {
    Block* b = g_blk;
    const int w = b->cur->lin / 32;
    const int my = b->warp_gen[w];
    b->warp_arrived[w]++;
    while (b->warp_gen[w] == my) {
        if (b->warp_arrived[w] >= b->warp_live[w]) { b->warp_arrived[w] = 0; b->warp_gen[w]++; break; }
        yield_fiber();
    }
}

}  // namespace

void mfref_syncthreads() {
    Block* b = g_blk;
    const int my = b->gen;
    b->arrived++;
    while (b->gen == my) {
        if (b->arrived >= b->live) { b->arrived = 0; b->gen++; break; }
        yield_fiber();
    }
}

unsigned mfref_shfl_down_bits(unsigned v, int offset, int width) {
    Block* b = g_blk;
    Fiber* me = b->cur;
    const int n = (int)b->f.size(), lin = me->lin, lane = lin % 32;
    unsigned* slots = b->slots.data() + (size_t)me->parity * n;
    me->parity ^= 1;
    slots[lin] = v;
    // one barrier: a lane can only overwrite this half again two shuffles later, i.e. after every lane of the warp has passed the
    // barrier of the shuffle in between and therefore finished reading here
    warp_barrier();
    const int src = lin + offset;
    unsigned r = v;   // out-of-range source lane: the caller's own value (CUDA semantics)
    if ((lane % width) + offset < width && src < n && src / 32 == lin / 32) r = slots[src];
    return r;
}

void mfref_launch(dim3 grid, dim3 block, const std::function<void()>& body) {
    const int n = (int)(block.x * block.y * block.z);
    Block blk;
    blk.f.resize(n);
    static char* pool = nullptr;
    static size_t pool_size = 0;
    if (pool_size < (size_t)n * kStack) {
        free(pool);
        pool_size = (size_t)n * kStack;
        pool = (char*)malloc(pool_size);
        if (!pool) { fprintf(stderr, "mfref: out of memory for fiber stacks\n"); abort(); }
    }
    blk.stacks = pool;
    blk.slots.resize(2 * (size_t)n);
    const int nw = (n + 31) / 32;
    blk.body = &body;
    Block* prev = g_blk;
    g_blk = &blk;
    gridDim = grid;
    blockDim = block;
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx) {
                blk.live = n; blk.arrived = 0; blk.gen = 0;
                blk.warp_arrived.assign(nw, 0); blk.warp_gen.assign(nw, 0); blk.warp_live.assign(nw, 0);
                for (int i = 0; i < n; ++i) {
                    Fiber& f = blk.f[i];
                    f.lin = i;
                    f.tid = uint3{(unsigned)i % block.x, ((unsigned)i / block.x) % block.y, (unsigned)i / (block.x * block.y)};
                    f.done = false;
                    f.parity = 0;
                    blk.warp_live[i / 32]++;
                    // stack tops staggered by a few cache lines: stacks exactly kStack apart would all fall into the same cache sets
                    ctx_make(&f.ctx, blk.stacks + (size_t)i * kStack, kStack - (size_t)((i * 37) % 256) * 64, fiber_entry);
                }
                blockIdx = uint3{bx, by, bz};
                blk.cur = &blk.f[0];
                threadIdx = blk.f[0].tid;
                ctx_switch(&blk.sched, &blk.f[0].ctx);
                if (blk.live != 0) { fprintf(stderr, "mfref: block returned with live fibers\n"); abort(); }
            }
    g_blk = prev;
}

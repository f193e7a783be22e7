// mfref_runtime.cpp -- the grid/block/thread execution model behind MFREF_LAUNCH (see mfref_cuda.h).
// TEST INFRASTRUCTURE ONLY.
//
// A block's threads are ucontext fibers resumed round-robin by the launching OS thread; threadIdx is rewritten before
// every resume.  __syncthreads() = block barrier over the fibers that have not returned; __shfl_down() = exchange through
// a per-block slot array bracketed by two warp barriers (warps are 32 consecutive linear thread ids, as in CUDA).
#include "mfref_cuda.h"

#include <ucontext.h>
#include <vector>

uint3 threadIdx, blockIdx;
dim3 blockDim, gridDim;

namespace {

constexpr size_t kStack = 256 * 1024;

struct Fiber {
    ucontext_t ctx;
    uint3 tid;
    int lin;
    bool done;
};

struct Block {
    std::vector<Fiber> f;
    char* stacks = nullptr;   // malloc'ed, never initialised: only the pages a fiber touches become resident
    std::vector<unsigned> slots;
    std::vector<int> warp_arrived, warp_gen, warp_live;
    int live = 0, arrived = 0, gen = 0;
    ucontext_t sched;
    Fiber* cur = nullptr;
    const std::function<void()>* body = nullptr;
};

Block* g_blk = nullptr;

void fiber_entry() {
    Block* b = g_blk;
    Fiber* me = b->cur;
    (*b->body)();
    me->done = true;
    b->live--;
    b->warp_live[me->lin / 32]--;
    swapcontext(&me->ctx, &b->sched);
}

inline void yield_fiber() {
    Block* b = g_blk;
    Fiber* me = b->cur;
    swapcontext(&me->ctx, &b->sched);
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
    const int lin = b->cur->lin, lane = lin % 32;
    b->slots[lin] = v;
    warp_barrier();
    const int src = lin + offset;
    unsigned r = v;   // out-of-range source lane: the caller's own value (CUDA semantics)
    if ((lane % width) + offset < width && src < (int)b->f.size() && src / 32 == lin / 32) r = b->slots[src];
    warp_barrier();
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
    blk.slots.resize(n);
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
                    blk.warp_live[i / 32]++;
                    getcontext(&f.ctx);
                    f.ctx.uc_stack.ss_sp = blk.stacks + (size_t)i * kStack;
                    f.ctx.uc_stack.ss_size = kStack;
                    f.ctx.uc_link = &blk.sched;
                    makecontext(&f.ctx, fiber_entry, 0);
                }
                while (blk.live > 0) {
                    for (int i = 0; i < n; ++i) {
                        Fiber& f = blk.f[i];
                        if (f.done) continue;
                        blk.cur = &f;
                        blockIdx = uint3{bx, by, bz};
                        threadIdx = f.tid;
                        swapcontext(&blk.sched, &f.ctx);
                    }
                }
            }
    g_blk = prev;
}

// Microbenchmark (round 6, review item 9): what does a barrier cost that never leaves ONE XCD?  The SO(3) pre-alignment is a chain of <= 10 launches
// over 19 200 pixels, each on the dependent-launch floor (5.5 us); its iterations could run inside one launch on the 32 compute units of one XCD if a
// round of {write partial, barrier, read all partials} were much cheaper than a launch.  grid_barrier.hip measured 6.56 us per round for workgroups
// with blockIdx % 8 == 0 under agent-scope release / acquire (an L2 write-back + invalidate per round).  Here the workgroups read HW_REG_XCC_ID and only
// those on XCC 0 take part; their atomics and exchanges are workgroup-SCOPE operations on global memory -- performed in the XCD's own L2, which is the
// coherence point of every compute unit that takes part -- and every round exchanges through slots of its own (no line is read twice, so no stale L1
// line can be met; stores are complete -- s_waitcnt -- before the arrival is counted).
//   mode 0: all 256 workgroups, agent-scope release / acquire                          (the device-wide barrier, for reference)
//   mode 1: XCC 0's workgroups, agent-scope release / acquire
//   mode 2: XCC 0's workgroups, workgroup-scope relaxed atomics, polling with a returning atomic (l2_read: executes in L2)
// Build: hipcc --offload-arch=gfx950 -O3 xcd_barrier.hip -o xcd_barrier        Run: timeout 60 ./xcd_barrier
#include <hip/hip_runtime.h>
#include <stdio.h>

constexpr int kBlocks = 256, kThreads = 256, kRounds = 200;

__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xfu; }   // HW_REG_XCC_ID, bits 3:0

// a read that is performed IN the L2: a returning atomic add of zero.  (The compiler folds __hip_atomic_fetch_add(p, 0) at workgroup scope into a
// load with sc0, which the compute unit's own L1 may serve -- a workgroup is coherent through it -- and the poll then never sees the other CUs' arrivals:
// the first version of this benchmark timed out.)
__device__ __forceinline__ unsigned l2_read(unsigned* p) {
    unsigned r; const unsigned zero = 0u;
    asm volatile("global_atomic_add %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(p), "v"(zero) : "memory");
    return r;
}

__global__ void k_probe(unsigned* xcc) { if (threadIdx.x == 0) xcc[blockIdx.x] = xcc_id(); }

__global__ __launch_bounds__(kThreads) void k_rounds(float* __restrict__ slots, unsigned* counter, unsigned* err, int mode, unsigned members, float* sink) {
    __shared__ float s[kThreads];
    __shared__ unsigned s_rank;
    if (mode != 0 && xcc_id() != 0u) return;
    if (threadIdx.x == 0) s_rank = __hip_atomic_fetch_add(&counter[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // my slot among the members
    __syncthreads();
    const unsigned rank = s_rank;
    if (rank >= members) return;   // (more workgroups on XCC 0 than the probe saw: they stay out)
    float v = (float)rank;
    for (int r = 0; r < kRounds; ++r) {
        float* out = slots + (size_t)r * kBlocks * 16;
        if (threadIdx.x < 16) out[rank * 16 + threadIdx.x] = v * 1e-9f + (float)threadIdx.x;
        const unsigned target = (unsigned)(r + 1) * members;
        if (mode == 2) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // the stores above have reached L2
            __syncthreads();
            if (threadIdx.x == 0) {
                __hip_atomic_fetch_add(&counter[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                const long long t0 = wall_clock64();
                while (l2_read(&counter[0]) < target)
                    if (wall_clock64() - t0 > 5000000ll) { __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            }
            __syncthreads();
        } else {
            __syncthreads();
            if (threadIdx.x == 0) {
                __hip_atomic_fetch_add(&counter[0], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                const long long t0 = wall_clock64();
                while (__hip_atomic_load(&counter[0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
                    if (wall_clock64() - t0 > 5000000ll) { __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            __syncthreads();
        }
        if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
        float acc = 0.f;
        for (unsigned f = threadIdx.x; f < members * 16; f += kThreads) acc += out[f];
        s[threadIdx.x] = acc;
        __syncthreads();
        for (int o = kThreads / 2; o > 0; o >>= 1) { if ((int)threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o]; __syncthreads(); }
        v = s[0];
        __syncthreads();
    }
    if (threadIdx.x == 0) sink[rank] = v;
}

int main() {
    hipStream_t st;
    (void)hipStreamCreate(&st);
    float *slots, *sink; unsigned *counter, *err, *xcc;
    (void)hipMalloc(&slots, (size_t)kRounds * kBlocks * 16 * 4); (void)hipMalloc(&sink, kBlocks * 4);
    (void)hipMalloc(&counter, 8); (void)hipMalloc(&err, 4); (void)hipMalloc(&xcc, kBlocks * 4);
    hipLaunchKernelGGL(k_probe, dim3(kBlocks), dim3(kThreads), 0, st, xcc);
    unsigned h[kBlocks];
    (void)hipMemcpy(h, xcc, sizeof(h), hipMemcpyDeviceToHost);
    unsigned on0 = 0, rr = 1;
    for (int i = 0; i < kBlocks; ++i) { on0 += h[i] == 0u; rr &= (h[i] == (unsigned)(i % 8)); }
    printf("%d workgroups: %u on XCC 0; blockIdx %% 8 == XCC for all of them: %s\n", kBlocks, on0, rr ? "yes" : "no");
    const char* names[3] = {"device-wide, agent release/acquire", "XCC 0, agent release/acquire", "XCC 0, L2-local atomics"};
    float expect = 0.f;
    for (int mode = 0; mode < 3; ++mode) {
        const unsigned members = mode == 0 ? kBlocks : on0;
        float best = 1e30f; unsigned herr = 0; float hs = 0.f;
        for (int rep = 0; rep < 5 && !herr; ++rep) {
            (void)hipMemsetAsync(counter, 0, 8, st); (void)hipMemsetAsync(err, 0, 4, st); (void)hipMemsetAsync(slots, 0, (size_t)kRounds * kBlocks * 16 * 4, st);
            hipEvent_t e0, e1;
            (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            (void)hipEventRecord(e0, st);
            hipLaunchKernelGGL(k_rounds, dim3(kBlocks), dim3(kThreads), 0, st, slots, counter, err, mode, members, sink);
            (void)hipEventRecord(e1, st);
            (void)hipStreamSynchronize(st);
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, e0, e1);
            (void)hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost);
            (void)hipMemcpy(&hs, sink, 4, hipMemcpyDeviceToHost);
            if (ms < best) best = ms;
        }
        if (mode == 1) expect = hs;
        if (herr) printf("  %-40s barrier timed out\n", names[mode]);
        else printf("  %-40s %2u workgroups  %.2f us per round   (result %.6g%s)\n", names[mode], members, 1e3f * best / kRounds, hs,
                    mode == 2 ? (hs == expect ? ", same as the fenced form" : ", DIFFERS from the fenced form") : "");
    }
    return 0;
}

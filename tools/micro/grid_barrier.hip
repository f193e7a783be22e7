// Microbenchmark (prepared in round 2, NOT YET RUN -- the round's GPU budget was spent): what does one device-wide barrier cost inside
// ONE persistent launch, compared with the 3.44 us of a dependent kernel launch that does the same partial-sum exchange
// (launch_floor.hip)?  DESIGN.md section 4 argues that fusing the 19 Gauss-Newton iterations into one persistent kernel cannot lift
// the floor because a device-wide barrier is the same cross-XCD round trip; this measures it instead of arguing.
//
// One launch of <<<240, 512>>> (one round of workgroups, all co-resident on 256 CUs) runs R rounds of
//   barrier   : thread 0 of every workgroup arrives on a device-scope counter and spins until all 240 have        -> barrier only
//   exchange  : every workgroup first writes its own 128 B partial, then the barrier, then reads all 240 x 128 B  -> the ICP exchange
//   exchange+gather : + a dependent image gather after the exchange (as the model-map gather after the solve)
//   xcd-local : the same exchange among the 30 workgroups with blockIdx % 8 == 0 only (workgroups are dealt round-robin to the 8 XCDs, so
//               these share one L2): is a barrier that never leaves an XCD cheaper?  (The coarse pyramid levels -- 19 200 and 76 800
//               pixels -- could run their 9 iterations on one XCD's 32 CUs.)
// Release / acquire at agent scope around the counter (the L2s of the 8 XCDs are not coherent with each other for plain accesses).
// Every spin is bounded: if a barrier does not complete within ~50 ms the kernel sets an error flag and returns -- it cannot hang.
// Build: hipcc --offload-arch=gfx950 -O3 grid_barrier.hip -o grid_barrier        Run: timeout 60 ./grid_barrier
#include <hip/hip_runtime.h>
#include <stdio.h>

constexpr int kBlocks = 240, kThreads = 512;

__device__ __forceinline__ bool grid_barrier(unsigned* counter, unsigned target, unsigned* err) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        const long long t0 = wall_clock64();
        while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (wall_clock64() - t0 > 5000000ll) { __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }   // 100 MHz counter: 50 ms
            __builtin_amdgcn_s_sleep(1);
        }
    }
    __syncthreads();
    return __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u;
}

// mode 0: barrier only; 1: + exchange; 2: + dependent gather; 3: exchange among the workgroups of XCD 0 only
__global__ __launch_bounds__(kThreads) void k_persistent(float* __restrict__ bufA, float* __restrict__ bufB, const float* __restrict__ img,
                                                         unsigned* counter, unsigned* err, int rounds, int mode, float* sink) {
    __shared__ float s[kThreads];
    float v = (float)blockIdx.x;
    if (mode == 3) {
        if (blockIdx.x % 8 != 0) return;
        const unsigned members = (kBlocks + 7) / 8;
        for (int r = 0; r < rounds; ++r) {
            float* out = (r & 1) ? bufB : bufA;
            if (threadIdx.x < 32) out[blockIdx.x * 32 + threadIdx.x] = v * 1e-9f + (float)threadIdx.x;
            if (!grid_barrier(counter, (unsigned)(r + 1) * members, err)) return;
            const float4* p4 = reinterpret_cast<const float4*>(out);
            float acc = 0.f;
            for (int f = threadIdx.x; f < (int)members * 8; f += kThreads) {
                const float4 q = p4[(f / 8) * 64 + (f % 8)];      // partial of workgroup 8 * (f / 8)
                acc += q.x + q.y + q.z + q.w;
            }
            s[threadIdx.x] = acc;
            __syncthreads();
            for (int o = kThreads / 2; o > 0; o >>= 1) { if (threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o]; __syncthreads(); }
            v = s[0];
            __syncthreads();
        }
        if (threadIdx.x == 0) sink[blockIdx.x] = v;
        return;
    }
    for (int r = 0; r < rounds; ++r) {
        float* out = (r & 1) ? bufB : bufA;
        if (mode >= 1 && threadIdx.x < 32) out[blockIdx.x * 32 + threadIdx.x] = v * 1e-9f + (float)threadIdx.x;
        if (!grid_barrier(counter, (unsigned)(r + 1) * kBlocks, err)) return;
        if (mode >= 1) {
            const float4* p4 = reinterpret_cast<const float4*>(out);
            float acc = 0.f;
            for (int f = threadIdx.x; f < kBlocks * 8; f += kThreads) {
                const float4 q = p4[f];   // ordered after thread 0's agent-scope acquire by the workgroup barrier inside grid_barrier()
                acc += q.x + q.y + q.z + q.w;
            }
            s[threadIdx.x] = acc;
            __syncthreads();
            for (int o = kThreads / 2; o > 0; o >>= 1) { if (threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o]; __syncthreads(); }
            v = s[0];
            __syncthreads();
            if (mode >= 2) v += img[((int)(v * 0.f) + blockIdx.x * 1280 + threadIdx.x) % 307200];
        }
    }
    if (threadIdx.x == 0) sink[blockIdx.x] = v;
}

int main() {
    hipStream_t st;
    hipStreamCreate(&st);
    float *a, *b, *img, *sink;
    unsigned *counter, *err;
    hipMalloc(&a, kBlocks * 32 * 4); hipMalloc(&b, kBlocks * 32 * 4); hipMalloc(&img, 307200 * 4); hipMalloc(&sink, kBlocks * 4);
    hipMalloc(&counter, 4); hipMalloc(&err, 4);
    hipMemset(a, 0, kBlocks * 32 * 4); hipMemset(b, 0, kBlocks * 32 * 4); hipMemset(img, 0, 307200 * 4);
    const int rounds = 200;
    const char* names[4] = {"barrier", "exchange", "exchange+gather", "xcd-local"};
    printf("us per round inside one persistent launch of <<<%d, %d>>> (%d rounds):\n", kBlocks, kThreads, rounds);
    for (int mode = 0; mode < 4; ++mode) {
        float best = 1e30f;
        unsigned herr = 0;
        for (int rep = 0; rep < 5 && !herr; ++rep) {
            hipMemsetAsync(counter, 0, 4, st); hipMemsetAsync(err, 0, 4, st);
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0, st);
            hipLaunchKernelGGL(k_persistent, dim3(kBlocks), dim3(kThreads), 0, st, a, b, img, counter, err, rounds, mode, sink);
            hipEventRecord(e1, st);
            hipStreamSynchronize(st);
            float ms = 0.f;
            hipEventElapsedTime(&ms, e0, e1);
            (void)hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost);
            if (ms < best) best = ms;
        }
        if (herr) printf("  %-16s barrier timed out (workgroups not co-resident?)\n", names[mode]);
        else printf("  %-16s %.2f   (launch itself included once: subtract ~%.2f)\n", names[mode], 1e3f * best / rounds, 5.0f / rounds);
    }
    return 0;
}

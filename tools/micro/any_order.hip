// Microbenchmark: does hipExtAnyOrderLaunch (a dispatch packet without the barrier bit) let a kernel start beside the kernel in front of it on the
// SAME stream on this runtime / GPU?  Two kernels of 64 workgroups that each spin ~50 us: back to back they take ~100 us, side by side ~50 us.
// Also: the same pair on two streams (the reference for "side by side"), and a chain A, B(any order), C(ordered) -- C must still wait for both.
// Build: hipcc --offload-arch=gfx950 -O3 any_order.hip -o any_order        Run: timeout 60 ./any_order
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>

__global__ void k_spin(long long ticks, int* out, int tag) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(4);
    if (threadIdx.x == 0) out[blockIdx.x] = tag;
}
__global__ void k_check(const int* a, const int* b, int n, int* bad) {   // C: reads what A and B wrote
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && (a[i] != 1 || b[i] != 2)) atomicAdd(bad, 1);
}

int main() {
    hipStream_t s0, s1;
    hipStreamCreate(&s0); hipStreamCreate(&s1);
    int *a, *b, *bad;
    hipMalloc(&a, 64 * 4); hipMalloc(&b, 64 * 4); hipMalloc(&bad, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const long long ticks = 5000;   // 100 MHz: 50 us
    for (int mode = 0; mode < 4; ++mode) {
        float best = 1e30f; int hbad = 0;
        for (int rep = 0; rep < 5; ++rep) {
            hipMemsetAsync(a, 0, 256, s0); hipMemsetAsync(b, 0, 256, s0); hipMemsetAsync(bad, 0, 4, s0);
            hipStreamSynchronize(s0);
            hipEventRecord(e0, s0);
            hipLaunchKernelGGL(k_spin, dim3(64), dim3(64), 0, s0, ticks, a, 1);
            if (mode == 0) hipLaunchKernelGGL(k_spin, dim3(64), dim3(64), 0, s0, ticks, b, 2);
            if (mode == 1 || mode == 3) hipExtLaunchKernelGGL(k_spin, dim3(64), dim3(64), 0, s0, nullptr, nullptr, hipExtAnyOrderLaunch, ticks, b, 2);
            if (mode == 2) { hipLaunchKernelGGL(k_spin, dim3(64), dim3(64), 0, s1, ticks, b, 2); hipEventRecord(e1, s1); hipStreamWaitEvent(s0, e1, 0); }
            if (mode == 3) hipLaunchKernelGGL(k_check, dim3(1), dim3(64), 0, s0, a, b, 64, bad);
            hipEventRecord(e1, s0);
            hipStreamSynchronize(s0); hipStreamSynchronize(s1);
            float ms = 0.f;
            hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
            int h = 0; (void)hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost); hbad += h;
        }
        const char* names[4] = {"A, B ordered (one stream)", "A, B any-order (one stream)", "A, B on two streams", "A, B any-order, C ordered reads both"};
        printf("%-40s %.1f us%s\n", names[mode], best * 1e3f, mode == 3 ? (hbad ? "   C SAW STALE DATA" : "   C saw both results") : "");
    }
    return 0;
}

// Microbenchmark for Model::clean's ticket counters (mf_surfel.hip: clean_body): how fast does the GPU hand out tickets when the workgroups of a
// launch draw them from L counters that lie `stride` bytes apart?  One returning device-scope atomicAdd per ticket, issued by one thread of a
// workgroup that waits for the value before it asks again (as a workgroup of the clean pass does).
// Build: hipcc --offload-arch=gfx950 -O3 ticket_lanes.hip -o ticket_lanes
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ __launch_bounds__(64) void k_tickets(int* ctr, int lanes, int stride_ints, int rounds, int* sink) {
    if (threadIdx.x != 0) return;
    int* mine = ctr + (size_t)(blockIdx.x % lanes) * stride_ints;
    int acc = 0;
    for (int r = 0; r < rounds; ++r) {
        const int t = atomicAdd(mine, 1);
        acc += t;
        if (t < 0) break;     // never: the next request depends on the value of this one
    }
    if (acc == 0x7fffffff) sink[0] = acc;
}

int main() {
    const size_t bytes = 64ull << 20;
    int *ctr, *sink;
    if (hipMalloc(&ctr, bytes) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) return 1;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grids[] = {1024, 256};
    const int lanes[] = {1, 8, 32, 256};
    const int strides[] = {128, 4096, 65536, 1 << 20};
    const int rounds = 64;
    for (int g : grids)
        for (int L : lanes)
            for (int sb : strides) {
                if (L == 1 && sb != 128) continue;
                if ((size_t)L * sb > bytes) continue;
                float best = 1e30f;
                for (int rep = 0; rep < 3; ++rep) {
                    hipMemset(ctr, 0, bytes);
                    hipDeviceSynchronize();
                    hipEventRecord(e0);
                    hipLaunchKernelGGL(k_tickets, dim3(g), dim3(64), 0, 0, ctr, L, sb / 4, rounds, sink);
                    hipEventRecord(e1);
                    hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1);
                    if (ms < best) best = ms;
                }
                printf("grid %5d  lanes %4d  stride %8d B : %8.1f us for %d tickets = %6.1f ns per ticket\n", g, L, sb, best * 1e3, g * rounds,
                       best * 1e6 / (g * rounds));
            }
    return 0;
}

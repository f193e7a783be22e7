// Microbenchmark: z-buffer style 64-bit atomicMin at device scope on one buffer vs. workgroup scope on per-XCD private
// buffers (merged afterwards).  Build: hipcc --offload-arch=gfx950 -O3 atomic_scope.hip -o atomic_scope
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 0xF; }

__device__ __forceinline__ unsigned long long make_key(int i, int j) {
    unsigned h = (unsigned)i * 2654435761u + (unsigned)j * 40503u;
    return ((unsigned long long)(h >> 4) << 32) | (unsigned)i;
}
// surfel i covers a 3x4 pixel footprint around a position that moves smoothly with i (column-major like the real maps)
__device__ __forceinline__ int pix(int i, int j, int W, int H) {
    const int col = (i / (H / 2)) * 2 % W, row = (i % (H / 2)) * 2;
    const int x = min(W - 1, col + (j & 3)), y = min(H - 1, row + (j >> 2));
    return y * W + x;
}
template <int MODE>
__global__ __launch_bounds__(256) void k_scatter(unsigned long long* keys, int n, int W, int H, int* xcc_hist) {
    const int P = W * H;
    unsigned long long* base = keys;
    if (MODE == 1) base = keys + (size_t)xcc_id() * P;
    if (MODE == 1 && threadIdx.x == 0 && xcc_hist) atomicAdd(&xcc_hist[xcc_id()], 1);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
#pragma unroll
        for (int j = 0; j < 12; ++j) {
            const int p = pix(i, j, W, H);
            const unsigned long long key = make_key(i, j);
            if (MODE == 0) atomicMin(&base[p], key);
            else __hip_atomic_fetch_min(&base[p], key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
}
__global__ void k_merge(const unsigned long long* keys8, unsigned long long* out, int P) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
    unsigned long long m = ~0ull;
    for (int x = 0; x < 8; ++x) m = min(m, keys8[(size_t)x * P + p]);
    out[p] = m;
}
int main() {
    const int W = 640, H = 480, P = W * H, n = 280000;
    unsigned long long *a, *b, *m; int* hist;
    hipMalloc(&a, (size_t)P * 8); hipMalloc(&b, (size_t)P * 8 * 8); hipMalloc(&m, (size_t)P * 8); hipMalloc(&hist, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms;
    for (int rep = 0; rep < 3; ++rep) {
        hipMemset(a, 0xFF, (size_t)P * 8); hipMemset(b, 0xFF, (size_t)P * 64); hipMemset(hist, 0, 64);
        hipDeviceSynchronize();
        hipEventRecord(e0); hipLaunchKernelGGL(k_scatter<0>, dim3(2048), dim3(256), 0, 0, a, n, W, H, nullptr); hipEventRecord(e1);
        hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); printf("device scope, 1 buffer : %.1f us\n", ms * 1e3);
        hipEventRecord(e0); hipLaunchKernelGGL(k_scatter<1>, dim3(2048), dim3(256), 0, 0, b, n, W, H, hist); hipEventRecord(e1);
        hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); printf("workgroup scope, 8 bufs: %.1f us\n", ms * 1e3);
        hipEventRecord(e0); hipLaunchKernelGGL(k_merge, dim3((P + 255) / 256), dim3(256), 0, 0, b, m, P); hipEventRecord(e1);
        hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); printf("merge 8 -> 1           : %.1f us\n", ms * 1e3);
    }
    std::vector<unsigned long long> ha(P), hm(P); int hh[16];
    hipMemcpy(ha.data(), a, (size_t)P * 8, hipMemcpyDeviceToHost); hipMemcpy(hm.data(), m, (size_t)P * 8, hipMemcpyDeviceToHost);
    hipMemcpy(hh, hist, 64, hipMemcpyDeviceToHost);
    int bad = 0; for (int p = 0; p < P; ++p) bad += ha[p] != hm[p];
    printf("mismatching texels: %d of %d; blocks per xcc:", bad, P);
    for (int x = 0; x < 8; ++x) printf(" %d", hh[x]);
    printf("\n");
    return 0;
}

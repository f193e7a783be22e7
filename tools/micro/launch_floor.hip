// Microbenchmark: what does ONE link of a chain of dependent kernel launches cost on this GPU, whatever the kernel computes?
// The Gauss-Newton loop of the tracker is such a chain (19 launches per frame, each needing the previous launch's result), so this
// is the floor under k_icp_iter's duration.  Four chains of 200 launches on one stream, timed with HIP events:
//   empty     : <<<1, 64>>> kernel that does nothing                      -> launch-to-launch overhead
//   hop       : <<<1, 64>>> reads one word the previous launch wrote, writes one word      -> + one dependent memory round trip
//   exchange  : <<<240, 512>>> every workgroup reads the 240 x 128 B the previous launch wrote (as the ICP prologue does) and
//               writes its own 128 B                                                       -> the partial-sum exchange
//   exchange2 : the same + a second dependent gather (an image read whose address depends on the exchanged data)
// Build: hipcc --offload-arch=gfx950 -O3 launch_floor.hip -o launch_floor
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ void k_empty() {}
__global__ void k_hop(const float* in, float* out) { if (threadIdx.x == 0) out[0] = in[0] + 1.f; }
__global__ __launch_bounds__(512) void k_exchange(const float* __restrict__ in, float* __restrict__ out, const float* __restrict__ img, int gather) {
    __shared__ float s[512];
    const float4* p4 = reinterpret_cast<const float4*>(in);
    float acc = 0.f;
    for (int f = threadIdx.x; f < 240 * 8; f += 512) { const float4 v = p4[f]; acc += v.x + v.y + v.z + v.w; }
    s[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 256; o > 0; o >>= 1) { if (threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o]; __syncthreads(); }
    float v = s[0];
    if (gather) {   // address depends on the exchanged value: a second dependent round trip, like the model-map gather after the solve
        const int j = ((int)(v * 0.f) + blockIdx.x * 1280 + threadIdx.x) % 307200;
        v += img[j];
    }
    if (threadIdx.x < 32) out[blockIdx.x * 32 + threadIdx.x] = v * 1e-9f + (float)threadIdx.x;
}

template <class F>
static float time_chain(F launch, int n, hipStream_t s) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 20; ++i) launch(i);
    hipStreamSynchronize(s);
    hipEventRecord(e0, s);
    for (int i = 0; i < n; ++i) launch(i);
    hipEventRecord(e1, s);
    hipStreamSynchronize(s);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    return 1e3f * ms / n;
}

int main() {
    hipStream_t s;
    hipStreamCreate(&s);
    float *a, *b, *img;
    hipMalloc(&a, 240 * 32 * 4); hipMalloc(&b, 240 * 32 * 4); hipMalloc(&img, 307200 * 4);
    hipMemset(a, 0, 240 * 32 * 4); hipMemset(b, 0, 240 * 32 * 4); hipMemset(img, 0, 307200 * 4);
    const int n = 200;
    printf("us per launch in a dependent chain of %d launches:\n", n);
    printf("  empty      %.2f\n", time_chain([&](int) { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s); }, n, s));
    printf("  hop        %.2f\n", time_chain([&](int i) { hipLaunchKernelGGL(k_hop, dim3(1), dim3(64), 0, s, (i & 1) ? a : b, (i & 1) ? b : a); }, n, s));
    printf("  exchange   %.2f\n", time_chain([&](int i) { hipLaunchKernelGGL(k_exchange, dim3(240), dim3(512), 0, s, (i & 1) ? a : b, (i & 1) ? b : a, img, 0); }, n, s));
    printf("  exchange2  %.2f\n", time_chain([&](int i) { hipLaunchKernelGGL(k_exchange, dim3(240), dim3(512), 0, s, (i & 1) ? a : b, (i & 1) ? b : a, img, 1); }, n, s));
    return 0;
}

// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on this GPU for the access widths the library uses: kernels that move a KNOWN
// number of bytes (256 MiB per pass, far beyond the 32 MiB of L2) with 4 B and 16 B per lane.  Run under
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace ...   and   rocprofv3 --pmc WRITE_SIZE --kernel-trace ...
// (separate passes); tools/make_pmc_json.py divides the known byte counts by the reported values to get the correction factors
// it then applies to the library's kernels (MI355X_MICROARCH.md, "HBM": only the 16 B/lane read factor (x2) is documented).
// Build: hipcc --offload-arch=gfx950 -O3 fetch_calib.hip -o fetch_calib
#include <hip/hip_runtime.h>
#include <stdio.h>

constexpr size_t kFloats = 64ull << 20;   // 256 MiB

__global__ __launch_bounds__(256) void calib_read4(const float* __restrict__ p, size_t n, float* __restrict__ out) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += p[i];
    if (acc == 123.456f) out[0] = acc;   // never true: keeps the loads alive without a write
}
__global__ __launch_bounds__(256) void calib_read16(const float4* __restrict__ p, size_t n4, float* __restrict__ out) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) { const float4 v = p[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 123.456f) out[0] = acc;
}
__global__ __launch_bounds__(256) void calib_write4(float* __restrict__ p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = (float)i;
}
__global__ __launch_bounds__(256) void calib_write16(float4* __restrict__ p, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) p[i] = make_float4((float)i, 1.f, 2.f, 3.f);
}

int main() {
    float *a, *out;
    if (hipMalloc(&a, kFloats * 4) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) return 1;
    hipMemset(a, 0, kFloats * 4);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(calib_read4, dim3(4096), dim3(256), 0, 0, a, kFloats, out);
        hipLaunchKernelGGL(calib_read16, dim3(4096), dim3(256), 0, 0, reinterpret_cast<const float4*>(a), kFloats / 4, out);
        hipLaunchKernelGGL(calib_write4, dim3(4096), dim3(256), 0, 0, a, kFloats);
        hipLaunchKernelGGL(calib_write16, dim3(4096), dim3(256), 0, 0, reinterpret_cast<float4*>(a), kFloats / 4);
    }
    hipDeviceSynchronize();
    printf("calibration kernels moved %zu bytes per launch\n", kFloats * 4);
    return 0;
}

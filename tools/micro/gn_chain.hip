// Microbenchmark: where do the ~3.3 k cycles between the "reduce" and "sync" stamps of k_icp_iter go?  One wavefront runs the pieces of the
// Gauss-Newton prologue (unpack + LDL^T, exp, pose composition) on a realistic system and stamps s_memtime between them; a second kernel
// times chains of dependent fp64 / fp32 FMAs and of v_rcp_f64 to give the per-instruction latencies the estimate needs.
// Build (from tools/micro): hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -I../../maskfusion_amd/csrc -I../../include gn_chain.hip -o gn_chain
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "../../maskfusion_amd/csrc/mf_odometry.hip"

using namespace mf;

__global__ void k_chain(const double* sys29, unsigned long long* out, double* sink) {
    __shared__ double s_sys[32];
    __shared__ GNState s_st;
    __shared__ float s_pose[24];
    __shared__ float s_T[16];
    if (threadIdx.x < 32) s_sys[threadIdx.x] = threadIdx.x < 29 ? sys29[threadIdx.x] : 0.0;
    if (threadIdx.x == 0) {
        for (int k = 0; k < 16; ++k) s_st.resultRt[k] = (k % 5 == 0) ? 1.0 : 0.0;
        for (int k = 0; k < 9; ++k) { s_st.Rprev[k] = s_st.Rprev_inv[k] = s_st.Rcurr[k] = (k % 4 == 0) ? 1.f : 0.f; }
        for (int k = 0; k < 3; ++k) s_st.tprev[k] = s_st.tcurr[k] = 0.f;
        s_st.ill = 0;
    }
    __syncthreads();
    unsigned long long t[6];
    t[0] = __builtin_amdgcn_s_memtime();
    double A[6][6], b[6], x[6], ratio;
    int shift = 0;
    for (int i = 0; i < 6; ++i)
        for (int j = i; j < 7; ++j) { const double v = s_sys[shift++]; if (j == 6) b[i] = v; else { A[i][j] = v; A[j][i] = v; } }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    t[1] = __builtin_amdgcn_s_memtime();                       // unpack (27 LDS reads)
    ldlt6_solve(A, b, x, &ratio);
    sink[threadIdx.x] = x[0] + x[1] + x[2] + x[3] + x[4] + x[5] + ratio;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    t[2] = __builtin_amdgcn_s_memtime();                       // LDL^T + substitution
    double Rw[3][3];
    rodrigues_d(x[3], x[4], x[5], Rw);
    sink[64 + threadIdx.x] = Rw[0][0] + Rw[0][1] + Rw[0][2] + Rw[1][0] + Rw[1][1] + Rw[1][2] + Rw[2][0] + Rw[2][1] + Rw[2][2];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    t[3] = __builtin_amdgcn_s_memtime();                       // exp
    gn_finish_wg(s_sys, &s_st, s_pose, s_T);
    __syncthreads();
    t[4] = __builtin_amdgcn_s_memtime();                       // the whole production function (solve + exp + composition + barriers)
    if (threadIdx.x == 0) for (int k = 0; k < 5; ++k) out[k] = t[k];
}

template <int kKind>
__global__ void k_latency(double seed, unsigned long long* out, double* sink) {
    double a = seed, b = 1.0000001, c = 1e-9;
    float fa = (float)seed, fb = 1.0000001f, fc = 1e-9f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int k = 0; k < 256; ++k) {
        if (kKind == 0) a = fma(a, b, c);
        else if (kKind == 1) fa = fmaf(fa, fb, fc);
        else if (kKind == 2) a = __builtin_amdgcn_rcp(a) + c;
        else a = a * b;
    }
    sink[threadIdx.x] = a + fa;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[0] = t1 - t0;
}

int main() {
    double h_sys[29];
    // a well-conditioned ICP-like system: A = J^T J of a few random rows + diagonal, packed upper triangle of the 7-vector rows
    double rows[40][7];
    unsigned s = 12345;
    for (auto& r : rows) for (double& v : r) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xFFFF) / 65536.0 - 0.5; }
    int k = 0;
    for (int i = 0; i < 6; ++i)
        for (int j = i; j < 7; ++j) { double acc = 0; for (auto& r : rows) acc += r[i] * r[j]; h_sys[k++] = acc * (j == 6 ? 1e-3 : 1.0) + (i == j ? 1.0 : 0.0); }
    h_sys[27] = 1e-3; h_sys[28] = 40.0;
    double* d_sys; unsigned long long* d_out; double* d_sink;
    hipMalloc(&d_sys, sizeof(h_sys)); hipMalloc(&d_out, 64); hipMalloc(&d_sink, 4096);
    hipMemcpy(d_sys, h_sys, sizeof(h_sys), hipMemcpyHostToDevice);
    unsigned long long h[8];
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(k_chain, dim3(1), dim3(64), 0, 0, d_sys, d_out, d_sink);
        hipMemcpy(h, d_out, 40, hipMemcpyDeviceToHost);
    }
    printf("one wavefront, s_memtime ticks (100 MHz constant clock x ? -- compare with the 3.3 k of icp_prof's solve column):\n");
    printf("  unpack %llu   LDL^T+subst %llu   exp %llu   gn_finish_wg (all of it again, + composition, 3 barriers) %llu\n", h[1] - h[0], h[2] - h[1], h[3] - h[2], h[4] - h[3]);
    const char* names[4] = {"256 dependent v_fma_f64", "256 dependent v_fma_f32", "256 dependent v_rcp_f64 + add", "256 dependent v_mul_f64"};
    for (int kind = 0; kind < 4; ++kind) {
        for (int rep = 0; rep < 3; ++rep) {
            if (kind == 0) hipLaunchKernelGGL(k_latency<0>, dim3(1), dim3(64), 0, 0, 1.5, d_out, d_sink);
            if (kind == 1) hipLaunchKernelGGL(k_latency<1>, dim3(1), dim3(64), 0, 0, 1.5, d_out, d_sink);
            if (kind == 2) hipLaunchKernelGGL(k_latency<2>, dim3(1), dim3(64), 0, 0, 1.5, d_out, d_sink);
            if (kind == 3) hipLaunchKernelGGL(k_latency<3>, dim3(1), dim3(64), 0, 0, 1.5, d_out, d_sink);
            hipMemcpy(h, d_out, 8, hipMemcpyDeviceToHost);
        }
        printf("  %-32s %llu ticks = %.2f per instruction\n", names[kind], h[0], h[0] / 256.0);
    }
    return 0;
}

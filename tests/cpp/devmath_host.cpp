// devmath_host.cpp -- the product's own device-side scalar math (maskfusion_amd/csrc/mf_device.h and a few functions cut out of
// mf_odometry.hip / mf_surfel.hip by tests/devmath.py), compiled for the HOST with g++ so that `-m "not gpu"` tests can hold it to the
// oracle, numpy and SciPy without a GPU.  Nothing here is shipped: it is a second compilation of product source for testing.
//
// Stand-ins for what only exists on the device: bit casts, rsqrtf, the v_rcp_f64 / v_rsq_f64 seeds of rcp_d / sqrt_d (here the exact
// quotient, which the Newton steps after them leave unchanged), wavefront shuffles (never executed by the functions under test).
// Compile with -ffp-contract=off: the functions compared bit for bit (shader_exp / shader_acos, surfel radius / confidence / colour
// code, the clean-window walk) carry `#pragma clang fp contract(off)` or contain no contractible pattern on the device either.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <algorithm>
#include <hip/hip_runtime.h>

static inline float __int_as_float(int i) { float f; memcpy(&f, &i, 4); return f; }
static inline float __uint_as_float(unsigned i) { float f; memcpy(&f, &i, 4); return f; }
static inline int __float_as_int(float f) { int i; memcpy(&i, &f, 4); return i; }
static inline unsigned __float_as_uint(float f) { unsigned i; memcpy(&i, &f, 4); return i; }
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
template <class T> static inline T __shfl_xor(T v, int, int = 64) { return v; }
static inline int __builtin_amdgcn_mbcnt_lo(unsigned, int b) { return b; }
static inline int __builtin_amdgcn_mbcnt_hi(unsigned, int b) { return b; }
static inline double __builtin_amdgcn_rcp(double x) { return 1.0 / x; }
static inline double __builtin_amdgcn_rsq(double x) { return 1.0 / sqrt(x); }
static inline unsigned long long atomicMin(unsigned long long* a, unsigned long long v) { unsigned long long o = *a; if (v < o) *a = v; return o; }
#define __HIP_MEMORY_SCOPE_AGENT 3
#define __hip_atomic_load(p, order, scope) (*(p))
using std::min;
using std::max;

#include "mf_device.h"
#include "mf_labels.h"

namespace mf {
DEVMATH_SLICES
}  // namespace mf

extern "C" {

float dm_shader_exp(float x) { return mf::shader_exp(x); }
float dm_shader_acos(float x) { return mf::shader_acos(x); }
float dm_surfel_radius(float depth, float nz, float fx, float fy, float cx, float cy) { return mf::surfel_radius(depth, nz, mf::Intr{fx, fy, cx, cy}); }
float dm_surfel_confidence(float x, float y, float w, float fx, float fy, float cx, float cy) { return mf::surfel_confidence(x, y, w, mf::Intr{fx, fy, cx, cy}); }
float dm_encode_color(float r, float g, float b) { return mf::encode_color(r, g, b); }
void dm_decode_color(float c, float* rgb) { const float3 v = mf::decode_color(c); rgb[0] = v.x; rgb[1] = v.y; rgb[2] = v.z; }
int dm_mask_id(int value, int n) { return mf::mask_id(value, n); }
void dm_m33_inverse(const float* m, float* inv) { mf::m33_inverse_f(m, inv); }
void dm_rodrigues(const double* w, double* R9) {
    double R[3][3];
    mf::rodrigues_d(w[0], w[1], w[2], R);
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) R9[r * 3 + c] = R[r][c];
}
void dm_rodrigues2(const float* R9, double* r3) { mf::rodrigues2_d(R9, r3); }
void dm_quat_from_rot(const float* R9, float* q4) { mf::quat_from_rot(R9, q4); }
// copy_unstable.vert's window along one axis as the device walks it: texels u[3], multiplicities m[3]
void dm_window_slots_literal(float c, int size, int* u, int* m) {
    int uu[3], mm[3];
    mf::window_slots_literal(c, size, uu, mm);
    for (int k = 0; k < 3; ++k) { u[k] = uu[k]; m[k] = mm[k]; }
}
// one Gauss-Newton step exactly as thread 0 of k_icp_iter / k_icp_finalize performs it: unpack the 29 sums, LDL^T, exp, compose.
// in: sys29 (27 packed products, residual, inliers), resultRt16 row-major, Rprev9, tprev3.  out: x6, resultRt16, Rcurr9, tcurr3, trR9, trt3,
// stats2 = {lastICPError, lastICPCount}
void dm_gn_solve_update(const double* sys29, const double* resultRt16, const float* Rprev9, const float* tprev3, double* x6, double* rt_out16,
                        float* Rcurr9, float* tcurr3, float* trR9, float* trt3, float* stats2) {
    mf::GNState in, out;
    memset(&in, 0, sizeof(in)); memset(&out, 0, sizeof(out));
    for (int k = 0; k < 16; ++k) in.resultRt[k] = resultRt16[k];
    for (int k = 0; k < 9; ++k) { in.Rprev[k] = Rprev9[k]; in.Rcurr[k] = Rprev9[k]; }
    for (int k = 0; k < 3; ++k) { in.tprev[k] = tprev3[k]; in.tcurr[k] = tprev3[k]; }
    mf::m33_inverse_f(in.Rprev, in.Rprev_inv);
    in.levelDone = -1;
    // the solve on its own, for x
    double A[6][6], b[6], x[6];
    int shift = 0;
    for (int i = 0; i < 6; ++i)
        for (int j = i; j < 7; ++j) {
            const double v = sys29[shift++];
            if (j == 6) b[i] = v; else { A[i][j] = v; A[j][i] = v; }
        }
    mf::ldlt6_solve(A, b, x);
    for (int k = 0; k < 6; ++k) x6[k] = x[k];
    mf::gn_solve_update_serial(sys29, in, out);
    for (int k = 0; k < 16; ++k) rt_out16[k] = out.resultRt[k];
    for (int k = 0; k < 9; ++k) { Rcurr9[k] = out.Rcurr[k]; trR9[k] = out.trR[k]; }
    for (int k = 0; k < 3; ++k) { tcurr3[k] = out.tcurr[k]; trt3[k] = out.trt[k]; }
    stats2[0] = out.lastICPError; stats2[1] = out.lastICPCount;
}
// pose_derive: Model::pose -> its inverse, and Model::computeFusionWeight(1) from pose / lastPose (row-major R, t)
void dm_pose_derive(const float* R9, const float* t3, const float* lastR9, const float* lastT3, float* Ri9, float* ti3, float* fusionWeight, int literal) {
    mf::PoseDev p;
    memset(&p, 0, sizeof(p));
    p.weightLiteral = literal;      // "literalFusionWeight": Model::rodrigues2 with the reference's float trace
    for (int k = 0; k < 9; ++k) { p.R[k] = R9[k]; p.lastR[k] = lastR9[k]; }
    for (int k = 0; k < 3; ++k) { p.t[k] = t3[k]; p.lastT[k] = lastT3[k]; }
    mf::pose_derive(p);
    for (int k = 0; k < 9; ++k) Ri9[k] = p.Ri[k];
    for (int k = 0; k < 3; ++k) ti3[k] = p.ti[k];
    *fusionWeight = p.fusionWeight;
}

}  // extern "C"

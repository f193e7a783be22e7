// mf_odometry.hip -- frame-to-model projective ICP with the whole Gauss-Newton loop resident on the device.
//
// Replaces (reference, relative to /root/reference):
//   RGBDOdometry::initICPModel            Core/Utils/RGBDOdometry.cpp:153-185  (copyMaps, resizeVMap/NMap, tranformMaps;
//                                         Core/Cuda/cudafuncs.cu:207-331,366-445) + FillIn (Core/Shaders/fill_*.frag)
//   icpStep / ICPReduction / reduceSum    Core/Cuda/reduce.cu:92-187,259-525
//   the host side of getIncrementalTransformation (LDLT, SE(3) update)  RGBDOdometry.cpp:327-497,
//                                         Core/Utils/OdometryProvider.h:32-90
//
// Design (MI355X): the reference runs 19 x {kernel, reduce kernel, sync, D2H, host LDLT} per model per frame.
// Here one launch per iteration does everything: its prologue reduces the previous launch's per-workgroup
// partial sums in a fixed order, solves the 6x6 system in fp64 and updates the pose (every workgroup does this
// redundantly and identically, so no workgroup ever waits on another inside a launch -- visibility comes from
// the kernel boundary only), then streams the current vertex/normal planes, gathers the model planes, and reduces 29
// accumulators through LDS (lane pairs by DPP, column-wise stores, 16-lane butterflies).  No host round trip, no atomics,
// bit-reproducible run to run.  With the photometric term an iteration is two launches (k_rgbd_iter, k_rgb_step).
// A launch per iteration is the cheapest device-wide synchronisation this GPU offers: the same loop inside one persistent
// launch (device-wide barriers) was measured 5.8x slower in round 3 and removed.
#include <string.h>

#include "mf_internal.h"
#include "mf_device.h"
#include "mf_rgbd_device.h"
#include "mf_bilateral_device.h"

namespace mf {

// ------------------------------------------------------------------------------------------------
// small fp64 / fp32 linear algebra for the per-iteration solve (single thread)
// ------------------------------------------------------------------------------------------------
// LDL^T in double, fully unrolled so that the 6x6 system lives in registers (a dynamically indexed local array would be
// spilled to scratch memory and turn this ~200-flop solve into ~40 us of dependent scratch round trips).  Stands in for
// Eigen::LDLT (RGBDOdometry.cpp:451-459); Eigen pivots on the diagonal, which for these symmetric positive definite
// systems changes the result only by rounding.  A non-positive / vanishing pivot zeroes that component (Eigen's
// behaviour for singular systems) instead of dividing by it.
__device__ __forceinline__ void ldlt6_solve(double (&A)[6][6], const double (&b)[6], double (&x)[6], double* min_pivot_ratio = nullptr) {
    double maxdiag = 0.0, minpiv = 1.7976931348623157e308;
#pragma unroll
    for (int i = 0; i < 6; ++i) maxdiag = fmax(maxdiag, fabs(A[i][i]));
    const double tol = maxdiag * 1e-14;
    double dinv[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const double d = A[k][k];
        const bool ok = d > tol;
        minpiv = fmin(minpiv, ok ? d : 0.0);
        dinv[k] = ok ? rcp_d(d) : 0.0;   // v_rcp_f64 + Newton: an IEEE division is a ~40-instruction dependent chain per pivot
        // the lower triangle (i >= j) is the working copy; col = column k of the current Schur complement
        double col[6];
#pragma unroll
        for (int i = k + 1; i < 6; ++i) col[i] = A[i][k];
#pragma unroll
        for (int i = k + 1; i < 6; ++i) {
            const double l = col[i] * dinv[k];
#pragma unroll
            for (int j = k + 1; j <= i; ++j) A[i][j] -= l * col[j];
            A[i][k] = l;  // L below the diagonal
        }
    }
    double y[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        double s = b[i];
#pragma unroll
        for (int j = 0; j < i; ++j) s -= A[i][j] * y[j];
        y[i] = s;
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) y[i] *= dinv[i];
#pragma unroll
    for (int i = 5; i >= 0; --i) {
        double s = y[i];
#pragma unroll
        for (int j = i + 1; j < 6; ++j) s -= A[j][i] * x[j];
        x[i] = s;
    }
    // smallest pivot over largest diagonal entry: >= 1 / cond(A) for a positive definite A (0 when a pivot vanished or A = 0)
    if (min_pivot_ratio) *min_pivot_ratio = maxdiag > 0.0 ? minpiv / maxdiag : 0.0;
}

// Wave-parallel 6x6 solve: Gauss-Jordan on the augmented [A|b] with one lane per element (lanes 0..41 of one
// wavefront, fp64, no pivoting -- the systems are symmetric positive definite; a vanishing pivot zeroes that component
// like Eigen::LDLT::solve does for singular systems).  Six rank-1 steps of {3 shuffles, 1 fma} replace a ~3 us
// single-thread LDL^T; the serial ldlt6_solve above is kept as the executable specification (tests compare both).
// sys: 27 packed upper-triangle products of the 7-vector row (reduce.cu:378-411 order) in LDS.  Returns x in every lane.
__device__ __forceinline__ int solve6_lane_index(int& r, int& c) {
    const int l = threadIdx.x & 63;
    r = l < 42 ? l / 7 : 0; c = l < 42 ? l % 7 : 0;
    const int i = r < c ? r : c, j = r < c ? c : r;                 // A is symmetric; column 6 is b
    return (c == 6) ? (r * 7 - (r * (r - 1)) / 2 + (6 - r)) : (i * 7 - (i * (i - 1)) / 2 + (j - i));
}
__device__ __forceinline__ void solve6_wave_core(double v, int r, int c, double (&x)[6]) {
    double maxdiag = 0.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) maxdiag = fmax(maxdiag, fabs(__shfl(v, k * 8, 64)));
    const double tol = maxdiag * 1e-14;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const double pivot = __shfl(v, k * 8, 64);
        const double rowk = __shfl(v, k * 7 + c, 64);
        const double colk = __shfl(v, r * 7 + k, 64);
        const double inv = pivot > tol ? rcp_d(pivot) : 0.0;
        const double scaled = rowk * inv;
        v = (r == k) ? scaled : v - colk * scaled;
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) x[k] = __shfl(v, k * 7 + 6, 64);
}
__device__ __forceinline__ void solve6_wave(const double* sys, double (&x)[6]) {
    int r, c;
    const int idx = solve6_lane_index(r, c);
    const double v = (double)(float)sys[idx];                        // the reference hands float A/b to the host
    solve6_wave_core(v, r, c, x);
}
// lastA = A_rgbd + w^2 A_icp, lastb = b_rgbd + w b_icp (RGBDOdometry.cpp:447-452; sic -- the ICP step is scaled by 1/w)
__device__ __forceinline__ void solve6_wave_rgbd(const double* sysIcp, const double* sysRgb, double wA, double wb, double (&x)[6]) {
    int r, c;
    const int idx = solve6_lane_index(r, c);
    const double v = (double)(float)sysRgb[idx] + ((c == 6) ? wb : wA) * (double)(float)sysIcp[idx];
    solve6_wave_core(v, r, c, x);
}

// Pose update from the Gauss-Newton step x: RGBDOdometry.cpp:428-474 (icp && !rgb branch) +
// OdometryProvider::computeUpdateSE3.  Everything is unrolled with compile-time indices (registers only).
__device__ __forceinline__ void gn_update_from_x(const double (&x)[6], float res, float inl, const GNState& in, GNState& out) {
    out.lastICPError = sqrtf(res) / inl;
    out.lastICPCount = inl;
    // resultRt <- [exp(w) | t] * resultRt
    double Rw[3][3];
    rodrigues_d(x[3], x[4], x[5], Rw);
    double nr[3][4];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c)
            nr[r][c] = Rw[r][0] * in.resultRt[0 * 4 + c] + Rw[r][1] * in.resultRt[1 * 4 + c] + Rw[r][2] * in.resultRt[2 * 4 + c] +
                       x[r] * in.resultRt[3 * 4 + c];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) out.resultRt[r * 4 + c] = nr[r][c];
#pragma unroll
    for (int c = 0; c < 4; ++c) out.resultRt[12 + c] = in.resultRt[12 + c];
    float trR[9], trt[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int c = 0; c < 3; ++c) trR[r * 3 + c] = (float)nr[r][c];
        trt[r] = (float)nr[r][3];
    }
    // currentT = [Rprev|tprev] * transform.inverse()
    float iR[9];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) iR[r * 3 + c] = trR[c * 3 + r];
    const float3 itv = mul33(iR, f3(trt[0], trt[1], trt[2]));
    const float it3[3] = {-itv.x, -itv.y, -itv.z};
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c)
            out.Rcurr[r * 3 + c] = in.Rprev[r * 3 + 0] * iR[0 * 3 + c] + in.Rprev[r * 3 + 1] * iR[1 * 3 + c] +
                                   in.Rprev[r * 3 + 2] * iR[2 * 3 + c];
    const float3 tv = mul33(in.Rprev, f3(it3[0], it3[1], it3[2]));
    out.tcurr[0] = tv.x + in.tprev[0]; out.tcurr[1] = tv.y + in.tprev[1]; out.tcurr[2] = tv.z + in.tprev[2];
#pragma unroll
    for (int k = 0; k < 9; ++k) { out.Rprev[k] = in.Rprev[k]; out.Rprev_inv[k] = in.Rprev_inv[k]; out.trR[k] = trR[k]; }
#pragma unroll
    for (int k = 0; k < 3; ++k) { out.tprev[k] = in.tprev[k]; out.trt[k] = trt[k]; }
    out.valid = 1;
    out.levelDone = in.levelDone; out.lastRGBError = in.lastRGBError; out.lastRGBCount = in.lastRGBCount; out.ill = in.ill;
}

// Serial form (one thread): unpack -> LDL^T -> update.  mf_k_gn_solve (tests/test_gpu_kernels.py) runs it next to the wave solver
// and the oracle's restatement of Eigen's LDLT / OdometryProvider::rodrigues / computeUpdateSE3.
__device__ __forceinline__ void gn_solve_update_serial(const double* sys, const GNState& in, GNState& out) {
    double A[6][6], b[6], x[6];
    int shift = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = i; j < 7; ++j) {
            // (the reference rounds A, b to fp32 on their way to the host; the fp32 per-workgroup partials already differ from
            // an all-fp64 sum by ~1e-7 relative, so the extra rounding -- 54 quarter-rate conversions on this single
            // thread -- is dropped)
            const double value = sys[shift++];
            if (j == 6) b[i] = value;
            else { A[i][j] = value; A[j][i] = value; }
        }
    ldlt6_solve(A, b, x);
    gn_update_from_x(x, (float)sys[27], (float)sys[28], in, out);
}

// The same update with the state resident in LDS, run by WAVEFRONT 0 ALONE (the other wavefronts of the workgroup return at once and wait
// at the caller's next barrier, as they always did).  Lanes 0..11 each solve the system redundantly in registers (so x needs no broadcast:
// the LDL^T is ~120 instructions whichever way it is cut) and compute ONE entry of the 3x4 block of resultRt; the float work after that --
// transform^-1, Rcurr = Rprev iR, tcurr = Rprev it + tprev -- is exchanged between those lanes with wavefront shuffles.
// Round 3 exchanged through LDS behind three workgroup barriers: the in-kernel stamps of round 4 (profiles/r04a_icp_prof.txt) priced those
// three steps at 500 + 650 + 480 of the prologue's ~4 000 cycles, more than the LDL^T (1 000) or exp + composition (1 000).  LDS
// instructions of ONE wavefront execute in issue order, so reading the old resultRt and then overwriting it needs no barrier either.
// The float expressions are those of the CPU restatement, operation by operation (contraction off): given the same T the pose is the
// oracle's bit for bit, in every kernel that calls this.
// s_st is updated in place: resultRt, Rcurr, tcurr, trR, trt, lastICPError / Count, valid, ill; everything else keeps its value.  s_pose
// receives Rcurr[9] tcurr[3] (slots 0..11; slots 12..23 = Rprev_inv, tprev are pose-independent and written by the caller once).
// The caller places a barrier between this call and any other wavefront's use of s_st / s_pose.
struct GnStamps { unsigned long long t[6]; };   // shader-clock stamps inside the solve (profiling; thread 0 of workgroup 0)
__device__ __forceinline__ void gn_finish_wg(const double* s_sys, GNState* s_st, float* s_pose, float* s_T /*[16], unused since round 4*/, bool stamps,
                                             GnStamps& st) {
    (void)s_T;
    if (threadIdx.x == 64) {   // wavefront 1 has nothing to do until the caller's barrier: the iteration's statistics (an IEEE square root and a
                               // division, ~150 cycles) leave wavefront 0's chain
        const float res = (float)s_sys[27], inl = (float)s_sys[28];
        s_st->lastICPError = sqrtf(res) / inl;
        s_st->lastICPCount = inl;
        s_st->valid = 1;
    }
    if (threadIdx.x >= 64) return;
    const int l = threadIdx.x;
    const int r = (l >> 2) & 3, c = l & 3;   // lanes 0..11: entry (r, c) of the 3x4 block
    // pose-independent operands of the float tail, fetched before the solve (their latency hides under it): row `row` of Rprev, tprev[q]
    const int rr = l / 3, cc = l - 3 * rr, q = l - 9;
    const bool isR = l < 9, isT = l >= 9 && l < 12;
    const int row = isR ? rr : (isT ? q : 0);
    const float p0 = s_st->Rprev[row * 3 + 0], p1 = s_st->Rprev[row * 3 + 1], p2 = s_st->Rprev[row * 3 + 2];
    const float tpq = s_st->tprev[isT ? q : 0];
    double nr = 0.0;
    int ill = 0;
    if (l < 12) {
        double A[6][6], b[6], x[6], ratio;
        int shift = 0;
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = i; j < 7; ++j) {
                const double value = s_sys[shift++];
                if (j == 6) b[i] = value;
                else { A[i][j] = value; A[j][i] = value; }
            }
        if (stamps) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); st.t[0] = __builtin_amdgcn_s_memtime(); }
        ldlt6_solve(A, b, x, &ratio);
        ill = (s_sys[28] < 6.0 || !(ratio >= 1e-8)) ? 1 : 0;   // outside the stated domain of this solver (finding F4): counted, see GNState::ill
        if (stamps) st.t[1] = __builtin_amdgcn_s_memtime() + (unsigned long long)(x[0] == 1.2345e300);   // (the stamp waits for the solve)
        double Rw[3][3];
        rodrigues_d(x[3], x[4], x[5], Rw);
        const double w0 = r == 0 ? Rw[0][0] : (r == 1 ? Rw[1][0] : Rw[2][0]);
        const double w1 = r == 0 ? Rw[0][1] : (r == 1 ? Rw[1][1] : Rw[2][1]);
        const double w2 = r == 0 ? Rw[0][2] : (r == 1 ? Rw[1][2] : Rw[2][2]);
        const double xr = r == 0 ? x[0] : (r == 1 ? x[1] : x[2]);
        const double* Rt = s_st->resultRt;
        nr = w0 * Rt[0 * 4 + c] + w1 * Rt[1 * 4 + c] + w2 * Rt[2 * 4 + c] + xr * Rt[3 * 4 + c];   // resultRt <- [exp(w) | t] * resultRt
        if (stamps) st.t[2] = __builtin_amdgcn_s_memtime() + (unsigned long long)(nr == 1.2345e300);
    }
    // Isometry3f transform T = float(resultRt[0..2][0..3]) lives in lanes r * 4 + c; currentT = [Rprev|tprev] * transform.inverse():
    // iR = trR^T, it = -iR trt, Rcurr = Rprev iR, tcurr = Rprev it + tprev.  The twelve entries are broadcast with v_readlane (a scalar
    // register each, a few cycles) and every lane selects its operands: no LDS round trip, no ds_bpermute latency on the chain.
    const float tv = (float)nr;
    const int tvb = __float_as_int(tv);
    float T[12];
#pragma unroll
    for (int j = 0; j < 12; ++j) T[j] = __int_as_float(__builtin_amdgcn_readlane(tvb, j));
    // lanes 0..8: a_k = T[cc][k]; lanes 9..11: a_k = T[k][q]
    const float a0 = isR ? (cc == 0 ? T[0] : (cc == 1 ? T[4] : T[8])) : (q == 0 ? T[0] : (q == 1 ? T[1] : T[2]));
    const float a1 = isR ? (cc == 0 ? T[1] : (cc == 1 ? T[5] : T[9])) : (q == 0 ? T[4] : (q == 1 ? T[5] : T[6]));
    const float a2 = isR ? (cc == 0 ? T[2] : (cc == 1 ? T[6] : T[10])) : (q == 0 ? T[8] : (q == 1 ? T[9] : T[10]));
    const float t0 = T[3], t1 = T[7], t2 = T[11];                                                   // trt
    float rcur, itq;
    {
#pragma clang fp contract(off)
        rcur = p0 * a0 + p1 * a1 + p2 * a2;                 // lanes 0..8: Rcurr[rr][cc] = sum_k Rprev[rr][k] iR[k][cc], iR[k][cc] = T[cc][k]
        itq = -(a0 * t0 + a1 * t1 + a2 * t2);               // lanes 9..11: it[q] = -(iR trt)[q], iR[q][k] = T[k][q]
    }
    const int itb = __float_as_int(itq);
    const float i0 = __int_as_float(__builtin_amdgcn_readlane(itb, 9)), i1 = __int_as_float(__builtin_amdgcn_readlane(itb, 10)),
                i2 = __int_as_float(__builtin_amdgcn_readlane(itb, 11));
    float tcur;
    {
#pragma clang fp contract(off)
        tcur = (p0 * i0 + p1 * i1 + p2 * i2) + tpq;         // lanes 9..11: tcurr[q]
    }
    if (l < 12) {
        s_st->resultRt[r * 4 + c] = nr;   // behind the broadcast: every lane of the wavefront has read the old block by now (one wavefront,
                                          // instructions in order -- and a schedule-independent order for the CPU-executed build)
        if (c < 3) s_st->trR[r * 3 + c] = tv; else s_st->trt[r] = tv;
    }
    if (isR) { s_st->Rcurr[l] = rcur; s_pose[l] = rcur; }
    else if (isT) { s_st->tcurr[q] = tcur; s_pose[9 + q] = tcur; }
    if (l == 0) {
        s_st->ill += ill;
        if (stamps) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); st.t[3] = __builtin_amdgcn_s_memtime(); st.t[4] = st.t[3]; st.t[5] = st.t[3]; }
    }
}
__device__ __forceinline__ void gn_finish_wg(const double* s_sys, GNState* s_st, float* s_pose, float* s_T /*[16]*/) {
    GnStamps unused;
    gn_finish_wg(s_sys, s_st, s_pose, s_T, false, unused);
}

// value of lane (lane ^ kXor) for a 32-bit pattern, kXor = 1 or 2: quad permutes on the DPP path (no LDS crossbar)
template <int kXor>
__device__ __forceinline__ float lane_xor_bits(int bits, int /*lane*/) {
    static_assert(kXor == 1 || kXor == 2, "quad permutes only");
    return __int_as_float(__builtin_amdgcn_update_dpp(0, bits, kXor == 1 ? 0xB1 : 0x4E, 0xf, 0xf, false));
}

// Fixed-order reduction of `nb` per-workgroup partials ([nb][32] floats) by a 256-thread workgroup -> sys[32] doubles
// in LDS.  The array is read as float4s with up to 10 independent 16 B loads in flight per lane (a one-load-at-a-time
// loop cost ~70 cycles per partial: 10 us at 300 workgroups; the partials live in other XCDs' L2s, so every batch is
// a fabric round trip).  Thread t < 256 owns components 4*(t%8)..+3 of workgroups t/8, t/8 + 32, ...
// Round 4 (in-kernel stamps, profiles/r04c_icp_prof.txt: this phase was 3.1 k of a launch's 13.6 k ticks, of which the memory round trip
// is ~0.9 k): a thread's own <= 10 values per component are added in fp32 -- they ARE fp32 sums of up to 1 536 products each, the extra
// rounding is below theirs -- instead of 40 conversions + 40 fp64 additions + 80 selects per thread on a quarter-rate pipe; the 32 row sums
// per component are then added in fp64 by FOUR lanes per component (8 rows each, in row order) and a quad butterfly on the DPP path
// instead of one lane walking 32 dependent additions.  The order is fixed, so every workgroup (and every run) gets bit-identical sums.
__device__ __forceinline__ double quad_sum_d(double v) {   // v + its three quad neighbours, same bits in all four lanes
    const int lane = threadIdx.x & 63;
    int lo = __double2loint(v), hi = __double2hiint(v);
    double o = __hiloint2double(__float_as_int(lane_xor_bits<1>(hi, lane)), __float_as_int(lane_xor_bits<1>(lo, lane)));
    v = (lane & 1) ? o + v : v + o;            // both lanes of a pair add (even lane's value) + (odd lane's value)
    lo = __double2loint(v); hi = __double2hiint(v);
    o = __hiloint2double(__float_as_int(lane_xor_bits<2>(hi, lane)), __float_as_int(lane_xor_bits<2>(lo, lane)));
    return (lane & 2) ? o + v : v + o;
}
__device__ __forceinline__ void reduce_rows(const double* s_seg /*[32][32]*/, double* s_sys /*[32]*/, int t /* 0..127 */) {
    const int comp = t >> 2, part = t & 3;
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += s_seg[(part * 8 + k) * 32 + comp];
    s = quad_sum_d(s);
    if (part == 0) s_sys[comp] = s;
}
__device__ __forceinline__ void partial_sums_f32(const float4* __restrict__ p4, int n4, int t, float (&a)[4]) {
    a[0] = a[1] = a[2] = a[3] = 0.f;
    for (int f = t; f < n4; f += 2560) {  // 10 independent 16 B loads in flight per lane
        float4 v[10];
#pragma unroll
        for (int u = 0; u < 10; ++u) v[u] = p4[min(f + 256 * u, n4 - 1)];  // unconditional: all ten issue back to back
#pragma unroll
        for (int u = 0; u < 10; ++u) {
            const bool in = f + 256 * u < n4;
            a[0] += in ? v[u].x : 0.f; a[1] += in ? v[u].y : 0.f;
            a[2] += in ? v[u].z : 0.f; a[3] += in ? v[u].w : 0.f;
        }
    }
}
__device__ __forceinline__ void reduce_partials(const float* __restrict__ partials, int nb, double* s_seg /*[32][32]*/,
                                                double* s_sys /*[32]*/) {
    // The first 256 threads of the workgroup load; any workgroup size that is a multiple of 256 may call this.
    if (threadIdx.x < 256) {
        float a[4];
        partial_sums_f32(reinterpret_cast<const float4*>(partials), nb * (kIcpSlots / 4), threadIdx.x, a);
        const int row = threadIdx.x >> 3, c4 = (threadIdx.x & 7) * 4;
        s_seg[row * 32 + c4 + 0] = (double)a[0]; s_seg[row * 32 + c4 + 1] = (double)a[1];
        s_seg[row * 32 + c4 + 2] = (double)a[2]; s_seg[row * 32 + c4 + 3] = (double)a[3];
    }
    __syncthreads();
    if (threadIdx.x < 128) reduce_rows(s_seg, s_sys, threadIdx.x);
    __syncthreads();
}

// Two partial arrays (ICP and photometric) and the correspondence counts in ONE memory latency: threads 0..255 reduce the
// first array, threads 256..511 the second, wavefront 0 issues the count loads ahead of its partial loads.  Needs a
// workgroup of >= 512 threads (the kernels here have exactly that); bit-identical sums to two reduce_partials calls.
__device__ __forceinline__ void reduce_partials_pair(const float* __restrict__ pa, const float* __restrict__ pb, const int2* __restrict__ cnt,
                                                     int nb, double* s_segA, double* s_segB, double* s_sysA, double* s_sysB, int* s_cnt) {
    const int half = threadIdx.x >> 8, t = threadIdx.x & 255;
    unsigned c = 0, g = 0;
    if (threadIdx.x < 64)
        for (int i = threadIdx.x; i < nb; i += 64) { const int2 v = cnt[i]; c += (unsigned)v.x; g += (unsigned)v.y; }
    if (half < 2) {
        float a[4];
        partial_sums_f32(reinterpret_cast<const float4*>(half ? pb : pa), nb * (kIcpSlots / 4), t, a);
        double* s_seg = half ? s_segB : s_segA;
        const int row = t >> 3, c4 = (t & 7) * 4;
        s_seg[row * 32 + c4 + 0] = (double)a[0]; s_seg[row * 32 + c4 + 1] = (double)a[1];
        s_seg[row * 32 + c4 + 2] = (double)a[2]; s_seg[row * 32 + c4 + 3] = (double)a[3];
    }
    if (threadIdx.x < 64) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { c += (unsigned)__shfl_xor((int)c, o, 64); g += (unsigned)__shfl_xor((int)g, o, 64); }
        if (threadIdx.x == 0) { s_cnt[0] = (int)c; s_cnt[1] = (int)g; }
    }
    __syncthreads();
    if (t < 128 && half < 2) reduce_rows(half ? s_segB : s_segA, half ? s_sysB : s_sysA, t);
    __syncthreads();
}

// Sum of 29 (padded to 32) accumulators over the 64 lanes of a wavefront by recursive halving: at step s each lane
// keeps the half of its remaining values selected by bit s of its lane id and hands the other half to lane ^ (1 << s),
// so 16 + 8 + 4 + 2 + 1 + 1 = 32 cross-lane moves do what 29 x 6 = 174 full-width reductions did (measured: the
// 174-DPP form cost ~2.2k cycles per wavefront and dominated the launch at 4 wavefronts per SIMD).
// On return lane l holds the wave total of component bitrev5(l & 31) in v[0] (components >= 29 are zero padding).
__device__ __forceinline__ int icp_component_of_lane(int lane) {
    const int l = lane & 31;
    return ((l & 1) << 4) | ((l & 2) << 2) | (l & 4) | ((l & 8) >> 2) | ((l & 16) >> 4);
}
// bitwise select (one v_bfi_b32): written on the bit patterns so that the compiler cannot turn "cond ? v[a] : v[b]" into
// a dynamically indexed register array (which it lowers to a 32-deep compare/select chain per access).
__device__ __forceinline__ float bitsel(unsigned mask, float a, float b) {
    return __uint_as_float((__float_as_uint(a) & mask) | (__float_as_uint(b) & ~mask));
}
// Cross-lane fetch "value of lane (lane ^ kXor)".  xor 1 / 2 are quad permutes, xor 8 a 16-lane row rotation, xor 4 two row
// rotations and a select -- all DPP modifiers on VALU instructions (no LDS crossbar traffic); only xor 16 / 32 cross a DPP row
// and go through ds_bpermute.  30 of the 32 exchanges of the halving tree are xor 1..8: with every exchange on the
// bpermute path the tree cost ~5 k cycles per launch at 16 wavefronts per CU (the crossbar serialises them).
template <int kXor>
__device__ __forceinline__ float lane_xor(float v, int lane) {
    const int iv = __float_as_int(v);
    if constexpr (kXor == 1) return __int_as_float(__builtin_amdgcn_update_dpp(0, iv, 0xB1, 0xf, 0xf, false));   // quad_perm:[1,0,3,2]
    else if constexpr (kXor == 2) return __int_as_float(__builtin_amdgcn_update_dpp(0, iv, 0x4E, 0xf, 0xf, false));   // quad_perm:[2,3,0,1]
    else if constexpr (kXor == 4) {
        const int dn = __builtin_amdgcn_update_dpp(0, iv, 0x124, 0xf, 0xf, false);   // row_ror:4  -> lane i reads i - 4 (mod 16)
        const int up = __builtin_amdgcn_update_dpp(0, iv, 0x12C, 0xf, 0xf, false);   // row_ror:12 -> lane i reads i + 4 (mod 16)
        return __int_as_float((lane & 4) ? dn : up);
    } else if constexpr (kXor == 8) return __int_as_float(__builtin_amdgcn_update_dpp(0, iv, 0x128, 0xf, 0xf, false));   // row_ror:8
    else return __shfl_xor(v, kXor, 64);
}

template <int kStep>
__device__ __forceinline__ void halving_step(float (&v)[32], int lane) {
    constexpr int half = 16 >> kStep;
    const unsigned up = ((lane >> kStep) & 1) ? 0xFFFFFFFFu : 0u;
#pragma unroll
    for (int k = 0; k < half; ++k) {
        const float send = bitsel(up, v[k], v[k + half]);
        const float keep = bitsel(up, v[k + half], v[k]);
        v[k] = keep + lane_xor<(1 << kStep)>(send, lane);
    }
}

__device__ __forceinline__ float wave_sum32_halving(float (&v)[32]) {
    const int lane = threadIdx.x & 63;
    halving_step<0>(v, lane);
    halving_step<1>(v, lane);
    halving_step<2>(v, lane);
    halving_step<3>(v, lane);
    halving_step<4>(v, lane);
    return v[0] + __shfl_xor(v[0], 32, 64);
}

// Sum of the 29 accumulators over ALL kT threads of a workgroup through LDS, in a fixed order: lane pairs first (one DPP add per
// accumulator), even lanes store column-wise [component][kT / 2] (consecutive lanes -> consecutive banks), then component c is
// summed by G = kT / 32 neighbouring threads (16 values each, read as four conflict-free float4 rows) and a G-lane DPP butterfly.
// ~85 instructions per wavefront and 60 KB of LDS traffic per 512-thread workgroup against ~300 instructions per wavefront for
// the register halving tree above (which stays for the RGB-D kernels): the tree was 2.4 k of the 15 k cycles of a level-0 launch.
// out: the workgroup's 32-float partial (components >= 29 are written as zeros).
template <int kT>
__device__ __forceinline__ void block_sum29_lds(const float (&acc)[32], float* s_red /*[29][kT / 2]*/, float* __restrict__ out) {
    constexpr int kHalf = kT / 2, G = kT / 32;
    static_assert(G == 8 || G == 16, "kT must be 256 or 512");
    const int tid = threadIdx.x, lane = tid & 63;
#pragma unroll
    for (int c = 0; c < 29; ++c) {
        const float v = acc[c] + lane_xor<1>(acc[c], lane);
        if ((lane & 1) == 0) s_red[c * kHalf + (tid >> 1)] = v;
    }
    __syncthreads();
    const int c = tid / G, g = tid % G;
    float s = 0.f;
    if (c < 29) {
        const float4* __restrict__ row = reinterpret_cast<const float4*>(s_red + c * kHalf);
        float4 v[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) v[m] = row[m * G + g];
#pragma unroll
        for (int m = 0; m < 4; ++m) s += (v[m].x + v[m].y) + (v[m].z + v[m].w);
    }
    s += lane_xor<1>(s, lane);
    s += lane_xor<2>(s, lane);
    s += lane_xor<4>(s, lane);
    if (G == 16) s += lane_xor<8>(s, lane);
    if (g == 0) out[c] = s;
}

// ------------------------------------------------------------------------------------------------
// The ICP iteration kernel.
// ------------------------------------------------------------------------------------------------
// Boundary floats of the two gates on the squared quantities (see icp_accumulate): d2 = max{y : sqrtf(y) <= distThres},
// s2 = min{y : sqrtf(y) >= angleThres}.  Host libm's sqrtf is correctly rounded, like the device's.
static void icp_gates(float distThres, float angleThres, float& dist2Max, float& sine2Min) {
    float d2 = distThres * distThres;
    while (sqrtf(d2) > distThres) d2 = nextafterf(d2, 0.f);
    while (sqrtf(nextafterf(d2, INFINITY)) <= distThres) d2 = nextafterf(d2, INFINITY);
    float s2 = angleThres * angleThres;
    while (sqrtf(s2) < angleThres) s2 = nextafterf(s2, INFINITY);
    while (s2 > 0.f && sqrtf(nextafterf(s2, 0.f)) >= angleThres) s2 = nextafterf(s2, 0.f);
    dist2Max = d2; sine2Min = s2;
}

// Gauss-Newton trace of a tracking step (debug tap "gn_trace", tests/test_gpu_parity_long.py): row r = [0..31] the reduced system of iteration r
// as the device summed it (fp64: 27 packed products, sum r^2, inliers), [32..47] resultRt, [48..56] Rcurr, [57..59] tcurr as iteration r USED
// them; row n_iterations holds the state after the last update.  Written by lanes u = 0..59 of whoever finishes an iteration.
__device__ __forceinline__ void gn_trace_write(double* __restrict__ trace, int it, bool have_sys, const double* s_sys, const GNState* s_st, int u) {
    if (!trace || u < 0) return;
    if (u < 32) { if (have_sys) trace[kGnTraceRow * (it - 1) + u] = s_sys[u]; }
    else if (u < 48) trace[kGnTraceRow * it + u] = s_st->resultRt[u - 32];
    else if (u < 57) trace[kGnTraceRow * it + u] = (double)s_st->Rcurr[u - 48];
    else if (u < 60) trace[kGnTraceRow * it + u] = (double)s_st->tcurr[u - 57];
}

struct IcpKArgs {
    const float* vc; const float* nc; const float* vp; const float* np;
    int W, H; Intr k;
    float dist2Max, sine2Min;      // the 0.10 m / sin 20 deg gates of reduce.cu:326-341 on the SQUARED quantities (icp_gates)
    const float* partials_in; int nb_in;
    float* partials_out;
    const GNState* st_in; GNState* st_out;
    float* log_out;
    double* trace; int it;         // optional (parity tests): the model's Gauss-Newton trace (mf_internal.h: kGnTraceRow) and this launch's iteration
    unsigned long long* prof_out;  // optional: 16 shader-clock stamps of workgroup 0 / thread 0 (8 of the launch + 6 inside the solve)
    const PoseDev* pose_in;        // first launch of a tracking step: seed the state from the model pose
    const So3Result* so3_in;       // ... and resultRt's rotation from the SO(3) pre-alignment (RGBDOdometry.cpp:338-344)
};

// Correspondence search for one pixel, split in two so that the gathers of all four pixels of a thread are in flight
// together (a per-pixel search-then-accumulate serialises four dependent L2 round trips: ~2 us per launch).
struct IcpCorr { float3 vcurr_g, vcurr_cp, ncurr_g; int j; bool ok; };

// Per-pixel arithmetic of the ICP kernels, every operation rounded on its own and in the order of the CPU restatement the
// parity tests check against (its m33_mul / f3_dot / f3_cross: a plain reading of Core/Cuda/reduce.cu:292-415), contraction off.  Two reasons:
//  * everything that DECIDES something -- the rounded projection (ux, uy), the 0.10 m distance gate, the sin 20 degrees gate, the
//    NaN tests -- is then bit-identical to the oracle, so the inlier set (not just its size) matches by construction, and the
//    seven row entries of every inlier do too; only the accumulation of the 28 products differs (fp32 block sums here, double there);
//  * k_icp_iter<512>, k_icp_iter<256>, k_icp_batch_pixels and k_rgbd_iter are separate functions, and left to itself the compiler
//    contracts a * b + c differently in each.  The 1e-7 relative spread that causes is invisible for a room-sized model, but a freshly
//    spawned object (2 000 pixels on two box faces: a 6x6 system with condition ~1e5) turns it into millimetres of pose: the batched
//    loop and the model-by-model loop must agree pixel for pixel (tests/test_gpu_multimodel.py::test_batched_tracking_...).
// Cost: ~20 instructions per pixel (3 mul + 2 add instead of 1 mul + 2 fma per matrix row).
__device__ __forceinline__ float3 icp_mul33(const float* R, float3 v) {
#pragma clang fp contract(off)
    return f3(R[0] * v.x + R[1] * v.y + R[2] * v.z, R[3] * v.x + R[4] * v.y + R[5] * v.z, R[6] * v.x + R[7] * v.y + R[8] * v.z);
}
__device__ __forceinline__ float icp_dot3(float3 a, float3 b) {
#pragma clang fp contract(off)
    return a.x * b.x + a.y * b.y + a.z * b.z;
}
__device__ __forceinline__ float3 icp_cross3(float3 a, float3 b) {
#pragma clang fp contract(off)
    return f3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}

template <class Args>
__device__ __forceinline__ IcpCorr icp_project(float vx, float vy, float vz, float nx, float ny, float nz, const float* Rc,
                                               float3 tc, const float* Rpi, float3 tp, const Args& a) {
#pragma clang fp contract(off)
    // search(), first half: Core/Cuda/reduce.cu:292-314
    IcpCorr c;
    const float3 g = icp_mul33(Rc, f3(vx, vy, vz));
    c.vcurr_g = f3(g.x + tc.x, g.y + tc.y, g.z + tc.z);
    c.vcurr_cp = icp_mul33(Rpi, f3(c.vcurr_g.x - tp.x, c.vcurr_g.y - tp.y, c.vcurr_g.z - tp.z));
    const int ux = __float2int_rn((c.vcurr_cp.x * a.k.fx) / c.vcurr_cp.z + a.k.cx);
    const int uy = __float2int_rn((c.vcurr_cp.y * a.k.fy) / c.vcurr_cp.z + a.k.cy);
    c.ok = !(ux < 0 || uy < 0 || ux >= a.W || uy >= a.H || c.vcurr_cp.z < 0) && !isnan(nx);
    c.j = c.ok ? uy * a.W + ux : 0;
    c.ncurr_g = icp_mul33(Rc, f3(nx, ny, nz));
    return c;
}

template <class Args>
__device__ __forceinline__ void icp_accumulate(const IcpCorr& c, float3 vprev_g, float3 nprev_g, const float* Rpi, float3 tp,
                                               const Args& a, float* acc) {
#pragma clang fp contract(off)
    // search(), second half + getProducts(): Core/Cuda/reduce.cu:326-415
    // dist = |vprev_g - vcurr_g| <= distThres and sine = |ncurr_g x nprev_g| < angleThres, decided on the squares: sqrtf is monotonic and
    // correctly rounded, so sqrtf(x) <= T  <=>  x <= max{y : sqrtf(y) <= T}, and sqrtf(x) < T  <=>  x < min{y : sqrtf(y) >= T}; the host finds
    // the two boundary floats once (icp_gates) and the kernel saves two IEEE square roots (~30 instructions) per pixel -- same inlier set, bit for bit.
    const float3 dv = f3(vprev_g.x - c.vcurr_g.x, vprev_g.y - c.vcurr_g.y, vprev_g.z - c.vcurr_g.z);
    const float dist2 = icp_dot3(dv, dv);
    const float3 cr = icp_cross3(c.ncurr_g, nprev_g);
    const float sine2 = icp_dot3(cr, cr);
    const bool found = c.ok && sine2 < a.sine2Min && dist2 <= a.dist2Max && !isnan(nprev_g.x);
    if (!found) return;
    const float3 s_cp = c.vcurr_cp;
    const float3 d_cp = icp_mul33(Rpi, f3(vprev_g.x - tp.x, vprev_g.y - tp.y, vprev_g.z - tp.z));
    const float3 n_cp = icp_mul33(Rpi, nprev_g);
    const float3 sxn = icp_cross3(s_cp, n_cp);
    const float row[7] = {n_cp.x, n_cp.y, n_cp.z, sxn.x, sxn.y, sxn.z,
                          icp_dot3(n_cp, f3(s_cp.x - d_cp.x, s_cp.y - d_cp.y, s_cp.z - d_cp.z))};
    int k = 0;
#pragma unroll
    for (int r = 0; r < 7; ++r)
#pragma unroll
        for (int cc = r; cc < 7; ++cc) { acc[k] = fmaf(row[r], row[cc], acc[k]); ++k; }
    acc[28] += 1.0f;
}

__device__ __forceinline__ void seed_state(const PoseDev& pose, GNState& s) {
    for (int k = 0; k < 16; ++k) s.resultRt[k] = (k % 5 == 0) ? 1.0 : 0.0;
    for (int k = 0; k < 9; ++k) { s.Rprev[k] = pose.R[k]; s.Rcurr[k] = pose.R[k]; s.trR[k] = (k % 4 == 0) ? 1.f : 0.f; }
    for (int k = 0; k < 3; ++k) { s.tprev[k] = pose.t[k]; s.tcurr[k] = pose.t[k]; s.trt[k] = 0.f; }
    m33_inverse_f(s.Rprev, s.Rprev_inv);  // RGBDOdometry.cpp:332
    s.lastICPError = 0.f; s.lastICPCount = 0.f; s.valid = 0;
    s.levelDone = -1; s.lastRGBError = 3.4028234664e38f; s.lastRGBCount = 0.f; s.ill = 0;
}

// Workgroup shape.  Per launch a CU spends its time in three VALU-throughput-bound stretches (one VALU instruction per 4
// cycles per SIMD, 4 SIMDs): the per-pixel chain (project -> gather -> gate -> 29 products, ~250 instructions per 64
// pixels), the 32-value halving reduction (~300 instructions PER WAVEFRONT, however few pixels it carried) and the
// single-wavefront solve.  With 16 wavefronts per CU at one pixel per thread the reduction alone was 16 x 300 x 4 / 4 =
// 4 800 cycles of every launch; 8 wavefronts carrying up to 3 pixels per thread halve that while two wavefronts per SIMD
// plus all gathers of a thread in flight together still hide the gather latency.
constexpr int kIcpThreads = 512;
constexpr int kIcpPx = 3;           // pixel slots per thread (a workgroup's chunk is <= kIcpPx * kIcpThreads pixels)
constexpr int kIcpMaxBlocks = 240;  // <= one workgroup per CU (256 CUs) with a little slack for busy CUs

__host__ __device__ inline int icp_chunk(int P, int nblocks) {
    const int c = (P + nblocks - 1) / nblocks;
    return ((c + 63) / 64) * 64;
}

// kT: threads per workgroup (512 at level 0; 256 at the coarse levels, where there are fewer 512-pixel chunks than CUs:
// twice the workgroups, half the wavefronts per CU -> half the per-CU halving-reduction time)
// kPx: pixel slots per thread = ceil(chunk / kT): 3 at level 0 of a VGA frame (1280-pixel chunks), 2 at level 1 (320), 1 at level 2 (256).
// The pixel phase is VALU-bound (~330 instructions per pixel; a SIMD issues one wave-instruction per 2 cycles), so a slot nobody needs is
// not free: round 3 ran three slots at every level -- at level 2 two thirds of the phase's instructions worked on masked pixels.  A slot
// whose 64 pixels all lie beyond the chunk is skipped by the WHOLE wavefront (a scalar branch, no exec-mask bookkeeping): at level 0 the
// chunk is 2.5 slots, wavefronts 4..7 skip the third one and every SIMD carries 5 slots instead of 6.  Masked or skipped, a slot adds
// nothing to the sums: results are bit-identical to the three-slot form.
template <int kT, int kPx>
__global__ __launch_bounds__(kT) void k_icp_iter(const IcpKArgs a) {
    constexpr bool kSkipIdleSlots = kT == 256 && kPx > 1;
    __shared__ double s_seg[32 * 32];
    __shared__ double s_sys[32];
    __shared__ float s_pose[24];  // Rcurr[9] tcurr[3] Rprev_inv[9] tprev[3]
    __shared__ float s_red[29 * (kT / 2)];
    __shared__ GNState s_st;      // the Gauss-Newton state: loaded (or seeded), updated in place by gn_finish_wg, stored by workgroup 0
    __shared__ float s_T[16];

    const int tid = threadIdx.x;
    // stage the Gauss-Newton state through LDS with one coalesced load (thread 0 would otherwise chase ~80 dependent
    // scalar loads after the solve)
    if (a.pose_in == nullptr) {
        if (tid < (int)(sizeof(GNState) / 4)) reinterpret_cast<uint32_t*>(&s_st)[tid] = reinterpret_cast<const uint32_t*>(a.st_in)[tid];
    } else if (tid == 0) {
        seed_state(*a.pose_in, s_st);   // RGBDOdometry.cpp:239-243,332-336: Rprev = Rcurr = pose, resultRt = I
        if (a.so3_in)
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c) s_st.resultRt[r * 4 + c] = a.so3_in->R[r * 3 + c];
    }
    const bool prof = a.prof_out != nullptr && blockIdx.x == 0 && tid == 0;
    unsigned long long stamp[8];
    GnStamps gst;
    if (prof) stamp[0] = __builtin_amdgcn_s_memtime();
    const int P = a.W * a.H;
    // A workgroup owns a contiguous chunk of <= kPx * kT pixels (icp_grid_blocks keeps the grid <= one workgroup
    // per CU so that a launch is ONE round of workgroups: 300 workgroups of 1024 px on 256 CUs ran as two rounds and
    // doubled the level-0 launch time).  Thread t handles pixels beg + t + q * kT, q < kPx.
    const int chunk = icp_chunk(P, gridDim.x);
    const int beg = blockIdx.x * chunk, end = min(P, beg + chunk);

    // (1) issue the pose-independent streamed loads first so their latency overlaps the solve below
    int idx[kPx]; bool act[kPx], won[kPx];
    float vx[kPx], vy[kPx], vz[kPx], nx[kPx], ny[kPx], nz[kPx];
#pragma unroll
    for (int q = 0; q < kPx; ++q) {
        // lanes past the end of the chunk re-read its last pixel and are masked out of the sums below: no divergent control flow
        // around the loads / projections / gathers (the exec-mask bookkeeping was ~10 % of the pixel phase); a slot that is past
        // the end for the whole wavefront is skipped with a scalar branch
        idx[q] = beg + q * kT + tid;
        act[q] = idx[q] < end;
        // (only where whole wavefronts are known to idle: a scalar branch is a scheduling fence for the loads around it, and at level 0 --
        // every wavefront carries at least two full slots -- the fenced form measured 0.2 us SLOWER per launch, profiles/r04b_*)
        won[q] = kSkipIdleSlots ? (__builtin_amdgcn_readfirstlane(beg + q * kT + (tid & ~63)) < end) : true;
        vx[q] = vy[q] = vz[q] = nx[q] = ny[q] = nz[q] = 0.f;
        if (won[q]) {
            const int i = min(idx[q], P - 1);
            vx[q] = a.vc[i]; vy[q] = a.vc[P + i]; vz[q] = a.vc[2 * P + i];
            nx[q] = a.nc[i]; ny[q] = a.nc[P + i]; nz[q] = a.nc[2 * P + i];
        }
    }

    // (2) prologue: finish the previous iteration (reduce -> solve -> pose), identically in every workgroup
    if (prof) stamp[1] = __builtin_amdgcn_s_memtime();
    if (a.nb_in > 0) {
        reduce_partials(a.partials_in, a.nb_in, s_seg, s_sys);   // (its barriers also publish s_st)
        if (prof) stamp[2] = __builtin_amdgcn_s_memtime();
        gn_finish_wg(s_sys, &s_st, s_pose, s_T, prof, gst);     // solve, exp, pose composition: state in place, Rcurr / tcurr -> s_pose[0..11]
    } else {
        __syncthreads();
    }
    if (tid < 12) {   // the pose-independent half of s_pose, and on the first launch of a tracking step the seeded Rcurr / tcurr
        s_pose[12 + tid] = tid < 9 ? s_st.Rprev_inv[tid] : s_st.tprev[tid - 9];
        if (a.nb_in == 0) s_pose[tid] = tid < 9 ? s_st.Rcurr[tid] : s_st.tcurr[tid - 9];
    }
    if (prof) { if (a.nb_in == 0) stamp[2] = stamp[1]; stamp[3] = __builtin_amdgcn_s_memtime(); }
    __syncthreads();
    if (prof) stamp[4] = __builtin_amdgcn_s_memtime();
    if (blockIdx.x == 0) {
        // workgroup 0 publishes the state (84 words) and the iteration log (32 floats): one store instruction each, after the barrier
        constexpr int kWords = (int)(sizeof(GNState) / 4);
        static_assert(kWords + 32 <= kT, "state + log lanes");
        if (tid < kWords) reinterpret_cast<uint32_t*>(a.st_out)[tid] = reinterpret_cast<const uint32_t*>(&s_st)[tid];
        else if (tid < kWords + 32 && a.log_out && a.nb_in > 0) a.log_out[tid - kWords] = (float)s_sys[tid - kWords];
        static_assert(kWords + 32 + 60 <= kT, "trace lanes");
        gn_trace_write(a.trace, a.it, a.nb_in > 0, s_sys, &s_st, tid - (kWords + 32));
    }

    float Rc[9], Rpi[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) { Rc[k] = s_pose[k]; Rpi[k] = s_pose[12 + k]; }
    const float3 tc = f3(s_pose[9], s_pose[10], s_pose[11]);
    const float3 tp = f3(s_pose[21], s_pose[22], s_pose[23]);

    // (3) normal equations of this thread's pixel
    float acc[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) acc[k] = 0.f;
    // all projections, then all gathers (independent loads in flight together), then the products
    IcpCorr cor[kPx];
    float3 pv[kPx], pn[kPx];
#pragma unroll
    for (int q = 0; q < kPx; ++q) {
        if (!won[q]) continue;
        cor[q] = icp_project(vx[q], vy[q], vz[q], nx[q], ny[q], nz[q], Rc, tc, Rpi, tp, a);
        cor[q].ok = cor[q].ok && act[q];
        cor[q].j = cor[q].ok ? cor[q].j : 0;
    }
#pragma unroll
    for (int q = 0; q < kPx; ++q) {
        if (!won[q]) continue;
        const int j = cor[q].j;
        pv[q] = f3(a.vp[j], a.vp[P + j], a.vp[2 * P + j]);
        pn[q] = f3(a.np[j], a.np[P + j], a.np[2 * P + j]);
    }
#pragma unroll
    for (int q = 0; q < kPx; ++q)
        if (won[q]) icp_accumulate(cor[q], pv[q], pn[q], Rpi, tp, a, acc);

    // (4) wavefront reduction (halving tree), one LDS stage across the wavefronts, one 128 B partial per workgroup
    if (prof) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); stamp[5] = __builtin_amdgcn_s_memtime(); }
    block_sum29_lds<kT>(acc, s_red, a.partials_out + blockIdx.x * kIcpSlots);
    if (prof) stamp[6] = __builtin_amdgcn_s_memtime();
    if (prof) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamp[7] = __builtin_amdgcn_s_memtime();
#pragma unroll
        for (int k = 0; k < 8; ++k) a.prof_out[k] = stamp[k];
#pragma unroll
        for (int k = 0; k < 6; ++k) a.prof_out[8 + k] = a.nb_in > 0 ? gst.t[k] : 0ull;
    }
}

// threads per workgroup of the ICP-only kernel for an image of P pixels
static int icp_threads_for(int P) { return ((P + kIcpThreads - 1) / kIcpThreads < kIcpMaxBlocks) ? 256 : kIcpThreads; }

static int icp_blocks_capped(int W, int H, int cap256) {
    const int P = W * H;
    const int T = icp_threads_for(P);
    const int cap = T == 256 ? cap256 : kIcpMaxBlocks;
    int nb = (P + T - 1) / T;
    if (nb > cap) {
        nb = cap;
        while (icp_chunk(P, nb) > kIcpPx * T) ++nb;  // larger images: more than one round of workgroups
    }
    return nb;
}
// grid of the RGB-D kernels (512 threads whatever the level) and the upper bound older callers size their scratch with
int icp_grid_blocks(int W, int H) { return icp_blocks_capped(W, H, kIcpMaxBlocks); }
// grid of the geometric kernel k_icp_iter.  A level that runs 256-thread workgroups (fewer than 240 chunks of 512 pixels: levels 1 and 2 of
// a VGA frame) gets one workgroup per 256 pixels up to 320 of them -- level 1: 300 workgroups, ONE pixel slot per thread, two workgroups on
// 44 of the 256 CUs (23 KB of LDS and 4 wavefronts each) -- instead of 240 chunks of 320 pixels whose second slot kept one wavefront in
// four busy and the other three waiting for it (round 4: the pixel phase is VALU-bound, a slot costs what it costs however few lanes use it).
int icp_geo_grid_blocks(int W, int H) { return icp_blocks_capped(W, H, 320); }

// pixel slots per thread the launch needs: ceil(chunk / threads), 1..kIcpPx
int icp_pixel_slots(int W, int H) {
    const int P = W * H, T = icp_threads_for(P);
    return (icp_chunk(P, icp_geo_grid_blocks(W, H)) + T - 1) / T;
}

void launch_icp_iteration(const IcpLaunch& l, hipStream_t s) {
    IcpKArgs a;
    a.vc = l.vmap_curr; a.nc = l.nmap_curr; a.vp = l.vmap_prev; a.np = l.nmap_prev;
    a.W = l.W; a.H = l.H; a.k = l.k; icp_gates(l.distThres, l.angleThres, a.dist2Max, a.sine2Min);
    a.partials_in = l.partials_in; a.nb_in = l.nblocks_in; a.partials_out = l.partials_out;
    a.st_in = l.state_in; a.st_out = l.state_out; a.log_out = l.log_out; a.prof_out = l.prof_out; a.pose_in = l.pose_in;
    a.trace = l.trace; a.it = l.it;
    a.so3_in = l.so3_in;
    const dim3 grid(icp_geo_grid_blocks(l.W, l.H));
    const int px = icp_pixel_slots(l.W, l.H);
    if (icp_threads_for(l.W * l.H) == 256) {
        if (px <= 1) hipLaunchKernelGGL((k_icp_iter<256, 1>), grid, dim3(256), 0, s, a);
        else if (px == 2) hipLaunchKernelGGL((k_icp_iter<256, 2>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((k_icp_iter<256, kIcpPx>), grid, dim3(256), 0, s, a);
    } else {
        if (px <= 2) hipLaunchKernelGGL((k_icp_iter<kIcpThreads, 2>), grid, dim3(kIcpThreads), 0, s, a);
        else hipLaunchKernelGGL((k_icp_iter<kIcpThreads, kIcpPx>), grid, dim3(kIcpThreads), 0, s, a);
    }
}

// ------------------------------------------------------------------------------------------------
// begin / finalize
// ------------------------------------------------------------------------------------------------
// last reduce + solve of a tracking step, then Model::pose / lastPose / statistics (RGBDOdometry.cpp:476-496, Model.cpp:437-446)
// and the object-model jump rule.  Called by all 256 threads of a workgroup.
// s_st: LDS copy of the state (st_in may already point at it); s_scr: 40 floats of LDS.
__device__ __forceinline__ void icp_finalize_body(const float* __restrict__ partials_in, int nb_in, const GNState* st_in,
                                                  PoseDev* __restrict__ pose, PoseDev* __restrict__ host_mirror, float* __restrict__ log_out,
                                                  float jump_limit, const So3Result* __restrict__ so3, double* s_seg, double* s_sys,
                                                  GNState* s_st, float* s_scr, double* __restrict__ trace = nullptr, int n_it = 0) {
    if (st_in != s_st && threadIdx.x < (int)(sizeof(GNState) / 4))
        reinterpret_cast<uint32_t*>(s_st)[threadIdx.x] = reinterpret_cast<const uint32_t*>(st_in)[threadIdx.x];
    if (nb_in > 0) {
        reduce_partials(partials_in, nb_in, s_seg, s_sys);     // (its barriers also publish s_st)
        gn_finish_wg(s_sys, s_st, s_scr, s_scr + 24);
        if (log_out && threadIdx.x >= 64 && threadIdx.x < 96) log_out[threadIdx.x - 64] = (float)s_sys[threadIdx.x - 64];
    }
    __syncthreads();
    gn_trace_write(trace, n_it, nb_in > 0, s_sys, s_st, (int)threadIdx.x - 128);   // the last system and the state it led to
    if (threadIdx.x == 0) {
        const GNState& st = *s_st;
        PoseDev p = *pose;
        for (int k = 0; k < 9; ++k) { p.lastR[k] = st.Rprev[k]; p.R[k] = st.Rcurr[k]; }
        for (int k = 0; k < 3; ++k) { p.lastT[k] = st.tprev[k]; p.t[k] = st.tcurr[k]; }
        p.lastICPError = st.lastICPError;
        p.lastICPCount = st.lastICPCount;
        p.rejected = 0; p.lastRGBError = 0.f; p.lastRGBCount = 0.f;
        if (so3) { p.lastSO3Error = so3->error; p.lastSO3Count = so3->count; p.so3Iterations = so3->iterations; }
        else { p.lastSO3Error = 0.f; p.lastSO3Count = 0.f; p.so3Iterations = 0; }
        for (int k = 0; k < 3; ++k) p.incT[k] = st.trt[k];
        p.illIterations = st.ill;
        if (jump_limit > 0.f && norm3(f3(st.trt[0], st.trt[1], st.trt[2])) >= jump_limit) p.alive = 0;   // `float d > 0.2` is a DOUBLE comparison upstream (MaskFusion.cpp:268): true from 0.2f on
        pose_derive(p);
        *pose = p;
        if (host_mirror) *host_mirror = p;
    }
}

__global__ __launch_bounds__(256) void k_icp_finalize(const float* __restrict__ partials_in, int nb_in,
                                                       const GNState* __restrict__ st_in, PoseDev* __restrict__ pose,
                                                       PoseDev* __restrict__ host_mirror, float* __restrict__ log_out,
                                                       float jump_limit, const So3Result* __restrict__ so3, double* __restrict__ trace, int n_it) {
    __shared__ double s_seg[32 * 32];
    __shared__ double s_sys[32];
    __shared__ GNState s_st;
    __shared__ float s_scr[40];
    icp_finalize_body(partials_in, nb_in, st_in, pose, host_mirror, log_out, jump_limit, so3, s_seg, s_sys, &s_st, s_scr, trace, n_it);
}

void launch_icp_finalize(const float* partials_in, int nblocks_in, const GNState* state_in, PoseDev* pose,
                         PoseDev* host_mirror, float* log_out, float jump_limit, const So3Result* so3, hipStream_t s, double* trace, int n_it) {
    hipLaunchKernelGGL(k_icp_finalize, dim3(1), dim3(256), 0, s, partials_in, nblocks_in, state_in, pose, host_mirror,
                       log_out, jump_limit, so3, trace, n_it);
}

// ------------------------------------------------------------------------------------------------
// Batched Gauss-Newton loop: iteration k of ALL tracked models per launch (see mf_internal.h).
// ------------------------------------------------------------------------------------------------
struct IcpSolveArgs { TrackBatch b; int it; int nb_in; const So3Result* so3; };

__global__ __launch_bounds__(256) void k_icp_batch_solve(const IcpSolveArgs a) {
    __shared__ double s_seg[32 * 32];
    __shared__ double s_sys[32];
    __shared__ GNState s_st;
    __shared__ float s_scr[40];
    const TrackModelDev* __restrict__ md = a.b.m[blockIdx.x];
    const int tid = threadIdx.x;
    if (a.it == 0) {   // RGBDOdometry.cpp:239-243,332-344: Rprev = Rcurr = pose, resultRt = I (or the SO(3) rotation)
        if (tid == 0) {
            GNState s;
            seed_state(*md->pose, s);
            if (a.so3)
                for (int r = 0; r < 3; ++r)
                    for (int c = 0; c < 3; ++c) s.resultRt[r * 4 + c] = a.so3->R[r * 3 + c];
            md->st[0] = s;
            if (md->trace) {
                for (int u = 32; u < 60; ++u) gn_trace_write(md->trace, 0, false, nullptr, &s, u);
            }
        }
        return;
    }
    const int prev = (a.it - 1) & 1;
    if (tid < (int)(sizeof(GNState) / 4)) reinterpret_cast<uint32_t*>(&s_st)[tid] = reinterpret_cast<const uint32_t*>(md->st + prev)[tid];
    reduce_partials(md->partials[prev], a.nb_in, s_seg, s_sys);   // (its barriers also publish s_st)
    gn_finish_wg(s_sys, &s_st, s_scr, s_scr + 24);                 // the same function as k_icp_iter: bit-identical states
    __syncthreads();
    // state (84 words) and log (32 floats) leave through one store instruction each instead of ~115 scalar stores of thread 0
    constexpr int kWords = (int)(sizeof(GNState) / 4);
    if (tid < kWords) reinterpret_cast<uint32_t*>(md->st + (a.it & 1))[tid] = reinterpret_cast<const uint32_t*>(&s_st)[tid];
    else if (tid < kWords + 32 && md->log) md->log[32 * (a.it - 1) + tid - kWords] = (float)s_sys[tid - kWords];
    gn_trace_write(md->trace, a.it, true, s_sys, &s_st, tid - (kWords + 32));
}

struct IcpPxArgs {
    TrackBatch b;
    const float* vc; const float* nc;
    int W, H; Intr k;
    float dist2Max, sine2Min;
    int level, parity, chunk;
    const float2* row_z;      // [H] {min z, max z} of the valid vertices of every row of vc (launch_row_zrange); nullptr: no slab culling
    // fused != 0 ("batchSolveInPixelPass"): the launch finishes the previous iteration itself -- every workgroup of a model reduces that model's partial
    // sums and solves, identically, as k_icp_iter's prologue does for a single model -- instead of reading the state k_icp_batch_solve wrote
    int fused, it, nb_in; const So3Result* so3;
};

// {min z, max z} over the valid vertices of every row of the three levels' vertex maps (planar [3][H][W]: z is plane 2)
struct RowZArgs { const float* vm[3]; int W, H; float2* out; };
__global__ __launch_bounds__(64) void k_row_zrange(const RowZArgs a) {
    int row = blockIdx.x, level = 0, W = a.W, H = a.H, base = 0;
    while (level < 2 && row >= H) { row -= H; base += H; W >>= 1; H >>= 1; ++level; }
    const float* __restrict__ z = a.vm[level] + 2 * (size_t)W * H + (size_t)row * W;
    float lo = INFINITY, hi = -INFINITY;
    for (int x = threadIdx.x; x < W; x += 64) {
        const float v = z[x];
        if (v == v) { lo = fminf(lo, v); hi = fmaxf(hi, v); }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { lo = fminf(lo, __shfl_xor(lo, off, 64)); hi = fmaxf(hi, __shfl_xor(hi, off, 64)); }
    if (threadIdx.x == 0) a.out[base + row] = make_float2(lo, hi);
}
void launch_row_zrange(const float* const vmap[3], int W, int H, float2* out, hipStream_t s) {
    RowZArgs a{{vmap[0], vmap[1], vmap[2]}, W, H, out};
    hipLaunchKernelGGL(k_row_zrange, dim3(H + (H >> 1) + (H >> 2)), dim3(64), 0, s, a);
}
constexpr int kBatchThreads = 256;
constexpr int kBatchPx = 3;   // pixels per thread and round: all their gathers are in flight together

__global__ __launch_bounds__(kBatchThreads) void k_icp_batch_pixels(const IcpPxArgs a) {
    __shared__ float s_red[29 * (kBatchThreads / 2)];
    __shared__ double s_seg[32 * 32];
    __shared__ double s_sys[32];
    __shared__ GNState s_st;
    __shared__ float s_scr[40];
    const TrackModelDev* __restrict__ md = a.b.m[blockIdx.y];
    const float* __restrict__ vp = md->vm[a.level];
    const float* __restrict__ np = md->nm[a.level];
    const int tid = threadIdx.x;
    constexpr int kWords = (int)(sizeof(GNState) / 4);
    static_assert(kWords + 32 + 60 <= kBatchThreads, "state + log + trace lanes");
    if (a.fused) {
        // k_icp_batch_solve's work as this launch's prologue (the same functions on the same data in the same order: the same bits), in every workgroup
        // of the model; workgroup 0 of the model publishes state, log and trace for the next launch and the finalize step.  One launch per iteration
        // instead of two: a dependent launch is 3.4 us before it does anything
        if (a.it == 0) {
            if (tid == 0) {
                seed_state(*md->pose, s_st);
                if (a.so3)
                    for (int r = 0; r < 3; ++r)
                        for (int c = 0; c < 3; ++c) s_st.resultRt[r * 4 + c] = a.so3->R[r * 3 + c];
            }
            __syncthreads();
            if (blockIdx.x == 0) {
                if (tid < kWords) reinterpret_cast<uint32_t*>(md->st)[tid] = reinterpret_cast<const uint32_t*>(&s_st)[tid];
                if (tid == 0 && md->trace)
                    for (int u = 32; u < 60; ++u) gn_trace_write(md->trace, 0, false, nullptr, &s_st, u);
            }
        } else {
            const int prev = (a.it - 1) & 1;
            if (tid < kWords) reinterpret_cast<uint32_t*>(&s_st)[tid] = reinterpret_cast<const uint32_t*>(md->st + prev)[tid];
            reduce_partials(md->partials[prev], a.nb_in, s_seg, s_sys);   // (its barriers also publish s_st)
            gn_finish_wg(s_sys, &s_st, s_scr, s_scr + 24);
            __syncthreads();
            if (blockIdx.x == 0) {
                if (tid < kWords) reinterpret_cast<uint32_t*>(md->st + (a.it & 1))[tid] = reinterpret_cast<const uint32_t*>(&s_st)[tid];
                else if (tid < kWords + 32 && md->log) md->log[32 * (a.it - 1) + tid - kWords] = (float)s_sys[tid - kWords];
                gn_trace_write(md->trace, a.it, true, s_sys, &s_st, tid - (kWords + 32));
            }
        }
    } else {
        if (tid < kWords) reinterpret_cast<uint32_t*>(&s_st)[tid] = reinterpret_cast<const uint32_t*>(md->st + a.parity)[tid];
        __syncthreads();
    }
    const GNState* st = &s_st;
    float Rc[9], Rpi[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) { Rc[k] = st->Rcurr[k]; Rpi[k] = st->Rprev_inv[k]; }
    const float3 tc = f3(st->tcurr[0], st->tcurr[1], st->tcurr[2]);
    const float3 tp = f3(st->tprev[0], st->tprev[1], st->tprev[2]);
    const int P = a.W * a.H;
    const int beg = blockIdx.x * a.chunk, end = min(P, beg + a.chunk);
    // Slab culling (exact).  A frame pixel contributes only if it projects onto a pixel of the model's maps that holds a normal (reduce.cu:316-352:
    // everything else is a NaN reject), and an OBJECT's maps hold normals on the object alone -- ~1 % of the image (Q3).  The workgroup's pixels
    // are rows r0 .. r1 of the frame's vertex map; every valid vertex of those rows lies on the ray of its pixel at a depth inside the rows' range
    // [zmin, zmax] (row_z), i.e. inside the frustum section spanned by the four corner rays of the row band between those depths -- a convex
    // set, the hull of its eight corners.  The iteration's relative pose maps it to the hull of the transformed corners; if all of them lie in front
    // of the model camera, the projection of a convex set is inside the convex hull of the projected corners, hence inside their bounding box.
    // If that box (grown by 2 px against the rounding of the per-pixel arithmetic) misses the model's rectangle of normals (TrackModelDev::rect,
    // grown likewise), no pixel of the workgroup can find a correspondence: its partial sums are zero, written without touching a map.
    __shared__ int s_skip;
    if (a.row_z && beg < end && !md->allow_fill) {     // (the background's prediction covers the image: nothing to cull)
        if (tid < 64) {
            const int r0 = beg / a.W, r1 = (end - 1) / a.W;
            float zlo = INFINITY, zhi = -INFINITY;
            for (int r = r0 + tid; r <= r1; r += 64) { const float2 z = a.row_z[r]; zlo = fminf(zlo, z.x); zhi = fmaxf(zhi, z.y); }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) { zlo = fminf(zlo, __shfl_xor(zlo, off, 64)); zhi = fmaxf(zhi, __shfl_xor(zhi, off, 64)); }
            // lanes 0..7: corner (x side, row side, depth end) of the slab
            const float cxs = (tid & 1) ? (float)(a.W - 1) : 0.f, cys = (tid & 2) ? (float)r1 : (float)r0, cz = (tid & 4) ? zhi : zlo;
            const float3 p = f3(cz * (cxs - a.k.cx) / a.k.fx, cz * (cys - a.k.cy) / a.k.fy, cz);           // createVMap, cudafuncs.cu:118-123
            const float3 g = icp_mul33(Rc, p);
            const float3 q = icp_mul33(Rpi, f3(g.x + tc.x - tp.x, g.y + tc.y - tp.y, g.z + tc.z - tp.z));
            const bool front = q.z > 1e-3f;
            float u = front ? q.x * a.k.fx / q.z + a.k.cx : 0.f, v = front ? q.y * a.k.fy / q.z + a.k.cy : 0.f;
            float ulo = tid < 8 ? u : INFINITY, uhi = tid < 8 ? u : -INFINITY, vlo = tid < 8 ? v : INFINITY, vhi = tid < 8 ? v : -INFINITY;
            int all_front = (tid >= 8 || front) ? 1 : 0;
#pragma unroll
            for (int off = 4; off > 0; off >>= 1) {
                ulo = fminf(ulo, __shfl_xor(ulo, off, 64)); uhi = fmaxf(uhi, __shfl_xor(uhi, off, 64));
                vlo = fminf(vlo, __shfl_xor(vlo, off, 64)); vhi = fmaxf(vhi, __shfl_xor(vhi, off, 64));
                all_front &= __shfl_xor(all_front, off, 64);
            }
            if (tid == 0) {
                // (level-0 rectangle: a coarser level's normals lie inside it shifted right by the level -- k_model_pyramid)
                const int rc0 = md->rect[0], rc1 = md->rect[1], rc2 = md->rect[2], rc3 = md->rect[3];
                bool skip = !(zlo <= zhi);                                        // no valid vertex in these rows at all
                if (!skip && all_front && ulo == ulo && uhi == uhi && vlo == vlo && vhi == vhi) {
                    const bool empty = rc0 > rc2 || rc1 > rc3;                   // the model's maps hold no normal
                    skip = empty || uhi + 2.5f < (float)(rc0 >> a.level) || ulo - 2.5f > (float)(rc2 >> a.level) || vhi + 2.5f < (float)(rc1 >> a.level) ||
                           vlo - 2.5f > (float)(rc3 >> a.level);
                }
                s_skip = skip ? 1 : 0;
            }
        }
        __syncthreads();
        if (s_skip) {
            if (tid < kIcpSlots) md->partials[a.parity][blockIdx.x * kIcpSlots + tid] = 0.f;
            return;
        }
    }
    float acc[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) acc[k] = 0.f;
    for (int base = beg; base < end; base += kBatchPx * kBatchThreads) {
        bool act[kBatchPx];
        float vx[kBatchPx], vy[kBatchPx], vz[kBatchPx], nx[kBatchPx], ny[kBatchPx], nz[kBatchPx];
#pragma unroll
        for (int q = 0; q < kBatchPx; ++q) {
            const int i0 = base + q * kBatchThreads + tid;
            act[q] = i0 < end;
            const int i = min(i0, P - 1);      // masked slots re-read a valid pixel: no divergent control flow (see k_icp_iter)
            vx[q] = a.vc[i]; vy[q] = a.vc[P + i]; vz[q] = a.vc[2 * P + i];
            nx[q] = a.nc[i]; ny[q] = a.nc[P + i]; nz[q] = a.nc[2 * P + i];
        }
        IcpCorr cor[kBatchPx];
        float3 pv[kBatchPx], pn[kBatchPx];
#pragma unroll
        for (int q = 0; q < kBatchPx; ++q) {
            cor[q] = icp_project(vx[q], vy[q], vz[q], nx[q], ny[q], nz[q], Rc, tc, Rpi, tp, a);
            cor[q].ok = cor[q].ok && act[q];
            cor[q].j = cor[q].ok ? cor[q].j : 0;
        }
#pragma unroll
        for (int q = 0; q < kBatchPx; ++q) {
            const int j = cor[q].j;
            pv[q] = f3(vp[j], vp[P + j], vp[2 * P + j]);
            pn[q] = f3(np[j], np[P + j], np[2 * P + j]);
        }
#pragma unroll
        for (int q = 0; q < kBatchPx; ++q) icp_accumulate(cor[q], pv[q], pn[q], Rpi, tp, a, acc);
    }
    block_sum29_lds<kBatchThreads>(acc, s_red, md->partials[a.parity] + blockIdx.x * kIcpSlots);
}

struct IcpFinArgs { TrackBatch b; int n_it; int nb_in; const So3Result* so3; };
__global__ __launch_bounds__(256) void k_icp_batch_finalize(const IcpFinArgs a) {
    __shared__ double s_seg[32 * 32];
    __shared__ double s_sys[32];
    __shared__ GNState s_st;
    __shared__ float s_scr[40];
    const TrackModelDev* __restrict__ md = a.b.m[blockIdx.x];
    const int last = (a.n_it - 1) & 1;
    icp_finalize_body(a.n_it > 0 ? md->partials[last] : nullptr, a.n_it > 0 ? a.nb_in : 0, md->st + (a.n_it > 0 ? last : 0), md->pose, md->pose_host,
                      (md->log && a.n_it > 0) ? md->log + 32 * (a.n_it - 1) : nullptr, md->jump_limit, a.so3, s_seg, s_sys, &s_st, s_scr, md->trace, a.n_it);
    if (threadIdx.x < 4) md->rect[threadIdx.x] = (threadIdx.x & 2) ? (int)0x80000000 : 0x7FFFFFFF;   // armed for the next frame's model pyramid
}

// Workgroups per model of the pixel pass: enough of them over all models to fill the GPU a few times over (there is no
// per-workgroup prologue to amortise, only one 32-value halving tree per wavefront), never less than one round of pixels.
int icp_batch_max_blocks(int W, int H) { return (W * H + kBatchPx * kBatchThreads - 1) / (kBatchPx * kBatchThreads); }
static int icp_batch_chunk(int P, int n_models) {
    const int round = kBatchPx * kBatchThreads;
    int per_model = (768 + n_models - 1) / n_models;                  // ~3 workgroups per CU in total
    int chunk = (P + per_model - 1) / per_model;
    chunk = ((chunk + round - 1) / round) * round;
    return chunk < round ? round : chunk;
}
int icp_batch_blocks(int W, int H, int n_models) {
    const int P = W * H, chunk = icp_batch_chunk(P, n_models);
    return (P + chunk - 1) / chunk;
}
void launch_icp_batch_solve(const TrackBatch& b, int it, int nb_in, const So3Result* so3, hipStream_t s) {
    IcpSolveArgs a{b, it, nb_in, so3};
    hipLaunchKernelGGL(k_icp_batch_solve, dim3(b.n), dim3(256), 0, s, a);
}
void launch_icp_batch_pixels(const TrackBatch& b, int it, int level, const float* vmap_curr, const float* nmap_curr, int W, int H, Intr k,
                             float distThres, float angleThres, hipStream_t s, const float2* row_z, bool fused, int nb_in, const So3Result* so3) {
    IcpPxArgs a;
    a.row_z = row_z;
    a.fused = fused ? 1 : 0; a.it = it; a.nb_in = nb_in; a.so3 = so3;
    a.b = b; a.vc = vmap_curr; a.nc = nmap_curr; a.W = W; a.H = H; a.k = k; icp_gates(distThres, angleThres, a.dist2Max, a.sine2Min);
    a.level = level; a.parity = it & 1; a.chunk = icp_batch_chunk(W * H, b.n);
    hipLaunchKernelGGL(k_icp_batch_pixels, dim3(icp_batch_blocks(W, H, b.n), b.n), dim3(kBatchThreads), 0, s, a);
}
void launch_icp_batch_finalize(const TrackBatch& b, int n_it, int nb_in, const So3Result* so3, hipStream_t s) {
    IcpFinArgs a{b, n_it, nb_in, so3};
    hipLaunchKernelGGL(k_icp_batch_finalize, dim3(b.n), dim3(256), 0, s, a);
}

// stand-alone Gauss-Newton update for the parity tests (mf_k_gn_solve): the production one-thread path (unpack -> LDL^T -> exp ->
// pose composition) and the wave-parallel Gauss-Jordan of the RGB-D kernels on the same system
struct GnSolveOut { double x_serial[6], x_wave[6], resultRt[16]; float Rcurr[9], tcurr[3], trR[9], trt[3], lastICPError, lastICPCount; };
__global__ __launch_bounds__(128) void k_gn_solve_test(const double* __restrict__ sys29, const double* __restrict__ resultRt,
                                                      const float* __restrict__ Rprev, const float* __restrict__ tprev, GnSolveOut* out) {
    __shared__ double s_sys[32];
    __shared__ GNState s_st;
    __shared__ float s_pose[24];
    __shared__ float s_T[16];
    if (threadIdx.x < 32) s_sys[threadIdx.x] = threadIdx.x < 29 ? sys29[threadIdx.x] : 0.0;
    if (threadIdx.x == 0) {
        GNState& in = s_st;
        for (int k = 0; k < 16; ++k) in.resultRt[k] = resultRt[k];
        for (int k = 0; k < 9; ++k) { in.Rprev[k] = Rprev[k]; in.Rcurr[k] = Rprev[k]; in.trR[k] = 0.f; }
        for (int k = 0; k < 3; ++k) { in.tprev[k] = tprev[k]; in.tcurr[k] = tprev[k]; in.trt[k] = 0.f; }
        m33_inverse_f(in.Rprev, in.Rprev_inv);
        in.lastICPError = in.lastICPCount = 0.f; in.valid = 0; in.levelDone = -1; in.lastRGBError = 0.f; in.lastRGBCount = 0.f; in.ill = 0;
    }
    __syncthreads();
    double xw[6];
    solve6_wave(s_sys, xw);
    // what k_icp_iter / k_icp_batch_solve / k_icp_finalize call: twelve lanes, state in LDS (the serial form -- gn_solve_update_serial --
    // stays as the executable specification: tests/test_devmath_host.py compiles it for the host)
    gn_finish_wg(s_sys, &s_st, s_pose, s_T);
    __syncthreads();
    if (threadIdx.x != 0) return;
    double A[6][6], b[6], x[6];
    int shift = 0;
    for (int i = 0; i < 6; ++i)
        for (int j = i; j < 7; ++j) {
            const double value = s_sys[shift++];
            if (j == 6) b[i] = value;
            else { A[i][j] = value; A[j][i] = value; }
        }
    ldlt6_solve(A, b, x);
    const GNState& st = s_st;
    for (int k = 0; k < 6; ++k) { out->x_serial[k] = x[k]; out->x_wave[k] = xw[k]; }
    for (int k = 0; k < 16; ++k) out->resultRt[k] = st.resultRt[k];
    for (int k = 0; k < 9; ++k) { out->Rcurr[k] = st.Rcurr[k]; out->trR[k] = st.trR[k]; }
    for (int k = 0; k < 3; ++k) { out->tcurr[k] = st.tcurr[k]; out->trt[k] = st.trt[k]; }
    out->lastICPError = st.lastICPError; out->lastICPCount = st.lastICPCount;
}
int gn_solve_standalone(const double* sys29, const double* resultRt16, const float* Rprev9, const float* tprev3, double* x_serial, double* x_wave,
                        double* resultRt_out, float* Rcurr9, float* tcurr3, float* stats2, hipStream_t s) {
    char* buf = nullptr;
    const size_t in_bytes = 29 * 8 + 16 * 8 + 12 * 4;
    if (hipMalloc((void**)&buf, in_bytes + sizeof(GnSolveOut)) != hipSuccess) return -1;
    double* d_sys = reinterpret_cast<double*>(buf);
    double* d_rt = d_sys + 29;
    float* d_R = reinterpret_cast<float*>(d_rt + 16);
    GnSolveOut* d_out = reinterpret_cast<GnSolveOut*>(buf + in_bytes);
    GnSolveOut h;
    bool ok = hipMemcpyAsync(d_sys, sys29, 29 * 8, hipMemcpyHostToDevice, s) == hipSuccess &&
              hipMemcpyAsync(d_rt, resultRt16, 16 * 8, hipMemcpyHostToDevice, s) == hipSuccess &&
              hipMemcpyAsync(d_R, Rprev9, 36, hipMemcpyHostToDevice, s) == hipSuccess &&
              hipMemcpyAsync(d_R + 9, tprev3, 12, hipMemcpyHostToDevice, s) == hipSuccess;
    if (ok) {
        hipLaunchKernelGGL(k_gn_solve_test, dim3(1), dim3(128), 0, s, d_sys, d_rt, d_R, d_R + 9, d_out);   // two wavefronts: gn_finish_wg's statistics run in the second
        ok = hipMemcpyAsync(&h, d_out, sizeof(h), hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
    }
    (void)hipFree(buf);
    if (!ok) return -2;
    memcpy(x_serial, h.x_serial, 48); memcpy(x_wave, h.x_wave, 48); memcpy(resultRt_out, h.resultRt, 128);
    memcpy(Rcurr9, h.Rcurr, 36); memcpy(tcurr3, h.tcurr, 12);
    stats2[0] = h.lastICPError; stats2[1] = h.lastICPCount;
    return 0;
}

// stand-alone icpStep for the parity tests: reduce partials to 32 floats
__global__ __launch_bounds__(256) void k_icp_reduce_only(const float* __restrict__ partials, int nb, float* __restrict__ out32) {
    __shared__ double s_seg[32 * 32];
    __shared__ double s_sys[32];
    reduce_partials(partials, nb, s_seg, s_sys);
    if (threadIdx.x < 32) out32[threadIdx.x] = (float)s_sys[threadIdx.x];
}

__global__ void k_state_from_args(GNState* st, const float* Rc, const float* tc, const float* Rpi, const float* tp) {
    if (threadIdx.x != 0) return;
    GNState s;
    for (int k = 0; k < 16; ++k) s.resultRt[k] = (k % 5 == 0) ? 1.0 : 0.0;
    for (int k = 0; k < 9; ++k) { s.Rcurr[k] = Rc[k]; s.Rprev_inv[k] = Rpi[k]; s.Rprev[k] = 0.f; s.trR[k] = 0.f; }
    for (int k = 0; k < 3; ++k) { s.tcurr[k] = tc[k]; s.tprev[k] = tp[k]; s.trt[k] = 0.f; }
    s.lastICPError = s.lastICPCount = 0.f; s.valid = 0; s.levelDone = -1; s.lastRGBError = 3.4028234664e38f; s.lastRGBCount = 0.f; s.ill = 0;
    *st = s;
}

void launch_icp_step_standalone(const float* Rcurr, const float* tcurr, const float* vc, const float* nc, const float* Rpi,
                                const float* tprev, Intr k, const float* vp, const float* np, float distThres,
                                float angleThres, int W, int H, float* partials, GNState* st, float* out32, hipStream_t s) {
    // partials: scratch of icp_geo_grid_blocks(W,H) * 32 floats; st: two GNState
    hipLaunchKernelGGL(k_state_from_args, dim3(1), dim3(64), 0, s, st, Rcurr, tcurr, Rpi, tprev);
    IcpLaunch l;
    l.vmap_curr = vc; l.nmap_curr = nc; l.vmap_prev = vp; l.nmap_prev = np; l.W = W; l.H = H; l.k = k;
    l.distThres = distThres; l.angleThres = angleThres; l.partials_in = nullptr; l.nblocks_in = 0;
    l.partials_out = partials; l.state_in = st; l.state_out = st + 1; l.log_out = nullptr; l.prof_out = nullptr;
    launch_icp_iteration(l, s);
    hipLaunchKernelGGL(k_icp_reduce_only, dim3(1), dim3(256), 0, s, partials, icp_geo_grid_blocks(W, H), out32);
}

// ------------------------------------------------------------------------------------------------
// RGB-D Gauss-Newton loop (icpWeight < 100 or rgbOnly): two launches per iteration.
//   A  k_rgbd_iter : [finish the previous iteration: reduce ICP + RGB partials, combine, solve, pose] ->
//                    ICP normal equations + photometric correspondences (RgbCorr per pixel, count / sum diff^2)
//   B  k_rgb_step  : [sigma = reduced count] -> photometric normal equations from the correspondences
// B exists because the photometric weight 1 / (count + |diff|) needs the GLOBAL correspondence count of the same
// iteration (RGBDOdometry.cpp:389-401,436); a launch boundary is the only grid-wide ordering point used here.
// The ICP-only loop keeps its own tuned kernel (k_icp_iter).
// ------------------------------------------------------------------------------------------------
struct RgbdKArgs {
    IcpKArgs icp;                 // maps, intrinsics, thresholds, ICP partials in/out, state in/out, pose_in
    RgbLevel L;                   // photometric inputs of THIS launch's level
    RgbCorr* corres;              // [W*H] out
    const float* rgb_partials_in; // [nb_in][32]  (B of the previous iteration)
    const int2* cnt_in;           // [nb_in]      (A of the previous iteration): {count, sum diff^2}
    int2* cnt_out;                // [gridDim.x]
    float icpWeight; int icpOn; int rgbOnly;
    int level, prev_level;
    const So3Result* so3_in;      // first launch: rotation seed of resultRt (nullptr: identity)
};

// Sum of nb int2 records by the first wavefront; result broadcast through LDS (s_cnt[2]).  int32 wrap-around like the
// reference's int2 sums (reduce.cu:716-772).
__device__ __forceinline__ void reduce_counts(const int2* __restrict__ cnt, int nb, int* s_cnt) {
    if (threadIdx.x < 64) {
        unsigned c = 0, g = 0;
        for (int i = threadIdx.x; i < nb; i += 64) { const int2 v = cnt[i]; c += (unsigned)v.x; g += (unsigned)v.y; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { c += (unsigned)__shfl_xor((int)c, o, 64); g += (unsigned)__shfl_xor((int)g, o, 64); }
        if (threadIdx.x == 0) { s_cnt[0] = (int)c; s_cnt[1] = (int)g; }
    }
    __syncthreads();
}

__device__ __forceinline__ float rgb_tmp_error(int count, int sigma) {  // RGBDOdometry.cpp:389
    return (float)(sqrt((double)sigma) / (double)count);
}

// Finishes the iteration whose partial sums are given; every workgroup calls it with all threads and gets the same
// answer.  Returns true in the one thread (0) that holds `out`.
__device__ __forceinline__ bool finish_rgbd_iteration(const float* icp_partials, const float* rgb_partials, const int2* cnt, int nb,
                                                      float icpWeight, int icpOn, int rgbOnly, int prev_level, const GNState& in,
                                                      GNState& out, double* s_seg, double* s_seg2, double* s_sys, double* s_sys2,
                                                      int* s_cnt, float* log_out) {
    if (blockDim.x >= 512) {
        reduce_partials_pair(icp_partials, rgb_partials, cnt, nb, s_seg, s_seg2, s_sys, s_sys2, s_cnt);
    } else {
        reduce_partials(icp_partials, nb, s_seg, s_sys);
        reduce_partials(rgb_partials, nb, s_seg, s_sys2);
        reduce_counts(cnt, nb, s_cnt);
    }
    if (threadIdx.x >= 64) return false;
    double x[6];
    const double w = (double)icpWeight;
    solve6_wave_rgbd(s_sys, s_sys2, icpOn ? w * w : 0.0, icpOn ? w : 0.0, x);
    if (threadIdx.x != 0) return false;
    if (in.levelDone == prev_level) { out = in; return true; }             // that level's loop was already left
    const int count = s_cnt[0], sigma = s_cnt[1];
    const float tmpError = rgb_tmp_error(count, sigma);
    if (rgbOnly && tmpError > in.lastRGBError) {                            // RGBDOdometry.cpp:392-394: break
        out = in;
        out.levelDone = prev_level;
        return true;
    }
    gn_update_from_x(x, (float)s_sys[27], (float)s_sys[28], in, out);
    if (!icpOn) { out.lastICPError = in.lastICPError; out.lastICPCount = in.lastICPCount; }
    out.lastRGBError = tmpError;
    out.lastRGBCount = (float)count;
    if (log_out)
        for (int k = 0; k < 32; ++k) log_out[k] = (float)s_sys2[k];
    return true;
}

__global__ __launch_bounds__(kIcpThreads) void k_rgbd_iter(const RgbdKArgs a) {
    __shared__ double s_seg[32 * 32];
    __shared__ double s_seg2[32 * 32];
    __shared__ double s_sys[32];
    __shared__ double s_sys2[32];
    __shared__ float s_pose[24];   // Rcurr[9] tcurr[3] Rprev_inv[9] tprev[3]
    __shared__ float s_krk[12];    // K R K^-1 [9], K t [3] of this iteration
    __shared__ float s_part[(kIcpThreads / 64) * kIcpSlots];
    __shared__ int s_cnt[2];
    __shared__ int s_icnt[(kIcpThreads / 64) * 2];
    __shared__ int s_skip;
    __shared__ GNState s_st;

    const int tid = threadIdx.x;
    const IcpKArgs& ia = a.icp;
    if (ia.pose_in == nullptr) {
        if (tid < (int)(sizeof(GNState) / 4)) reinterpret_cast<uint32_t*>(&s_st)[tid] = reinterpret_cast<const uint32_t*>(ia.st_in)[tid];
    } else if (tid == 0) {
        seed_state(*ia.pose_in, s_st);
        if (a.so3_in)  // RGBDOdometry.cpp:338-344
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c) s_st.resultRt[r * 4 + c] = a.so3_in->R[r * 3 + c];
    }
    __syncthreads();
    const int P = ia.W * ia.H;
    const int chunk = icp_chunk(P, gridDim.x);
    const int beg = blockIdx.x * chunk, end = min(P, beg + chunk);

    // prologue: finish the previous iteration, then derive what this iteration needs
    GNState st;
    bool mine;
    if (ia.nb_in > 0) {
        mine = finish_rgbd_iteration(ia.partials_in, a.rgb_partials_in, a.cnt_in, ia.nb_in, a.icpWeight, a.icpOn, a.rgbOnly, a.prev_level,
                                     s_st, st, s_seg, s_seg2, s_sys, s_sys2, s_cnt, (blockIdx.x == 0) ? ia.log_out : nullptr);
    } else {
        mine = tid == 0;
        if (mine) st = s_st;
    }
    if (mine) {
        if (a.level != a.prev_level) st.lastRGBError = 3.4028234664e38f;   // RGBDOdometry.cpp:359
#pragma unroll
        for (int k = 0; k < 9; ++k) { s_pose[k] = st.Rcurr[k]; s_pose[12 + k] = st.Rprev_inv[k]; }
#pragma unroll
        for (int k = 0; k < 3; ++k) { s_pose[9 + k] = st.tcurr[k]; s_pose[21 + k] = st.tprev[k]; }
        float krk[9], kt[3];
        krk_from_result(st.resultRt, ia.k, krk, kt);
        for (int k = 0; k < 9; ++k) s_krk[k] = krk[k];
        for (int k = 0; k < 3; ++k) s_krk[9 + k] = kt[k];
        s_skip = st.levelDone == a.level;
        if (blockIdx.x == 0) *ia.st_out = st;
    }
    __syncthreads();
    const bool skip = s_skip != 0;
    const int lane = tid & 63, wave = tid >> 6;

    // ---- ICP normal equations (as k_icp_iter, <= 2 px per thread)
    float acc[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) acc[k] = 0.f;
    if (a.icpOn && !skip) {
        float Rc[9], Rpi[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) { Rc[k] = s_pose[k]; Rpi[k] = s_pose[12 + k]; }
        const float3 tc = f3(s_pose[9], s_pose[10], s_pose[11]);
        const float3 tp = f3(s_pose[21], s_pose[22], s_pose[23]);
        for (int i = beg + tid; i < end; i += kIcpThreads) {
            const IcpCorr c = icp_project(ia.vc[i], ia.vc[P + i], ia.vc[2 * P + i], ia.nc[i], ia.nc[P + i], ia.nc[2 * P + i], Rc, tc, Rpi,
                                          tp, ia);
            const int j = c.j;
            const float3 pv = f3(ia.vp[j], ia.vp[P + j], ia.vp[2 * P + j]);
            const float3 pn = f3(ia.np[j], ia.np[P + j], ia.np[2 * P + j]);
            icp_accumulate(c, pv, pn, Rpi, tp, ia, acc);
        }
    }
    const float wsum = wave_sum32_halving(acc);
    if (lane < 32) s_part[wave * kIcpSlots + icp_component_of_lane(lane)] = wsum;

    // ---- photometric correspondences of this iteration
    unsigned cnt = 0, sig = 0;
    if (!skip) {
        float krk[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) krk[k] = s_krk[k];
        const float3 kt = f3(s_krk[9], s_krk[10], s_krk[11]);
        for (int i = beg + tid; i < end; i += kIcpThreads) {
            const int y = i / a.L.W, x = i - y * a.L.W;
            RgbCorr c;
            if (rgb_residual_px(a.L, krk, kt, x, y, c)) { cnt += 1u; sig += (unsigned)(int)(c.diff * c.diff); }
            a.corres[i] = c;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { cnt += (unsigned)__shfl_xor((int)cnt, o, 64); sig += (unsigned)__shfl_xor((int)sig, o, 64); }
    if (lane == 0) { s_icnt[wave * 2] = (int)cnt; s_icnt[wave * 2 + 1] = (int)sig; }
    __syncthreads();
    if (tid < kIcpSlots) {
        float sv = 0.f;
        if (tid < 29) {
#pragma unroll
            for (int w = 0; w < kIcpThreads / 64; ++w) sv += s_part[w * kIcpSlots + tid];
        }
        ia.partials_out[blockIdx.x * kIcpSlots + tid] = sv;
    }
    if (tid == 0) {
        unsigned c = 0, g = 0;
        for (int w = 0; w < kIcpThreads / 64; ++w) { c += (unsigned)s_icnt[w * 2]; g += (unsigned)s_icnt[w * 2 + 1]; }
        a.cnt_out[blockIdx.x] = make_int2((int)c, (int)g);
    }
}

struct RgbStepKArgs {
    RgbLevel L; Intr k; int level;
    const RgbCorr* corres; const int2* cnt_in; int nb;   // of the A launch of the same iteration
    const GNState* st;                                    // state written by that launch
    int rgbOnly; float sobelScale;
    float* partials_out;                                  // [gridDim.x][32]
};

__global__ __launch_bounds__(kIcpThreads) void k_rgb_step(const RgbStepKArgs a) {
    __shared__ float s_part[(kIcpThreads / 64) * kIcpSlots];
    __shared__ int s_cnt[2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    reduce_counts(a.cnt_in, a.nb, s_cnt);
    const int count = s_cnt[0], sigma = s_cnt[1];
    const float tmpError = rgb_tmp_error(count, sigma);
    float sigmaVal = (tmpError == 0.f) ? 1.f : (float)count;   // RGBDOdometry.cpp:390
    if (a.rgbOnly) sigmaVal = -1.f;                             // :399-401
    const bool skip = a.st->levelDone == a.level;
    const int P = a.L.W * a.L.H;
    const int chunk = icp_chunk(P, gridDim.x);
    const int beg = blockIdx.x * chunk, end = min(P, beg + chunk);
    float acc[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) acc[k] = 0.f;
    if (!skip) {
        for (int i = beg + tid; i < end; i += kIcpThreads) {
            const RgbCorr c = a.corres[i];
            if (c.u0 < 0) continue;
            const int y = i / a.L.W, x = i - y * a.L.W;
            rgb_step_px(a.L, c, x, y, sigmaVal, a.k, a.sobelScale, acc);
        }
    }
    const float wsum = wave_sum32_halving(acc);
    if (lane < 32) s_part[wave * kIcpSlots + icp_component_of_lane(lane)] = wsum;
    __syncthreads();
    if (tid < kIcpSlots) {
        float sv = 0.f;
        if (tid < 29) {
#pragma unroll
            for (int w = 0; w < kIcpThreads / 64; ++w) sv += s_part[w * kIcpSlots + tid];
        }
        a.partials_out[blockIdx.x * kIcpSlots + tid] = sv;
    }
}

__global__ __launch_bounds__(256) void k_rgbd_finalize(const float* __restrict__ icp_partials, const float* __restrict__ rgb_partials,
                                                        const int2* __restrict__ cnt, int nb, float icpWeight, int icpOn, int rgbOnly,
                                                        int rgbOn, int prev_level, const GNState* __restrict__ st_in,
                                                        const So3Result* __restrict__ so3, PoseDev* __restrict__ pose,
                                                        PoseDev* __restrict__ host_mirror, float* __restrict__ log_out, float jump_limit) {
    __shared__ double s_seg[32 * 32];
    __shared__ double s_sys[32];
    __shared__ double s_sys2[32];
    __shared__ int s_cnt[2];
    __shared__ GNState s_st;
    if (threadIdx.x < (int)(sizeof(GNState) / 4)) reinterpret_cast<uint32_t*>(&s_st)[threadIdx.x] = reinterpret_cast<const uint32_t*>(st_in)[threadIdx.x];
    __syncthreads();
    GNState st;
    bool mine;
    if (nb > 0) {
        mine = finish_rgbd_iteration(icp_partials, rgb_partials, cnt, nb, icpWeight, icpOn, rgbOnly, prev_level, s_st, st, s_seg, s_seg,
                                     s_sys, s_sys2, s_cnt, log_out);
    } else {
        mine = threadIdx.x == 0;
        if (mine) st = s_st;
    }
    if (!mine) return;
    PoseDev p = *pose;
    p.rejected = 0;
    const float dx = st.tcurr[0] - st.tprev[0], dy = st.tcurr[1] - st.tprev[1], dz = st.tcurr[2] - st.tprev[2];
    if (rgbOn && (double)sqrtf(dx * dx + dy * dy + dz * dz) > 0.3) {   // RGBDOdometry.cpp:477-481
        for (int k = 0; k < 9; ++k) st.Rcurr[k] = st.Rprev[k];
        for (int k = 0; k < 3; ++k) { st.tcurr[k] = st.tprev[k]; st.trt[k] = 0.f; }
        p.rejected = 1;
    }
    for (int k = 0; k < 9; ++k) { p.lastR[k] = st.Rprev[k]; p.R[k] = st.Rcurr[k]; }
    for (int k = 0; k < 3; ++k) { p.lastT[k] = st.tprev[k]; p.t[k] = st.tcurr[k]; }
    p.lastICPError = st.lastICPError; p.lastICPCount = st.lastICPCount;
    p.lastRGBError = st.lastRGBError; p.lastRGBCount = st.lastRGBCount;
    if (so3) { p.lastSO3Error = so3->error; p.lastSO3Count = so3->count; p.so3Iterations = so3->iterations; }
    else { p.lastSO3Error = 0.f; p.lastSO3Count = 0.f; p.so3Iterations = 0; }
    for (int k = 0; k < 3; ++k) p.incT[k] = st.trt[k];
    p.illIterations = st.ill;   // (the photometric loop does not count: its combined system goes through the wave Gauss-Jordan)
    if (jump_limit > 0.f && norm3(f3(st.trt[0], st.trt[1], st.trt[2])) >= jump_limit) p.alive = 0;   // `float d > 0.2` is a DOUBLE comparison upstream (MaskFusion.cpp:268): true from 0.2f on
    pose_derive(p);
    *pose = p;
    if (host_mirror) *host_mirror = p;
}

void launch_rgbd_iteration(const RgbdLaunch& l, hipStream_t s) {
    RgbdKArgs a;
    const IcpLaunch& il = l.icp;
    a.icp.vc = il.vmap_curr; a.icp.nc = il.nmap_curr; a.icp.vp = il.vmap_prev; a.icp.np = il.nmap_prev;
    a.icp.W = il.W; a.icp.H = il.H; a.icp.k = il.k; icp_gates(il.distThres, il.angleThres, a.icp.dist2Max, a.icp.sine2Min);
    a.icp.partials_in = il.partials_in; a.icp.nb_in = il.nblocks_in; a.icp.partials_out = il.partials_out;
    a.icp.st_in = il.state_in; a.icp.st_out = il.state_out; a.icp.log_out = il.log_out; a.icp.prof_out = nullptr; a.icp.pose_in = il.pose_in; a.icp.so3_in = nullptr;
    a.L = l.L; a.corres = l.corres; a.rgb_partials_in = l.rgb_partials_in; a.cnt_in = l.cnt_in; a.cnt_out = l.cnt_out;
    a.icpWeight = l.icpWeight; a.icpOn = l.icpOn; a.rgbOnly = l.rgbOnly; a.level = l.level; a.prev_level = l.prev_level; a.so3_in = l.so3_in;
    const int nb = icp_grid_blocks(il.W, il.H);
    hipLaunchKernelGGL(k_rgbd_iter, dim3(nb), dim3(kIcpThreads), 0, s, a);
    RgbStepKArgs b;
    b.L = l.L; b.k = il.k; b.level = l.level; b.corres = l.corres; b.cnt_in = l.cnt_out; b.nb = nb; b.st = il.state_out;
    b.rgbOnly = l.rgbOnly; b.sobelScale = l.sobelScale; b.partials_out = l.rgb_partials_out;
    hipLaunchKernelGGL(k_rgb_step, dim3(nb), dim3(kIcpThreads), 0, s, b);
}

void launch_rgbd_finalize(const float* icp_partials, const float* rgb_partials, const int2* cnt, int nb, float icpWeight, int icpOn,
                          int rgbOnly, int rgbOn, int prev_level, const GNState* st_in, const So3Result* so3, PoseDev* pose,
                          PoseDev* host_mirror, float* log_out, float jump_limit, hipStream_t s) {
    hipLaunchKernelGGL(k_rgbd_finalize, dim3(1), dim3(256), 0, s, icp_partials, rgb_partials, cnt, nb, icpWeight, icpOn, rgbOnly, rgbOn,
                       prev_level, st_in, so3, pose, host_mirror, log_out, jump_limit);
}

// ------------------------------------------------------------------------------------------------
// Model-side pyramid: copyMaps + resize x2 + transform x3 (+ fill-in) in ONE pass.  One thread owns a 4x4 block of
// level-0 pixels = 2x2 of level 1 = 1 pixel of level 2, so the averages follow the reference's operation order
// ((x00 + x01 + x10 + x11) / 4, level 2 from level-1 values) and nothing is re-read from HBM.
// ------------------------------------------------------------------------------------------------
struct MapPx { float3 v, n; bool vok, nok; };

__device__ __forceinline__ MapPx load_model_px(const float4* __restrict__ predV, const float4* __restrict__ predN,
                                               const float* __restrict__ fillDepth, bool useFill, int x, int y, int W,
                                               int H, Intr k) {
    const int p = y * W + x;
    float4 v4 = predV[p], n4 = predN[p];
    if (useFill) {
        if (v4.z == 0) {  // fill_vertex.frag:37-53
            const float z = fillDepth[p];
            v4 = make_float4(((float)x - k.cx) * z * (1.0f / k.fx), ((float)y - k.cy) * z * (1.0f / k.fy), z, 1.f);
        }
        if (n4.z == 0) {  // fill_normal.frag:34-50
            const float3 vp = get_vertex(fillDepth, W, H, x, y, (float)x, (float)y, k);
            const float3 n = get_normal_forward(fillDepth, W, H, x, y, vp, k);
            n4 = make_float4(n.x, n.y, n.z, 1.f);
        }
    }
    MapPx r;
    if (!(v4.z == 0)) {  // copyMapsKernel, cudafuncs.cu:286-305
        r.v = f3(v4.x, v4.y, v4.z); r.n = f3(n4.x, n4.y, n4.z);
        r.vok = !isnan(r.v.x); r.nok = !isnan(r.n.x);
    } else {
        r.v = r.n = f3(qnan(), qnan(), qnan());
        r.vok = r.nok = false;
    }
    return r;
}

// R a with the fused multiply-adds written out.  This file allows contraction, and under "fast" the back end decides per use WHICH product of
// (R0 ax + R1 ay) + R2 az keeps its own rounding -- by operand order after scheduling: moving this kernel's body into a function (round 6) turned
// the level-2 normals' R1 ay into R0 ax and every fifth normal moved by an ulp, enough to shift a tracked pose by 1e-7 and flip a surfel of a
// count-exact test.  The form below is the one every use of the kernel compiled to through rounds 1-5; written out it no longer depends on context.
__device__ __forceinline__ float3 mul33_fixed(const float* R, float3 a) {
#pragma clang fp contract(off)
    return f3(fmaf(R[2], a.z, fmaf(R[0], a.x, R[1] * a.y)), fmaf(R[5], a.z, fmaf(R[3], a.x, R[4] * a.y)),
              fmaf(R[8], a.z, fmaf(R[6], a.x, R[7] * a.y)));
}

__device__ __forceinline__ void store_tx(float* __restrict__ vm, float* __restrict__ nm, int P, int i, float3 v, bool vok,
                                         float3 n, bool nok, const float* R, float3 t) {
    // tranformMapsKernel, cudafuncs.cu:207-249
    float3 vd = f3(qnan(), qnan(), qnan()), nd = vd;
    if (vok) vd = mul33_fixed(R, v) + t;
    if (nok) nd = mul33_fixed(R, n);
    vm[i] = vd.x; vm[P + i] = vd.y; vm[2 * P + i] = vd.z;
    nm[i] = nd.x; nm[P + i] = nd.y; nm[2 * P + i] = nd.z;
}

struct PyrArgs {
    const float4* predV; const float4* predN; const float* fillDepth; const FrameDev* frame; const PoseDev* pose;
    float R[9]; float t[3]; int hostPose;
    float* vm[3]; float* nm[3];
    int W, H; Intr k;
    TrackBatch b;   // b.n > 0: grid.z = model, everything but fillDepth / W / H / k comes from the model's block
};

// value of lane (quad base + kB) for every lane of a quad: one DPP quad_perm broadcast, no LDS traffic
template <int kB>
__device__ __forceinline__ float quad_bcast(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), kB | (kB << 2) | (kB << 4) | (kB << 6), 0xf, 0xf, false));
}
template <int kB>
__device__ __forceinline__ float3 quad_bcast3(float3 v) { return f3(quad_bcast<kB>(v.x), quad_bcast<kB>(v.y), quad_bcast<kB>(v.z)); }

// One thread per LEVEL-1 pixel (2x2 level-0 pixels); the four lanes of a DPP quad hold the 2x2 level-1 block of one level-2
// pixel and exchange their values with quad broadcasts, so the averages keep the reference's operation order
// ((x00 + x01 + x10 + x11) / 4, level 2 from level-1 values).  (A thread per level-2 pixel -- 19 200 threads walking 16
// pixels each -- left three quarters of the CUs idle: 15 us.)
__device__ __forceinline__ void model_pyramid_body(const PyrArgs& a0, const int blk_x, const int blk_y, const int blk_z) {
    PyrArgs a = a0;
    if (a0.b.n > 0) {
        const TrackModelDev* __restrict__ md = a0.b.m[blk_z];
        a.predV = md->predV; a.predN = md->predN; a.frame = md->frame; a.pose = md->pose; a.hostPose = 0;
        a.fillDepth = md->allow_fill ? a0.fillDepth : nullptr;
#pragma unroll
        for (int i = 0; i < 3; ++i) { a.vm[i] = md->vm[i]; a.nm[i] = md->nm[i]; }
    }
    const int W2 = a.W >> 2, H2 = a.H >> 2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = lane & 3, bx = b & 1, by = b >> 1;
    const int x2 = blk_x * 16 + (lane >> 2);
    const int y2 = blk_y * 4 + wave;
    const bool inside = x2 < W2 && y2 < H2;   // whole quads are in or out: the DPP exchanges below stay well defined
    float R[9]; float3 t;
    if (a.hostPose) {
#pragma unroll
        for (int k = 0; k < 9; ++k) R[k] = a.R[k];
        t = f3(a.t[0], a.t[1], a.t[2]);
    } else {
#pragma unroll
        for (int k = 0; k < 9; ++k) R[k] = a.pose->R[k];
        t = f3(a.pose->t[0], a.pose->t[1], a.pose->t[2]);
    }
    const bool useFill = a.fillDepth != nullptr && a.frame != nullptr && a.frame->useFillIn != 0;
    const int W = a.W, H = a.H, W1 = W >> 1, H1 = H >> 1;
    const int P0 = W * H, P1 = W1 * H1, P2 = W2 * H2;

    float3 v1 = f3(qnan(), qnan(), qnan()), n1 = v1;
    bool v1ok = false, n1ok = false;
    bool n0any = false;                       // (batched tracker) this thread's level-0 pixels that hold a normal: their box
    int n0x0 = 0, n0y0 = 0, n0x1 = 0, n0y1 = 0;
    if (inside) {
        MapPx px[4];
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const int x = 4 * x2 + 2 * bx + dx, y = 4 * y2 + 2 * by + dy;
                px[dy * 2 + dx] = load_model_px(a.predV, a.predN, a.fillDepth, useFill, x, y, W, H, a.k);
                store_tx(a.vm[0], a.nm[0], P0, y * W + x, px[dy * 2 + dx].v, px[dy * 2 + dx].vok, px[dy * 2 + dx].n,
                         px[dy * 2 + dx].nok, R, t);
                if (px[dy * 2 + dx].nok) {
                    n0x0 = n0any ? min(n0x0, x) : x; n0y0 = n0any ? min(n0y0, y) : y; n0x1 = n0any ? max(n0x1, x) : x; n0y1 = n0any ? max(n0y1, y) : y;
                    n0any = true;
                }
            }
        // resizeMapKernel<false/true>, cudafuncs.cu:366-417: order x00 + x01 + x10 + x11
        v1ok = px[0].vok && px[1].vok && px[2].vok && px[3].vok;
        n1ok = px[0].nok && px[1].nok && px[2].nok && px[3].nok;
        v1 = f3((px[0].v.x + px[1].v.x + px[2].v.x + px[3].v.x) / 4, (px[0].v.y + px[1].v.y + px[2].v.y + px[3].v.y) / 4,
                (px[0].v.z + px[1].v.z + px[2].v.z + px[3].v.z) / 4);
        n1 = normalized_rsqrt(f3((px[0].n.x + px[1].n.x + px[2].n.x + px[3].n.x) / 4,
                                 (px[0].n.y + px[1].n.y + px[2].n.y + px[3].n.y) / 4,
                                 (px[0].n.z + px[1].n.z + px[2].n.z + px[3].n.z) / 4));
        if (!v1ok) v1 = f3(qnan(), qnan(), qnan());
        if (!n1ok) n1 = f3(qnan(), qnan(), qnan());
        n1ok = n1ok && !isnan(n1.x);
        const int x1 = 2 * x2 + bx, y1 = 2 * y2 + by;
        store_tx(a.vm[1], a.nm[1], P1, y1 * W1 + x1, v1, v1ok, n1, n1ok, R, t);
    }
    // level 2: the quad's four level-1 values in block order b = by * 2 + bx (executed by every lane: DPP needs them active)
    const float okv = v1ok ? 1.f : 0.f, okn = n1ok ? 1.f : 0.f;
    const float3 va = quad_bcast3<0>(v1), vb = quad_bcast3<1>(v1), vc = quad_bcast3<2>(v1), vd = quad_bcast3<3>(v1);
    const float3 na = quad_bcast3<0>(n1), nb = quad_bcast3<1>(n1), nc = quad_bcast3<2>(n1), nd = quad_bcast3<3>(n1);
    // (every broadcast is executed by every lane -- no short-circuit: the wavefront reductions below need the lanes in step)
    const float ov0 = quad_bcast<0>(okv), ov1 = quad_bcast<1>(okv), ov2 = quad_bcast<2>(okv), ov3 = quad_bcast<3>(okv);
    const float on0 = quad_bcast<0>(okn), on1 = quad_bcast<1>(okn), on2 = quad_bcast<2>(okn), on3 = quad_bcast<3>(okn);
    const bool v2ok = (ov0 != 0.f) & (ov1 != 0.f) & (ov2 != 0.f) & (ov3 != 0.f);
    bool n2ok = (on0 != 0.f) & (on1 != 0.f) & (on2 != 0.f) & (on3 != 0.f);
    if (inside && b == 0) {
        float3 v2 = f3((va.x + vb.x + vc.x + vd.x) / 4, (va.y + vb.y + vc.y + vd.y) / 4, (va.z + vb.z + vc.z + vd.z) / 4);
        float3 n2 = normalized_rsqrt(f3((na.x + nb.x + nc.x + nd.x) / 4, (na.y + nb.y + nc.y + nd.y) / 4, (na.z + nb.z + nc.z + nd.z) / 4));
        if (!v2ok) v2 = f3(qnan(), qnan(), qnan());
        if (!n2ok) n2 = f3(qnan(), qnan(), qnan());
        n2ok = n2ok && !isnan(n2.x);
        store_tx(a.vm[2], a.nm[2], P2, y2 * W2 + x2, v2, v2ok, n2, n2ok, R, t);
    }
    if (a0.b.n > 0 && !a0.b.m[blk_z]->allow_fill) {
        // batched tracker, object models: the rectangle of level-0 pixels that hold a normal (TrackModelDev::rect) -- a superset of the pixels store_tx
        // wrote a normal to (it writes NaN wherever the flag is off), which is all the pixel pass of the Gauss-Newton loop can pair a frame pixel
        // with.  A level-1 / level-2 normal needs all four normals below it (resizeMapKernel), so the rectangle shifted right by the level bounds
        // those levels too.  An object covers ~1 % of the image: nearly every wavefront sees no normal at all and leaves after one ballot.
        if (__ballot(inside && n0any) != 0ull) {
            int* __restrict__ rect = a0.b.m[blk_z]->rect;
            const bool on = inside && n0any;
            const int x0 = wave_min_i(on ? n0x0 : 0x7FFFFFFF), y0 = wave_min_i(on ? n0y0 : 0x7FFFFFFF);
            const int x1 = wave_max_i(on ? n0x1 : (int)0x80000000), y1 = wave_max_i(on ? n0y1 : (int)0x80000000);
            if (lane == 0) { atomicMin(&rect[0], x0); atomicMin(&rect[1], y0); atomicMax(&rect[2], x1); atomicMax(&rect[3], y1); }
        }
    }
}

__global__ __launch_bounds__(256) void k_model_pyramid(const PyrArgs a) { model_pyramid_body(a, (int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z); }

// The depth filter of the frame and the model-side pyramid of the same frame's tracking step in ONE launch ("fusedPreprocessLaunch"): the two are
// independent -- the filter reads the new depth image, the pyramid the previous frame's prediction -- and complement each other (the filter is
// VALU-bound on 1 208 workgroups at VGA, the pyramid a 20 MB stream on 300), but on one stream two launches run one after the other.  Workgroups
// below nbil run bilateral_body (mf_bilateral_device.h), the others model_pyramid_body with their index as the 2-D block of k_model_pyramid:
// the same instructions on the same data as the two kernels, hence the same bits.
// (six wavefronts per SIMD = six workgroups per compute unit: at VGA all 1 508 workgroups are resident at once; the pyramid half would take 83 registers)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6, 8))) void k_bilateral_model_pyramid(const float* __restrict__ depth, float* __restrict__ out, int W, int H, int nbil,
                                                                 int interleave, const PyrArgs a) {
    __shared__ float tile[kBLdsH * kBLdsW];
    // Which half a workgroup belongs to.  Workgroups are dispatched in index order: with the filter's first and the pyramids' behind them, a launch
    // that does not fit the GPU at once (1280 x 960: 4 808 + 1 200 per tracked model) ran the two halves one after the other.  The pyramids' workgroups
    // are therefore spread evenly among the filter's (a Bresenham walk: workgroup b is a pyramid workgroup when floor((b + 1) npyr / total) steps),
    // so that the VALU-bound half and the bandwidth-bound half are in flight together from the first round on.
    const int gx = ((W >> 2) + 15) / 16, gy = ((H >> 2) + 3) / 4, per = gx * gy;
    // (a launch that IS resident at once -- VGA, one tracked model: 1 508 workgroups -- keeps the filter's workgroups first: their index is what
    // places a tile on the XCD that holds its neighbours' halo rows, worth 1 us there)
    const int total = (int)gridDim.x, npyr = total - nbil, b = (int)blockIdx.x;
    int before, after;
    if (interleave) { before = (int)(((long long)b * npyr) / total); after = (int)(((long long)(b + 1) * npyr) / total); }
    else { before = b < nbil ? 0 : b - nbil; after = b < nbil ? 0 : b - nbil + 1; }
    if (after == before) { bilateral_body(depth, out, W, H, tile, b - before); return; }
    // (a.b.n > 0: the batched tracker's pyramids, model by model -- k_model_pyramid's grid.z unrolled)
    const int j = before;
    model_pyramid_body(a, (j % per) % gx, (j % per) / gx, j / per);
}

void launch_model_pyramid(const float4* predV, const float4* predN, const float* fillDepth, const FrameDev* frame,
                          const PoseDev* pose, const float* R9t3_host_or_null, float* const vmaps[3], float* const nmaps[3],
                          int W, int H, Intr k, hipStream_t s) {
    PyrArgs a;
    a.predV = predV; a.predN = predN; a.fillDepth = fillDepth; a.frame = frame; a.pose = pose;
    a.hostPose = R9t3_host_or_null != nullptr;
    for (int i = 0; i < 9; ++i) a.R[i] = a.hostPose ? R9t3_host_or_null[i] : 0.f;
    for (int i = 0; i < 3; ++i) a.t[i] = a.hostPose ? R9t3_host_or_null[9 + i] : 0.f;
    for (int i = 0; i < 3; ++i) { a.vm[i] = vmaps[i]; a.nm[i] = nmaps[i]; }
    a.W = W; a.H = H; a.k = k;
    a.b.n = 0;
    dim3 grid(((W >> 2) + 15) / 16, ((H >> 2) + 3) / 4);
    hipLaunchKernelGGL(k_model_pyramid, grid, dim3(256), 0, s, a);
}

void launch_bilateral_model_pyramid(const float* depth, float* depthF, const float4* predV, const float4* predN, const float* fillDepth,
                                    const FrameDev* frame, const PoseDev* pose, float* const vmaps[3], float* const nmaps[3], int W, int H, Intr k,
                                    hipStream_t s) {
    PyrArgs a;
    a.predV = predV; a.predN = predN; a.fillDepth = fillDepth; a.frame = frame; a.pose = pose;
    a.hostPose = 0;
    for (int i = 0; i < 9; ++i) a.R[i] = 0.f;
    for (int i = 0; i < 3; ++i) a.t[i] = 0.f;
    for (int i = 0; i < 3; ++i) { a.vm[i] = vmaps[i]; a.nm[i] = nmaps[i]; }
    a.W = W; a.H = H; a.k = k;
    a.b.n = 0;
    const int nbil = bilateral_grid(W, H), npyr = (((W >> 2) + 15) / 16) * (((H >> 2) + 3) / 4);
    hipLaunchKernelGGL(k_bilateral_model_pyramid, dim3(nbil + npyr), dim3(256), 0, s, depth, depthF, W, H, nbil, (nbil + npyr > 1536) ? 1 : 0, a);
}

// ... and for the batched tracker: the filter beside launch_model_pyramid_batch's work
void launch_bilateral_model_pyramid_batch(const float* depth, float* depthF, const TrackBatch& b, const float* fillDepth, int W, int H, Intr k,
                                          hipStream_t s) {
    PyrArgs a;
    memset(&a, 0, sizeof(a));
    a.fillDepth = fillDepth; a.W = W; a.H = H; a.k = k; a.b = b;
    const int nbil = bilateral_grid(W, H), npyr = (((W >> 2) + 15) / 16) * (((H >> 2) + 3) / 4) * b.n;
    hipLaunchKernelGGL(k_bilateral_model_pyramid, dim3(nbil + npyr), dim3(256), 0, s, depth, depthF, W, H, nbil, (nbil + npyr > 1536) ? 1 : 0, a);
}

void launch_model_pyramid_batch(const TrackBatch& b, const float* fillDepth, int W, int H, Intr k, hipStream_t s) {
    PyrArgs a;
    memset(&a, 0, sizeof(a));
    a.fillDepth = fillDepth; a.W = W; a.H = H; a.k = k; a.b = b;
    dim3 grid(((W >> 2) + 15) / 16, ((H >> 2) + 3) / 4, b.n);
    hipLaunchKernelGGL(k_model_pyramid, grid, dim3(256), 0, s, a);
}

}  // namespace mf

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
// the kernel boundary only), then streams the current vertex/normal planes (16 B per lane), gathers the model
// planes, and reduces 29 accumulators with DPP row operations + one LDS stage.  No host round trip, no atomics,
// bit-reproducible run to run.
#include "mf_device.h"

namespace mf {

// ------------------------------------------------------------------------------------------------
// small fp64 / fp32 linear algebra for the per-iteration solve (single thread)
// ------------------------------------------------------------------------------------------------
__device__ void m33_inverse_f(const float* m, float* inv) {  // cofactor inverse (Eigen fixed-size stand-in)
    const float c00 = m[4] * m[8] - m[5] * m[7];
    const float c01 = m[5] * m[6] - m[3] * m[8];
    const float c02 = m[3] * m[7] - m[4] * m[6];
    const float det = m[0] * c00 + m[1] * c01 + m[2] * c02;
    const float id = 1.0f / det;
    inv[0] = c00 * id; inv[1] = (m[2] * m[7] - m[1] * m[8]) * id; inv[2] = (m[1] * m[5] - m[2] * m[4]) * id;
    inv[3] = c01 * id; inv[4] = (m[0] * m[8] - m[2] * m[6]) * id; inv[5] = (m[2] * m[3] - m[0] * m[5]) * id;
    inv[6] = c02 * id; inv[7] = (m[1] * m[6] - m[0] * m[7]) * id; inv[8] = (m[0] * m[4] - m[1] * m[3]) * id;
}

// LDL^T with diagonal pivoting in double; zero pivots give zero components (Eigen::LDLT::solve behaviour).
__device__ void ldlt6_solve(double* A /*36, destroyed*/, double* b /*6, destroyed*/, double* x) {
    int perm[6];
    double maxdiag = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i) { perm[i] = i; maxdiag = fmax(maxdiag, fabs(A[i * 6 + i])); }
    const double tol = maxdiag * 1e-300 + 1e-300;
    for (int k = 0; k < 6; ++k) {
        int p = k;
        for (int i = k + 1; i < 6; ++i)
            if (fabs(A[i * 6 + i]) > fabs(A[p * 6 + p])) p = i;
        if (p != k) {
            for (int j = 0; j < 6; ++j) { double t = A[k * 6 + j]; A[k * 6 + j] = A[p * 6 + j]; A[p * 6 + j] = t; }
            for (int j = 0; j < 6; ++j) { double t = A[j * 6 + k]; A[j * 6 + k] = A[j * 6 + p]; A[j * 6 + p] = t; }
            { double t = b[k]; b[k] = b[p]; b[p] = t; }
            { int t = perm[k]; perm[k] = perm[p]; perm[p] = t; }
        }
        const double d = A[k * 6 + k];
        if (fabs(d) <= tol) continue;
        for (int i = k + 1; i < 6; ++i) {
            const double l = A[i * 6 + k] / d;
            for (int j = k + 1; j < 6; ++j) A[i * 6 + j] -= l * A[k * 6 + j];
            A[i * 6 + k] = l;
        }
    }
    double y[6], z[6];
    for (int i = 0; i < 6; ++i) {
        double s = b[i];
        for (int j = 0; j < i; ++j) s -= A[i * 6 + j] * y[j];
        y[i] = s;
    }
    for (int i = 0; i < 6; ++i) y[i] = (fabs(A[i * 6 + i]) > tol) ? y[i] / A[i * 6 + i] : 0.0;
    for (int i = 5; i >= 0; --i) {
        double s = y[i];
        for (int j = i + 1; j < 6; ++j) s -= A[j * 6 + i] * z[j];
        z[i] = s;
    }
    for (int i = 0; i < 6; ++i) x[perm[i]] = z[i];
}

// OdometryProvider::rodrigues (Core/Utils/OdometryProvider.h:32-67)
__device__ void rodrigues_d(const double* w, double* R) {
    for (int k = 0; k < 9; ++k) R[k] = (k % 4 == 0) ? 1.0 : 0.0;
    double rx = w[0], ry = w[1], rz = w[2];
    const double theta = sqrt(rx * rx + ry * ry + rz * rz);
    if (theta >= 2.2204460492503131e-16) {
        const double c = cos(theta), s = sin(theta), c1 = 1. - c;
        const double itheta = 1. / theta;
        rx *= itheta; ry *= itheta; rz *= itheta;
        const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        const double rrt[9] = {rx * rx, rx * ry, rx * rz, rx * ry, ry * ry, ry * rz, rx * rz, ry * rz, rz * rz};
        const double rx_[9] = {0, -rz, ry, rz, 0, -rx, -ry, rx, 0};
        for (int k = 0; k < 9; ++k) R[k] = c * I[k] + c1 * rrt[k] + s * rx_[k];
    }
}

// One Gauss-Newton update from the reduced system `sys` (27 upper-tri products, sum r^2, inliers):
// RGBDOdometry.cpp:428-474 (icp && !rgb branch) + OdometryProvider::computeUpdateSE3.
__device__ void gn_solve_update(const double* sys, const GNState& in, GNState& out) {
    out = in;
    double A[36], b[6], x[6];
    int shift = 0;
    for (int i = 0; i < 6; ++i)
        for (int j = i; j < 7; ++j) {
            const double value = (double)(float)sys[shift++];  // the reference hands float A/b to the host
            if (j == 6) b[i] = value;
            else A[j * 6 + i] = A[i * 6 + j] = value;
        }
    const float res = (float)sys[27], inl = (float)sys[28];
    out.lastICPError = sqrtf(res) / inl;
    out.lastICPCount = inl;
    ldlt6_solve(A, b, x);
    // resultRt <- [exp(w) | t] * resultRt
    double Rw[9], Rt[16], nr[16];
    rodrigues_d(x + 3, Rw);
    for (int k = 0; k < 16; ++k) Rt[k] = (k % 5 == 0) ? 1.0 : 0.0;
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) Rt[r * 4 + c] = Rw[r * 3 + c];
        Rt[r * 4 + 3] = x[r];
    }
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) {
            double s = 0;
            for (int k = 0; k < 4; ++k) s += Rt[r * 4 + k] * in.resultRt[k * 4 + c];
            nr[r * 4 + c] = s;
        }
    for (int k = 0; k < 16; ++k) out.resultRt[k] = nr[k];
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) out.trR[r * 3 + c] = (float)nr[r * 4 + c];
        out.trt[r] = (float)nr[r * 4 + 3];
    }
    // currentT = [Rprev|tprev] * transform.inverse()
    float iR[9];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) iR[r * 3 + c] = out.trR[c * 3 + r];
    const float3 itv = mul33(iR, f3(out.trt[0], out.trt[1], out.trt[2]));
    const float it3[3] = {-itv.x, -itv.y, -itv.z};
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c)
            out.Rcurr[r * 3 + c] = in.Rprev[r * 3 + 0] * iR[0 * 3 + c] + in.Rprev[r * 3 + 1] * iR[1 * 3 + c] +
                                   in.Rprev[r * 3 + 2] * iR[2 * 3 + c];
    const float3 tv = mul33(in.Rprev, f3(it3[0], it3[1], it3[2]));
    out.tcurr[0] = tv.x + in.tprev[0]; out.tcurr[1] = tv.y + in.tprev[1]; out.tcurr[2] = tv.z + in.tprev[2];
    out.valid = 1;
}

// Fixed-order reduction of `nb` per-workgroup partials ([nb][32] floats) by a 256-thread workgroup -> sys[32] doubles
// in LDS.  Thread t sums component (t & 31) over workgroups (t >> 5), (t >> 5) + 8, ...; lanes 0..31 then add the
// 8 segment sums in order.
__device__ __forceinline__ void reduce_partials(const float* __restrict__ partials, int nb, double* s_seg /*[8][32]*/,
                                                double* s_sys /*[32]*/) {
    const int c = threadIdx.x & 31, seg = threadIdx.x >> 5;
    double acc = 0.0;
    for (int b = seg; b < nb; b += 8) acc += (double)partials[b * kIcpSlots + c];
    s_seg[seg * 32 + c] = acc;
    __syncthreads();
    if (threadIdx.x < 32) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k) s += s_seg[k * 32 + threadIdx.x];
        s_sys[threadIdx.x] = s;
    }
    __syncthreads();
}

// ------------------------------------------------------------------------------------------------
// wave64 sum via DPP row operations (one VALU op per step); the total lands in lane 63.
// ------------------------------------------------------------------------------------------------
template <int kCtrl>
__device__ __forceinline__ float dpp_add(float v) {
    const int o = __builtin_amdgcn_update_dpp(0, __float_as_int(v), kCtrl, 0xf, 0xf, false);
    return v + __int_as_float(o);
}
__device__ __forceinline__ float wave_sum_to_lane63(float v) {
#if defined(MF_REDUCE_SHFL)
    return wave_sum(v);
#else
    v = dpp_add<0xb1>(v);   // quad_perm:[1,0,3,2]
    v = dpp_add<0x4e>(v);   // quad_perm:[2,3,0,1]
    v = dpp_add<0x124>(v);  // row_ror:4
    v = dpp_add<0x128>(v);  // row_ror:8
    v = dpp_add<0x142>(v);  // row_bcast:15
    v = dpp_add<0x143>(v);  // row_bcast:31
    return v;
#endif
}

// ------------------------------------------------------------------------------------------------
// The ICP iteration kernel.  256 threads, 4 consecutive pixels per thread (16 B streamed loads per plane).
// ------------------------------------------------------------------------------------------------
struct IcpKArgs {
    const float* vc; const float* nc; const float* vp; const float* np;
    int W, H; Intr k;
    float distThres, angleThres;
    const float* partials_in; int nb_in;
    float* partials_out;
    const GNState* st_in; GNState* st_out;
    float* log_out;
};

__device__ __forceinline__ void icp_pixel(float vx, float vy, float vz, float nx, float ny, float nz, const float* Rc,
                                          float3 tc, const float* Rpi, float3 tp, const IcpKArgs& a, int P, float* acc) {
    // search(): Core/Cuda/reduce.cu:292-353
    const float3 vcurr = f3(vx, vy, vz);
    const float3 vcurr_g = mul33(Rc, vcurr) + tc;
    const float3 vcurr_cp = mul33(Rpi, vcurr_g - tp);
    const int ux = __float2int_rn(vcurr_cp.x * a.k.fx / vcurr_cp.z + a.k.cx);
    const int uy = __float2int_rn(vcurr_cp.y * a.k.fy / vcurr_cp.z + a.k.cy);
    if (ux < 0 || uy < 0 || ux >= a.W || uy >= a.H || vcurr_cp.z < 0) return;
    const int j = uy * a.W + ux;
    const float3 vprev_g = f3(a.vp[j], a.vp[P + j], a.vp[2 * P + j]);
    const float3 nprev_g = f3(a.np[j], a.np[P + j], a.np[2 * P + j]);
    const float3 ncurr_g = mul33(Rc, f3(nx, ny, nz));
    const float dist = norm3(vprev_g - vcurr_g);
    const float sine = norm3(cross3(ncurr_g, nprev_g));
    if (!(sine < a.angleThres && dist <= a.distThres && !isnan(nx) && !isnan(nprev_g.x))) return;
    // getProducts(): Core/Cuda/reduce.cu:355-415
    const float3 s_cp = vcurr_cp;
    const float3 d_cp = mul33(Rpi, vprev_g - tp);
    const float3 n_cp = mul33(Rpi, nprev_g);
    const float3 sxn = cross3(s_cp, n_cp);
    const float row[7] = {n_cp.x, n_cp.y, n_cp.z, sxn.x, sxn.y, sxn.z, dot3(n_cp, s_cp - d_cp)};
    int k = 0;
#pragma unroll
    for (int r = 0; r < 7; ++r)
#pragma unroll
        for (int c = r; c < 7; ++c) acc[k++] += row[r] * row[c];
    acc[28] += 1.0f;
}

__global__ __launch_bounds__(256) void k_icp_iter(const IcpKArgs a) {
    __shared__ double s_seg[8 * 32];
    __shared__ double s_sys[32];
    __shared__ float s_pose[24];  // Rcurr[9] tcurr[3] Rprev_inv[9] tprev[3]
    __shared__ float s_part[4 * kIcpSlots];

    const int tid = threadIdx.x;
    const int P = a.W * a.H;
    const int q = blockIdx.x * 256 + tid;  // pixel quad
    const bool active = q * 4 < P;

    // (1) issue the pose-independent streamed loads first so their latency overlaps the solve below
    float4 vx, vy, vz, nx, ny, nz;
    if (active) {
        const float4* vc4 = reinterpret_cast<const float4*>(a.vc);
        const float4* nc4 = reinterpret_cast<const float4*>(a.nc);
        const int P4 = P >> 2;
        vx = vc4[q]; vy = vc4[P4 + q]; vz = vc4[2 * P4 + q];
        nx = nc4[q]; ny = nc4[P4 + q]; nz = nc4[2 * P4 + q];
    }

    // (2) prologue: finish the previous iteration (reduce -> solve -> pose), identically in every workgroup
    if (a.nb_in > 0) {
        reduce_partials(a.partials_in, a.nb_in, s_seg, s_sys);
        if (tid == 0) {
            GNState st;
            gn_solve_update(s_sys, *a.st_in, st);
#pragma unroll
            for (int k = 0; k < 9; ++k) { s_pose[k] = st.Rcurr[k]; s_pose[12 + k] = st.Rprev_inv[k]; }
#pragma unroll
            for (int k = 0; k < 3; ++k) { s_pose[9 + k] = st.tcurr[k]; s_pose[21 + k] = st.tprev[k]; }
            if (blockIdx.x == 0) {
                *a.st_out = st;
                if (a.log_out)
                    for (int k = 0; k < 32; ++k) a.log_out[k] = (float)s_sys[k];
            }
        }
    } else if (tid == 0) {
        const GNState& st = *a.st_in;
#pragma unroll
        for (int k = 0; k < 9; ++k) { s_pose[k] = st.Rcurr[k]; s_pose[12 + k] = st.Rprev_inv[k]; }
#pragma unroll
        for (int k = 0; k < 3; ++k) { s_pose[9 + k] = st.tcurr[k]; s_pose[21 + k] = st.tprev[k]; }
        if (blockIdx.x == 0) *a.st_out = st;
    }
    __syncthreads();

    float Rc[9], Rpi[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) { Rc[k] = s_pose[k]; Rpi[k] = s_pose[12 + k]; }
    const float3 tc = f3(s_pose[9], s_pose[10], s_pose[11]);
    const float3 tp = f3(s_pose[21], s_pose[22], s_pose[23]);

    // (3) normal equations of this thread's 4 pixels
    float acc[29];
#pragma unroll
    for (int k = 0; k < 29; ++k) acc[k] = 0.f;
    if (active) {
        icp_pixel(vx.x, vy.x, vz.x, nx.x, ny.x, nz.x, Rc, tc, Rpi, tp, a, P, acc);
        icp_pixel(vx.y, vy.y, vz.y, nx.y, ny.y, nz.y, Rc, tc, Rpi, tp, a, P, acc);
        icp_pixel(vx.z, vy.z, vz.z, nx.z, ny.z, nz.z, Rc, tc, Rpi, tp, a, P, acc);
        icp_pixel(vx.w, vy.w, vz.w, nx.w, ny.w, nz.w, Rc, tc, Rpi, tp, a, P, acc);
    }

    // (4) wavefront reduction (DPP), one LDS stage across the 4 wavefronts, one 128 B partial per workgroup
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int k = 0; k < 29; ++k) {
        const float s = wave_sum_to_lane63(acc[k]);
        if (lane == 63) s_part[wave * kIcpSlots + k] = s;
    }
    __syncthreads();
    if (tid < kIcpSlots) {
        float s = 0.f;
        if (tid < 29) s = ((s_part[tid] + s_part[kIcpSlots + tid]) + s_part[2 * kIcpSlots + tid]) + s_part[3 * kIcpSlots + tid];
        a.partials_out[blockIdx.x * kIcpSlots + tid] = s;
    }
}

int icp_grid_blocks(int W, int H) { return (W * H / 4 + 255) / 256; }

void launch_icp_iteration(const IcpLaunch& l, hipStream_t s) {
    IcpKArgs a;
    a.vc = l.vmap_curr; a.nc = l.nmap_curr; a.vp = l.vmap_prev; a.np = l.nmap_prev;
    a.W = l.W; a.H = l.H; a.k = l.k; a.distThres = l.distThres; a.angleThres = l.angleThres;
    a.partials_in = l.partials_in; a.nb_in = l.nblocks_in; a.partials_out = l.partials_out;
    a.st_in = l.state_in; a.st_out = l.state_out; a.log_out = l.log_out;
    hipLaunchKernelGGL(k_icp_iter, dim3(icp_grid_blocks(l.W, l.H)), dim3(256), 0, s, a);
}

// ------------------------------------------------------------------------------------------------
// begin / finalize
// ------------------------------------------------------------------------------------------------
__global__ void k_icp_begin(const PoseDev* __restrict__ pose, GNState* __restrict__ st) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    GNState s;
    for (int k = 0; k < 16; ++k) s.resultRt[k] = (k % 5 == 0) ? 1.0 : 0.0;
    for (int k = 0; k < 9; ++k) { s.Rprev[k] = pose->R[k]; s.Rcurr[k] = pose->R[k]; s.trR[k] = (k % 4 == 0) ? 1.f : 0.f; }
    for (int k = 0; k < 3; ++k) { s.tprev[k] = pose->t[k]; s.tcurr[k] = pose->t[k]; s.trt[k] = 0.f; }
    m33_inverse_f(s.Rprev, s.Rprev_inv);  // RGBDOdometry.cpp:332
    s.lastICPError = 0.f; s.lastICPCount = 0.f; s.valid = 0; s.pad = 0;
    *st = s;
}

void launch_icp_begin(const PoseDev* pose, GNState* st, hipStream_t s) {
    hipLaunchKernelGGL(k_icp_begin, dim3(1), dim3(64), 0, s, pose, st);
}

// Model::rodrigues2 (Core/Model/Model.cpp:891-932); the SVD re-orthonormalisation U V^T is done by Newton polar
// iterations (identical to rounding for near-rotations).
__device__ void rodrigues2_d(const float* Rin, double* r) {
    double R[9], Rn[9];
    for (int k = 0; k < 9; ++k) R[k] = Rin[k];
    for (int it = 0; it < 4; ++it) {
        const double c00 = R[4] * R[8] - R[5] * R[7], c01 = R[5] * R[6] - R[3] * R[8], c02 = R[3] * R[7] - R[4] * R[6];
        const double det = R[0] * c00 + R[1] * c01 + R[2] * c02;
        const double cof[9] = {c00, c01, c02,
                               R[2] * R[7] - R[1] * R[8], R[0] * R[8] - R[2] * R[6], R[1] * R[6] - R[0] * R[7],
                               R[1] * R[5] - R[2] * R[4], R[2] * R[3] - R[0] * R[5], R[0] * R[4] - R[1] * R[3]};
        for (int k = 0; k < 9; ++k) Rn[k] = 0.5 * (R[k] + cof[k] / det);
        for (int k = 0; k < 9; ++k) R[k] = Rn[k];
    }
    double rx = R[7] - R[5], ry = R[2] - R[6], rz = R[3] - R[1];
    const double s = sqrt((rx * rx + ry * ry + rz * rz) * 0.25);
    double cth = (R[0] + R[4] + R[8] - 1) * 0.5;
    cth = cth > 1. ? 1. : cth < -1. ? -1. : cth;
    double theta = acos(cth);
    if (s < 1e-5) {
        if (cth > 0) rx = ry = rz = 0;
        else {
            double tt = (R[0] + 1) * 0.5;
            rx = sqrt(fmax(tt, 0.0));
            tt = (R[4] + 1) * 0.5;
            ry = sqrt(fmax(tt, 0.0)) * (R[1] < 0 ? -1.0 : 1.0);
            tt = (R[8] + 1) * 0.5;
            rz = sqrt(fmax(tt, 0.0)) * (R[2] < 0 ? -1.0 : 1.0);
            if (fabs(rx) < fabs(ry) && fabs(rx) < fabs(rz) && (R[5] > 0) != (ry * rz > 0)) rz = -rz;
            theta /= sqrt(rx * rx + ry * ry + rz * rz);
            rx *= theta; ry *= theta; rz *= theta;
        }
    } else {
        const double vth = 1 / (2 * s) * theta;
        rx *= vth; ry *= vth; rz *= vth;
    }
    r[0] = rx; r[1] = ry; r[2] = rz;
}

// Derived members of PoseDev from (R,t) and (lastR,lastT): inverse and Model::computeFusionWeight(1.0)
// (Core/Model/Model.cpp:449-464).
__device__ void pose_derive(PoseDev& p) {
    m33_inverse_f(p.R, p.Ri);
    const float3 v = mul33(p.Ri, f3(p.t[0], p.t[1], p.t[2]));
    p.ti[0] = -v.x; p.ti[1] = -v.y; p.ti[2] = -v.z;
    // getLastTransform() = pose^-1 * lastPose
    float Rd[9];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c)
            Rd[r * 3 + c] = p.Ri[r * 3] * p.lastR[c] + p.Ri[r * 3 + 1] * p.lastR[3 + c] + p.Ri[r * 3 + 2] * p.lastR[6 + c];
    float3 td = mul33(p.Ri, f3(p.lastT[0], p.lastT[1], p.lastT[2]));
    td = f3(td.x + p.ti[0], td.y + p.ti[1], td.z + p.ti[2]);
    double rv[3];
    rodrigues2_d(Rd, rv);
    const float rn = (float)sqrt(rv[0] * rv[0] + rv[1] * rv[1] + rv[2] * rv[2]);
    float weighting = fmaxf(norm3(td), rn);
    const float largest = 0.01f, minWeight = 0.5f;
    if (weighting > largest) weighting = largest;
    p.fusionWeight = fmaxf(1.0f - (weighting / largest), minWeight);
}

__global__ __launch_bounds__(256) void k_icp_finalize(const float* __restrict__ partials_in, int nb_in,
                                                       const GNState* __restrict__ st_in, PoseDev* __restrict__ pose,
                                                       PoseDev* __restrict__ host_mirror, float* __restrict__ log_out) {
    __shared__ double s_seg[8 * 32];
    __shared__ double s_sys[32];
    GNState st;
    if (nb_in > 0) reduce_partials(partials_in, nb_in, s_seg, s_sys);
    if (threadIdx.x == 0) {
        if (nb_in > 0) {
            gn_solve_update(s_sys, *st_in, st);
            if (log_out)
                for (int k = 0; k < 32; ++k) log_out[k] = (float)s_sys[k];
        } else st = *st_in;
        PoseDev p = *pose;
        for (int k = 0; k < 9; ++k) { p.lastR[k] = st.Rprev[k]; p.R[k] = st.Rcurr[k]; }
        for (int k = 0; k < 3; ++k) { p.lastT[k] = st.tprev[k]; p.t[k] = st.tcurr[k]; }
        p.lastICPError = st.lastICPError;
        p.lastICPCount = st.lastICPCount;
        pose_derive(p);
        *pose = p;
        if (host_mirror) *host_mirror = p;
    }
}

void launch_icp_finalize(const float* partials_in, int nblocks_in, const GNState* state_in, PoseDev* pose,
                         PoseDev* host_mirror, float* log_out, hipStream_t s) {
    hipLaunchKernelGGL(k_icp_finalize, dim3(1), dim3(256), 0, s, partials_in, nblocks_in, state_in, pose, host_mirror,
                       log_out);
}

// stand-alone icpStep for the parity tests: reduce partials to 32 floats
__global__ __launch_bounds__(256) void k_icp_reduce_only(const float* __restrict__ partials, int nb, float* __restrict__ out32) {
    __shared__ double s_seg[8 * 32];
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
    s.lastICPError = s.lastICPCount = 0.f; s.valid = 0; s.pad = 0;
    *st = s;
}

void launch_icp_step_standalone(const float* Rcurr, const float* tcurr, const float* vc, const float* nc, const float* Rpi,
                                const float* tprev, Intr k, const float* vp, const float* np, float distThres,
                                float angleThres, int W, int H, float* partials, GNState* st, float* out32, hipStream_t s) {
    // partials: scratch of icp_grid_blocks(W,H) * 32 floats; st: two GNState
    hipLaunchKernelGGL(k_state_from_args, dim3(1), dim3(64), 0, s, st, Rcurr, tcurr, Rpi, tprev);
    IcpLaunch l;
    l.vmap_curr = vc; l.nmap_curr = nc; l.vmap_prev = vp; l.nmap_prev = np; l.W = W; l.H = H; l.k = k;
    l.distThres = distThres; l.angleThres = angleThres; l.partials_in = nullptr; l.nblocks_in = 0;
    l.partials_out = partials; l.state_in = st; l.state_out = st + 1; l.log_out = nullptr;
    launch_icp_iteration(l, s);
    hipLaunchKernelGGL(k_icp_reduce_only, dim3(1), dim3(256), 0, s, partials, icp_grid_blocks(W, H), out32);
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

__device__ __forceinline__ void store_tx(float* __restrict__ vm, float* __restrict__ nm, int P, int i, float3 v, bool vok,
                                         float3 n, bool nok, const float* R, float3 t) {
    // tranformMapsKernel, cudafuncs.cu:207-249
    float3 vd = f3(qnan(), qnan(), qnan()), nd = vd;
    if (vok) vd = mul33(R, v) + t;
    if (nok) nd = mul33(R, n);
    vm[i] = vd.x; vm[P + i] = vd.y; vm[2 * P + i] = vd.z;
    nm[i] = nd.x; nm[P + i] = nd.y; nm[2 * P + i] = nd.z;
}

struct PyrArgs {
    const float4* predV; const float4* predN; const float* fillDepth; const FrameDev* frame; const PoseDev* pose;
    float R[9]; float t[3]; int hostPose;
    float* vm[3]; float* nm[3];
    int W, H; Intr k;
};

__global__ __launch_bounds__(256) void k_model_pyramid(const PyrArgs a) {
    const int W2 = a.W >> 2, H2 = a.H >> 2;
    const int x2 = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y2 = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x2 >= W2 || y2 >= H2) return;
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

    float3 v1[4], n1[4]; bool v1ok[4], n1ok[4];
#pragma unroll
    for (int by = 0; by < 2; ++by) {
#pragma unroll
        for (int bx = 0; bx < 2; ++bx) {
            MapPx px[4];
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    const int x = 4 * x2 + 2 * bx + dx, y = 4 * y2 + 2 * by + dy;
                    px[dy * 2 + dx] = load_model_px(a.predV, a.predN, a.fillDepth, useFill, x, y, W, H, a.k);
                    store_tx(a.vm[0], a.nm[0], P0, y * W + x, px[dy * 2 + dx].v, px[dy * 2 + dx].vok, px[dy * 2 + dx].n,
                             px[dy * 2 + dx].nok, R, t);
                }
            // resizeMapKernel<false/true>, cudafuncs.cu:366-417: order x00 + x01 + x10 + x11
            const int b = by * 2 + bx;
            v1ok[b] = px[0].vok && px[1].vok && px[2].vok && px[3].vok;
            n1ok[b] = px[0].nok && px[1].nok && px[2].nok && px[3].nok;
            v1[b] = f3((px[0].v.x + px[1].v.x + px[2].v.x + px[3].v.x) / 4, (px[0].v.y + px[1].v.y + px[2].v.y + px[3].v.y) / 4,
                       (px[0].v.z + px[1].v.z + px[2].v.z + px[3].v.z) / 4);
            n1[b] = normalized_rsqrt(f3((px[0].n.x + px[1].n.x + px[2].n.x + px[3].n.x) / 4,
                                        (px[0].n.y + px[1].n.y + px[2].n.y + px[3].n.y) / 4,
                                        (px[0].n.z + px[1].n.z + px[2].n.z + px[3].n.z) / 4));
            if (!v1ok[b]) v1[b] = f3(qnan(), qnan(), qnan());
            if (!n1ok[b]) n1[b] = f3(qnan(), qnan(), qnan());
            n1ok[b] = n1ok[b] && !isnan(n1[b].x);
            const int x1 = 2 * x2 + bx, y1 = 2 * y2 + by;
            store_tx(a.vm[1], a.nm[1], P1, y1 * W1 + x1, v1[b], v1ok[b], n1[b], n1ok[b], R, t);
        }
    }
    const bool v2ok = v1ok[0] && v1ok[1] && v1ok[2] && v1ok[3];
    bool n2ok = n1ok[0] && n1ok[1] && n1ok[2] && n1ok[3];
    float3 v2 = f3((v1[0].x + v1[1].x + v1[2].x + v1[3].x) / 4, (v1[0].y + v1[1].y + v1[2].y + v1[3].y) / 4,
                   (v1[0].z + v1[1].z + v1[2].z + v1[3].z) / 4);
    float3 n2 = normalized_rsqrt(f3((n1[0].x + n1[1].x + n1[2].x + n1[3].x) / 4, (n1[0].y + n1[1].y + n1[2].y + n1[3].y) / 4,
                                    (n1[0].z + n1[1].z + n1[2].z + n1[3].z) / 4));
    if (!v2ok) v2 = f3(qnan(), qnan(), qnan());
    if (!n2ok) n2 = f3(qnan(), qnan(), qnan());
    n2ok = n2ok && !isnan(n2.x);
    store_tx(a.vm[2], a.nm[2], P2, y2 * W2 + x2, v2, v2ok, n2, n2ok, R, t);
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
    dim3 grid(((W >> 2) + 63) / 64, ((H >> 2) + 3) / 4);
    hipLaunchKernelGGL(k_model_pyramid, grid, dim3(256), 0, s, a);
}

}  // namespace mf

// mf_device.h -- device-side helpers (gfx950, wave64).
#pragma once

#include <hip/hip_runtime.h>
#include "mf_internal.h"

namespace mf {

__device__ __forceinline__ float qnan() { return __int_as_float(0x7fffffff); }

__device__ __forceinline__ float3 f3(float x, float y, float z) { return make_float3(x, y, z); }
__device__ __forceinline__ float3 operator-(float3 a, float3 b) { return f3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ float3 operator+(float3 a, float3 b) { return f3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ float3 operator*(float3 a, float s) { return f3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ float3 cross3(float3 a, float3 b) {
    return f3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
// every product rounded on its own: cross(v, v) is then EXACTLY zero (with fused multiply-adds it is a rounding residual pointing
// anywhere), which is what turns a degenerate depth neighbourhood into a NaN normal -- the "invalid" mark of every map here
__device__ __forceinline__ float3 cross3_exact(float3 a, float3 b) {
#pragma clang fp contract(off)
    return f3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
__device__ __forceinline__ float dot3(float3 a, float3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ float norm3(float3 a) { return sqrtf(dot3(a, a)); }
// CUDA-side normalized(): a * rsqrt(dot) (Core/Cuda/operators.cuh:80-84).  The reference's rsqrtf is an approximate instruction (2 ulp)
// no CPU build reproduces, and v_rsq_f32 is a different approximation (1 ulp); the CPU restatement evaluates 1 / sqrt with two correctly
// rounded operations, and so does the device since round 3: the normal maps -- and with them the geometric edge map, whose 0.3 threshold
// decides label pixels -- are then bit-identical to the restatement's on the same depth (the 8-object scene showed ~0.3 % of the label
// pixels flipping on 1-ulp normals).  ~20 instructions per normal in three once-per-frame kernels.
__device__ __forceinline__ float3 normalized_rsqrt(float3 a) { return a * (1.0f / sqrtf(dot3(a, a))); }
// GLSL normalize(): a / length(a)
__device__ __forceinline__ float3 normalize_gl(float3 a) {
    const float l = sqrtf(dot3(a, a));
    return f3(a.x / l, a.y / l, a.z / l);
}
__device__ __forceinline__ float3 mul33(const float* R, float3 a) {  // row-major
    return f3(R[0] * a.x + R[1] * a.y + R[2] * a.z, R[3] * a.x + R[4] * a.y + R[5] * a.z,
              R[6] * a.x + R[7] * a.y + R[8] * a.z);
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// 64-lane sum; result valid in lane 0 (and, with the xor form, in every lane)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__device__ __forceinline__ int wave_min_i(int v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = min(v, __shfl_xor(v, off, 64));
    return v;
}
__device__ __forceinline__ int wave_max_i(int v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = max(v, __shfl_xor(v, off, 64));
    return v;
}
// order-preserving map float -> int (and back): integer atomicMin / atomicMax then order floats, negative ones included
__device__ __forceinline__ int box_enc(float f) { const int b = __float_as_int(f); return b >= 0 ? b : b ^ 0x7FFFFFFF; }
__device__ __forceinline__ float box_dec(int e) { return __int_as_float(e >= 0 ? e : e ^ 0x7FFFFFFF); }

// rank of this lane among the set lanes of a ballot (wave64)
__device__ __forceinline__ int lane_rank(unsigned long long mask) {
    const int lo = __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0);
    return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), lo);
}

// z-buffer update: one 64-bit atomicMin.  (Measured: pre-testing with a relaxed read to skip losing fragments made the
// scatter passes 5-15 % SLOWER -- the atomics are not what bounds them.)
__device__ __forceinline__ void zmin_key(unsigned long long* addr, unsigned long long key) {
    atomicMin(addr, key);
}
// The same behind a plain read: a fragment that cannot win is dropped without an atomic.  The read may be stale (device-scope atomics are
// performed memory-side, past the L2 the read hits) -- a stale value is an OLDER, larger key, so a fragment is only ever dropped when it
// loses against the current one too: same keys, bit for bit.  For the passes of OBJECT models, whose surfels pile up on a few thousand
// pixels (the 3.4 M-surfel object maps of configs[4]: ~85 fragments per pixel, nearly all of them losers); on a background map at one or two
// fragments per pixel the extra read costs more than it saves (measured in round 2: +5-15 %).
__device__ __forceinline__ void zmin_key_pretested(unsigned long long* addr, unsigned long long key) {
    if (key < __hip_atomic_load(addr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(addr, key);   // (agent scope: past the per-CU L1, which would keep serving the launch's first value)
}

// ---- exp() and acos() of the surfel shaders (surfels.glsl:44, data.vert:167): GLSL leaves their last bits to the GPU vendor.
// This library and the CPU checker the parity tests compare it with evaluate them with the SAME
// sequence of individually rounded fp32 operations (Cephes-style range reduction + polynomial, <= 2 ulp), so that a confidence
// or an angle test can never differ between them in the last bit -- which the life cycle of a surfel would amplify into a
// different keep / merge decision a few frames later. ----
__device__ __forceinline__ float shader_exp(float x) {
#pragma clang fp contract(off)
    // x <= 0 here (the argument is -(r/400)^2 / 0.72); valid for |x| < 87 
    const float n = rintf(x * 1.44269504088896341f);
    float r = x - n * 0.693359375f;
    r = r - n * -2.12194440e-4f;
    float p = 1.9875691500E-4f;
    p = p * r + 1.3981999507E-3f;
    p = p * r + 8.3334519073E-3f;
    p = p * r + 4.1665795894E-2f;
    p = p * r + 1.6666665459E-1f;
    p = p * r + 5.0000001201E-1f;
    const float y = p * (r * r) + r + 1.0f;
    return y * __uint_as_float((unsigned)((int)n + 127) << 23);   // exact scaling by 2^n 
}
__device__ __forceinline__ float shader_asin_core(float a) {
#pragma clang fp contract(off)
    const float z = a * a;
    float p = 4.2163199048E-2f;
    p = p * z + 2.4181311049E-2f;
    p = p * z + 4.5470025998E-2f;
    p = p * z + 7.4953002686E-2f;
    p = p * z + 1.6666752422E-1f;
    return p * z * a + a;
}
__device__ __forceinline__ float shader_acos(float x) {
#pragma clang fp contract(off)
    if (x > 0.5f) return 2.0f * shader_asin_core(sqrtf(0.5f * (1.0f - x)));
    if (x < -0.5f) return 3.14159265358979323846f - 2.0f * shader_asin_core(sqrtf(0.5f * (1.0f + x)));
    return 1.57079632679489661923f - shader_asin_core(x);   // NaN falls through to here and stays NaN 
}

// ---- surfel / shader helpers (Core/Shaders/surfels.glsl, color_encoding.glsl, geometry.glsl) ----
__device__ __forceinline__ float surfel_radius(float depth, float norm_z, Intr k) {  // surfels.glsl:19-34
    const float camz = 1.0f / k.fx, camw = 1.0f / k.fy;
    const float meanFocal = ((1.0f / fabsf(camz)) + (1.0f / fabsf(camw))) / 2.0f;
    const float radius = (depth / meanFocal) * 1.41421356237f;
    return fminf(2.0f * radius, radius / fabsf(norm_z));
}
__device__ __forceinline__ float surfel_confidence(float x, float y, float weighting, Intr k) {  // surfels.glsl:36-46
    const float dx = x - k.cx, dy = y - k.cy;
    const float radialDist = sqrtf(dx * dx + dy * dy) / 400.f;
    return shader_exp(-(radialDist * radialDist) / 0.72f) * weighting;
}
__device__ __forceinline__ float encode_color(float r, float g, float b) {  // color_encoding.glsl:19-25
    int rgb = (int)roundf(r * 255.0f);
    rgb = (rgb << 8) + (int)roundf(g * 255.0f);
    rgb = (rgb << 8) + (int)roundf(b * 255.0f);
    return (float)rgb;
}
__device__ __forceinline__ float3 decode_color(float c) {  // color_encoding.glsl:27-34
    const int ci = (int)c;
    return f3((float)((ci >> 16) & 0xFF) / 255.0f, (float)((ci >> 8) & 0xFF) / 255.0f, (float)(ci & 0xFF) / 255.0f);
}
// nearest fetch, GL_CLAMP_TO_EDGE
__device__ __forceinline__ float texf(const float* img, int W, int H, int x, int y) {
    return img[clampi(y, 0, H - 1) * W + clampi(x, 0, W - 1)];
}
// geometry.glsl:21-26; cam = (cx, cy, 1/fx, 1/fy)
// (products rounded before anything is subtracted from them: the normals below difference two vertices a pixel apart -- ~1e-3 of
// their magnitude -- so a product fused into that subtraction shows up as 1e-4 in the normal)
__device__ __forceinline__ float3 get_vertex(const float* depth, int W, int H, int px, int py, float x, float y, Intr k) {
#pragma clang fp contract(off)
    const float z = texf(depth, W, H, px, py);
    return f3((x - k.cx) * z * (1.0f / k.fx), (y - k.cy) * z * (1.0f / k.fy), z);
}
// geometry.glsl:28-40 (central differences)
__device__ __forceinline__ float3 get_normal_central(const float* depth, int W, int H, int px, int py, float x, float y,
                                                     float3 vPos, Intr k) {
    const float3 xf = get_vertex(depth, W, H, px + 1, py, x + 1, y, k);
    const float3 xb = get_vertex(depth, W, H, px - 1, py, x - 1, y, k);
    const float3 yf = get_vertex(depth, W, H, px, py + 1, x, y + 1, k);
    const float3 yb = get_vertex(depth, W, H, px, py - 1, x, y - 1, k);
    const float3 del_x = ((xb + vPos) * 0.5f) - ((xf + vPos) * 0.5f);
    const float3 del_y = ((yb + vPos) * 0.5f) - ((yf + vPos) * 0.5f);
    return normalize_gl(cross3_exact(del_x, del_y));
}
// geometry.glsl:42-62 (forward differences, integer pixel coordinates)
__device__ __forceinline__ float3 get_normal_forward(const float* depth, int W, int H, int px, int py, float3 vPos, Intr k) {
    const float3 vx = get_vertex(depth, W, H, px + 1, py, (float)(px + 1), (float)py, k);
    const float3 vy = get_vertex(depth, W, H, px, py + 1, (float)px, (float)(py + 1), k);
    return normalize_gl(cross3_exact(vx - vPos, vy - vPos));
}

// ---- pose math shared by the odometry and the object-model kernels ----
// fp64 reciprocal / square root from the hardware seeds (v_rcp_f64 / v_rsq_f64, ~2^-26) plus Newton steps: ~8 instructions
// instead of the ~40 of an IEEE division; last-ulp differences are irrelevant for a Gauss-Newton step.
// Image kernels: workgroup b of a launch runs on XCD b % 8 (observed placement: used for speed, never for correctness), and every XCD has
// its own 4 MiB L2.  A row-major tile order therefore deals neighbouring tiles -- which share their halo rows -- to eight different L2s,
// and each L2 ends up fetching the whole image from HBM (k_bilateral: 9.5 MB fetched for a 1.2 MB image).  This maps the launch's linear
// workgroup id to a tile index such that XCD k owns the k-th CONTIGUOUS eighth of the tile list; launch 8 * ceil(tiles / 8) workgroups
// and skip indices >= tiles.
__device__ __forceinline__ int xcd_contiguous_tile(int linear_wg, int tiles) {
    const int per = (tiles + 7) >> 3;
    return (linear_wg & 7) * per + (linear_wg >> 3);
}
__host__ __device__ inline int xcd_padded_grid(int tiles) { return ((tiles + 7) >> 3) << 3; }

__device__ __forceinline__ double rcp_d(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = r * (2.0 - x * r);
    r = r * (2.0 - x * r);
    return r;
}
__device__ __forceinline__ double sqrt_d(double x) {  // x > 0
    double r = __builtin_amdgcn_rsq(x);
    r = r * (1.5 - 0.5 * x * r * r);
    r = r * (1.5 - 0.5 * x * r * r);
    double t = x * r;
    return t + 0.5 * r * (x - t * t);
}

__device__ inline void m33_inverse_f(const float* m, float* inv) {  // cofactor inverse (Eigen fixed-size stand-in)
#pragma clang fp contract(off)   // every operation rounded on its own, as the CPU restatement computes it: the inverse pose must not depend on the
                                 // kernel it was derived in (round 4: see pose_derive)
    const float c00 = m[4] * m[8] - m[5] * m[7];
    const float c01 = m[5] * m[6] - m[3] * m[8];
    const float c02 = m[3] * m[7] - m[4] * m[6];
    const float det = m[0] * c00 + m[1] * c01 + m[2] * c02;
    const float id = 1.0f / det;
    inv[0] = c00 * id; inv[1] = (m[2] * m[7] - m[1] * m[8]) * id; inv[2] = (m[1] * m[5] - m[2] * m[4]) * id;
    inv[3] = c01 * id; inv[4] = (m[0] * m[8] - m[2] * m[6]) * id; inv[5] = (m[2] * m[3] - m[0] * m[5]) * id;
    inv[6] = c02 * id; inv[7] = (m[1] * m[6] - m[0] * m[7]) * id; inv[8] = (m[0] * m[4] - m[1] * m[3]) * id;
}


// Model::rodrigues2 (Core/Model/Model.cpp:891-932); the SVD re-orthonormalisation U V^T is done by Newton polar
// iterations (identical to rounding for near-rotations).
__device__ inline void rodrigues2_d(const float* Rin, double* r) {
#pragma clang fp contract(off)
    double R[9], Rn[9];
    for (int k = 0; k < 9; ++k) R[k] = Rin[k];
    for (int it = 0; it < 4; ++it) {
        const double c00 = R[4] * R[8] - R[5] * R[7], c01 = R[5] * R[6] - R[3] * R[8], c02 = R[3] * R[7] - R[4] * R[6];
        const double det = R[0] * c00 + R[1] * c01 + R[2] * c02;
        const double cof[9] = {c00, c01, c02,
                               R[2] * R[7] - R[1] * R[8], R[0] * R[8] - R[2] * R[6], R[1] * R[6] - R[0] * R[7],
                               R[1] * R[5] - R[2] * R[4], R[2] * R[3] - R[0] * R[5], R[0] * R[4] - R[1] * R[3]};
        const double idet = rcp_d(det);
        for (int k = 0; k < 9; ++k) Rn[k] = 0.5 * (R[k] + cof[k] * idet);
        for (int k = 0; k < 9; ++k) R[k] = Rn[k];
    }
    double rx = R[7] - R[5], ry = R[2] - R[6], rz = R[3] - R[1];
    const double s2 = (rx * rx + ry * ry + rz * rz) * 0.25;
    const double s = s2 > 0.0 ? sqrt_d(s2) : 0.0;
    double cth = (R[0] + R[4] + R[8] - 1) * 0.5;
    cth = cth > 1. ? 1. : cth < -1. ? -1. : cth;
    // asin series for the usual small inter-frame rotation (error < 1e-15 below 0.1 rad); libm acos otherwise
    double theta;
    if (s < 0.1 && cth > 0) {
        const double q = s * s;
        theta = s * (1.0 + q * (1.0 / 6 + q * (3.0 / 40 + q * (15.0 / 336 + q * (105.0 / 3456 + q * (945.0 / 42240 + q * (10395.0 / 599040)))))));
    } else theta = acos(cth);
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
        const double vth = rcp_d(2 * s) * theta;
        rx *= vth; ry *= vth; rz *= vth;
    }
    r[0] = rx; r[1] = ry; r[2] = rz;
}

// Model::rodrigues2 as the reference's text evaluates it (finding F5): U V^T, its off-diagonal differences and its TRACE are float there
// (Eigen::Matrix3f), so cos(theta) = (trace - 1) / 2 is quantised in steps of 1.2e-7 and theta = acos(c) in steps of ~4.9e-4 rad near 0:
// a rotation below that reads as zero, one above it as a multiple of the step.  The polar factor is computed in double (IEEE divisions,
// no contraction: the same bits as the CPU restatement) and rounded to float where the reference's SVD product is float.
__device__ inline void rodrigues2_literal_d(const float* Rin, double* r) {
#pragma clang fp contract(off)
#pragma clang fp contract(off)
    double R[9], Rn[9];
    for (int k = 0; k < 9; ++k) R[k] = Rin[k];
    for (int it = 0; it < 4; ++it) {
        const double c00 = R[4] * R[8] - R[5] * R[7], c01 = R[5] * R[6] - R[3] * R[8], c02 = R[3] * R[7] - R[4] * R[6];
        const double det = R[0] * c00 + R[1] * c01 + R[2] * c02;
        const double cof[9] = {c00, c01, c02,
                               R[2] * R[7] - R[1] * R[8], R[0] * R[8] - R[2] * R[6], R[1] * R[6] - R[0] * R[7],
                               R[1] * R[5] - R[2] * R[4], R[2] * R[3] - R[0] * R[5], R[0] * R[4] - R[1] * R[3]};
        const double idet = 1.0 / det;   // ONE IEEE division per iteration (nine cost ~3 us of the single-thread finalize chain); the CPU restatement evaluates the same expression
        for (int k = 0; k < 9; ++k) Rn[k] = 0.5 * (R[k] + cof[k] * idet);
        for (int k = 0; k < 9; ++k) R[k] = Rn[k];
    }
    float Rf[9];
    for (int k = 0; k < 9; ++k) Rf[k] = (float)R[k];
    double rx = (double)(Rf[7] - Rf[5]), ry = (double)(Rf[2] - Rf[6]), rz = (double)(Rf[3] - Rf[1]);
    const double s = sqrt((rx * rx + ry * ry + rz * rz) * 0.25);
    const float tr = (Rf[0] + Rf[4]) + Rf[8];
    double cth = (double)(tr - 1.0f) * 0.5;
    cth = cth > 1. ? 1. : cth < -1. ? -1. : cth;
    double theta = acos(cth);
    if (s < 1e-5) {
        if (cth > 0) rx = ry = rz = 0;
        else {
            double tt = ((double)Rf[0] + 1) * 0.5;
            rx = sqrt(fmax(tt, 0.0));
            tt = ((double)Rf[4] + 1) * 0.5;
            ry = sqrt(fmax(tt, 0.0)) * (Rf[1] < 0 ? -1.0 : 1.0);
            tt = ((double)Rf[8] + 1) * 0.5;
            rz = sqrt(fmax(tt, 0.0)) * (Rf[2] < 0 ? -1.0 : 1.0);
            if (fabs(rx) < fabs(ry) && fabs(rx) < fabs(rz) && (Rf[5] > 0) != (ry * rz > 0)) rz = -rz;
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
__device__ inline void pose_derive(PoseDev& p) {
    // Contraction off, operands in the CPU restatement's order.  This function is inlined into kernels of several files; mf_odometry.hip (the
    // Gauss-Newton finalize kernels) is compiled with FMA contraction, the surfel / pose-override kernels without.  Left to the compiler, the
    // inverse pose of a TRACKED frame came out one ulp away from the restatement's (an overridden pose did not), surfels on a pixel border
    // projected into the neighbouring texel and two of 76 800 association decisions of a teacher-forced frame differed on the MI355X while the
    // CPU-executed kernels agreed everywhere (round 4, tests/test_gpu_parity_long.py::test_s2_eight_objects_tracked_teacher_forced).
#pragma clang fp contract(off)
    m33_inverse_f(p.R, p.Ri);
    p.ti[0] = -(p.Ri[0] * p.t[0] + p.Ri[1] * p.t[1] + p.Ri[2] * p.t[2]);
    p.ti[1] = -(p.Ri[3] * p.t[0] + p.Ri[4] * p.t[1] + p.Ri[5] * p.t[2]);
    p.ti[2] = -(p.Ri[6] * p.t[0] + p.Ri[7] * p.t[1] + p.Ri[8] * p.t[2]);
    // getLastTransform() = pose^-1 * lastPose
    float Rd[9];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c)
            Rd[r * 3 + c] = p.Ri[r * 3] * p.lastR[c] + p.Ri[r * 3 + 1] * p.lastR[3 + c] + p.Ri[r * 3 + 2] * p.lastR[6 + c];
    float3 td = f3(p.Ri[0] * p.lastT[0] + p.Ri[1] * p.lastT[1] + p.Ri[2] * p.lastT[2], p.Ri[3] * p.lastT[0] + p.Ri[4] * p.lastT[1] + p.Ri[5] * p.lastT[2],
                   p.Ri[6] * p.lastT[0] + p.Ri[7] * p.lastT[1] + p.Ri[8] * p.lastT[2]);
    td = f3(td.x + p.ti[0], td.y + p.ti[1], td.z + p.ti[2]);
    double rv[3];
    if (p.weightLiteral) rodrigues2_literal_d(Rd, rv);
    else rodrigues2_d(Rd, rv);
    const float rn = (float)sqrt(rv[0] * rv[0] + rv[1] * rv[1] + rv[2] * rv[2]);
    const float tn = sqrtf(td.x * td.x + td.y * td.y + td.z * td.z);
    float weighting = fmaxf(tn, rn);
    const float largest = 0.01f, minWeight = 0.5f;
    if (weighting > largest) weighting = largest;
    p.fusionWeight = fmaxf(1.0f - (weighting / largest), minWeight);
}


// Pose log entry (MaskFusion.cpp:580-596): cam->world for the background, obj->world = globalPose * pose^-1 for objects;
// translation + Eigen::Quaternionf(rotation) as x y z w.  Kept on the device so that logging costs no host visit.
__device__ inline void quat_from_rot(const float* m /*row-major*/, float* q /*x y z w*/) {  // Eigen/src/Geometry/Quaternion.h
    float t = m[0] + m[4] + m[8];
    if (t > 0.f) {
        t = sqrtf(t + 1.0f);
        q[3] = 0.5f * t;
        t = 0.5f / t;
        q[0] = (m[7] - m[5]) * t; q[1] = (m[2] - m[6]) * t; q[2] = (m[3] - m[1]) * t;
    } else {
        // the three cases written out with constant indices (i = largest diagonal entry, j = i + 1, k = j + 1 mod 3): an index computed at run
        // time puts m[] and q[] into scratch memory -- 48 bytes per lane that every launch of every kernel containing this (the tiled
        // prediction's epilogue among them) then has to set up
        int i = 0;
        if (m[4] > m[0]) i = 1;
        if (m[8] > (i == 0 ? m[0] : m[4])) i = 2;
        if (i == 0) {        // j = 1, k = 2
            t = sqrtf(m[0] - m[4] - m[8] + 1.0f);
            q[0] = 0.5f * t; t = 0.5f / t;
            q[3] = (m[7] - m[5]) * t; q[1] = (m[3] + m[1]) * t; q[2] = (m[6] + m[2]) * t;
        } else if (i == 1) { // j = 2, k = 0
            t = sqrtf(m[4] - m[8] - m[0] + 1.0f);
            q[1] = 0.5f * t; t = 0.5f / t;
            q[3] = (m[2] - m[6]) * t; q[2] = (m[7] + m[5]) * t; q[0] = (m[1] + m[3]) * t;
        } else {             // j = 0, k = 1
            t = sqrtf(m[8] - m[0] - m[4] + 1.0f);
            q[2] = 0.5f * t; t = 0.5f / t;
            q[3] = (m[3] - m[1]) * t; q[0] = (m[2] + m[6]) * t; q[1] = (m[5] + m[7]) * t;
        }
    }
}
__device__ inline void pose_log_entry(const PoseDev* pose, const PoseDev* bg /*nullptr for the background itself*/, float* o /*[8]*/) {
    float R[9], t[3];
    if (!bg) {
        for (int k = 0; k < 9; ++k) R[k] = pose->R[k];
        for (int k = 0; k < 3; ++k) t[k] = pose->t[k];
    } else {
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) R[r * 3 + c] = bg->R[r * 3] * pose->Ri[c] + bg->R[r * 3 + 1] * pose->Ri[3 + c] + bg->R[r * 3 + 2] * pose->Ri[6 + c];
        const float3 v = mul33(bg->R, f3(pose->ti[0], pose->ti[1], pose->ti[2]));
        t[0] = v.x + bg->t[0]; t[1] = v.y + bg->t[1]; t[2] = v.z + bg->t[2];
    }
    o[0] = t[0]; o[1] = t[1]; o[2] = t[2];
    quat_from_rot(R, o + 3);
    o[7] = 0.f;
}

// sin / cos for |th| <= 0.5 by Taylor series (error < 1e-17): keeps libm's large-argument reduction (and its scratch
// arrays) out of the per-iteration kernel.
__device__ __forceinline__ void sincos_small(double th, double& s, double& c) {
    const double t2 = th * th;  // Horner form with reciprocal-factorial coefficients: no fp64 divisions (~40 instructions each)
    s = th * (1.0 + t2 * (-1.0 / 6 + t2 * (1.0 / 120 + t2 * (-1.0 / 5040 + t2 * (1.0 / 362880 + t2 * (-1.0 / 39916800 +
        t2 * (1.0 / 6227020800.0 + t2 * (-1.0 / 1307674368000.0))))))));
    c = 1.0 + t2 * (-0.5 + t2 * (1.0 / 24 + t2 * (-1.0 / 720 + t2 * (1.0 / 40320 + t2 * (-1.0 / 3628800 + t2 * (1.0 / 479001600.0 +
        t2 * (-1.0 / 87178291200.0 + t2 * (1.0 / 20922789888000.0))))))));
}

// OdometryProvider::rodrigues (Core/Utils/OdometryProvider.h:32-67).  Rotations beyond 0.5 rad (never produced by a
// converging Gauss-Newton step) are built by repeated squaring of the rotation by theta / 2^k.
__device__ __forceinline__ void rodrigues_d(double wx, double wy, double wz, double (&R)[3][3]) {
    const double xx = wx * wx, yy = wy * wy, zz = wz * wz;
    const double th2 = xx + yy + zz;
    if (th2 <= 0.25) {
        // The case every converging Gauss-Newton step is in.  R = I + A [w]x + B [w]x^2 with A = sin(th)/th and B = (1 - cos th)/th^2,
        // both even power series in th -- so no square root, no reciprocal and no sin / cos of th itself: this function sits on the
        // one-thread dependency chain every iteration launch waits for (solve -> pose), where each of those cost a dozen dependent
        // fp64 operations.  Nine terms each (truncation < 1e-19 at th^2 = 0.25), evaluated in Estrin form (depth 5 instead of 9).
        const double u = th2, u2 = u * u, u4 = u2 * u2;
        const double A = ((1.0 + u * (-1.0 / 6)) + u2 * (1.0 / 120 + u * (-1.0 / 5040))) +
                         u4 * (((1.0 / 362880 + u * (-1.0 / 39916800)) + u2 * (1.0 / 6227020800.0 + u * (-1.0 / 1307674368000.0))) +
                               u4 * (1.0 / 355687428096000.0));
        const double B = ((0.5 + u * (-1.0 / 24)) + u2 * (1.0 / 720 + u * (-1.0 / 40320))) +
                         u4 * (((1.0 / 3628800 + u * (-1.0 / 479001600.0)) + u2 * (1.0 / 87178291200.0 + u * (-1.0 / 20922789888000.0))) +
                               u4 * (1.0 / 6402373705728000.0));
        const double xy = B * (wx * wy), xz = B * (wx * wz), yz = B * (wy * wz);
        const double ax = A * wx, ay = A * wy, az = A * wz;
        R[0][0] = 1.0 - B * (yy + zz); R[0][1] = xy - az;             R[0][2] = xz + ay;
        R[1][0] = xy + az;             R[1][1] = 1.0 - B * (xx + zz); R[1][2] = yz - ax;
        R[2][0] = xz - ay;             R[2][1] = yz + ax;             R[2][2] = 1.0 - B * (xx + yy);
        return;
    }
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) R[r][c] = (r == c) ? 1.0 : 0.0;
    const double theta = th2 > 0.0 ? sqrt_d(th2) : 0.0;
    if (theta >= 2.2204460492503131e-16) {
        int halvings = 0;
        double th = theta;
        while (th > 0.5 && halvings < 64) { th *= 0.5; ++halvings; }
        double s, c;
        sincos_small(th, s, c);
        const double c1 = 1. - c, itheta = rcp_d(theta);
        const double rx = wx * itheta, ry = wy * itheta, rz = wz * itheta;
        R[0][0] = c + c1 * rx * rx;      R[0][1] = c1 * rx * ry - s * rz; R[0][2] = c1 * rx * rz + s * ry;
        R[1][0] = c1 * rx * ry + s * rz; R[1][1] = c + c1 * ry * ry;      R[1][2] = c1 * ry * rz - s * rx;
        R[2][0] = c1 * rx * rz - s * ry; R[2][1] = c1 * ry * rz + s * rx; R[2][2] = c + c1 * rz * rz;
        for (int h = 0; h < halvings; ++h) {
            double Q[3][3];
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int cc = 0; cc < 3; ++cc) Q[r][cc] = R[r][0] * R[0][cc] + R[r][1] * R[1][cc] + R[r][2] * R[2][cc];
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int cc = 0; cc < 3; ++cc) R[r][cc] = Q[r][cc];
        }
    }
}


}  // namespace mf

// mf_rgbd_device.h -- per-pixel device functions of the photometric term and the SO(3) pre-alignment.
//
// Replaces (reference, relative to /root/reference):
//   RGBResidual::getProducts   Core/Cuda/reduce.cu:812-895     -> rgb_residual_px
//   RGBReduction::getProducts  Core/Cuda/reduce.cu:547-626     -> rgb_step_px
//   SO3Reduction::getProducts  Core/Cuda/reduce.cu:1033-1124   -> so3_px
//   bgr2IntensityKernel / applyKernel / projectPointsKernel    Core/Cuda/cudafuncs.cu:626-639,658-683,722-738
//
// Every function that feeds an integer truncation / rounding is compiled with contraction off and spells its fused
// multiply-adds out (the nvcc-style form), so that the integer results are bit-identical to the oracle's.
#pragma once

#include "mf_internal.h"
#include "mf_device.h"

namespace mf {

__device__ __forceinline__ uint8_t intensity_of(float c0, float c1, float c2) {
#pragma clang fp contract(off)
    const float v = fmaf(c2, 0.587f, fmaf(c1, 0.299f, c0 * 0.114f));
    return (uint8_t)(int)v;
}

// The tests of RGBResidual::getProducts that do not depend on the pose (reduce.cu:823-851): border, 4x4 window of the next
// image all > 0, squared gradient magnitude >= minScale.
__device__ __forceinline__ bool rgb_gate_px(const uint8_t* __restrict__ nextImage, int valx, int valy, float minScale, int W, int H,
                                            int x, int y) {
    if (!(x < W - 5 && y < H - 1)) return false;
    bool valid = true;
    for (int u = max(y - 2, 0); u < min(y + 2, H); ++u)
        for (int v = max(x - 2, 0); v < min(x + 2, W); ++v) valid = valid && (nextImage[u * W + v] > 0);
    if (!valid) return false;
    const float mTwo = (float)((valx * valx) + (valy * valy));
    return mTwo >= minScale;
}

// Returns true (and fills c) iff pixel (x, y) of the next image finds a photometric correspondence in the last image.
__device__ __forceinline__ bool rgb_residual_px(const RgbLevel& L, const float* __restrict__ krk, float3 kt, int x, int y,
                                                RgbCorr& c) {
#pragma clang fp contract(off)
    c.u0 = -1; c.v0 = -1; c.diff = 0.f;
    const int W = L.W, H = L.H;
    if (L.gate) {
        if (!L.gate[y * W + x]) return false;   // the same three tests, evaluated once per frame (rgb_gate_px)
    } else if (!rgb_gate_px(L.nextImage, L.dIdx[y * W + x], L.dIdy[y * W + x], L.minScale, W, H, x, y)) return false;
    const float d1 = L.nextDepth[y * W + x];
    if (isnan(d1)) return false;
    const float fx_ = (float)x, fy_ = (float)y;
    const float l2 = fmaf(krk[7], fy_, krk[6] * fx_) + krk[8];
    const float l0 = fmaf(krk[1], fy_, krk[0] * fx_) + krk[2];
    const float l1 = fmaf(krk[4], fy_, krk[3] * fx_) + krk[5];
    const float td1 = fmaf(d1, l2, kt.z);
    const int u0 = __float2int_rn(fmaf(d1, l0, kt.x) / td1);
    const int v0 = __float2int_rn(fmaf(d1, l1, kt.y) / td1);
    if (!(u0 >= 0 && v0 >= 0 && u0 < W && v0 < H)) return false;
    const float d0 = L.lastDepth[v0 * W + u0];
    const int li = L.lastImage[v0 * W + u0];
    if (!(d0 > 0 && fabsf(td1 - d0) <= L.maxDepthDelta && li != 0)) return false;
    c.u0 = (int16_t)u0; c.v0 = (int16_t)v0;
    c.diff = (float)L.nextImage[y * W + x] - (float)li;
    return true;
}

// 29 accumulators in acc[0..28] (27 products of the 7-vector row, row[6]^2, inliers), reduce.cu:591-626 order.
__device__ __forceinline__ void rgb_step_px(const RgbLevel& L, const RgbCorr& c, int x, int y, float sigma, Intr k,
                                            float sobelScale, float* acc) {
#pragma clang fp contract(off)
    float w = sigma + fabsf(c.diff);
    w = w > 1.1920929e-07f ? 1.0f / w : 1.0f;
    if (sigma == -1.f) w = 1.f;
    float row[7];
    row[6] = -w * c.diff;
    // projectPointsKernel on the fly (cudafuncs.cu:722-738): the cloud of the LAST depth at (u0, v0)
    const float z = L.lastDepth[c.v0 * L.W + c.u0];
    const float X = ((float)c.u0 - k.cx) * z * (1.0f / k.fx);
    const float Y = ((float)c.v0 - k.cy) * z * (1.0f / k.fy);
    const float invz = 1.0f / z;
    const float dI_dx_val = w * sobelScale * (float)L.dIdx[y * L.W + x];
    const float dI_dy_val = w * sobelScale * (float)L.dIdy[y * L.W + x];
    const float v0 = dI_dx_val * k.fx * invz;
    const float v1 = dI_dy_val * k.fy * invz;
    const float v2 = -(v0 * X + v1 * Y) * invz;
    row[0] = v0; row[1] = v1; row[2] = v2;
    row[3] = -z * v1 + Y * v2;
    row[4] = z * v0 - X * v2;
    row[5] = -Y * v0 + X * v1;
    int q = 0;
#pragma unroll
    for (int r = 0; r < 7; ++r)
#pragma unroll
        for (int cc = r; cc < 7; ++cc) acc[q++] += row[r] * row[cc];
    acc[28] += 1.0f;
}

__device__ __forceinline__ float2 so3_gradient(const uint8_t* __restrict__ img, int W, int x, int y) {  // reduce.cu:1015-1031
    const float actu = (float)img[y * W + x];
    float back = (float)img[y * W + x - 1], fore = (float)img[y * W + x + 1];
    float2 g;
    g.x = ((back + actu) / 2.0f) - ((fore + actu) / 2.0f);
    back = (float)img[(y - 1) * W + x]; fore = (float)img[(y + 1) * W + x];
    g.y = ((back + actu) / 2.0f) - ((fore + actu) / 2.0f);
    return g;
}

// acc[0..8]: products of the 4-vector row (aa ab ac ad bb bc bd cc cd), acc[9]: residual, acc[10]: inliers
__device__ __forceinline__ void so3_px(const uint8_t* __restrict__ lastImage, const uint8_t* __restrict__ nextImage, int W, int H,
                                       const float* __restrict__ basis /*imageBasis[9] kinv[9] krlr[9]*/, int x, int y, float* acc) {
#pragma clang fp contract(off)
    const float3 p = f3((float)x, (float)y, 1.0f);
    const float3 wp = mul33(basis, p);
    const int wx = __float2int_rn(wp.x / wp.z), wy = __float2int_rn(wp.y / wp.z);
    if (!(wx >= 1 && wx < W - 1 && wy >= 1 && wy < H - 1 && x >= 1 && x < W - 1 && y >= 1 && y < H - 1)) return;
    const float2 gn = so3_gradient(nextImage, W, wx, wy);
    const float2 gl = so3_gradient(lastImage, W, x, y);
    const float gx = (gn.x + gl.x) / 2.0f, gy = (gn.y + gl.y) / 2.0f;
    const float3 point = mul33(basis + 9, p);
    const float z2 = point.z * point.z;
    const float* m = basis + 18;
    const float a = m[0], b = m[1], c = m[2], d = m[3], e = m[4], f = m[5], g = m[6], h = m[7], i = m[8];
    const float fy_ = (float)y, fx_ = (float)x;
    const float3 left = f3(((point.z * (d * gy + a * gx)) - (gy * g * fy_) - (gx * g * fx_)) / z2,
                           ((point.z * (e * gy + b * gx)) - (gy * h * fy_) - (gx * h * fx_)) / z2,
                           ((point.z * (f * gy + c * gx)) - (gy * i * fy_) - (gx * i * fx_)) / z2);
    const float3 jac = cross3(left, point);
    const float row[4] = {jac.x, jac.y, jac.z, -((float)nextImage[wy * W + wx] - (float)lastImage[y * W + x])};
    int q = 0;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int cc = r; cc < 4; ++cc) acc[q++] += row[r] * row[cc];
    acc[9] += row[3] * row[3];
    acc[10] += 1.0f;
}

// K * R * K^-1 and K * t of the inverse of the (rigid) double 4x4 resultRt, cast to float: RGBDOdometry.cpp:363-373.
__device__ __forceinline__ void krk_from_result(const double* resultRt, Intr k, float* krk, float* kt) {
#pragma clang fp contract(off)
    const double* m = resultRt;
    const double r00 = m[0], r01 = m[1], r02 = m[2], r10 = m[4], r11 = m[5], r12 = m[6], r20 = m[8], r21 = m[9], r22 = m[10];
    const double c00 = r11 * r22 - r12 * r21, c01 = r12 * r20 - r10 * r22, c02 = r10 * r21 - r11 * r20;
    const double det = r00 * c00 + r01 * c01 + r02 * c02, id = 1.0 / det;
    double Ri[9];
    Ri[0] = c00 * id; Ri[1] = (r02 * r21 - r01 * r22) * id; Ri[2] = (r01 * r12 - r02 * r11) * id;
    Ri[3] = c01 * id; Ri[4] = (r00 * r22 - r02 * r20) * id; Ri[5] = (r02 * r10 - r00 * r12) * id;
    Ri[6] = c02 * id; Ri[7] = (r01 * r20 - r00 * r21) * id; Ri[8] = (r00 * r11 - r01 * r10) * id;
    double ti[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) ti[r] = -(Ri[r * 3 + 0] * m[3] + Ri[r * 3 + 1] * m[7] + Ri[r * 3 + 2] * m[11]);
    const double K[9] = {(double)k.fx, 0, (double)k.cx, 0, (double)k.fy, (double)k.cy, 0, 0, 1};
    const double Ki[9] = {1.0 / (double)k.fx, 0, -(double)k.cx / (double)k.fx, 0, 1.0 / (double)k.fy, -(double)k.cy / (double)k.fy, 0, 0, 1};
    double KR[9], KRK[9];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) KR[r * 3 + c] = K[r * 3 + 0] * Ri[0 * 3 + c] + K[r * 3 + 1] * Ri[1 * 3 + c] + K[r * 3 + 2] * Ri[2 * 3 + c];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) KRK[r * 3 + c] = KR[r * 3 + 0] * Ki[0 * 3 + c] + KR[r * 3 + 1] * Ki[1 * 3 + c] + KR[r * 3 + 2] * Ki[2 * 3 + c];
#pragma unroll
    for (int q = 0; q < 9; ++q) krk[q] = (float)KRK[q];
#pragma unroll
    for (int r = 0; r < 3; ++r) kt[r] = (float)(K[r * 3 + 0] * ti[0] + K[r * 3 + 1] * ti[1] + K[r * 3 + 2] * ti[2]);
}

}  // namespace mf

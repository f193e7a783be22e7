// mf_rgbd.hip -- image-side kernels of the photometric term and the SO(3) pre-alignment.
//
// Replaces (reference, relative to /root/reference):
//   RGBDOdometry::populateRGBDData / initRGBModel / initRGB / initFirstRGB   Core/Utils/RGBDOdometry.cpp:187-225
//     verticesToDepth, imageBGRToIntensity, pyrDownUcharGauss                  Core/Cuda/cudafuncs.cu:602-639,534-588
//   computeDerivativeImages                                                   Core/Cuda/cudafuncs.cu:658-718
//   the SO(3) block of getIncrementalTransformation + so3Step                 RGBDOdometry.cpp:264-324, reduce.cu:999-1202
//
// The SO(3) loop (19 200 px at VGA level 2, <= 10 iterations) never leaves the device: each iteration is one launch whose
// prologue finishes the previous one (reduction, 3x3 solve, rotation update, convergence tests), instead of ten {kernel,
// reduce kernel, sync, D2H, host solve} rounds.
#pragma clang fp contract(off)  // before the headers: their inline helpers must not be fused either

#include "mf_internal.h"
#include "mf_rgbd_device.h"

namespace mf {

// ---------------- intensity ----------------
__global__ __launch_bounds__(256) void k_intensity(const uint8_t* __restrict__ img, int channels, uint8_t* __restrict__ dst, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint8_t* p = img + (size_t)i * channels;
    dst[i] = intensity_of((float)p[0], (float)p[1], (float)p[2]);
}
void launch_intensity(const uint8_t* img, int channels, uint8_t* dst, int n, hipStream_t s) {
    hipLaunchKernelGGL(k_intensity, dim3((n + 255) / 256), dim3(256), 0, s, img, channels, dst, n);
}

// Level 0 of the "last" pyramids of one model: depth from the vertex map initICPModel was given (prediction, or the
// fill-in where the prediction is empty), intensity from the matching image (written at predict time).
__global__ __launch_bounds__(256) void k_rgbd_last_l0(const float4* __restrict__ predV, const float* __restrict__ fillDepth,
                                                      const uint8_t* __restrict__ predGray, const uint8_t* __restrict__ fillGray,
                                                      const FrameDev* __restrict__ frame, float cutOff, float* __restrict__ depth0,
                                                      uint8_t* __restrict__ image0, int n, int frameToFrameRGB) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const bool useFill = fillDepth != nullptr && frame->useFillIn != 0;
    float z = predV[i].z;
    if (useFill && z == 0) z = fillDepth[i];                       // fill_vertex.frag:37-53
    depth0[i] = (z > cutOff || z <= 0) ? qnan() : z;               // verticesToDepthKernel
    image0[i] = (useFill || (frameToFrameRGB != 0 && fillGray != nullptr)) ? fillGray[i] : predGray[i];   // Model.cpp:395-401
}
void launch_rgbd_last_l0(const float4* predV, const float* fillDepth, const uint8_t* predGray, const uint8_t* fillGray,
                         const FrameDev* frame, float* depth0, uint8_t* image0, int n, hipStream_t s, int frameToFrameRGB) {
    hipLaunchKernelGGL(k_rgbd_last_l0, dim3((n + 255) / 256), dim3(256), 0, s, predV, fillDepth, predGray, fillGray, frame, 6.0f,
                       depth0, image0, n, frameToFrameRGB);  // maxDepthRGB = 6, RGBDOdometry.cpp:34
}

// ---------------- pyrDownUcharGauss (cudafuncs.cu:534-588; border quirk Q9, zero texels skipped) ----------------
__device__ __forceinline__ float gauss5w(int r, int c) {
    const int a = (r == 0 || r == 4) ? 1 : ((r == 2) ? 6 : 4);
    const int b = (c == 0 || c == 4) ? 1 : ((c == 2) ? 6 : 4);
    return (float)(a * b);
}
__device__ __forceinline__ void pyrdown_u8_body(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int sw, int sh, int bx, int by) {
    const int dw = sw / 2, dh = sh / 2;
    const int x = bx * 64 + (threadIdx.x & 63), y = by * 4 + (threadIdx.x >> 6);
    if (x >= dw || y >= dh) return;
    const int D = 5;
    const int tx = min(2 * x - D / 2 + D, sw - 1), ty = min(2 * y - D / 2 + D, sh - 1);
    float sum = 0.f;
    int count = 0;
    for (int cy = max(0, 2 * y - D / 2); cy < ty; ++cy)
        for (int cx = max(0, 2 * x - D / 2); cx < tx; ++cx) {
            const int v = src[cy * sw + cx];
            if (v > 0) {
                const float w = gauss5w(ty - cy - 1, tx - cx - 1);
                sum += (float)v * w;
                count += (int)w;
            }
        }
    const float r = sum / (float)count;             // 0 / 0 -> NaN -> 0 (cvt semantics of the reference)
    dst[y * dw + x] = isnan(r) ? (uint8_t)0 : (uint8_t)(int)r;
}
__global__ __launch_bounds__(256) void k_pyrdown_u8(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int sw, int sh) {
    pyrdown_u8_body(src, dst, sw, sh, blockIdx.x, blockIdx.y);
}
void launch_pyrdown_u8(const uint8_t* src, uint8_t* dst, int sw, int sh, hipStream_t s) {
    dim3 grid((sw / 2 + 63) / 64, (sh / 2 + 3) / 4);
    hipLaunchKernelGGL(k_pyrdown_u8, grid, dim3(256), 0, s, src, dst, sw, sh);
}
// pyrDownGaussF (cudafuncs.cu:510-532) with the per-pixel expressions of k_pyrdown_f (mf_preproc.hip; both files round every operation on
// its own): the copy that rides in the combined launches below
__device__ __forceinline__ float gauss5f(int i) { return i == 2 ? 6.f : ((i == 1 || i == 3) ? 4.f : 1.f); }
__device__ __forceinline__ void pyrdown_f_body(const float* __restrict__ src, float* __restrict__ dst, int sw, int sh, int bx, int by) {
    const int dw = sw >> 1, dh = sh >> 1;
    const int x = bx * 64 + (threadIdx.x & 63);
    const int y = by * 4 + (threadIdx.x >> 6);
    if (x >= dw || y >= dh) return;
    const int tx = min(2 * x + 3, sw - 1);
    const int ty = min(2 * y + 3, sh - 1);
    float sum = 0.f;
    int count = 0;
    for (int cy = max(0, 2 * y - 2); cy < ty; ++cy) {
        for (int cx = max(0, 2 * x - 2); cx < tx; ++cx) {
            const float v = src[cy * sw + cx];
            if (!isnan(v)) {
                const float w = gauss5f(ty - cy - 1) * gauss5f(tx - cx - 1);
                sum += v * w;
                count += (int)w;
            }
        }
    }
    dst[y * dw + x] = sum / (float)count;
}

// ---------------- computeDerivativeImages ----------------
__device__ __forceinline__ void derivative_body(const uint8_t* __restrict__ src, int16_t* __restrict__ dx, int16_t* __restrict__ dy, int W, int H,
                                                float minScale, uint8_t* __restrict__ gate, int bx, int by) {
    const int x = bx * 64 + (threadIdx.x & 63), y = by * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    const float gx[9] = {0.52201f, 0.00000f, -0.52201f, 0.79451f, -0.00000f, -0.79451f, 0.52201f, 0.00000f, -0.52201f};
    const float gy[9] = {0.52201f, 0.79451f, 0.52201f, 0.00000f, 0.00000f, 0.00000f, -0.52201f, -0.79451f, -0.52201f};
    float dxVal = 0.f, dyVal = 0.f;
    int k = 8;
    for (int j = max(y - 1, 0); j <= min(y + 1, H - 1); ++j)
        for (int i = max(x - 1, 0); i <= min(x + 1, W - 1); ++i) {
            const float sv = (float)src[j * W + i];
            // the kernel index walks backwards over the CLAMPED window (border quirk); selects instead of a dynamically
            // indexed constant array
            float wx = 0.f, wy = 0.f;
#pragma unroll
            for (int q = 0; q < 9; ++q) { wx = (q == k) ? gx[q] : wx; wy = (q == k) ? gy[q] : wy; }
            dxVal = fmaf(sv, wx, dxVal);
            dyVal = fmaf(sv, wy, dyVal);
            --k;
        }
    const int16_t sx = (int16_t)dxVal, sy = (int16_t)dyVal;
    dx[y * W + x] = sx;
    dy[y * W + x] = sy;
    if (gate) gate[y * W + x] = rgb_gate_px(src, sx, sy, minScale, W, H, x, y) ? 1 : 0;
}
__global__ __launch_bounds__(256) void k_derivative(const uint8_t* __restrict__ src, int16_t* __restrict__ dx,
                                                    int16_t* __restrict__ dy, int W, int H, float minScale, uint8_t* __restrict__ gate) {
    derivative_body(src, dx, dy, W, H, minScale, gate, blockIdx.x, blockIdx.y);
}
void launch_derivative(const uint8_t* src, int16_t* dx, int16_t* dy, int W, int H, float minScale, uint8_t* gate, hipStream_t s) {
    dim3 grid((W + 63) / 64, (H + 3) / 4);
    hipLaunchKernelGGL(k_derivative, grid, dim3(256), 0, s, src, dx, dy, W, H, minScale, gate);
}

// Up to three INDEPENDENT small image jobs in one launch (grid.z = job): the derivative images of the three pyramid levels; one level of a
// model's "last" depth pyramid together with the same level of its "last" intensity pyramid.  Each of these is a 5-8 us launch on an image
// of at most 0.3 M pixels -- launch latency, not work -- and the reference-default frame carried nine of them one after the other (72 us of
// 617, profiles/r04_rgbd_kernel_stats.csv).  The bodies are the single-job kernels' bodies: same bits.
__global__ __launch_bounds__(256) void k_small_jobs(const SmallJobs a) {
    const SmallJob& j = a.j[blockIdx.z];
    switch (j.kind) {
        case 0: pyrdown_u8_body(static_cast<const uint8_t*>(j.src), static_cast<uint8_t*>(j.dst), j.W, j.H, blockIdx.x, blockIdx.y); break;
        case 1: pyrdown_f_body(static_cast<const float*>(j.src), static_cast<float*>(j.dst), j.W, j.H, blockIdx.x, blockIdx.y); break;
        default: derivative_body(static_cast<const uint8_t*>(j.src), static_cast<int16_t*>(j.dst), static_cast<int16_t*>(j.dst2), j.W, j.H, j.minScale,
                                 j.gate, blockIdx.x, blockIdx.y); break;
    }
}
void launch_small_jobs(const SmallJobs& a, hipStream_t s) {
    int gx = 1, gy = 1;
    for (int i = 0; i < a.n; ++i) {
        const int w = a.j[i].kind == 2 ? a.j[i].W : a.j[i].W / 2, h = a.j[i].kind == 2 ? a.j[i].H : a.j[i].H / 2;   // output size of the job
        gx = max(gx, (w + 63) / 64); gy = max(gy, (h + 3) / 4);
    }
    hipLaunchKernelGGL(k_small_jobs, dim3(gx, gy, a.n), dim3(256), 0, s, a);
}

// ---------------- the frame's intensity pyramid, derivative and gate images in ONE launch ----------------
// imageBGRToIntensity + 2 x pyrDownUcharGauss + computeDerivativeImages x 3 (+ the pose-independent gate of reduce.cu:823-851 x 3) were four
// dependent launches of 7-8 us each on images of at most 0.3 M pixels (launch latency, not work).  Here a workgroup owns a 32 x 32-pixel tile of
// level 0 (16 x 16 of level 1, 8 x 8 of level 2), computes the intensity of the tile and of the halo its coarser levels and their windows
// need -- 57 x 57 texels of level 0 -> 27 x 27 of level 1 -> 12 x 12 of level 2 -- into the LDS, and runs the same per-pixel expressions as the
// single kernels above on the LDS copies (the window rules depend on the pixel's GLOBAL coordinates only): same bytes, one launch.
constexpr int kRpThreads = 1024;   // 256 threads took 43 us for what four launches do in 30 (a workgroup's phases are serial: 13 + 3 + 1 + 6 rounds per thread); 1024: a quarter of the rounds
constexpr int kRpT2 = 8, kRpR2 = kRpT2 + 4, kRpR1 = 2 * kRpT2 + 11, kRpR0 = 4 * kRpT2 + 25;
struct RgbPyrArgs {
    const uint8_t* rgb; int W, H;
    uint8_t* gray[3]; int16_t* dIdx[3]; int16_t* dIdy[3]; uint8_t* gate[3]; float minScale[3];
    int derivatives;      // 0: the three intensity levels only (SO(3) without the photometric term)
};
// pyrdown_u8_body's pixel through an accessor (global level coordinates -> texel)
template <class Src>
__device__ __forceinline__ uint8_t pyrdown_u8_px(Src src, int sw, int sh, int x, int y) {
    const int D = 5;
    const int tx = min(2 * x - D / 2 + D, sw - 1), ty = min(2 * y - D / 2 + D, sh - 1);
    float sum = 0.f;
    int count = 0;
    for (int cy = max(0, 2 * y - D / 2); cy < ty; ++cy)
        for (int cx = max(0, 2 * x - D / 2); cx < tx; ++cx) {
            const int v = src(cx, cy);
            if (v > 0) {
                const float w = gauss5w(ty - cy - 1, tx - cx - 1);
                sum += (float)v * w;
                count += (int)w;
            }
        }
    const float r = sum / (float)count;
    return isnan(r) ? (uint8_t)0 : (uint8_t)(int)r;
}
// derivative_body's pixel (and its gate) through an accessor
template <class Src>
__device__ __forceinline__ void derivative_px(Src src, int W, int H, int x, int y, float minScale, int16_t& sx, int16_t& sy, bool& gate) {
    const float gx[9] = {0.52201f, 0.00000f, -0.52201f, 0.79451f, -0.00000f, -0.79451f, 0.52201f, 0.00000f, -0.52201f};
    const float gy[9] = {0.52201f, 0.79451f, 0.52201f, 0.00000f, 0.00000f, 0.00000f, -0.52201f, -0.79451f, -0.52201f};
    float dxVal = 0.f, dyVal = 0.f;
    int k = 8;
    for (int j = max(y - 1, 0); j <= min(y + 1, H - 1); ++j)
        for (int i = max(x - 1, 0); i <= min(x + 1, W - 1); ++i) {
            const float sv = (float)src(i, j);
            float wx = 0.f, wy = 0.f;
#pragma unroll
            for (int q = 0; q < 9; ++q) { wx = (q == k) ? gx[q] : wx; wy = (q == k) ? gy[q] : wy; }
            dxVal = fmaf(sv, wx, dxVal);
            dyVal = fmaf(sv, wy, dyVal);
            --k;
        }
    sx = (int16_t)dxVal; sy = (int16_t)dyVal;
    // rgb_gate_px (mf_rgbd_device.h) on the accessor
    gate = false;
    if (x < W - 5 && y < H - 1) {
        bool valid = true;
        for (int u = max(y - 2, 0); u < min(y + 2, H); ++u)
            for (int v = max(x - 2, 0); v < min(x + 2, W); ++v) valid = valid && (src(v, u) > 0);
        if (valid) gate = (float)(((int)sx * (int)sx) + ((int)sy * (int)sy)) >= minScale;
    }
}
__global__ __launch_bounds__(kRpThreads) void k_rgb_pyramid(const RgbPyrArgs a) {
    __shared__ uint8_t s_g0[kRpR0 * kRpR0], s_g1[kRpR1 * kRpR1], s_g2[kRpR2 * kRpR2];
    const int W0 = a.W, H0 = a.H, W1 = W0 / 2, H1 = H0 / 2, W2 = W1 / 2, H2 = H1 / 2;
    const int tilesX = (W2 + kRpT2 - 1) / kRpT2;
    const int X2 = ((int)blockIdx.x % tilesX) * kRpT2, Y2 = ((int)blockIdx.x / tilesX) * kRpT2;
    const int ox2 = X2 - 2, oy2 = Y2 - 2, ox1 = 2 * X2 - 6, oy1 = 2 * Y2 - 6, ox0 = 4 * X2 - 14, oy0 = 4 * Y2 - 14;
    // level 0: intensity of the tile and its halo
    for (int t = threadIdx.x; t < kRpR0 * kRpR0; t += kRpThreads) {
        const int x = ox0 + t % kRpR0, y = oy0 + t / kRpR0;
        uint8_t v = 0;
        if (x >= 0 && y >= 0 && x < W0 && y < H0) {
            const uint8_t* p = a.rgb + ((size_t)y * W0 + x) * 3;
            v = intensity_of((float)p[0], (float)p[1], (float)p[2]);
        }
        s_g0[t] = v;
    }
    __syncthreads();
    auto g0 = [&](int x, int y) -> int { return s_g0[(y - oy0) * kRpR0 + (x - ox0)]; };
    for (int t = threadIdx.x; t < kRpR1 * kRpR1; t += kRpThreads) {
        const int x = ox1 + t % kRpR1, y = oy1 + t / kRpR1;
        s_g1[t] = (x >= 0 && y >= 0 && x < W1 && y < H1) ? pyrdown_u8_px(g0, W0, H0, x, y) : (uint8_t)0;
    }
    __syncthreads();
    auto g1 = [&](int x, int y) -> int { return s_g1[(y - oy1) * kRpR1 + (x - ox1)]; };
    for (int t = threadIdx.x; t < kRpR2 * kRpR2; t += kRpThreads) {
        const int x = ox2 + t % kRpR2, y = oy2 + t / kRpR2;
        s_g2[t] = (x >= 0 && y >= 0 && x < W2 && y < H2) ? pyrdown_u8_px(g1, W1, H1, x, y) : (uint8_t)0;
    }
    __syncthreads();
    auto g2 = [&](int x, int y) -> int { return s_g2[(y - oy2) * kRpR2 + (x - ox2)]; };
    // the tile's own pixels of the three levels: intensity out, derivative / gate images
    for (int t = threadIdx.x; t < 16 * kRpT2 * kRpT2; t += kRpThreads) {
        const int x = 4 * X2 + t % (4 * kRpT2), y = 4 * Y2 + t / (4 * kRpT2);
        if (x < W0 && y < H0) {
            a.gray[0][y * W0 + x] = (uint8_t)g0(x, y);
            if (a.derivatives) {
                int16_t sx, sy; bool gt;
                derivative_px(g0, W0, H0, x, y, a.minScale[0], sx, sy, gt);
                a.dIdx[0][y * W0 + x] = sx; a.dIdy[0][y * W0 + x] = sy;
                if (a.gate[0]) a.gate[0][y * W0 + x] = gt ? 1 : 0;
            }
        }
    }
    for (int t = threadIdx.x; t < 4 * kRpT2 * kRpT2; t += kRpThreads) {
        const int x = 2 * X2 + t % (2 * kRpT2), y = 2 * Y2 + t / (2 * kRpT2);
        if (x < W1 && y < H1) {
            a.gray[1][y * W1 + x] = (uint8_t)g1(x, y);
            if (a.derivatives) {
                int16_t sx, sy; bool gt;
                derivative_px(g1, W1, H1, x, y, a.minScale[1], sx, sy, gt);
                a.dIdx[1][y * W1 + x] = sx; a.dIdy[1][y * W1 + x] = sy;
                if (a.gate[1]) a.gate[1][y * W1 + x] = gt ? 1 : 0;
            }
        }
    }
    for (int t = threadIdx.x; t < kRpT2 * kRpT2; t += kRpThreads) {
        const int x = X2 + t % kRpT2, y = Y2 + t / kRpT2;
        if (x < W2 && y < H2) {
            a.gray[2][y * W2 + x] = (uint8_t)g2(x, y);
            if (a.derivatives) {
                int16_t sx, sy; bool gt;
                derivative_px(g2, W2, H2, x, y, a.minScale[2], sx, sy, gt);
                a.dIdx[2][y * W2 + x] = sx; a.dIdy[2][y * W2 + x] = sy;
                if (a.gate[2]) a.gate[2][y * W2 + x] = gt ? 1 : 0;
            }
        }
    }
}
// true when the launch was made (sizes the tiling covers: both halvings exact); false: the caller runs the single kernels
bool launch_rgb_pyramid(const uint8_t* rgb, int W, int H, uint8_t* const gray[3], int16_t* const dIdx[3], int16_t* const dIdy[3], uint8_t* const gate[3],
                        const float minScale[3], bool derivatives, hipStream_t s) {
    if ((W & 3) || (H & 3)) return false;
    RgbPyrArgs a;
    a.rgb = rgb; a.W = W; a.H = H; a.derivatives = derivatives ? 1 : 0;
    for (int i = 0; i < 3; ++i) { a.gray[i] = gray[i]; a.dIdx[i] = dIdx[i]; a.dIdy[i] = dIdy[i]; a.gate[i] = gate[i]; a.minScale[i] = minScale[i]; }
    const int W2 = W / 4, H2 = H / 4;
    hipLaunchKernelGGL(k_rgb_pyramid, dim3(((W2 + kRpT2 - 1) / kRpT2) * ((H2 + kRpT2 - 1) / kRpT2)), dim3(kRpThreads), 0, s, a);
    return true;
}

// ---------------- SO(3) pre-alignment: all iterations in one workgroup ----------------
__device__ __forceinline__ void so3_bases(const double* R, Intr k, float* basis) {  // RGBDOdometry.cpp:290-302
    const double K[9] = {(double)k.fx, 0, (double)k.cx, 0, (double)k.fy, (double)k.cy, 0, 0, 1};
    const double Ki[9] = {1.0 / (double)k.fx, 0, -(double)k.cx / (double)k.fx, 0, 1.0 / (double)k.fy, -(double)k.cy / (double)k.fy, 0, 0, 1};
    double KR[9], H[9];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) KR[r * 3 + c] = K[r * 3 + 0] * R[0 * 3 + c] + K[r * 3 + 1] * R[1 * 3 + c] + K[r * 3 + 2] * R[2 * 3 + c];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) H[r * 3 + c] = KR[r * 3 + 0] * Ki[0 * 3 + c] + KR[r * 3 + 1] * Ki[1 * 3 + c] + KR[r * 3 + 2] * Ki[2 * 3 + c];
    for (int q = 0; q < 9; ++q) { basis[q] = (float)H[q]; basis[9 + q] = (float)Ki[q]; basis[18 + q] = (float)KR[q]; }
}

// Symmetric 3x3 solve.  Eigen's float LDLT (RGBDOdometry.cpp:313) is replaced by the closed form in double, rounded to
// float: it differs from the float factorisation by rounding only (the oracle keeps the float LDLT).
__device__ __forceinline__ void solve3(const float* A, const float* b, float* x) {
    const double a00 = A[0], a01 = A[1], a02 = A[2], a11 = A[4], a12 = A[5], a22 = A[8];
    const double c00 = a11 * a22 - a12 * a12, c01 = a12 * a02 - a01 * a22, c02 = a01 * a12 - a11 * a02;
    const double c11 = a00 * a22 - a02 * a02, c12 = a01 * a02 - a00 * a12, c22 = a00 * a11 - a01 * a01;
    const double det = a00 * c00 + a01 * c01 + a02 * c02;
    if (!(fabs(det) > 0.0)) { x[0] = x[1] = x[2] = 0.f; return; }
    const double id = 1.0 / det;
    x[0] = (float)((c00 * b[0] + c01 * b[1] + c02 * b[2]) * id);
    x[1] = (float)((c01 * b[0] + c11 * b[1] + c12 * b[2]) * id);
    x[2] = (float)((c02 * b[0] + c12 * b[1] + c22 * b[2]) * id);
}

// One launch per SO(3) iteration, structured like the ICP loop: the prologue of launch j (every workgroup, redundantly and
// bit-identically) finishes iteration j-1 -- reduces its per-workgroup partial sums in workgroup order, applies the
// convergence / divergence rules, solves the 3x3 system and updates the rotation -- then the pixels of iteration j are spread
// over ~75 workgroups.  Once the loop has ended (`done`), the remaining launches only forward the state (~2.5 us each).
// (All iterations inside ONE 1024-thread workgroup cost 186 us per frame: 19 200 pixels x ~200 instructions on a single CU.)
constexpr int kSo3Threads = 256;
constexpr int kSo3Slots = 12;   // 11 accumulators padded

struct So3State {
    double R[9], lastR[9];
    float R_lr[9];
    float lastError, lastCount, err, cnt;
    int iters, done;
};

__device__ __forceinline__ void so3_state_init(So3State& st) {
    for (int q = 0; q < 9; ++q) { st.R[q] = st.lastR[q] = (q % 4 == 0) ? 1.0 : 0.0; st.R_lr[q] = (q % 4 == 0) ? 1.f : 0.f; }
    st.lastError = st.lastCount = 3.4028234664e38f / 2;
    st.err = st.cnt = 0.f; st.iters = 0; st.done = 0;
}

// end of one iteration (RGBDOdometry.cpp:301-323) given the reduced sums
__device__ __forceinline__ void so3_finish_iteration(const float* sum, So3State& st) {
    ++st.iters;
    const float jtj[9] = {sum[0], sum[1], sum[2], sum[1], sum[4], sum[5], sum[2], sum[5], sum[7]};
    const float jtr[3] = {sum[3], sum[6], sum[8]};
    st.err = sqrtf(sum[9]) / sum[10];
    st.cnt = sum[10];
    if (st.err < st.lastError && fabsf(st.lastError - st.cnt) < 0.001f) {            // sic (RGBDOdometry.cpp:305)
        st.done = 1;
    } else if (st.err > st.lastError + 0.001f) {                                        // diverging, :307-312
        st.err = st.lastError; st.cnt = st.lastCount;
        for (int q = 0; q < 9; ++q) st.R[q] = st.lastR[q];
        st.done = 1;
    } else {
        st.lastError = st.err; st.lastCount = st.cnt;
        for (int q = 0; q < 9; ++q) st.lastR[q] = st.R[q];
        float delta[3];
        solve3(jtj, jtr, delta);
        double Rw[3][3];
        rodrigues_d((double)delta[0], (double)delta[1], (double)delta[2], Rw);
        float nl[9];
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c)
                nl[r * 3 + c] = (float)Rw[r][0] * st.R_lr[0 * 3 + c] + (float)Rw[r][1] * st.R_lr[1 * 3 + c] + (float)Rw[r][2] * st.R_lr[2 * 3 + c];
        for (int q = 0; q < 9; ++q) { st.R_lr[q] = nl[q]; st.R[q] = (double)nl[q]; }
    }
}

// loads the state, finishes the previous iteration from its partials; returns with the state in s_st (all threads synced)
__device__ __forceinline__ void so3_prologue(const So3State* __restrict__ st_in, const float* __restrict__ part_in, int nb_in, int first,
                                             So3State& s_st, float* s_p /*[nb_in * kSo3Slots]*/, float* s_sum /*[12]*/) {
    const int tid = threadIdx.x;
    if (first) {
        if (tid == 0) so3_state_init(s_st);
    } else {
        if (tid < (int)(sizeof(So3State) / 4)) reinterpret_cast<uint32_t*>(&s_st)[tid] = reinterpret_cast<const uint32_t*>(st_in)[tid];
        for (int i = tid; i < nb_in * kSo3Slots; i += kSo3Threads) s_p[i] = part_in[i];
    }
    __syncthreads();
    if (!first && !s_st.done) {
        if (tid < 11) {
            float v = 0.f;
            for (int b = 0; b < nb_in; ++b) v += s_p[b * kSo3Slots + tid];
            s_sum[tid] = v;
        }
        __syncthreads();
        if (tid == 0) so3_finish_iteration(s_sum, s_st);
    }
    __syncthreads();
}

constexpr int kSo3MaxBlocks = 512;   // LDS staging of the previous launch's partials

__global__ __launch_bounds__(kSo3Threads) void k_so3_iter(const uint8_t* __restrict__ lastImage, const uint8_t* __restrict__ nextImage,
                                                           int W, int H, Intr k, const So3State* __restrict__ st_in,
                                                           So3State* __restrict__ st_out, const float* __restrict__ part_in, int nb_in,
                                                           float* __restrict__ part_out, int first) {
    __shared__ So3State s_st;
    __shared__ float s_p[kSo3MaxBlocks * kSo3Slots];
    __shared__ float s_sum[kSo3Slots];
    __shared__ float s_basis[27];
    __shared__ float s_red[(kSo3Threads / 64) * kSo3Slots];
    so3_prologue(st_in, part_in, nb_in, first, s_st, s_p, s_sum);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool done = s_st.done != 0;
    if (tid == 0) {
        if (!done) so3_bases(s_st.R, k, s_basis);
        if (blockIdx.x == 0) *st_out = s_st;
    }
    if (done) return;
    __syncthreads();
    float acc[11];
#pragma unroll
    for (int q = 0; q < 11; ++q) acc[q] = 0.f;
    const int p = blockIdx.x * kSo3Threads + tid;
    if (p < W * H) {
        const int y = p / W, x = p - y * W;
        so3_px(lastImage, nextImage, W, H, s_basis, x, y, acc);
    }
#pragma unroll
    for (int q = 0; q < 11; ++q) {
        float v = acc[q];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if (lane == 0) s_red[wave * kSo3Slots + q] = v;
    }
    __syncthreads();
    if (tid < kSo3Slots) {
        float v = 0.f;
        if (tid < 11)
            for (int w = 0; w < kSo3Threads / 64; ++w) v += s_red[w * kSo3Slots + tid];
        part_out[blockIdx.x * kSo3Slots + tid] = v;
    }
}

__global__ __launch_bounds__(kSo3Threads) void k_so3_final(const So3State* __restrict__ st_in, const float* __restrict__ part_in, int nb_in,
                                                            So3Result* __restrict__ out) {
    __shared__ So3State s_st;
    __shared__ float s_p[kSo3MaxBlocks * kSo3Slots];
    __shared__ float s_sum[kSo3Slots];
    so3_prologue(st_in, part_in, nb_in, 0, s_st, s_p, s_sum);
    if (threadIdx.x == 0) {
        for (int q = 0; q < 9; ++q) out->R[q] = s_st.R[q];
        out->error = s_st.err; out->count = s_st.cnt; out->iterations = s_st.iters; out->pad = 0;
    }
}

size_t so3_scratch_bytes(int W2, int H2) {
    const size_t nb = ((size_t)W2 * H2 + kSo3Threads - 1) / kSo3Threads;
    return 2 * sizeof(So3State) + 2 * nb * kSo3Slots * sizeof(float) + 64;
}

int launch_so3_prealign(const uint8_t* lastImage2, const uint8_t* nextImage2, int W2, int H2, Intr k2, So3Result* out, void* scratch,
                        hipStream_t s) {
    const int nb = (W2 * H2 + kSo3Threads - 1) / kSo3Threads;
    if (nb > kSo3MaxBlocks) return -1;
    So3State* st = reinterpret_cast<So3State*>(scratch);
    float* part = reinterpret_cast<float*>(st + 2);
    for (int j = 0; j < 10; ++j)   // RGBDOdometry.cpp:283: at most ten iterations
        hipLaunchKernelGGL(k_so3_iter, dim3(nb), dim3(kSo3Threads), 0, s, lastImage2, nextImage2, W2, H2, k2, st + (j & 1), st + ((j + 1) & 1),
                           part + (size_t)((j + 1) & 1) * nb * kSo3Slots, nb, part + (size_t)(j & 1) * nb * kSo3Slots, j == 0 ? 1 : 0);
    hipLaunchKernelGGL(k_so3_final, dim3(1), dim3(kSo3Threads), 0, s, st + 0, part + (size_t)1 * nb * kSo3Slots, nb, out);
    return 0;
}

// ---------------- stand-alone residual / step (parity tests; the tracking loop uses the fused kernels) ----------------
__global__ __launch_bounds__(256) void k_rgb_residual_only(const RgbLevel L, const float* __restrict__ krk_kt, RgbCorr* __restrict__ corres,
                                                           int* __restrict__ sums /*[2]: count, sigma (pre-zeroed)*/) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= L.W * L.H) return;
    const int y = i / L.W, x = i - y * L.W;
    float krk[9];
    for (int q = 0; q < 9; ++q) krk[q] = krk_kt[q];
    RgbCorr c;
    const bool ok = rgb_residual_px(L, krk, f3(krk_kt[9], krk_kt[10], krk_kt[11]), x, y, c);
    corres[i] = c;
    if (ok) { atomicAdd(&sums[0], 1); atomicAdd(&sums[1], (int)(c.diff * c.diff)); }
}
__global__ __launch_bounds__(256) void k_rgb_step_only(const RgbLevel L, const RgbCorr* __restrict__ corres, float sigma, Intr k,
                                                       float sobelScale, double* __restrict__ out32 /*pre-zeroed*/) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= L.W * L.H) return;
    const RgbCorr c = corres[i];
    if (c.u0 < 0) return;
    float acc[32];
    for (int q = 0; q < 32; ++q) acc[q] = 0.f;
    rgb_step_px(L, c, i % L.W, i / L.W, sigma, k, sobelScale, acc);
    for (int q = 0; q < 29; ++q) atomicAdd(&out32[q], (double)acc[q]);
}
void launch_rgb_residual_only(const RgbLevel& L, const float* krk_kt, RgbCorr* corres, int* sums, hipStream_t s) {
    hipLaunchKernelGGL(k_rgb_residual_only, dim3((L.W * L.H + 255) / 256), dim3(256), 0, s, L, krk_kt, corres, sums);
}
void launch_rgb_step_only(const RgbLevel& L, const RgbCorr* corres, float sigma, Intr k, float sobelScale, double* out32, hipStream_t s) {
    hipLaunchKernelGGL(k_rgb_step_only, dim3((L.W * L.H + 255) / 256), dim3(256), 0, s, L, corres, sigma, k, sobelScale, out32);
}

}  // namespace mf

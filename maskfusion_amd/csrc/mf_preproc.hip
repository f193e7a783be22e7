// mf_preproc.hip -- per-frame image preprocessing on plain HBM arrays.
//   bilateral depth filter   <- MaskFusion::filterDepth + depth_bilateral_metric.frag (Core/MaskFusion.cpp:650-657)
//   depth pyramid            <- pyrDownGaussF (Core/Cuda/cudafuncs.cu:333-364, 510-532)
//   vertex + normal maps     <- createVMap + createNMap (Core/Cuda/cudafuncs.cu:109-205), fused into one pass
// every float op individually rounded in this file (the preprocessing feeds normals, which amplify 1-ulp depth
// differences ~1000x; keeping these maps within rounding of a plain reading of the reference keeps parity tight).
// BEFORE the header: its inline helpers (cross3, dot3, normalized_rsqrt) are compiled under whatever is in force where they are
// DEFINED -- until round 3 the pragma sat below the include, createNMap's cross product was fused, and the normal maps differed from
// the restatement's in the last bit on hardware only (the CPU-executed build has contraction off everywhere): 0.03-0.2 % of the label
// pixels of the 8-object scene.
#pragma clang fp contract(off)
#include "mf_device.h"
#include "mf_bilateral_device.h"

namespace mf {

// 13x13 bilateral: the body lives in mf_bilateral_device.h (shared with the launch that runs it beside the model-side pyramid)
__global__ __launch_bounds__(256) void k_bilateral(const float* __restrict__ depth, float* __restrict__ out, int W, int H) {
    __shared__ float tile[kBLdsH * kBLdsW];
    bilateral_body(depth, out, W, H, tile, (int)blockIdx.x);
}

void launch_bilateral(const float* depth, float* out, int W, int H, hipStream_t s) {
    hipLaunchKernelGGL(k_bilateral, dim3(bilateral_grid(W, H)), dim3(256), 0, s, depth, out, W, H);
}

// ------------------------------------------------------------------------------------------------
// 5x5 Gaussian half-sampling that skips NaNs, with the reference's border quirk: the upper loop
// bounds clamp to cols-1 / rows-1 exclusive and the kernel is indexed from the far corner.
// ------------------------------------------------------------------------------------------------
// binomial row {1,4,6,4,1}; the 5x5 kernel of pyrDownGaussF (cudafuncs.cu:517-521) is its outer product
__device__ __forceinline__ float gauss5(int i) { return i == 2 ? 6.f : ((i == 1 || i == 3) ? 4.f : 1.f); }

__global__ __launch_bounds__(256) void k_pyrdown_f(const float* __restrict__ src, float* __restrict__ dst, int sw, int sh) {
    const int dw = sw >> 1, dh = sh >> 1;
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= dw || y >= dh) return;
    const int tx = min(2 * x + 3, sw - 1);
    const int ty = min(2 * y + 3, sh - 1);
    float sum = 0.f;
    int count = 0;
    for (int cy = max(0, 2 * y - 2); cy < ty; ++cy) {
        for (int cx = max(0, 2 * x - 2); cx < tx; ++cx) {
            const float v = src[cy * sw + cx];
            if (!isnan(v)) {
                const float w = gauss5(ty - cy - 1) * gauss5(tx - cx - 1);
                sum += v * w;
                count += (int)w;
            }
        }
    }
    dst[y * dw + x] = sum / (float)count;
}

void launch_pyrdown_f(const float* src, float* dst, int sw, int sh, hipStream_t s) {
    const int dw = sw / 2, dh = sh / 2;
    dim3 grid((dw + 63) / 64, (dh + 3) / 4);
    hipLaunchKernelGGL(k_pyrdown_f, grid, dim3(256), 0, s, src, dst, sw, sh);
}

// ------------------------------------------------------------------------------------------------
// Vertex map and normal map of one level in one pass: the three vertices a normal needs are recomputed from
// depth (bitwise the values createVMap stores), so the normal never waits for a vertex-map round trip.
// Planar SoA [3][H][W] outputs keep the ICP loads of 64 consecutive lanes on one 256 B line per plane.
// Invalid pixels get NaN in all three planes (the reference writes x = NaN only; consumers test x only).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool vertex_from_depth(float z, int u, int v, Intr k, float fx_inv, float fy_inv, float cutoff,
                                                  float3& out) {
    if (z > 0.0f && z < cutoff) {
        out = f3(z * ((float)u - k.cx) * fx_inv, z * ((float)v - k.cy) * fy_inv, z);
        return true;
    }
    out = f3(qnan(), qnan(), qnan());
    return false;
}

__global__ __launch_bounds__(256) void k_vmap_nmap(const float* __restrict__ depth, float* __restrict__ vmap,
                                                   float* __restrict__ nmap, int W, int H, Intr k, float cutoff) {
    const int u = blockIdx.x * 64 + (threadIdx.x & 63);
    const int v = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (u >= W || v >= H) return;
    const int P = W * H, i = v * W + u;
    const float fx_inv = 1.f / k.fx, fy_inv = 1.f / k.fy;
    float3 v00, v01, v10;
    const bool ok00 = vertex_from_depth(depth[i], u, v, k, fx_inv, fy_inv, cutoff, v00);
    vmap[i] = v00.x; vmap[P + i] = v00.y; vmap[2 * P + i] = v00.z;
    float3 n = f3(qnan(), qnan(), qnan());
    if (u < W - 1 && v < H - 1) {
        const bool ok01 = vertex_from_depth(depth[i + 1], u + 1, v, k, fx_inv, fy_inv, cutoff, v01);
        const bool ok10 = vertex_from_depth(depth[i + W], u, v + 1, k, fx_inv, fy_inv, cutoff, v10);
        if (ok00 && ok01 && ok10) n = normalized_rsqrt(cross3(v01 - v00, v10 - v00));
    }
    nmap[i] = n.x; nmap[P + i] = n.y; nmap[2 * P + i] = n.z;
}

void launch_vmap_nmap(const float* depth, float* vmap, float* nmap, int W, int H, Intr k, float cutoff, hipStream_t s) {
    dim3 grid((W + 63) / 64, (H + 3) / 4);
    hipLaunchKernelGGL(k_vmap_nmap, grid, dim3(256), 0, s, depth, vmap, nmap, W, H, k, cutoff);
}

// ------------------------------------------------------------------------------------------------
// Model::generateCUDATextures in ONE launch (Core/Model/Model.cpp:350-389): pyrDownGaussF x2 + createVMap/createNMap x3.
// A 256-thread workgroup owns a 4x4 tile of level 2 = 8x8 of level 1 = 16x16 of level 0 (1200 workgroups at VGA; 8x8 tiles
// = 300 workgroups left the chip half empty: 17 us).  It stages the 29x29 level-0 depths those need (5x5 taps of 5x5 taps
// + the +1 neighbours of the normals), builds the 13x13 level-1 and 5x5 level-2 depths in LDS with exactly the per-pixel expressions of k_pyrdown_f (same loop bounds, same summation order: results are
// bit-identical to the level-by-level kernels), and writes the six planar maps.  Five dependent launches (2 x 7.8 us +
// 3 x 4.8 us: each one launch-latency bound) become one; the two smaller depth levels never visit HBM.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float pyrdown_px(const float* __restrict__ src /*LDS*/, int ldw, int ox, int oy, int x, int y, int sw, int sh) {
    const int tx = min(2 * x + 3, sw - 1);
    const int ty = min(2 * y + 3, sh - 1);
    float sum = 0.f;
    int count = 0;
    if (2 * x >= 2 && 2 * y >= 2 && tx == 2 * x + 3 && ty == 2 * y + 3) {
        // away from the image border the loops below are the full 5 x 5 window with the binomial weights in their natural order: the same taps in
        // the same order, unrolled and without a branch per tap -- a NaN tap adds +0 to the sum (which is never -0: it starts at +0 and every
        // addend is a depth >= 0 times a weight) and 0 to the count, i.e. nothing, as when it is skipped
        const float* p = src + (2 * y - 2 - oy) * ldw + (2 * x - 2 - ox);
#pragma unroll
        for (int j = 0; j < 5; ++j) {
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const float v = p[j * ldw + i];
                const float w = gauss5(4 - j) * gauss5(4 - i);     // compile-time constant
                const bool ok = !isnan(v);
                sum += ok ? v * w : 0.f;
                count += ok ? (int)w : 0;
            }
        }
        return sum / (float)count;
    }
    for (int cy = max(0, 2 * y - 2); cy < ty; ++cy) {
        for (int cx = max(0, 2 * x - 2); cx < tx; ++cx) {
            const float v = src[(cy - oy) * ldw + (cx - ox)];
            if (!isnan(v)) {
                const float w = gauss5(ty - cy - 1) * gauss5(tx - cx - 1);
                sum += v * w;
                count += (int)w;
            }
        }
    }
    return sum / (float)count;
}

__device__ __forceinline__ void vmap_nmap_px(const float* __restrict__ d /*LDS*/, int ldw, int ox, int oy, int u, int v, int W, int H,
                                             Intr k, float cutoff, float* __restrict__ vmap, float* __restrict__ nmap) {
    const int P = W * H, i = v * W + u;
    const float fx_inv = 1.f / k.fx, fy_inv = 1.f / k.fy;
    const float* p = d + (v - oy) * ldw + (u - ox);
    float3 v00, v01, v10;
    const bool ok00 = vertex_from_depth(p[0], u, v, k, fx_inv, fy_inv, cutoff, v00);
    vmap[i] = v00.x; vmap[P + i] = v00.y; vmap[2 * P + i] = v00.z;
    float3 n = f3(qnan(), qnan(), qnan());
    if (u < W - 1 && v < H - 1) {
        const bool ok01 = vertex_from_depth(p[1], u + 1, v, k, fx_inv, fy_inv, cutoff, v01);
        const bool ok10 = vertex_from_depth(p[ldw], u, v + 1, k, fx_inv, fy_inv, cutoff, v10);
        if (ok00 && ok01 && ok10) n = normalized_rsqrt(cross3(v01 - v00, v10 - v00));
    }
    nmap[i] = n.x; nmap[P + i] = n.y; nmap[2 * P + i] = n.z;
}

struct FramePyrArgs {
    const float* depth; int W, H; Intr k; float cutoff;
    float* vmap[3]; float* nmap[3];
};

constexpr int kFpT2 = 4;                                   // level-2 tile side of a workgroup (16x16 level-0 pixels)
constexpr int kFpL2 = kFpT2 + 1, kFpL1 = 2 * kFpT2 + 5, kFpL0 = 2 * kFpL1 + 3;   // 5, 13, 29 with their halos

__global__ __launch_bounds__(256) void k_frame_pyramid(const FramePyrArgs a) {
    __shared__ float s0[kFpL0 * kFpL0];
    __shared__ float s1[kFpL1 * kFpL1];
    __shared__ float s2[kFpL2 * kFpL2];
    const int W0 = a.W, H0 = a.H, W1 = W0 >> 1, H1 = H0 >> 1, W2 = W0 >> 2, H2 = H0 >> 2;
    const int tiles_x = (W2 + kFpT2 - 1) / kFpT2, tiles = tiles_x * ((H2 + kFpT2 - 1) / kFpT2);
    const int tile = xcd_contiguous_tile(blockIdx.x, tiles);     // XCD k works on the k-th band of tile rows (mf_device.h)
    if (tile >= tiles) return;
    const int X2 = (tile % tiles_x) * kFpT2, Y2 = (tile / tiles_x) * kFpT2;  // tile origin at level 2
    const int ox1 = 2 * X2 - 2, oy1 = 2 * Y2 - 2;                // LDS origins (may be negative)
    const int ox0 = 2 * ox1 - 2, oy0 = 2 * oy1 - 2;
    const int tid = threadIdx.x;
    for (int i = tid; i < kFpL0 * kFpL0; i += 256) {
        const int ly = i / kFpL0, lx = i - ly * kFpL0;
        const int gx = ox0 + lx, gy = oy0 + ly;
        s0[i] = (gx >= 0 && gx < W0 && gy >= 0 && gy < H0) ? a.depth[gy * W0 + gx] : qnan();
    }
    __syncthreads();
    // (each level's vertex / normal maps are written as soon as its depths stand in LDS: their stores drain under the next level's arithmetic)
    const Intr k0 = a.k;
    const Intr k1 = Intr{a.k.fx / 2.f, a.k.fy / 2.f, a.k.cx / 2.f, a.k.cy / 2.f};
    const Intr k2 = Intr{a.k.fx / 4.f, a.k.fy / 4.f, a.k.cx / 4.f, a.k.cy / 4.f};
    for (int l = tid; l < 16 * kFpT2 * kFpT2; l += 256) {
        const int u = 4 * X2 + l % (4 * kFpT2), v = 4 * Y2 + l / (4 * kFpT2);
        if (u < W0 && v < H0) vmap_nmap_px(s0, kFpL0, ox0, oy0, u, v, W0, H0, k0, a.cutoff, a.vmap[0], a.nmap[0]);
    }
    for (int i = tid; i < kFpL1 * kFpL1; i += 256) {
        const int ly = i / kFpL1, lx = i - ly * kFpL1;
        const int gx = ox1 + lx, gy = oy1 + ly;
        s1[i] = (gx >= 0 && gx < W1 && gy >= 0 && gy < H1) ? pyrdown_px(s0, kFpL0, ox0, oy0, gx, gy, W0, H0) : qnan();
    }
    __syncthreads();
    for (int l = tid; l < 4 * kFpT2 * kFpT2; l += 256) {
        const int u = 2 * X2 + l % (2 * kFpT2), v = 2 * Y2 + l / (2 * kFpT2);
        if (u < W1 && v < H1) vmap_nmap_px(s1, kFpL1, ox1, oy1, u, v, W1, H1, k1, a.cutoff, a.vmap[1], a.nmap[1]);
    }
    for (int i = tid; i < kFpL2 * kFpL2; i += 256) {
        const int ly = i / kFpL2, lx = i - ly * kFpL2;
        const int gx = X2 + lx, gy = Y2 + ly;
        s2[i] = (gx < W2 && gy < H2) ? pyrdown_px(s1, kFpL1, ox1, oy1, gx, gy, W1, H1) : qnan();
    }
    __syncthreads();
    for (int l = tid; l < kFpT2 * kFpT2; l += 256) {
        const int u = X2 + l % kFpT2, v = Y2 + l / kFpT2;
        if (u < W2 && v < H2) vmap_nmap_px(s2, kFpL2, X2, Y2, u, v, W2, H2, k2, a.cutoff, a.vmap[2], a.nmap[2]);
    }
}

void launch_frame_pyramid(const float* depth, float* const vmap[3], float* const nmap[3], int W, int H, Intr k, float cutoff,
                          hipStream_t s) {
    FramePyrArgs a;
    a.depth = depth; a.W = W; a.H = H; a.k = k; a.cutoff = cutoff;
    for (int i = 0; i < 3; ++i) { a.vmap[i] = vmap[i]; a.nmap[i] = nmap[i]; }
    const int tiles = (((W >> 2) + kFpT2 - 1) / kFpT2) * (((H >> 2) + kFpT2 - 1) / kFpT2);
    hipLaunchKernelGGL(k_frame_pyramid, dim3(xcd_padded_grid(tiles)), dim3(256), 0, s, a);
}

}  // namespace mf

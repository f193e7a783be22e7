// mf_bilateral_device.h -- the 13x13 bilateral depth filter's workgroup body (depth_bilateral_metric.frag:30-76), shared by k_bilateral
// (mf_preproc.hip) and the launch that runs it beside the model-side pyramid of the same frame (mf_odometry.hip: k_bilateral_model_pyramid).
// Every float operation of the body is rounded on its own (the pragma inside the function: the including file may allow contraction).
#pragma once
#include "mf_device.h"

namespace mf {

// ------------------------------------------------------------------------------------------------
// 13x13 bilateral.  One 256-thread workgroup (4 wavefronts, 64 lanes along x) filters a 64x4 tile staged through LDS
// with a 6-pixel halo: HBM/L2 traffic is 4 B out per pixel and the 169 taps come from LDS (row-contiguous ds_read_b32,
// conflict free).  The kernel is VALU/latency bound (169 exp per pixel), so the tile is kept small: 1200 workgroups
// give every SIMD 4-5 resident wavefronts to hide the dependent exp/fma chains (a 64x16 tile = 300 workgroups ran 4x
// slower at one wavefront per SIMD).
// ------------------------------------------------------------------------------------------------
constexpr int kBR = 6;
constexpr int kBTileW = 64, kBTileH = 4;
constexpr int kBLdsW = kBTileW + 2 * kBR;  // 76
constexpr int kBLdsH = kBTileH + 2 * kBR;  // 16
constexpr float kBOutside = 1e15f;

// The range/space weight exp(-(s2 * a + c2 * b)) is evaluated as 2^-(s2 * a' + c2 * b') with log2(e) folded into the
// constants and the hardware v_exp_f32 (1 ulp on 2^t; the argument carries |t| * 2^-24 <= 1e-6 relative for every weight
// above 1e-6).  libm's expf cost ~25 instructions x 169 taps and made the kernel VALU bound at 40 us.  The filter output
// is a weighted MEAN of nearly equal depths, so a 1e-6 relative weight error moves it by ~1e-9 relative: the measured
// difference to the oracle's expf path is a few ulp (tests/test_gpu_kernels.py::test_bilateral).
// (Two pixels per thread with 2-wide packed fp32 math -- 6.5 instead of 10 VALU instructions per tap -- was tried: 31 us
// against 19 us; half as many wavefronts left the exp / LDS latencies exposed.)
// tile: kBLdsH * kBLdsW floats of LDS; block: the workgroup's index in a grid of xcd_padded_grid(tiles) workgroups
__device__ __forceinline__ void bilateral_body(const float* __restrict__ depth, float* __restrict__ out, int W, int H, float* tile, int block) {
#pragma clang fp contract(off)
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int tiles_x = (W + kBTileW - 1) / kBTileW, tiles = tiles_x * ((H + kBTileH - 1) / kBTileH);
    const int tile_id = xcd_contiguous_tile(block, tiles);   // each XCD's L2 fetches its own band of the image (+ halo), not all of it
    if (tile_id >= tiles) return;
    const int x0 = (tile_id % tiles_x) * kBTileW, y0 = (tile_id / tiles_x) * kBTileH;
    // stage.  The shader clips its loops at the image border; here an out-of-image tap holds kBOutside = 1e15: its range term is
    // -8e32, 2^that is exactly 0, and it adds tmp * 0 = +0 to both sums -- the same bits as skipping it, without a compare, an exec
    // mask and a branch per tap (round 3: the 169 branches also kept every ds_read on its own s_waitcnt).
    for (int i = threadIdx.x; i < kBLdsH * kBLdsW; i += 256) {
        const int ly = i / kBLdsW, lx = i - ly * kBLdsW;
        const int gx = x0 + lx - kBR, gy = y0 + ly - kBR;
        tile[i] = (gx >= 0 && gx < W && gy >= 0 && gy < H) ? depth[gy * W + gx] : kBOutside;
    }
    __syncthreads();
    const float sigma_space2_inv_half = 0.024691358f * 1.44269504088896340736f;   // x log2(e)
    const float sigma_color2_inv_half = 555.556f * 1.44269504088896340736f;
    const int gx = x0 + tx, gy = y0 + ty;
    if (gx >= W || gy >= H) return;
    const float value = tile[(ty + kBR) * kBLdsW + tx + kBR];
    float res = 0.f;
    if (!(value <= 0.03f)) {   // the shader's gate as written (`if (value <= 0.03f) 0 else filter`, :34): a NaN centre is filtered, to NaN
        // {sum of tap * weight, sum of weights} as ONE two-lane value: the two additions of a tap are a single v_pk_add_f32 -- two independent IEEE
        // additions, the same bits -- whether or not the including file lets the SLP vectoriser find the pair (mf_odometry.hip does not)
        typedef float pair_t __attribute__((vector_size(8)));
        pair_t acc = {0.f, 0.f};
#pragma unroll
        for (int dy = -kBR; dy <= kBR; ++dy) {
            const float* row = &tile[(ty + kBR + dy) * kBLdsW + tx + kBR];
            const float fy2 = (float)(dy * dy);
#pragma unroll
            for (int dx = -kBR; dx <= kBR; ++dx) {
                const float tmp = row[dx];
                const float space_term = -(((float)(dx * dx) + fy2) * sigma_space2_inv_half);   // compile-time constant
                const float color2 = (value - tmp) * (value - tmp);
                const float weight = __builtin_amdgcn_exp2f(space_term - color2 * sigma_color2_inv_half);   // exactly 0 for an outside tap
                const pair_t add = {tmp * weight, weight};
                acc += add;
            }
        }
        res = acc[0] / acc[1];
    }
    out[gy * W + gx] = res;
}

inline int bilateral_grid(int W, int H) { return xcd_padded_grid(((W + kBTileW - 1) / kBTileW) * ((H + kBTileH - 1) / kBTileH)); }

}  // namespace mf

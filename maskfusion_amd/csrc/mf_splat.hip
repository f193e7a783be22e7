// mf_splat.hip -- tiled splat prediction (ModelProjection::combinedPredict, Core/Model/ModelProjection.cpp:187-268 +
// Core/Shaders/splat.vert, combo_splat.frag): the same raster rule and keys as k_splat_scatter / k_splat_resolve in
// mf_surfel.hip (which stay as the executable specification: tests compare the two bit for bit), organised so that the
// z-test happens in LDS.
//
// Why: the scatter form issues one 64-bit global atomicMin per covered pixel (4.3 M per VGA frame at 270 k surfels).
// Device-scope atomics are performed memory-side (rocprofv3: TCC_HIT ~ 0, every request a miss; a line touched by an
// atomic is dropped from L2, so a read-before-atomic filter costs as much as the atomic) and the kernel sat at 81 us,
// issue-stalled (SQ_WAIT_INST_ANY 68 % of wave cycles).  Here surfels are binned to 16x16-pixel tiles (two passes with
// LDS histograms: ~13 k global atomics instead of 4.3 M), one workgroup per tile runs the z-test with ds_min_u64 on a 2 KB
// LDS tile, and the winner's attributes are written straight to the prediction maps -- no key buffer, no resolve pass.
#pragma clang fp contract(off)

#include "mf_internal.h"
#include "mf_device.h"
#include "mf_rgbd_device.h"

namespace mf {

constexpr int kTile = 16;
constexpr int kBinThreads = 1024;
constexpr int kMaxTiles = 8192;          // bounds the LDS histograms (VGA: 1200 tiles, 1280x960: 4800); larger images use the scatter form

struct SplatSetup { float3 h, nrm; float sqrRad, pn; int px0, px1, py0, py1; };

// splat.vert:40-105 for surfel i; false if it draws nothing.  Verbatim the per-surfel part of k_splat_scatter.
__device__ __forceinline__ bool splat_setup(const Surfels& src, int i, float time, const float* Ri, float3 ti, int W, int H, Intr k,
                                            float maxDepth, float confThreshold, int timeDelta, SplatSetup& o) {
    const float4 pc = src.pc[i];
    if (pc.w < confThreshold) return false;
    const float lastTime = src.ct[i].w;
    const float3 h = mul33(Ri, f3(pc.x, pc.y, pc.z)) + ti;
    if (h.z > maxDepth || h.z < 0 || time - lastTime > (float)timeDelta || lastTime > time) return false;  // splat.vert:58
    const float u = ((k.fx * h.x) / h.z) + k.cx, v = ((k.fy * h.y) / h.z) + k.cy;
    if (!(u >= 0.f && u <= (float)W && v >= 0.f && v <= (float)H)) return false;
    const float4 n4 = src.nr[i];
    const float3 nrm = normalize_gl(mul33(Ri, f3(n4.x, n4.y, n4.z)));
    const float rad = n4.w;
    const float3 x1 = normalize_gl(f3(nrm.y - nrm.z, -nrm.x, nrm.x)) * (rad * 1.41421356f);
    const float3 y1 = cross3(nrm, x1);
    float xs0 = INFINITY, xs1 = -INFINITY, ys0 = INFINITY, ys1 = -INFINITY;
    const float3 corners[4] = {h + x1, h + y1, h - y1, h - x1};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float pxq = ((k.fx * corners[q].x) / corners[q].z) + k.cx;
        const float pyq = ((k.fy * corners[q].y) / corners[q].z) + k.cy;
        xs0 = fminf(xs0, pxq); xs1 = fmaxf(xs1, pxq);
        ys0 = fminf(ys0, pyq); ys1 = fmaxf(ys1, pyq);
    }
    float size = fmaxf(0.f, fmaxf(fabsf(xs1 - xs0), fabsf(ys1 - ys0)));
    if (!(size > 0.f)) return false;
    size = fminf(size, 64.0f);
    const float half = size * 0.5f;
    o.px0 = max(0, (int)ceilf(u - half - 0.5f)); o.px1 = min(W - 1, (int)ceilf(u + half - 0.5f) - 1);
    o.py0 = max(0, (int)ceilf(v - half - 0.5f)); o.py1 = min(H - 1, (int)ceilf(v + half - 0.5f) - 1);
    if (o.px0 > o.px1 || o.py0 > o.py1) return false;
    o.h = h; o.nrm = nrm; o.sqrRad = rad * rad; o.pn = dot3(h, nrm);
    return true;
}

struct BinArgs {
    Surfels src; const FrameDev* frame; const PoseDev* pose; int W, H; Intr k; float maxDepth, confThreshold; int timeDelta;
    int tilesX, tilesY;
    int* tile_count;      // [tiles]   zero on entry (each tile workgroup re-zeroes its own counter when it is done)
    int* tile_cursor;     // [tiles]   zeroed by pass 1, used by pass 2
    int* tile_base;       // [tiles+1] exclusive scan of tile_count, written by pass 2 for pass 3
    int* entries;         // [entries_cap]
    int entries_cap;
    short4* bbox;         // [surfels] px0, px1, py0, py1 (px0 > px1: culled)
    FrameDev* frame_rw;   // overflow flag (pad[1])
};

// Pass 1: per-surfel sprite box + per-tile counts (LDS histogram per 1024-thread workgroup, flushed with one global atomic
// per touched tile).
__global__ __launch_bounds__(kBinThreads) void k_splat_bin_count(const BinArgs a) {
    extern __shared__ int s_hist[];
    const int nt = a.tilesX * a.tilesY;
    for (int t = threadIdx.x; t < nt; t += kBinThreads) s_hist[t] = 0;
    if (blockIdx.x == 0)
        for (int t = threadIdx.x; t < nt; t += kBinThreads) a.tile_cursor[t] = 0;
    __syncthreads();
    const int n = a.frame->count;
    const float time = (float)a.frame->tick;
    float Ri[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) Ri[q] = a.pose->Ri[q];
    const float3 ti = f3(a.pose->ti[0], a.pose->ti[1], a.pose->ti[2]);
    for (int i = blockIdx.x * kBinThreads + threadIdx.x; i < n; i += gridDim.x * kBinThreads) {
        SplatSetup su;
        short4 bb = make_short4(1, 0, 1, 0);
        if (splat_setup(a.src, i, time, Ri, ti, a.W, a.H, a.k, a.maxDepth, a.confThreshold, a.timeDelta, su)) {
            bb = make_short4((short)su.px0, (short)su.px1, (short)su.py0, (short)su.py1);
            for (int ty = su.py0 / kTile; ty <= su.py1 / kTile; ++ty)
                for (int tx = su.px0 / kTile; tx <= su.px1 / kTile; ++tx) atomicAdd(&s_hist[ty * a.tilesX + tx], 1);
        }
        a.bbox[i] = bb;
    }
    __syncthreads();
    for (int t = threadIdx.x; t < nt; t += kBinThreads)
        if (s_hist[t]) atomicAdd(&a.tile_count[t], s_hist[t]);
}

// Exclusive scan of the tile counts by one workgroup (nt <= kMaxTiles) into LDS; returns nothing, fills s_base[0..nt].
__device__ __forceinline__ void scan_tiles(const int* __restrict__ counts, int nt, int* s_base, int* s_tmp, int nthreads) {
    // each thread sums a contiguous chunk, one warp-free serial scan over the (<= 1024) chunk sums by thread 0 is avoided by
    // a Hillis-Steele pass in LDS
    const int per = (nt + nthreads - 1) / nthreads;
    const int b = threadIdx.x * per, e = min(nt, b + per);
    int sum = 0;
    for (int t = b; t < e; ++t) sum += counts[t];
    s_tmp[threadIdx.x] = sum;
    __syncthreads();
    for (int off = 1; off < nthreads; off <<= 1) {
        const int v = (threadIdx.x >= off) ? s_tmp[threadIdx.x - off] : 0;
        __syncthreads();
        s_tmp[threadIdx.x] += v;
        __syncthreads();
    }
    int run = s_tmp[threadIdx.x] - sum;   // exclusive prefix of this thread's chunk
    for (int t = b; t < e; ++t) { s_base[t] = run; run += counts[t]; }
    if (threadIdx.x == nthreads - 1) s_base[nt] = s_tmp[nthreads - 1];
    __syncthreads();
}

// Pass 2: fill the per-tile surfel lists.  A workgroup counts its own entries per tile in LDS, reserves a contiguous
// range per touched tile with ONE global atomicAdd, then hands out slots inside the range with LDS atomics.
__global__ __launch_bounds__(kBinThreads) void k_splat_bin_fill(const BinArgs a) {
    extern __shared__ int s_mem[];
    const int nt = a.tilesX * a.tilesY;
    int* s_base = s_mem;                 // [nt + 1] exclusive scan of tile_count
    int* s_cnt = s_mem + (nt + 1);       // [nt] this workgroup's entries per tile, then its reserved start
    int* s_fill = s_cnt + nt;            // [nt] slots handed out
    int* s_tmp = s_fill + nt;            // [kBinThreads]
    scan_tiles(a.tile_count, nt, s_base, s_tmp, kBinThreads);
    if (blockIdx.x == 0)
        for (int t = threadIdx.x; t <= nt; t += kBinThreads) a.tile_base[t] = s_base[t];
    for (int t = threadIdx.x; t < nt; t += kBinThreads) { s_cnt[t] = 0; s_fill[t] = 0; }
    __syncthreads();
    const int n = a.frame->count;
    for (int i = blockIdx.x * kBinThreads + threadIdx.x; i < n; i += gridDim.x * kBinThreads) {
        const short4 bb = a.bbox[i];
        if (bb.x > bb.y) continue;
        for (int ty = bb.z / kTile; ty <= bb.w / kTile; ++ty)
            for (int tx = bb.x / kTile; tx <= bb.y / kTile; ++tx) atomicAdd(&s_cnt[ty * a.tilesX + tx], 1);
    }
    __syncthreads();
    for (int t = threadIdx.x; t < nt; t += kBinThreads)
        if (s_cnt[t]) s_cnt[t] = s_base[t] + atomicAdd(&a.tile_cursor[t], s_cnt[t]);
    __syncthreads();
    for (int i = blockIdx.x * kBinThreads + threadIdx.x; i < n; i += gridDim.x * kBinThreads) {
        const short4 bb = a.bbox[i];
        if (bb.x > bb.y) continue;
        for (int ty = bb.z / kTile; ty <= bb.w / kTile; ++ty)
            for (int tx = bb.x / kTile; tx <= bb.y / kTile; ++tx) {
                const int t = ty * a.tilesX + tx;
                const int slot = s_cnt[t] + atomicAdd(&s_fill[t], 1);
                if (slot < a.entries_cap) a.entries[slot] = i;
                else a.frame_rw->pad[1] = 1;   // list overflow: reported by mf_sync (never silently dropped)
            }
    }
}

struct TileArgs {
    Surfels src; FrameDev* frame; const PoseDev* pose; int W, H; Intr k; float maxDepth, confThreshold; int timeDelta;
    int tilesX, tilesY;
    int* tile_count; const int* tile_base; const int* entries; int entries_cap;
    float4* predV; float4* predN; uchar4* predImage; uint16_t* predTime;
    const uint8_t* rgb; uint8_t* predGray; uint8_t* fillGray;
};

// Pass 3: one 256-thread workgroup per 16x16 tile: LDS z-test over the tile's surfel list, then the fragment outputs
// (combo_splat.frag) of every pixel of the tile.
__global__ __launch_bounds__(256) void k_splat_tile(const TileArgs a) {
    __shared__ unsigned long long s_key[kTile * kTile];
    __shared__ float4 s_ray[kTile * kTile];
    __shared__ int s_range[2];
    const int tile = blockIdx.x;
    const int tx0 = (tile % a.tilesX) * kTile, ty0 = (tile / a.tilesX) * kTile;
    s_key[threadIdx.x] = kEmptyKey;
    {   // the viewing ray of every pixel of the tile, once (combo_splat.frag:40-42); the per-surfel loops below re-used to
        // spend two thirds of their instructions recomputing it (two divisions + a normalisation per covered pixel)
        const float fcx = (float)(tx0 + (threadIdx.x & (kTile - 1))) + 0.5f, fcy = (float)(ty0 + (threadIdx.x >> 4)) + 0.5f;
        const float3 l = normalize_gl(f3((fcx - a.k.cx) / a.k.fx, (fcy - a.k.cy) / a.k.fy, 1.0f));
        s_ray[threadIdx.x] = make_float4(l.x, l.y, l.z, 0.f);
    }
    if (threadIdx.x == 0) {
        s_range[0] = a.tile_base[tile];
        s_range[1] = a.tile_count[tile];
        a.tile_count[tile] = 0;   // consumed: pass 1 of the next prediction starts from zero
    }
    __syncthreads();
    const int beg = s_range[0], cnt = min(s_range[1], max(0, a.entries_cap - s_range[0]));
    const float time = (float)a.frame->tick;
    float Ri[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) Ri[q] = a.pose->Ri[q];
    const float3 ti = f3(a.pose->ti[0], a.pose->ti[1], a.pose->ti[2]);
    const Intr k = a.k;
    for (int e = threadIdx.x; e < cnt; e += 256) {
        const int i = a.entries[beg + e];
        SplatSetup su;
        if (!splat_setup(a.src, i, time, Ri, ti, a.W, a.H, k, a.maxDepth, a.confThreshold, a.timeDelta, su)) continue;
        const int x0 = max(su.px0, tx0), x1 = min(su.px1, tx0 + kTile - 1);
        const int y0 = max(su.py0, ty0), y1 = min(su.py1, ty0 + kTile - 1);
        for (int py = y0; py <= y1; ++py) {
            for (int px = x0; px <= x1; ++px) {
                const int lp = (py - ty0) * kTile + (px - tx0);
                const float4 r4 = s_ray[lp];
                const float3 l = f3(r4.x, r4.y, r4.z);
                const float3 cp = l * (su.pn / dot3(l, su.nrm));
                const float3 diff = cp - su.h;
                if (!(dot3(diff, diff) <= su.sqrRad)) continue;
                if (!(cp.z > 0.f)) continue;
                const unsigned long long key = ((unsigned long long)__float_as_uint(cp.z) << 32) | (unsigned)i;
                atomicMin(&s_key[lp], key);
            }
        }
    }
    __syncthreads();
    const int px = tx0 + (threadIdx.x & (kTile - 1)), py = ty0 + (threadIdx.x >> 4);
    if (px >= a.W || py >= a.H) return;
    const int p = py * a.W + px;
    const unsigned long long key = s_key[threadIdx.x];
    if (key == kEmptyKey) {
        a.predV[p] = a.predN[p] = make_float4(0, 0, 0, 0);
        a.predImage[p] = make_uchar4(0, 0, 0, 0);
        a.predTime[p] = 0;
        if (a.predGray) a.predGray[p] = 0;
        if (a.fillGray && a.rgb) a.fillGray[p] = intensity_of((float)a.rgb[p * 3], (float)a.rgb[p * 3 + 1], (float)a.rgb[p * 3 + 2]);
        return;
    }
    const int i = (int)(unsigned)(key & 0xFFFFFFFFull);
    const float z = __uint_as_float((unsigned)(key >> 32));
    const float4 pc = a.src.pc[i], c4 = a.src.ct[i], n4 = a.src.nr[i];
    const float3 n = normalize_gl(mul33(a.pose->Ri, f3(n4.x, n4.y, n4.z)));
    const float fcx = (float)px + 0.5f, fcy = (float)py + 0.5f;
    a.predV[p] = make_float4((fcx - k.cx) * z * (1.f / k.fx), (fcy - k.cy) * z * (1.f / k.fy), z, pc.w);  // combo_splat.frag:56
    a.predN[p] = make_float4(n.x, n.y, n.z, n4.w);
    const int ci = (int)c4.x;
    const uchar4 col = make_uchar4((ci >> 16) & 0xFF, (ci >> 8) & 0xFF, ci & 0xFF, 255);
    a.predImage[p] = col;
    a.predTime[p] = (uint16_t)(unsigned)c4.z;
    if (a.predGray || a.fillGray) {
        const uint8_t gv = intensity_of((float)col.x, (float)col.y, (float)col.z);
        if (a.predGray) a.predGray[p] = gv;
        if (a.fillGray) {
            const bool empty = col.x == 0 && col.y == 0 && col.z == 0;
            a.fillGray[p] = (empty && a.rgb) ? intensity_of((float)a.rgb[p * 3], (float)a.rgb[p * 3 + 1], (float)a.rgb[p * 3 + 2]) : gv;
        }
    }
    // MaskFusion::requiresFillIn (MaskFusion.cpp:630-648): nearest sample of the 20x down-sampled colour prediction
    if ((px % 20) == 10 && (py % 20) == 10 && px / 20 < a.W / 20 && py / 20 < a.H / 20 && col.x > 0 && col.y > 0 && col.z > 0)
        atomicAdd(&a.frame->cover, 1);
}

size_t splat_tiles_scratch_ints(int W, int H) { return (size_t)((W + kTile - 1) / kTile) * ((H + kTile - 1) / kTile); }

int launch_splat_tiled(Surfels src, FrameDev* frame, const PoseDev* pose, int W, int H, Intr k, float maxDepth, float confThreshold,
                       int timeDelta, int* tile_count, int* tile_cursor, int* tile_base, int* entries, int entries_cap, void* bbox,
                       float4* predV,
                       float4* predN, uchar4* predImage, uint16_t* predTime, const uint8_t* rgb, uint8_t* predGray, uint8_t* fillGray,
                       hipStream_t s) {
    const int tilesX = (W + kTile - 1) / kTile, tilesY = (H + kTile - 1) / kTile, nt = tilesX * tilesY;
    if (nt > kMaxTiles) return -1;
    BinArgs b;
    b.src = src; b.frame = frame; b.pose = pose; b.W = W; b.H = H; b.k = k; b.maxDepth = maxDepth; b.confThreshold = confThreshold;
    b.timeDelta = timeDelta; b.tilesX = tilesX; b.tilesY = tilesY; b.tile_count = tile_count; b.tile_cursor = tile_cursor;
    b.tile_base = tile_base; b.entries = entries; b.entries_cap = entries_cap; b.bbox = reinterpret_cast<short4*>(bbox); b.frame_rw = frame;
    const int nblocks = 256;
    hipLaunchKernelGGL(k_splat_bin_count, dim3(nblocks), dim3(kBinThreads), (size_t)nt * sizeof(int), s, b);
    hipLaunchKernelGGL(k_splat_bin_fill, dim3(nblocks), dim3(kBinThreads), (size_t)(3 * nt + 1 + kBinThreads) * sizeof(int), s, b);
    TileArgs t;
    t.src = src; t.frame = frame; t.pose = pose; t.W = W; t.H = H; t.k = k; t.maxDepth = maxDepth; t.confThreshold = confThreshold;
    t.timeDelta = timeDelta; t.tilesX = tilesX; t.tilesY = tilesY; t.tile_count = tile_count; t.tile_base = tile_base;
    t.entries = entries; t.entries_cap = entries_cap;
    t.predV = predV; t.predN = predN; t.predImage = predImage; t.predTime = predTime; t.rgb = rgb; t.predGray = predGray;
    t.fillGray = fillGray;
    hipLaunchKernelGGL(k_splat_tile, dim3(nt), dim3(256), 0, s, t);
    return 0;
}

}  // namespace mf

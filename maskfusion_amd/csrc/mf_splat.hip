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

constexpr int kTile = 16;                // tile width in pixels
constexpr int kTileHMax = 32;            // tile height: 24 (default since round 6), 16, 20 or 32 pixels ("tileHeight", SplatTuning::tile_h) -- a launch parameter.
                                         // 1 200 workgroups of 16 x 16 tiles at VGA are more than the 1 024 that are resident at once; 800 of 16 x 24 are one
                                         // round: prediction stage 53.0 -> 50.8 us at VGA, the 1280 x 960 frame 861 -> 848 us (profiles/r06zj_ab.txt); 16 x 32: 52.6 us at VGA, 843 us at
                                         // 1280 x 960, configs[4] unchanged (r06zk_ab.txt)
constexpr int kBinThreads = 1024;
constexpr int kMaxTiles = 8192;          // bounds the LDS histograms (VGA: 1200 tiles, 1280x960: 4800); larger images use the scatter form

struct SplatSetup { float3 h, nrm; float sqrRad, pn; int px0, px1, py0, py1; };
struct TileStamps { unsigned long long t[8]; };   // "splatProfile": shader-clock stamps of a tile workgroup's thread 0 (by reference + constant
                                                  // indices: an array handed over as a pointer lives in scratch memory, for every launch)

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
    // gl_PointSize is clamped to the implementation's point size range (OpenGL 3.3 core, 3.4 "Points"; NVIDIA: [1, 2047]): a sprite
    // smaller than a pixel -- a surfel seen from more than ~4x its creation distance -- still covers the pixel its centre falls into
    size = fminf(fmaxf(size, 1.0f), 64.0f);
    const float half = size * 0.5f;
    o.px0 = max(0, (int)ceilf(u - half - 0.5f)); o.px1 = min(W - 1, (int)ceilf(u + half - 0.5f) - 1);
    o.py0 = max(0, (int)ceilf(v - half - 0.5f)); o.py1 = min(H - 1, (int)ceilf(v + half - 0.5f) - 1);
    if (o.px0 > o.px1 || o.py0 > o.py1) return false;
    o.h = h; o.nrm = nrm; o.sqrRad = rad * rad; o.pn = dot3(h, nrm);
    return true;
}

// lanes that share one sprite in the tile z-test (1, 2, 4, 8, 16; mf_set_param "spriteLanes"): an A/B switch for measurements, per context
// (SplatTuning travels with the launch; rounds 2-3 kept both knobs in process-wide statics that one context's setting changed for all)
int splat_sprite_lanes(int n) { return (n == 1 || n == 2 || n == 8 || n == 16) ? n : 4; }
// threads per tile workgroup (256, 512, 1024; "tileThreads"): a tile has 256 pixels, the threads beyond them only walk the sprite list.
// 1200 tiles of 256 threads are 4.7 wavefronts per SIMD on 256 CUs -- too few to hide the list's dependent gathers and the division
// chain of the pixel test (tools/splat_prof.py: a tile workgroup lives ~32 k cycles, ~4 k of them issuing)
// Measured (profiles/r03k_*, prediction stage incl. binning): 256 threads 62.6 us, 512 threads 58.0 us, 1024 threads 65.9 us (4 lanes per
// sprite each) -- 512 is the default.
int splat_tile_threads(int n) { return (n == 256 || n == 320 || n == 384 || n == 1024) ? n : 512; }
// tile height ("tileHeight"): 16, 20 or 24 rows of 16 pixels; the tile's pixels need a thread each
int splat_tile_height(int h) { return (h == 16 || h == 20 || h == 32) ? h : 24; }

struct BinArgs {
    Surfels src; const FrameDev* frame; const PoseDev* pose; int W, H; Intr k; float maxDepth, confThreshold; int timeDelta;
    int tilesX, tilesY, tileH;
    int* tile_count;      // [tiles]  zero on entry (each tile workgroup re-zeroes its own counter when it is done)
    int* entries;         // [tiles][tile_cap]
    int tile_cap;
    FrameDev* frame_rw;   // overflow flag (pad[1])
    float4* rec0; float4* rec1; short4* bbox;   // [surfels] sprite set-up kept for the tile pass: {h, r^2}, {n, h.n}, pixel box
    const int* vis_list; const int* vis_count;  // nullptr: every surfel; else the runs of src (Surfels::box) k_cull listed -- no other run can draw
};

// Binning in one pass: every tile owns a fixed slice of `entries`, so no scan is needed.  A 1024-thread workgroup counts
// its entries per tile in LDS, reserves a contiguous range per touched tile with ONE global atomicAdd (~13 k per frame
// instead of one global atomic per covered pixel), then hands out slots inside the range with LDS atomics.  The order of
// a tile's list is not deterministic; the z-test that consumes it is order independent.
__global__ __launch_bounds__(kBinThreads) void k_splat_bin(const BinArgs a) {
    extern __shared__ int s_mem[];
    const int nt = a.tilesX * a.tilesY;
    int* s_cnt = s_mem;           // [nt] this workgroup's entries per tile, then the start of its reserved range
    int* s_fill = s_mem + nt;     // [nt] slots handed out
    const int n = a.frame->phys;      // device-resident: the grid is fixed and walks the buffer in chunks of 2048 slots
    // by runs: the listed ones (k_cull), or every run of the buffer's table; a dense buffer without a table: slot by slot
    const int table_runs = a.frame->runs;
    const bool by_runs = a.vis_list != nullptr || table_runs > 0;
    const int nruns = a.vis_list ? *a.vis_count : table_runs;
    const float time = (float)a.frame->tick;
    float Ri[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) Ri[q] = a.pose->Ri[q];
    const float3 ti = f3(a.pose->ti[0], a.pose->ti[1], a.pose->ti[2]);
  // a round = 2048 surfel slots: 2048 consecutive surfels of the buffer, or -- with a visibility list -- the next 2048 / kRun listed runs
  constexpr int kRunsPerRound = 2 * kBinThreads / kRun;
  const int rounds = by_runs ? (nruns + kRunsPerRound - 1) / kRunsPerRound : (n + 2 * kBinThreads - 1) / (2 * kBinThreads);
  for (int chunk = blockIdx.x; chunk < rounds; chunk += gridDim.x) {
    for (int t = threadIdx.x; t < nt; t += kBinThreads) { s_cnt[t] = 0; s_fill[t] = 0; }
    __syncthreads();
    // 2 surfels per thread; their sprite boxes stay in registers between the two phases
    int idx[2]; short4 bb[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        int i = (chunk * 2 + r) * kBinThreads + threadIdx.x;
        if (by_runs) {
            const int slot = r * kBinThreads + (int)threadIdx.x, v = chunk * kRunsPerRound + slot / kRun;
            i = n;
            if (v < nruns) {
                const int run = a.vis_list ? a.vis_list[v] : v;
                const int beg = run_start(a.src.box, run), len = run_len(a.src.box, run);
                if (slot % kRun < len) i = beg + slot % kRun;
            }
        }
        idx[r] = i;
        bb[r] = make_short4(1, 0, 1, 0);
        if (i < n) {
            SplatSetup su;
            if (splat_setup(a.src, i, time, Ri, ti, a.W, a.H, a.k, a.maxDepth, a.confThreshold, a.timeDelta, su)) {
                bb[r] = make_short4((short)su.px0, (short)su.px1, (short)su.py0, (short)su.py1);
                // the tile pass meets this surfel ~1.8 times (once per overlapped tile): it reads the set-up instead of
                // repeating its ~300 instructions (ten IEEE divisions) each time
                a.rec0[i] = make_float4(su.h.x, su.h.y, su.h.z, su.sqrRad);
                a.rec1[i] = make_float4(su.nrm.x, su.nrm.y, su.nrm.z, su.pn);
                for (int ty = su.py0 / a.tileH; ty <= su.py1 / a.tileH; ++ty)
                    for (int tx = su.px0 / kTile; tx <= su.px1 / kTile; ++tx) atomicAdd(&s_cnt[ty * a.tilesX + tx], 1);
            }
            a.bbox[i] = bb[r];   // also for surfels that draw nothing (empty box): the overflow path of the tile pass scans every box
        }
    }
    __syncthreads();
    for (int t = threadIdx.x; t < nt; t += kBinThreads)
        if (s_cnt[t]) s_cnt[t] = atomicAdd(&a.tile_count[t], s_cnt[t]);
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        if (bb[r].x > bb[r].y) continue;
        for (int ty = bb[r].z / a.tileH; ty <= bb[r].w / a.tileH; ++ty)
            for (int tx = bb[r].x / kTile; tx <= bb[r].y / kTile; ++tx) {
                const int t = ty * a.tilesX + tx;
                const int slot = s_cnt[t] + atomicAdd(&s_fill[t], 1);
                // a full list is not an error and drops nothing: tile_count keeps counting, and a tile whose count exceeds its
                // slice scans the sprite boxes of the whole map instead of its list (k_splat_tile); pad[1] only records that it happened
                if (slot < a.tile_cap) a.entries[(size_t)t * a.tile_cap + slot] = idx[r];
                else a.frame_rw->pad[1] = 1;
            }
    }
    __syncthreads();
  }
}

struct TileArgs {
    Surfels src; FrameDev* frame; const PoseDev* pose; int W, H; Intr k; float maxDepth, confThreshold; int timeDelta;
    int tilesX, tilesY, tileH;
    int* tile_count; const int* entries; int tile_cap;
    const float4* rec0; const float4* rec1; const short4* bbox;
    float4* predV; float4* predN; uchar4* predImage; uint16_t* predTime;
    const uint8_t* rgb; uint8_t* predGray; uint8_t* fillGray;
    int fillPassthrough;             // fill_rgb.frag's `passthrough` (frameToFrameRGB, Model.cpp:981): the fill-in image is the raw frame everywhere
    const int* vis_list; const int* vis_count;   // as BinArgs
    int advance; FrameAdvance adv;   // advance != 0: the last workgroup also runs the end-of-frame bookkeeping (k_frame_advance)
    unsigned long long* prof;        // optional ("splatProfile"): [tiles][8] shader-clock stamps of thread 0 + the tile's list length
};

// The z-test of one 16x16 tile in LDS, shared by the prediction (payload = surfel index) and the global projection (payload =
// model order / id): rays of the tile's pixels, the tile's sprite list (or, after a list overflow, every sprite box of the map),
// ds_min_u64 per covered pixel.  On return s_key[pixel of the tile] holds the winning key (all threads have passed a barrier).
template <bool kIndexPayload, int kSpriteLanes>
__device__ __forceinline__ void tile_ztest(int tile, int tilesX, int tileH, Intr k, int* tile_count, const int* entries, int tile_cap,
                                           const FrameDev* frame, const float4* __restrict__ rec0, const float4* __restrict__ rec1,
                                           const short4* __restrict__ bbox, unsigned payload, unsigned long long* s_key, float4* s_ray,
                                           int* s_range, bool prof, TileStamps& stamp, const int* __restrict__ vis_list,
                                           const int* __restrict__ vis_count, const int4* __restrict__ runs) {
    const int tx0 = (tile % tilesX) * kTile, ty0 = (tile / tilesX) * tileH;
    if ((int)threadIdx.x < kTile * tileH) {
    s_key[threadIdx.x] = kEmptyKey;
    {   // the viewing ray of every pixel of the tile, once (combo_splat.frag:40-42); the per-surfel loops below re-used to
        // spend two thirds of their instructions recomputing it (two divisions + a normalisation per covered pixel)
        const float fcx = (float)(tx0 + (threadIdx.x & (kTile - 1))) + 0.5f, fcy = (float)(ty0 + (threadIdx.x >> 4)) + 0.5f;
        const float3 l = normalize_gl(f3((fcx - k.cx) / k.fx, (fcy - k.cy) / k.fy, 1.0f));
        s_ray[threadIdx.x] = make_float4(l.x, l.y, l.z, 0.f);
    }
    }
    if (threadIdx.x == 0) {
        s_range[0] = tile_count[tile];
        tile_count[tile] = 0;   // consumed: the next binning pass starts from zero
    }
    __syncthreads();
    if (prof) { stamp.t[1] = __builtin_amdgcn_s_memtime(); stamp.t[6] = (unsigned long long)s_range[0]; }
    const bool overflow = s_range[0] > tile_cap;   // more sprites than list slots (the reference has no such limit): scan every box
    // (after an overflow: every surfel whose sprite box the binning pass wrote -- the runs of the visibility list, every run of the table, or
    // -- a dense buffer without a table -- every slot)
    // (looked at after an overflow only: the common path waits for nothing but its list)
    const int table_runs = overflow ? frame->runs : 0;
    const bool by_runs = vis_list != nullptr || table_runs > 0;
    const int cnt = overflow ? (by_runs ? (vis_list ? *vis_count : table_runs) * kRun : frame->count) : s_range[0];
    const int* __restrict__ list = entries + (size_t)tile * tile_cap;
    auto entry = [&](int e) -> int {
        if (!overflow) return list[e];
        if (!by_runs) return e;
        const int run = vis_list ? vis_list[e / kRun] : e / kRun;
        return e % kRun < run_len(runs, run) ? run_start(runs, run) + e % kRun : -1;
    };
    // kSpriteLanes neighbouring lanes share one sprite and take every kSpriteLanes-th pixel of its clipped box (with one lane per sprite a
    // wavefront runs as long as its largest box).  The lanes read the same list entry / records (one request) and the z-test is order
    // independent, so the keys are the same bits.  Measured on MI355X (profiles/r03h_*): prediction stage 65.5 us with 1 lane,
    // 62.2 / 62.6 with 2 / 4, 66.3 with 8, 81.4 with 16 -- the boxes are small (4.3 px a side on the bench stream), and the pass is half
    // VALU (two IEEE divisions' worth per pixel test), half gather latency; 4 is the default.
    constexpr int kLX = kSpriteLanes >= 8 ? 4 : (kSpriteLanes >= 2 ? 2 : 1), kLY = kSpriteLanes / kLX;
    const int sub = threadIdx.x & (kSpriteLanes - 1), sx = sub % kLX, sy = sub / kLX;
    // Software pipeline over the list (round 4).  An entry costs three DEPENDENT gathers -- list[e] -> bbox[i] -> rec0[i], rec1[i] -- and the
    // in-kernel stamps of round 3 (tools/splat_prof.py) showed a tile workgroup issuing for ~4 k of its ~32 k cycles: it waits.  Every round
    // now first issues the loads of LATER rounds -- the list entry two rounds ahead, the box and the two records of the next round's sprite
    // (every listed sprite overlaps the tile, so its records are always needed) -- and only then tests the sprite whose data arrived during
    // the previous round: one memory latency per round, overlapped with the pixel tests, instead of three in a row.  Same keys, bit for bit.
    const int stride = (int)blockDim.x / kSpriteLanes;   // blockDim.x = 256, 512 or 1024 ("tileThreads")
    const int e0 = threadIdx.x / kSpriteLanes;
    const short4 kNoBox = make_short4(1, 0, 1, 0);
    int i0 = e0 < cnt ? entry(e0) : -1;
    int i1 = e0 + stride < cnt ? entry(e0 + stride) : -1;
    short4 bb0 = i0 >= 0 ? bbox[i0] : kNoBox;
    float4 r0 = i0 >= 0 ? rec0[i0] : make_float4(0, 0, 0, 0), r1 = i0 >= 0 ? rec1[i0] : make_float4(0, 0, 0, 0);
    for (int e = e0; e < cnt; e += stride) {
        const int i = i0;
        const short4 bb = bb0;
        const float4 c0 = r0, c1 = r1;
        // loads of the rounds to come (nothing below depends on them)
        const int e2 = e + 2 * stride;
        const int i2 = e2 < cnt ? entry(e2) : -1;
        if (i1 >= 0) { bb0 = bbox[i1]; r0 = rec0[i1]; r1 = rec1[i1]; } else bb0 = kNoBox;
        i0 = i1; i1 = i2;
        const int x0 = max((int)bb.x, tx0), x1 = min((int)bb.y, tx0 + kTile - 1);
        const int y0 = max((int)bb.z, ty0), y1 = min((int)bb.w, ty0 + tileH - 1);
        if (x0 > x1 || y0 > y1) continue;
        SplatSetup su;
        su.h = f3(c0.x, c0.y, c0.z); su.sqrRad = c0.w; su.nrm = f3(c1.x, c1.y, c1.z); su.pn = c1.w;
        // the kSpriteLanes lanes of a sprite form a kLX x kLY block that strides over the box: two plain nested loops (a flat pixel counter
        // needed a wrap-around loop per pixel), and the next pixel's ray is on its way from LDS while this one is tested
        for (int py = y0 + sy; py <= y1; py += kLY) {
            const int row = (py - ty0) * kTile - tx0;
            int px = x0 + sx;
            float4 nxt = s_ray[row + min(px, x1)];
            for (; px <= x1; px += kLX) {
                const int lp = row + px;
                const float4 r4 = nxt;
                nxt = s_ray[row + min(px + kLX, x1)];
                const float3 l = f3(r4.x, r4.y, r4.z);
                const float3 cp = l * (su.pn / dot3(l, su.nrm));
                const float3 diff = cp - su.h;
                if (!(dot3(diff, diff) <= su.sqrRad)) continue;
                if (!(cp.z > 0.f)) continue;
                const unsigned long long key = ((unsigned long long)__float_as_uint(cp.z) << 32) | (kIndexPayload ? (unsigned)i : payload);
                atomicMin(&s_key[lp], key);   // (looking at s_key before the atomic -- most sprites lose -- was measured: no gain)
            }
        }
    }
    if (prof) stamp.t[2] = __builtin_amdgcn_s_memtime();
    __syncthreads();
    if (prof) stamp.t[3] = __builtin_amdgcn_s_memtime();
}

// Pass 3: one workgroup (512 threads by default, the first 256 own the pixels) per 16x16 tile: LDS z-test over the tile's surfel list, then the fragment outputs
// (combo_splat.frag) of every pixel of the tile.
template <int kSpriteLanes>
__global__ __launch_bounds__(1024) void k_splat_tile(const TileArgs a) {
    __shared__ unsigned long long s_key[kTile * kTileHMax];
    __shared__ float4 s_ray[kTile * kTileHMax];
    __shared__ int s_range[1];
    __shared__ int s_cover;
    // XCD k draws the k-th contiguous eighth of the tile list (mf_device.h): a sprite overlaps 1.8 tiles on average, and neighbouring
    // tiles now find its records in the same L2.  The grid is padded to a multiple of 8; a padding workgroup draws nothing but still
    // takes its ticket below.
    const int tile = xcd_contiguous_tile(blockIdx.x, a.tilesX * a.tilesY);
    const bool live = tile < a.tilesX * a.tilesY;
    const int tx0 = (tile % a.tilesX) * kTile, ty0 = (tile / a.tilesX) * a.tileH;
    const Intr k = a.k;
    if (threadIdx.x == 0) s_cover = 0;   // (ordered before its use by the barriers inside tile_ztest)
    TileStamps stamp;
    const bool prof = a.prof != nullptr && live && threadIdx.x == 0;
    if (prof) {
#pragma unroll
        for (int q = 0; q < 8; ++q) stamp.t[q] = 0;
        stamp.t[0] = __builtin_amdgcn_s_memtime();
    }
    if (live) tile_ztest<true, kSpriteLanes>(tile, a.tilesX, a.tileH, k, a.tile_count, a.entries, a.tile_cap, a.frame, a.rec0, a.rec1, a.bbox, 0u, s_key, s_ray, s_range,
                                            prof, stamp, a.vis_list, a.vis_count, a.src.box);
    const int px = tx0 + (threadIdx.x & (kTile - 1)), py = ty0 + (threadIdx.x >> 4);
    int covered = 0;   // this pixel is one of the 20x down-sampled samples of MaskFusion::requiresFillIn and carries a colour
    // (gathering the winners' records along the tile's COLUMNS and transposing the outputs through the LDS -- what made the index map's resolve
    // twice as fast -- changes nothing here: 61.3 against 60.0 us for the stage, profiles/r05r_ab.txt; a sprite covers ~4 x 4 pixels, neighbouring
    // pixels share their winner either way)
    if (live && (int)threadIdx.x < kTile * a.tileH && px < a.W && py < a.H) {   // (threads beyond the tile's 256 pixels only helped with the list)
      const int p = py * a.W + px;
      const unsigned long long key = s_key[threadIdx.x];
      if (key == kEmptyKey) {
        a.predV[p] = a.predN[p] = make_float4(0, 0, 0, 0);
        a.predImage[p] = make_uchar4(0, 0, 0, 0);
        a.predTime[p] = 0;
        if (a.predGray) a.predGray[p] = 0;
        if (a.fillGray && a.rgb) a.fillGray[p] = intensity_of((float)a.rgb[p * 3], (float)a.rgb[p * 3 + 1], (float)a.rgb[p * 3 + 2]);
      } else {
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
                const bool empty = (col.x == 0 && col.y == 0 && col.z == 0) || a.fillPassthrough != 0;
                a.fillGray[p] = (empty && a.rgb) ? intensity_of((float)a.rgb[p * 3], (float)a.rgb[p * 3 + 1], (float)a.rgb[p * 3 + 2]) : gv;
            }
        }
        // MaskFusion::requiresFillIn (MaskFusion.cpp:630-648): nearest sample of the 20x down-sampled colour prediction
        covered = ((px % 20) == 10 && (py % 20) == 10 && px / 20 < a.W / 20 && py / 20 < a.H / 20 && col.x > 0 && col.y > 0 && col.z > 0) ? 1 : 0;
      }
    }
    // One 64-bit atomic per workgroup carries both its coverage count and its "finished" ticket (the count travels IN the atomic, so
    // the workgroup that draws the last ticket holds the launch's total without any ordering between workgroups).  That workgroup
    // hands the total on: to frame->cover for a stand-alone prediction, or straight into the end-of-frame bookkeeping that used to be
    // its own single-thread launch (k_frame_advance, ~4.7 us per model and frame).  Nothing else in this launch reads what it writes.
    if (prof) {
        stamp.t[4] = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamp.t[5] = __builtin_amdgcn_s_memtime();
#pragma unroll
        for (int q = 0; q < 8; ++q) a.prof[(size_t)tile * 8 + q] = stamp.t[q];
    }
    if (covered) atomicAdd(&s_cover, 1);   // at most two such pixels per 16x16 tile
    __syncthreads();
    if (threadIdx.x == 0) {
        const int wg_cover = s_cover;
        const unsigned long long old = atomicAdd(&a.frame->done_cover, (1ull << 32) | (unsigned long long)(unsigned)wg_cover);
        if ((unsigned)(old >> 32) == gridDim.x - 1u) {
            const int cover = (int)(unsigned)(old & 0xFFFFFFFFull) + wg_cover;
            FrameDev* f = a.frame;
            f->done_cover = 0ull;
            if (a.advance) {
                if (a.adv.log_slot) pose_log_entry(a.pose, a.adv.bg_pose, a.adv.log_slot);
                const int rw = a.W / 20, rh = a.H / 20;
                f->pad[0] = f->useFillIn;   // decision the tracking step of THIS frame ran with (mf_get_last_fillin)
                f->useFillIn = ((float)(f->cover + cover) / (float)(rw * rh) < 0.75f) ? 1 : 0;
                f->cover = 0;
                f->tick += 1;
                MF_FRAME_BBOX_ADVANCE(f);
                if (a.adv.host_mirror) *a.adv.host_mirror = *f;
            } else {
                f->cover += cover;
            }
        }
    }
}

// GlobalProjection (Core/Model/GlobalProjection.cpp:43-114, splat_models.vert / combo_splat_models.frag) of ONE model through the
// same tile lists: the z-test winner of every pixel of the tile is merged into the key image with a plain read-modify-write --
// a pixel belongs to exactly one workgroup of this launch, and the other models' launches are ordered on the stream.  Same keys
// as k_global_scatter (mf_segment.hip), which stays for small (object) models: one global atomic per covered pixel is memory-side
// work (see the header) and cost ~170 us per frame for a 250 k-surfel background model against ~25 us here.
struct GlobalTileArgs {
    const FrameDev* frame; int W, H; Intr k; int tilesX, tilesY, tileH;
    int* tile_count; const int* entries; int tile_cap;
    const float4* rec0; const float4* rec1; const short4* bbox;
    unsigned payload; unsigned long long* keys;
    const int* vis_list; const int* vis_count; const int4* runs;   // as BinArgs (runs = src.box)
};
template <int kSpriteLanes>
__global__ __launch_bounds__(1024) void k_global_tile(const GlobalTileArgs a) {
    __shared__ unsigned long long s_key[kTile * kTileHMax];
    __shared__ float4 s_ray[kTile * kTileHMax];
    __shared__ int s_range[1];
    const int tile = xcd_contiguous_tile(blockIdx.x, a.tilesX * a.tilesY);
    if (tile >= a.tilesX * a.tilesY) return;
    TileStamps unused;
    tile_ztest<false, kSpriteLanes>(tile, a.tilesX, a.tileH, a.k, a.tile_count, a.entries, a.tile_cap, a.frame, a.rec0, a.rec1, a.bbox, a.payload, s_key, s_ray, s_range,
                                    false, unused, a.vis_list, a.vis_count, a.runs);
    const int px = (tile % a.tilesX) * kTile + (threadIdx.x & (kTile - 1)), py = (tile / a.tilesX) * a.tileH + (threadIdx.x >> 4);
    if ((int)threadIdx.x >= kTile * a.tileH || px >= a.W || py >= a.H) return;
    const unsigned long long key = s_key[threadIdx.x];
    if (key == kEmptyKey) return;
    const int p = py * a.W + px;
    if (key < a.keys[p]) a.keys[p] = key;
}

int launch_global_tiled(Surfels src, FrameDev* frame, const PoseDev* pose, int W, int H, Intr k, float maxDepth, float confThreshold,
                        int timeDelta, int order, int id, int* tile_count, int* entries, int entries_cap, float4* rec0, float4* rec1, void* bbox,
                        unsigned long long* keys, hipStream_t s, SplatTuning tune, const VisList* vis) {
    const int tileH = splat_tile_height(tune.tile_h);
    const int tilesX = (W + kTile - 1) / kTile, tilesY = (H + tileH - 1) / tileH, nt = tilesX * tilesY;
    if (nt > kMaxTiles) return -1;
    const int g_sprite_lanes = splat_sprite_lanes(tune.sprite_lanes), g_tile_threads = std::max(splat_tile_threads(tune.tile_threads), ((kTile * tileH + 63) / 64) * 64);
    BinArgs b;
    b.src = src; b.frame = frame; b.pose = pose; b.W = W; b.H = H; b.k = k; b.maxDepth = maxDepth; b.confThreshold = confThreshold;
    b.timeDelta = timeDelta; b.tilesX = tilesX; b.tilesY = tilesY; b.tileH = tileH; b.tile_count = tile_count;
    b.entries = entries; b.tile_cap = entries_cap / nt; b.frame_rw = frame;
    b.rec0 = rec0; b.rec1 = rec1; b.bbox = reinterpret_cast<short4*>(bbox);
    b.vis_list = vis ? vis->list : nullptr; b.vis_count = vis ? vis->count : nullptr;
    // two 1024-thread workgroups fit on a CU (60 VGPRs): 512 workgroups are ONE round of chunks up to a million surfels (with 256 a
    // 0.6 M-surfel map was 293 chunks = two rounds, the second one on 37 CUs)
    const int nblocks = min(512, (src.cap + 2 * kBinThreads - 1) / (2 * kBinThreads));
    hipLaunchKernelGGL(k_splat_bin, dim3(nblocks), dim3(kBinThreads), (size_t)2 * nt * sizeof(int), s, b);
    GlobalTileArgs t;
    t.frame = frame; t.W = W; t.H = H; t.k = k; t.tilesX = tilesX; t.tilesY = tilesY; t.tileH = tileH; t.tile_count = tile_count; t.entries = entries; t.tile_cap = b.tile_cap;
    t.rec0 = rec0; t.rec1 = rec1; t.bbox = reinterpret_cast<const short4*>(bbox);
    t.payload = ((unsigned)order << 8) | ((unsigned)id & 255u); t.keys = keys;
    t.vis_list = b.vis_list; t.vis_count = b.vis_count; t.runs = src.box;
    switch (g_sprite_lanes) {
        case 1: hipLaunchKernelGGL(k_global_tile<1>, dim3(xcd_padded_grid(nt)), dim3(g_tile_threads), 0, s, t); break;
        case 2: hipLaunchKernelGGL(k_global_tile<2>, dim3(xcd_padded_grid(nt)), dim3(g_tile_threads), 0, s, t); break;
        case 8: hipLaunchKernelGGL(k_global_tile<8>, dim3(xcd_padded_grid(nt)), dim3(g_tile_threads), 0, s, t); break;
        case 16: hipLaunchKernelGGL(k_global_tile<16>, dim3(xcd_padded_grid(nt)), dim3(g_tile_threads), 0, s, t); break;
        default: hipLaunchKernelGGL(k_global_tile<4>, dim3(xcd_padded_grid(nt)), dim3(g_tile_threads), 0, s, t); break;
    }
    return 0;
}

size_t splat_tiles_scratch_ints(int W, int H) { return (size_t)((W + kTile - 1) / kTile) * ((H + kTile - 1) / kTile); }

int launch_splat_tiled(Surfels src, FrameDev* frame, const PoseDev* pose, int W, int H, Intr k, float maxDepth, float confThreshold,
                       int timeDelta, int* tile_count, int* entries, int entries_cap, float4* rec0, float4* rec1, void* bbox, float4* predV,
                       float4* predN, uchar4* predImage, uint16_t* predTime, const uint8_t* rgb, uint8_t* predGray, uint8_t* fillGray, hipStream_t s,
                       const FrameAdvance* advance, int fillPassthrough, unsigned long long* prof, SplatTuning tune, const VisList* vis) {
    const int tileH = splat_tile_height(tune.tile_h);
    const int tilesX = (W + kTile - 1) / kTile, tilesY = (H + tileH - 1) / tileH, nt = tilesX * tilesY;
    if (nt > kMaxTiles) return -1;
    const int g_sprite_lanes = splat_sprite_lanes(tune.sprite_lanes), g_tile_threads = std::max(splat_tile_threads(tune.tile_threads), ((kTile * tileH + 63) / 64) * 64);
    BinArgs b;
    b.src = src; b.frame = frame; b.pose = pose; b.W = W; b.H = H; b.k = k; b.maxDepth = maxDepth; b.confThreshold = confThreshold;
    b.timeDelta = timeDelta; b.tilesX = tilesX; b.tilesY = tilesY; b.tileH = tileH; b.tile_count = tile_count;
    b.entries = entries; b.tile_cap = entries_cap / nt; b.frame_rw = frame;
    b.rec0 = rec0; b.rec1 = rec1; b.bbox = reinterpret_cast<short4*>(bbox);
    b.vis_list = vis ? vis->list : nullptr; b.vis_count = vis ? vis->count : nullptr;
    // two 1024-thread workgroups fit on a CU (60 VGPRs): 512 workgroups are ONE round of chunks up to a million surfels (with 256 a
    // 0.6 M-surfel map was 293 chunks = two rounds, the second one on 37 CUs)
    const int nblocks = min(512, (src.cap + 2 * kBinThreads - 1) / (2 * kBinThreads));
    hipLaunchKernelGGL(k_splat_bin, dim3(nblocks), dim3(kBinThreads), (size_t)2 * nt * sizeof(int), s, b);
    TileArgs t;
    t.src = src; t.frame = frame; t.pose = pose; t.W = W; t.H = H; t.k = k; t.maxDepth = maxDepth; t.confThreshold = confThreshold;
    t.timeDelta = timeDelta; t.tilesX = tilesX; t.tilesY = tilesY; t.tileH = tileH; t.tile_count = tile_count;
    t.entries = entries; t.tile_cap = b.tile_cap; t.rec0 = rec0; t.rec1 = rec1; t.bbox = reinterpret_cast<const short4*>(bbox);
    t.predV = predV; t.predN = predN; t.predImage = predImage; t.predTime = predTime; t.rgb = rgb; t.predGray = predGray;
    t.fillGray = fillGray;
    t.fillPassthrough = fillPassthrough;
    t.prof = prof;
    t.vis_list = b.vis_list; t.vis_count = b.vis_count;
    t.advance = advance ? 1 : 0;
    t.adv = advance ? *advance : FrameAdvance{nullptr, nullptr, nullptr};
    switch (g_sprite_lanes) {
        case 1: hipLaunchKernelGGL(k_splat_tile<1>, dim3(xcd_padded_grid(nt)), dim3(g_tile_threads), 0, s, t); break;
        case 2: hipLaunchKernelGGL(k_splat_tile<2>, dim3(xcd_padded_grid(nt)), dim3(g_tile_threads), 0, s, t); break;
        case 8: hipLaunchKernelGGL(k_splat_tile<8>, dim3(xcd_padded_grid(nt)), dim3(g_tile_threads), 0, s, t); break;
        case 16: hipLaunchKernelGGL(k_splat_tile<16>, dim3(xcd_padded_grid(nt)), dim3(g_tile_threads), 0, s, t); break;
        default: hipLaunchKernelGGL(k_splat_tile<4>, dim3(xcd_padded_grid(nt)), dim3(g_tile_threads), 0, s, t); break;
    }
    return 0;
}

}  // namespace mf

// mf_labels_gpu.hip -- device form of the label stage of MfSegmentation::performSegmentation
// (Core/Segmentation/MfSegmentation.cpp:220-522), SURVEY.md 8f-2.
//
// The reference runs this stage on the CPU with OpenCV (README.md:52 names it the bottleneck); the host restatement in
// mf_labels.hip (the executable specification, checked bit for bit against the oracle) cost ~6.3 ms of a 7 ms VGA frame plus
// 2.1 MB of D2H / H2D copies.  Here every step is a kernel over HBM-resident images and the frame's only host visit reads
// back two words (new-model decision) and the per-model `alive` flags:
//   connected components  : union-find on pixel indices (atomicMin on parents), root = first pixel in raster order, so an
//                           ordered scan over the roots numbers the components exactly like cv::connectedComponentsWithStats
//   removeEdges           : five Jacobi sweeps (ping-pong label images)
//   overlap votes         : integer histograms with atomics (order independent), decisions by component / by mask
//   label closing         : grey dilate / erode with cv::getStructuringElement(MORPH_ELLIPSE)'s rows
// Integer logic only (one float depth test, evaluated like the host code), so the result is identical to the host form.
#include "mf_internal.h"
#include "mf_labels.h"

namespace mf {

namespace {

__device__ __forceinline__ int uf_find(const int* __restrict__ L, int i) {
    for (;;) {
        const int p = __hip_atomic_load(&L[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (p == i) return i;
        i = p;
    }
}
__device__ __forceinline__ void uf_unite(int* L, int a, int b) {
    for (;;) {
        a = uf_find(L, a); b = uf_find(L, b);
        if (a == b) return;
        if (a < b) { const int t = a; a = b; b = t; }   // a > b: the smaller index becomes the parent
        const int old = atomicMin(&L[a], b);
        if (old == a) return;
        a = old;
    }
}

// Run-length aggregation of atomics inside a wavefront: label images are piecewise constant along a row, so 64 consecutive
// pixels hold a handful of runs of equal keys.  The first lane of each run issues ONE atomic for the whole run -- one
// atomic per pixel on a few hot addresses serialised at ~11 ns each (3-4 ms per histogram at VGA).
// Returns the run length in the run's first lane, 0 elsewhere.  All 64 lanes must call it.
__device__ __forceinline__ int run_length(bool valid, unsigned long long key, bool force_break) {
    const int lane = threadIdx.x & 63;
    const unsigned long long prev = __shfl_up(key, 1, 64);
    const unsigned long long validMask = __ballot(valid);
    const bool prevValid = lane > 0 && ((validMask >> (lane - 1)) & 1ull);
    const bool leader = valid && (!prevValid || key != prev || force_break);
    const unsigned long long breaks = __ballot(leader) | ~validMask;
    if (!leader) return 0;
    const unsigned long long above = (lane == 63) ? 0ull : (breaks & ~((2ull << lane) - 1ull));
    const int end = above ? (__ffsll((long long)above) - 1) : 64;
    return end - lane;
}

struct LabTables {          // small device-resident tables
    int nLive; int liveId[64]; int liveCls[64];
    int idToIndex[256];     // std::map default: 0
    int idExact[256];       // -1 if no live model has this id
    int classIDs[256]; int nMasks;
    int nComp;              // number of components incl. background (label 0)
    int maskPixels[256]; int maskToID[256];
    int hasNewLabel, newClassID, overflow, pad;
};

__global__ void k_lab_models(LabTables* T, const int* ids, const int* cls, const PoseDev* const* poses, int nModels, const int* classIDs,
                             int nMasks) {
    if (blockIdx.x != 0) return;
    // (the four 256-entry tables are cleared by the whole wavefront: one thread walking them was 8 of this launch's 11 us)
    for (int k = threadIdx.x; k < 256; k += blockDim.x) { T->idToIndex[k] = 0; T->idExact[k] = -1; T->maskPixels[k] = 0; T->classIDs[k] = 0; }
    __syncthreads();
    if (threadIdx.x != 0) return;
    int n = 0;
    for (int m = 0; m < nModels; ++m) {
        if (m > 0 && poses[m]->alive == 0) continue;   // dropped by the jump rule in this frame (MaskFusion.cpp:268-272)
        T->liveId[n] = ids[m]; T->liveCls[n] = cls[m];
        T->idToIndex[ids[m] & 255] = n; T->idExact[ids[m] & 255] = n;
        ++n;
    }
    T->nLive = n;
    T->nMasks = nMasks;
    for (int k = 0; k < nMasks && k < 256; ++k) T->classIDs[k] = classIDs[k];
    T->hasNewLabel = 0; T->newClassID = -1; T->overflow = 0; T->nComp = 1;
}

// Foreground test + horizontal runs: a wavefront covers 64 consecutive pixels; every foreground pixel is linked straight to
// the first pixel of its run inside the wavefront (ballot + bit scan, no atomics).  What remains for the union-find are
// the links across wavefront boundaries and between vertically adjacent runs (k_lab_merge): a few thousand unions with
// shallow trees instead of one union per pixel pair with row-long parent chains (that form took 10 ms at VGA).
__global__ __launch_bounds__(256) void k_lab_init(const uint8_t* __restrict__ binIn, const uint8_t* __restrict__ mask, const LabTables* T,
                                                  int personClassID, uint8_t* __restrict__ ignoreMap, int* __restrict__ L, int W, int P) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    uint8_t b = 0;
    if (i < P) {
        b = binIn[i];
        if (T->nMasks) {   // MfSegmentation.cpp:221-235
            const bool person = T->classIDs[mask_id(mask[i], T->nMasks)] == personClassID;
            ignoreMap[i] = person ? 255 : 0;
            if (person) b = 0;
        } else if (ignoreMap[i]) b = 0;
    }
    const int lane = threadIdx.x & 63;
    const unsigned long long fg = __ballot(b != 0);
    const unsigned long long rowStart = __ballot(i < P && (i % W) == 0);
    const unsigned long long starts = fg & (~(fg << 1) | rowStart);          // first pixel of a run inside this wavefront
    if (i >= P) return;
    if (!b) { L[i] = -1; return; }
    const unsigned long long upto = starts & ((lane == 63) ? ~0ull : ((2ull << lane) - 1ull));
    const int startLane = 63 - __clzll((long long)upto);
    L[i] = i - (lane - startLane);
}

__global__ __launch_bounds__(256) void k_lab_merge(int* L, int W, int P) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P || L[i] < 0) return;
    const int x = i % W;
    // a run that continues from the previous wavefront
    if ((i & 63) == 0 && x > 0 && L[i - 1] >= 0) uf_unite(L, i, i - 1);
    // vertical link, once per overlap of two runs (at its first column)
    if (i >= W && L[i - W] >= 0 && !(x > 0 && L[i - 1] >= 0 && L[i - W - 1] >= 0)) uf_unite(L, i, i - W);
}

__global__ __launch_bounds__(256) void k_lab_flatten(int* L, int P) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P || L[i] < 0) return;
    L[i] = uf_find(L, i);   // roots keep L[i] == i; writing a root's index into a child never un-roots anything
}

__device__ __forceinline__ int block_total(int v, int* s_w) {   // sum over a 256-thread workgroup
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = v;
    __syncthreads();
    const int t = s_w[0] + s_w[1] + s_w[2] + s_w[3];
    __syncthreads();
    return t;
}

__global__ __launch_bounds__(256) void k_lab_count_roots(const int* __restrict__ L, int P, int* __restrict__ blockCounts) {
    __shared__ int s_w[4];
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int isRoot = (i < P && L[i] == i) ? 1 : 0;
    const int t = block_total(isRoot, s_w);
    if (threadIdx.x == 0) blockCounts[blockIdx.x] = t;
}

// component ids in raster order of the first pixel (cv::connectedComponentsWithStats numbering), areas zeroed
__global__ __launch_bounds__(256) void k_lab_number(const int* __restrict__ L, int P, const int* __restrict__ blockCounts, int nBlocks,
                                                    int* __restrict__ compId, int* __restrict__ area, int4* __restrict__ bbox, int W,
                                                    int H, LabTables* T) {
    __shared__ int s_w[4];
    int before = 0;
    for (int b = threadIdx.x; b < blockIdx.x; b += 256) before += blockCounts[b];
    const int base = block_total(before, s_w);
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool isRoot = i < P && L[i] == i;
    const unsigned long long m = __ballot(isRoot);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) s_w[wave] = __popcll(m);
    __syncthreads();
    int off = base;
    for (int w = 0; w < wave; ++w) off += s_w[w];
    if (isRoot) {
        const int id = 1 + off + __popcll(m & ((1ull << lane) - 1ull));
        compId[i] = id;
        area[id] = 0;
        bbox[id] = make_int4(W, H, -1, -1);
    }
    if (blockIdx.x == nBlocks - 1 && threadIdx.x == 0) {
        T->nComp = 1 + base + s_w[0] + s_w[1] + s_w[2] + s_w[3];
        area[0] = 0;
    }
}

// labels + the statistics of cv::connectedComponentsWithStats that the stage uses later: area and bounding box.
// Two levels of aggregation before anything touches global memory: runs of equal labels inside a wavefront (ballot), then
// adjacent runs with the same label across the 16 wavefronts of a 1024-pixel workgroup (LDS).  A component that covers half
// the image would otherwise receive one atomic per wavefront on the same five addresses (~12 k x 11 ns = 117 us at VGA).
constexpr int kRelabelThreads = 1024;
__global__ __launch_bounds__(kRelabelThreads) void k_lab_relabel(const int* __restrict__ L, const int* __restrict__ compId,
                                                                 int* __restrict__ lab, int* __restrict__ area, int4* __restrict__ bbox,
                                                                 int W, int P) {
    __shared__ int s_key[kRelabelThreads], s_len[kRelabelThreads];
    __shared__ int4 s_box[kRelabelThreads];
    __shared__ int s_wcount[kRelabelThreads / 64];
    const int i = blockIdx.x * kRelabelThreads + threadIdx.x;
    const bool in = i < P;
    const int c = (in && L[i] >= 0) ? compId[L[i]] : 0;
    if (in) lab[i] = c;
    const int x = in ? i % W : 0, y = in ? i / W : 0;
    // background statistics (label 0) are never read; wave-level runs break at row starts so that a run has one y, ordered x
    const int len = run_length(in && c != 0, (unsigned long long)c, x == 0);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long leaders = __ballot(len > 0);
    if (lane == 0) s_wcount[wave] = __popcll(leaders);
    __syncthreads();
    int slot = 0, n = 0;
    for (int w = 0; w < kRelabelThreads / 64; ++w) { if (w < wave) slot += s_wcount[w]; n += s_wcount[w]; }
    if (len > 0) {
        slot += __popcll(leaders & ((1ull << lane) - 1ull));
        s_key[slot] = c; s_len[slot] = len; s_box[slot] = make_int4(x, y, x + len - 1, y);
    }
    __syncthreads();
    // entry e starts a merged run if its label differs from the previous entry's; it folds the entries that follow it
    for (int e = threadIdx.x; e < n; e += kRelabelThreads) {
        const int key = s_key[e];
        if (e > 0 && s_key[e - 1] == key) continue;
        int a_sum = s_len[e];
        int4 b = s_box[e];
        for (int f = e + 1; f < n && s_key[f] == key; ++f) {
            a_sum += s_len[f];
            const int4 g = s_box[f];
            b.x = min(b.x, g.x); b.y = min(b.y, g.y); b.z = max(b.z, g.z); b.w = max(b.w, g.w);
        }
        atomicAdd(&area[key], a_sum);
        atomicMin(&bbox[key].x, b.x); atomicMin(&bbox[key].y, b.y); atomicMax(&bbox[key].z, b.z); atomicMax(&bbox[key].w, b.w);
    }
}

// removeEdges, MfSegmentation.cpp:243-291: a pixel that is edge (0) or in a < 50 px component takes the label of the first
// 8-neighbour (row-major order) within 8 mm whose component has > 50 px; neighbours are read from the previous sweep.
// The five sweeps in one launch: a workgroup stages a 32x8 tile with a 5-pixel halo (labels + depth) in LDS and iterates
// on a region that shrinks by one pixel per sweep, so every output pixel sees exactly the five Jacobi steps of the
// launch-per-sweep form (5 x 7.5 us -> one launch).
constexpr int kSwTW = 32, kSwTH = 8, kSwHalo = 5;
constexpr int kSwLW = kSwTW + 2 * kSwHalo, kSwLH = kSwTH + 2 * kSwHalo;   // 42 x 18
__global__ __launch_bounds__(256) void k_lab_sweep5(const int* __restrict__ in, int* __restrict__ out, const int* __restrict__ area,
                                                    const float* __restrict__ depth, int W, int H) {
    __shared__ int s_lab[2][kSwLH * kSwLW];
    __shared__ float s_d[kSwLH * kSwLW];
    const int x0 = blockIdx.x * kSwTW - kSwHalo, y0 = blockIdx.y * kSwTH - kSwHalo;
    for (int i = threadIdx.x; i < kSwLH * kSwLW; i += 256) {
        const int ly = i / kSwLW, lx = i - ly * kSwLW;
        const int gx = x0 + lx, gy = y0 + ly;
        const bool inside = gx >= 0 && gx < W && gy >= 0 && gy < H;
        s_lab[0][i] = inside ? in[gy * W + gx] : 0;
        s_d[i] = inside ? depth[gy * W + gx] : 0.f;
    }
    __syncthreads();
    int cur = 0;
    for (int it = 1; it <= 5; ++it) {
        const int rw = kSwLW - 2 * it, rh = kSwLH - 2 * it;   // region still exact after `it` sweeps
        for (int i = threadIdx.x; i < rw * rh; i += 256) {
            const int ly = it + i / rw, lx = it + i % rw;
            const int gx = x0 + lx, gy = y0 + ly;
            const int li = ly * kSwLW + lx;
            int c = s_lab[cur][li];
            if (gx >= 1 && gx < W - 1 && gy >= 1 && gy < H - 1 && !(c != 0 && area[c] >= 50)) {
                const float d = s_d[li];
                const int ox[8] = {-1, 0, 1, -1, 1, -1, 0, 1}, oy[8] = {-1, -1, -1, 0, 0, 1, 1, 1};
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int lj = (ly + oy[k]) * kSwLW + (lx + ox[k]);
                    const int n = s_lab[cur][lj];
                    if (n != 0 && (double)fabsf(s_d[lj] - d) < 0.008 && area[n] > 50) { c = n; break; }
                }
            }
            s_lab[1 - cur][li] = c;
        }
        __syncthreads();
        cur = 1 - cur;
    }
    const int lx = kSwHalo + (threadIdx.x & 31), ly = kSwHalo + (threadIdx.x >> 5);
    const int gx = x0 + lx, gy = y0 + ly;
    if (gx < W && gy < H) out[gy * W + gx] = s_lab[cur][ly * kSwLW + lx];
}

__global__ __launch_bounds__(256) void k_lab_zero_tables(LabTables* T, int* compMask, int* compModel, int* compToMask, int* compFollow,
                                                         unsigned* overlap, int cap) {
    const int nComp = T->nComp, nMasks = T->nMasks, nLive = T->nLive;
    const long long needMask = (long long)nComp * nMasks, needModel = (long long)nComp * nLive;
    if (needMask > cap || needModel > cap) { if (blockIdx.x == 0 && threadIdx.x == 0) T->overflow = 1; return; }
    const int stride = gridDim.x * 256, t0 = blockIdx.x * 256 + threadIdx.x;
    for (long long k = t0; k < needMask; k += stride) compMask[k] = 0;
    for (long long k = t0; k < needModel; k += stride) compModel[k] = 0;
    for (int k = t0; k < nComp; k += stride) { compToMask[k] = 0; compFollow[k] = 0; }
    for (int k = t0; k < nLive * 256; k += stride) overlap[k] = 0u;
}

__global__ __launch_bounds__(256) void k_lab_hist(const int* __restrict__ lab, const uint8_t* __restrict__ mask,
                                                  const uint8_t* __restrict__ proj, const LabTables* T, int* compMask, int* compModel, int P) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (T->overflow) return;
    const bool in = i < P;
    const int c = in ? lab[i] : 0;
    const int mi = in ? T->idToIndex[proj[i]] : 0;
    const int mv = (in && T->nMasks) ? mask_id(mask[i], T->nMasks) : 0;
    // row 0 of both tables (the background "component") is never read
    const int len = run_length(in && c != 0, ((unsigned long long)c << 16) | ((unsigned long long)mi << 8) | (unsigned long long)mv, false);
    if (len) {
        atomicAdd(&compModel[(size_t)c * T->nLive + mi], len);                       // :299-305
        if (T->nMasks) atomicAdd(&compMask[(size_t)c * T->nMasks + mv], len);       // :307-318
    }
}

// per component: mask assignment (65 % rule, :320-345) and the model it follows if no mask claims it (60 % rule, :500-522)
__global__ __launch_bounds__(256) void k_lab_comp_decide(LabTables* T, const int* __restrict__ area, const int* __restrict__ compMask,
                                                         const int* __restrict__ compModel, int* compToMask, int* compFollow,
                                                         int minMappedComponentSize) {
    if (T->overflow) return;
    const int nComp = T->nComp, nMasks = T->nMasks, nLive = T->nLive;
    for (int c = 1 + blockIdx.x * 256 + threadIdx.x; c < nComp; c += gridDim.x * 256) {
        const int csize = area[c];
        int assigned = 0;
        if (nMasks && csize > minMappedComponentSize) {
            const int t = (int)(0.65f * csize);
            for (int m = 1; m < nMasks; ++m)
                if (compMask[(size_t)c * nMasks + m] > t) { assigned = m; atomicAdd(&T->maskPixels[m], csize); }
        }
        compToMask[c] = assigned;
        int bestM = 0, ov = compModel[(size_t)c * nLive];
        for (int m = 1; m < nLive; ++m)
            if (compModel[(size_t)c * nLive + m] > ov) { ov = compModel[(size_t)c * nLive + m]; bestM = m; }
        const int modelID = T->liveId[bestM];
        compFollow[c] = (modelID > 0 && (float)ov > 0.6f * (float)csize) ? modelID : 0;
    }
}

__global__ __launch_bounds__(256) void k_lab_full1(const int* __restrict__ lab, const int* __restrict__ compToMask,
                                                   const uint8_t* __restrict__ ignoreMap, uint8_t* __restrict__ full, int P) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    full[i] = ignoreMap[i] ? (uint8_t)255 : (uint8_t)compToMask[lab[i]];
}

// grey-level dilate / erode with cv::getStructuringElement(MORPH_ELLIPSE, (2r+1)^2); constant border = neutral element
template <bool kDilate>
__global__ __launch_bounds__(256) void k_lab_morph(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int W, int H, int r) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    int v = kDilate ? 0 : 255;
    const double inv_r2 = r ? 1.0 / ((double)r * r) : 0.0;
    for (int i = 0; i < 2 * r + 1; ++i) {
        const int yy = y + i - r;
        if (yy < 0 || yy >= H) continue;
        const int dy = i - r;
        const int dx = (int)lrint(r * sqrt((r * r - dy * dy) * inv_r2));
        const int lo = max(r - dx, 0), hi = min(r + dx + 1, 2 * r + 1);
        for (int j = lo; j < hi; ++j) {
            const int xx = x + j - r;
            if (xx < 0 || xx >= W) continue;
            const int s = in[yy * W + xx];
            v = kDilate ? max(v, s) : min(v, s);
        }
    }
    out[y * W + x] = (uint8_t)v;
}

__global__ __launch_bounds__(256) void k_lab_overlap(const uint8_t* __restrict__ full, const uint8_t* __restrict__ proj, const LabTables* T,
                                                     unsigned* overlap, int P) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (T->nMasks == 0 || T->overflow) return;
    const bool in = i < P;
    const int b = in ? T->idExact[proj[i]] : -1;
    const int f = in ? full[i] : 0;
    // the background model's row (b == 0) is never read (:455 starts at j = 1)
    const int len = run_length(in && b > 0, ((unsigned long long)(unsigned)b << 8) | (unsigned long long)f, false);
    if (len) atomicAdd(&overlap[b * 256 + f], (unsigned)len);   // :441-447
}

// mask -> model id with the new-model rule (:449-498); sequential by construction ("the first mask that qualifies")
__global__ void k_lab_mask_decide(LabTables* T, const unsigned* __restrict__ overlap, int personClassID, float minMaskModelOverlap,
                                  unsigned long long minNew, unsigned long long maxNew, int nextModelID, int allowNew, int* result_host) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    for (int k = 0; k < 256; ++k) T->maskToID[k] = 0;
    T->maskToID[255] = 255;
    const int nMasks = T->nMasks, nLive = T->nLive;
    if (nMasks && !T->overflow) {
        for (int m = 1; m < nMasks; ++m) T->maskToID[m] = (T->classIDs[m] == personClassID) ? 255 : 0;
        for (int midx = 1; midx < nMasks; ++midx) {
            if (T->maskToID[midx] == 255) continue;
            int best = 0;
            unsigned bestOverlap = 0;
            for (int j = 1; j < nLive; ++j)
                if (overlap[j * 256 + midx] > bestOverlap) { bestOverlap = overlap[j * 256 + midx]; best = j; }
            const bool classMatches = T->liveCls[best] == T->classIDs[midx];
            const int mp = T->maskPixels[midx];
            if ((float)bestOverlap < minMaskModelOverlap * (float)mp) best = 0;
            if (best != 0 && classMatches) {
                T->maskToID[midx] = T->liveId[best];
            } else if (!T->hasNewLabel && allowNew && (unsigned long long)mp > minNew && (unsigned long long)mp < maxNew && best == 0) {
                T->maskToID[midx] = nextModelID;
                T->hasNewLabel = 1;
                T->newClassID = T->classIDs[midx];
            } else {
                T->maskToID[midx] = 255;
            }
        }
    }
    result_host[0] = T->hasNewLabel; result_host[1] = T->newClassID; result_host[2] = T->overflow; result_host[3] = T->nComp;
}

__global__ __launch_bounds__(256) void k_lab_full2(const int* __restrict__ lab, const int* __restrict__ compToMask,
                                                   const int* __restrict__ compFollow, const int4* __restrict__ bbox, const LabTables* T,
                                                   uint8_t* __restrict__ full, int W, int H, int P) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    int v = T->maskToID[full[i]];
    const int c = lab[i];
    if (c >= 1 && compToMask[c] == 0 && compFollow[c] > 0) {   // :500-522
        // the reference repaints only inside the component's bounding box as reported by connectedComponentsWithStats, i.e.
        // BEFORE the edge-growing sweeps (inclusive "+width/+height" bounds, clamped): pixels the sweeps added outside keep 0
        const int4 b = bbox[c];
        const int x = i % W, y = i / W;
        if (x >= b.x && x <= min(b.z + 1, W - 1) && y >= b.y && y <= min(b.w + 1, H - 1)) v = compFollow[c];
    }
    full[i] = (uint8_t)v;
}

}  // namespace

size_t labels_gpu_table_bytes() { return sizeof(LabTables); }

void launch_labels_gpu(const LabelsGpuArgs& a, hipStream_t s) {
    const int P = a.W * a.H, nb = (P + 255) / 256;
    LabTables* T = reinterpret_cast<LabTables*>(a.tables);
    const dim3 g2((a.W + 63) / 64, (a.H + 3) / 4);
    hipLaunchKernelGGL(k_lab_models, dim3(1), dim3(64), 0, s, T, a.model_ids, a.model_cls, a.model_poses, a.nModels, a.class_ids, a.nMasks);
    hipLaunchKernelGGL(k_lab_init, dim3(nb), dim3(256), 0, s, a.binary, a.mask, T, a.prm.personClassID, a.ignoreMap, a.L, a.W, P);
    hipLaunchKernelGGL(k_lab_merge, dim3(nb), dim3(256), 0, s, a.L, a.W, P);
    hipLaunchKernelGGL(k_lab_flatten, dim3(nb), dim3(256), 0, s, a.L, P);
    hipLaunchKernelGGL(k_lab_count_roots, dim3(nb), dim3(256), 0, s, a.L, P, a.blockCounts);
    hipLaunchKernelGGL(k_lab_number, dim3(nb), dim3(256), 0, s, a.L, P, a.blockCounts, nb, a.compId, a.area, a.bbox, a.W, a.H, T);
    hipLaunchKernelGGL(k_lab_relabel, dim3((P + kRelabelThreads - 1) / kRelabelThreads), dim3(kRelabelThreads), 0, s, a.L, a.compId, a.lab[0],
                       a.area, a.bbox, a.W, P);
    int cur = 0;
    if (a.prm.removeEdges) {
        const dim3 gs((a.W + kSwTW - 1) / kSwTW, (a.H + kSwTH - 1) / kSwTH);
        hipLaunchKernelGGL(k_lab_sweep5, gs, dim3(256), 0, s, a.lab[0], a.lab[1], a.area, a.depth, a.W, a.H);
        cur = 1;
    }
    const int* lab = a.lab[cur];
    hipLaunchKernelGGL(k_lab_zero_tables, dim3(512), dim3(256), 0, s, T, a.compMask, a.compModel, a.compToMask, a.compFollow, a.overlap,
                       a.table_cap);
    hipLaunchKernelGGL(k_lab_hist, dim3(nb), dim3(256), 0, s, lab, a.mask, a.proj, T, a.compMask, a.compModel, P);
    hipLaunchKernelGGL(k_lab_comp_decide, dim3(64), dim3(256), 0, s, T, a.area, a.compMask, a.compModel, a.compToMask, a.compFollow,
                       a.prm.minMappedComponentSize);
    hipLaunchKernelGGL(k_lab_full1, dim3(nb), dim3(256), 0, s, lab, a.compToMask, a.ignoreMap, a.full, P);
    if (a.nMasks > 0 && a.prm.morphMaskIterations > 0) {   // :424-426 (iterations == 0 is a copy)
        uint8_t* src = a.full; uint8_t* dst = a.tmp_u8;
        for (int it = 0; it < a.prm.morphMaskIterations; ++it) {
            hipLaunchKernelGGL(k_lab_morph<true>, g2, dim3(256), 0, s, src, dst, a.W, a.H, a.prm.morphMaskRadius);
            uint8_t* t = src; src = dst; dst = t;
        }
        for (int it = 0; it < a.prm.morphMaskIterations; ++it) {
            hipLaunchKernelGGL(k_lab_morph<false>, g2, dim3(256), 0, s, src, dst, a.W, a.H, a.prm.morphMaskRadius);
            uint8_t* t = src; src = dst; dst = t;
        }
        if (src != a.full) (void)hipMemcpyAsync(a.full, src, (size_t)P, hipMemcpyDeviceToDevice, s);
    }
    hipLaunchKernelGGL(k_lab_overlap, dim3(nb), dim3(256), 0, s, a.full, a.proj, T, a.overlap, P);
    const unsigned long long total = (unsigned long long)P;
    hipLaunchKernelGGL(k_lab_mask_decide, dim3(1), dim3(64), 0, s, T, a.overlap, a.prm.personClassID, a.prm.minMaskModelOverlap,
                       (unsigned long long)(a.prm.minRelSizeNew * total), (unsigned long long)(a.prm.maxRelSizeNew * total),
                       a.nextModelID, a.allowNew ? 1 : 0, a.result_host);
    hipLaunchKernelGGL(k_lab_full2, dim3(nb), dim3(256), 0, s, lab, a.compToMask, a.compFollow, a.bbox, T, a.full, a.W, a.H, P);
}

}  // namespace mf

// mf_surfel.hip -- surfel map maintenance on plain HBM arrays (no OpenGL, no transform feedback, no textures).
//
// Replaces (reference, relative to /root/reference):
//   first-frame init         Core/Shaders/vertex_feedback.vert/.geom, init_unstable.vert; Core/Model/Model.cpp:240-285
//   index map                Core/Shaders/index_map.vert/.frag; Core/Model/ModelProjection.cpp:100-152
//   data association         Core/Shaders/data.vert/.geom/.frag; Core/Model/Model.cpp:466-581
//   surfel update            Core/Shaders/update.vert; Core/Model/Model.cpp:583-646
//   clean / append / decay   Core/Shaders/copy_unstable.vert:53-157, .geom; Core/Model/Model.cpp:649-772
//   splat prediction         Core/Shaders/splat.vert, combo_splat.frag; Core/Model/ModelProjection.cpp:187-268
//
// Design (MI355X):
//   * surfels are three float4 streams (48 B/surfel, 16 B per lane per stream, fully coalesced), double buffered;
//   * the GL rasteriser + z-buffer becomes a 64-bit atomicMin on (z bits << 32 | surfel index) per pixel followed by
//     a per-pixel resolve that recomputes the attributes of the winner (ties go to the lower index = GL order);
//   * the 453 MB "update map" the reference clears and scatters into every frame becomes one int per surfel that
//     receives atomicMin(column-major candidate index) -- "first writer wins" without ordering the writers;
//   * transform feedback (ordered stream compaction) becomes a fixed-grid two-pass compaction: pass 1 writes keep
//     flags + per-workgroup counts, pass 2 sums the counts of the workgroups before it in its prologue and copies
//     survivors in order with wavefront ballots.  Surfel order is therefore exactly the reference's (old surfels
//     first, then new ones in column-major pixel order), and no workgroup ever spins on another.
//   * every per-surfel grid is fixed-size and bounded by a device-resident count: no host round trip per pass.
// Every float op individually rounded in this file: several passes re-derive the same quantity in different kernels
// (a surfel's camera-space position in the index-map resolve and again in the clean pass) and then compare them for
// strict inequality; with per-kernel FMA contraction those self-comparisons flip at random.  It also keeps these
// kernels within rounding of a plain reading of the shaders (the oracle is built with -ffp-contract=off).
#pragma clang fp contract(off)
#include "mf_device.h"
#include "mf_walk.h"
#include "mf_rgbd_device.h"

namespace mf {

// ------------------------------------------------------------------------------------------------
// fills
// ------------------------------------------------------------------------------------------------
__global__ void k_fill_keys(unsigned long long* keys, int n) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) keys[i] = kEmptyKey;
}
void launch_fill_keys(unsigned long long* keys, int n, hipStream_t s) {
    hipLaunchKernelGGL(k_fill_keys, dim3(min((n + 255) / 256, 2048)), dim3(256), 0, s, keys, n);
}
__global__ void k_fill_int(int* p, int v, int n) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = v;
}
void launch_fill_int(int* p, int v, int n, hipStream_t s) {
    hipLaunchKernelGGL(k_fill_int, dim3(min((n + 255) / 256, 2048)), dim3(256), 0, s, p, v, n);
}

// ------------------------------------------------------------------------------------------------
// ordered compaction machinery (fixed grid of kCompactBlocks workgroups x 256 threads)
// ------------------------------------------------------------------------------------------------
// number of quarter-rate candidate pixels of frame `tick` (data.vert:117): x % 2 == y % 2 == tick % 2
__device__ __forceinline__ int cand_count(int W, int H, int tick) {
    const int par = tick & 1;
    return ((W - par + 1) / 2) * ((H - par + 1) / 2);
}

// (the grid of an ordered-compaction launch is at most kCompactBlocks workgroups -- the size of its count array -- and the host's choice below that:
// compact_blocks_for)
__device__ __forceinline__ int chunk_size(int n) {
    const int blocks = (int)gridDim.x;
    const int c = (n + blocks - 1) / blocks;
    return ((c + 255) / 256) * 256;
}

// sum of 256 per-thread ints -> every thread gets the block total; s_w: 4 ints of LDS
__device__ __forceinline__ int block_sum_i(int v, int* s_w) {
    v = wave_sum_i(v);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = v;
    __syncthreads();
    const int tot = s_w[0] + s_w[1] + s_w[2] + s_w[3];
    __syncthreads();
    return tot;
}

// exclusive prefix of block_counts for this workgroup (prologue of pass 2)
__device__ __forceinline__ int block_base(const int* __restrict__ block_counts, int* s_w) {
    int v = 0;
    for (int b = threadIdx.x; b < (int)blockIdx.x; b += 256) v += block_counts[b];
    return block_sum_i(v, s_w);
}

// ------------------------------------------------------------------------------------------------
// first-frame initialisation: one record per pixel in column-major slots + keep flag, then compaction
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_init_records(const uint8_t* __restrict__ rgb, const float* __restrict__ depthRaw,
                                                      const float* __restrict__ depthF, int W, int H, Intr k, float maxDepth,
                                                      const FrameDev* __restrict__ frame, float4* __restrict__ rec,
                                                      uint8_t* __restrict__ flags) {
    const int i = blockIdx.x * 64 + (threadIdx.x & 63);
    const int j = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (i >= W || j >= H) return;
    const int slot = i * H + j;  // FeedbackBuffer.cpp:44-50: column-major vertex order
    const float x = (float)i + 0.5f, y = (float)j + 0.5f;
    const float3 vraw = get_vertex(depthRaw, W, H, i, j, x, y, k);
    if (vraw.z <= 0 || vraw.z > maxDepth) { flags[slot] = 0; return; }
    const float3 vfil = get_vertex(depthF, W, H, i, j, x, y, k);
    const float3 n = get_normal_central(depthF, W, H, i, j, x, y, vfil, k);
    const uint8_t* p = rgb + (size_t)(j * W + i) * 3;
    rec[slot * 3 + 0] = make_float4(vraw.x, vraw.y, vraw.z, surfel_confidence(x, y, 1.0f, k));
    rec[slot * 3 + 1] = make_float4((float)((p[0] << 16) + (p[1] << 8) + p[2]), 0.f, 1.f, (float)frame->tick);
    rec[slot * 3 + 2] = make_float4(n.x, n.y, n.z, surfel_radius(vfil.z, n.z, k));
    flags[slot] = 1;
}

void launch_init_surfels(const uint8_t* rgb, const float* depthRaw, const float* depthF, int W, int H, Intr k, float maxDepth,
                         const FrameDev* frame, float4* rec, uint8_t* flags, hipStream_t s) {
    dim3 grid((W + 63) / 64, (H + 3) / 4);
    hipLaunchKernelGGL(k_init_records, grid, dim3(256), 0, s, rgb, depthRaw, depthF, W, H, k, maxDepth, frame, rec, flags);
}

__global__ __launch_bounds__(256) void k_count_flags(const uint8_t* __restrict__ flags, int n, int* __restrict__ block_counts) {
    __shared__ int s_w[4];
    const int chunk = chunk_size(n);
    const int beg = blockIdx.x * chunk, end = min(n, beg + chunk);
    int v = 0;
    for (int i = beg + threadIdx.x; i < end; i += 256) v += flags[i] ? 1 : 0;
    const int tot = block_sum_i(v, s_w);
    if (threadIdx.x == 0) block_counts[blockIdx.x] = tot;
}

__global__ __launch_bounds__(256) void k_compact_records(const float4* __restrict__ rec, const uint8_t* __restrict__ flags, int n,
                                                         Surfels dst, FrameDev* __restrict__ frame,
                                                         const int* __restrict__ block_counts, int* __restrict__ host_count) {
    __shared__ int s_w[4];
    int base = block_base(block_counts, s_w);
    const int chunk = chunk_size(n);
    const int beg = blockIdx.x * chunk, end = min(n, beg + chunk);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i0 = beg; i0 < end; i0 += 256) {
        const int i = i0 + threadIdx.x;
        const bool keep = i < end && flags[i];
        const unsigned long long m = __ballot(keep);
        if (lane == 0) s_w[wave] = __popcll(m);
        __syncthreads();
        int off = base;
        for (int w = 0; w < wave; ++w) off += s_w[w];
        const int tot = s_w[0] + s_w[1] + s_w[2] + s_w[3];
        if (keep) {
            const int o = off + lane_rank(m);
            if (o < dst.cap) { dst.pc[o] = rec[i * 3 + 0]; dst.ct[o] = rec[i * 3 + 1]; dst.nr[o] = rec[i * 3 + 2]; }
        }
        base += tot;
        __syncthreads();
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
        frame->count = frame->phys = min(base, dst.cap);   // a dense buffer without a run table
        frame->runs = 0; frame->first = 0; frame->first_run = 0;
        if (host_count) *host_count = min(base, dst.cap);
    }
}

void launch_compact_records(const float4* rec, const uint8_t* flags, int n, Surfels dst, FrameDev* frame, int* block_counts,
                            int* host_count_mirror, hipStream_t s) {
    hipLaunchKernelGGL(k_count_flags, dim3(kCompactBlocks), dim3(256), 0, s, flags, n, block_counts);
    hipLaunchKernelGGL(k_compact_records, dim3(kCompactBlocks), dim3(256), 0, s, rec, flags, n, dst, frame, block_counts,
                       host_count_mirror);
}

// ------------------------------------------------------------------------------------------------
// index map: scatter (z-test as 64-bit atomicMin) + resolve
// Raster rule: a 1-px point lands in texel (floor(u), floor(v)); LESS on z; lower index wins ties.
// ------------------------------------------------------------------------------------------------
// (bodies are __device__ functions: the single-model kernels call them with their own arguments, the batched object-model kernels at the
// end of this file with one model's arguments picked by blockIdx.z)
// One surfel per lane; `live` lanes hold a surfel index i (the others only take part in the wavefront exchange below).
__device__ __forceinline__ void index_scatter_one(const Surfels& src, int i, bool live, float time, const float* Ri, float3 ti, int W, int H, Intr k,
                                                  float maxDepth, int timeDelta, unsigned long long* __restrict__ keys, int transposed,
                                                  bool pretest = false) {
    int p = -1;
    unsigned long long key = kEmptyKey;
    if (live) {
        const float4 pc = src.pc[i];
        const float lastTime = src.ct[i].w;
        const float3 h = mul33(Ri, f3(pc.x, pc.y, pc.z)) + ti;
        if (!(h.z > maxDepth || h.z <= 0 || time - lastTime > (float)timeDelta)) {  // index_map.vert:46
            const float u = ((k.fx * h.x) / h.z) + k.cx;
            const float v = ((k.fy * h.y) / h.z) + k.cy;
            if (u >= 0.f && u < (float)W && v >= 0.f && v < (float)H) {
                // transposed: column-major key image.  Surfels are stored in column-major creation order (data.vert), so the surfels of a wavefront
                // then meet in a few cache lines of keys instead of one line per image row: on the 26.9 M-surfel map of configs[4] the same pass takes
                // 75 us column-major and 133-168 us row-major (profiles/r05_c4_kernel_stats.csv, r05p_c4_kernel_stats.csv), at VGA the copy-update with
                // the scatter riding on it 17.4 against 23.2 us.  (The packed object maps of that scenario, ~85 surfels per pixel, behave the other way
                // round -- 257 against 190 us -- and are scattered row-major.)  The RESOLVE brings the result into the order its consumer reads it in.
                p = transposed ? (int)floorf(u) * H + (int)floorf(v) : (int)floorf(v) * W + (int)floorf(u);
                key = ((unsigned long long)__float_as_uint(h.z) << 32) | (unsigned)i;
            }
        }
    }
    // Neighbouring surfels of a buffer are neighbours in space (creation order), and in a dense map several in a row land on the SAME texel:
    // each lane takes over the smaller key of the lanes 1, 2 and 4 below it that hit its texel, and only the last lane of such a group (of up
    // to eight) goes to memory with the group's minimum -- the z-test is a minimum, so the keys in memory are the same bits, with a fraction of the
    // device-scope atomics (which retire at a few tens of nanoseconds each when they meet on an address).
#pragma unroll
    for (int d = 1; d <= 4; d <<= 1) {
        const int pn = __shfl_up(p, d, 64);
        const unsigned lo = __shfl_up((unsigned)(key & 0xFFFFFFFFull), d, 64), hi = __shfl_up((unsigned)(key >> 32), d, 64);
        const unsigned long long kn = ((unsigned long long)hi << 32) | lo;
        if ((int)(threadIdx.x & 63) >= d && pn == p && kn < key) key = kn;
    }
    // (three steps cover the 7 lanes below: a lane writes when the next lane hits another texel, and every 8th lane writes in any case, so
    // that a longer group is written in pieces none of whose members is lost)
    const int p_next = __shfl_down(p, 1, 64);
    const bool last_of_group = (threadIdx.x & 7) == 7 || p_next != p;
    if (p >= 0 && last_of_group) {
        if (pretest) zmin_key_pretested(&keys[p], key);
        else zmin_key(&keys[p], key);
    }
}

__device__ __forceinline__ void index_scatter_body(Surfels src, const FrameDev* __restrict__ frame,
                                                   const PoseDev* __restrict__ pose, int W, int H, Intr k, float maxDepth,
                                                   int timeDelta, unsigned long long* __restrict__ keys, int transposed,
                                                   const int* __restrict__ vis_list = nullptr, const int* __restrict__ vis_count = nullptr,
                                                   bool pretest = false) {
    const float time = (float)frame->tick;
    float Ri[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) Ri[q] = pose->Ri[q];
    const float3 ti = f3(pose->ti[0], pose->ti[1], pose->ti[2]);
    for_each_surfel_slice<1>(src, frame, vis_list, vis_count, [&](int i, bool live) {
        index_scatter_one(src, i, live, time, Ri, ti, W, H, k, maxDepth, timeDelta, keys, transposed, pretest);
    });
}

__global__ __launch_bounds__(256) void k_index_scatter(Surfels src, const FrameDev* __restrict__ frame,
                                                       const PoseDev* __restrict__ pose, int W, int H, Intr k, float maxDepth,
                                                       int timeDelta, unsigned long long* __restrict__ keys, int transposed,
                                                       const int* __restrict__ vis_list, const int* __restrict__ vis_count) {
    index_scatter_body(src, frame, pose, W, H, k, maxDepth, timeDelta, keys, transposed, vis_list, vis_count);
}

void launch_index_scatter(Surfels src, const FrameDev* frame, const PoseDev* pose, int W, int H, Intr k, float maxDepth,
                          int timeDelta, unsigned long long* keys, bool transposed, hipStream_t s, int blocks, const VisList* vis) {
    hipLaunchKernelGGL(k_index_scatter, dim3(blocks), dim3(256), 0, s, src, frame, pose, W, H, k, maxDepth, timeDelta, keys,
                       transposed ? 1 : 0, vis ? vis->list : nullptr, vis ? vis->count : nullptr);
}

// ------------------------------------------------------------------------------------------------
// run table (Surfels::box) from scratch, and the visibility test over it
// ------------------------------------------------------------------------------------------------
// what a run's table entry says about its surfels, per thread: box of the positions, newest lastTime, lowest confidence (order-preserving ints)
struct RunAcc {
    int lo[3], hi[3], tmax, cmin;
    __device__ __forceinline__ void reset() { lo[0] = lo[1] = lo[2] = kBoxEmptyMin; hi[0] = hi[1] = hi[2] = kBoxEmptyMax; tmax = kBoxEmptyMax; cmin = kBoxEmptyMin; }
    __device__ __forceinline__ void add(float4 pc, float lastTime) {
        if (pc.x == pc.x && pc.y == pc.y && pc.z == pc.z) {   // (a NaN position is never in view)
            const int ex = box_enc(pc.x), ey = box_enc(pc.y), ez = box_enc(pc.z);
            lo[0] = min(lo[0], ex); lo[1] = min(lo[1], ey); lo[2] = min(lo[2], ez);
            hi[0] = max(hi[0], ex); hi[1] = max(hi[1], ey); hi[2] = max(hi[2], ez);
        }
        // (a NaN time stamp passes every "seen within timeDelta" test of the passes -- !(time - NaN > delta) -- : such a surfel counts as seen now)
        tmax = max(tmax, lastTime == lastTime ? box_enc(lastTime) : 0x7F800000);
        if (pc.w == pc.w) cmin = min(cmin, box_enc(pc.w));   // (a NaN confidence is below no threshold: the age rule of clean never drops it)
    }
};
// block reduction of one run's entry: per-thread (already reduced over the thread's own surfels) -> s_red[wavefront][8]
__device__ __forceinline__ void run_box_reduce(const RunAcc& t, int (*s_red)[8]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int lo[3], hi[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) { lo[q] = wave_min_i(t.lo[q]); hi[q] = wave_max_i(t.hi[q]); }
    const int tmax = wave_max_i(t.tmax), cmin = wave_min_i(t.cmin);
    if (lane == 0) {
#pragma unroll
        for (int q = 0; q < 3; ++q) { s_red[wave][q] = lo[q]; s_red[wave][4 + q] = hi[q]; }
        s_red[wave][3] = tmax; s_red[wave][7] = cmin;
    }
}
// one thread: the four wavefronts' partial results -> the table entry of run r (start slot, live surfels)
__device__ __forceinline__ void run_box_store(int4* __restrict__ box, int r, int start, int len, const int (*s_red)[8]) {
    int4 a, b;
    a.x = min(min(s_red[0][0], s_red[1][0]), min(s_red[2][0], s_red[3][0]));
    a.y = min(min(s_red[0][1], s_red[1][1]), min(s_red[2][1], s_red[3][1]));
    a.z = min(min(s_red[0][2], s_red[1][2]), min(s_red[2][2], s_red[3][2]));
    a.w = max(max(s_red[0][3], s_red[1][3]), max(s_red[2][3], s_red[3][3]));
    b.x = max(max(s_red[0][4], s_red[1][4]), max(s_red[2][4], s_red[3][4]));
    b.y = max(max(s_red[0][5], s_red[1][5]), max(s_red[2][5], s_red[3][5]));
    b.z = max(max(s_red[0][6], s_red[1][6]), max(s_red[2][6], s_red[3][6]));
    b.w = start;
    const int cmin = min(min(s_red[0][7], s_red[1][7]), min(s_red[2][7], s_red[3][7]));
    box[kBoxStride * r] = a; box[kBoxStride * r + 1] = b; box[kBoxStride * r + 2] = make_int4(len, cmin, 0, 0);
}

// A DENSE buffer (slots [0, count)) gets its table: fixed runs of kRun slots.  refresh != 0: the buffer HAS a table (frame->runs > 0) and only the
// entries' contents are recomputed from the surfels (after an in-place update outside a frame: merged surfels moved, their time stamps changed).
__global__ __launch_bounds__(256) void k_run_table(Surfels s, FrameDev* __restrict__ frame, int refresh) {
    __shared__ int s_red[4][8];
    const int n = frame->count;
    const int runs = refresh ? frame->runs : (n + kRun - 1) / kRun;
    for (int r = blockIdx.x; r < runs; r += gridDim.x) {
        const int start = refresh ? run_start(s.box, r) : r * kRun, len = refresh ? run_len(s.box, r) : min(n - r * kRun, kRun);
        RunAcc acc;
        acc.reset();
        for (int i = start + (int)threadIdx.x; i < start + len; i += 256) acc.add(s.pc[i], s.ct[i].w);
        run_box_reduce(acc, s_red);
        __syncthreads();
        if (threadIdx.x == 0) run_box_store(s.box, r, start, len, s_red);
        __syncthreads();
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && !refresh) {
        frame->runs = runs; frame->phys = n; frame->first = 0; frame->first_run = 0;
    }
}
void launch_run_table(Surfels s, FrameDev* frame, hipStream_t st, bool refresh) {
    hipLaunchKernelGGL(k_run_table, dim3(1024), dim3(256), 0, st, s, frame, refresh ? 1 : 0);
}
// entries of a buffer's run table: the fixed runs of a full buffer + the runs Model::clean appends, one frame's new surfels after the other
// (<= P / 4 surfels in <= P / (4 kRun) + 1 runs per frame), until the host compacts the buffer (mf_frame.inl: runs_ub)
size_t run_table_runs(long capacity, long pixels) { return (size_t)(2 * ((capacity + kRun - 1) / kRun) + 4 * ((pixels / 4 + kRun - 1) / kRun + 2) + 64); }
size_t run_table_entries(long capacity, long pixels) { return (size_t)kBoxStride * run_table_runs(capacity, pixels); }

// every corner of a run's box outside the SAME half-space => the whole box is: near / far, then the four image sides as planes through the eye
// (u < -2  <=>  fx x + (cx + 2) z < 0 for z > 0).  The per-surfel tests of the passes are u in [0, W] x [0, H] and 0 <= z <= maxDepth on individually
// rounded floats: the box is tested against the image grown by 2 px and the depth range grown by 1 cm -- metres against rounding errors of
// micrometres.  zlo / zhi: the depth range of the box's corners in the camera frame.
// outside (optional): bit q set = the whole box lies outside the image on side q (0 left, 1 right, 2 top, 3 bottom)
__device__ __forceinline__ bool run_box_in_frustum(int4 a, int4 b, const PoseDev* __restrict__ pose, int W, int H, Intr k, float maxDepth, float& zlo, float& zhi,
                                                   int* outside = nullptr) {
    const float lo[3] = {box_dec(a.x), box_dec(a.y), box_dec(a.z)}, hi[3] = {box_dec(b.x), box_dec(b.y), box_dec(b.z)};
    int out_near = 1, out_far = 1, out_l = 1, out_r = 1, out_t = 1, out_b = 1;
    zlo = INFINITY; zhi = -INFINITY;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const float3 p = f3((c & 1) ? hi[0] : lo[0], (c & 2) ? hi[1] : lo[1], (c & 4) ? hi[2] : lo[2]);
        const float3 h = mul33(pose->Ri, p) + f3(pose->ti[0], pose->ti[1], pose->ti[2]);
        zlo = fminf(zlo, h.z); zhi = fmaxf(zhi, h.z);
        out_near &= h.z < -0.01f;
        out_far &= h.z > maxDepth + 0.01f;
        out_l &= k.fx * h.x + (k.cx + 2.f) * h.z < 0.f;
        out_r &= k.fx * h.x + (k.cx - (float)W - 2.f) * h.z > 0.f;
        out_t &= k.fy * h.y + (k.cy + 2.f) * h.z < 0.f;
        out_b &= k.fy * h.y + (k.cy - (float)H - 2.f) * h.z > 0.f;
    }
    if (outside) *outside = out_l | (out_r << 1) | (out_t << 2) | (out_b << 3);
    return !(out_near | out_far | out_l | out_r | out_t | out_b);
}
// the listed runs go to `list` in no particular order; the last workgroup to finish publishes their number and re-arms the counters
__device__ __forceinline__ void run_list_append(bool listed, int r, int* __restrict__ list, int* __restrict__ count, int* __restrict__ ctl) {
    const unsigned long long m = __ballot(listed);
    int base = 0;
    if ((threadIdx.x & 63) == 0 && m) base = atomicAdd(&ctl[0], __popcll(m));
    base = __shfl(base, 0, 64);
    if (listed) list[base + lane_rank(m)] = r;
}
__device__ __forceinline__ bool run_list_finish(int* __restrict__ count, int* __restrict__ ctl) {
    __syncthreads();
    if (threadIdx.x != 0) return false;
    // (every workgroup's slot reservations have returned before its barrier: relaxed atomics suffice, an agent-scope release would write the L2 back)
    const int done = atomicAdd(&ctl[1], 1);
    if (done != (int)gridDim.x - 1) return false;
    count[0] = __hip_atomic_load(&ctl[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&ctl[0], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&ctl[1], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return true;
}

// One thread per run: conservative frustum test of the run's box under `pose` and the activity test (no surfel seen within timeDelta: every
// projection pass drops all of them, index_map.vert:46, splat.vert:58).
__global__ __launch_bounds__(256) void k_cull(Surfels s, const FrameDev* __restrict__ frame, const PoseDev* __restrict__ pose, int W, int H, Intr k,
                                              float maxDepth, int timeDelta, int* __restrict__ list, int* __restrict__ count, int* __restrict__ ctl) {
    const int runs = frame->runs;
    const float time = (float)frame->tick;
    const int r = blockIdx.x * 256 + threadIdx.x;
    bool vis = false;
    if (r < runs) {
        const int4 a = s.box[kBoxStride * r], b = s.box[kBoxStride * r + 1];
        if (run_len(s.box, r) > 0 && a.x <= b.x && a.y <= b.y && a.z <= b.z && !(time - box_dec(a.w) > (float)timeDelta)) {
            float zlo, zhi;
            vis = run_box_in_frustum(a, b, pose, W, H, k, maxDepth, zlo, zhi);
        }
    }
    run_list_append(vis, r, list, count, ctl);
    (void)run_list_finish(count, ctl);
}
void launch_cull(Surfels s, const FrameDev* frame, const PoseDev* pose, int W, int H, Intr k, float maxDepth, int timeDelta, int* list, int* count,
                 int* ctl, int max_runs, hipStream_t st) {
    hipLaunchKernelGGL(k_cull, dim3((max_runs + 255) / 256), dim3(256), 0, st, s, frame, pose, W, H, k, maxDepth, timeDelta, list, count, ctl);
}

// Two consumers, two output shapes (the index_map.frag attachments that each of them samples):
//   packed == nullptr (the pass that feeds fuse): index + vertConf + normRad as separate images;
//   packed != nullptr (the pass that feeds clean): ONE 32 B record per texel {vertConf.xyzw | colorTime.z, colorTime.w,
//     index bits, 0} -- clean gathers 9 texels per surfel, and three separate images cost it three cache lines per tap.
// Resolve: the winning surfel of every texel -> the maps the consumer reads.  Two consumers, two orders: data.vert's window (fuse_data) reads
// {index, vertConf, normRad(, colorTime)} row-major; copy_unstable.vert's window (clean) reads the packed map {position, confidence | initTime,
// lastTime, index, 0} COLUMN-major (surfels are stored in column-major creation order: consecutive surfels then read consecutive texels).  The
// key image comes in either order too (index_scatter_one).  Where the two orders agree a thread handles texel p of both; where they differ a
// workgroup handles a 16 x 16-pixel tile and transposes it through the LDS: keys are read and records written in pieces of 128-512 bytes.
// Every form also resets the key it read (ready for the next scatter: saves a separate clear pass).
struct ResolvedTexel { int index; float4 vc, nr, ct, p0, p1; };
template <bool kPacked>
__device__ __forceinline__ ResolvedTexel resolve_texel(const Surfels& src, const PoseDev* __restrict__ pose, unsigned long long key, bool want_ct, int first) {
    ResolvedTexel r;
    r.index = 0;
    r.vc = r.nr = r.ct = r.p0 = r.p1 = make_float4(0, 0, 0, 0);
    if (key == kEmptyKey) return r;
    const int i = (int)(unsigned)(key & 0xFFFFFFFFull);
    // The index image holds the VERTEX ID of the winner (index_map.frag), and vertex 0 -- the first surfel of the buffer -- cannot be told from
    // the cleared texture: it occludes like any other surfel and is then read as "no surfel" (data.vert, copy_unstable.vert test index > 0).  In a
    // sparse buffer the first surfel is slot frame->first.
    const int id = i == first ? 0 : i;
    const float4 pc = src.pc[i];
    const float3 h = mul33(pose->Ri, f3(pc.x, pc.y, pc.z)) + f3(pose->ti[0], pose->ti[1], pose->ti[2]);
    if (kPacked) {
        const float4 c4 = src.ct[i];
        r.p0 = make_float4(h.x, h.y, h.z, pc.w);
        r.p1 = make_float4(c4.z, c4.w, __int_as_float(id), 0.f);
    } else {
        const float4 n4 = src.nr[i];
        const float3 n = normalize_gl(mul33(pose->Ri, f3(n4.x, n4.y, n4.z)));
        r.index = id;
        r.vc = make_float4(h.x, h.y, h.z, pc.w);
        r.nr = make_float4(n.x, n.y, n.z, n4.w);
        if (want_ct) r.ct = src.ct[i];   // colorTime image: only the clean pass of a freshly spawned model reads it from here
    }
    return r;
}
// frame planes that travel with the packed map (copy_unstable.vert:139-156 looks the filtered depth and the mask up at the surfel's own texel: in the
// packed, column-major order these are neighbours of its window taps; in the row-major images every lane pulled a sector of its own):
// depthF -> the packed record's spare word, mask -> maskT (column-major bytes)
struct ResolveOut {
    int* index; float4* vc; float4* nr; float4* ct; float4* packed; const float* depthF; const uint8_t* mask; uint8_t* maskT;
    const FrameDev* frame;    // the buffer's frame state (`first`)
    // with `packed`, optional: what Model::clean's mask-disagreement rule (copy_unstable.vert:139-156) can meet in this frame, for the run culling of
    // the in-place clean (k_cull_clean): order-preserving ints -- {min, max} filtered depth over the texels of the LEFT column, the RIGHT column, the
    // TOP row, the BOTTOM row whose mask is foreign to the model (neither its id nor >= 255), then the min over ALL foreign texels (kDecayStats
    // ints); armed (empty) between frames
    int* decay_stats; int maskID;
};
// a workgroup's texels -> the decay statistics (one atomic per statistic and wavefront that saw such a texel)
__device__ __forceinline__ void decay_stats_add(const ResolveOut& o, bool in_image, int x, int y, int W, int H, float depth, int maskValue) {
    if (!o.decay_stats) return;
    const bool foreign = in_image && maskValue != o.maskID && maskValue < 255 && depth == depth;
    const int e = box_enc(depth);
    const bool side[4] = {foreign && x == 0, foreign && x == W - 1, foreign && y == 0, foreign && y == H - 1};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (__ballot(side[q]) == 0ull) continue;      // (wavefront-uniform: most wavefronts hold no border texel at all)
        const int lo = wave_min_i(side[q] ? e : kBoxEmptyMin), hi = wave_max_i(side[q] ? e : kBoxEmptyMax);
        if ((threadIdx.x & 63) == 0) { atomicMin(&o.decay_stats[2 * q], lo); atomicMax(&o.decay_stats[2 * q + 1], hi); }
    }
    if (__ballot(foreign) != 0ull) {
        const int amin = wave_min_i(foreign ? e : kBoxEmptyMin);
        if ((threadIdx.x & 63) == 0) atomicMin(&o.decay_stats[8], amin);
    }
}
// row-major keys -> row-major maps: texel p of the key image is texel p of the outputs
template <bool kPacked>
__device__ __forceinline__ void index_resolve_same_body(Surfels src, const PoseDev* __restrict__ pose, unsigned long long* __restrict__ keys, int P, ResolveOut o) {
    static_assert(!kPacked, "the packed map goes through a tile kernel");
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
    const unsigned long long key = keys[p];
    keys[p] = kEmptyKey;
    const ResolvedTexel r = resolve_texel<false>(src, pose, key, o.ct != nullptr, o.frame->first);
    o.index[p] = r.index; o.vc[p] = r.vc; o.nr[p] = r.nr;
    if (o.ct) o.ct[p] = r.ct;
}
// keys and outputs in different orders (column-major keys -> row-major maps, row-major keys -> column-major packed map): one tile per workgroup
constexpr int kResolveTile = 16;
template <bool kPacked>
__device__ __forceinline__ void index_resolve_transposing_body(Surfels src, const PoseDev* __restrict__ pose, unsigned long long* __restrict__ keys, int W, int H,
                                                               ResolveOut o, int tile) {
    __shared__ float4 s_a[kResolveTile][kResolveTile + 1], s_b[kResolveTile][kResolveTile + 1], s_c[kPacked ? 1 : kResolveTile][kResolveTile + 1];
    __shared__ int s_i[kPacked ? 1 : kResolveTile][kResolveTile + 1];
    __shared__ uint8_t s_mk[kPacked ? kResolveTile : 1][kResolveTile + 1];
    const int tilesX = (W + kResolveTile - 1) / kResolveTile;
    const int x0 = (tile % tilesX) * kResolveTile, y0 = (tile / tilesX) * kResolveTile;
    const int f = threadIdx.x & (kResolveTile - 1), g = threadIdx.x / kResolveTile;   // f: the fast coordinate of consecutive lanes
    {   // read the keys in THEIR order: packed output <- row-major keys (x fast), maps <- column-major keys (y fast)
        const int lx = kPacked ? f : g, ly = kPacked ? g : f;
        const int x = x0 + lx, y = y0 + ly;
        ResolvedTexel r = resolve_texel<kPacked>(src, pose, kEmptyKey, false, 0);
        int mk = 0;
        if (x < W && y < H) {
            const int p = kPacked ? y * W + x : x * H + y;
            const unsigned long long key = keys[p];
            keys[p] = kEmptyKey;
            r = resolve_texel<kPacked>(src, pose, key, o.ct != nullptr, o.frame->first);
            if (kPacked) { r.p1.w = o.depthF[p]; mk = o.mask[p]; s_mk[lx][ly] = (uint8_t)mk; }
        }
        if (kPacked) decay_stats_add(o, x < W && y < H, x, y, W, H, r.p1.w, mk);
        if (kPacked) { s_a[lx][ly] = r.p0; s_b[lx][ly] = r.p1; }
        else { s_i[lx][ly] = r.index; s_a[lx][ly] = r.vc; s_b[lx][ly] = r.nr; if (o.ct) s_c[lx][ly] = r.ct; }
    }
    __syncthreads();
    {   // write the outputs in THEIR order: packed column-major (y fast), maps row-major (x fast)
        const int lx = kPacked ? g : f, ly = kPacked ? f : g;
        const int x = x0 + lx, y = y0 + ly;
        if (x < W && y < H) {
            if (kPacked) {
                const int tp = x * H + y;
                o.packed[2 * tp] = s_a[lx][ly];
                o.packed[2 * tp + 1] = s_b[lx][ly];
                o.maskT[tp] = s_mk[lx][ly];
            } else {
                const int p = y * W + x;
                o.index[p] = s_i[lx][ly]; o.vc[p] = s_a[lx][ly]; o.nr[p] = s_b[lx][ly];
                if (o.ct) o.ct[p] = s_c[lx][ly];
            }
        }
    }
}
// column-major keys -> the packed column-major map: keys and records in the same order, only the frame planes are transposed
__device__ __forceinline__ void index_resolve_packed_body(Surfels src, const PoseDev* __restrict__ pose, unsigned long long* __restrict__ keys, int W, int H,
                                                          ResolveOut o, int tile) {
    __shared__ float s_d[kResolveTile][kResolveTile + 1];
    __shared__ uint8_t s_mk[kResolveTile][kResolveTile + 1];
    const int tilesX = (W + kResolveTile - 1) / kResolveTile;
    const int x0 = (tile % tilesX) * kResolveTile, y0 = (tile / tilesX) * kResolveTile;
    const int f = threadIdx.x & (kResolveTile - 1), g = threadIdx.x / kResolveTile;
    {
        const int x = x0 + f, y = y0 + g;
        float d = 0.f;
        int mk = 0;
        if (x < W && y < H) { d = o.depthF[y * W + x]; mk = o.mask[y * W + x]; s_d[f][g] = d; s_mk[f][g] = (uint8_t)mk; }
        decay_stats_add(o, x < W && y < H, x, y, W, H, d, mk);
    }
    __syncthreads();
    const int x = x0 + g, y = y0 + f;
    if (x < W && y < H) {
        const int tp = x * H + y;
        const unsigned long long key = keys[tp];
        keys[tp] = kEmptyKey;
        ResolvedTexel r = resolve_texel<true>(src, pose, key, false, o.frame->first);
        r.p1.w = s_d[g][f];
        o.packed[2 * tp] = r.p0;
        o.packed[2 * tp + 1] = r.p1;
        o.maskT[tp] = s_mk[g][f];
    }
}
__global__ __launch_bounds__(256) void k_index_resolve(Surfels src, const PoseDev* __restrict__ pose, unsigned long long* __restrict__ keys, int P, ResolveOut o) {
    index_resolve_same_body<false>(src, pose, keys, P, o);
}
__global__ __launch_bounds__(256) void k_index_resolve_packed(Surfels src, const PoseDev* __restrict__ pose, unsigned long long* __restrict__ keys, int W, int H,
                                                              ResolveOut o) {
    index_resolve_packed_body(src, pose, keys, W, H, o, (int)blockIdx.x);
}
template <bool kPacked>
__global__ __launch_bounds__(256) void k_index_resolve_transposing(Surfels src, const PoseDev* __restrict__ pose, unsigned long long* __restrict__ keys, int W, int H,
                                                                   ResolveOut o) {
    index_resolve_transposing_body<kPacked>(src, pose, keys, W, H, o, (int)blockIdx.x);
}
int resolve_tiles(int W, int H) { return ((W + kResolveTile - 1) / kResolveTile) * ((H + kResolveTile - 1) / kResolveTile); }

// packed != nullptr: the packed column-major map (index / vc / nr / ct unused) with the frame's filtered depth in its spare word and the mask
// transposed into maskT; else the row-major maps.  keys_transposed: the order the scatter used.
void launch_index_resolve(Surfels src, const FrameDev* frame, const PoseDev* pose, unsigned long long* keys, int W, int H, int* index, float4* vc,
                          float4* nr, float4* ct, float4* packed, const float* depthF, const uint8_t* mask, uint8_t* maskT, bool keys_transposed,
                          hipStream_t s, int* decay_stats, int maskID) {
    const int P = W * H;
    const ResolveOut o{index, vc, nr, ct, packed, depthF, mask, maskT, frame, packed ? decay_stats : nullptr, maskID};
    const dim3 flat((P + 255) / 256), tiles(resolve_tiles(W, H));
    if (packed) {
        if (keys_transposed) hipLaunchKernelGGL(k_index_resolve_packed, tiles, dim3(256), 0, s, src, pose, keys, W, H, o);
        else hipLaunchKernelGGL(k_index_resolve_transposing<true>, tiles, dim3(256), 0, s, src, pose, keys, W, H, o);
    } else {
        if (keys_transposed) hipLaunchKernelGGL(k_index_resolve_transposing<false>, tiles, dim3(256), 0, s, src, pose, keys, W, H, o);
        else hipLaunchKernelGGL(k_index_resolve, flat, dim3(256), 0, s, src, pose, keys, P, o);
    }
}

// ------------------------------------------------------------------------------------------------
// data association (data.vert).  Threads walk candidates row-major (coalesced image reads); the candidate's
// column-major index c = xi * nyc + yi is only its slot / its priority in the first-writer-wins merge.
// ------------------------------------------------------------------------------------------------
struct FuseDataArgs {
    const uint8_t* rgb; const float* depthRaw; const float* depthF; const uint8_t* mask; int maskID;
    const FrameDev* frame; const PoseDev* pose; float weightMultiplier; float maxDepth;
    int bboxLimit;                 // 1: object models limit their fusion depth by lastBoundingBox (upstream with its GUI; "objectBoundingBoxLimit")
    int W, H; Intr k;
    const int* index; const float4* vc; const float4* nr;
    uint8_t* cand_op; float4* cand_rec; int* upd_first;
    int* cand_best;                // surfel a merge candidate was associated with (read by the update pass; untouched for the others)
};

__device__ __forceinline__ void fuse_data_body(const FuseDataArgs& a) {
    const int time = a.frame->tick;
    const int par = time & 1;
    const int W = a.W, H = a.H;
    const int nxc = (W - par + 1) / 2, nyc = (H - par + 1) / 2;
    const int xi = blockIdx.x * 64 + (threadIdx.x & 63);
    const int yi = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (xi >= nxc || yi >= nyc) return;
    const int c = xi * nyc + yi;
    const int px = 2 * xi + par, py = 2 * yi + par;
    const Intr k = a.k;
    uint8_t op = 0;
    const float x = (float)px + 0.5f, y = (float)py + 0.5f;
    const float3 vLocal = get_vertex(a.depthRaw, W, H, px, py, x, y, k);
    bool valid = a.mask[py * W + px] == a.maskID;
    valid = valid && !(texf(a.depthRaw, W, H, px - 1, py) == 0 || texf(a.depthRaw, W, H, px, py - 1) == 0 ||
                       texf(a.depthRaw, W, H, px + 1, py) == 0 || texf(a.depthRaw, W, H, px, py + 1) == 0);
    // Model::fuse's maxDepth uniform = min(depthCutoff, model.maxDepth, bb_max_z) (Model.cpp:480-501,527): an OBJECT model whose
    // bounding box exists (the GUI drew it after the previous frame) does not grow in depth beyond the box + 5 %.  The box's two corners are
    // taken to the camera frame with pose^-1 and only their z is used.
    float maxDepth = a.maxDepth;
    if (a.maskID != 0 && a.bboxLimit) {
        const FrameDev* f = a.frame;
        if (f->bbox[0] <= f->bbox[3] && f->bbox[1] <= f->bbox[4] && f->bbox[2] <= f->bbox[5]) {   // !lastBoundingBox.isEmpty()
            const float* Ri = a.pose->Ri; const float* ti = a.pose->ti;
            const float bbscale = 0.001f;
            const float zmin = ((Ri[6] * (bbscale * (float)f->bbox[0]) + Ri[7] * (bbscale * (float)f->bbox[1])) + Ri[8] * (bbscale * (float)f->bbox[2])) + ti[2];
            const float zmax = ((Ri[6] * (bbscale * (float)f->bbox[3]) + Ri[7] * (bbscale * (float)f->bbox[4])) + Ri[8] * (bbscale * (float)f->bbox[5])) + ti[2];
            const float lo = zmin < zmax ? zmin : zmax, hi = zmin < zmax ? zmax : zmin;
            maxDepth = fminf(maxDepth, hi + 0.05f * fabsf(hi - lo));
        }
    }
    valid = valid && (vLocal.z > 0 && vLocal.z <= maxDepth);
    if (valid) {
        float R[9];
#pragma unroll
        for (int q = 0; q < 9; ++q) R[q] = a.pose->R[q];
        const float3 t = f3(a.pose->t[0], a.pose->t[1], a.pose->t[2]);
        const float3 vGlobal = mul33(R, vLocal) + t;
        const float3 vF = get_vertex(a.depthF, W, H, px, py, x, y, k);
        const float3 nLocal = get_normal_central(a.depthF, W, H, px, py, x, y, vF, k);
        const float3 nGlobal = mul33(R, nLocal);
        const uint8_t* pc = a.rgb + (size_t)(py * W + px) * 3;
        const float weighting = a.pose->fusionWeight * a.weightMultiplier;

        const float xl = (x - k.cx) * (1.0f / k.fx), yl = (y - k.cy) * (1.0f / k.fy);
        const float lambda = sqrtf(xl * xl + yl * yl + 1);
        const float3 ray = f3(xl, yl, 1);
        float bestDist = 1000;
        int best = 0;
        bool merge = false;
        // data.vert:139-141 window: pixel-centre offsets {-1,-0.5,0,+0.5} -> texels {x-1,x,x,x+1}; a revisited texel can
        // never replace itself (strict <), so the 3x3 distinct texels in first-visit order are equivalent.
        // All 27 window loads are issued unconditionally and together: guarding vertConf / normRad behind the index and
        // z tests made every tap a chain of three dependent gathers (17.5 us for 77 k candidates).
        int cur9[9]; float4 vc9[9], nr9[9];
#pragma unroll
        for (int q = 0; q < 9; ++q) {
            const int da = q / 3 - 1, db = q % 3 - 1;
            const int tp = clampi(py + db, 0, H - 1) * W + clampi(px + da, 0, W - 1);
            cur9[q] = a.index[tp];
            vc9[q] = a.vc[tp];
            nr9[q] = a.nr[tp];
        }
#pragma unroll
        for (int q = 0; q < 9; ++q) {
            const int current = cur9[q];
            if (current > 0) {
                const float4 vc = vc9[q];
                const float zdiff = vc.z - vLocal.z;
                if (fabsf(zdiff * lambda) < 0.05f) {
                    const float dist = norm3(cross3(ray, f3(vc.x, vc.y, vc.z)));
                    const float4 nr = nr9[q];
                    const float3 nn = f3(nr.x, nr.y, nr.z);
                    const float ang = shader_acos(dot3(nn, nLocal) / (norm3(nn) * norm3(nLocal)));
                    if (dist < bestDist && (fabsf(nr.z) < 0.75f || fabsf(ang) < 0.5f)) {
                        merge = true; bestDist = dist; best = current;
                    }
                }
            }
        }
        op = merge ? 1 : 2;
        a.cand_rec[c * 3 + 0] = make_float4(vGlobal.x, vGlobal.y, vGlobal.z, surfel_confidence(x, y, weighting, k));
        a.cand_rec[c * 3 + 1] = make_float4((float)((pc[0] << 16) + (pc[1] << 8) + pc[2]), 0.f, (float)time, merge ? -1.f : -2.f);
        a.cand_rec[c * 3 + 2] = make_float4(nGlobal.x, nGlobal.y, nGlobal.z, surfel_radius(vF.z, nLocal.z, k));
        if (merge) {
            atomicMin(&a.upd_first[best], c);  // first writer (lowest column-major index) wins
            a.cand_best[c] = best;
        }
    }
    a.cand_op[c] = op;
}

__global__ __launch_bounds__(256) void k_fuse_data(const FuseDataArgs a) { fuse_data_body(a); }

void launch_fuse_data(const uint8_t* rgb, const float* depthRaw, const float* depthF, const uint8_t* mask, int maskID,
                      const FrameDev* frame, const PoseDev* pose, float weightMultiplier, float maxDepth, int W, int H, Intr k,
                      const int* index, const float4* vc, const float4* nr, uint8_t* cand_op, float4* cand_rec, int* upd_first,
                      int* cand_best, hipStream_t s, int bboxLimit) {
    FuseDataArgs a{rgb, depthRaw, depthF, mask, maskID, frame, pose, weightMultiplier, maxDepth, bboxLimit, W, H, k,
                   index, vc, nr, cand_op, cand_rec, upd_first, cand_best};
    dim3 grid(((W + 1) / 2 + 63) / 64, ((H + 1) / 2 + 3) / 4);
    hipLaunchKernelGGL(k_fuse_data, grid, dim3(256), 0, s, a);
}

// ------------------------------------------------------------------------------------------------
// surfel update (update.vert), IN PLACE.  The reference copies the whole buffer through transform feedback (Model.cpp:583-646) although
// only the surfels a candidate was merged into change -- at most P / 4 of them, of 26 M in a full map.  Here one thread per candidate
// updates "its" surfel where it stands (first writer wins: the candidate with the lowest column-major index owns the merge, decided by
// the atomicMin of the association pass) and re-arms the surfel's merge slot; no surfel reads another, so the values are the ones the
// copying form wrote (rounds 1-4: an N-sized pass of 100 B per surfel with the second index scatter riding on it).
// ------------------------------------------------------------------------------------------------
// update.vert:40-96 for one surfel and the candidate merged into it (records in registers)
__device__ __forceinline__ void surfel_merge(float4& pc, float4& ct, float4& nr, float4 mp, float4 mc, float4 mn, float time) {
    const float c_k = pc.w, a = mp.w;
    if (mn.w < (1.0f + 0.5f) * nr.w) {
        pc = make_float4(((c_k * pc.x) + (a * mp.x)) / (c_k + a), ((c_k * pc.y) + (a * mp.y)) / (c_k + a),
                         ((c_k * pc.z) + (a * mp.z)) / (c_k + a), c_k + a);
        const float3 oc = decode_color(ct.x), nc = decode_color(mc.x);
        ct = make_float4(encode_color(((c_k * oc.x) + (a * nc.x)) / (c_k + a), ((c_k * oc.y) + (a * nc.y)) / (c_k + a),
                                      ((c_k * oc.z) + (a * nc.z)) / (c_k + a)),
                         ct.y, ct.z, time);
        const float4 av = make_float4(((c_k * nr.x) + (a * mn.x)) / (c_k + a), ((c_k * nr.y) + (a * mn.y)) / (c_k + a),
                                      ((c_k * nr.z) + (a * mn.z)) / (c_k + a), ((c_k * nr.w) + (a * mn.w)) / (c_k + a));
        const float3 nn = normalize_gl(f3(av.x, av.y, av.z));
        nr = make_float4(nn.x, nn.y, nn.z, av.w);
    } else {
        pc.w = c_k + a;
        ct.w = time;
    }
}

__device__ __forceinline__ void fuse_update_body(Surfels s, const FrameDev* __restrict__ frame, int* __restrict__ upd_first,
                                                 const uint8_t* __restrict__ cand_op, const int* __restrict__ cand_best,
                                                 const float4* __restrict__ cand_rec, int W, int H) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= cand_count(W, H, frame->tick)) return;
    if (cand_op[c] != 1) return;
    const int i = cand_best[c];
    if (i < 0 || i >= frame->phys) return;
    if (upd_first[i] != c) return;          // an earlier candidate owns this surfel's merge
    upd_first[i] = kNoUpdate;
    float4 pc = s.pc[i], ct = s.ct[i], nr = s.nr[i];
    surfel_merge(pc, ct, nr, cand_rec[c * 3 + 0], cand_rec[c * 3 + 1], cand_rec[c * 3 + 2], (float)frame->tick);
    s.pc[i] = pc; s.ct[i] = ct; s.nr[i] = nr;
}

// The same as a COPY src -> dst over the whole buffer with the second index scatter (predictIndices after fuse, MaskFusion.cpp:556) riding on
// the values just written (rounds 1-4): the form for SMALL maps, where one 12 us pass beats an in-place update + a scatter launch of its own
// (8 + 10 us at VGA) and the whole frame is a chain of such launches; on a 26.9 M-surfel map the copy is 100 B per surfel, 0.56 ms.
struct IndexScatterArgs { const PoseDev* pose; int W, H; Intr k; float maxDepth; int timeDelta; unsigned long long* keys; int transposed; };
__device__ __forceinline__ void fuse_update_copy_body(Surfels src, Surfels dst, const FrameDev* __restrict__ frame, int* __restrict__ upd_first,
                                                      const float4* __restrict__ cand_rec, const IndexScatterArgs& ix) {
    const int n = frame->count;
    const float time = (float)frame->tick;
    float Ri[9];
    float3 ti = f3(0, 0, 0);
    if (ix.keys) {
#pragma unroll
        for (int q = 0; q < 9; ++q) Ri[q] = ix.pose->Ri[q];
        ti = f3(ix.pose->ti[0], ix.pose->ti[1], ix.pose->ti[2]);
    }
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        float4 pc = src.pc[i], ct = src.ct[i], nr = src.nr[i];
        const int m = upd_first[i];
        if (m != kNoUpdate) {
            upd_first[i] = kNoUpdate;
            surfel_merge(pc, ct, nr, cand_rec[m * 3 + 0], cand_rec[m * 3 + 1], cand_rec[m * 3 + 2], time);
        }
        dst.pc[i] = pc; dst.ct[i] = ct; dst.nr[i] = nr;
        if (ix.keys) {   // k_index_scatter on the updated surfel (index_map.vert:40-60)
            const float3 h = mul33(Ri, f3(pc.x, pc.y, pc.z)) + ti;
            if (!(h.z > ix.maxDepth || h.z <= 0 || time - ct.w > (float)ix.timeDelta)) {
                const float u = ((ix.k.fx * h.x) / h.z) + ix.k.cx;
                const float v = ((ix.k.fy * h.y) / h.z) + ix.k.cy;
                if (u >= 0.f && u < (float)ix.W && v >= 0.f && v < (float)ix.H) {
                    const int p = ix.transposed ? (int)floorf(u) * ix.H + (int)floorf(v) : (int)floorf(v) * ix.W + (int)floorf(u);
                    zmin_key(&ix.keys[p], ((unsigned long long)__float_as_uint(h.z) << 32) | (unsigned)i);
                }
            }
        }
    }
}
__global__ __launch_bounds__(256) void k_fuse_update_copy(Surfels src, Surfels dst, const FrameDev* __restrict__ frame, int* __restrict__ upd_first,
                                                          const float4* __restrict__ cand_rec, const IndexScatterArgs ix) {
    fuse_update_copy_body(src, dst, frame, upd_first, cand_rec, ix);
}
void launch_fuse_update_copy(Surfels src, Surfels dst, const FrameDev* frame, int* upd_first, const float4* cand_rec, const PoseDev* pose,
                             int W, int H, Intr k, float maxDepth, int timeDelta, unsigned long long* keys_or_null, bool transposed,
                             hipStream_t s, int blocks) {
    IndexScatterArgs ix{pose, W, H, k, maxDepth, timeDelta, keys_or_null, transposed ? 1 : 0};
    hipLaunchKernelGGL(k_fuse_update_copy, dim3(blocks), dim3(256), 0, s, src, dst, frame, upd_first, cand_rec, ix);
}

__global__ __launch_bounds__(256) void k_fuse_update(Surfels s, const FrameDev* __restrict__ frame, int* __restrict__ upd_first,
                                                     const uint8_t* __restrict__ cand_op, const int* __restrict__ cand_best,
                                                     const float4* __restrict__ cand_rec, int W, int H) {
    fuse_update_body(s, frame, upd_first, cand_op, cand_best, cand_rec, W, H);
}

static inline int cand_blocks(int W, int H) { return (((W + 1) / 2) * ((H + 1) / 2) + 255) / 256; }

void launch_fuse_update(Surfels s, const FrameDev* frame, int* upd_first, const uint8_t* cand_op, const int* cand_best, const float4* cand_rec,
                        int W, int H, hipStream_t st) {
    hipLaunchKernelGGL(k_fuse_update, dim3(cand_blocks(W, H)), dim3(256), 0, st, s, frame, upd_first, cand_op, cand_best, cand_rec, W, H);
}

// ------------------------------------------------------------------------------------------------
// clean (copy_unstable.vert:53-157): pass 1 = per-element test + new confidence + per-workgroup counts;
// pass 2 = ordered copy.  Element i < count is an old surfel, element count + c is candidate c (only op == 2 live).
// ------------------------------------------------------------------------------------------------
struct CleanArgs {
    Surfels src, dst; FrameDev* frame; const PoseDev* pose; int W, H; Intr k;
    int timeDelta; float confThreshold; float outlierCoeff; int maskID; int transposed;
    int literal;   // 1: the window is walked with copy_unstable.vert's own fp32 induction variable (4 or 5 steps per axis), 0: 4 x 4
    const int* index; const float4* vc; const float4* ct;   // separate images (only when the packed map is absent)
    const float4* packed;                                    // {vertConf | initTime, lastTime, index, 0} per texel
    const float* depthF; const uint8_t* mask;
    const uint8_t* maskT;                                    // the mask in the packed map's order (only with `packed`)
    const uint8_t* cand_op; const float4* cand_rec;
    uint8_t* flags; float* newconf;    // keep flag / new confidence per element: pass 1 -> pass 2 of the two-launch form
    int* block_counts;                 // [kCompactBlocks] survivors per workgroup (two-launch form)
    int* host_count;
    unsigned long long* host_append; unsigned seq;   // append: CleanIn::host_append / seq
    int append;                        // two-launch form, 1: the buffer's own surfels have been cleaned where they stand (k_clean_runs) -- the two passes
                                       // handle the frame's candidates only and append the survivors at frame->phys (dst = src)
    const int* run_list; const int* run_count;   // k_clean_runs: the runs to visit (k_cull_clean); nullptr: every run of the table
    int* ctl;                          // k_clean_runs: kCleanCtlInts ints, zero between launches
};

// The window of copy_unstable.vert:85-86 along one axis, exactly as the shader text walks it: `for (i = c - 2s; i < c + 2s; i += s)` on an
// fp32 induction variable makes 4 steps in exact arithmetic and 4 OR 5 in fp32, depending on the rounding of the centre c (27 % of
// the x positions at 640 columns make 5); every step is a tap that COUNTS towards `count > 8` / `zCount > 4`.  A tap's texel is the
// floor of its coordinate snapped to 1/256 texel -- the rule under which the reference's own shader, compiled from its source
// (oracle/_ref/libmf_glsl.so), was executed and the CPU restatement's literal mode was checked against it bit for bit.  The taps
// are monotone and span two pixels: at most three distinct texels u[] with multiplicities m[] (an unused slot has m = 0).
__device__ __forceinline__ void window_slots_literal(float c, int size, int (&u)[3], int (&m)[3]) {
    const float fs = (float)size;
    const float step = (1.0f / (fs * 1.0f)) * 0.5f;     // indexXStep = stepX * 0.5 / scale, scale = FACTOR = 1
    const float half = (1.0f * step) * 2.0f;            // scale * indexXStep * windowMultiplier
    const float end = c + half;
    u[0] = u[1] = u[2] = 0; m[0] = m[1] = m[2] = 0;
    int s = -1, last = -1;                               // texels are >= 0 after the clamp
    float i = c - half;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        if (i < end) {
            const int t = clampi((int)floorf(rintf(i * fs * 256.0f) * (1.0f / 256.0f)), 0, size - 1);
            if (t != last) { s = min(s + 1, 2); u[s] = t; last = t; }
            m[s] += 1;
        }
        i += step;
    }
}

// decay: which factor the mask-disagreement rule applied to the confidence (0 none, 1: k, 2: 0.25 k) -- clean_decayed() re-applies it
// nr_lazy != nullptr: the surfel's normal / radius record is only fetched when the window is walked (it is not needed otherwise); `nr` is then ignored
__device__ __forceinline__ bool clean_test(const CleanArgs& a, float4 pc, float4 ct, float4 nr, float time, const float* Ri,
                                           float3 ti, float& newconf, int& decay, const float4* nr_lazy = nullptr) {
    const int W = a.W, H = a.H;
    bool test = true;
    const float3 lp = mul33(Ri, f3(pc.x, pc.y, pc.z)) + ti;
    const float x = ((a.k.fx * lp.x) / lp.z) + a.k.cx;
    const float y = ((a.k.fy * lp.y) / lp.z) + a.k.cy;
    int count = 0, zCount = 0;
    if (time - ct.w < (float)a.timeDelta && lp.z > 0 && x > 0 && y > 0 && x < (float)W && y < (float)H) {
        if (nr_lazy) nr = *nr_lazy;
        const float3 ln = normalize_gl(mul33(Ri, f3(nr.x, nr.y, nr.z)));
        // copy_unstable.vert:86-87 samples a 4x4 window at offsets {-1,-0.5,0,+0.5} px; nearest fetches land on only 2 or
        // 3 distinct texels per axis, so the window is walked as <= 3x3 distinct texels weighted by their multiplicity
        // (identical counts, ~2.5x fewer gathers).
        int txs[4], tys[4];
        txs[0] = clampi((int)floorf(x - 1.0f), 0, W - 1); txs[1] = clampi((int)floorf(x - 0.5f), 0, W - 1);
        txs[2] = clampi((int)floorf(x), 0, W - 1);        txs[3] = clampi((int)floorf(x + 0.5f), 0, W - 1);
        tys[0] = clampi((int)floorf(y - 1.0f), 0, H - 1); tys[1] = clampi((int)floorf(y - 0.5f), 0, H - 1);
        tys[2] = clampi((int)floorf(y), 0, H - 1);        tys[3] = clampi((int)floorf(y + 0.5f), 0, H - 1);
        // the four fetches are monotone: t0 <= t1 <= t2 <= t3 and t1 is t0 or t2 -> three static slots {t0, t2, t3}
        int ux[3] = {txs[0], txs[2], txs[3]};
        int mx[3] = {1 + (txs[1] == txs[0]), 1 + (txs[1] != txs[0]) + (txs[3] == txs[2]), (txs[3] != txs[2]) ? 1 : 0};
        int uy[3] = {tys[0], tys[2], tys[3]};
        int my[3] = {1 + (tys[1] == tys[0]), 1 + (tys[1] != tys[0]) + (tys[3] == tys[2]), (tys[3] != tys[2]) ? 1 : 0};
        if (a.literal) {   // the shader text's own trip count (see window_slots_literal)
            window_slots_literal(x / (float)W, W, ux, mx);
            window_slots_literal(y / (float)H, H, uy, my);
        }
#pragma unroll
        for (int ia = 0; ia < 3; ++ia) {
#pragma unroll
            for (int ib = 0; ib < 3; ++ib) {
                const int tp = a.transposed ? ux[ia] * H + uy[ib] : uy[ib] * W + ux[ia];
                const int mult = mx[ia] * my[ib];
                if (mult <= 0) continue;   // (requesting the taps' records in groups before looking at any -- 6 + 3, all 9 -- was tried in rounds 2 and 5: slower at VGA, DESIGN.md "rejected")
                float4 v, c;
                int idx;
                if (a.packed) {
                    v = a.packed[2 * tp];
                    const float4 r1 = a.packed[2 * tp + 1];
                    c = make_float4(0.f, 0.f, r1.x, r1.y);
                    idx = __float_as_int(r1.z);
                } else {
                    idx = a.index[tp];
                    v = a.vc[tp];
                    c = a.ct[tp];
                }
                if (idx > 0) {
                    const float dx = v.x - lp.x, dy = v.y - lp.y;
                    if (c.z < ct.z && v.w > a.confThreshold && v.z > lp.z && v.z - lp.z < 0.01f &&
                        sqrtf(dx * dx + dy * dy) < nr.w * 1.4f)
                        count += mult;
                    if (c.w == time && v.w > a.confThreshold && v.z > lp.z && v.z - lp.z > 0.01f && fabsf(ln.z) > 0.85f)
                        zCount += mult;
                }
            }
        }
    }
    if (count > 8 || zCount > 4) test = false;
    float w = ct.w;
    if (w == -2.f) w = time;
    if (w == -1.f || ((time - w) > 20 && pc.w < a.confThreshold)) test = false;
    if (w > 0 && time - w > (float)a.timeDelta) test = true;
    // mask-disagreement decay, copy_unstable.vert:139-156 (nearest fetch, clamp to edge, NaN -> texel 0)
    const int fx_ = isnan(x) ? 0 : clampi((int)fminf(fmaxf(floorf(x), -1.f), (float)W), 0, W - 1);
    const int fy_ = isnan(y) ? 0 : clampi((int)fminf(fmaxf(floorf(y), -1.f), (float)H), 0, H - 1);
    // (with the packed map: the filtered depth rides in its spare word, the mask in the column-major plane beside it -- launch_index_resolve)
    const float wDepth = a.packed ? a.packed[2 * (fx_ * H + fy_) + 1].w : a.depthF[fy_ * W + fx_];
    const int maskValue = a.packed ? a.maskT[fx_ * H + fy_] : a.mask[fy_ * W + fx_];
    newconf = pc.w;
    decay = 0;
    if (maskValue != a.maskID && maskValue < 255 && (wDepth > lp.z - 0.05f && wDepth < lp.z + 0.05f)) {
        const float kk = 0.5f + 0.5f * (1 - a.outlierCoeff / 10.0f);
        if (maskValue == 0) { newconf *= kk; decay = 1; }
        else if (a.maskID == 0) { newconf *= 0.25f * kk; decay = 2; }
        else { newconf *= kk; decay = 1; }
    }
    return test;
}
// the confidence clean_test returned for a record, from the record and the decay code alone (the same float operations)
__device__ __forceinline__ float clean_decayed(const CleanArgs& a, float conf, int decay) {
    const float kk = 0.5f + 0.5f * (1 - a.outlierCoeff / 10.0f);
    if (decay == 1) conf *= kk;
    else if (decay == 2) conf *= 0.25f * kk;
    return conf;
}

// ------------------------------------------------------------------------------------------------
// clean, two launches over a static partition: pass 1 = per-element test + new confidence + per-workgroup counts, pass 2 = ordered copy.
//   * SMALL maps (append == 0; rounds 1-4): every element -- the buffer's surfels and the frame's candidates -- src -> dst, a dense buffer;
//   * the tail of the IN-PLACE clean of big maps (append == 1, below): the candidates only; their survivors are appended behind the buffer's last
//     run, and their runs to its table.
// The surviving records, their order and the count are the same bits either way (tests/test_gpu_switches.py::test_clean_forms_agree).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void clean_small_flags_body(const CleanArgs& a) {
    __shared__ int s_w[4];
    // elements below `count` are the buffer's slots, the others the candidates; `first`: the first element these passes handle
    const int count = a.append ? a.frame->phys : a.frame->count;
    const int first = a.append ? count : 0;
    const int ncand = cand_count(a.W, a.H, a.frame->tick);
    const int total = count + ncand;
    const float time = (float)a.frame->tick;
    float Ri[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) Ri[q] = a.pose->Ri[q];
    const float3 ti = f3(a.pose->ti[0], a.pose->ti[1], a.pose->ti[2]);
    const int chunk = chunk_size(total - first);
    const int beg = first + blockIdx.x * chunk, end = min(total, beg + chunk);
    int kept = 0;
    for (int i = beg + threadIdx.x; i < end; i += 256) {
        bool keep = false;
        float nc = 0.f;
        int dk = 0;
        if (i < count) {
            keep = clean_test(a, a.src.pc[i], a.src.ct[i], a.src.nr[i], time, Ri, ti, nc, dk);
        } else {
            const int c = i - count;
            if (a.cand_op[c] == 2)
                keep = clean_test(a, a.cand_rec[c * 3 + 0], a.cand_rec[c * 3 + 1], a.cand_rec[c * 3 + 2], time, Ri, ti, nc, dk);
        }
        a.flags[i - first] = keep ? 1 : 0;
        a.newconf[i - first] = nc;
        kept += keep ? 1 : 0;
    }
    const int tot = block_sum_i(kept, s_w);
    if (threadIdx.x == 0) {
        a.block_counts[blockIdx.x] = tot;
        if (blockIdx.x == 0) {
            a.frame->countNext = count;  // snapshots for pass 2 (see FrameDev)
            a.frame->runsNext = a.frame->runs;
            if (a.maskID != 0 && !a.append)   // the compaction pass (next launch) accumulates this clean pass's box (append: on top of k_clean_runs')
                for (int q = 0; q < 6; ++q) a.frame->bbox_acc[q] = q < 3 ? kBBoxEmptyMin : kBBoxEmptyMax;
        }
    }
    if (a.append) {
        // the table entries pass 2 fills: new run j holds the output slots [count + j kRun, count + (j + 1) kRun)
        const int r0 = a.frame->runs, nnew = (ncand + kRun - 1) / kRun + 1;
        for (int j = blockIdx.x * 256 + threadIdx.x; j < nnew; j += gridDim.x * 256) {
            a.dst.box[kBoxStride * (r0 + j)] = make_int4(kBoxEmptyMin, kBoxEmptyMin, kBoxEmptyMin, kBoxEmptyMax);
            a.dst.box[kBoxStride * (r0 + j) + 1] = make_int4(kBoxEmptyMax, kBoxEmptyMax, kBoxEmptyMax, count + j * kRun);
            a.dst.box[kBoxStride * (r0 + j) + 2] = make_int4(0, kBoxEmptyMin, 0, 0);
        }
    }
}

__global__ __launch_bounds__(256) void k_clean_small_flags(const CleanArgs a) { clean_small_flags_body(a); }

__device__ __forceinline__ void clean_small_compact_body(const CleanArgs& a) {
    __shared__ int s_w[4];
    __shared__ int s_bb[6];
    // Model::lastBoundingBox of an OBJECT model (Model.cpp:315-345 + draw_global_surface.vert:55-78: the box of the surfels the GUI draws --
    // confidence above the model's threshold -- in millimetres, truncated): accumulated here, where the frame's final records pass through
    // registers anyway; Model::fuse of the NEXT frame limits its depth with it (Model.cpp:480-501).  The background (id 0) never uses one.
    const bool bbox_on = a.maskID != 0;
    int bmin[3] = {kBBoxEmptyMin, kBBoxEmptyMin, kBBoxEmptyMin}, bmax[3] = {kBBoxEmptyMax, kBBoxEmptyMax, kBBoxEmptyMax};
    if (bbox_on && threadIdx.x < 6) s_bb[threadIdx.x] = threadIdx.x < 3 ? kBBoxEmptyMin : kBBoxEmptyMax;
    const int count = a.frame->countNext;
    const int first = a.append ? count : 0;
    const int total = count + cand_count(a.W, a.H, a.frame->tick);
    const int r0 = a.frame->runsNext;     // (append) the table entry of the first new run
    const float time = (float)a.frame->tick;
    const int chunk = chunk_size(total - first);
    const int beg = first + blockIdx.x * chunk, end = min(total, beg + chunk);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int base = 0;
    for (int i0 = beg; i0 < end; i0 += 256) {
        // every load of the slice is issued before anything depends on one of them (the keep flag, the record, the new
        // confidence, and -- first slice -- the workgroup's base offset): one memory latency per slice instead of three
        const int i = i0 + threadIdx.x;
        const bool in = i < end;
        uint8_t flag = 0;
        float nc = 0.f;
        float4 pc = make_float4(0, 0, 0, 0), ct = pc, nr = pc;
        if (in) {
            flag = a.flags[i - first];
            nc = a.newconf[i - first];
            if (i < count) { pc = a.src.pc[i]; ct = a.src.ct[i]; nr = a.src.nr[i]; }
            else { const int c = i - count; pc = a.cand_rec[c * 3 + 0]; ct = a.cand_rec[c * 3 + 1]; nr = a.cand_rec[c * 3 + 2]; }
        }
        if (i0 == beg) base = block_base(a.block_counts, s_w) + first;
        const bool keep = in && flag;
        const unsigned long long m = __ballot(keep);
        if (lane == 0) s_w[wave] = __popcll(m);
        __syncthreads();
        int off = base;
        for (int w = 0; w < wave; ++w) off += s_w[w];
        const int tot = s_w[0] + s_w[1] + s_w[2] + s_w[3];
        const int o = off + lane_rank(m);
        const bool written = keep && o < a.dst.cap;
        if (written) {
            pc.w = nc;
            if (ct.w == -2.f) ct.w = time;  // copy_unstable.vert:131
            a.dst.pc[o] = pc; a.dst.ct[o] = ct; a.dst.nr[o] = nr;
            if (bbox_on && pc.w > a.confThreshold) {   // draw_global_surface.vert:55 (unstable == 0), :69-78
                const int x = (int)(1000.f * pc.x), y = (int)(1000.f * pc.y), z = (int)(1000.f * pc.z);
                bmin[0] = min(bmin[0], x); bmin[1] = min(bmin[1], y); bmin[2] = min(bmin[2], z);
                bmax[0] = max(bmax[0], x); bmax[1] = max(bmax[1], y); bmax[2] = max(bmax[2], z);
            }
        }
        if (a.append) {
            // the appended surfels' runs: a wavefront's survivors are consecutive output slots, in at most two runs
            const int j = written ? (o - count) / kRun : 0x7FFFFFFF;
            const int j0 = wave_min_i(j);
            if (j0 != 0x7FFFFFFF) {
                for (int g = 0; g < 2; ++g) {
                    const bool mine = written && j == j0 + g;
                    const unsigned long long mm = __ballot(mine);
                    if (mm == 0ull) continue;
                    RunAcc acc;
                    acc.reset();
                    if (mine) acc.add(pc, ct.w);
                    int v[8] = {acc.lo[0], acc.lo[1], acc.lo[2], acc.tmax, acc.hi[0], acc.hi[1], acc.hi[2], acc.cmin};
#pragma unroll
                    for (int q = 0; q < 8; ++q) v[q] = (q < 3 || q == 7) ? wave_min_i(v[q]) : wave_max_i(v[q]);
                    if (lane == 0) {
                        int* e = reinterpret_cast<int*>(&a.dst.box[kBoxStride * (r0 + j0 + g)]);
                        atomicMin(&e[0], v[0]); atomicMin(&e[1], v[1]); atomicMin(&e[2], v[2]); atomicMax(&e[3], v[3]);
                        atomicMax(&e[4], v[4]); atomicMax(&e[5], v[5]); atomicMax(&e[6], v[6]);
                        atomicAdd(&e[8], __popcll(mm)); atomicMin(&e[9], v[7]);
                    }
                }
            }
        }
        base += tot;
        __syncthreads();
    }
    if (bbox_on) {   // (the loop's barriers order the initialisation of s_bb before these; a workgroup without elements skips both)
        if (beg < end) {
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                if (bmin[q] != kBBoxEmptyMin) atomicMin(&s_bb[q], bmin[q]);
                if (bmax[q] != kBBoxEmptyMax) atomicMax(&s_bb[3 + q], bmax[q]);
            }
            __syncthreads();
            if (threadIdx.x < 3 && s_bb[threadIdx.x] != kBBoxEmptyMin) atomicMin(&a.frame->bbox_acc[threadIdx.x], s_bb[threadIdx.x]);
            else if (threadIdx.x >= 3 && threadIdx.x < 6 && s_bb[threadIdx.x] != kBBoxEmptyMax) atomicMax(&a.frame->bbox_acc[threadIdx.x], s_bb[threadIdx.x]);
        }
    }
    if (blockIdx.x != gridDim.x - 1) return;
    if (beg >= end) base = block_base(a.block_counts, s_w) + first;   // the last workgroup owns no elements: all threads take part
    if (threadIdx.x == 0) {
        const int end_slot = min(base, a.dst.cap);
        if (a.append) {
            // (nobody else reads these during the launch: the other workgroups work from the snapshots)
            const int n = a.frame->count + (end_slot - count);
            a.frame->count = n;
            a.frame->phys = end_slot;
            a.frame->runs = r0 + (end_slot - count + kRun - 1) / kRun;
            // the first live surfel: runs only lose surfels, so the first non-empty run moves forward only; if every old run is empty it is the first
            // appended surfel (an empty map has none: the value is never used)
            int fr = a.frame->first_run;
            while (fr < r0 && run_len(a.dst.box, fr) == 0) ++fr;
            a.frame->first_run = fr;
            a.frame->first = fr < r0 ? run_start(a.dst.box, fr) : count;
            if (a.host_count) *a.host_count = n;
            if (a.host_append) *a.host_append = append_mirror(a.seq, r0 + (end_slot - count + kRun - 1) / kRun, end_slot);
        } else {
            a.frame->count = a.frame->phys = end_slot;
            a.frame->runs = 0;   // a dense buffer; this form writes no run table
            a.frame->first = 0; a.frame->first_run = 0;
            if (a.host_count) *a.host_count = end_slot;
        }
    }
}

__global__ __launch_bounds__(256) void k_clean_small_compact(const CleanArgs a) { clean_small_compact_body(a); }

// ------------------------------------------------------------------------------------------------
// clean, IN PLACE (round 6): Model::clean of a big map as O(surfels its tests can change), not O(N).
// The reference streams the whole buffer through transform feedback every frame (Model.cpp:649-772: N x 48 B read, N' x 48 B written), and so did
// rounds 1-5 here (round 5: one launch with a decoupled look-back, 3.6 GB moved in 1.28 ms on the 26.9 M-surfel map of configs[4], >= 78 % of it
// copies of surfels no test of copy_unstable.vert:53-157 can touch).  What the pass can do to a surfel of the buffer:
//   (a) drop it by the window rules (:77-106)          -- only a surfel inside the image, in front of the camera, seen within timeDelta;
//   (b) drop it by the age rule (:118-125)             -- only an UNSTABLE surfel (confidence below the threshold) last seen within timeDelta;
//   (c) lower its confidence, mask disagreement (:139-156) -- a surfel whose texel (clamped to the image border when it projects outside) carries a
//       foreign mask value and a filtered depth within 5 cm of its own.
// The buffer is kept as RUNS (Surfels::box): k_cull_clean lists the runs in which (a), (b) or (c) can apply at all -- from the run's box, its newest
// time stamp, its lowest confidence, and the frame's border / foreign-mask depth ranges (ResolveOut::decay_stats) -- and k_clean_runs visits only
// those: tests every surfel, and where a run loses surfels moves its survivors up INSIDE the run (their order is kept; the slots behind them stay
// unused).  The order of the surfels -- the reference's transform-feedback order, which decides z-test ties and which surfel is vertex 0 -- is the
// order of the runs; nothing ever moves between runs, so no workgroup waits for another (the round-5 look-back and its ticket order are gone).
// The frame's new surfels are appended behind the last run by the two-launch form above (append = 1).  When the slots behind the last run, or
// the table, run out, the host compacts the buffer (k_run_offsets + k_densify, mf_frame.inl) -- a copy of the live surfels, every few dozen frames.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_cull_clean(Surfels s, const FrameDev* __restrict__ frame, const PoseDev* __restrict__ pose, int W, int H, Intr k,
                                                    int timeDelta, float confThreshold, int* __restrict__ decay_stats, int* __restrict__ list,
                                                    int* __restrict__ count, int* __restrict__ ctl) {
    const int runs = frame->runs;
    const float time = (float)frame->tick;
    const int r = blockIdx.x * 256 + threadIdx.x;
    // what the frame holds for rule (c): filtered depth of the border texels with a foreign mask, side by side; lowest filtered depth of ANY foreign texel
    int e_side[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) e_side[q] = decay_stats[q];
    const int e_amin = decay_stats[8];
    const bool has_any = e_amin != kBoxEmptyMin;
    bool visit = false;
    if (r < runs) {
        const int4 a = s.box[kBoxStride * r], b = s.box[kBoxStride * r + 1], c = s.box[kBoxStride * r + 2];
        if (c.x > 0) {
            const bool recent = !(time - box_dec(a.w) > (float)timeDelta);
            if (c.y != kBoxEmptyMin && box_dec(c.y) < confThreshold && recent) visit = true;              // (b)
            else if (a.x <= b.x && a.y <= b.y && a.z <= b.z) {     // (a run of NaN positions only: every comparison of (a) and (c) fails)
                float zlo, zhi;
                int outside = 0;
                if (run_box_in_frustum(a, b, pose, W, H, k, INFINITY, zlo, zhi, &outside)) visit = true;  // (a), and (c) inside the image
                else if (zlo > 0.06f) {
                    // In front of the camera and outside the image: every surfel's texel is clamped to the border.  A box that lies beyond the LEFT
                    // edge as a whole puts all of them into the left column (any row), and so on: rule (c) needs a foreign texel THERE whose depth is
                    // within 5 cm of a surfel's -- one side that rules it out is enough.
                    visit = true;
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if ((outside >> q) & 1) {
                            const bool has = e_side[2 * q] != kBoxEmptyMin;
                            if (!has || box_dec(e_side[2 * q + 1]) <= zlo - 0.06f || box_dec(e_side[2 * q]) >= zhi + 0.06f) visit = false;
                        }
                    if (outside == 0) visit = has_any;      // (outside by the near plane only: cannot happen with zlo > 0 -- kept conservative)
                } else if (zhi < -0.06f)     // behind the camera: any texel, but only a NEGATIVE filtered depth is within 5 cm (the filter writes none)
                    visit = has_any && box_dec(e_amin) < zhi + 0.06f;
                else visit = has_any;        // around the camera plane: any texel, depths around zero
            }
        }
    }
    run_list_append(visit, r, list, count, ctl);
    if (run_list_finish(count, ctl)) {       // every workgroup has read the statistics: armed for the next frame's resolve pass
        for (int q = 0; q < kDecayStats; ++q)
            __hip_atomic_store(&decay_stats[q], (q == 8 || (q & 1) == 0) ? kBoxEmptyMin : kBoxEmptyMax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
__global__ void k_arm_decay_stats(int* st) {
    if (threadIdx.x == 0 && blockIdx.x == 0)
        for (int q = 0; q < kDecayStats; ++q) st[q] = (q == 8 || (q & 1) == 0) ? kBoxEmptyMin : kBoxEmptyMax;
}
void launch_arm_decay_stats(int* stats, hipStream_t st) { hipLaunchKernelGGL(k_arm_decay_stats, dim3(1), dim3(64), 0, st, stats); }
void launch_cull_clean(Surfels s, const FrameDev* frame, const PoseDev* pose, int W, int H, Intr k, int timeDelta, float confThreshold, int* decay_stats,
                       int* list, int* count, int* ctl, int max_runs, hipStream_t st) {
    hipLaunchKernelGGL(k_cull_clean, dim3((max_runs + 255) / 256), dim3(256), 0, st, s, frame, pose, W, H, k, timeDelta, confThreshold, decay_stats, list,
                       count, ctl);
}

// One run per workgroup and round (its <= kRun slots: two per thread); the runs are dealt round-robin over the grid (the listed runs come in no
// particular order, expensive -- in view -- and cheap ones mixed).
__device__ __forceinline__ void clean_runs_body(const CleanArgs& a) {
    __shared__ int s_cnt[2][4];
    __shared__ int s_red[4][8];
    __shared__ int s_bb[6];
    static_assert(kRun == 512, "two slots per thread");
    // Model::lastBoundingBox of an OBJECT model: see clean_small_compact_body (here: the box of the buffer's own survivors; the append pass adds
    // the new surfels').  An object model's launch therefore visits EVERY run.
    const bool bbox_on = a.maskID != 0;
    int bmin[3] = {kBBoxEmptyMin, kBBoxEmptyMin, kBBoxEmptyMin}, bmax[3] = {kBBoxEmptyMax, kBBoxEmptyMax, kBBoxEmptyMax};
    if (bbox_on && threadIdx.x < 6) s_bb[threadIdx.x] = threadIdx.x < 3 ? kBBoxEmptyMin : kBBoxEmptyMax;
    const int nlist = a.run_list ? *a.run_count : a.frame->runs;     // (frame->runs / count change behind this launch only)
    const float time = (float)a.frame->tick;
    float Ri[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) Ri[q] = a.pose->Ri[q];
    const float3 ti = f3(a.pose->ti[0], a.pose->ti[1], a.pose->ti[2]);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int wg_died = 0;          // surfels this workgroup dropped (thread 0's copy is the one that counts)
    for (int v = blockIdx.x; v < nlist; v += gridDim.x) {
        const int r = a.run_list ? a.run_list[v] : v;
        const int start = run_start(a.src.box, r), len = run_len(a.src.box, r);
        // The test needs HALF of a record -- position + confidence, the two time stamps: 24 of its 48 bytes; the normal / radius record only for a
        // surfel in view (clean_test fetches it there), the rest only for a survivor that moves.  Slots threadIdx.x and 256 + threadIdx.x of the run.
        const int off0 = (int)threadIdx.x, off1 = 256 + (int)threadIdx.x;
        const bool live0 = off0 < len, live1 = off1 < len;
        float4 pc0 = make_float4(0, 0, 0, 0), pc1 = pc0;
        float2 tm0 = make_float2(0, 0), tm1 = tm0;
        if (live0) { pc0 = a.src.pc[start + off0]; tm0 = *reinterpret_cast<const float2*>(reinterpret_cast<const float*>(&a.src.ct[start + off0]) + 2); }
        if (live1) { pc1 = a.src.pc[start + off1]; tm1 = *reinterpret_cast<const float2*>(reinterpret_cast<const float*>(&a.src.ct[start + off1]) + 2); }
        // (Requesting every load of an element's test at once -- normal / radius, the nine window records, the mask texel: 135 registers -- changed
        // nothing on the configs[4] maps, 0.43 ms either way: the pass is bound by the instructions of the test, ~2 000 per surfel in view.)
        float nc0 = 0.f, nc1 = 0.f;
        int dk = 0;
        const bool keep0 = live0 && clean_test(a, pc0, make_float4(0.f, 0.f, tm0.x, tm0.y), make_float4(0, 0, 0, 0), time, Ri, ti, nc0, dk, &a.src.nr[start + off0]);
        const bool keep1 = live1 && clean_test(a, pc1, make_float4(0.f, 0.f, tm1.x, tm1.y), make_float4(0, 0, 0, 0), time, Ri, ti, nc1, dk, &a.src.nr[start + off1]);
        RunAcc acc;
        acc.reset();
        if (keep0) acc.add(make_float4(pc0.x, pc0.y, pc0.z, nc0), tm0.y);
        if (keep1) acc.add(make_float4(pc1.x, pc1.y, pc1.z, nc1), tm1.y);
        const unsigned long long m0 = __ballot(keep0), m1 = __ballot(keep1);
        if (lane == 0) { s_cnt[0][wave] = __popcll(m0); s_cnt[1][wave] = __popcll(m1); }
        int o0 = lane_rank(m0), o1 = lane_rank(m1);
        run_box_reduce(acc, s_red);
        __syncthreads();
        const int c0 = s_cnt[0][0] + s_cnt[0][1] + s_cnt[0][2] + s_cnt[0][3];
        const int kept = c0 + s_cnt[1][0] + s_cnt[1][1] + s_cnt[1][2] + s_cnt[1][3];
        for (int w = 0; w < wave; ++w) { o0 += s_cnt[0][w]; o1 += s_cnt[1][w]; }
        o1 += c0;
        const bool conf0 = keep0 && __float_as_int(nc0) != __float_as_int(pc0.w), conf1 = keep1 && __float_as_int(nc1) != __float_as_int(pc1.w);
        if (kept != len) {
            // the run lost surfels: its survivors behind the first hole move up, whole records, in order
            const bool mv0 = keep0 && o0 != off0, mv1 = keep1 && o1 != off1;
            float4 c40 = make_float4(0, 0, 0, 0), n40 = c40, c41 = c40, n41 = c40;
            if (mv0) { c40 = a.src.ct[start + off0]; n40 = a.src.nr[start + off0]; }
            if (mv1) { c41 = a.src.ct[start + off1]; n41 = a.src.nr[start + off1]; }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every record that moves is in registers before any slot of the run is rewritten
            __syncthreads();
            if (mv0) { a.src.pc[start + o0] = make_float4(pc0.x, pc0.y, pc0.z, nc0); a.src.ct[start + o0] = c40; a.src.nr[start + o0] = n40; }
            else if (conf0) reinterpret_cast<float*>(&a.src.pc[start + off0])[3] = nc0;
            if (mv1) { a.src.pc[start + o1] = make_float4(pc1.x, pc1.y, pc1.z, nc1); a.src.ct[start + o1] = c41; a.src.nr[start + o1] = n41; }
            else if (conf1) reinterpret_cast<float*>(&a.src.pc[start + off1])[3] = nc1;
        } else {     // nobody moves: only a decayed confidence is written
            if (conf0) reinterpret_cast<float*>(&a.src.pc[start + off0])[3] = nc0;
            if (conf1) reinterpret_cast<float*>(&a.src.pc[start + off1])[3] = nc1;
        }
        if (threadIdx.x == 0) {
            run_box_store(a.src.box, r, start, kept, s_red);
            wg_died += len - kept;
        }
        if (bbox_on) {   // draw_global_surface.vert:55 (unstable == 0), :69-78
            if (keep0 && nc0 > a.confThreshold) {
                const int x = (int)(1000.f * pc0.x), y = (int)(1000.f * pc0.y), z = (int)(1000.f * pc0.z);
                bmin[0] = min(bmin[0], x); bmin[1] = min(bmin[1], y); bmin[2] = min(bmin[2], z);
                bmax[0] = max(bmax[0], x); bmax[1] = max(bmax[1], y); bmax[2] = max(bmax[2], z);
            }
            if (keep1 && nc1 > a.confThreshold) {
                const int x = (int)(1000.f * pc1.x), y = (int)(1000.f * pc1.y), z = (int)(1000.f * pc1.z);
                bmin[0] = min(bmin[0], x); bmin[1] = min(bmin[1], y); bmin[2] = min(bmin[2], z);
                bmax[0] = max(bmax[0], x); bmax[1] = max(bmax[1], y); bmax[2] = max(bmax[2], z);
            }
        }
        __syncthreads();   // s_cnt / s_red are rewritten by the next round
    }
    if (bbox_on) {
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            if (bmin[q] != kBBoxEmptyMin) atomicMin(&s_bb[q], bmin[q]);
            if (bmax[q] != kBBoxEmptyMax) atomicMax(&s_bb[3 + q], bmax[q]);
        }
        __syncthreads();
        // (RETURNING atomics whose results are waited for: the "finished" ticket below must not overtake them -- a fire-and-forget atomic is not
        // ordered against a later one to another address)
        int seen = 0;
        if (threadIdx.x < 3 && s_bb[threadIdx.x] != kBBoxEmptyMin) seen = atomicMin(&a.frame->bbox_tmp[threadIdx.x], s_bb[threadIdx.x]);
        else if (threadIdx.x >= 3 && threadIdx.x < 6 && s_bb[threadIdx.x] != kBBoxEmptyMax) seen = atomicMax(&a.frame->bbox_tmp[threadIdx.x], s_bb[threadIdx.x]);
        if (threadIdx.x < 6) s_bb[threadIdx.x] = seen;      // (consumes the returned values)
    }
    // The last workgroup to FINISH installs the launch's results: one 64-bit atomic per workgroup carries both its "finished" ticket and the number
    // of surfels it dropped -- the value travels IN the atomic, so the workgroup that draws the last ticket holds the launch's total without any
    // ordering between workgroups.
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long* done64 = reinterpret_cast<unsigned long long*>(a.ctl + 2);
        const unsigned long long old = atomicAdd(done64, (1ull << 32) | (unsigned long long)(unsigned)wg_died);
        if ((unsigned)(old >> 32) == gridDim.x - 1u) {
            const int n = a.frame->count - ((int)(unsigned)(old & 0xFFFFFFFFull) + wg_died);
            a.frame->count = n;
            if (a.host_count) *a.host_count = n;
            if (bbox_on) {   // the box of a frame is the box of its LAST clean pass (the reference's render pass sees the final buffer)
                for (int q = 0; q < 6; ++q) {
                    a.frame->bbox_acc[q] = __hip_atomic_load(&a.frame->bbox_tmp[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(&a.frame->bbox_tmp[q], q < 3 ? kBBoxEmptyMin : kBBoxEmptyMax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            __hip_atomic_store(done64, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

__global__ __launch_bounds__(256) void k_clean_runs(const CleanArgs a) { clean_runs_body(a); }

// workgroups of a k_clean_runs launch for a buffer of ~`elements` surfels: the runs are dealt round-robin, any grid covers any table
int clean_runs_grid(long elements) {
    const long runs = (elements + kRun - 1) / kRun;
    return (int)(runs < 64 ? 64 : (runs > kCleanGridMax ? kCleanGridMax : runs));
}

// ------------------------------------------------------------------------------------------------
// compaction of a sparse buffer: src's runs -> dst, dense (slots [0, count), no table).  Host-driven (mf_frame.inl: densify), when the slots
// behind the last run or the table entries run out, before a download, before a model's passes return to the small-map forms.
// ------------------------------------------------------------------------------------------------
// offs[r] = live surfels of the runs before run r (offs[runs] = all of them): one workgroup
__global__ __launch_bounds__(1024) void k_run_offsets(Surfels s, const FrameDev* __restrict__ frame, int* __restrict__ offs) {
    __shared__ int s_w[16];
    const int runs = frame->runs;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int carry = 0;
    for (int base = 0; base < runs; base += 1024) {
        const int r = base + (int)threadIdx.x;
        const int v = r < runs ? run_len(s.box, r) : 0;
        int x = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int y = __shfl_up(x, d, 64);
            if (lane >= d) x += y;
        }
        if (lane == 63) s_w[wave] = x;
        __syncthreads();
        int before = 0, total = 0;
        for (int w = 0; w < 16; ++w) { if (w < wave) before += s_w[w]; total += s_w[w]; }
        if (r < runs) offs[r] = carry + before + x - v;
        carry += total;
        __syncthreads();
    }
    if (threadIdx.x == 0) offs[runs] = carry;
}
__global__ __launch_bounds__(256) void k_densify(Surfels src, Surfels dst, const FrameDev* __restrict__ frame, const int* __restrict__ offs) {
    const int runs = frame->runs;
    for (int r = blockIdx.x; r < runs; r += gridDim.x) {
        const int start = run_start(src.box, r), len = run_len(src.box, r), o = offs[r];
        for (int q = threadIdx.x; q < len; q += 256) {
            if (o + q >= dst.cap) break;
            dst.pc[o + q] = src.pc[start + q]; dst.ct[o + q] = src.ct[start + q]; dst.nr[o + q] = src.nr[start + q];
        }
    }
}
__global__ void k_densify_finish(FrameDev* __restrict__ frame, const int* __restrict__ offs, int cap, int* __restrict__ host_count) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int n = min(offs[frame->runs], cap);
    frame->count = frame->phys = n;
    frame->runs = 0; frame->first = 0; frame->first_run = 0;
    if (host_count) *host_count = n;
}
void launch_densify(Surfels src, Surfels dst, FrameDev* frame, int* offs, int* host_count, hipStream_t st) {
    hipLaunchKernelGGL(k_run_offsets, dim3(1), dim3(1024), 0, st, src, frame, offs);
    hipLaunchKernelGGL(k_densify, dim3(2048), dim3(256), 0, st, src, dst, frame, offs);
    hipLaunchKernelGGL(k_densify_finish, dim3(1), dim3(64), 0, st, frame, offs, dst.cap, host_count);
}

// ------------------------------------------------------------------------------------------------
// splat prediction: scatter (per-surfel sprite loop, ray-disc test, 64-bit atomicMin) + resolve
// Raster rule: sprite side s centred on (u,v) covers pixel (px,py) iff u - s/2 <= px + 0.5 < u + s/2; LESS on the
// corrected z; lower index wins ties; sprites wider than 64 px are clamped.
// ------------------------------------------------------------------------------------------------

// kLanes neighbouring lanes share one surfel and split its sprite's pixels between them as a kLX x kLY block (1: the thread-per-surfel form).
// The object models' launches use 4: a few thousand sprites of 4-10 px a side kept a handful of threads busy for ~36 us each (round 4 trace),
// the rest of the GPU idle; the keys are the same bits in any split (atomicMin is order independent).
template <int kLanes>
__device__ __forceinline__ void splat_scatter_body(Surfels src, const FrameDev* __restrict__ frame,
                                                   const PoseDev* __restrict__ pose, int W, int H, Intr k, float maxDepth,
                                                   float confThreshold, int timeDelta, unsigned long long* __restrict__ keys, bool pretest = false) {
    const float time = (float)frame->tick;  // combinedPredict(time = tick, maxTime = tick)
    float Ri[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) Ri[q] = pose->Ri[q];
    const float3 ti = f3(pose->ti[0], pose->ti[1], pose->ti[2]);
    constexpr int kLX = kLanes >= 2 ? 2 : 1, kLY = kLanes / kLX;
    const int sub = threadIdx.x % kLanes, sx = sub % kLX, sy = sub / kLX;
    for_each_surfel_slice<kLanes>(src, frame, nullptr, nullptr, [&](int i, bool live) {
        if (!live) return;
        const float4 pc = src.pc[i];
        if (pc.w < confThreshold) return;
        const float lastTime = src.ct[i].w;
        const float3 h = mul33(Ri, f3(pc.x, pc.y, pc.z)) + ti;
        if (h.z > maxDepth || h.z < 0 || time - lastTime > (float)timeDelta || lastTime > time) return;  // splat.vert:58
        const float u = ((k.fx * h.x) / h.z) + k.cx, v = ((k.fy * h.y) / h.z) + k.cy;
        if (!(u >= 0.f && u <= (float)W && v >= 0.f && v <= (float)H)) return;
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
        if (!(size > 0.f)) return;
        size = fminf(fmaxf(size, 1.0f), 64.0f);   // GL clamps gl_PointSize to the point size range: at least 1 px (see mf_splat.hip)
        const float half = size * 0.5f;
        const int px0 = max(0, (int)ceilf(u - half - 0.5f)), px1 = min(W - 1, (int)ceilf(u + half - 0.5f) - 1);
        const int py0 = max(0, (int)ceilf(v - half - 0.5f)), py1 = min(H - 1, (int)ceilf(v + half - 0.5f) - 1);
        const float sqrRad = rad * rad;
        const float pn = dot3(h, nrm);
        for (int py = py0 + sy; py <= py1; py += kLY) {
            for (int px = px0 + sx; px <= px1; px += kLX) {
                const float fcx = (float)px + 0.5f, fcy = (float)py + 0.5f;
                const float3 l = normalize_gl(f3((fcx - k.cx) / k.fx, (fcy - k.cy) / k.fy, 1.0f));
                const float3 cp = l * (pn / dot3(l, nrm));
                const float3 diff = cp - h;
                if (!(dot3(diff, diff) <= sqrRad)) continue;
                if (!(cp.z > 0.f)) continue;
                const unsigned long long key = ((unsigned long long)__float_as_uint(cp.z) << 32) | (unsigned)i;
                if (pretest) zmin_key_pretested(&keys[py * W + px], key);   // (the object models' launches)
                else zmin_key(&keys[py * W + px], key);
            }
        }
    });
}

__global__ __launch_bounds__(256) void k_splat_scatter(Surfels src, const FrameDev* __restrict__ frame,
                                                       const PoseDev* __restrict__ pose, int W, int H, Intr k, float maxDepth,
                                                       float confThreshold, int timeDelta, unsigned long long* __restrict__ keys) {
    splat_scatter_body<1>(src, frame, pose, W, H, k, maxDepth, confThreshold, timeDelta, keys);
}

void launch_splat_scatter(Surfels src, const FrameDev* frame, const PoseDev* pose, int W, int H, Intr k, float maxDepth,
                          float confThreshold, int timeDelta, unsigned long long* keys, hipStream_t s, int blocks) {
    hipLaunchKernelGGL(k_splat_scatter, dim3(blocks), dim3(256), 0, s, src, frame, pose, W, H, k, maxDepth, confThreshold,
                       timeDelta, keys);
}

__device__ __forceinline__ void splat_resolve_body(Surfels src, const PoseDev* __restrict__ pose,
                                                   unsigned long long* __restrict__ keys, int W, int H, Intr k,
                                                   float4* __restrict__ predV, float4* __restrict__ predN,
                                                   uchar4* __restrict__ predImage, uint16_t* __restrict__ predTime,
                                                   FrameDev* __restrict__ frame, const uint8_t* __restrict__ rgb,
                                                   uint8_t* __restrict__ predGray, uint8_t* __restrict__ fillGray, int fillPassthrough) {
    const int px = blockIdx.x * 64 + (threadIdx.x & 63);
    const int py = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (px >= W || py >= H) return;
    const int p = py * W + px;
    const unsigned long long key = keys[p];
    keys[p] = kEmptyKey;
    if (key == kEmptyKey) {
        predV[p] = predN[p] = make_float4(0, 0, 0, 0);
        predImage[p] = make_uchar4(0, 0, 0, 0);
        predTime[p] = 0;
        // intensity images for the photometric term of the NEXT tracking step (imageBGRToIntensity of the RGB projection,
        // and of the fill-in image: fill_rgb.frag takes the raw frame where the projection is empty)
        if (predGray) predGray[p] = 0;
        if (fillGray && rgb) fillGray[p] = intensity_of((float)rgb[p * 3], (float)rgb[p * 3 + 1], (float)rgb[p * 3 + 2]);
        return;
    }
    const int i = (int)(unsigned)(key & 0xFFFFFFFFull);
    const float z = __uint_as_float((unsigned)(key >> 32));
    const float4 pc = src.pc[i], c4 = src.ct[i], n4 = src.nr[i];
    const float3 n = normalize_gl(mul33(pose->Ri, f3(n4.x, n4.y, n4.z)));
    const float fcx = (float)px + 0.5f, fcy = (float)py + 0.5f;
    predV[p] = make_float4((fcx - k.cx) * z * (1.f / k.fx), (fcy - k.cy) * z * (1.f / k.fy), z, pc.w);  // combo_splat.frag:56
    predN[p] = make_float4(n.x, n.y, n.z, n4.w);
    const int ci = (int)c4.x;
    const uchar4 col = make_uchar4((ci >> 16) & 0xFF, (ci >> 8) & 0xFF, ci & 0xFF, 255);
    predImage[p] = col;
    predTime[p] = (uint16_t)(unsigned)c4.z;
    if (predGray || fillGray) {
        const uint8_t gv = intensity_of((float)col.x, (float)col.y, (float)col.z);
        if (predGray) predGray[p] = gv;
        if (fillGray) {
            // fill_rgb.frag:31-34: the raw frame where the projection is empty -- or everywhere with `passthrough` (frameToFrameRGB, Model.cpp:981)
            const bool empty = (col.x == 0 && col.y == 0 && col.z == 0) || fillPassthrough != 0;
            fillGray[p] = (empty && rgb) ? intensity_of((float)rgb[p * 3], (float)rgb[p * 3 + 1], (float)rgb[p * 3 + 2]) : gv;
        }
    }
    // MaskFusion::requiresFillIn (MaskFusion.cpp:630-648): nearest sample of the 20x down-sampled colour prediction
    if ((px % 20) == 10 && (py % 20) == 10 && px / 20 < W / 20 && py / 20 < H / 20 && col.x > 0 && col.y > 0 && col.z > 0)
        atomicAdd(&frame->cover, 1);
}

__global__ __launch_bounds__(256) void k_splat_resolve(Surfels src, const PoseDev* __restrict__ pose,
                                                       unsigned long long* __restrict__ keys, int W, int H, Intr k,
                                                       float4* __restrict__ predV, float4* __restrict__ predN,
                                                       uchar4* __restrict__ predImage, uint16_t* __restrict__ predTime,
                                                       FrameDev* __restrict__ frame, const uint8_t* __restrict__ rgb,
                                                       uint8_t* __restrict__ predGray, uint8_t* __restrict__ fillGray, int fillPassthrough) {
    splat_resolve_body(src, pose, keys, W, H, k, predV, predN, predImage, predTime, frame, rgb, predGray, fillGray, fillPassthrough);
}

void launch_splat_resolve(Surfels src, const PoseDev* pose, unsigned long long* keys, int W, int H, Intr k, float4* predV,
                          float4* predN, uchar4* predImage, uint16_t* predTime, FrameDev* frame, const uint8_t* rgb,
                          uint8_t* predGray, uint8_t* fillGray, hipStream_t s, int fillPassthrough) {
    dim3 grid((W + 63) / 64, (H + 3) / 4);
    hipLaunchKernelGGL(k_splat_resolve, grid, dim3(256), 0, s, src, pose, keys, W, H, k, predV, predN, predImage, predTime, frame, rgb,
                       predGray, fillGray, fillPassthrough);
}

// ------------------------------------------------------------------------------------------------
// end of frame: tick++ and the fill-in decision for the next tracking step
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void frame_advance_body(FrameDev* frame, int W, int H, FrameDev* host_mirror, const PoseDev* pose, const PoseDev* bg_pose,
                                                   float* log_slot) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (log_slot) pose_log_entry(pose, bg_pose, log_slot);
    const int rw = W / 20, rh = H / 20;
    frame->pad[0] = frame->useFillIn;  // decision the tracking step of THIS frame ran with (mf_get_last_fillin)
    frame->useFillIn = ((float)frame->cover / (float)(rw * rh) < 0.75f) ? 1 : 0;
    frame->cover = 0;
    frame->tick += 1;
    MF_FRAME_BBOX_ADVANCE(frame);
    if (host_mirror) *host_mirror = *frame;
}
__global__ void k_frame_advance(FrameDev* frame, int W, int H, FrameDev* host_mirror, const PoseDev* pose, const PoseDev* bg_pose,
                                float* log_slot) {
    frame_advance_body(frame, W, H, host_mirror, pose, bg_pose, log_slot);
}
void launch_frame_advance(FrameDev* frame, int W, int H, FrameDev* host_mirror, const PoseDev* pose, const PoseDev* bg_pose,
                          float* log_slot, hipStream_t s) {
    hipLaunchKernelGGL(k_frame_advance, dim3(1), dim3(64), 0, s, frame, W, H, host_mirror, pose, bg_pose, log_slot);
}

// the pose-log entry alone (MaskFusion.cpp:580-596): what the frame advance writes, as a launch of its own behind a captured frame
__global__ void k_pose_log(const PoseDev* __restrict__ pose, const PoseDev* __restrict__ bg_pose, float* __restrict__ slot) {
    if (threadIdx.x == 0 && blockIdx.x == 0) pose_log_entry(pose, bg_pose, slot);
}
void launch_pose_log(const PoseDev* pose, const PoseDev* bg_pose, float* slot, hipStream_t s) {
    hipLaunchKernelGGL(k_pose_log, dim3(1), dim3(64), 0, s, pose, bg_pose, slot);
}

static CleanArgs clean_args(const CleanIn& in, Surfels src, Surfels dst) {
    CleanArgs a;
    a.transposed = in.transposed ? 1 : 0;
    a.literal = in.literalWindow ? 1 : 0;
    a.src = src; a.dst = dst; a.frame = in.frame; a.pose = in.pose; a.W = in.W; a.H = in.H; a.k = in.k; a.timeDelta = in.timeDelta;
    a.confThreshold = in.confThreshold; a.outlierCoeff = in.outlierCoeff; a.maskID = in.maskID; a.index = in.index; a.vc = in.vc; a.ct = in.ct;
    a.packed = in.packed;
    a.depthF = in.depthF; a.mask = in.mask; a.maskT = in.maskT; a.cand_op = in.cand_op; a.cand_rec = in.cand_rec;
    a.flags = in.flags; a.newconf = in.newconf; a.block_counts = in.block_counts; a.host_count = in.host_count;
    a.host_append = in.host_append; a.seq = in.seq;
    a.append = 0; a.run_list = nullptr; a.run_count = nullptr; a.ctl = nullptr;
    return a;
}
// Workgroups of the two ordered-compaction launches for `elements` elements (the host's last known count: any grid is correct).  A VGA map
// (0.75 M elements) is fastest on 2 048 (303.6 against 309-313 us per frame on 1 024 and 305.0 on 4 096); from ~1.5 M elements on, 1 024 workgroups
// with longer slices win: 1280 x 960 on its natural map 847 -> 835 us, the four 3.7 M-element object maps of configs[4] 4.16 -> 4.07-4.12 ms
// (profiles/r06zo_ab.txt, r06zp_ab.txt).
int compact_blocks_for(long elements) { return elements >= 1500000L ? kCompactBlocks / 2 : kCompactBlocks; }
void launch_clean_small(const CleanIn& in, Surfels src, Surfels dst, hipStream_t s, int blocks) {
    const CleanArgs a = clean_args(in, src, dst);
    hipLaunchKernelGGL(k_clean_small_flags, dim3(blocks), dim3(256), 0, s, a);
    hipLaunchKernelGGL(k_clean_small_compact, dim3(blocks), dim3(256), 0, s, a);
}
void launch_clean_runs(const CleanIn& in, Surfels buf, const VisList* runs, int* ctl, int blocks, hipStream_t s) {
    CleanArgs a = clean_args(in, buf, buf);
    a.run_list = runs ? runs->list : nullptr; a.run_count = runs ? runs->count : nullptr; a.ctl = ctl;
    hipLaunchKernelGGL(k_clean_runs, dim3(blocks), dim3(256), 0, s, a);
}
void launch_clean_append(const CleanIn& in, Surfels buf, hipStream_t s) {
    CleanArgs a = clean_args(in, buf, buf);
    a.append = 1;
    // (the candidates are <= P / 4 elements: a grid of one 256-element slice per workgroup, at most kCompactBlocks of them)
    hipLaunchKernelGGL(k_clean_small_flags, dim3(kCompactBlocks), dim3(256), 0, s, a);
    hipLaunchKernelGGL(k_clean_small_compact, dim3(kCompactBlocks), dim3(256), 0, s, a);
}

// ------------------------------------------------------------------------------------------------
// The surfel passes of every OBJECT model of a frame, one launch per pass (grid.z = model; mf_internal.h: ObjBatch).  Same bodies, same
// arguments as the model-by-model launches of enqueue_fuse_clean / enqueue_predict (mf_context.hip) -- the results are bit-identical
// (tests/test_gpu_multimodel.py::test_object_model_launch_switches_change_nothing) -- but a frame with M objects costs 9 launches
// instead of 9 M.  Core/MaskFusion.cpp:539-569 runs the models one after the other; nothing couples them.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_obj_index_scatter(const ObjBatch b) {
    const ObjPassArgs& m = b.m[blockIdx.z];
    index_scatter_body(m.a, m.frame, m.pose, b.W, b.H, b.k, b.maxDepthProcessed, b.timeDelta, m.keys, 0, nullptr, nullptr, true);
}
__global__ __launch_bounds__(256) void k_obj_index_resolve(const ObjBatch b) {   // (the object models' keys are row-major in both passes)
    const ObjPassArgs& m = b.m[blockIdx.z];
    index_resolve_same_body<false>(m.a, m.pose, m.keys, b.W * b.H, ResolveOut{m.index, m.ivc, m.inr, nullptr, nullptr, nullptr, nullptr, nullptr, m.frame, nullptr, 0});
}
__global__ __launch_bounds__(256) void k_obj_index_resolve_packed(const ObjBatch b) {   // the pass that feeds clean(): grid.x = 16 x 16-pixel tiles
    const ObjPassArgs& m = b.m[blockIdx.z];
    index_resolve_transposing_body<true>(b.updateCopy ? m.b : m.a, m.pose, m.keys, b.W, b.H,
                                         ResolveOut{nullptr, nullptr, nullptr, nullptr, m.iclean, b.depthF, b.mask, b.maskT, m.frame, nullptr, 0}, (int)blockIdx.x);
}
__global__ __launch_bounds__(256) void k_obj_fuse_data(const ObjBatch b) {
    const ObjPassArgs& m = b.m[blockIdx.z];
    const FuseDataArgs a{b.rgb, b.depthRaw, b.depthF, b.mask, m.maskID, m.frame, m.pose, m.weightMultiplier, m.fuseMaxDepth, b.bboxLimit, b.W, b.H, b.k,
                         m.index, m.ivc, m.inr, m.cand_op, m.cand_rec, m.upd_first, m.cand_best};
    fuse_data_body(a);
}
__global__ __launch_bounds__(256) void k_obj_fuse_update(const ObjBatch b) {
    const ObjPassArgs& m = b.m[blockIdx.z];
    fuse_update_body(m.a, m.frame, m.upd_first, m.cand_op, m.cand_best, m.cand_rec, b.W, b.H);
}
__global__ __launch_bounds__(256) void k_obj_fuse_update_copy(const ObjBatch b) {   // small models: a -> b with the second index scatter riding
    const ObjPassArgs& m = b.m[blockIdx.z];
    const IndexScatterArgs ix{m.pose, b.W, b.H, b.k, b.maxDepthProcessed, b.timeDelta, m.keys, 0};
    fuse_update_copy_body(m.a, m.b, m.frame, m.upd_first, m.cand_rec, ix);
}
__device__ __forceinline__ CleanArgs obj_clean_args(const ObjBatch& b, const ObjPassArgs& m, int append) {
    CleanArgs a;
    // copy-update: fuse copied a -> b, clean goes b -> a (the live buffer stays); in-place update + two-launch clean: clean goes a -> b;
    // in-place clean (cleanSmall == 0): everything happens in a
    a.src = b.updateCopy ? m.b : m.a; a.dst = b.updateCopy ? m.a : (b.cleanSmall ? m.b : m.a);
    a.frame = m.frame; a.pose = m.pose; a.W = b.W; a.H = b.H; a.k = b.k; a.timeDelta = b.timeDelta;
    a.confThreshold = m.confThreshold; a.outlierCoeff = b.outlierCoeff; a.maskID = m.maskID; a.transposed = 1; a.literal = b.cleanLiteral;
    a.index = m.index; a.vc = m.ivc; a.ct = nullptr; a.packed = m.iclean; a.depthF = b.depthF; a.mask = b.mask; a.maskT = b.maskT;
    a.cand_op = m.cand_op; a.cand_rec = m.cand_rec; a.flags = m.flags; a.newconf = m.newconf;
    a.block_counts = m.block_counts; a.host_count = m.host_count;
    a.host_append = m.host_append; a.seq = m.clean_seq;
    a.append = append; a.run_list = nullptr; a.run_count = nullptr; a.ctl = m.clean_ctl;
    return a;
}
__global__ __launch_bounds__(256) void k_obj_clean_runs(const ObjBatch b) { clean_runs_body(obj_clean_args(b, b.m[blockIdx.z], 0)); }
__global__ __launch_bounds__(256) void k_obj_clean_small_flags(const ObjBatch b) { clean_small_flags_body(obj_clean_args(b, b.m[blockIdx.z], b.cleanSmall ? 0 : 1)); }
__global__ __launch_bounds__(256) void k_obj_clean_small_compact(const ObjBatch b) { clean_small_compact_body(obj_clean_args(b, b.m[blockIdx.z], b.cleanSmall ? 0 : 1)); }
__global__ __launch_bounds__(256) void k_obj_splat_scatter(const ObjBatch b) {
    const ObjPassArgs& m = b.m[blockIdx.z];
    // (both forms write the same keys: the z-test is a minimum)
    if (b.denseSprites) splat_scatter_body<1>(m.a, m.frame, m.pose, b.W, b.H, b.k, b.maxDepthProcessed, m.confThreshold, b.timeDelta, m.keys, true);
    else splat_scatter_body<4>(m.a, m.frame, m.pose, b.W, b.H, b.k, b.maxDepthProcessed, m.confThreshold, b.timeDelta, m.keys, true);
}
__global__ __launch_bounds__(256) void k_obj_splat_resolve(const ObjBatch b) {
    const ObjPassArgs& m = b.m[blockIdx.z];
    splat_resolve_body(m.a, m.pose, m.keys, b.W, b.H, b.k, m.predV, m.predN, m.predImage, m.predTime, m.frame, b.rgb, m.predGray, nullptr, 0);
}
__global__ void k_obj_frame_advance(const ObjBatch b) {
    const ObjPassArgs& m = b.m[blockIdx.z];
    frame_advance_body(m.frame, b.W, b.H, m.host_frame, m.pose, b.bg_pose, m.log_slot);
}

void launch_obj_fuse_clean(const ObjBatch& b, int blocks, int clean_blocks, hipStream_t s, int compact_blocks) {
    const int P = b.W * b.H;
    const dim3 surfels(blocks, 1, b.n), pixels((P + 255) / 256, 1, b.n), compact(clean_blocks, 1, b.n);
    const dim3 cands(((b.W + 1) / 2 + 63) / 64, ((b.H + 1) / 2 + 3) / 4, b.n);
    hipLaunchKernelGGL(k_obj_index_scatter, surfels, dim3(256), 0, s, b);
    hipLaunchKernelGGL(k_obj_index_resolve, pixels, dim3(256), 0, s, b);
    hipLaunchKernelGGL(k_obj_fuse_data, cands, dim3(256), 0, s, b);
    if (b.updateCopy) {
        hipLaunchKernelGGL(k_obj_fuse_update_copy, surfels, dim3(256), 0, s, b);
    } else {
        hipLaunchKernelGGL(k_obj_fuse_update, dim3(cand_blocks(b.W, b.H), 1, b.n), dim3(256), 0, s, b);
        hipLaunchKernelGGL(k_obj_index_scatter, surfels, dim3(256), 0, s, b);   // predictIndices after fuse (MaskFusion.cpp:556): the same pass on the updated buffer
    }
    hipLaunchKernelGGL(k_obj_index_resolve_packed, dim3(resolve_tiles(b.W, b.H), 1, b.n), dim3(256), 0, s, b);
    // two-launch form src -> dst, or (big models) the buffer's own surfels in place, run by run -- an object model's launch visits every run: its
    // bounding box is the box of ALL its drawn surfels -- and then the frame's candidates appended by the two-launch form
    if (!b.cleanSmall) hipLaunchKernelGGL(k_obj_clean_runs, compact, dim3(256), 0, s, b);
    hipLaunchKernelGGL(k_obj_clean_small_flags, dim3(compact_blocks, 1, b.n), dim3(256), 0, s, b);
    hipLaunchKernelGGL(k_obj_clean_small_compact, dim3(compact_blocks, 1, b.n), dim3(256), 0, s, b);
}
void launch_obj_predict_advance(const ObjBatch& b, int blocks, hipStream_t s) {
    hipLaunchKernelGGL(k_obj_splat_scatter, dim3(blocks, 1, b.n), dim3(256), 0, s, b);
    hipLaunchKernelGGL(k_obj_splat_resolve, dim3((b.W + 63) / 64, (b.H + 3) / 4, b.n), dim3(256), 0, s, b);
    hipLaunchKernelGGL(k_obj_frame_advance, dim3(1, 1, b.n), dim3(64), 0, s, b);
}

}  // namespace mf

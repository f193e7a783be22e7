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

__device__ __forceinline__ int chunk_size(int n) {
    const int c = (n + kCompactBlocks - 1) / kCompactBlocks;
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
        frame->count = min(base, dst.cap);
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

// vis_list == nullptr: every surfel of the buffer; else only the runs k_cull listed (Surfels::box) -- the others hold no surfel that could
// pass the tests above, so the keys are the same bits either way
__device__ __forceinline__ void index_scatter_body(Surfels src, const FrameDev* __restrict__ frame,
                                                   const PoseDev* __restrict__ pose, int W, int H, Intr k, float maxDepth,
                                                   int timeDelta, unsigned long long* __restrict__ keys, int transposed,
                                                   const int* __restrict__ vis_list = nullptr, const int* __restrict__ vis_count = nullptr,
                                                   bool pretest = false) {
    const int n = frame->count;
    const float time = (float)frame->tick;
    float Ri[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) Ri[q] = pose->Ri[q];
    const float3 ti = f3(pose->ti[0], pose->ti[1], pose->ti[2]);
    if (vis_list) {
        const int nv = *vis_count;
        for (int v = blockIdx.x; v < nv; v += gridDim.x) {
            const int r = vis_list[v];
            const int beg = src.box[2 * r + 1].w, end = min(n, src.box[2 * r + 3].w);
            for (int i0 = beg; i0 < end; i0 += 256)     // (wavefront-uniform bounds: every lane takes part in the exchange)
                index_scatter_one(src, i0 + (int)threadIdx.x, i0 + (int)threadIdx.x < end, time, Ri, ti, W, H, k, maxDepth, timeDelta, keys, transposed, pretest);
        }
        return;
    }
    for (int i0 = blockIdx.x * 256; i0 < n; i0 += gridDim.x * 256)
        index_scatter_one(src, i0 + (int)threadIdx.x, i0 + (int)threadIdx.x < n, time, Ri, ti, W, H, k, maxDepth, timeDelta, keys, transposed, pretest);
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
// block reduction of one run's box: per-thread (already reduced over the thread's own surfels) -> int4 pair; s_red: 4 x 8 ints of LDS
__device__ __forceinline__ void run_box_reduce(int (&lo)[3], int (&hi)[3], int tmax, int (*s_red)[8]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < 3; ++q) { lo[q] = wave_min_i(lo[q]); hi[q] = wave_max_i(hi[q]); }
    tmax = wave_max_i(tmax);
    if (lane == 0) {
#pragma unroll
        for (int q = 0; q < 3; ++q) { s_red[wave][q] = lo[q]; s_red[wave][4 + q] = hi[q]; }
        s_red[wave][3] = tmax;
    }
}
__device__ __forceinline__ void run_box_accumulate(float4 pc, float lastTime, int (&lo)[3], int (&hi)[3], int& tmax) {
    if (pc.x == pc.x && pc.y == pc.y && pc.z == pc.z) {   // (a NaN position is never in view)
        const int ex = box_enc(pc.x), ey = box_enc(pc.y), ez = box_enc(pc.z);
        lo[0] = min(lo[0], ex); lo[1] = min(lo[1], ey); lo[2] = min(lo[2], ez);
        hi[0] = max(hi[0], ex); hi[1] = max(hi[1], ey); hi[2] = max(hi[2], ez);
    }
    if (lastTime == lastTime) tmax = max(tmax, box_enc(lastTime));
}
// thread 0 .. : the four wavefronts' partial boxes -> the table entry of run r starting at slot `start`
__device__ __forceinline__ void run_box_store(int4* __restrict__ box, int r, int start, const int (*s_red)[8]) {
    int4 a, b;
    a.x = min(min(s_red[0][0], s_red[1][0]), min(s_red[2][0], s_red[3][0]));
    a.y = min(min(s_red[0][1], s_red[1][1]), min(s_red[2][1], s_red[3][1]));
    a.z = min(min(s_red[0][2], s_red[1][2]), min(s_red[2][2], s_red[3][2]));
    a.w = max(max(s_red[0][3], s_red[1][3]), max(s_red[2][3], s_red[3][3]));
    b.x = max(max(s_red[0][4], s_red[1][4]), max(s_red[2][4], s_red[3][4]));
    b.y = max(max(s_red[0][5], s_red[1][5]), max(s_red[2][5], s_red[3][5]));
    b.z = max(max(s_red[0][6], s_red[1][6]), max(s_red[2][6], s_red[3][6]));
    b.w = start;
    box[2 * r] = a; box[2 * r + 1] = b;
}

__global__ __launch_bounds__(256) void k_run_table(Surfels s, FrameDev* __restrict__ frame) {
    __shared__ int s_red[4][8];
    const int n = frame->count;
    const int runs = (n + kRun - 1) / kRun;
    for (int r = blockIdx.x; r < runs; r += gridDim.x) {
        int lo[3] = {kBoxEmptyMin, kBoxEmptyMin, kBoxEmptyMin}, hi[3] = {kBoxEmptyMax, kBoxEmptyMax, kBoxEmptyMax}, tmax = kBoxEmptyMax;
        for (int i = r * kRun + (int)threadIdx.x; i < min(n, (r + 1) * kRun); i += 256) run_box_accumulate(s.pc[i], s.ct[i].w, lo, hi, tmax);
        run_box_reduce(lo, hi, tmax, s_red);
        __syncthreads();
        if (threadIdx.x == 0) run_box_store(s.box, r, r * kRun, s_red);
        __syncthreads();
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        s.box[2 * runs + 1] = make_int4(0, 0, 0, n);   // end of the last run
        frame->runs = runs;
    }
}
void launch_run_table(Surfels s, FrameDev* frame, hipStream_t st) {
    hipLaunchKernelGGL(k_run_table, dim3(1024), dim3(256), 0, st, s, frame);
}
size_t run_table_entries(long elements) { return (size_t)(2 * ((elements + kRun - 1) / kRun + 2)); }

// One thread per run: conservative frustum test of the run's box under `pose` (the per-surfel tests of the passes that consume the list are
// u in [0, W] x [0, H] and 0 <= z <= maxDepth on individually rounded floats: the box is tested against the image grown by 2 px and the depth
// range grown by 1 cm -- metres against rounding errors of micrometres) and the activity test (no surfel seen within timeDelta: every pass
// drops all of them, index_map.vert:46, splat.vert:58).  Runs that pass are appended to `list` in no particular order.
__global__ __launch_bounds__(256) void k_cull(Surfels s, const FrameDev* __restrict__ frame, const PoseDev* __restrict__ pose, int W, int H, Intr k,
                                              float maxDepth, int timeDelta, int* __restrict__ list, int* __restrict__ count, int* __restrict__ ctl) {
    const int runs = frame->runs;
    const float time = (float)frame->tick;
    const int r = blockIdx.x * 256 + threadIdx.x;
    bool vis = false;
    if (r < runs) {
        const int4 a = s.box[2 * r], b = s.box[2 * r + 1];
        const int len = s.box[2 * r + 3].w - b.w;
        if (len > 0 && a.x <= b.x && a.y <= b.y && a.z <= b.z && !(time - box_dec(a.w) > (float)timeDelta)) {
            const float lo[3] = {box_dec(a.x), box_dec(a.y), box_dec(a.z)}, hi[3] = {box_dec(b.x), box_dec(b.y), box_dec(b.z)};
            // every corner outside the SAME half-space => the whole box is: near / far, then the four image sides as planes through the eye
            // (u < -2  <=>  fx x + (cx + 2) z < 0 for z > 0; points with z <= 0 fail the depth test anyway)
            int out_near = 1, out_far = 1, out_l = 1, out_r = 1, out_t = 1, out_b = 1;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float3 p = f3((c & 1) ? hi[0] : lo[0], (c & 2) ? hi[1] : lo[1], (c & 4) ? hi[2] : lo[2]);
                const float3 h = mul33(pose->Ri, p) + f3(pose->ti[0], pose->ti[1], pose->ti[2]);
                out_near &= h.z < -0.01f;
                out_far &= h.z > maxDepth + 0.01f;
                out_l &= k.fx * h.x + (k.cx + 2.f) * h.z < 0.f;
                out_r &= k.fx * h.x + (k.cx - (float)W - 2.f) * h.z > 0.f;
                out_t &= k.fy * h.y + (k.cy + 2.f) * h.z < 0.f;
                out_b &= k.fy * h.y + (k.cy - (float)H - 2.f) * h.z > 0.f;
            }
            vis = !(out_near | out_far | out_l | out_r | out_t | out_b);
        }
    }
    const unsigned long long m = __ballot(vis);
    int base = 0;
    if ((threadIdx.x & 63) == 0 && m) base = atomicAdd(&ctl[0], __popcll(m));
    base = __shfl(base, 0, 64);
    if (vis) list[base + lane_rank(m)] = r;
    // the last workgroup to finish publishes the count and re-arms the counters
    __syncthreads();
    if (threadIdx.x == 0) {
        // (every workgroup's slot reservations have returned before its barrier: relaxed atomics suffice, an agent-scope release would write the L2 back)
        const int done = atomicAdd(&ctl[1], 1);
        if (done == (int)gridDim.x - 1) {
            count[0] = __hip_atomic_load(&ctl[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&ctl[0], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&ctl[1], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
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
__device__ __forceinline__ ResolvedTexel resolve_texel(const Surfels& src, const PoseDev* __restrict__ pose, unsigned long long key, bool want_ct) {
    ResolvedTexel r;
    r.index = 0;
    r.vc = r.nr = r.ct = r.p0 = r.p1 = make_float4(0, 0, 0, 0);
    if (key == kEmptyKey) return r;
    const int i = (int)(unsigned)(key & 0xFFFFFFFFull);
    const float4 pc = src.pc[i];
    const float3 h = mul33(pose->Ri, f3(pc.x, pc.y, pc.z)) + f3(pose->ti[0], pose->ti[1], pose->ti[2]);
    if (kPacked) {
        const float4 c4 = src.ct[i];
        r.p0 = make_float4(h.x, h.y, h.z, pc.w);
        r.p1 = make_float4(c4.z, c4.w, __int_as_float(i), 0.f);
    } else {
        const float4 n4 = src.nr[i];
        const float3 n = normalize_gl(mul33(pose->Ri, f3(n4.x, n4.y, n4.z)));
        r.index = i;
        r.vc = make_float4(h.x, h.y, h.z, pc.w);
        r.nr = make_float4(n.x, n.y, n.z, n4.w);
        if (want_ct) r.ct = src.ct[i];   // colorTime image: only the clean pass of a freshly spawned model reads it from here
    }
    return r;
}
// frame planes that travel with the packed map (copy_unstable.vert:139-156 looks the filtered depth and the mask up at the surfel's own texel: in the
// packed, column-major order these are neighbours of its window taps; in the row-major images every lane pulled a sector of its own):
// depthF -> the packed record's spare word, mask -> maskT (column-major bytes)
struct ResolveOut { int* index; float4* vc; float4* nr; float4* ct; float4* packed; const float* depthF; const uint8_t* mask; uint8_t* maskT; };

// row-major keys -> row-major maps: texel p of the key image is texel p of the outputs
template <bool kPacked>
__device__ __forceinline__ void index_resolve_same_body(Surfels src, const PoseDev* __restrict__ pose, unsigned long long* __restrict__ keys, int P, ResolveOut o) {
    static_assert(!kPacked, "the packed map goes through a tile kernel");
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
    const unsigned long long key = keys[p];
    keys[p] = kEmptyKey;
    const ResolvedTexel r = resolve_texel<false>(src, pose, key, o.ct != nullptr);
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
        ResolvedTexel r = resolve_texel<kPacked>(src, pose, kEmptyKey, false);
        if (x < W && y < H) {
            const int p = kPacked ? y * W + x : x * H + y;
            const unsigned long long key = keys[p];
            keys[p] = kEmptyKey;
            r = resolve_texel<kPacked>(src, pose, key, o.ct != nullptr);
            if (kPacked) { r.p1.w = o.depthF[p]; s_mk[lx][ly] = o.mask[p]; }
        }
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
        if (x < W && y < H) { s_d[f][g] = o.depthF[y * W + x]; s_mk[f][g] = o.mask[y * W + x]; }
    }
    __syncthreads();
    const int x = x0 + g, y = y0 + f;
    if (x < W && y < H) {
        const int tp = x * H + y;
        const unsigned long long key = keys[tp];
        keys[tp] = kEmptyKey;
        ResolvedTexel r = resolve_texel<true>(src, pose, key, false);
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
void launch_index_resolve(Surfels src, const PoseDev* pose, unsigned long long* keys, int W, int H, int* index, float4* vc,
                          float4* nr, float4* ct, float4* packed, const float* depthF, const uint8_t* mask, uint8_t* maskT, bool keys_transposed,
                          hipStream_t s) {
    const int P = W * H;
    const ResolveOut o{index, vc, nr, ct, packed, depthF, mask, maskT};
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
    if (i < 0 || i >= frame->count) return;
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
    uint8_t* flags; float* newconf;    // keep flag / new confidence per element: the two-launch form's intermediate; for the one-launch form optional taps
    int* block_counts;                 // [kCompactBlocks] survivors per workgroup (two-launch form)
    int* host_count;
    unsigned long long* scan_state;    // [chunks] decoupled look-back: (launch epoch << 34 | status << 32 | survivors)
    int* ctl;                          // kCleanCtlInts ints, zero between launches: finished workgroups + the ticket counters (clean_body)
    unsigned epoch;                    // distinguishes this launch's entries of scan_state from older ones (never reset)
    int ticket_lanes;                  // counters the chunks are drawn from (1 .. kCleanTicketLanes, <= compute units and <= workgroups launched)
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
// clean, the form for SMALL maps: two launches over a static partition (rounds 1-4).  A round of the one-launch form below is ~100 us of
// dependent latency whatever it moves; on a map of a few hundred thousand surfels that IS the pass (k_clean at VGA: 100 us against
// 25 + 8 us for these two), on 27 M surfels it is amortised (1.04 against 0.95 + 0.53 ms).  launch_clean picks by the element count; the
// surviving records, their order and the count are the same bits either way (tests/test_gpu_switches.py::test_clean_forms_agree).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void clean_small_flags_body(const CleanArgs& a) {
    __shared__ int s_w[4];
    const int count = a.frame->count;
    const int total = count + cand_count(a.W, a.H, a.frame->tick);
    const float time = (float)a.frame->tick;
    float Ri[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) Ri[q] = a.pose->Ri[q];
    const float3 ti = f3(a.pose->ti[0], a.pose->ti[1], a.pose->ti[2]);
    const int chunk = chunk_size(total);
    const int beg = blockIdx.x * chunk, end = min(total, beg + chunk);
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
        a.flags[i] = keep ? 1 : 0;
        a.newconf[i] = nc;
        kept += keep ? 1 : 0;
    }
    const int tot = block_sum_i(kept, s_w);
    if (threadIdx.x == 0) {
        a.block_counts[blockIdx.x] = tot;
        if (blockIdx.x == 0) {
            a.frame->countNext = count;  // snapshot for pass 2 (see FrameDev)
            if (a.maskID != 0)   // the compaction pass (next launch) accumulates this clean pass's box
                for (int q = 0; q < 6; ++q) a.frame->bbox_acc[q] = q < 3 ? kBBoxEmptyMin : kBBoxEmptyMax;
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
    const int total = count + cand_count(a.W, a.H, a.frame->tick);
    const float time = (float)a.frame->tick;
    const int chunk = chunk_size(total);
    const int beg = blockIdx.x * chunk, end = min(total, beg + chunk);
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
            flag = a.flags[i];
            nc = a.newconf[i];
            if (i < count) { pc = a.src.pc[i]; ct = a.src.ct[i]; nr = a.src.nr[i]; }
            else { const int c = i - count; pc = a.cand_rec[c * 3 + 0]; ct = a.cand_rec[c * 3 + 1]; nr = a.cand_rec[c * 3 + 2]; }
        }
        if (i0 == beg) base = block_base(a.block_counts, s_w);
        const bool keep = in && flag;
        const unsigned long long m = __ballot(keep);
        if (lane == 0) s_w[wave] = __popcll(m);
        __syncthreads();
        int off = base;
        for (int w = 0; w < wave; ++w) off += s_w[w];
        const int tot = s_w[0] + s_w[1] + s_w[2] + s_w[3];
        if (keep) {
            const int o = off + lane_rank(m);
            pc.w = nc;
            if (ct.w == -2.f) ct.w = time;  // copy_unstable.vert:131
            if (o < a.dst.cap) {
                a.dst.pc[o] = pc; a.dst.ct[o] = ct; a.dst.nr[o] = nr;
                if (bbox_on && pc.w > a.confThreshold) {   // draw_global_surface.vert:55 (unstable == 0), :69-78
                    const int x = (int)(1000.f * pc.x), y = (int)(1000.f * pc.y), z = (int)(1000.f * pc.z);
                    bmin[0] = min(bmin[0], x); bmin[1] = min(bmin[1], y); bmin[2] = min(bmin[2], z);
                    bmax[0] = max(bmax[0], x); bmax[1] = max(bmax[1], y); bmax[2] = max(bmax[2], z);
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
    if (beg >= end) base = block_base(a.block_counts, s_w);   // the last workgroup owns no elements: all threads take part
    if (threadIdx.x == 0) {
        a.frame->count = min(base, a.dst.cap);
        a.frame->runs = 0;   // (this form writes no run table: the host does not cull a buffer it produced)
        if (a.host_count) *a.host_count = min(base, a.dst.cap);
    }
}

__global__ __launch_bounds__(256) void k_clean_small_compact(const CleanArgs a) { clean_small_compact_body(a); }

// ------------------------------------------------------------------------------------------------
// clean (copy_unstable.vert:53-157) in ONE launch: test + ordered compaction with a decoupled look-back.
// Rounds 1-4 ran two launches (k_clean_flags: test -> keep flags + new confidences + per-workgroup counts; k_clean_compact: prefix of the
// counts, ordered copy) over a static partition: the workgroups whose slice lay in view did all the window gathers while the others idled, and
// every flag / confidence took a round trip through HBM (154 B per surfel moved; 1.28 + 0.58 ms on the 26.9 M-surfel map of configs[4]).
// Here a workgroup draws a chunk of kCleanChunk consecutive elements (element i < count: old surfel i; element count + c: candidate c, live
// only with op == 2) from a ticket counter and
//   sweep 1   fetches the half of each record the test needs (position + confidence, the two time stamps: 24 B), tests it -- keep bit and decay
//             code stay in two registers per thread, the fetched half goes to the LDS --, and requests the other half of the survivors;
//   look-back publishes the chunk's number of survivors and obtains the number of survivors of all earlier chunks from the published values
//             (Merrill & Garland's decoupled look-back; the ticket order guarantees that every earlier chunk is owned by a workgroup that is
//             already running, so the wait terminates);
//   sweep 2   writes the survivors to their final slots -- the order of the output is the order of the input, as transform feedback keeps it --
//             and the bounding boxes of the new buffer's runs (Surfels::box).
// Every byte of the input is read ONCE: 96 B per surfel moved plus the window gathers of the surfels in view.
// Round 5 history (profiles/r05b_* .. r05n_*; tools/clean_prof.py times the phases of every chunk with an instrumented build): records held in
// registers, chunks of 512 / 1024: 1.2-1.9 ms; one ticket counter, 32 of them, 128 B .. 64 KB apart: no difference (a counter hands out a ticket
// every 11 ns, 32 of them one every 0.5 ns: tools/micro/ticket_lanes); 2048 elements, both sweeps from memory (144 B per surfel): 1.33 ms;
// all of a thread's loads in flight at once instead of four dependent batches: 1.35 ms; this form: 1.29 ms.  Whatever the form, a chunk's phases
// stretch with the number of workgroups in flight -- a dependent round trip to memory takes 6-8 us while the pass runs, with 768 or 1024
// workgroups -- and the pass ends up at ~10 chunks per microsecond: the memory system is saturated by ~3 TB/s of mixed traffic (reads of six
// streams in 32 KB pieces whose order the tickets decide, writes, 64-byte gathers), not by this kernel's instruction stream.
// ------------------------------------------------------------------------------------------------
constexpr int kCleanPerThread = 8;
constexpr int kCleanChunk = 256 * kCleanPerThread;
constexpr int kSubRuns = kCleanChunk / kRun;   // the survivors of a chunk are kSubRuns consecutive runs of the new buffer's run table
constexpr int kSlicesPerRun = kRun / 256;
static_assert(kCleanChunk % kRun == 0 && kRun % 256 == 0, "runs are whole slices of a chunk");
constexpr int kLookPerLane = 4;   // states of earlier chunks a lane reads per look-back step: 256 per step and wavefront (16 per lane was slower: 27 against 17 us per look-back)
constexpr unsigned kScanAggregate = 1u, kScanInclusive = 2u;
constexpr int kTicketStride = 32, kTicketBase = 32;   // ctl: [2..3] finished workgroups << 32 | survivors (64 bit), [kTicketBase + g kTicketStride] ticket counter of lane g (128 B apart)

__device__ __forceinline__ unsigned long long scan_load(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void scan_store(unsigned long long* p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// exclusive prefix of chunk `chunk` (> 0), by wavefront 0.  A step reads the states of the 256 chunks before the part already summed -- lane l
// those at distance 4 l .. 4 l + 3, nearest first -- waits until each of them has published something for THIS launch, and adds them up to and
// including the nearest one that carries an inclusive prefix.  (Round 5, first version: 64 states per step.  Every step is a round trip to memory
// -- the states are read at agent scope, past the non-coherent L2 -- and with ~1000 chunks in flight a chunk walked ~16 of them, ~25 us, before it
// could write: the pass was bound by its look-back, 26 k chunks in ~0.9 ms.  256 per step cut the walk to four.)
__device__ __forceinline__ int scan_look_back(const unsigned long long* __restrict__ state, int chunk, unsigned epoch, int& gave_up) {
    const int lane = threadIdx.x & 63;
    int exclusive = 0;
    int idx = chunk - 1;          // nearest chunk not yet accounted for
    for (;;) {
        unsigned long long v[kLookPerLane];
        int spins = 0;
        for (;;) {
            bool ready = true;
#pragma unroll
            for (int q = 0; q < kLookPerLane; ++q) {
                const int mine = idx - (lane * kLookPerLane + q);
                v[q] = mine >= 0 ? scan_load(&state[mine]) : 0ull;
                ready = ready && (mine < 0 || (unsigned)(v[q] >> 34) == epoch);
            }
            if (__ballot(!ready) == 0ull) break;
            if (++spins > (1 << 22)) { gave_up = 1; break; }   // never in a correct run: a bounded wait cannot hang the GPU
            __builtin_amdgcn_s_sleep(1);
        }
        // this lane's values up to and including its nearest inclusive entry (if it has one)
        int mine_sum = 0;
        bool has_incl = false;
#pragma unroll
        for (int q = 0; q < kLookPerLane; ++q) {
            const int mine = idx - (lane * kLookPerLane + q);
            const bool valid = mine >= 0 && (unsigned)(v[q] >> 34) == epoch;
            if (valid && !has_incl) {
                mine_sum += (int)(unsigned)(v[q] & 0xFFFFFFFFull);
                has_incl = ((unsigned)(v[q] >> 32) & 3u) == kScanInclusive;
            }
        }
        const unsigned long long incl_mask = __ballot(has_incl);
        const int stop = incl_mask ? __builtin_ctzll(incl_mask) : 63;
        exclusive += wave_sum_i(lane <= stop ? mine_sum : 0);
        if (incl_mask || idx - 64 * kLookPerLane < 0 || gave_up) break;
        idx -= 64 * kLookPerLane;
    }
    return exclusive;
}

#ifdef MF_CLEAN_PROF
// tooling build (tools/clean_prof.py): per chunk of the background's clean pass {chunk start, ticket, sweep 1, look-back, sweep 2} in 100 MHz ticks
__device__ unsigned g_clean_prof[1 << 16][8];
#define MF_PROF_T(x) const unsigned long long x = wall_clock64()
#else
#define MF_PROF_T(x)
#endif
__device__ __forceinline__ void clean_body(const CleanArgs& a) {
    __shared__ int s_chunk, s_base;
    __shared__ int s_cnt[kCleanPerThread][4];
    __shared__ int s_bb[6];
    __shared__ int s_red[kSubRuns][4][8];   // the boxes of the chunk's runs (Surfels::box of dst): per-wavefront partial results
    __shared__ float4 s_pc[kCleanPerThread][256];   // the half of the chunk's records that sweep 1 fetched (48 KB with s_tm: three workgroups per CU)
    __shared__ float2 s_tm[kCleanPerThread][256];
    // Model::lastBoundingBox of an OBJECT model (Model.cpp:315-345 + draw_global_surface.vert:55-78: the box of the surfels the GUI draws --
    // confidence above the model's threshold -- in millimetres, truncated): accumulated here, where the frame's final records pass through
    // registers anyway; Model::fuse of the NEXT frame limits its depth with it (Model.cpp:480-501).  The background (id 0) never uses one.
    const bool bbox_on = a.maskID != 0;
    int bmin[3] = {kBBoxEmptyMin, kBBoxEmptyMin, kBBoxEmptyMin}, bmax[3] = {kBBoxEmptyMax, kBBoxEmptyMax, kBBoxEmptyMax};
    if (bbox_on && threadIdx.x < 6) s_bb[threadIdx.x] = threadIdx.x < 3 ? kBBoxEmptyMin : kBBoxEmptyMax;
    // frame->count stays what it is for the whole launch: the new count is installed by the LAST workgroup to finish (below)
    const int count = a.frame->count;
    const int total = count + cand_count(a.W, a.H, a.frame->tick);
    const int nchunks = (total + kCleanChunk - 1) / kCleanChunk;
    const float time = (float)a.frame->tick;
    float Ri[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) Ri[q] = a.pose->Ri[q];
    const float3 ti = f3(a.pose->ti[0], a.pose->ti[1], a.pose->ti[2]);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // Chunks are handed out by `lanes` (<= kCleanTicketLanes) counters, lane g (= this workgroup's index mod lanes) serving chunks g, g + lanes, ...:
    // one counter serves a ticket every ~11 ns, 32 of them one every 0.5 ns (tools/micro/ticket_lanes.hip).  (Handing the chunks out in blocks of
    // 4 / 16 / 64 consecutive ones per XCD, so that an XCD's workgroups walk a contiguous piece of the buffer: 1.32 / 1.34 / 1.56 ms against 1.33 --
    // the memory phases get a little shorter, the waits in the look-back longer.)  Forward progress: a chunk only ever waits for LOWER chunks; the lowest chunk not
    // yet finished is either owned by a running workgroup or next in line on a lane whose workgroups (the first `lanes` workgroups of the
    // grid are dispatched first, one per lane: the host keeps lanes <= the number of compute units) are all working on lower chunks, which finish.
    const int lanes = a.ticket_lanes;
    const int tlane = (int)(blockIdx.x % lanes);
    int wg_kept = 0;          // survivors of the chunks this workgroup handled (thread 0's copy is the one that counts)
    for (;;) {
        MF_PROF_T(t_0);
        if (threadIdx.x == 0) s_chunk = tlane + lanes * atomicAdd(&a.ctl[kTicketBase + tlane * kTicketStride], 1);
        __syncthreads();
        const int chunk = s_chunk;
        if (chunk >= nchunks) break;
        MF_PROF_T(t_1);
        const int first = chunk * kCleanChunk + (int)threadIdx.x;
        // ---- sweep 1: test.  Element j of this thread is first + 256 j.  The test needs HALF of a record -- position + confidence, the two time
        // stamps: 24 of its 48 bytes; the normal / radius record only for an element in view (clean_test fetches it there) -- and that half stays
        // in the LDS for sweep 2, which fetches the other half: every byte of the input is read once (96 B per surfel moved; rounds 1-4: 154).
        unsigned keepmask = 0u, decay = 0u;
#pragma unroll 1
        for (int sr = 0; sr < kSubRuns; ++sr) {
            int rlo[3] = {kBoxEmptyMin, kBoxEmptyMin, kBoxEmptyMin}, rhi[3] = {kBoxEmptyMax, kBoxEmptyMax, kBoxEmptyMax}, rtime = kBoxEmptyMax;
            float4 pc[kSlicesPerRun];
            float2 tm[kSlicesPerRun];
            bool live[kSlicesPerRun];
#pragma unroll
            for (int q = 0; q < kSlicesPerRun; ++q) {   // the slices' records are requested before anything depends on one of them
                const int i = first + 256 * (sr * kSlicesPerRun + q);
                live[q] = i < total;
                pc[q] = make_float4(0, 0, 0, 0); tm[q] = make_float2(0, 0);
                if (i < count) {
                    pc[q] = a.src.pc[i];
                    tm[q] = *reinterpret_cast<const float2*>(reinterpret_cast<const float*>(&a.src.ct[i]) + 2);
                } else if (live[q]) {
                    const int c = i - count;
                    live[q] = a.cand_op[c] == 2;      // op == 1 records carry w = -1 and are dropped, op == 0 slots hold nothing
                    if (live[q]) {
                        pc[q] = a.cand_rec[c * 3 + 0];
                        tm[q] = *reinterpret_cast<const float2*>(reinterpret_cast<const float*>(&a.cand_rec[c * 3 + 1]) + 2);
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < kSlicesPerRun; ++q) {
                const int j = sr * kSlicesPerRun + q;
                const int i = first + 256 * j;
                const float4 ct = make_float4(0.f, 0.f, tm[q].x, tm[q].y);
                const float4* nrp = i < count ? &a.src.nr[i] : &a.cand_rec[(i - count) * 3 + 2];
                float nc = 0.f;
                int dk = 0;
                const bool keep = live[q] && clean_test(a, pc[q], ct, make_float4(0, 0, 0, 0), time, Ri, ti, nc, dk, nrp);
                if (a.flags) {
                    if (i < total) { a.flags[i] = keep ? 1 : 0; a.newconf[i] = live[q] ? nc : 0.f; }
                }
                s_pc[j][threadIdx.x] = pc[q];
                s_tm[j][threadIdx.x] = tm[q];
                keepmask |= (keep ? 1u : 0u) << j;
                decay |= (unsigned)dk << (2 * j);
                if (keep) run_box_accumulate(pc[q], ct.w == -2.f ? time : ct.w, rlo, rhi, rtime);   // (copy_unstable.vert:131: -2 becomes the time)
                const unsigned long long m = __ballot(keep);
                if (lane == 0) s_cnt[j][wave] = __popcll(m);
            }
            run_box_reduce(rlo, rhi, rtime, s_red[sr]);
        }
        // The OTHER half of every surviving record (colour word + the unused word, normal + radius) is requested now: the loads travel while
        // the workgroup waits for its place in the output.
        float2 cus[kCleanPerThread];
        float4 nrs[kCleanPerThread];
#pragma unroll
        for (int j = 0; j < kCleanPerThread; ++j) {
            cus[j] = make_float2(0, 0); nrs[j] = make_float4(0, 0, 0, 0);
            if ((keepmask >> j) & 1u) {
                const int i = first + 256 * j;
                const float4* rec1 = i < count ? &a.src.ct[i] : &a.cand_rec[(i - count) * 3 + 1];
                cus[j] = *reinterpret_cast<const float2*>(rec1);
                nrs[j] = i < count ? a.src.nr[i] : a.cand_rec[(i - count) * 3 + 2];
            }
        }
        __syncthreads();
        MF_PROF_T(t_2);
        if (wave == 0) {
            int tot = 0;
#pragma unroll
            for (int j = 0; j < kCleanPerThread; ++j) tot += s_cnt[j][0] + s_cnt[j][1] + s_cnt[j][2] + s_cnt[j][3];
            const unsigned long long tag = (unsigned long long)a.epoch << 34;
            int excl = 0, gave_up = 0;
            if (chunk > 0) {
                if (lane == 0) scan_store(&a.scan_state[chunk], tag | ((unsigned long long)kScanAggregate << 32) | (unsigned)tot);
                excl = scan_look_back(a.scan_state, chunk, a.epoch, gave_up);
            }
            if (lane == 0) {
                scan_store(&a.scan_state[chunk], tag | ((unsigned long long)kScanInclusive << 32) | (unsigned)(excl + tot));
                s_base = excl;
                wg_kept += tot;
                if (gave_up) a.frame->pad[2] = 1;
                int start = excl;     // the chunk's survivors are kSubRuns consecutive runs of the new buffer
                for (int sr = 0; sr < kSubRuns; ++sr) {
                    run_box_store(a.dst.box, chunk * kSubRuns + sr, min(start, a.dst.cap), s_red[sr]);
                    for (int q = 0; q < kSlicesPerRun; ++q) {
                        const int j = sr * kSlicesPerRun + q;
                        start += s_cnt[j][0] + s_cnt[j][1] + s_cnt[j][2] + s_cnt[j][3];
                    }
                }
                if (chunk == nchunks - 1) a.dst.box[2 * nchunks * kSubRuns + 1] = make_int4(0, 0, 0, min(excl + tot, a.dst.cap));   // end of the last run
            }
        }
        __syncthreads();
        MF_PROF_T(t_3);
        // ---- sweep 2: the survivors go to their final slots, in order
        int off = s_base;
#pragma unroll
        for (int j = 0; j < kCleanPerThread; ++j) {
            const bool keep = (keepmask >> j) & 1u;
            const unsigned long long m = __ballot(keep);
            int o = off + lane_rank(m);
            for (int w = 0; w < wave; ++w) o += s_cnt[j][w];
            if (keep && o < a.dst.cap) {
                float4 pc = s_pc[j][threadIdx.x];
                const float2 tm = s_tm[j][threadIdx.x];
                pc.w = clean_decayed(a, pc.w, (int)((decay >> (2 * j)) & 3u));
                const float4 ct = make_float4(cus[j].x, cus[j].y, tm.x, tm.y == -2.f ? time : tm.y);   // copy_unstable.vert:131
                a.dst.pc[o] = pc; a.dst.ct[o] = ct; a.dst.nr[o] = nrs[j];
                if (bbox_on && pc.w > a.confThreshold) {   // draw_global_surface.vert:55 (unstable == 0), :69-78
                    const int x = (int)(1000.f * pc.x), y = (int)(1000.f * pc.y), z = (int)(1000.f * pc.z);
                    bmin[0] = min(bmin[0], x); bmin[1] = min(bmin[1], y); bmin[2] = min(bmin[2], z);
                    bmax[0] = max(bmax[0], x); bmax[1] = max(bmax[1], y); bmax[2] = max(bmax[2], z);
                }
            }
            off += s_cnt[j][0] + s_cnt[j][1] + s_cnt[j][2] + s_cnt[j][3];
        }
        __syncthreads();   // s_chunk / s_cnt / s_base / s_red are rewritten by the next round
#ifdef MF_CLEAN_PROF
        if (threadIdx.x == 0 && a.maskID == 0 && chunk < (1 << 16)) {
            const unsigned long long t_4 = wall_clock64();
            unsigned* o = g_clean_prof[chunk];
            o[0] = (unsigned)t_0; o[1] = (unsigned)(t_1 - t_0); o[2] = (unsigned)(t_2 - t_1); o[3] = (unsigned)(t_3 - t_2); o[4] = (unsigned)(t_4 - t_3);
            o[5] = blockIdx.x; o[6] = (unsigned)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 0xF; o[7] = (unsigned)s_base;
        }
#endif
    }
    if (bbox_on) {
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            if (bmin[q] != kBBoxEmptyMin) atomicMin(&s_bb[q], bmin[q]);
            if (bmax[q] != kBBoxEmptyMax) atomicMax(&s_bb[3 + q], bmax[q]);
        }
        __syncthreads();
        if (threadIdx.x < 3 && s_bb[threadIdx.x] != kBBoxEmptyMin) atomicMin(&a.frame->bbox_tmp[threadIdx.x], s_bb[threadIdx.x]);
        else if (threadIdx.x >= 3 && threadIdx.x < 6 && s_bb[threadIdx.x] != kBBoxEmptyMax) atomicMax(&a.frame->bbox_tmp[threadIdx.x], s_bb[threadIdx.x]);
    }
    // The last workgroup to FINISH installs the launch's results: by then nobody reads frame->count or the tickets any more.  One 64-bit atomic
    // per workgroup carries both its "finished" ticket and the number of survivors it wrote: the value travels IN the atomic, so the
    // workgroup that draws the last ticket holds the launch's total without any ordering between workgroups (an agent-scope release / acquire
    // pair here cost an L2 write-back per workgroup); the box atomics of this workgroup have completed behind the barrier above.
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long* done64 = reinterpret_cast<unsigned long long*>(a.ctl + 2);
        const unsigned long long old = atomicAdd(done64, (1ull << 32) | (unsigned long long)(unsigned)wg_kept);
        if ((unsigned)(old >> 32) == gridDim.x - 1u) {
            const int n = min((int)(unsigned)(old & 0xFFFFFFFFull) + wg_kept, a.dst.cap);
            a.frame->countNext = n;
            a.frame->count = n;
            a.frame->runs = nchunks * kSubRuns;
            if (a.host_count) *a.host_count = n;
            if (bbox_on) {   // the box of a frame is the box of its LAST clean pass (the reference's render pass sees the final buffer)
                for (int q = 0; q < 6; ++q) {
                    a.frame->bbox_acc[q] = __hip_atomic_load(&a.frame->bbox_tmp[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(&a.frame->bbox_tmp[q], q < 3 ? kBBoxEmptyMin : kBBoxEmptyMax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            for (int g = 0; g < kCleanTicketLanes; ++g) __hip_atomic_store(&a.ctl[kTicketBase + g * kTicketStride], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(done64, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

__global__ __launch_bounds__(256) void k_clean(const CleanArgs a) { clean_body(a); }
#ifdef MF_CLEAN_PROF
}  // namespace mf
extern "C" int mf_debug_clean_prof(unsigned* out, int chunks) {
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(mf::g_clean_prof), (size_t)chunks * 32) == hipSuccess ? 0 : -1;
}
namespace mf {
#endif

// workgroups of a clean launch for `elements` elements (an upper bound or an estimate: the chunks are drawn from a ticket counter, any
// grid covers any count)
int clean_grid(long elements) {
    const long chunks = (elements + kCleanChunk - 1) / kCleanChunk;
    return (int)(chunks < kCleanTicketLanes ? kCleanTicketLanes : (chunks > kCleanGridMax ? kCleanGridMax : chunks));
}
size_t clean_scan_entries(long max_elements) { return (size_t)((max_elements + kCleanChunk - 1) / kCleanChunk + 1); }
static_assert(kTicketBase + kCleanTicketLanes * kTicketStride <= kCleanCtlInts, "ticket counters fit the control block");

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
                                                   float confThreshold, int timeDelta, unsigned long long* __restrict__ keys) {
    const int n = frame->count;
    const float time = (float)frame->tick;  // combinedPredict(time = tick, maxTime = tick)
    float Ri[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) Ri[q] = pose->Ri[q];
    const float3 ti = f3(pose->ti[0], pose->ti[1], pose->ti[2]);
    constexpr int kLX = kLanes >= 2 ? 2 : 1, kLY = kLanes / kLX;
    const int sub = threadIdx.x % kLanes, sx = sub % kLX, sy = sub / kLX;
    for (int i = (blockIdx.x * 256 + threadIdx.x) / kLanes; i < n; i += gridDim.x * 256 / kLanes) {
        const float4 pc = src.pc[i];
        if (pc.w < confThreshold) continue;
        const float lastTime = src.ct[i].w;
        const float3 h = mul33(Ri, f3(pc.x, pc.y, pc.z)) + ti;
        if (h.z > maxDepth || h.z < 0 || time - lastTime > (float)timeDelta || lastTime > time) continue;  // splat.vert:58
        const float u = ((k.fx * h.x) / h.z) + k.cx, v = ((k.fy * h.y) / h.z) + k.cy;
        if (!(u >= 0.f && u <= (float)W && v >= 0.f && v <= (float)H)) continue;
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
        if (!(size > 0.f)) continue;
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
                if (kLanes > 1) zmin_key_pretested(&keys[py * W + px], key);   // (the object models' launches)
                else zmin_key(&keys[py * W + px], key);
            }
        }
    }
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

void launch_clean(Surfels src, Surfels dst, FrameDev* frame, const PoseDev* pose, int W, int H, Intr k, int timeDelta,
                  float confThreshold, float outlierCoeff, int maskID, const int* index, const float4* vc, const float4* ct,
                  const float4* packed, const float* depthF, const uint8_t* mask, const uint8_t* maskT, const uint8_t* cand_op, const float4* cand_rec, uint8_t* flags,
                  float* newconf, int* block_counts, unsigned long long* scan_state, int* ctl, unsigned epoch, int blocks, int ticket_lanes,
                  int* host_count_mirror, bool transposed, bool literalWindow, bool small_map, hipStream_t s) {
    CleanArgs a;
    a.transposed = transposed ? 1 : 0;
    a.literal = literalWindow ? 1 : 0;
    a.src = src; a.dst = dst; a.frame = frame; a.pose = pose; a.W = W; a.H = H; a.k = k; a.timeDelta = timeDelta;
    a.confThreshold = confThreshold; a.outlierCoeff = outlierCoeff; a.maskID = maskID; a.index = index; a.vc = vc; a.ct = ct;
    a.packed = packed;
    a.depthF = depthF; a.mask = mask; a.maskT = maskT; a.cand_op = cand_op; a.cand_rec = cand_rec;
    a.flags = flags; a.newconf = newconf; a.block_counts = block_counts; a.host_count = host_count_mirror;
    a.scan_state = scan_state; a.ctl = ctl; a.epoch = epoch; a.ticket_lanes = min(ticket_lanes, blocks);
    if (small_map) {
        hipLaunchKernelGGL(k_clean_small_flags, dim3(kCompactBlocks), dim3(256), 0, s, a);
        hipLaunchKernelGGL(k_clean_small_compact, dim3(kCompactBlocks), dim3(256), 0, s, a);
        return;
    }
    hipLaunchKernelGGL(k_clean, dim3(blocks), dim3(256), 0, s, a);
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
    index_resolve_same_body<false>(m.a, m.pose, m.keys, b.W * b.H, ResolveOut{m.index, m.ivc, m.inr, nullptr, nullptr, nullptr, nullptr, nullptr});
}
__global__ __launch_bounds__(256) void k_obj_index_resolve_packed(const ObjBatch b) {   // the pass that feeds clean(): grid.x = 16 x 16-pixel tiles
    const ObjPassArgs& m = b.m[blockIdx.z];
    index_resolve_transposing_body<true>(b.updateCopy ? m.b : m.a, m.pose, m.keys, b.W, b.H,
                                         ResolveOut{nullptr, nullptr, nullptr, nullptr, m.iclean, b.depthF, b.mask, b.maskT}, (int)blockIdx.x);
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
__device__ __forceinline__ CleanArgs obj_clean_args(const ObjBatch& b, const ObjPassArgs& m) {
    CleanArgs a;
    // copy-update: fuse copied a -> b, clean goes b -> a (the live buffer stays); in place: clean goes a -> b
    a.src = b.updateCopy ? m.b : m.a; a.dst = b.updateCopy ? m.a : m.b;
    a.frame = m.frame; a.pose = m.pose; a.W = b.W; a.H = b.H; a.k = b.k; a.timeDelta = b.timeDelta;
    a.confThreshold = m.confThreshold; a.outlierCoeff = b.outlierCoeff; a.maskID = m.maskID; a.transposed = 1; a.literal = b.cleanLiteral;
    a.index = m.index; a.vc = m.ivc; a.ct = nullptr; a.packed = m.iclean; a.depthF = b.depthF; a.mask = b.mask; a.maskT = b.maskT;
    a.cand_op = m.cand_op; a.cand_rec = m.cand_rec; a.flags = b.cleanSmall ? m.flags : nullptr; a.newconf = b.cleanSmall ? m.newconf : nullptr;
    a.block_counts = m.block_counts; a.host_count = m.host_count;
    a.scan_state = m.scan_state; a.ctl = m.clean_ctl; a.epoch = b.cleanEpoch; a.ticket_lanes = b.cleanTicketLanes;
    return a;
}
__global__ __launch_bounds__(256) void k_obj_clean(const ObjBatch b) { clean_body(obj_clean_args(b, b.m[blockIdx.z])); }
__global__ __launch_bounds__(256) void k_obj_clean_small_flags(const ObjBatch b) { clean_small_flags_body(obj_clean_args(b, b.m[blockIdx.z])); }
__global__ __launch_bounds__(256) void k_obj_clean_small_compact(const ObjBatch b) { clean_small_compact_body(obj_clean_args(b, b.m[blockIdx.z])); }
__global__ __launch_bounds__(256) void k_obj_splat_scatter(const ObjBatch b) {
    const ObjPassArgs& m = b.m[blockIdx.z];
    splat_scatter_body<4>(m.a, m.frame, m.pose, b.W, b.H, b.k, b.maxDepthProcessed, m.confThreshold, b.timeDelta, m.keys);
}
__global__ __launch_bounds__(256) void k_obj_splat_resolve(const ObjBatch b) {
    const ObjPassArgs& m = b.m[blockIdx.z];
    splat_resolve_body(m.a, m.pose, m.keys, b.W, b.H, b.k, m.predV, m.predN, m.predImage, m.predTime, m.frame, b.rgb, m.predGray, nullptr, 0);
}
__global__ void k_obj_frame_advance(const ObjBatch b) {
    const ObjPassArgs& m = b.m[blockIdx.z];
    frame_advance_body(m.frame, b.W, b.H, m.host_frame, m.pose, b.bg_pose, m.log_slot);
}

void launch_obj_fuse_clean(const ObjBatch& b, int blocks, int clean_blocks, hipStream_t s) {
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
    if (b.cleanSmall) {
        hipLaunchKernelGGL(k_obj_clean_small_flags, dim3(kCompactBlocks, 1, b.n), dim3(256), 0, s, b);
        hipLaunchKernelGGL(k_obj_clean_small_compact, dim3(kCompactBlocks, 1, b.n), dim3(256), 0, s, b);
    } else {
        hipLaunchKernelGGL(k_obj_clean, compact, dim3(256), 0, s, b);
    }
}
void launch_obj_predict_advance(const ObjBatch& b, int blocks, hipStream_t s) {
    hipLaunchKernelGGL(k_obj_splat_scatter, dim3(blocks, 1, b.n), dim3(256), 0, s, b);
    hipLaunchKernelGGL(k_obj_splat_resolve, dim3((b.W + 63) / 64, (b.H + 3) / 4, b.n), dim3(256), 0, s, b);
    hipLaunchKernelGGL(k_obj_frame_advance, dim3(1, 1, b.n), dim3(64), 0, s, b);
}

}  // namespace mf

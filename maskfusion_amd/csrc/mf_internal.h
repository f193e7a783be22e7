// mf_internal.h -- shared declarations of the HIP hot path (gfx950 only; no CPU fallback, no CUDA shims).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mf {

constexpr int kLevels = 3;            // RGBDOdometry::NUM_PYRS (Core/Utils/RGBDOdometry.h:81)
constexpr int kIcpSlots = 32;         // 29 accumulators padded to 32 floats (128 B)
constexpr int kCompactBlocks = 2048;  // fixed grid of the ordered-compaction passes (a VGA map: one 256-element slice per workgroup)
constexpr unsigned long long kEmptyKey = 0xFFFFFFFFFFFFFFFFull;
constexpr int kNoUpdate = 0x7FFFFFFF;

struct Intr {
    float fx, fy, cx, cy;
};

// Device-resident pose block of one model.  Written by kernels only (the Gauss-Newton loop never returns to the
// host); mirrored to pinned host memory at the end of a frame.
struct PoseDev {
    float R[9], t[3];          // model pose (camera -> model frame), row-major
    float Ri[9], ti[3];        // inverse (t_inv uniform of the reference shaders)
    float lastR[9], lastT[3];  // Model::lastPose
    float fusionWeight;        // Model::computeFusionWeight(1.0)
    float lastICPError, lastICPCount;
    float initR[9], initT[3];  // Model::initialC2Winv (objects; Model.h:263-264)
    float incT[3];             // translation of the last tracking increment (MaskFusion.cpp:268)
    int alive;                 // 0 once the 0.2 m jump test dropped the model in this frame
    int rejected;              // 1 if the last tracking step was reverted by the 0.3 m rule (RGBDOdometry.cpp:477-481)
    float lastRGBError, lastRGBCount, lastSO3Error, lastSO3Count;
    int so3Iterations;
    int weightLiteral;         // 1: computeFusionWeight's log map with the reference's float trace ("literalFusionWeight", DESIGN.md finding F5)
    int illIterations;         // Gauss-Newton iterations of the last geometric tracking step whose system was outside the solver's stated domain (GNState::ill)
};

// Result of the SO(3) pre-alignment kernel (RGBDOdometry.cpp:264-324); seeds resultRt of the Gauss-Newton loop.
struct So3Result {
    double R[9];
    float error, count;
    int iterations, pad;
};

// Gauss-Newton state carried from one ICP launch to the next (double-buffered: launch k reads [k-1], block 0 writes [k]).
struct GNState {
    double resultRt[16];       // row-major, RGBDOdometry.cpp:336
    float Rprev[9], tprev[3], Rprev_inv[9];
    float Rcurr[9], tcurr[3];
    float lastICPError, lastICPCount;
    float trR[9], trt[3];      // Isometry3f transform (increment)
    int valid;
    int levelDone;             // pyramid level whose loop was left early (rgbOnly rule, RGBDOdometry.cpp:392-394); -1: none
    float lastRGBError, lastRGBCount;
    int ill;                   // iterations so far whose 6x6 system had fewer than 6 inliers or a pivot below 1e-8 of the largest diagonal entry
                               // (=> cond(A) > 1e8): there the unpivoted fp64 LDL^T here and the reference's pivoted Eigen::LDLT on float-rounded
                               // sums may return different steps (DESIGN.md finding F4); everywhere else they agree to rounding
};

// Scalars that live on the device so that no kernel launch needs a host round trip.
struct FrameDev {
    int tick;                  // MaskFusion::tick
    int count;                 // Model::count (live surfels)
    int countNext;             // snapshot of `count` (dense buffer) / `phys` (sparse buffer) taken by clean pass 1 and read by pass 2, whose last
                               // workgroup then rewrites them (no workgroup of pass 2 reads them, so no intra-launch ordering is assumed)
    int runs;                  // entries of the live buffer's run table (Surfels::box); 0: the buffer is DENSE and has no table -- its surfels are
                               // slots [0, count); > 0: the surfels are the slots [start_r, start_r + len_r) of the runs, in run order (a SPARSE
                               // buffer once a run has lost surfels: Model::clean then works in place, mf_surfel.hip "clean, in place")
    int cover;                 // predicted-colour coverage count (requiresFillIn)
    int useFillIn;             // decision taken for the current tracking step
    int pad[3];
    // Model::lastBoundingBox in millimetres (Model.cpp:315-345; {min xyz, max xyz}, empty = min > max) as of the end of the last frame, and
    // the one being accumulated by this frame's clean pass (object models only; the frame advance moves it over)
    int bbox[6], bbox_acc[6];
    int bbox_tmp[6];           // ... and what the workgroups of a running clean launch have merged so far (empty between launches)
    int phys;                  // first slot behind the last run (= count for a dense buffer): where Model::clean appends the frame's new surfels
    int first;                 // slot of the FIRST live surfel (0 for a dense buffer): the surfel whose vertex id is 0 and which the index map
                               // therefore cannot tell from "no surfel" (index_map.frag writes the id into a texture cleared to 0)
    int runsNext;              // snapshot of `runs` taken by clean pass 1 for pass 2 (like countNext)
    int first_run;             // ... and the run it starts (runs only ever lose surfels: the first non-empty run moves forward, never back)
    unsigned long long done_cover;   // k_splat_tile: (workgroups finished << 32) | coverage count of this launch; zero between launches
};

constexpr int kBBoxEmptyMin = 100000, kBBoxEmptyMax = -100000;   // Model.cpp:315: {1e5, 1e5, 1e5, -1e5, -1e5, -1e5}
constexpr int kBBoxNotRun = 0x7FFFFFFF;   // bbox_acc[0] between the frame advance and the next clean pass: no clean has accumulated a box since
// (device + host) reset of the two boxes of a FrameDev
#define MF_FRAME_BBOX_RESET(f) do { for (int q_ = 0; q_ < 3; ++q_) { (f)->bbox[q_] = (f)->bbox_acc[q_] = (f)->bbox_tmp[q_] = mf::kBBoxEmptyMin; \
                                                                     (f)->bbox[3 + q_] = (f)->bbox_acc[3 + q_] = (f)->bbox_tmp[3 + q_] = mf::kBBoxEmptyMax; } \
                                    (f)->bbox_acc[0] = mf::kBBoxNotRun; } while (0)
// (the box of a frame is the box of its LAST clean pass, like the reference's render pass over the final buffer -- on a spawn frame the spawn
// pass's clean output is not part of it: every clean launch REPLACES bbox_acc with what its workgroups merged into bbox_tmp)
// end of a frame (the GUI's renderPointCloud runs after processFrame, GUI/MainController.cpp:704-717): the accumulated box becomes
// lastBoundingBox; a frame without a clean pass for this model (rgbOnly, model-level calls) leaves the buffer, hence the box, as it was
#define MF_FRAME_BBOX_ADVANCE(f) do { if ((f)->bbox_acc[0] != mf::kBBoxNotRun) for (int q_ = 0; q_ < 6; ++q_) (f)->bbox[q_] = (f)->bbox_acc[q_]; \
                                      (f)->bbox_acc[0] = mf::kBBoxNotRun; } while (0)

struct Surfels {               // SoA of float4, 48 B per surfel in three coalesced streams
    float4* pc;                // position + confidence
    float4* ct;                // colour, unused, initTime, lastTime
    float4* nr;                // normal + radius
    int cap;                   // capacity in surfels (writes beyond it are dropped, like a full transform-feedback buffer)
    // Run table: the buffer as consecutive RUNS of at most kRun slots (after Model::initialise / an uploaded map / a compaction: slots
    // [r kRun, (r + 1) kRun); behind them the runs Model::clean appended, one frame's new surfels after the other), kBoxStride int4 per run:
    //   {enc(min x), enc(min y), enc(min z), enc(newest lastTime)}, {enc(max x), enc(max y), enc(max z), first slot of the run},
    //   {live surfels of the run (they are its FIRST slots, in order), enc(lowest confidence), 0, 0}
    // (enc: order-preserving float -> int, mf_device.h box_enc).
    // Surfels are stored in creation order, i.e. in spatially coherent runs; a full map is mostly out of view (the 26.5 M-surfel map of
    // configs[4]: 86 %), and the projection passes (index map x 2, prediction, GlobalProjection) only visit the runs whose box meets the viewing
    // frustum and that hold a surfel seen within timeDelta (k_cull) -- the reference streams the whole buffer through every pass
    // (ModelProjection.cpp:100-152,187-268).  Model::clean of a big map visits only the runs in which its tests can change something and
    // compacts each of them where it stands: the order of the surfels (= the reference's transform-feedback order) is the order of the runs.
    int4* box;
};
constexpr int kRun = 512;
constexpr int kBoxStride = 3;
__host__ __device__ inline int run_start(const int4* box, int r) { return box[kBoxStride * r + 1].w; }
__host__ __device__ inline int run_len(const int4* box, int r) { return box[kBoxStride * r + 2].x; }
struct VisList { const int* list; const int* count; };   // device memory: run indices (any order) and how many (launch_cull)
constexpr int kBoxEmptyMin = 0x7FFFFFFF, kBoxEmptyMax = (int)0x80000000;

// Persistent per-model block of the tracker (device memory, written once when the model is created): what the batched
// Gauss-Newton kernels need to find a model's maps, state and partial sums from blockIdx alone.
struct TrackModelDev {
    const float4* predV; const float4* predN;    // prediction consumed by initICPModel
    PoseDev* pose; PoseDev* pose_host; const FrameDev* frame;
    float* vm[3]; float* nm[3];                   // model-side vertex / normal pyramid (global frame)
    float* partials[2]; GNState* st;              // ping-pong per-workgroup partial sums [nb][32]; st[0], st[1]
    float* log;                                   // [19][32] reduced systems (background model) or nullptr
    double* trace;                                // [20][kGnTraceRow] Gauss-Newton trace (mf_odometry.hip: gn_trace_write) or nullptr
    float jump_limit;                             // object models: 0.2 m rule of MaskFusion.cpp:268-272; background: 0
    int allow_fill;                               // Model::allowsFillIn (background only)
    int* rect;                                    // {x0, y0, x1, y1}: the pixels of nm[0] that hold a normal (an object's prediction is NaN outside the object,
                                                  // Q3; the coarser levels' normals lie inside it shifted by the level) -- written by the batched model
                                                  // pyramid for object models, read by the batched pixel pass (slab culling), re-armed {INT_MAX, INT_MAX,
                                                  // INT_MIN, INT_MIN} by the batched finalize
};
constexpr int kMaxTrackBatch = 32;
constexpr int kGnTraceRow = 64;
struct TrackBatch { const TrackModelDev* m[kMaxTrackBatch]; int n; };   // by value in the kernel arguments

// ---------------- preprocessing ----------------
void launch_bilateral_model_pyramid(const float* depth, float* depthF, const float4* predV, const float4* predN, const float* fillDepth,
                                    const FrameDev* frame, const PoseDev* pose, float* const vmaps[3], float* const nmaps[3], int W, int H, Intr k,
                                    hipStream_t s);   // the filter + launch_model_pyramid's work in one launch (mf_odometry.hip)
void launch_bilateral(const float* depth, float* out, int W, int H, hipStream_t s);
void launch_pyrdown_f(const float* src, float* dst, int sw, int sh, hipStream_t s);
void launch_vmap_nmap(const float* depth, float* vmap, float* nmap, int W, int H, Intr k, float cutoff, hipStream_t s);
// depth pyramid (2 x pyrDownGaussF) + vertex/normal maps of the three levels in one launch; k: level-0 intrinsics
void launch_frame_pyramid(const float* depth, float* const vmap[3], float* const nmap[3], int W, int H, Intr k, float cutoff,
                          hipStream_t s);

// ---------------- odometry ----------------
// Fused RGBDOdometry::initICPModel.  pose: device PoseDev (R,t used).  fill-in inputs may be null (no fill-in).
void launch_model_pyramid(const float4* predV, const float4* predN, const float* fillDepth, const FrameDev* frame,
                          const PoseDev* pose, const float* R9t3_host_or_null, float* const vmaps[3],
                          float* const nmaps[3], int W, int H, Intr k, hipStream_t s);
// One Gauss-Newton iteration: [reduce+solve of the previous launch] -> normal equations of this level.
struct IcpLaunch {
    const float* vmap_curr; const float* nmap_curr; const float* vmap_prev; const float* nmap_prev;
    int W, H; Intr k;
    float distThres, angleThres;
    const float* partials_in; int nblocks_in;   // previous launch (nullptr/0 for the first)
    float* partials_out;
    const GNState* state_in; GNState* state_out;
    float* log_out;                              // optional [32] floats of the reduced system solved in this launch
    double* trace = nullptr; int it = 0;         // optional: the model's Gauss-Newton trace ([20][kGnTraceRow]) and this launch's iteration index
    unsigned long long* prof_out = nullptr;      // optional [16] shader-clock stamps (workgroup 0)
    const PoseDev* pose_in = nullptr;            // first launch only: seed the Gauss-Newton state from this pose
    const So3Result* so3_in = nullptr;           // first launch only: SO(3) pre-alignment seeds resultRt's rotation
};
int icp_grid_blocks(int W, int H);       // workgroups of the RGB-D iteration kernels for this level
int icp_geo_grid_blocks(int W, int H);   // workgroups of the geometric kernel k_icp_iter (more of them at the coarse levels: one pixel slot per thread)
void launch_icp_iteration(const IcpLaunch& a, hipStream_t s);
// ---- the same loop for SEVERAL models at once (MaskFusion.cpp:247-276 tracks them one after the other): one launch serves
// iteration k of every tracked model.  Split in two kernels per iteration -- "solve" (one workgroup per model: reduce the
// previous iteration's partials, LDL^T, pose) and "pixels" (grid.y = model, no prologue, any number of workgroup rounds) --
// because the redundant per-workgroup prologue of k_icp_iter would be paid once per model and per round.
int icp_batch_max_blocks(int W, int H);                       // upper bound of workgroups per model of launch_icp_batch_pixels
int icp_batch_blocks(int W, int H, int n_models);             // workgroups per model it will use for this image size
void launch_model_pyramid_batch(const TrackBatch& b, const float* fillDepth, int W, int H, Intr k, hipStream_t s);
void launch_bilateral_model_pyramid_batch(const float* depth, float* depthF, const TrackBatch& b, const float* fillDepth, int W, int H, Intr k,
                                          hipStream_t s);   // the depth filter + launch_model_pyramid_batch's work in one launch
// it = 0 seeds the states from the model poses (+ SO(3) rotation); it >= 1 finishes iteration it-1 whose pixel pass used nb_in blocks
void launch_icp_batch_solve(const TrackBatch& b, int it, int nb_in, const So3Result* so3_or_null, hipStream_t s);
// row_z: launch_row_zrange's output for this level (nullptr: no culling -- every workgroup walks its pixels)
void launch_icp_batch_pixels(const TrackBatch& b, int it, int level, const float* vmap_curr, const float* nmap_curr, int W, int H, Intr k,
                             float distThres, float angleThres, hipStream_t s, const float2* row_z = nullptr, bool fused = false, int nb_in = 0,
                             const So3Result* so3_or_null = nullptr);   // fused: launch_icp_batch_solve(b, it, nb_in, so3) is this launch's prologue
// depth range of every row of the frame's three vertex maps: out[0 .. H) level 0, [H .. H + H/2) level 1, [.. + H/4) level 2: {min z, max z} over the
// row's valid vertices ({+inf, -inf}: none)
void launch_row_zrange(const float* const vmap[3], int W, int H, float2* out, hipStream_t s);
// last solve (iteration n_it - 1) + Model pose / lastPose / statistics / jump rule
void launch_icp_batch_finalize(const TrackBatch& b, int n_it, int nb_in, const So3Result* so3_or_null, hipStream_t s);
// Last reduce+solve, then pose / lastPose / inverse / fusion weight update and host mirror.
// jump_limit > 0: object-model rule of MaskFusion.cpp:268-272 (|increment translation| > limit => pose->alive = 0)
void launch_icp_finalize(const float* partials_in, int nblocks_in, const GNState* state_in, PoseDev* pose,
                         PoseDev* host_mirror, float* log_out, float jump_limit, const So3Result* so3, hipStream_t s,
                         double* trace = nullptr, int n_it = 0);
// Stand-alone icpStep (parity tests): host-provided poses, output 32 floats.
void launch_icp_step_standalone(const float* Rcurr, const float* tcurr, const float* vc, const float* nc,
                                const float* Rpi, const float* tprev, Intr k, const float* vp, const float* np,
                                float distThres, float angleThres, int W, int H, float* partials, GNState* st2,
                                float* out32, hipStream_t s);

struct RgbCorr { int16_t u0, v0; float diff; };  // DataTerm (types.cuh:75-81) packed to 8 B; u0 < 0: no correspondence

struct RgbLevel {  // one pyramid level of RGBDOdometry's photometric inputs
    const int16_t* dIdx; const int16_t* dIdy;          // nextdIdx / nextdIdy
    const float* lastDepth; const float* nextDepth;    // Q1: both come from the model prediction
    const uint8_t* lastImage; const uint8_t* nextImage;
    int W, H;
    float minScale;        // minimumGradientMagnitudes[level]^2 / sobelScale^2
    float maxDepthDelta;   // 0.07
    const uint8_t* gate;   // optional: 1 where the iteration-independent tests of RGBResidual pass (border, 4x4 window of the
                           // next image > 0, gradient magnitude), written once per frame by the derivative kernel; nullptr =
                           // evaluate them per pixel
};

// ---------------- photometric term + SO(3) (mf_rgbd.hip, mf_odometry.hip) ----------------
void launch_intensity(const uint8_t* img, int channels, uint8_t* dst, int n, hipStream_t s);
void launch_pyrdown_u8(const uint8_t* src, uint8_t* dst, int sw, int sh, hipStream_t s);
void launch_derivative(const uint8_t* src, int16_t* dx, int16_t* dy, int W, int H, float minScale, uint8_t* gate /*or null*/, hipStream_t s);
// up to three independent small image jobs in ONE launch (mf_rgbd.hip): kind 0 pyrDownUcharGauss (W, H = source size), 1 pyrDownGaussF (source size),
// 2 computeDerivativeImages (+ gate image; W, H = image size)
struct SmallJob { int kind; const void* src; void* dst; void* dst2; uint8_t* gate; int W, H; float minScale; };
struct SmallJobs { SmallJob j[3]; int n; };
void launch_small_jobs(const SmallJobs& a, hipStream_t s);
// intensity pyramid (3 levels) + derivative / gate images of a frame in one launch; false (nothing launched) for sizes its tiling does not cover
bool launch_rgb_pyramid(const uint8_t* rgb, int W, int H, uint8_t* const gray[3], int16_t* const dIdx[3], int16_t* const dIdy[3], uint8_t* const gate[3],
                        const float minScale[3], bool derivatives, hipStream_t s);
// level 0 of a model's "last" depth / intensity pyramids (populateRGBDData of initRGBModel; Q1: initRGB re-uses the depth)
// frameToFrameRGB != 0 (and a fill-in image given): initRGBModel takes the fill-in image whatever the fill-in decision (Model.cpp:399-400)
void launch_rgbd_last_l0(const float4* predV, const float* fillDepth, const uint8_t* predGray, const uint8_t* fillGray,
                         const FrameDev* frame, float* depth0, uint8_t* image0, int n, hipStream_t s, int frameToFrameRGB = 0);
// scratch: so3_scratch_bytes(W2, H2) of device memory; returns -1 if the image needs more workgroups than the staging allows
size_t so3_scratch_bytes(int W2, int H2);
int launch_so3_prealign(const uint8_t* lastImage2, const uint8_t* nextImage2, int W2, int H2, Intr k2, So3Result* out, void* scratch,
                        hipStream_t s);
void launch_rgb_residual_only(const RgbLevel& L, const float* krk_kt, RgbCorr* corres, int* sums, hipStream_t s);
void launch_rgb_step_only(const RgbLevel& L, const RgbCorr* corres, float sigma, Intr k, float sobelScale, double* out32, hipStream_t s);
// One RGB-D Gauss-Newton iteration = two launches (see mf_odometry.hip)
struct RgbdLaunch {
    IcpLaunch icp;
    RgbLevel L;
    RgbCorr* corres;
    const float* rgb_partials_in; float* rgb_partials_out;
    const int2* cnt_in; int2* cnt_out;
    float icpWeight; int icpOn; int rgbOnly; float sobelScale;
    int level, prev_level;
    const So3Result* so3_in;
};
void launch_rgbd_iteration(const RgbdLaunch& l, hipStream_t s);
void launch_rgbd_finalize(const float* icp_partials, const float* rgb_partials, const int2* cnt, int nb, float icpWeight, int icpOn,
                          int rgbOnly, int rgbOn, int prev_level, const GNState* st_in, const So3Result* so3, PoseDev* pose,
                          PoseDev* host_mirror, float* log_out, float jump_limit, hipStream_t s);

// ---------------- surfels ----------------
void launch_init_surfels(const uint8_t* rgb, const float* depthRaw, const float* depthF, int W, int H, Intr k,
                         float maxDepth, const FrameDev* frame, float4* rec /*[P][3]*/, uint8_t* flags, hipStream_t s);
// Grid of the four grid-stride surfel kernels below: 2048 workgroups by default; a model that is known to be small (an object model
// of a few thousand surfels) may be launched with fewer -- the loops are grid-stride, the result does not depend on the grid.
constexpr int kSurfelGridBlocks = 2048;
// vis: nullptr = every surfel; else the runs launch_cull found possibly visible
void launch_index_scatter(Surfels src, const FrameDev* frame, const PoseDev* pose, int W, int H, Intr k,
                          float maxDepth, int timeDelta, unsigned long long* keys, bool transposed /* column-major key image */, hipStream_t s,
                          int blocks = kSurfelGridBlocks, const VisList* vis = nullptr);
// run table of a DENSE buffer s (slots [0, frame->count)): fixed runs of kRun slots (after Model::initialise / an uploaded map / a compaction; the
// in-place clean pass maintains it afterwards).  refresh: s has a table -- only the entries' contents are recomputed from the surfels
void launch_run_table(Surfels s, FrameDev* frame, hipStream_t st, bool refresh = false);
size_t run_table_runs(long capacity, long pixels);      // runs a buffer's table can hold (the host compacts the buffer before they run out)
size_t run_table_entries(long capacity, long pixels);   // int4 entries of that table (kBoxStride per run)
// the runs of s whose box meets the viewing frustum (image bounds + 2 px, -1 cm .. maxDepth + 1 cm) and that hold a surfel seen within
// timeDelta: their indices -> list (any order), their number -> count[0]; ctl: 2 ints, zero between launches
void launch_cull(Surfels s, const FrameDev* frame, const PoseDev* pose, int W, int H, Intr k, float maxDepth, int timeDelta, int* list, int* count,
                 int* ctl, int max_runs, hipStream_t st);
// packed == nullptr: index / vertConf / normRad (+ colorTime if ct != nullptr) images; else one 32 B record per texel.  decay_stats (with packed,
// optional): kDecayStats ints the pass accumulates the frame's mask-disagreement depth ranges into for launch_cull_clean (armed by
// launch_arm_decay_stats / k_cull_clean; maskID: the model's id)
void launch_index_resolve(Surfels src, const FrameDev* frame, const PoseDev* pose, unsigned long long* keys, int W, int H, int* index,
                          float4* vc, float4* nr, float4* ct /*or null*/, float4* packed /*or null: the packed column-major map instead of the row-major maps*/,
                          const float* depthF, const uint8_t* mask, uint8_t* maskT /* with packed: the frame planes that travel with it */,
                          bool keys_transposed /* the order the scatter used */, hipStream_t s, int* decay_stats = nullptr, int maskID = 0);
void launch_fuse_data(const uint8_t* rgb, const float* depthRaw, const float* depthF, const uint8_t* mask,
                      int maskID, const FrameDev* frame, const PoseDev* pose, float weightMultiplier, float maxDepth,
                      int W, int H, Intr k, const int* index, const float4* vc, const float4* nr, uint8_t* cand_op,
                      float4* cand_rec, int* upd_first, int* cand_best, hipStream_t s, int bboxLimit = 1);
// update.vert IN PLACE: one thread per candidate, the winning candidate of a surfel (upd_first) merges into it where it stands
void launch_fuse_update(Surfels s, const FrameDev* frame, int* upd_first, const uint8_t* cand_op, const int* cand_best, const float4* cand_rec,
                        int W, int H, hipStream_t st);
// the same as a copy src -> dst over the whole buffer; keys_or_null != nullptr: the index-map scatter of the pass that feeds clean rides on it
// (the form for small maps)
void launch_fuse_update_copy(Surfels src, Surfels dst, const FrameDev* frame, int* upd_first, const float4* cand_rec, const PoseDev* pose,
                             int W, int H, Intr k, float maxDepth, int timeDelta, unsigned long long* keys_or_null, bool transposed,
                             hipStream_t s, int blocks = kSurfelGridBlocks);
// Model::clean (copy_unstable.vert:53-157, Model.cpp:649-772).  What every form is given: the shader's uniforms and textures + the frame's
// candidates (new-surfel records of the association pass) + the two-launch form's intermediates
struct CleanIn {
    FrameDev* frame; const PoseDev* pose; int W, H; Intr k; int timeDelta; float confThreshold, outlierCoeff; int maskID;
    const int* index; const float4* vc; const float4* ct;    // the index map as separate images, or
    const float4* packed; const uint8_t* maskT;              // ... the packed column-major map with the mask beside it
    const float* depthF; const uint8_t* mask;
    const uint8_t* cand_op; const float4* cand_rec;
    uint8_t* flags; float* newconf; int* block_counts;       // [elements], [elements], [kCompactBlocks]
    int* host_count;                                         // pinned mirror of the surfel count
    unsigned long long* host_append; unsigned seq;           // in place: pinned mirror of where the buffer ends after THIS pass, tagged with the pass's
                                                             // sequence number (append_mirror() below); nullptr: none
    bool transposed;                                         // layout of index / vc / ct / packed: column-major
    bool literalWindow;                                      // the window walked with the shader's own fp32 trip count
};
// seq (20 bits) | runs of the table (18 bits) | first slot behind the last run (26 bits): one 64-bit store the host can read at any time --
// its bounds on what the device has appended then start from a recent exact value instead of the last compaction (mf_frame.inl: prepare_in_place)
constexpr int kAppendSeqBits = 20, kAppendRunBits = 18, kAppendPhysBits = 26;
__host__ __device__ inline unsigned long long append_mirror(unsigned seq, int runs, int phys) {
    return ((unsigned long long)(seq & ((1u << kAppendSeqBits) - 1u)) << (kAppendRunBits + kAppendPhysBits)) | ((unsigned long long)(unsigned)runs << kAppendPhysBits) |
           (unsigned long long)(unsigned)phys;
}
// two launches (flags + ordered copy), src -> dst: dst is a DENSE buffer without a run table.  src must be dense.
int compact_blocks_for(long elements);   // workgroups of an ordered-compaction launch for that many elements (<= kCompactBlocks)
void launch_clean_small(const CleanIn& in, Surfels src, Surfels dst, hipStream_t s, int blocks = kCompactBlocks);
// IN PLACE (mf_surfel.hip "clean, in place"; buf has a run table): launch_clean_runs tests the surfels of the listed runs (nullptr: of every run)
// and compacts each run where it stands; launch_clean_append tests the frame's candidates and appends the survivors behind the last run.
// ctl: kCleanCtlInts ints, zero between launches; blocks: clean_runs_grid(elements expected)
constexpr int kCleanGridMax = 2048;
constexpr int kCleanCtlInts = 8;
int clean_runs_grid(long elements);
void launch_clean_runs(const CleanIn& in, Surfels buf, const VisList* runs, int* ctl, int blocks, hipStream_t s);
void launch_clean_append(const CleanIn& in, Surfels buf, hipStream_t s);
// the runs launch_clean_runs has to visit for the BACKGROUND-style culling of a model: those in which one of the pass's rules can apply at all
// (decay_stats: launch_index_resolve's, re-armed here; list / count / ctl as launch_cull)
constexpr int kDecayStats = 9;     // ResolveOut::decay_stats (mf_surfel.hip): {min, max} x {left, right, top, bottom border} + min over all foreign texels
void launch_arm_decay_stats(int* stats /*[kDecayStats]*/, hipStream_t st);   // once, when the statistics block is allocated
void launch_cull_clean(Surfels s, const FrameDev* frame, const PoseDev* pose, int W, int H, Intr k, int timeDelta, float confThreshold, int* decay_stats,
                       int* list, int* count, int* ctl, int max_runs, hipStream_t st);
// compaction of a sparse buffer: the surfels of src's runs -> dst, dense (no table); offs: scratch of run_table_runs() + 1 ints
void launch_densify(Surfels src, Surfels dst, FrameDev* frame, int* offs, int* host_count, hipStream_t st);
// generic ordered compaction of [n_dev] records (3 x float4 each, record-major) -> surfels, sets frame->count
void launch_compact_records(const float4* rec, const uint8_t* flags, int n, Surfels dst, FrameDev* frame,
                            int* block_counts, int* host_count_mirror, hipStream_t s);
void launch_splat_scatter(Surfels src, const FrameDev* frame, const PoseDev* pose, int W, int H, Intr k,
                          float maxDepth, float confThreshold, int timeDelta, unsigned long long* keys,
                          hipStream_t s, int blocks = kSurfelGridBlocks);
void launch_splat_resolve(Surfels src, const PoseDev* pose, unsigned long long* keys, int W, int H, Intr k,
                          float4* predV, float4* predN, uchar4* predImage, uint16_t* predTime, FrameDev* frame,
                          const uint8_t* rgb /*or null*/, uint8_t* predGray /*or null*/, uint8_t* fillGray /*or null*/,
                          hipStream_t s, int fillPassthrough = 0 /* fill_rgb.frag's `passthrough` (frameToFrameRGB) */);
// Tiled form of scatter + resolve (mf_splat.hip): bins surfels to 16x16 tiles, z-test in LDS, writes the maps directly.
// tile_count must be zero on the first call (it is left zero).  Returns -1 if the image has too many tiles for the LDS
// histograms (use the scatter form then).
size_t splat_tiles_scratch_ints(int W, int H);
struct SplatTuning { int sprite_lanes = 4, tile_threads = 512, tile_h = 24; };   // A/B knobs of the tile passes ("spriteLanes", "tileThreads"), owned by the context
// The end-of-frame bookkeeping (k_frame_advance: pose log entry, fill-in decision for the next tracking step, tick++, host mirror) as
// the epilogue of the tiled prediction: its last workgroup to finish runs it, one launch less per model and frame.
struct FrameAdvance { FrameDev* host_mirror; const PoseDev* bg_pose; float* log_slot; };
int launch_splat_tiled(Surfels src, FrameDev* frame, const PoseDev* pose, int W, int H, Intr k, float maxDepth, float confThreshold,
                       int timeDelta, int* tile_count, int* entries /*[tiles][entries_cap / tiles]*/, int entries_cap,
                       float4* rec0 /*[src.cap]*/, float4* rec1 /*[src.cap]*/, void* bbox /*[src.cap] x 8 B*/, float4* predV, float4* predN,
                       uchar4* predImage, uint16_t* predTime, const uint8_t* rgb, uint8_t* predGray, uint8_t* fillGray, hipStream_t s,
                       const FrameAdvance* advance = nullptr, int fillPassthrough = 0, unsigned long long* prof = nullptr /*[tiles][8] stamps*/,
                       SplatTuning tune = SplatTuning(), const VisList* vis = nullptr);
// ---- the surfel passes of ALL object models of a frame, one launch per pass (grid.z = model) ----
// An object model holds a few thousand surfels: each of its ~11 per-frame launches is pure launch latency (~85 us per object and frame in
// round 2's profile).  The batched kernels are the single-model kernels' bodies called with one model's arguments, picked from a device
// array by blockIdx.z; every model brings its own scratch (index maps, key image, candidate records, ...), which the single-model path shares.
struct ObjPassArgs {
    Surfels a, b;                      // live buffer when the frame's fusion starts / the other one (small models: fuse a -> b, clean b -> a; from
                                       // inPlaceElements on: fuse in place in a, clean a -> b, then b is live; big ones: everything in place in a)
    FrameDev* frame; PoseDev* pose;
    int maskID; float confThreshold, fuseMaxDepth, weightMultiplier;
    unsigned long long* keys; int* index; float4* ivc; float4* inr; float4* iclean;
    uint8_t* cand_op; float4* cand_rec; int* upd_first; int* cand_best; int* clean_ctl; int* host_count;
    unsigned long long* host_append; unsigned clean_seq;   // CleanIn::host_append / seq
    uint8_t* flags; float* newconf; int* block_counts;   // the two-launch clean form's intermediates
    float4* predV; float4* predN; uchar4* predImage; uint16_t* predTime; uint8_t* predGray;
    FrameDev* host_frame; float* log_slot;
    unsigned global_payload;           // GlobalProjection: order << 8 | id
};
struct ObjBatch {
    const ObjPassArgs* m; int n;
    int W, H; Intr k; float maxDepthProcessed, globalMaxDepth; int timeDelta; float outlierCoeff; int cleanLiteral, bboxLimit;
    int denseSprites;                  // 1: a model of the batch is above inPlaceElements -- the sprite passes (prediction, GlobalProjection) run one thread
                                       // per surfel: such a map holds sub-pixel sprites, and four lanes per surfel repeat its set-up four times
    int updateCopy;                    // 1: every model of the batch is below inPlaceElements -- update.vert as a copy a -> b, clean b -> a
    int cleanSmall;                    // 1: the two-launch clean form src -> dst; 0: every model of the batch has a run table -- clean in place
    const uint8_t* rgb; const float* depthRaw; const float* depthF; const uint8_t* mask; const PoseDev* bg_pose;
    uint8_t* maskT;                    // the mask in column-major order, beside the packed maps (every model's resolve writes the same bytes)
    unsigned long long* global_keys;
};
void launch_obj_global_scatter(const ObjBatch& b, int blocks, hipStream_t s);          // GlobalProjection of every object model (mf_segment.hip)
void launch_obj_fuse_clean(const ObjBatch& b, int blocks, int clean_blocks, hipStream_t s, int compact_blocks = kCompactBlocks);   // predictIndices -> fuse -> predictIndices -> clean
void launch_obj_predict_advance(const ObjBatch& b, int blocks, hipStream_t s);         // combinedPredict (scatter form) + the end-of-frame bookkeeping
void launch_fill_keys(unsigned long long* keys, int n, hipStream_t s);
void launch_fill_int(int* p, int v, int n, hipStream_t s);
// ---------------- multi-model coupling (mf_segment.hip) ----------------
void launch_edge_map(const float* vmap, const float* nmap, float* out, int W, int H, float wD, float wC, hipStream_t s);
void launch_edge_binary(const float* edge, uint8_t* out, uint8_t* tmp, int W, int H, float threshold, int radius,
                        int iterations, hipStream_t s);
void launch_global_scatter(Surfels src, const FrameDev* frame, const PoseDev* pose, int W, int H, Intr k, float maxDepth,
                           float confThreshold, int timeDelta, int order, int id, unsigned long long* keys, hipStream_t s,
                           int blocks = kSurfelGridBlocks);
void launch_global_resolve(unsigned long long* keys, uint8_t* ids, int P, hipStream_t s);
int gn_solve_standalone(const double* sys29, const double* resultRt16, const float* Rprev9, const float* tprev3, double* x_serial, double* x_wave,
                        double* resultRt_out, float* Rcurr9, float* tcurr3, float* stats2, hipStream_t s);
int launch_global_tiled(Surfels src, FrameDev* frame, const PoseDev* pose, int W, int H, Intr k, float maxDepth, float confThreshold,
                        int timeDelta, int order, int id, int* tile_count, int* entries, int entries_cap, float4* rec0, float4* rec1, void* bbox,
                        unsigned long long* keys, hipStream_t s, SplatTuning tune = SplatTuning(), const VisList* vis = nullptr);
void launch_spawn_pose(PoseDev* obj, const PoseDev* bg, FrameDev* objFrame, const FrameDev* bgFrame, PoseDev* host_mirror,
                       hipStream_t s);
void launch_static_pose(PoseDev* obj, const PoseDev* bg, PoseDev* host_mirror, hipStream_t s);
void launch_override_pose(PoseDev* pose, const float* in_pose16_colmajor, int mode, PoseDev* host_mirror, hipStream_t s);
void launch_model_state(const PoseDev* pose, const FrameDev* frame, float* out16, hipStream_t s);

// end-of-frame bookkeeping: tick++, cover -> useFillIn decision for the next frame
// ... and the pose-log entry of this frame (MaskFusion.cpp:580-596) when log != nullptr
void launch_pose_log(const PoseDev* pose, const PoseDev* bg_pose /*nullptr: the background itself*/, float* slot, hipStream_t s);
void launch_frame_advance(FrameDev* frame, int W, int H, FrameDev* host_mirror, const PoseDev* pose, const PoseDev* bg_pose,
                          float* log_slot, hipStream_t s);

}  // namespace mf

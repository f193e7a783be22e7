// mf_context.hip -- the mf_ctx object and the C ABI of include/maskfusion_amd.h.
//
// One context = one GPU, one HIP stream, one model list (background + object models).  MaskFusion::processFrame
// (Core/MaskFusion.cpp:200-607) becomes a fixed sequence of asynchronous launches on that stream.  With a single model
// ("-static") the host never reads anything back inside a frame; with multiple models there is exactly one host
// synchronisation per frame, where the reference also leaves the GPU (MfSegmentation's CPU stage).
#include "../../include/maskfusion_amd.h"
#include "mf_internal.h"
#include "mf_labels.h"

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <string.h>
#include <memory>
#include <chrono>
#include <string>
#include <vector>

using namespace mf;

namespace {

// One surfel model (Core/Model/Model.h): persistent per-model device state.  Everything else (maps, index map, candidate
// buffers, z-buffer keys) is scratch shared by all models: the reference's per-model passes never overlap in time.
struct ModelState {
    int id = 0, classID = -1;
    bool isStatic = true, allowFillIn = false;
    unsigned age = 0;
    float confThr = 0.f, maxDepth = FLT_MAX;
    Surfels surf[2];
    int cur = 0, cap = 0;
    PoseDev* d_pose = nullptr; FrameDev* d_frame = nullptr;
    float4* d_predV = nullptr; float4* d_predN = nullptr; uchar4* d_predImage = nullptr; uint16_t* d_predTime = nullptr;
    uint8_t* d_predGray = nullptr; uint8_t* d_fillGray = nullptr;  // intensity of the RGB projection / of the fill-in image
    bool pred_gray_valid = false;          // the last prediction of this model wrote d_predGray (/ d_fillGray): it ran with a photometric term configured
    // RGBDOdometry of the model (Model::frameToModel): model-side pyramid, Gauss-Newton state, per-workgroup partial sums
    float* d_vmap_g[3] = {}; float* d_nmap_g[3] = {}; GNState* d_gn = nullptr; float* d_partials[2] = {nullptr, nullptr};
    TrackModelDev* d_track = nullptr;      // the block the batched tracker kernels find all of that through
    float* d_icp_log = nullptr;            // [20][32] reduced systems of the model's last tracking step, one row per iteration (debug tap "icp_log")
    double* d_gn_trace = nullptr;          // [20][kGnTraceRow] systems in fp64 + the state every iteration used (debug tap "gn_trace"; geometric loop)
    // object models: private scratch of the surfel passes, so that the passes of ALL objects of a frame can be one launch each ("batchObjectPasses")
    struct ObjScratch {
        unsigned long long* keys = nullptr; int* index = nullptr; float4* ivc = nullptr; float4* inr = nullptr; float4* iclean = nullptr;
        uint8_t* cand_op = nullptr; float4* cand_rec = nullptr; int* upd_first = nullptr; int* cand_best = nullptr;
        unsigned long long* scan_state = nullptr; int* clean_ctl = nullptr;
    } scr;
    float* d_poselog = nullptr;            // Model::poseLog on the device: ring of [cap][8] floats (t, q xyzw, pad)
    std::vector<int64_t> log_ts;           // timestamps of the entries (host side of PoseLogItem)
    PoseDev* h_pose = nullptr; FrameDev* h_frame = nullptr; int* h_count = nullptr;
    TrackModelDev track_host;              // host copy of *d_track (pose_host is filled in once h_pose exists)
    std::vector<void*> allocs;
    ~ModelState() {
        for (void* p : allocs) (void)hipFree(p);
        if (h_pose) (void)hipHostFree(h_pose);
        if (h_frame) (void)hipHostFree(h_frame);
        if (h_count) (void)hipHostFree(h_count);
    }
};

// Device scratch of the GPU label stage (mf_labels_gpu.hip)
struct LabelsScratch {
    int P = 0, table_cap = 0;
    uint8_t* ignoreMap = nullptr; uint8_t* tmp_u8 = nullptr;
    int* L = nullptr; int* compId = nullptr; int* area = nullptr; int* lab[2] = {nullptr, nullptr}; int* blockCounts = nullptr;
    int4* bbox = nullptr;
    int* compMask = nullptr; int* compModel = nullptr; int* compToMask = nullptr; int* compFollow = nullptr;
    unsigned* overlap = nullptr; void* tables = nullptr;
    int* d_small = nullptr;      // class ids [256] | model ids [64] | model classes [64]
    const PoseDev** d_poses = nullptr;   // [64]
    int* h_small = nullptr; const PoseDev** h_poses = nullptr; int* h_result = nullptr;   // pinned
    std::vector<void*> dev, host;
    ~LabelsScratch() {
        for (void* p : dev) (void)hipFree(p);
        for (void* p : host) (void)hipHostFree(p);
    }
    template <typename T> bool dalloc(T** p, size_t n) {
        void* q = nullptr;
        if (hipMalloc(&q, n * sizeof(T)) != hipSuccess) return false;
        if (hipMemset(q, 0, n * sizeof(T)) != hipSuccess) return false;
        dev.push_back(q); *p = reinterpret_cast<T*>(q);
        return true;
    }
    template <typename T> bool halloc(T** p, size_t n) {
        void* q = nullptr;
        if (hipHostMalloc(&q, n * sizeof(T)) != hipSuccess) return false;
        memset(q, 0, n * sizeof(T));
        host.push_back(q); *p = reinterpret_cast<T*>(q);
        return true;
    }
    bool init(int P_) {
        P = P_;
        table_cap = 8 << 20;   // components x (masks | models) entries per vote table; exceeding it is reported, never truncated
        return dalloc(&ignoreMap, (size_t)P) && dalloc(&tmp_u8, (size_t)P) && dalloc(&L, (size_t)P) && dalloc(&compId, (size_t)P) &&
               dalloc(&area, (size_t)P + 1) && dalloc(&bbox, (size_t)P + 1) && dalloc(&lab[0], (size_t)P) && dalloc(&lab[1], (size_t)P) &&
               dalloc(&blockCounts, (size_t)(P + 255) / 256) && dalloc(&compMask, (size_t)table_cap) &&
               dalloc(&compModel, (size_t)table_cap) && dalloc(&compToMask, (size_t)P + 1) && dalloc(&compFollow, (size_t)P + 1) &&
               dalloc(&overlap, (size_t)64 * 256) && dalloc((char**)&tables, labels_gpu_table_bytes()) && dalloc(&d_small, 384) &&
               dalloc(&d_poses, 64) && halloc(&h_small, 384) && halloc(&h_poses, 64) && halloc(&h_result, 4);
    }
    // uploads the small per-frame tables and enqueues the stage; models: {id, classID, device pose}
    int enqueue(const SegParams& prm, int W, int H, const uint8_t* d_binary, const float* d_depth, const uint8_t* d_mask,
                const int32_t* class_ids, int n_masks, const uint8_t* d_proj, const std::vector<SegModelInfo>& models,
                const std::vector<const PoseDev*>& poses, int nextModelID, bool allowNew, uint8_t* d_full, hipStream_t s) {
        if (models.size() > 64 || n_masks > 256) return MF_EINVAL;
        memset(h_small, 0, 384 * sizeof(int));
        for (int k = 0; k < n_masks; ++k) h_small[k] = class_ids[k];
        for (size_t m = 0; m < models.size(); ++m) { h_small[256 + m] = models[m].id; h_small[320 + m] = models[m].classID; h_poses[m] = poses[m]; }
        if (hipMemcpyAsync(d_small, h_small, 384 * sizeof(int), hipMemcpyHostToDevice, s) != hipSuccess ||
            hipMemcpyAsync(d_poses, h_poses, 64 * sizeof(PoseDev*), hipMemcpyHostToDevice, s) != hipSuccess)
            return MF_EHIP;
        LabelsGpuArgs a;
        a.prm = prm; a.W = W; a.H = H; a.binary = d_binary; a.depth = d_depth; a.mask = d_mask; a.proj = d_proj;
        a.class_ids = d_small; a.nMasks = n_masks; a.model_ids = d_small + 256; a.model_cls = d_small + 320; a.model_poses = d_poses;
        a.nModels = (int)models.size(); a.nextModelID = nextModelID; a.allowNew = allowNew; a.ignoreMap = ignoreMap; a.full = d_full;
        a.tmp_u8 = tmp_u8; a.L = L; a.compId = compId; a.area = area; a.bbox = bbox; a.lab[0] = lab[0]; a.lab[1] = lab[1]; a.blockCounts = blockCounts;
        a.compMask = compMask; a.compModel = compModel; a.table_cap = table_cap; a.compToMask = compToMask; a.compFollow = compFollow;
        a.overlap = overlap; a.tables = tables; a.result_host = h_result;
        launch_labels_gpu(a, s);
        return MF_OK;
    }
};

}  // namespace

struct mf_ctx {
    mf_config cfg;
    int W, H, P;
    Intr K;
    hipStream_t stream = nullptr;      // tracking + fusion (everything that depends on the pose)
    hipStream_t stream_pre = nullptr;  // per-frame preprocessing, which depends on nothing but the frame: it runs one frame
                                       // ahead under the latency-bound Gauss-Newton loop of the previous frame
    hipEvent_t ev_pre_done[2] = {nullptr, nullptr};    // preprocessing of frame k finished      (pre -> main)
    hipEvent_t ev_main_done[2] = {nullptr, nullptr};   // frame k finished tracking (so k-1 is complete)  (main -> pre)
    bool clean_literal = true;                         // Model::clean walks its window with the shader text's fp32 trip count ("cleanLiteralWindow")
    bool global_tiles = true;                          // A/B + test knob ("globalTiles"): 0 = every model through k_global_scatter
    bool early_bg_fusion = true;                       // A/B knob ("earlyBackgroundFusion"): 0 = the host visit drains the stream
    hipEvent_t ev_labels = nullptr;                    // the label stage of this frame has written its result words
    hipEvent_t ev_staged = nullptr;                    // mf_stage_frame_dev: the producers of the staged buffers, ordered on `stream`, have run (main -> pre)
    long frame_no = 0;
    long bg_fused_frame = -1;          // mf_fuse_background has fused the background of this staged frame (mf_fuse_models then skips it)
    bool labels_pending = false;       // between mf_perform_segmentation_begin and _end
    int lastF = 0;
    int overlap = 0;                   // 1: preprocessing on stream_pre, one frame ahead ("overlapPreprocessing").  Measured on
                                       // MI355X (tools/host_rate.py, same box A/B): 473-483 us/frame either way -- the filter's
                                       // waves delay the whole-CU Gauss-Newton / fusion workgroups (+30 us on the main chain)
                                       // and the two cross-queue barriers cost the rest, so it is off by default.
    std::string err;
    int host_tick = 1;
    bool map_ready = false;            // the background map exists (first frame processed, Model::initialise or an uploaded map)
    bool tracked_once = false;         // a tracking step has run (its stage timings are meaningful)
    bool timings_on = false, icp_prof_on = false;
    // A/B switches for object models (a few thousand surfels each; their per-frame cost is launch overhead).  Measured on MI355X, 12-model S2
    // scene (profiles/r03a_bench_2s_object_switches.txt): 394 frames/s without, 395 / 399 with one, 407 with both -> on since round 3:
    bool object_small_grids = true;                    // "objectSmallGrids": the grid-stride surfel kernels with a grid sized from the model's last known count
    bool object_scatter_splat = true;                  // "objectScatterSplat": object models are predicted with the scatter form instead of tile lists
#ifndef MF_DEFAULT_LITERAL_FUSION_WEIGHT
#define MF_DEFAULT_LITERAL_FUSION_WEIGHT 1             // default since round 3 (finding F5: the reference's own arithmetic); 0 = the accurate double log map
#endif
    bool weight_literal = MF_DEFAULT_LITERAL_FUSION_WEIGHT != 0;   // "literalFusionWeight": Model::rodrigues2 with the reference's float trace (finding F5)
    bool bbox_limit = true;                            // "objectBoundingBoxLimit": Model::fuse limits an object's depth by lastBoundingBox (Model.cpp:480-501; upstream: whenever its GUI draws the models)
    bool batch_objects = true;                         // "batchObjectPasses": the surfel passes of all object models of a frame as one launch per pass (grid.z = model)
    static constexpr int kObjArgSlots = 8;             // argument arrays of the batched launches: pinned staging + device copy, a small ring (two per frame)
    ObjPassArgs* d_obj_args[kObjArgSlots] = {}; ObjPassArgs* h_obj_args[kObjArgSlots] = {}; hipEvent_t ev_obj_args[kObjArgSlots] = {};
    unsigned obj_arg_slot = 0;
    bool ftf_rgb = false;                              // MaskFusion::frameToFrameRGB ("-ftf"; Model.cpp:399-400,981): the photometric term tracks against the previous RAW frame
    double host_us[5] = {0, 0, 0, 0, 0}; long host_calls = 0;   // mf_process_frame's host time: wait for the slot + staging copy | upload enqueue | frame enqueue | whole call | the wait alone ("hostStageUs" ... "hostWaitUs")
    // "hostLockstep" (default on): the call waits for frame k-2 to have RUN before it enqueues frame k's upload, so the host is at most two
    // frames ahead and the upload's dependency (the frame that last read the device block) is already satisfied when it is enqueued.  Left
    // to run three frames ahead, every frame pays ~40 us for cross-queue dependencies that are still open at enqueue time
    // (profiles/r04o_host_ab.json: 352 -> 324 us per frame; device-resident: 305)
    bool host_lockstep = true;
    // "hostWaitUpload" (default on, needs hostLockstep): the call also waits for its OWN upload (~60 us of the ~300 the host has to spare per
    // frame) before it enqueues the frame, which then needs no cross-queue wait at all: 322 -> 317 us per frame (profiles/r04q_host_ab.json;
    // device-resident frames: 305).
    bool host_wait_upload = true;

    // frame-level
    uint8_t* d_rgb = nullptr; float* d_depth = nullptr; uint8_t* d_mask_in = nullptr; uint8_t* d_zero_mask = nullptr;
    // mf_process_frame (host pointers, the reference's FrameData boundary) without a synchronisation per frame ("hostInputAsync", default on):
    // the caller's buffers are copied into a pinned staging slot (so they may be reused at once), the slot goes to the device on its own
    // stream under the previous frame's kernels, and the call returns when the frame is enqueued; every getter synchronises, as they do
    // behind mf_process_frame_dev.  Two slots each: frame k+2 reuses what frame k used.
    bool host_async = true;
    hipStream_t stream_in = nullptr;
    uint8_t* h_in[2] = {nullptr, nullptr};                 // pinned: depth (4P) | rgb (3P) | mask (P), each part 256-B aligned: ONE upload per frame
    uint8_t* d_in_block[2] = {nullptr, nullptr};           // the same layout in HBM; d_in_* below are views into it
    size_t in_off_rgb = 0, in_off_mask = 0, in_bytes = 0;
    uint8_t* d_in_rgb[2] = {}; float* d_in_depth[2] = {}; uint8_t* d_in_mask[2] = {};   // slot 0 = d_rgb / d_depth / d_mask_in
    hipEvent_t ev_in_copied[2] = {nullptr, nullptr};       // the slot's H2D copies have completed   (in -> main / pre, and the host before it refills the slot)
    hipEvent_t ev_in_consumed[2] = {nullptr, nullptr};     // the frame that read the slot has been processed   (main -> in)
    unsigned in_slot = 0;
    uint8_t* d_mask_tex = nullptr;  // textureMask: the last full segmentation (Core/MaskFusion.cpp:297)
    float* d_depthF[3] = {nullptr, nullptr, nullptr};  // ring: frame k filters into [k % 3], fill-in reads [(k - 1) % 3]
    float* d_vmap[2][3] = {}; float* d_nmap[2][3] = {};
    // shared scratch
    // photometric term + SO(3) (a5, a8-a10, a12)
    uint8_t* d_gray[2][3] = {};            // intensity pyramid of the frame, by frame parity ([prev] = lastNextImage)
    long gray_frame[2] = {-1, -1};         // frame_no each set was computed for
    long deriv_frame = -1;                 // frame_no the derivative images / gate images (d_dIdx, d_dIdy, d_rgb_gate) were computed for
    int16_t* d_dIdx[3] = {}; int16_t* d_dIdy[3] = {}; uint8_t* d_rgb_gate[3] = {};
    float* d_lastDepth[3] = {}; uint8_t* d_lastImage[3] = {};   // per-model scratch: populateRGBDData(last)
    RgbCorr* d_corres = nullptr; float* d_rgb_partials[2] = {nullptr, nullptr}; int2* d_cnt[2] = {nullptr, nullptr};
    So3Result* d_so3 = nullptr; char* d_so3_scratch = nullptr;
    // tiled splat prediction (mf_splat.hip)
    SplatTuning splat_tune;                // "spriteLanes" / "tileThreads" of THIS context
    int* d_tile_count = nullptr; int* d_tile_entries = nullptr; int tile_entries_cap = 0; int tile_entries_alloc = 0; int splat_tiles = 1;
    float4* d_splat_rec0 = nullptr; float4* d_splat_rec1 = nullptr; uint2* d_splat_bbox = nullptr;   // per-surfel sprite set-up
    const uint8_t* cur_rgb = nullptr;      // device rgb of the frame being processed (fill-in intensity at predict time)
    const float* cur_depth = nullptr;      // device raw depth of the frame being processed / staged (Model-level entry points)
    int batch_tracking = 1;                // 0: track the models one after the other even when a batch is possible ("batchTracking")
    int model_api_packed = 0;              // mf_model_predict_indices also builds the packed column-major map clean() uses in-frame
    std::vector<int32_t> mask_classes;     // FrameData::classIDs for mf_process_frame_dev (mf_set_mask_class_ids)
    std::vector<int> trackable;            // MaskFusion::trackableClassIds (empty: every class is trackable)
    // pose logs of dropped models (MaskFusion::inactiveModels, exportPoses).  A drop must not stall the frame it happens in (a scene of tracked
    // objects drops and re-spawns one every few frames): the entries are copied device-to-device, in stream order, into an arena allocated
    // once (`arena_off` / `n` say where) and only read back when somebody exports them; `p` is filled then (or at once when the arena is full)
    struct RetiredLog { int id; std::vector<int64_t> ts; std::vector<float> p; size_t arena_off = 0, n = 0; bool in_arena = false; };
    std::vector<RetiredLog> retired;
    float* d_retired = nullptr; size_t retired_cap = 0, retired_used = 0;   // [retired_cap][8] floats
    std::vector<std::unique_ptr<ModelState>> pool;   // MaskFusion::preallocatedModels (buffers allocated ahead of the spawn)
    unsigned long long* d_keys = nullptr;
    int* d_index = nullptr; float4* d_ivc = nullptr; float4* d_ict = nullptr; float4* d_inr = nullptr;
    float4* d_iclean = nullptr;            // packed column-major index map of the clean pass: 2 x float4 per texel
    uint8_t* d_cand_op = nullptr; float4* d_cand_rec = nullptr; int* d_upd_first = nullptr;
    uint8_t* d_flags = nullptr; float* d_newconf = nullptr; int* d_block_counts = nullptr;
    int* d_cand_best = nullptr;            // surfel a merge candidate was associated with (fuse_data -> fuse_update)
    unsigned long long* d_scan_state = nullptr; int* d_clean_ctl = nullptr; unsigned clean_epoch = 0;   // Model::clean's decoupled look-back (mf_surfel.hip)
    // visibility list of the projection passes (Surfels::box, k_cull): the runs of ONE buffer that can be in view under ONE pose; vis_tag says whose
    int* d_vis_list = nullptr; int* d_vis_count = nullptr; int* d_cull_ctl = nullptr; int vis_max_runs = 0;
    struct { const void* model = nullptr; long frame = -1; int cur = -1; } vis_tag;
    int ticket_lanes = 1;                  // ticket counters of the clean pass: min(kCleanTicketLanes, compute units of the device)
    bool cull_runs = true;                 // "cullRuns": 0 = every projection pass streams the whole buffer (A/B switch, executable specification)
    int cull_min_surfels = 2000000;        // "cullMinSurfels": maps below this size are streamed whole
    unsigned long long* d_icp_prof = nullptr;
    unsigned long long* d_splat_prof = nullptr; bool splat_prof_on = false;   // "splatProfile": [tiles][8] stamps of the background's tile pass
    // multi-model coupling
    float* d_edge = nullptr; uint8_t* d_bin = nullptr; uint8_t* d_tmp_u8 = nullptr; uint8_t* d_proj_ids = nullptr;
    uint8_t* h_bin = nullptr; uint8_t* h_ids = nullptr; float* h_depth = nullptr; uint8_t* h_mask = nullptr; uint8_t* h_full = nullptr;
    std::vector<uint8_t> ignoreMap;
    std::unique_ptr<LabelsScratch> labels;   // device label stage ("gpuLabels", default on)
    int gpu_labels = 1;
    SegParams seg;
    int nextID = 0, spawnOffset = 0;
    int cap_max = 0;

    std::vector<std::unique_ptr<ModelState>> models;

    hipEvent_t ev[MF_N_TIMINGS + 1] = {};
    hipEvent_t ev_mm[5] = {};                    // multi-model frame: after global projection | label stage | background fuse+clean | first object pass | (spare)
    bool mm_marked = false; float mm_host_wait_ms = 0.f;
    hipEvent_t ev_icp[2] = {nullptr, nullptr};   // first / after-last Gauss-Newton launch of the background model
    hipEvent_t ev_icp_mid = nullptr;             // ... and right before its first level-0 iteration (coarse levels | level 0)
    bool icp_mid_recorded = false;
    float last_ms[MF_N_TIMINGS] = {};
    std::vector<void*> allocs;
    std::vector<void*> host_allocs;
};

#define MF_HIP(ctx, call)                                                                         \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            char buf_[512];                                                                       \
            snprintf(buf_, sizeof(buf_), "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            if (ctx) (ctx)->err = buf_;                                                           \
            return MF_EHIP;                                                                       \
        }                                                                                         \
    } while (0)

template <typename T>
static int dev_alloc(mf_ctx* c, std::vector<void*>& owner, T** p, size_t n, int fill = 0) {
    void* q = nullptr;
    MF_HIP(c, hipMalloc(&q, n * sizeof(T)));
    MF_HIP(c, hipMemsetAsync(q, fill, n * sizeof(T), c->stream));
    owner.push_back(q);
    *p = reinterpret_cast<T*>(q);
    return MF_OK;
}
template <typename T>
static int host_alloc(mf_ctx* c, T** p, size_t n) {
    void* q = nullptr;
    MF_HIP(c, hipHostMalloc(&q, n * sizeof(T)));
    memset(q, 0, n * sizeof(T));
    c->host_allocs.push_back(q);
    *p = reinterpret_cast<T*>(q);
    return MF_OK;
}

extern "C" int mf_default_config(mf_config* cfg, int32_t width, int32_t height, float fx, float fy, float cx, float cy) {
    if (!cfg) return MF_EINVAL;
    memset(cfg, 0, sizeof(*cfg));
    cfg->width = width; cfg->height = height; cfg->fx = fx; cfg->fy = fy; cfg->cx = cx; cfg->cy = cy;
    cfg->device = 0;
    cfg->time_delta = 200; cfg->conf_global = 4.f; cfg->conf_object = 2.f; cfg->depth_cutoff = 3.f;
    cfg->icp_weight = 10.f; cfg->fast_odom = 0; cfg->so3 = 1; cfg->pyramid = 1; cfg->max_depth_processed = 20.f;
    cfg->outlier_coefficient = 0.9f;
    cfg->num_gsurfels = 9437184; cfg->num_osurfels = 1048576;
    cfg->enable_multiple_models = 1;
    cfg->model_spawn_offset = 20; cfg->track_all_models = 1; cfg->max_models = 32;
    cfg->rgb_only = 0;
    cfg->pose_log_capacity = 65536;   // enablePoseLogging = true (Core/MaskFusion.h:402)
    return MF_OK;
}

static __global__ void k_set_weight_literal(PoseDev* p, int literal, PoseDev* host_mirror) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    p->weightLiteral = literal;
    if (host_mirror) host_mirror->weightLiteral = literal;
}
static __global__ void k_pose_identity(PoseDev* p, int weight_literal) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    PoseDev q;
    memset(&q, 0, sizeof(q));
    q.weightLiteral = weight_literal;
    for (int k = 0; k < 9; ++k) q.R[k] = q.Ri[k] = q.lastR[k] = q.initR[k] = (k % 4 == 0) ? 1.f : 0.f;
    q.fusionWeight = 1.f;
    q.alive = 1;
    *p = q;
}
static __global__ void k_frame_init(FrameDev* f, int tick) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    f->tick = tick; f->count = 0; f->countNext = 0; f->runs = 0; f->cover = 0; f->useFillIn = 0;
    f->pad[0] = f->pad[1] = f->pad[2] = 0;
    f->done_cover = 0ull;
    MF_FRAME_BBOX_RESET(f);
}

static int surfel_capacity(int num) {  // Model::TEXTURE_DIMENSION_*^2 (Core/Model/Model.cpp:101-105)
    const int dim = 64 * (int)(sqrt((double)num) / 64);
    return dim * dim;
}

// Model::Model (Core/Model/Model.cpp:115-223): buffers of one model
// An object model's private scratch of the batched surfel passes: its own index maps / key image / candidate records, 141 B per pixel + 9 B per
// surfel slot (43 MB at VGA, 173 MB at 1280x960).  Only the batched passes ("batchObjectPasses", >= 2 objects) touch it, so it is allocated when
// a model first takes part in one -- or ahead of time for preallocated models, whose point is that nothing is allocated at spawn time
// (every object model used to carry it from its creation on, batched or not).
static int ensure_obj_scratch(mf_ctx* c, ModelState& m) {
    if (m.scr.keys) return MF_OK;
    const size_t P = (size_t)c->P, cap = (size_t)m.cap;
    int rc;
#define A(call) do { rc = (call); if (rc != MF_OK) return rc; } while (0)
    A(dev_alloc(c, m.allocs, &m.scr.keys, P, 0xFF));
    A(dev_alloc(c, m.allocs, &m.scr.index, P));
    A(dev_alloc(c, m.allocs, &m.scr.ivc, P));
    A(dev_alloc(c, m.allocs, &m.scr.inr, P));
    A(dev_alloc(c, m.allocs, &m.scr.iclean, P * 2));
    A(dev_alloc(c, m.allocs, &m.scr.cand_op, P));
    A(dev_alloc(c, m.allocs, &m.scr.cand_rec, P * 3));
    A(dev_alloc(c, m.allocs, &m.scr.upd_first, cap));
    A(dev_alloc(c, m.allocs, &m.scr.cand_best, P));
    A(dev_alloc(c, m.allocs, &m.scr.scan_state, clean_scan_entries((long)cap + (long)P)));
    A(dev_alloc(c, m.allocs, &m.scr.clean_ctl, (size_t)kCleanCtlInts));
#undef A
    launch_fill_int(m.scr.upd_first, kNoUpdate, (int)cap, c->stream);
    return MF_OK;
}

static int create_model(mf_ctx* c, int id, float confThr, bool allowFillIn, int cap, std::unique_ptr<ModelState>& out, bool with_scratch = false) {
    std::unique_ptr<ModelState> m(new ModelState());
    m->id = id; m->confThr = confThr; m->allowFillIn = allowFillIn; m->cap = cap;
    const size_t P = (size_t)c->P;
    int rc;
#define A(call) do { rc = (call); if (rc != MF_OK) return rc; } while (0)
    for (int b = 0; b < 2; ++b) {
        A(dev_alloc(c, m->allocs, &m->surf[b].pc, (size_t)cap));
        A(dev_alloc(c, m->allocs, &m->surf[b].ct, (size_t)cap));
        A(dev_alloc(c, m->allocs, &m->surf[b].nr, (size_t)cap));
        A(dev_alloc(c, m->allocs, &m->surf[b].box, run_table_entries((long)cap + (long)c->P)));
        m->surf[b].cap = cap;
    }
    A(dev_alloc(c, m->allocs, &m->d_pose, 1));
    A(dev_alloc(c, m->allocs, &m->d_frame, 1));
    A(dev_alloc(c, m->allocs, &m->d_predV, P));
    A(dev_alloc(c, m->allocs, &m->d_predN, P));
    A(dev_alloc(c, m->allocs, &m->d_predImage, P));
    A(dev_alloc(c, m->allocs, &m->d_predTime, P));
    A(dev_alloc(c, m->allocs, &m->d_predGray, P));
    if (allowFillIn) A(dev_alloc(c, m->allocs, &m->d_fillGray, P));
    if (c->cfg.pose_log_capacity > 0) A(dev_alloc(c, m->allocs, &m->d_poselog, (size_t)c->cfg.pose_log_capacity * 8));
    for (int i = 0; i < 3; ++i) {
        const size_t lp = (size_t)(c->W >> i) * (c->H >> i);
        A(dev_alloc(c, m->allocs, &m->d_vmap_g[i], lp * 3));
        A(dev_alloc(c, m->allocs, &m->d_nmap_g[i], lp * 3));
    }
    {
        int nbm = icp_batch_max_blocks(c->W, c->H);
        for (int i = 0; i < 3; ++i) nbm = std::max(nbm, std::max(icp_grid_blocks(c->W >> i, c->H >> i), icp_geo_grid_blocks(c->W >> i, c->H >> i)));
        const size_t nbmax = (size_t)nbm;
        for (int b = 0; b < 2; ++b) A(dev_alloc(c, m->allocs, &m->d_partials[b], nbmax * kIcpSlots));
    }
    A(dev_alloc(c, m->allocs, &m->d_gn, 2));
    A(dev_alloc(c, m->allocs, &m->d_track, 1));
    A(dev_alloc(c, m->allocs, &m->d_icp_log, (size_t)20 * 32));
    A(dev_alloc(c, m->allocs, &m->d_gn_trace, (size_t)20 * kGnTraceRow));
    if (!allowFillIn && with_scratch) A(ensure_obj_scratch(c, *m));
#undef A
    {
        TrackModelDev t;
        memset(&t, 0, sizeof(t));
        t.predV = m->d_predV; t.predN = m->d_predN; t.pose = m->d_pose; t.frame = m->d_frame;
        for (int i = 0; i < 3; ++i) { t.vm[i] = m->d_vmap_g[i]; t.nm[i] = m->d_nmap_g[i]; }
        t.partials[0] = m->d_partials[0]; t.partials[1] = m->d_partials[1]; t.st = m->d_gn;
        t.log = m->d_icp_log;                                      // 128 B per iteration and model
        t.trace = m->d_gn_trace;
        t.jump_limit = allowFillIn ? 0.f : 0.2f;                   // MaskFusion.cpp:268-272 applies to object models
        t.allow_fill = allowFillIn ? 1 : 0;
        m->track_host = t;
    }
    hipLaunchKernelGGL(k_pose_identity, dim3(1), dim3(64), 0, c->stream, m->d_pose, c->weight_literal ? 1 : 0);
    hipLaunchKernelGGL(k_frame_init, dim3(1), dim3(64), 0, c->stream, m->d_frame, c->host_tick);
    if (hipHostMalloc((void**)&m->h_pose, sizeof(PoseDev)) != hipSuccess || hipHostMalloc((void**)&m->h_frame, sizeof(FrameDev)) != hipSuccess ||
        hipHostMalloc((void**)&m->h_count, sizeof(int)) != hipSuccess)
        return MF_ENOMEM;
    memset(m->h_pose, 0, sizeof(PoseDev));
    for (int k = 0; k < 9; ++k) m->h_pose->R[k] = m->h_pose->Ri[k] = (k % 4 == 0) ? 1.f : 0.f;
    m->h_pose->alive = 1;
    m->h_pose->fusionWeight = 1.f;                         // pose == lastPose: computeFusionWeight(1) = 1 before the first tracking step
    m->h_pose->weightLiteral = c->weight_literal ? 1 : 0;
    memset(m->h_frame, 0, sizeof(FrameDev));
    m->h_frame->tick = c->host_tick;
    *m->h_count = 0;
    m->track_host.pose_host = m->h_pose;
    MF_HIP(c, hipMemcpyAsync(m->d_track, &m->track_host, sizeof(TrackModelDev), hipMemcpyHostToDevice, c->stream));
    MF_HIP(c, hipStreamSynchronize(c->stream));   // track_host is pageable: the copy must not outlive a moved ModelState
    out = std::move(m);
    return MF_OK;
}

extern "C" int mf_create(const mf_config* cfg, mf_ctx** out) {
    if (!cfg || !out) return MF_EINVAL;
    *out = nullptr;
    if (cfg->width <= 0 || cfg->height <= 0 || (cfg->width % 8) || (cfg->height % 8)) return MF_EINVAL;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || cfg->device >= ndev) return MF_ENODEV;
    mf_ctx* c = new mf_ctx();
    c->cfg = *cfg;
    if (c->cfg.model_spawn_offset <= 0) c->cfg.model_spawn_offset = 20;
    if (c->cfg.max_models <= 0) c->cfg.max_models = 32;
    c->W = cfg->width; c->H = cfg->height; c->P = c->W * c->H;
    c->K = Intr{cfg->fx, cfg->fy, cfg->cx, cfg->cy};
    auto fail = [&](int code) { mf_destroy(c); return code; };
    if (hipSetDevice(cfg->device) != hipSuccess) return fail(MF_ENODEV);
    {
        int cus = 1;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, cfg->device) != hipSuccess) { (void)hipGetLastError(); cus = 1; }
        c->ticket_lanes = cus < 1 ? 1 : (cus > kCleanTicketLanes ? kCleanTicketLanes : cus);
    }
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) return fail(MF_ENODEV);
    if (hipStreamCreateWithFlags(&c->stream_pre, hipStreamNonBlocking) != hipSuccess) return fail(MF_ENODEV);
    for (int i = 0; i < 2; ++i)
        if (hipEventCreateWithFlags(&c->ev_pre_done[i], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&c->ev_main_done[i], hipEventDisableTiming) != hipSuccess)
            return fail(MF_ENODEV);
    const int W = c->W, H = c->H, P = c->P;
    const int cap_bg = surfel_capacity(cfg->num_gsurfels), cap_obj = surfel_capacity(cfg->num_osurfels);
    if (cap_bg <= 0 || cap_obj <= 0) return fail(MF_EINVAL);
    c->cap_max = cap_bg > cap_obj ? cap_bg : cap_obj;
    int rc = MF_OK;
#define A(call) do { rc = (call); if (rc != MF_OK) return fail(rc); } while (0)
    // one block per input slot, depth | rgb | mask: a host frame arrives with one copy (mf_process_frame)
    auto up256 = [](size_t v) { return (v + 255) & ~(size_t)255; };
    c->in_off_rgb = up256((size_t)P * sizeof(float));
    c->in_off_mask = up256(c->in_off_rgb + (size_t)P * 3);
    c->in_bytes = up256(c->in_off_mask + (size_t)P);
    for (int i = 0; i < 2; ++i) {
        A(dev_alloc(c, c->allocs, &c->d_in_block[i], c->in_bytes));
        c->d_in_depth[i] = reinterpret_cast<float*>(c->d_in_block[i]);
        c->d_in_rgb[i] = c->d_in_block[i] + c->in_off_rgb;
        c->d_in_mask[i] = c->d_in_block[i] + c->in_off_mask;
    }
    c->d_rgb = c->d_in_rgb[0]; c->d_depth = c->d_in_depth[0]; c->d_mask_in = c->d_in_mask[0];
    for (int i = 0; i < 2; ++i) {
        A(host_alloc(c, &c->h_in[i], c->in_bytes));
        if (hipEventCreateWithFlags(&c->ev_in_copied[i], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&c->ev_in_consumed[i], hipEventDisableTiming) != hipSuccess) return fail(MF_EHIP);
    }
    if (hipStreamCreateWithFlags(&c->stream_in, hipStreamNonBlocking) != hipSuccess) return fail(MF_EHIP);
    A(dev_alloc(c, c->allocs, &c->d_zero_mask, (size_t)P));
    A(dev_alloc(c, c->allocs, &c->d_mask_tex, (size_t)P));
    for (int b = 0; b < 3; ++b) A(dev_alloc(c, c->allocs, &c->d_depthF[b], (size_t)P));
    for (int i = 0; i < 3; ++i) {
        const size_t lp = (size_t)(W >> i) * (H >> i);
        for (int set = 0; set < 2; ++set) {
            A(dev_alloc(c, c->allocs, &c->d_vmap[set][i], lp * 3));
            A(dev_alloc(c, c->allocs, &c->d_nmap[set][i], lp * 3));
        }
    }
    for (int i = 0; i < 3; ++i) {
        const size_t lp = (size_t)(W >> i) * (H >> i);
        for (int set = 0; set < 2; ++set) A(dev_alloc(c, c->allocs, &c->d_gray[set][i], lp));
        A(dev_alloc(c, c->allocs, &c->d_dIdx[i], lp));
        A(dev_alloc(c, c->allocs, &c->d_dIdy[i], lp));
        A(dev_alloc(c, c->allocs, &c->d_rgb_gate[i], lp));
        A(dev_alloc(c, c->allocs, &c->d_lastDepth[i], lp));
        A(dev_alloc(c, c->allocs, &c->d_lastImage[i], lp));
    }
    A(dev_alloc(c, c->allocs, &c->d_corres, (size_t)P));
    for (int b = 0; b < 2; ++b) {
        A(dev_alloc(c, c->allocs, &c->d_rgb_partials[b], (size_t)icp_grid_blocks(W, H) * kIcpSlots));
        A(dev_alloc(c, c->allocs, &c->d_cnt[b], (size_t)icp_grid_blocks(W, H)));
    }
    A(dev_alloc(c, c->allocs, &c->d_so3, 1));
    A(dev_alloc(c, c->allocs, &c->d_so3_scratch, so3_scratch_bytes(W >> 2, H >> 2)));
    {
        const size_t nt = splat_tiles_scratch_ints(W, H);
        const size_t maxcap = (size_t)std::max(surfel_capacity(cfg->num_gsurfels), surfel_capacity(cfg->num_osurfels));
        A(dev_alloc(c, c->allocs, &c->d_tile_count, nt));
        // every tile owns entries_cap / tiles list slots: 4x the surfel capacity in total (a surfel overlaps 1-4 tiles), i.e.
        // room for every surfel of a full map to land in a quarter of the image (an overflow is an error, never a drop)
        // ... and never less than 16 list slots per pixel of a tile, so that small maps can still pile up in one place
        c->tile_entries_cap = (int)std::min<size_t>(std::max<size_t>(4 * maxcap, (size_t)16 * P), (size_t)1 << 30);
        c->tile_entries_alloc = c->tile_entries_cap;
        // (round 3 tried 48-byte list entries carrying the sprite set-up -- one coalesced stream per tile instead of list -> box -> record
        // gathers: the tile pass gained 2 us, the binning pass lost 3.5 us and the lists grew to 1.8 GB; measured, reverted)
        A(dev_alloc(c, c->allocs, &c->d_tile_entries, (size_t)c->tile_entries_cap));
        A(dev_alloc(c, c->allocs, &c->d_splat_rec0, maxcap));
        A(dev_alloc(c, c->allocs, &c->d_splat_rec1, maxcap));
        A(dev_alloc(c, c->allocs, &c->d_splat_bbox, maxcap));
    }
    A(dev_alloc(c, c->allocs, &c->d_keys, (size_t)P, 0xFF));
    A(dev_alloc(c, c->allocs, &c->d_index, (size_t)P));
    A(dev_alloc(c, c->allocs, &c->d_ivc, (size_t)P));
    A(dev_alloc(c, c->allocs, &c->d_ict, (size_t)P));
    A(dev_alloc(c, c->allocs, &c->d_iclean, (size_t)P * 2));
    A(dev_alloc(c, c->allocs, &c->d_inr, (size_t)P));
    A(dev_alloc(c, c->allocs, &c->d_cand_op, (size_t)P));
    A(dev_alloc(c, c->allocs, &c->d_cand_rec, (size_t)P * 3));
    A(dev_alloc(c, c->allocs, &c->d_upd_first, (size_t)c->cap_max));
    A(dev_alloc(c, c->allocs, &c->d_flags, (size_t)c->cap_max + P));
    A(dev_alloc(c, c->allocs, &c->d_newconf, (size_t)c->cap_max + P));
    A(dev_alloc(c, c->allocs, &c->d_block_counts, (size_t)kCompactBlocks));
    A(dev_alloc(c, c->allocs, &c->d_cand_best, (size_t)P));
    A(dev_alloc(c, c->allocs, &c->d_scan_state, clean_scan_entries((long)c->cap_max + (long)P)));
    A(dev_alloc(c, c->allocs, &c->d_clean_ctl, (size_t)kCleanCtlInts));
    c->vis_max_runs = (int)(run_table_entries((long)c->cap_max + (long)P) / 2);
    A(dev_alloc(c, c->allocs, &c->d_vis_list, (size_t)c->vis_max_runs));
    A(dev_alloc(c, c->allocs, &c->d_vis_count, 1));
    A(dev_alloc(c, c->allocs, &c->d_cull_ctl, 2));
    A(dev_alloc(c, c->allocs, &c->d_icp_prof, (size_t)20 * 16));
    A(dev_alloc(c, c->allocs, &c->d_edge, (size_t)P));
    A(dev_alloc(c, c->allocs, &c->d_bin, (size_t)P));
    A(dev_alloc(c, c->allocs, &c->d_tmp_u8, (size_t)P));
    A(dev_alloc(c, c->allocs, &c->d_proj_ids, (size_t)P));
    A(host_alloc(c, &c->h_bin, (size_t)P));
    A(host_alloc(c, &c->h_ids, (size_t)P));
    A(host_alloc(c, &c->h_depth, (size_t)P));
    A(host_alloc(c, &c->h_mask, (size_t)P));
    A(host_alloc(c, &c->h_full, (size_t)P));
    launch_fill_int(c->d_upd_first, kNoUpdate, c->cap_max, c->stream);
    // globalModel = Model(getNextModelID(true), initConfidenceGlobal, fillIn = true) (Core/MaskFusion.cpp:80)
    std::unique_ptr<ModelState> bg;
    A(create_model(c, c->nextID++, cfg->conf_global, true, cap_bg, bg));
    c->models.push_back(std::move(bg));
#undef A
    c->ignoreMap.assign(P, 0);
    c->labels.reset(new LabelsScratch());
    if (!c->labels->init(P)) return fail(MF_ENOMEM);
    for (int i = 0; i <= MF_N_TIMINGS; ++i)
        if (hipEventCreate(&c->ev[i]) != hipSuccess) return fail(MF_EHIP);
    for (int i = 0; i < 2; ++i)
        if (hipEventCreate(&c->ev_icp[i]) != hipSuccess) return fail(MF_EHIP);
    for (int i = 0; i < 5; ++i)
        if (hipEventCreate(&c->ev_mm[i]) != hipSuccess) return fail(MF_EHIP);
    if (hipEventCreate(&c->ev_icp_mid) != hipSuccess) return fail(MF_EHIP);
    for (int i = 0; i < mf_ctx::kObjArgSlots; ++i) {
        if (hipMalloc((void**)&c->d_obj_args[i], sizeof(ObjPassArgs) * 64) != hipSuccess ||
            hipHostMalloc((void**)&c->h_obj_args[i], sizeof(ObjPassArgs) * 64) != hipSuccess ||
            hipEventCreateWithFlags(&c->ev_obj_args[i], hipEventDisableTiming) != hipSuccess)
            return fail(MF_ENOMEM);
    }
    if (hipStreamSynchronize(c->stream) != hipSuccess) return fail(MF_EHIP);
    *out = c;
    return MF_OK;
}

extern "C" void mf_destroy(mf_ctx* c) {
    if (!c) return;
    if (c->stream_pre) (void)hipStreamSynchronize(c->stream_pre);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    c->models.clear();
    for (void* p : c->allocs) (void)hipFree(p);
    if (c->d_splat_prof) (void)hipFree(c->d_splat_prof);
    for (void* p : c->host_allocs) (void)hipHostFree(p);
    for (int i = 0; i <= MF_N_TIMINGS; ++i)
        if (c->ev[i]) (void)hipEventDestroy(c->ev[i]);
    for (int i = 0; i < 2; ++i)
        if (c->ev_icp[i]) (void)hipEventDestroy(c->ev_icp[i]);
    if (c->ev_icp_mid) (void)hipEventDestroy(c->ev_icp_mid);
    for (int i = 0; i < 5; ++i)
        if (c->ev_mm[i]) (void)hipEventDestroy(c->ev_mm[i]);
    for (int i = 0; i < 2; ++i) {
        if (c->ev_pre_done[i]) (void)hipEventDestroy(c->ev_pre_done[i]);
        if (c->ev_main_done[i]) (void)hipEventDestroy(c->ev_main_done[i]);
    }
    if (c->ev_labels) (void)hipEventDestroy(c->ev_labels);
    if (c->ev_staged) (void)hipEventDestroy(c->ev_staged);
    for (int i = 0; i < 2; ++i) {
        if (c->ev_in_copied[i]) (void)hipEventDestroy(c->ev_in_copied[i]);
        if (c->ev_in_consumed[i]) (void)hipEventDestroy(c->ev_in_consumed[i]);
    }
    if (c->stream_in) { (void)hipStreamSynchronize(c->stream_in); (void)hipStreamDestroy(c->stream_in); }
    for (int i = 0; i < mf_ctx::kObjArgSlots; ++i) {
        if (c->d_obj_args[i]) (void)hipFree(c->d_obj_args[i]);
        if (c->h_obj_args[i]) (void)hipHostFree(c->h_obj_args[i]);
        if (c->ev_obj_args[i]) (void)hipEventDestroy(c->ev_obj_args[i]);
    }
    if (c->stream_pre) (void)hipStreamDestroy(c->stream_pre);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

extern "C" const char* mf_last_error(const mf_ctx* c) { return c ? c->err.c_str() : "null context"; }

static void mark(mf_ctx* c, int i, hipStream_t s = nullptr) {
    if (c->timings_on) (void)hipEventRecord(c->ev[i], s ? s : c->stream);
}

// ------------------------------------------------------------------------------------------------
// per-model stages
// ------------------------------------------------------------------------------------------------
// minimumGradientMagnitudes[level]^2 / sobelScale^2 (RGBDOdometry.cpp:31-32,102-105,381)
static float rgb_min_scale(int level) {
    const double minGrad[3] = {5.0, 3.0, 1.0}, sobelScale = 1.0 / 8.0;
    return (float)(pow(minGrad[level], 2.0) / pow(sobelScale, 2.0));
}

// rgb = rgbOnly || icpWeight < 100 (RGBDOdometry.cpp:238)
static bool photometric_on(const mf_ctx* c) { return c->cfg.rgb_only != 0 || c->cfg.icp_weight < 100.f; }

// Model::performTracking (Core/Model/Model.cpp:427-447): initICP (model pyramid + fill-in, RGB pyramids), the optional
// SO(3) pre-alignment, then the Gauss-Newton loop (ICP only: one launch per iteration; with the photometric term: two).
static void enqueue_track(mf_ctx* c, ModelState& m, const float* fillDepth, float jump_limit, long frame_k) {
    const mf_config& g = c->cfg;
    const int set = (int)(frame_k & 1);
    float* const* cur_vmap = c->d_vmap[set];
    float* const* cur_nmap = c->d_nmap[set];
    const int W = c->W, H = c->H;
    hipStream_t s = c->stream;
    launch_model_pyramid(m.d_predV, m.d_predN, m.allowFillIn ? fillDepth : nullptr, m.d_frame, m.d_pose, nullptr, m.d_vmap_g,
                         m.d_nmap_g, W, H, c->K, s);
    const bool rgb = photometric_on(c);
    const bool icp = !g.rgb_only && g.icp_weight > 0.f;
    // the previous frame's intensity pyramid is RGBDOdometry::lastNextImage (identical for every tracked model)
    const bool so3 = g.so3 != 0 && c->gray_frame[set ^ 1] == frame_k - 1 && c->gray_frame[set] == frame_k;
    if (so3)
        (void)launch_so3_prealign(c->d_gray[set ^ 1][2], c->d_gray[set][2], W >> 2, H >> 2, Intr{g.fx / 4, g.fy / 4, g.cx / 4, g.cy / 4},
                                  c->d_so3, c->d_so3_scratch, s);
    const So3Result* so3_seed = so3 ? c->d_so3 : nullptr;
    if (rgb) {
        // initRGBModel + initRGB (Model.cpp:395-406; Q1: both depth pyramids come from the vertex map initICPModel was given)
        launch_rgbd_last_l0(m.d_predV, m.allowFillIn ? fillDepth : nullptr, m.d_predGray, m.d_fillGray, m.d_frame, c->d_lastDepth[0],
                            c->d_lastImage[0], W * H, s, (c->ftf_rgb && m.allowFillIn) ? 1 : 0);
        for (int i = 0; i + 1 < 3; ++i) {   // one level of the depth pyramid and of the intensity pyramid per launch (independent of each other)
            SmallJobs jobs;
            jobs.n = 2;
            jobs.j[0] = SmallJob{1, c->d_lastDepth[i], c->d_lastDepth[i + 1], nullptr, nullptr, W >> i, H >> i, 0.f};
            jobs.j[1] = SmallJob{0, c->d_lastImage[i], c->d_lastImage[i + 1], nullptr, nullptr, W >> i, H >> i, 0.f};
            launch_small_jobs(jobs, s);
        }
    }
    const int iters[3] = {g.fast_odom ? 3 : 10, g.pyramid ? 5 : 0, g.pyramid ? 4 : 0};  // RGBDOdometry.cpp:327-329
    const float sobelScale = 1.0f / 8.0f;                                                // 1 / 2^sobelSize, :31-32
    const bool timed = c->timings_on && &m == c->models[0].get();
    if (timed) { (void)hipEventRecord(c->ev_icp[0], s); c->tracked_once = true; c->icp_mid_recorded = false; }
    int k = 0, nb_prev = 0, prev_level = -1;
    // every launch of the loop and its finalize
    auto issue_loop = [&](bool with_marks) {
    for (int lvl = 2; lvl >= 0; --lvl) {
        const float div = (float)(1 << lvl);
        if (lvl == 0 && with_marks && timed) { (void)hipEventRecord(c->ev_icp_mid, s); c->icp_mid_recorded = true; }   // coarse levels | level 0 (bench.py: roofline.levels)
        for (int j = 0; j < iters[lvl]; ++j) {
            IcpLaunch l;
            l.vmap_curr = cur_vmap[lvl]; l.nmap_curr = cur_nmap[lvl];
            l.vmap_prev = m.d_vmap_g[lvl]; l.nmap_prev = m.d_nmap_g[lvl];
            l.W = W >> lvl; l.H = H >> lvl; l.k = Intr{g.fx / div, g.fy / div, g.cx / div, g.cy / div};
            l.distThres = 0.10f; l.angleThres = sinf(20.f * 3.14159254f / 180.f);  // RGBDOdometry.h:35-36
            l.partials_in = nb_prev ? m.d_partials[(k + 1) & 1] : nullptr;
            l.nblocks_in = nb_prev;
            l.partials_out = m.d_partials[k & 1];
            l.state_in = &m.d_gn[k & 1]; l.state_out = &m.d_gn[(k + 1) & 1];
            l.log_out = (k > 0) ? m.d_icp_log + 32 * (k - 1) : nullptr;
            l.trace = rgb ? nullptr : m.d_gn_trace; l.it = k;
            l.prof_out = (c->icp_prof_on && m.id == 0 && !rgb) ? c->d_icp_prof + 16 * k : nullptr;
            l.pose_in = (k == 0) ? m.d_pose : nullptr;
            l.so3_in = (k == 0) ? so3_seed : nullptr;
            if (!rgb) {
                launch_icp_iteration(l, s);
            } else {
                RgbdLaunch r;
                r.icp = l;
                r.L.dIdx = c->d_dIdx[lvl]; r.L.dIdy = c->d_dIdy[lvl];
                r.L.lastDepth = c->d_lastDepth[lvl]; r.L.nextDepth = c->d_lastDepth[lvl];
                r.L.lastImage = c->d_lastImage[lvl]; r.L.nextImage = c->d_gray[set][lvl];
                r.L.W = l.W; r.L.H = l.H;
                r.L.minScale = rgb_min_scale(lvl);
                r.L.gate = c->d_rgb_gate[lvl];
                r.L.maxDepthDelta = 0.07f;                                          // maxDepthDeltaRGB, :33
                r.corres = c->d_corres;
                r.rgb_partials_in = nb_prev ? c->d_rgb_partials[(k + 1) & 1] : nullptr;
                r.rgb_partials_out = c->d_rgb_partials[k & 1];
                r.cnt_in = nb_prev ? c->d_cnt[(k + 1) & 1] : nullptr;
                r.cnt_out = c->d_cnt[k & 1];
                r.icpWeight = g.icp_weight; r.icpOn = icp ? 1 : 0; r.rgbOnly = g.rgb_only ? 1 : 0; r.sobelScale = sobelScale;
                r.level = lvl; r.prev_level = (prev_level < 0) ? lvl : prev_level;
                r.so3_in = l.so3_in;
                launch_rgbd_iteration(r, s);
            }
            nb_prev = rgb ? icp_grid_blocks(l.W, l.H) : icp_geo_grid_blocks(l.W, l.H);
            prev_level = lvl;
            ++k;
        }
    }
    if (with_marks && timed) (void)hipEventRecord(c->ev_icp[1], s);
    float* log_out = (k > 0) ? m.d_icp_log + 32 * (k - 1) : nullptr;
    if (!rgb)
        launch_icp_finalize(nb_prev ? m.d_partials[(k + 1) & 1] : nullptr, nb_prev, &m.d_gn[k & 1], m.d_pose, m.h_pose, log_out,
                            jump_limit, so3_seed, s, m.d_gn_trace, k);
    else
        launch_rgbd_finalize(nb_prev ? m.d_partials[(k + 1) & 1] : nullptr, nb_prev ? c->d_rgb_partials[(k + 1) & 1] : nullptr,
                             nb_prev ? c->d_cnt[(k + 1) & 1] : nullptr, nb_prev, g.icp_weight, icp ? 1 : 0, g.rgb_only ? 1 : 0, 1,
                             prev_level, &m.d_gn[k & 1], so3_seed, m.d_pose, m.h_pose, log_out, jump_limit, s);
    };
    issue_loop(true);
}

// The same for several models at once (geometric term only; MaskFusion.cpp:247-276 tracks the models one after the other, their
// steps are independent): the model pyramid, every Gauss-Newton iteration and the final pose update of ALL of them are one
// launch each, so that a frame with M tracked models costs ~40 launches instead of M x 21 latency-bound ones.
static void enqueue_track_batch(mf_ctx* c, const std::vector<ModelState*>& ms, const float* fillDepth, long frame_k) {
    const mf_config& g = c->cfg;
    const int set = (int)(frame_k & 1);
    const int W = c->W, H = c->H;
    hipStream_t s = c->stream;
    TrackBatch b;
    memset(&b, 0, sizeof(b));
    b.n = (int)ms.size();
    for (int i = 0; i < b.n; ++i) b.m[i] = ms[i]->d_track;
    launch_model_pyramid_batch(b, fillDepth, W, H, c->K, s);
    const bool so3 = g.so3 != 0 && c->gray_frame[set ^ 1] == frame_k - 1 && c->gray_frame[set] == frame_k;
    if (so3)   // one pre-alignment serves every model: it only looks at the two frames (RGBDOdometry.cpp:264-324)
        (void)launch_so3_prealign(c->d_gray[set ^ 1][2], c->d_gray[set][2], W >> 2, H >> 2, Intr{g.fx / 4, g.fy / 4, g.cx / 4, g.cy / 4},
                                  c->d_so3, c->d_so3_scratch, s);
    const So3Result* so3_seed = so3 ? c->d_so3 : nullptr;
    const int iters[3] = {g.fast_odom ? 3 : 10, g.pyramid ? 5 : 0, g.pyramid ? 4 : 0};
    const bool timed = c->timings_on && ms[0] == c->models[0].get();
    if (timed) { (void)hipEventRecord(c->ev_icp[0], s); c->tracked_once = true; c->icp_mid_recorded = false; }
    int it = 0, nb_prev = 0;
    for (int lvl = 2; lvl >= 0; --lvl) {
        const float div = (float)(1 << lvl);
        for (int j = 0; j < iters[lvl]; ++j) {
            launch_icp_batch_solve(b, it, nb_prev, it == 0 ? so3_seed : nullptr, s);
            launch_icp_batch_pixels(b, it, lvl, c->d_vmap[set][lvl], c->d_nmap[set][lvl], W >> lvl, H >> lvl,
                                    Intr{g.fx / div, g.fy / div, g.cx / div, g.cy / div}, 0.10f, sinf(20.f * 3.14159254f / 180.f), s);
            nb_prev = icp_batch_blocks(W >> lvl, H >> lvl, b.n);
            ++it;
        }
    }
    if (timed) (void)hipEventRecord(c->ev_icp[1], s);
    if (it == 0) launch_icp_batch_solve(b, 0, 0, so3_seed, s);   // no iterations at all: the states still have to exist
    launch_icp_batch_finalize(b, it, nb_prev, so3_seed, s);
}

// workgroups for the grid-stride surfel kernels of model m: all of them for the background, and for an object model -- with
// "objectSmallGrids" -- twice what its last known surfel count needs (the count lives on the device; *h_count is its pinned mirror as of the
// last clean pass: a stale value only costs a few more trips round the grid-stride loop, never a result)
static int surfel_blocks(const mf_ctx* c, const ModelState& m) {
    if (!c->object_small_grids || m.id == 0) return kSurfelGridBlocks;
    const long n = 2L * (long)*m.h_count + 4096;
    const long b = (n + 255) / 256;
    return (int)(b < 32 ? 32 : (b > kSurfelGridBlocks ? kSurfelGridBlocks : b));
}

// predictIndices -> fuse -> [predictIndices] -> clean for one model (Core/MaskFusion.cpp:541-563 / :344-353)
// The runs of m's live buffer that can be in view under m's current pose (k_cull), for the projection passes of this frame: culled once per
// buffer and pose -- GlobalProjection and the two index-map passes of a frame share one list (the in-place update moves no surfel of a run
// that is not listed: only surfels the first index map drew are merged), the prediction after clean() gets its own (new buffer).
// Depth range: the widest any consumer uses.  nullptr: culling is off.
static const VisList* ensure_vis(mf_ctx* c, ModelState& m, VisList& out) {
    // (a small map is cheaper to stream than to cull: the test is a launch of its own on a chain of launches that are each a few microseconds.
    // The count is the pinned mirror as of the model's last clean pass; which side of the threshold a frame falls on changes no result)
    if (!c->cull_runs || *m.h_count < c->cull_min_surfels) return nullptr;
    out.list = c->d_vis_list; out.count = c->d_vis_count;
    if (c->vis_tag.model == &m && c->vis_tag.frame == c->frame_no && c->vis_tag.cur == m.cur) return &out;
    const mf_config& g = c->cfg;
    launch_cull(m.surf[m.cur], m.d_frame, m.d_pose, c->W, c->H, c->K, fmaxf(g.depth_cutoff, g.max_depth_processed), g.time_delta, c->d_vis_list,
                c->d_vis_count, c->d_cull_ctl, (int)(run_table_entries((long)m.cap + (long)c->P) / 2), c->stream);
    c->vis_tag.model = &m; c->vis_tag.frame = c->frame_no; c->vis_tag.cur = m.cur;
    return &out;
}

// workgroups of a clean launch for model m: sized from its last known count (pinned mirror; the chunks are drawn from a ticket counter,
// so a stale value costs a workgroup a few more rounds, never a result)
static int clean_blocks(const mf_ctx* c, const ModelState& m) {
    return clean_grid((long)*m.h_count + (long)c->P / 2);
}
static unsigned next_clean_epoch(mf_ctx* c) {
    c->clean_epoch = (c->clean_epoch + 1u) & 0x3FFFFFFFu;
    if (c->clean_epoch == 0) c->clean_epoch = 1;
    return c->clean_epoch;
}

static void enqueue_fuse_clean(mf_ctx* c, ModelState& m, const uint8_t* d_rgb, const float* d_depth, const float* depthF,
                               const uint8_t* mask, float fuseDepthCutoff, float weightMultiplier, bool secondIndexPass, bool marks) {
    const mf_config& g = c->cfg;
    const int W = c->W, H = c->H;
    hipStream_t s = c->stream;
    const int src = m.cur, dst = 1 - m.cur;
    const int blocks = surfel_blocks(c, m);
    VisList vl;
    const VisList* vis = ensure_vis(c, m, vl);
    launch_index_scatter(m.surf[src], m.d_frame, m.d_pose, W, H, c->K, g.max_depth_processed, g.time_delta, c->d_keys, false, s, blocks, vis);
    launch_index_resolve(m.surf[src], m.d_pose, c->d_keys, W, H, c->d_index, c->d_ivc, c->d_inr, secondIndexPass ? nullptr : c->d_ict,
                         nullptr, s);
    if (marks) mark(c, 4);
    // Model::fuse maxDepth uniform: min(depthCutoff, model.maxDepth, bb_max_z) (Model.cpp:527); bb_max_z from the model's bounding box on the device
    launch_fuse_data(d_rgb, d_depth, depthF, mask, m.id, m.d_frame, m.d_pose, weightMultiplier, fminf(fuseDepthCutoff, m.maxDepth), W, H,
                     c->K, c->d_index, c->d_ivc, c->d_inr, c->d_cand_op, c->d_cand_rec, c->d_upd_first, c->d_cand_best, s, c->bbox_limit ? 1 : 0);
    if (marks) mark(c, 5);
    // update.vert in place: only the surfels a candidate merged into are touched (the reference copies the whole buffer, Model.cpp:583-646)
    launch_fuse_update(m.surf[src], m.d_frame, c->d_upd_first, c->d_cand_op, c->d_cand_best, c->d_cand_rec, W, H, s);
    if (marks) mark(c, 6);
    if (secondIndexPass) {   // predictIndices on the updated buffer (:556), column-major packed texels for clean's window gathers
        launch_index_scatter(m.surf[src], m.d_frame, m.d_pose, W, H, c->K, g.max_depth_processed, g.time_delta, c->d_keys, true, s, blocks, vis);
        launch_index_resolve(m.surf[src], m.d_pose, c->d_keys, W, H, nullptr, nullptr, nullptr, nullptr, c->d_iclean, s);
    }
    launch_clean(m.surf[src], m.surf[dst], m.d_frame, m.d_pose, W, H, c->K, g.time_delta, m.confThr, g.outlier_coefficient, m.id,
                 c->d_index, c->d_ivc, c->d_ict, secondIndexPass ? c->d_iclean : nullptr, depthF, mask, c->d_cand_op, c->d_cand_rec, nullptr, nullptr,
                 c->d_scan_state, c->d_clean_ctl, next_clean_epoch(c), clean_blocks(c, m), c->ticket_lanes, m.h_count, secondIndexPass, c->clean_literal, s);
    m.cur = dst;   // one copying pass per frame (clean): the live buffer alternates
}

// MaskFusion::predict for one model: combinedPredict(maxDepthProcessed, tick, tick, timeDelta) -- the fill-in half
// (performFillIn) is evaluated lazily by the next tracking step from the retained filtered depth.
// advance: the end-of-frame bookkeeping of this model (processFrame's tail); the tiled prediction runs it as its epilogue, the scatter
// form is followed by k_frame_advance.
static void enqueue_predict(mf_ctx* c, ModelState& m, const FrameAdvance* advance = nullptr) {
    m.pred_gray_valid = photometric_on(c);
    if (c->splat_tiles && !(c->object_scatter_splat && m.id != 0)) {
        const bool gray = photometric_on(c);
        VisList vl;
        const VisList* vis = ensure_vis(c, m, vl);
        if (launch_splat_tiled(m.surf[m.cur], m.d_frame, m.d_pose, c->W, c->H, c->K, c->cfg.max_depth_processed, m.confThr,
                               c->cfg.time_delta, c->d_tile_count, c->d_tile_entries, c->tile_entries_cap, c->d_splat_rec0, c->d_splat_rec1,
                               c->d_splat_bbox, m.d_predV, m.d_predN, m.d_predImage, m.d_predTime, c->cur_rgb,
                               gray ? m.d_predGray : nullptr, gray ? m.d_fillGray : nullptr, c->stream, advance, c->ftf_rgb ? 1 : 0,
                               (c->splat_prof_on && m.id == 0) ? c->d_splat_prof : nullptr, c->splat_tune, vis) == 0)
            return;
    }
    launch_splat_scatter(m.surf[m.cur], m.d_frame, m.d_pose, c->W, c->H, c->K, c->cfg.max_depth_processed, m.confThr,
                         c->cfg.time_delta, c->d_keys, c->stream, surfel_blocks(c, m));
    const bool gray = photometric_on(c) ;
    launch_splat_resolve(m.surf[m.cur], m.d_pose, c->d_keys, c->W, c->H, c->K, m.d_predV, m.d_predN, m.d_predImage, m.d_predTime,
                         m.d_frame, c->cur_rgb, gray ? m.d_predGray : nullptr, gray ? m.d_fillGray : nullptr, c->stream, c->ftf_rgb ? 1 : 0);
    if (advance) launch_frame_advance(m.d_frame, c->W, c->H, advance->host_mirror, m.d_pose, advance->bg_pose, advance->log_slot, c->stream);
}

// ObjBatch of the object models in `ms` (mf_internal.h): one entry per model, staged through a pinned slot of the ring and copied to the
// device on the stream; the slot's event guards its reuse (the host is at most a frame ahead in a multi-model scene: it waits for the
// label stage every frame).  weightMultiplier / log slots are filled by the caller where they matter.
static int make_obj_batch(mf_ctx* c, const std::vector<ModelState*>& ms, const std::vector<int>& orders, const uint8_t* d_rgb, const float* d_depth,
                          const float* depthF, const uint8_t* mask, float weightMultiplier, const std::vector<float*>* log_slots, ObjBatch& b, int& blocks) {
    const mf_config& g = c->cfg;
    const int slot = (int)(c->obj_arg_slot++ % mf_ctx::kObjArgSlots);
    MF_HIP(c, hipEventSynchronize(c->ev_obj_args[slot]));
    ObjPassArgs* h = c->h_obj_args[slot];
    const bool gray = photometric_on(c);
    blocks = 32;
    for (size_t i = 0; i < ms.size(); ++i) {
        ModelState& m = *ms[i];
        ObjPassArgs& a = h[i];
        if (!m.scr.keys) { int rc = ensure_obj_scratch(c, m); if (rc != MF_OK) return rc; }   // first batched pass of a model created at spawn time
        a.a = m.surf[m.cur]; a.b = m.surf[1 - m.cur];
        a.frame = m.d_frame; a.pose = m.d_pose;
        a.maskID = m.id; a.confThreshold = m.confThr; a.fuseMaxDepth = fminf(g.depth_cutoff, m.maxDepth); a.weightMultiplier = weightMultiplier;
        a.keys = m.scr.keys; a.index = m.scr.index; a.ivc = m.scr.ivc; a.inr = m.scr.inr; a.iclean = m.scr.iclean;
        a.cand_op = m.scr.cand_op; a.cand_rec = m.scr.cand_rec; a.upd_first = m.scr.upd_first; a.cand_best = m.scr.cand_best;
        a.scan_state = m.scr.scan_state; a.clean_ctl = m.scr.clean_ctl; a.host_count = m.h_count;
        a.predV = m.d_predV; a.predN = m.d_predN; a.predImage = m.d_predImage; a.predTime = m.d_predTime; a.predGray = gray ? m.d_predGray : nullptr;
        a.host_frame = m.h_frame; a.log_slot = log_slots ? (*log_slots)[i] : nullptr;
        a.global_payload = ((unsigned)orders[i] << 8) | ((unsigned)m.id & 255u);
        blocks = std::max(blocks, surfel_blocks(c, m));
    }
    MF_HIP(c, hipMemcpyAsync(c->d_obj_args[slot], h, sizeof(ObjPassArgs) * ms.size(), hipMemcpyHostToDevice, c->stream));
    MF_HIP(c, hipEventRecord(c->ev_obj_args[slot], c->stream));
    b.m = c->d_obj_args[slot]; b.n = (int)ms.size();
    b.W = c->W; b.H = c->H; b.k = c->K; b.maxDepthProcessed = g.max_depth_processed; b.globalMaxDepth = g.depth_cutoff; b.timeDelta = g.time_delta;
    b.outlierCoeff = g.outlier_coefficient; b.cleanLiteral = c->clean_literal ? 1 : 0; b.bboxLimit = c->bbox_limit ? 1 : 0; b.cleanEpoch = 0; b.cleanTicketLanes = 1;
    b.rgb = d_rgb; b.depthRaw = d_depth; b.depthF = depthF; b.mask = mask; b.bg_pose = c->models[0]->d_pose; b.global_keys = c->d_keys;
    return MF_OK;
}
// every object model of the list, with its position in the list (the GlobalProjection payload)
static void object_models(mf_ctx* c, std::vector<ModelState*>& ms, std::vector<int>& orders) {
    ms.clear(); orders.clear();
    for (size_t i = 1; i < c->models.size(); ++i) { ms.push_back(c->models[i].get()); orders.push_back((int)i); }
}
static bool batch_objects_now(const mf_ctx* c) { return c->batch_objects && c->models.size() >= 3 && c->models.size() <= 65; }

static int check_launch(mf_ctx* c) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { c->err = std::string("launch failed: ") + hipGetErrorString(e); return MF_EHIP; }
    return MF_OK;
}

// getNextModelID(true) (Core/MaskFusion.cpp:715-731)
static int take_next_model_id(mf_ctx* c) {
    const int next = c->nextID;
    for (;;) {
        c->nextID = (c->nextID + 1) & 255;
        bool occupied = false;
        for (auto& m : c->models) occupied |= (m->id == c->nextID);
        if (!occupied) break;
    }
    return next;
}

// filterDepth (Core/MaskFusion.cpp:217) + Model::generateCUDATextures (Model.cpp:350-389) + the frame's intensity pyramid and
// derivative images, for frame index k (buffer set k & 1, filtered-depth ring slot k % 3).
// With overlapPreprocessing it runs on its own stream and starts when frame k-1 has finished TRACKING: frame k-2 (the last
// user of this buffer set and of depthF[k % 3]) is then complete, and the filter overlaps the atomic-/latency-bound fusion
// kernels of frame k-1 rather than its Gauss-Newton launches, which need a whole CU per workgroup and stall behind resident
// filter waves.
static int enqueue_preprocess(mf_ctx* c, const uint8_t* d_rgb, const float* d_depth, long k, bool with_maps) {
    const int W = c->W, H = c->H, P = c->P;
    hipStream_t s = c->stream;
    const mf_config& g = c->cfg;
    const int set = (int)(k & 1);
    float* depthF = c->d_depthF[k % 3];
    hipStream_t sp = c->overlap ? c->stream_pre : s;
    if (c->overlap) MF_HIP(c, hipStreamWaitEvent(sp, c->ev_main_done[set ^ 1], 0));
    mark(c, 0, sp);
    launch_bilateral(d_depth, depthF, W, H, sp);
    if (with_maps) {   // the frame that initialises the map is never tracked against: no vertex / normal maps needed
        launch_frame_pyramid(depthF, c->d_vmap[set], c->d_nmap[set], W, H, c->K, g.depth_cutoff, sp);
    }
    c->cur_rgb = d_rgb;
    c->cur_depth = d_depth;
    if (photometric_on(c) || g.so3) {
        // imageBGRToIntensity + pyrDownUcharGauss of the frame (initRGB / initFirstRGB) and, for the photometric term,
        // computeDerivativeImages (RGBDOdometry.cpp:245-250)
        launch_intensity(d_rgb, 3, c->d_gray[set][0], P, sp);
        for (int i = 0; i + 1 < 3; ++i) launch_pyrdown_u8(c->d_gray[set][i], c->d_gray[set][i + 1], W >> i, H >> i, sp);
        c->gray_frame[set] = k;
        if (photometric_on(c)) {
            SmallJobs jobs;   // the three levels' derivative / gate images: one launch
            jobs.n = 3;
            for (int i = 0; i < 3; ++i)
                jobs.j[i] = SmallJob{2, c->d_gray[set][i], c->d_dIdx[i], c->d_dIdy[i], c->d_rgb_gate[i], W >> i, H >> i, rgb_min_scale(i)};
            launch_small_jobs(jobs, sp);
            c->deriv_frame = k;
        }
    }
    mark(c, 1, sp);
    if (c->overlap) {
        MF_HIP(c, hipEventRecord(c->ev_pre_done[set], sp));
        MF_HIP(c, hipStreamWaitEvent(s, c->ev_pre_done[set], 0));
    }
    return MF_OK;
}

static int download_pose_log(mf_ctx* c, ModelState& m, std::vector<int64_t>& ts, std::vector<float>& p7);

// GlobalProjection::project for one model (fixed confidence threshold 12, GlobalProjection.cpp:43-107).  The background model goes
// through the tile lists (its ~10^5..10^6 sprites cover millions of pixels: one memory-side atomic each in the scatter form);
// object models (a few thousand sprites) keep the scatter form, which costs them one short launch.  Both write the same keys.
static void enqueue_global_projection(mf_ctx* c, ModelState& m, int order) {
    const mf_config& g = c->cfg;
    if (m.id == 0 && c->splat_tiles && c->global_tiles) {
        VisList vl;
        const VisList* vis = ensure_vis(c, m, vl);
        if (launch_global_tiled(m.surf[m.cur], m.d_frame, m.d_pose, c->W, c->H, c->K, g.depth_cutoff, 12.0f, g.time_delta, order, m.id, c->d_tile_count,
                                c->d_tile_entries, c->tile_entries_cap, c->d_splat_rec0, c->d_splat_rec1, c->d_splat_bbox, c->d_keys, c->stream, c->splat_tune,
                                vis) == 0)
            return;
    }
    launch_global_scatter(m.surf[m.cur], m.d_frame, m.d_pose, c->W, c->H, c->K, g.depth_cutoff, 12.0f, g.time_delta, order, m.id, c->d_keys,
                          c->stream, surfel_blocks(c, m));
}

// spawnObjectModel (Core/MaskFusion.cpp:671-684): pose = I, makeStatic(globalPose); moveNewModelToList
static int spawn_object(mf_ctx* c, int id, int classID) {
    const mf_config& g = c->cfg;
    hipStream_t s = c->stream;
    ModelState& bg = *c->models[0];
    std::unique_ptr<ModelState> nm;
    if (!c->pool.empty()) {   // :673-676: take a preallocated model
        nm = std::move(c->pool.front());
        c->pool.erase(c->pool.begin());
        nm->id = id;
        nm->confThr = g.conf_object;
        nm->age = 0; nm->isStatic = true; nm->log_ts.clear(); nm->cur = 0;
        hipLaunchKernelGGL(k_pose_identity, dim3(1), dim3(64), 0, s, nm->d_pose, c->weight_literal ? 1 : 0);
        hipLaunchKernelGGL(k_frame_init, dim3(1), dim3(64), 0, s, nm->d_frame, c->host_tick);
        nm->h_frame->tick = c->host_tick;
    } else {
        int rc = create_model(c, id, g.conf_object, false, surfel_capacity(g.num_osurfels), nm);
        if (rc != MF_OK) return rc;
    }
    nm->classID = classID;
    launch_spawn_pose(nm->d_pose, bg.d_pose, nm->d_frame, bg.d_frame, nm->h_pose, s);
    c->models.push_back(std::move(nm));
    // the private scratch of the batched object passes (>= 2 objects) is allocated HERE, outside the enqueue path of a frame (hipMalloc
    // synchronises the device); a failure only switches the batched passes off -- the model-by-model passes need no private scratch
    if (c->batch_objects && c->models.size() >= 3)
        for (size_t i = 1; i < c->models.size(); ++i)
            if (ensure_obj_scratch(c, *c->models[i]) != MF_OK) { (void)hipGetLastError(); c->batch_objects = false; c->err.clear(); break; }
    return MF_OK;
}

// The tracking loop of processFrame (Core/MaskFusion.cpp:247-276) over models[first..]: every model that is tracked this frame goes
// into one batch (geometric term) or is tracked on its own (photometric term: its scratch images are shared); static objects then follow
// the background's NEW pose (models[0]'s pose: on a context that holds only objects the caller has overridden it with the owner's).
static void enqueue_tracking_loop(mf_ctx* c, size_t first, bool track_all, const float* depthF_prev, long k) {
    ModelState& bg = *c->models[0];
    std::vector<ModelState*> tracked, follow;
    if (first == 0) tracked.push_back(&bg);
    for (size_t i = 1; i < c->models.size(); ++i) {
        ModelState& m = *c->models[i];
        // trackable = trackableClassIds.empty() || trackableClassIds.count(classID), :261
        bool trackable = c->trackable.empty();
        for (int id : c->trackable) trackable |= (id == m.classID);
        if ((!m.isStatic || track_all) && trackable) tracked.push_back(&m);   // jump rule of :268-272 in the finalize step
        else follow.push_back(&m);
    }
    if (!photometric_on(c) && tracked.size() >= 2 && (int)tracked.size() <= kMaxTrackBatch && c->batch_tracking) {
        enqueue_track_batch(c, tracked, depthF_prev, k);
    } else {
        for (ModelState* m : tracked) enqueue_track(c, *m, m == &bg ? depthF_prev : nullptr, m == &bg ? 0.f : 0.2f, k);
    }
    for (ModelState* m : follow) launch_static_pose(m->d_pose, bg.d_pose, m->h_pose, c->stream);   // updateStaticPose, :274
}

// The fusion loop of processFrame (Core/MaskFusion.cpp:539-565) over models[first..]: predictIndices -> fuse -> predictIndices -> clean;
// the object models go through one launch per pass when there are at least two of them ("batchObjectPasses").
static int enqueue_fusion_loop(mf_ctx* c, size_t first, bool multi, const uint8_t* d_rgb, const float* d_depth, const float* depthF,
                               const uint8_t* mask, float weight_multiplier) {
    const mf_config& g = c->cfg;
    const bool batch = multi && batch_objects_now(c);
    for (size_t i = first; i < (batch ? (size_t)1 : c->models.size()); ++i)
        enqueue_fuse_clean(c, *c->models[i], d_rgb, d_depth, depthF, mask, g.depth_cutoff, weight_multiplier, true, i == 0);
    if (batch) {   // every object model: one launch per pass
        std::vector<ModelState*> objs; std::vector<int> orders;
        object_models(c, objs, orders);
        ObjBatch ob; int blocks = 0;
        int rc = make_obj_batch(c, objs, orders, d_rgb, d_depth, depthF, mask, weight_multiplier, nullptr, ob, blocks);
        if (rc != MF_OK) return rc;
        ob.cleanEpoch = next_clean_epoch(c);
        int cblocks = 8;
        ob.cleanTicketLanes = 1;
        for (ModelState* m : objs) cblocks = std::max(cblocks, clean_blocks(c, *m));
        ob.cleanTicketLanes = std::min(c->ticket_lanes, cblocks);
        launch_obj_fuse_clean(ob, blocks, cblocks, c->stream);
        for (ModelState* m : objs) m->cur = 1 - m->cur;   // fuse in place, clean a -> b: b is the live buffer now
    }
    return MF_OK;
}

// predict() (Core/MaskFusion.cpp:569) + tick++ (:573) + the pose log entry (:580-596) + incrementAge (:600) over models[first..].
// first == 1: models[0] is the stand-in of a background that lives in another context -- it is not drawn, but its frame state advances.
static int enqueue_predict_loop(mf_ctx* c, size_t first, bool may_batch, int64_t timestamp, const uint8_t* d_rgb, const float* d_depth,
                                const float* depthF, const uint8_t* mask, float weight_multiplier) {
    const mf_config& g = c->cfg;
    ModelState& bg = *c->models[0];
    auto log_slot = [&](ModelState& m) -> float* {   // MaskFusion.cpp:580-596
        if (!m.d_poselog) return nullptr;
        float* slot = m.d_poselog + (m.log_ts.size() % (size_t)g.pose_log_capacity) * 8;
        m.log_ts.push_back(timestamp);
        return slot;
    };
    if (first > 0) {
        launch_frame_advance(bg.d_frame, c->W, c->H, bg.h_frame, bg.d_pose, nullptr, log_slot(bg), c->stream);
        bg.age++;
    }
    if (may_batch && batch_objects_now(c)) {
        // predict() + tick++ + pose log of the object models in three launches; the background keeps its tiled prediction
        if (first == 0) {
            const FrameAdvance adv0{bg.h_frame, nullptr, log_slot(bg)};
            enqueue_predict(c, bg, &adv0);
            bg.age++;
        }
        std::vector<ModelState*> objs; std::vector<int> orders;
        object_models(c, objs, orders);
        std::vector<float*> slots;
        for (ModelState* m : objs) {
            slots.push_back(log_slot(*m));
            m->pred_gray_valid = photometric_on(c);
            m->age++;
        }
        ObjBatch ob; int blocks = 0;
        int rc = make_obj_batch(c, objs, orders, d_rgb, d_depth, depthF, mask, weight_multiplier, &slots, ob, blocks);
        if (rc != MF_OK) return rc;
        launch_obj_predict_advance(ob, blocks, c->stream);
        return MF_OK;
    }
    for (size_t i = first; i < c->models.size(); ++i) {
        ModelState& m = *c->models[i];
        const FrameAdvance adv{m.h_frame, i == 0 ? nullptr : bg.d_pose, log_slot(m)};
        enqueue_predict(c, m, &adv);   // ... with tick++ / the fill-in decision / the pose log entry as its epilogue
        m.age++;  // incrementAge, :600
    }
    return MF_OK;
}

// inactivateModel (Core/MaskFusion.cpp:699-713) for models[i], i > 0: the pose log moves to the retired list, the MODEL -- its surfel
// buffers, maps and scratch, ~100 MB at VGA -- goes back to the preallocation pool instead of being freed (upstream deletes the data; what a
// later spawn gets is a fresh model either way: spawn_object re-initialises pose, frame state and counters on the stream).  Neither step
// waits for the GPU: round 3 freed ~30 allocations here and allocated them again (plus a drained stream and a 2 MB read-back for the log) at
// the next spawn -- 1.5 of the 2.4 ms of a frame of bench.py --config 2s, whose scene drops and re-spawns a tracked box every other frame.
static int materialise_retired(mf_ctx* c);
static int retire_model(mf_ctx* c, size_t i) {
    ModelState* m = c->models[i].get();   // (everything that can fail happens before the model leaves the list)
    const size_t cap = (size_t)c->cfg.pose_log_capacity, n_all = m->log_ts.size(), n = n_all < cap ? n_all : cap;
    if (m->d_poselog && n > 0) {
        mf_ctx::RetiredLog r;
        r.id = m->id;
        r.ts.assign(m->log_ts.end() - (long)n, m->log_ts.end());
        r.n = n;
        if (!c->d_retired) {
            c->retired_cap = 1u << 17;   // 131 072 entries of 32 B: 4 MB, once
            void* q = nullptr;
            if (hipMalloc(&q, c->retired_cap * 8 * sizeof(float)) == hipSuccess) { c->d_retired = (float*)q; c->allocs.push_back(q); }
            else { (void)hipGetLastError(); c->retired_cap = 0; }
        }
        if (c->d_retired && c->retired_used + n > c->retired_cap && n <= c->retired_cap) {
            int rc = materialise_retired(c);   // arena full: read it back once (a synchronisation every ~130 k retired entries) and start over
            if (rc != MF_OK) return rc;
        }
        if (c->d_retired && c->retired_used + n <= c->retired_cap) {
            // chronological order: entries n_all - n .. n_all - 1 of a ring of `cap` slots -> at most two contiguous pieces
            const size_t first = (n_all - n) % cap, run1 = (first + n <= cap) ? n : cap - first;
            MF_HIP(c, hipMemcpyAsync(c->d_retired + c->retired_used * 8, m->d_poselog + first * 8, run1 * 8 * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
            if (run1 < n)
                MF_HIP(c, hipMemcpyAsync(c->d_retired + (c->retired_used + run1) * 8, m->d_poselog, (n - run1) * 8 * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
            r.arena_off = c->retired_used; r.in_arena = true;
            c->retired_used += n;
        } else {
            std::vector<int64_t> ts;
            int rc = download_pose_log(c, *m, ts, r.p);   // arena full: the synchronising path
            if (rc != MF_OK) return rc;
        }
        c->retired.push_back(std::move(r));
    }
    std::unique_ptr<ModelState> owned = std::move(c->models[i]);
    c->models.erase(c->models.begin() + (long)i);
    owned->id = -1; owned->classID = -1; owned->age = 0; owned->isStatic = true; owned->log_ts.clear(); owned->cur = 0; owned->pred_gray_valid = false;
    owned->maxDepth = FLT_MAX;
    *owned->h_count = 0;
    if (c->vis_tag.model == owned.get()) c->vis_tag.model = nullptr;
    c->pool.push_back(std::move(owned));
    return MF_OK;
}
// the retired logs' entries on the host (export time: a synchronisation is fine there)
static int materialise_retired(mf_ctx* c) {
    bool any = false;
    for (auto& r : c->retired) any |= r.in_arena;
    if (!any) return MF_OK;
    MF_HIP(c, hipStreamSynchronize(c->stream));
    for (auto& r : c->retired) {
        if (!r.in_arena) continue;
        std::vector<float> raw(r.n * 8);
        MF_HIP(c, hipMemcpy(raw.data(), c->d_retired + r.arena_off * 8, r.n * 8 * sizeof(float), hipMemcpyDeviceToHost));
        r.p.clear();
        for (size_t e = 0; e < r.n; ++e) r.p.insert(r.p.end(), raw.data() + e * 8, raw.data() + e * 8 + 7);
        r.in_arena = false;
    }
    c->retired_used = 0;   // every entry is on the host now: the arena starts over
    return MF_OK;
}

static int process_frame_impl(mf_ctx* c, const uint8_t* d_rgb, const float* d_depth, const uint8_t* d_mask_in,
                              const int32_t* class_ids, int n_masks, float weight_multiplier, int64_t timestamp = 0,
                              const float* in_pose16 = nullptr, bool bootstrap = false) {
    const int W = c->W, H = c->H, P = c->P;
    hipStream_t s = c->stream;
    const mf_config& g = c->cfg;
    const bool multi = g.enable_multiple_models != 0;
    // -static (enableMultipleModels == false): everything is background (MaskFusion.cpp:223-230)
    const uint8_t* mask = multi ? c->d_mask_tex : c->d_zero_mask;
    const long k = c->frame_no;
    const int set = (int)(k & 1);
    float* depthF = c->d_depthF[k % 3];
    float* depthF_prev = c->d_depthF[(k + 2) % 3];
    ModelState& bg = *c->models[0];
    bool main_done_recorded = false;
    bool bg_fused = false;
    c->mm_marked = false;

    int prc = enqueue_preprocess(c, d_rgb, d_depth, k, c->map_ready);
    if (prc != MF_OK) return prc;

    if (!c->map_ready) {
        c->map_ready = true;
        mark(c, 2); mark(c, 3); mark(c, 4); mark(c, 5); mark(c, 6);
        // :235-238
        launch_init_surfels(d_rgb, d_depth, depthF, W, H, c->K, g.max_depth_processed, bg.d_frame, c->d_cand_rec, c->d_flags, s);
        bg.cur = 0;
        launch_compact_records(c->d_cand_rec, c->d_flags, P, bg.surf[0], bg.d_frame, c->d_block_counts, bg.h_count, s);
        launch_run_table(bg.surf[0], bg.d_frame, s);
        mark(c, 7);
    } else if (in_pose16 && !bootstrap) {
        // the caller supplies the camera pose: no tracking, no segmentation, object poses untouched
        // (MaskFusion.cpp:243,413-415 -- the whole "regular" block is skipped)
        mark(c, 2);
        launch_override_pose(bg.d_pose, in_pose16, 0, bg.h_pose, s);
        mark(c, 3); mark(c, 4);
        if (!g.rgb_only)   // :539
          for (size_t i = 0; i < c->models.size(); ++i)
            enqueue_fuse_clean(c, *c->models[i], d_rgb, d_depth, depthF, mask, g.depth_cutoff, weight_multiplier, true, i == 0);
        mark(c, 7);
    } else {
        mark(c, 2);
        // tracking, :247-276.  Every model that is tracked this frame goes into one batch (geometric term) or is tracked on its
        // own (photometric term: its scratch images are shared); static objects then follow the background's NEW pose
        enqueue_tracking_loop(c, 0, g.track_all_models != 0, depthF_prev, k);
        if (bootstrap && in_pose16) launch_override_pose(bg.d_pose, in_pose16, 1, bg.h_pose, s);   // :280-283 (after the object loop)
        mark(c, 3);
        if (c->overlap) { MF_HIP(c, hipEventRecord(c->ev_main_done[set], s)); main_done_recorded = true; }

        if (multi) {
            // GlobalProjection::project(models, tick, tick, timeDelta, depthCutoff) (:289) with its fixed threshold 12
            if (batch_objects_now(c)) {
                enqueue_global_projection(c, bg, 0);
                std::vector<ModelState*> objs; std::vector<int> orders;
                object_models(c, objs, orders);
                ObjBatch ob; int blocks = 0;
                int rc = make_obj_batch(c, objs, orders, d_rgb, d_depth, depthF, mask, weight_multiplier, nullptr, ob, blocks);
                if (rc != MF_OK) return rc;
                launch_obj_global_scatter(ob, blocks, s);
            } else {
                for (size_t i = 0; i < c->models.size(); ++i) enqueue_global_projection(c, *c->models[i], (int)i);
            }
            launch_global_resolve(c->d_keys, c->d_proj_ids, P, s);
            if (c->timings_on) (void)hipEventRecord(c->ev_mm[0], s);
            // MfSegmentation::performSegmentation, device half (MfSegmentation.cpp:149-208)
            launch_edge_map(c->d_vmap[set][0], c->d_nmap[set][0], c->d_edge, W, H, c->seg.weightDistance, c->seg.weightConvexity, s);
            launch_edge_binary(c->d_edge, c->d_bin, c->d_tmp_u8, W, H, c->seg.threshold, c->seg.morphEdgeRadius,
                               c->seg.morphEdgeIterations, s);
            const bool haveMasks = d_mask_in && class_ids && n_masks > 0;
            static const int32_t kNoClass[1] = {0};
            SegResult res;
            if (c->gpu_labels) {
                // label stage on the device (mf_labels_gpu.hip): the model table still holds the objects the jump rule may
                // have dropped in this frame -- the kernels read their `alive` flags -- and the only host visit of the frame
                // reads back the new-model decision and those flags
                if (c->spawnOffset < g.model_spawn_offset) c->spawnOffset++;  // :294
                std::vector<SegModelInfo> infos;
                std::vector<const PoseDev*> poses;
                for (auto& m : c->models) { infos.push_back(SegModelInfo{m->id, m->classID}); poses.push_back(m->d_pose); }
                int rc = c->labels->enqueue(c->seg, W, H, c->d_bin, d_depth, haveMasks ? d_mask_in : nullptr, haveMasks ? class_ids : kNoClass,
                                            haveMasks ? n_masks : 0, c->d_proj_ids, infos, poses, c->nextID,
                                            c->spawnOffset >= g.model_spawn_offset, c->d_mask_tex, s);   // writes textureMask (:297)
                if (rc != MF_OK) return rc;
                // The host has to look at the label stage's decision (new model? which objects did the jump rule drop?) before it
                // can enqueue the objects' fusion -- but not before the BACKGROUND's: that model is never spawned or dropped and its
                // fusion only reads the label image on the stream.  So it goes in first and the host waits on an event recorded
                // right behind the label stage: by the time it wakes up and has enqueued the object work, the GPU is still busy
                // with the background's fuse / clean passes (~0.15 ms at VGA) -- the stream never drains inside a frame.  (Model
                // order inside the fusion loop is free: every model's passes run back to back on one stream and touch only its own
                // surfels; upstream fuses the new model first, MaskFusion.cpp:342-353,539-565.)
                if (!c->ev_labels) MF_HIP(c, hipEventCreateWithFlags(&c->ev_labels, hipEventDisableTiming));
                MF_HIP(c, hipEventRecord(c->ev_labels, s));
                if (c->timings_on) (void)hipEventRecord(c->ev_mm[1], s);
                if (!g.rgb_only && c->early_bg_fusion) {
                    enqueue_fuse_clean(c, bg, d_rgb, d_depth, depthF, mask, g.depth_cutoff, weight_multiplier, true, true);
                    bg_fused = true;
                }
                if (c->timings_on) (void)hipEventRecord(c->ev_mm[2], s);
                const auto t_wait0 = std::chrono::steady_clock::now();
                MF_HIP(c, hipEventSynchronize(c->ev_labels));
                c->mm_host_wait_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_wait0).count();
                if (c->timings_on) { (void)hipEventRecord(c->ev_mm[3], s); c->mm_marked = true; }
                if (c->labels->h_result[2]) { c->err = "label stage: vote tables overflowed (too many components x masks)"; return MF_ESTATE; }
                res.hasNewLabel = c->labels->h_result[0] != 0;
                res.newClassID = c->labels->h_result[1];
            } else {
                MF_HIP(c, hipMemcpyAsync(c->h_bin, c->d_bin, (size_t)P, hipMemcpyDeviceToHost, s));
                MF_HIP(c, hipMemcpyAsync(c->h_ids, c->d_proj_ids, (size_t)P, hipMemcpyDeviceToHost, s));
                MF_HIP(c, hipMemcpyAsync(c->h_depth, d_depth, (size_t)P * sizeof(float), hipMemcpyDeviceToHost, s));
                if (haveMasks) MF_HIP(c, hipMemcpyAsync(c->h_mask, d_mask_in, (size_t)P, hipMemcpyDeviceToHost, s));
                MF_HIP(c, hipStreamSynchronize(s));  // the one host visit of a multi-model frame (the reference leaves the GPU here too)
            }

            // inactivateModel for objects the jump rule dropped (:268-272); data is deleted (no re-detection upstream)
            for (size_t i = 1; i < c->models.size();) {
                if (c->models[i]->h_pose->alive == 0) {
                    int rc = retire_model(c, i);
                    if (rc != MF_OK) return rc;
                } else ++i;
            }
            if (!c->gpu_labels) {
                if (c->spawnOffset < g.model_spawn_offset) c->spawnOffset++;  // :294
                std::vector<SegModelInfo> infos;
                for (auto& m : c->models) infos.push_back(SegModelInfo{m->id, m->classID});
                segmentation_host(c->seg, W, H, c->h_bin, c->h_depth, c->h_mask, haveMasks ? class_ids : kNoClass, haveMasks ? n_masks : 0,
                                  c->h_ids, infos, c->nextID, c->spawnOffset >= g.model_spawn_offset, c->ignoreMap, c->h_full, res);
                MF_HIP(c, hipMemcpyAsync(c->d_mask_tex, c->h_full, (size_t)P, hipMemcpyHostToDevice, s));  // :297
            }
            bool spawned = false;
            if (res.hasNewLabel && (int)c->models.size() < g.max_models) {
                int rc = spawn_object(c, take_next_model_id(c), res.newClassID);
                if (rc != MF_OK) return rc;
                c->spawnOffset = 0;
                spawned = true;
            }
            for (size_t i = 1; i < c->models.size(); ++i) c->models[i]->maxDepth = 30.f + 30.f * 1.2f;  // :335-339 (depthMean = depthStd = 30)
            if (spawned)  // :342-353: predictIndices; fuse(maxDepthProcessed, weight 100); clean (no second index pass)
                enqueue_fuse_clean(c, *c->models.back(), d_rgb, d_depth, depthF, mask, g.max_depth_processed, 100.f, false, false);
            for (size_t i = 1; i < c->models.size(); ++i)  // :369-374
                c->models[i]->confThr = fminf(4.5f, (float)c->models[i]->age / 25.0f);
        }
        // (the predict() at MaskFusion.cpp:423 only feeds the dead loop-closure block and is overwritten at :569)
        // fusion, :539-565: if (!rgbOnly && trackingOk && !lost)
        if (!g.rgb_only) {
            int rc = enqueue_fusion_loop(c, bg_fused ? 1 : 0, multi, d_rgb, d_depth, depthF, mask, weight_multiplier);
            if (rc != MF_OK) return rc;
        }
        mark(c, 7);
    }
    {
        int rc = enqueue_predict_loop(c, 0, multi && c->map_ready && k > 0, timestamp, d_rgb, d_depth, depthF, mask, weight_multiplier);
        if (rc != MF_OK) return rc;
    }
    mark(c, 8);
    // every branch records the event (the caller-supplied-pose branch and the first frame do it here, at the end of the frame)
    if (c->overlap && !main_done_recorded) MF_HIP(c, hipEventRecord(c->ev_main_done[set], s));
    c->lastF = (int)(k % 3);
    c->frame_no++;
    c->host_tick++;
    return check_launch(c);
}

extern "C" int mf_process_frame_dev(mf_ctx* c, const uint8_t* d_rgb, const float* d_depth, const uint8_t* d_mask, int64_t timestamp,
                                    float weight_multiplier) {
    if (!c || !d_rgb || !d_depth) return MF_EINVAL;
    // FrameData::classIDs of device-resident masks: the table of mf_set_mask_class_ids (every id class 0 until it is set)
    std::vector<int32_t> cls;
    if (d_mask && c->cfg.enable_multiple_models) {
        if (c->mask_classes.empty()) cls.assign(256, 0);
        else cls = c->mask_classes;
    }
    return process_frame_impl(c, d_rgb, d_depth, d_mask, cls.empty() ? nullptr : cls.data(), (int)cls.size(), weight_multiplier,
                              timestamp);
}

// FrameData::classIDs (Core/FrameData.h:25-48) for frames handed over as device pointers: class_ids[v] is the class of mask value v
extern "C" int mf_set_mask_class_ids(mf_ctx* c, const int32_t* class_ids, int32_t n) {
    if (!c || n < 0 || n > 256 || (n > 0 && !class_ids)) return MF_EINVAL;
    c->mask_classes.assign(class_ids, class_ids + n);
    return MF_OK;
}

extern "C" int mf_sync(mf_ctx* c) {
    if (!c) return MF_EINVAL;
    MF_HIP(c, hipStreamSynchronize(c->stream));
    if (c->timings_on) {
        // event i marks the START of stage i; stage i lasts until event i+1 (labels: see the header)
        float t[MF_N_TIMINGS] = {};
        for (int i = 0; i < 8; ++i) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, c->ev[i], c->ev[i + 1]) == hipSuccess) t[i] = ms;
        }
        float run = 0.f;  // "Run" = the pose-dependent chain on the main stream; preprocessing overlaps the previous frame
        if (hipEventElapsedTime(&run, c->ev[2], c->ev[8]) == hipSuccess) t[8] = run;
        float init = 0.f, iters = 0.f;
        t[1] = 0.f;
        if (c->tracked_once && hipEventElapsedTime(&init, c->ev[2], c->ev_icp[0]) == hipSuccess) t[1] = init;
        if (c->tracked_once && hipEventElapsedTime(&iters, c->ev_icp[0], c->ev_icp[1]) == hipSuccess) t[9] = iters;
        float coarse = 0.f, fine = 0.f;   // launch-per-iteration loop of a single model only (the batched / graph forms record no mid event)
        if (c->tracked_once && c->icp_mid_recorded && hipEventElapsedTime(&coarse, c->ev_icp[0], c->ev_icp_mid) == hipSuccess &&
            hipEventElapsedTime(&fine, c->ev_icp_mid, c->ev_icp[1]) == hipSuccess) { t[10] = coarse; t[11] = fine; }
        if (c->mm_marked) {   // multi-model frame with the device label stage: what labels 3..7 do not show (see the header)
            float v = 0.f;
            if (hipEventElapsedTime(&v, c->ev[3], c->ev_mm[0]) == hipSuccess) t[12] = v;
            if (hipEventElapsedTime(&v, c->ev_mm[0], c->ev_mm[1]) == hipSuccess) t[13] = v;
            if (hipEventElapsedTime(&v, c->ev_mm[1], c->ev_mm[2]) == hipSuccess) t[14] = v;
            if (hipEventElapsedTime(&v, c->ev_mm[2], c->ev_mm[3]) == hipSuccess) t[15] = v;
            if (hipEventElapsedTime(&v, c->ev_mm[3], c->ev[7]) == hipSuccess) t[16] = v;
            t[17] = c->mm_host_wait_ms;
        }
        memcpy(c->last_ms, t, sizeof(t));
    }
    return MF_OK;
}

extern "C" int mf_process_frame(mf_ctx* c, const uint8_t* rgb, const float* depth, const uint8_t* mask, const int32_t* class_ids,
                                int32_t n_masks, int64_t timestamp, const float* in_pose16, float weight_multiplier, int32_t bootstrap) {
    if (!c || !rgb || !depth) return MF_EINVAL;
    if (bootstrap && !in_pose16) { c->err = "bootstrap needs in_pose (MaskFusion.cpp:281)"; return MF_EINVAL; }
    if (c->host_async) {
        // pinned double buffer + asynchronous upload: no hipStreamSynchronize per frame on the boundary a drop-in
        // user sees (MaskFusion.cpp:212-216 uploads FrameData every frame)
        const size_t P = (size_t)c->P;
        const int slot = (int)(c->in_slot++ & 1u);
        const auto t_0 = std::chrono::steady_clock::now();
        MF_HIP(c, hipEventSynchronize(c->ev_in_copied[slot]));        // the staging slot's previous upload (two frames ago) has left it
        const auto t_w = std::chrono::steady_clock::now();
        uint8_t* h = c->h_in[slot];
        memcpy(h, depth, P * sizeof(float));                           // depth | colour | mask: one packed block, one upload
        memcpy(h + c->in_off_rgb, rgb, P * 3);
        if (mask) memcpy(h + c->in_off_mask, mask, P);
        const auto t_1 = std::chrono::steady_clock::now();
        hipStream_t sup = c->stream_in;
        if (c->host_lockstep) MF_HIP(c, hipEventSynchronize(c->ev_in_consumed[slot]));   // (frame k-2 has run: at most two frames are queued)
        MF_HIP(c, hipStreamWaitEvent(sup, c->ev_in_consumed[slot], 0));                  // the frame that read this device block is done
        const size_t up_bytes = mask ? c->in_off_mask + P : c->in_off_rgb + P * 3;
        MF_HIP(c, hipMemcpyAsync(c->d_in_block[slot], h, up_bytes, hipMemcpyHostToDevice, sup));
        MF_HIP(c, hipEventRecord(c->ev_in_copied[slot], sup));
        // (single-model frames: ~60 us of the ~300 the host has to spare.  A multi-model call synchronises in mid-frame for the label stage's
        // decision and has no time to spare: its upload stays a cross-queue wait under the previous frame's tail)
        if (c->host_wait_upload && c->host_lockstep && c->cfg.enable_multiple_models == 0) MF_HIP(c, hipEventSynchronize(c->ev_in_copied[slot]));
        else MF_HIP(c, hipStreamWaitEvent(c->stream, c->ev_in_copied[slot], 0));
        if (c->overlap) MF_HIP(c, hipStreamWaitEvent(c->stream_pre, c->ev_in_copied[slot], 0));
        const auto t_2 = std::chrono::steady_clock::now();
        int rc = process_frame_impl(c, c->d_in_rgb[slot], c->d_in_depth[slot], mask ? c->d_in_mask[slot] : nullptr, class_ids, n_masks,
                                    weight_multiplier, timestamp, in_pose16, bootstrap != 0);
        (void)hipEventRecord(c->ev_in_consumed[slot], c->stream);     // (also on a failed frame: the slot must become reusable)
        const auto t_3 = std::chrono::steady_clock::now();
        auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
        c->host_us[0] += us(t_0, t_1); c->host_us[1] += us(t_1, t_2); c->host_us[2] += us(t_2, t_3); c->host_us[3] += us(t_0, t_3); c->host_us[4] += us(t_0, t_w); c->host_calls++;
        return rc;
    }
    // blocking form: staged on the input stream (the frame is first read there); the previous frame has completed (this call syncs)
    hipStream_t sin = c->overlap ? c->stream_pre : c->stream;
    MF_HIP(c, hipMemcpyAsync(c->d_rgb, rgb, (size_t)c->P * 3, hipMemcpyHostToDevice, sin));
    MF_HIP(c, hipMemcpyAsync(c->d_depth, depth, (size_t)c->P * sizeof(float), hipMemcpyHostToDevice, sin));
    if (mask) MF_HIP(c, hipMemcpyAsync(c->d_mask_in, mask, (size_t)c->P, hipMemcpyHostToDevice, sin));
    int rc = process_frame_impl(c, c->d_rgb, c->d_depth, mask ? c->d_mask_in : nullptr, class_ids, n_masks, weight_multiplier, timestamp,
                                in_pose16, bootstrap != 0);
    if (rc != MF_OK) return rc;
    return mf_sync(c);
}

// the coverage count behind MaskFusion::requiresFillIn belongs to ONE projection: a prediction outside processFrame starts it over
// (inside a frame k_frame_advance has already consumed and zeroed it)
static __global__ void k_reset_cover(FrameDev* f) { if (threadIdx.x == 0 && blockIdx.x == 0) f->cover = 0; }
// ... and re-takes the decision k_frame_advance took for the next tracking step (MaskFusion::requiresFillIn, :630-648)
static __global__ void k_fillin_decision(FrameDev* f, int W, int H, FrameDev* host_mirror) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    f->useFillIn = ((float)f->cover / (float)((W / 20) * (H / 20)) < 0.75f) ? 1 : 0;
    f->cover = 0;
    if (host_mirror) *host_mirror = *f;
}
static void launch_fillin_decision(FrameDev* f, int W, int H, FrameDev* host_mirror, hipStream_t s) {
    hipLaunchKernelGGL(k_fillin_decision, dim3(1), dim3(64), 0, s, f, W, H, host_mirror);
}

extern "C" int mf_predict(mf_ctx* c) {
    if (!c) return MF_EINVAL;
    const uint8_t* keep = c->cur_rgb;
    c->cur_rgb = nullptr;  // the caller's frame buffer may be gone: the fill-in intensity keeps its last contents
    for (auto& m : c->models) {
        hipLaunchKernelGGL(k_reset_cover, dim3(1), dim3(64), 0, c->stream, m->d_frame);
        enqueue_predict(c, *m);
    }
    for (auto& m : c->models) launch_fillin_decision(m->d_frame, c->W, c->H, m->h_frame, c->stream);
    c->cur_rgb = keep;
    return check_launch(c);
}

// MaskFusion::preallocateModels (Core/MaskFusion.cpp:144-149): object-model buffers allocated ahead of time, so that a spawn
// inside a frame costs two tiny kernels instead of ~100 MB of hipMalloc + memset
extern "C" int mf_preallocate_models(mf_ctx* c, uint32_t count) {
    if (!c) return MF_EINVAL;
    for (uint32_t i = 0; i < count; ++i) {
        std::unique_ptr<ModelState> m;
        int rc = create_model(c, -1, c->cfg.conf_object, false, surfel_capacity(c->cfg.num_osurfels), m, c->batch_objects);
        if (rc != MF_OK) return rc;
        c->pool.push_back(std::move(m));
    }
    return check_launch(c);
}

static __global__ void k_set_tick(FrameDev* f, int tick, FrameDev* host_mirror) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    f->tick = tick;
    if (host_mirror) *host_mirror = *f;
}
// MaskFusion::setTick (Core/MaskFusion.h:206): the run loop uses it to start at / skip to a frame number
extern "C" int mf_set_tick(mf_ctx* c, int32_t tick) {
    if (!c || tick < 1) return MF_EINVAL;
    if (!c->map_ready && tick != 1) { c->err = "setTick before the first frame would skip the map initialisation"; return MF_ESTATE; }
    c->host_tick = tick;
    for (auto& m : c->models) hipLaunchKernelGGL(k_set_tick, dim3(1), dim3(64), 0, c->stream, m->d_frame, tick, m->h_frame);
    return check_launch(c);
}

static ModelState* model_at(mf_ctx* c, int32_t i);

// ------------------------------------------------------------------------------------------------
// Model-level entry points: the public operations of Model (Core/Model/Model.h:126-162,233-268) one by one, on a frame
// staged with mf_stage_frame.  MaskFusion::processFrame is a fixed composition of these (mf_process_frame enqueues the
// same launches); they exist so that a caller can drive a model the way the reference's own callers do, and so that each
// surfel pass can be compared with the oracle in isolation.
// ------------------------------------------------------------------------------------------------
static __global__ void k_set_count(FrameDev* f, int count, int* host_count) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    f->count = count; f->countNext = count;
    if (host_count) *host_count = count;
}
static void set_model_tick(mf_ctx* c, ModelState& m, int tick) {
    hipLaunchKernelGGL(k_set_tick, dim3(1), dim3(64), 0, c->stream, m.d_frame, tick, m.h_frame);
}
static long staged_frame(const mf_ctx* c) { return c->frame_no - 1; }   // index of the frame staged / processed last
static const uint8_t* current_mask(const mf_ctx* c) { return c->cfg.enable_multiple_models ? c->d_mask_tex : c->d_zero_mask; }

// upload + MaskFusion::filterDepth (:217) + Model::generateCUDATextures (Model.cpp:350-389) + intensity pyramid: everything of
// processFrame that does not touch a model.  mask: model id per pixel = what textureMask holds for fuse / clean (NULL: left as it is)
extern "C" int mf_stage_frame(mf_ctx* c, const uint8_t* rgb, const float* depth, const uint8_t* mask) {
    if (c) c->vis_tag.model = nullptr;   // (a visibility list belongs to one frame and one pose)
    if (!c || !rgb || !depth) return MF_EINVAL;
    hipStream_t sin = c->overlap ? c->stream_pre : c->stream;
    MF_HIP(c, hipStreamSynchronize(c->stream));
    MF_HIP(c, hipMemcpyAsync(c->d_rgb, rgb, (size_t)c->P * 3, hipMemcpyHostToDevice, sin));
    MF_HIP(c, hipMemcpyAsync(c->d_depth, depth, (size_t)c->P * sizeof(float), hipMemcpyHostToDevice, sin));
    if (mask) MF_HIP(c, hipMemcpyAsync(c->d_mask_tex, mask, (size_t)c->P, hipMemcpyHostToDevice, sin));
    int rc = enqueue_preprocess(c, c->d_rgb, c->d_depth, c->frame_no, true);   // a staged frame always has its vertex / normal maps
    if (rc != MF_OK) return rc;
    c->lastF = (int)(c->frame_no % 3);
    c->frame_no++;
    return mf_sync(c);
}

// The same for a frame that already sits in device memory (e.g. the buffer an RCCL broadcast landed in): nothing is copied, nothing is
// synchronised -- the caller orders the producers of the three buffers on the context's stream (mf_get_stream) and keeps rgb / depth alive
// and unmodified until the frame's model-level calls have completed there.
extern "C" int mf_stage_frame_dev(mf_ctx* c, const uint8_t* d_rgb, const float* d_depth, const uint8_t* d_mask) {
    if (c) c->vis_tag.model = nullptr;   // (a visibility list belongs to one frame and one pose)
    if (!c || !d_rgb || !d_depth) return MF_EINVAL;
    if (d_mask) MF_HIP(c, hipMemcpyAsync(c->d_mask_tex, d_mask, (size_t)c->P, hipMemcpyDeviceToDevice, c->stream));
    if (c->overlap) {
        // with overlapPreprocessing the filter / pyramid kernels run on stream_pre, which otherwise only waits for the previous frame's tracking:
        // the buffers' producers (an RCCL broadcast, a copy) are ordered on `stream` by contract, so the preprocessing stream waits for them here
        if (!c->ev_staged) MF_HIP(c, hipEventCreateWithFlags(&c->ev_staged, hipEventDisableTiming));
        MF_HIP(c, hipEventRecord(c->ev_staged, c->stream));
        MF_HIP(c, hipStreamWaitEvent(c->stream_pre, c->ev_staged, 0));
    }
    int rc = enqueue_preprocess(c, d_rgb, d_depth, c->frame_no, true);
    if (rc != MF_OK) return rc;
    c->lastF = (int)(c->frame_no % 3);
    c->frame_no++;
    return check_launch(c);
}

// Model::initialise (Core/Model/Model.cpp:240-285): the map of `model` becomes the staged frame's point cloud
extern "C" int mf_model_initialise(mf_ctx* c, int32_t model) {
    ModelState* m = model_at(c, model);
    if (!m || c->frame_no == 0) return MF_EINVAL;
    const long k = staged_frame(c);
    launch_init_surfels(c->cur_rgb, c->cur_depth, c->d_depthF[k % 3], c->W, c->H, c->K, c->cfg.max_depth_processed, m->d_frame, c->d_cand_rec,
                        c->d_flags, c->stream);
    launch_compact_records(c->d_cand_rec, c->d_flags, c->P, m->surf[m->cur], m->d_frame, c->d_block_counts, m->h_count, c->stream);
    launch_run_table(m->surf[m->cur], m->d_frame, c->stream);
    c->vis_tag.model = nullptr;
    if (model == 0) c->map_ready = true;
    return check_launch(c);
}

// test / tooling tap with no upstream twin: replace the surfel buffer of `model` (count records of 12 floats, mf_download_map's layout)
extern "C" int mf_model_upload_map(mf_ctx* c, int32_t model, const float* surfels, uint32_t count) {
    ModelState* m = model_at(c, model);
    if (!m || (!surfels && count) || (int)count > m->cap) return MF_EINVAL;
    MF_HIP(c, hipStreamSynchronize(c->stream));
    std::vector<float4> a(count), b(count), d(count);
    for (uint32_t i = 0; i < count; ++i) {
        memcpy(&a[i], surfels + (size_t)i * 12, 16);
        memcpy(&b[i], surfels + (size_t)i * 12 + 4, 16);
        memcpy(&d[i], surfels + (size_t)i * 12 + 8, 16);
    }
    const Surfels& s = m->surf[m->cur];
    if (count) {
        MF_HIP(c, hipMemcpy(s.pc, a.data(), count * sizeof(float4), hipMemcpyHostToDevice));
        MF_HIP(c, hipMemcpy(s.ct, b.data(), count * sizeof(float4), hipMemcpyHostToDevice));
        MF_HIP(c, hipMemcpy(s.nr, d.data(), count * sizeof(float4), hipMemcpyHostToDevice));
    }
    hipLaunchKernelGGL(k_set_count, dim3(1), dim3(64), 0, c->stream, m->d_frame, (int)count, m->h_count);
    launch_run_table(m->surf[m->cur], m->d_frame, c->stream);
    c->vis_tag.model = nullptr;
    if (model == 0) c->map_ready = true;   // the map exists: the next mf_process_frame tracks instead of initialising
    return check_launch(c);
}

// Model::overridePose (Core/Model/Model.h:235-238): lastPose = pose; pose = p
extern "C" int mf_model_override_pose(mf_ctx* c, int32_t model, const float* pose16) {
    ModelState* m = model_at(c, model);
    if (!m || !pose16) return MF_EINVAL;
    launch_override_pose(m->d_pose, pose16, 0, m->h_pose, c->stream);
    c->vis_tag.model = nullptr;   // a visibility list belongs to ONE pose
    return check_launch(c);
}

// Model::computeFusionWeight(weightMultiplier) (Core/Model/Model.cpp:449-464) from the model's pose and lastPose
extern "C" int mf_model_fusion_weight(mf_ctx* c, int32_t model, float weight_multiplier, float* out) {
    ModelState* m = model_at(c, model);
    if (!m || !out) return MF_EINVAL;
    int rc = mf_sync(c);
    if (rc != MF_OK) return rc;
    *out = m->h_pose->fusionWeight * weight_multiplier;
    return MF_OK;
}

// Model::performTracking(frameToFrameRGB, rgbOnly, icpWeight, pyramid, fastOdom, so3, maxDepthProcessed, rgb, logTimestamp,
// tryFillIn) (Core/Model/Model.h:135-136, Model.cpp:427-447) against the staged frame.  frameToFrameRGB must be 0 (the
// reference's only caller passes false, MaskFusion.cpp:248).
extern "C" int mf_model_perform_tracking(mf_ctx* c, int32_t model, int32_t frame_to_frame_rgb, int32_t rgb_only, float icp_weight,
                                         int32_t pyramid, int32_t fast_odom, int32_t so3, float max_depth_processed, int64_t log_timestamp,
                                         int32_t try_fill_in) {
    if (c) c->vis_tag.model = nullptr;   // (a visibility list belongs to one frame and one pose)
    ModelState* m = model_at(c, model);
    (void)log_timestamp;   // only forwarded to a debug print upstream
    if (!m || c->frame_no == 0) return MF_EINVAL;
    const long k = staged_frame(c);
    // The intensity pyramid, the derivative images and the gate images of the staged frame were built by mf_stage_frame under the CONTEXT's
    // configuration, and the intensity of the last prediction by that prediction: a per-call photometric term on a context that never
    // computed them would track against stale or empty images.  Refuse loudly instead (the context must be created with icpWeight < 100
    // or rgbOnly for a photometric term; SO(3) is guarded inside enqueue_track and simply not run without its two pyramids).
    if (rgb_only != 0 || icp_weight < 100.f) {
        const int set = (int)(k & 1);
        if (c->gray_frame[set] != k || c->deriv_frame != k || !m->pred_gray_valid) {
            c->err = "performTracking: a photometric term was requested (rgbOnly or icpWeight < 100) but the staged frame / the last prediction carry no "
                     "intensity and derivative images -- the context was configured without one (icpWeight >= 100, rgbOnly = false) when they were built";
            return MF_ESTATE;
        }
    }
    const mf_config keep = c->cfg;
    const bool keep_ftf = c->ftf_rgb;
    c->ftf_rgb = frame_to_frame_rgb != 0;   // which image initRGBModel takes (Model.cpp:399-400); the fill-in image itself was built by the last prediction
    c->cfg.rgb_only = rgb_only; c->cfg.icp_weight = icp_weight; c->cfg.pyramid = pyramid; c->cfg.fast_odom = fast_odom; c->cfg.so3 = so3;
    c->cfg.max_depth_processed = max_depth_processed;
    // tryFillIn = MaskFusion::requiresFillIn(model) (:630-648): the decision itself is taken on the device from the coverage of the
    // last prediction; here it only gates whether the fill-in source (the previous frame's filtered depth) is offered at all
    // object models carry the 0.2 m jump rule of the caller (MaskFusion.cpp:268-272): pose->alive = 0 marks "remove this model"
    enqueue_track(c, *m, (try_fill_in && m->allowFillIn) ? c->d_depthF[(k + 2) % 3] : nullptr, model == 0 ? 0.f : 0.2f, k);
    c->cfg = keep;
    c->ftf_rgb = keep_ftf;
    return check_launch(c);
}

// Model::predictIndices(time, maxDepth, timeDelta) (Core/Model/Model.h:162, ModelProjection.cpp:100-152)
extern "C" int mf_model_predict_indices(mf_ctx* c, int32_t model, int32_t time, float max_depth, int32_t time_delta) {
    ModelState* m = model_at(c, model);
    if (!m) return MF_EINVAL;
    hipStream_t s = c->stream;
    set_model_tick(c, *m, time);
    launch_index_scatter(m->surf[m->cur], m->d_frame, m->d_pose, c->W, c->H, c->K, max_depth, time_delta, c->d_keys, false, s);
    launch_index_resolve(m->surf[m->cur], m->d_pose, c->d_keys, c->W, c->H, c->d_index, c->d_ivc, c->d_inr, c->d_ict, nullptr, s);
    if (c->model_api_packed) {   // the layout mf_process_frame feeds clean() with: packed records, column-major
        launch_index_scatter(m->surf[m->cur], m->d_frame, m->d_pose, c->W, c->H, c->K, max_depth, time_delta, c->d_keys, true, s);
        launch_index_resolve(m->surf[m->cur], m->d_pose, c->d_keys, c->W, c->H, nullptr, nullptr, nullptr, nullptr, c->d_iclean, s);
    }
    return check_launch(c);
}

// Model::fuse(time, rgb, mask, depthRaw, depthFiltered, depthCutoff, weightMultiplier) (Core/Model/Model.h:142-143,
// Model.cpp:466-647) with the staged frame's textures: data association against the index map of the last
// mf_model_predict_indices of THIS model, then the update pass (in place).
extern "C" int mf_model_fuse(mf_ctx* c, int32_t model, int32_t time, float depth_cutoff, float weight_multiplier) {
    ModelState* m = model_at(c, model);
    if (!m || c->frame_no == 0) return MF_EINVAL;
    hipStream_t s = c->stream;
    const long k = staged_frame(c);
    set_model_tick(c, *m, time);
    launch_fuse_data(c->cur_rgb, c->cur_depth, c->d_depthF[k % 3], current_mask(c), m->id, m->d_frame, m->d_pose, weight_multiplier,
                     fminf(depth_cutoff, m->maxDepth), c->W, c->H, c->K, c->d_index, c->d_ivc, c->d_inr, c->d_cand_op, c->d_cand_rec,
                     c->d_upd_first, c->d_cand_best, s, c->bbox_limit ? 1 : 0);
    launch_fuse_update(m->surf[m->cur], m->d_frame, c->d_upd_first, c->d_cand_op, c->d_cand_best, c->d_cand_rec, c->W, c->H, s);   // in place
    return check_launch(c);
}

// Model::clean(time, graph, timeDelta, depthCutoff, isFern, depthFiltered, mask) (Core/Model/Model.h:146-147, Model.cpp:649-772):
// uses the index map of the last mf_model_predict_indices and the new-surfel records of the last mf_model_fuse of THIS model
extern "C" int mf_model_clean(mf_ctx* c, int32_t model, int32_t time, int32_t time_delta, float depth_cutoff) {
    ModelState* m = model_at(c, model);
    (void)depth_cutoff;   // the maxDepth uniform of copy_unstable.vert is never read (:53-157)
    if (!m || c->frame_no == 0) return MF_EINVAL;
    const long k = staged_frame(c);
    set_model_tick(c, *m, time);
    const int src = m->cur, dst = 1 - m->cur;
    const bool packed = c->model_api_packed != 0;
    launch_clean(m->surf[src], m->surf[dst], m->d_frame, m->d_pose, c->W, c->H, c->K, time_delta, m->confThr, c->cfg.outlier_coefficient, m->id,
                 c->d_index, c->d_ivc, c->d_ict, packed ? c->d_iclean : nullptr, c->d_depthF[k % 3], current_mask(c), c->d_cand_op, c->d_cand_rec,
                 c->d_flags, c->d_newconf, c->d_scan_state, c->d_clean_ctl, next_clean_epoch(c), clean_blocks(c, *m), c->ticket_lanes, m->h_count, packed, c->clean_literal,
                 c->stream);
    m->cur = dst;
    return check_launch(c);
}

// Model::combinedPredict(maxDepth, time, maxTime, timeDelta, ACTIVE) (Core/Model/Model.h:158, ModelProjection.cpp:187-268);
// the reference only ever calls it with time == maxTime (MaskFusion.cpp:616-628)
extern "C" int mf_model_combined_predict(mf_ctx* c, int32_t model, float max_depth, int32_t time, int32_t max_time, int32_t time_delta) {
    ModelState* m = model_at(c, model);
    if (!m || time != max_time) return MF_EINVAL;
    set_model_tick(c, *m, time);
    const mf_config keep = c->cfg;
    c->cfg.max_depth_processed = max_depth; c->cfg.time_delta = time_delta;
    hipLaunchKernelGGL(k_reset_cover, dim3(1), dim3(64), 0, c->stream, m->d_frame);
    enqueue_predict(c, *m);
    c->cfg = keep;
    return check_launch(c);
}

// the tail of processFrame for a frame driven through the Model-level calls: tick++ (:573), fill-in decision for the next
// tracking step (requiresFillIn), pose log entry (:580-596), age++ (:600)
extern "C" int mf_end_frame(mf_ctx* c, int64_t timestamp) {
    if (!c) return MF_EINVAL;
    ModelState& bg = *c->models[0];
    for (auto& m : c->models) {
        float* slot = nullptr;
        if (m->d_poselog) {
            slot = m->d_poselog + (m->log_ts.size() % (size_t)c->cfg.pose_log_capacity) * 8;
            m->log_ts.push_back(timestamp);
        }
        launch_frame_advance(m->d_frame, c->W, c->H, m->h_frame, m->d_pose, m.get() == &bg ? nullptr : bg.d_pose, slot, c->stream);
        m->age++;
    }
    c->host_tick++;
    return check_launch(c);
}

// ------------------------------------------------------------------------------------------------
// The three per-model LOOPS of MaskFusion::processFrame as calls over this context's model list, for a caller that sequences a frame
// itself (maskfusion_amd/sharded.py: one scene, its models spread over several contexts).  They run exactly what mf_process_frame runs for
// these loops -- the batched Gauss-Newton loop over all tracked models, one launch per surfel pass for all object models -- where the
// Model-level calls above cost ~21 + ~12 launches per model.  first_model = 0: the whole list; 1: models[0] is the stand-in of a
// background that lives in another context (its pose is set with mf_model_override_pose; it is neither tracked, fused nor drawn).
// Configuration (rgbOnly, icpWeight, pyramid, fastOdom, so3, depth limits, timeDelta) is the context's.
// ------------------------------------------------------------------------------------------------
// the tracking loop, Core/MaskFusion.cpp:247-276 (trackable classes, static objects follow the background, the 0.2 m jump rule)
extern "C" int mf_track_models(mf_ctx* c, int32_t first_model, int32_t track_all_models) {
    if (c) c->vis_tag.model = nullptr;   // (a visibility list belongs to one frame and one pose)
    if (!c || c->frame_no == 0 || first_model < 0 || first_model > 1 || (first_model == 0 && !c->map_ready)) return MF_EINVAL;
    const long k = staged_frame(c);
    if (photometric_on(c)) {   // the same guard as mf_model_perform_tracking
        const int set = (int)(k & 1);
        bool ok = c->gray_frame[set] == k && c->deriv_frame == k;
        for (size_t i = (size_t)first_model; i < c->models.size(); ++i) ok = ok && c->models[i]->pred_gray_valid;
        if (!ok) { c->err = "mf_track_models: photometric term configured, but the staged frame / a prediction carries no intensity images"; return MF_ESTATE; }
    }
    enqueue_tracking_loop(c, (size_t)first_model, track_all_models != 0, c->d_depthF[(k + 2) % 3], k);
    return check_launch(c);
}
// the fusion loop, Core/MaskFusion.cpp:539-565, preceded -- when spawned_model >= 1 -- by the spawn-frame pass of that model
// (:342-353: predictIndices; fuse(maxDepthProcessed, weight 100); clean) and by the per-frame object parameters (:335-339, :369-374)
extern "C" int mf_fuse_models(mf_ctx* c, int32_t first_model, float weight_multiplier, int32_t spawned_model) {
    if (!c || c->frame_no == 0 || first_model < 0 || first_model > 1 || (first_model == 0 && !c->map_ready) || spawned_model == 0 ||
        spawned_model >= (int32_t)c->models.size())
        return MF_EINVAL;
    const mf_config& g = c->cfg;
    const long k = staged_frame(c);
    const float* depthF = c->d_depthF[k % 3];
    const uint8_t* mask = current_mask(c);
    for (size_t i = 1; i < c->models.size(); ++i) c->models[i]->maxDepth = 30.f + 30.f * 1.2f;
    if (spawned_model > 0)
        enqueue_fuse_clean(c, *c->models[spawned_model], c->cur_rgb, c->cur_depth, depthF, mask, g.max_depth_processed, 100.f, false, false);
    for (size_t i = 1; i < c->models.size(); ++i) c->models[i]->confThr = fminf(4.5f, (float)c->models[i]->age / 25.0f);
    if (!g.rgb_only) {
        if (first_model == 0 && c->bg_fused_frame == k) first_model = 1;   // mf_fuse_background has run for this frame
        int rc = enqueue_fusion_loop(c, (size_t)first_model, g.enable_multiple_models != 0, c->cur_rgb, c->cur_depth, depthF, mask, weight_multiplier);
        if (rc != MF_OK) return rc;
    }
    return check_launch(c);
}
// predict() (:569) and the tail of the frame (tick++ :573, pose log :580-596, incrementAge :600) -- the end of a frame driven through
// mf_stage_frame / mf_track_models / mf_fuse_models (do not call mf_end_frame as well)
extern "C" int mf_predict_models(mf_ctx* c, int32_t first_model, int64_t timestamp) {
    if (!c || c->frame_no == 0 || first_model < 0 || first_model > 1 || (first_model == 0 && !c->map_ready)) return MF_EINVAL;
    const long k = staged_frame(c);
    int rc = enqueue_predict_loop(c, (size_t)first_model, c->cfg.enable_multiple_models != 0, timestamp, c->cur_rgb, c->cur_depth, c->d_depthF[k % 3],
                                  current_mask(c), 1.0f);
    if (rc != MF_OK) return rc;
    c->host_tick++;
    return check_launch(c);
}
// mf_model_state_dev for every model of the list: d_out16[i * 16 ..] = state of models[i] (one call per frame instead of one per model)
extern "C" int mf_models_state_dev(mf_ctx* c, float* d_out16, int32_t capacity) {
    if (!c || !d_out16 || capacity < (int32_t)c->models.size()) return MF_EINVAL;
    for (size_t i = 0; i < c->models.size(); ++i) launch_model_state(c->models[i]->d_pose, c->models[i]->d_frame, d_out16 + 16 * i, c->stream);
    return check_launch(c);
}

// Model::makeNonStatic / makeStatic(globalPose) / isNonstatic (Core/Model/Model.h:263-268): a non-static object model is
// tracked even when trackAllModels is off; makeStatic re-anchors it to the background's current pose
extern "C" int mf_make_nonstatic(mf_ctx* c, int32_t model) {
    ModelState* m = model_at(c, model);
    if (!m) return MF_EINVAL;
    m->isStatic = false;
    return MF_OK;
}
static __global__ void k_make_static(PoseDev* obj, const PoseDev* bg, PoseDev* host_mirror) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    // initialC2Winv = pose * globalPose^-1
    PoseDev p = *obj;
    for (int r = 0; r < 3; ++r)
        for (int col = 0; col < 3; ++col)
            p.initR[r * 3 + col] = p.R[r * 3] * bg->Ri[col] + p.R[r * 3 + 1] * bg->Ri[3 + col] + p.R[r * 3 + 2] * bg->Ri[6 + col];
    for (int r = 0; r < 3; ++r)
        p.initT[r] = p.R[r * 3] * bg->ti[0] + p.R[r * 3 + 1] * bg->ti[1] + p.R[r * 3 + 2] * bg->ti[2] + p.t[r];
    *obj = p;
    if (host_mirror) *host_mirror = p;
}
extern "C" int mf_make_static(mf_ctx* c, int32_t model) {
    if (c) c->vis_tag.model = nullptr;   // (a visibility list belongs to one frame and one pose)
    ModelState* m = model_at(c, model);
    if (!m || model == 0) return MF_EINVAL;
    hipLaunchKernelGGL(k_make_static, dim3(1), dim3(64), 0, c->stream, m->d_pose, c->models[0]->d_pose, m->h_pose);
    m->isStatic = true;
    return check_launch(c);
}
// MaskFusion::setTrackableClassIds (Core/MaskFusion.h:246, MaskFusion.cpp:261,940); n = 0 clears the set (everything trackable)
extern "C" int mf_set_trackable_class_ids(mf_ctx* c, const int32_t* ids, int32_t n) {
    if (!c || n < 0 || (n > 0 && !ids)) return MF_EINVAL;
    c->trackable.assign(ids, ids + n);
    return MF_OK;
}

// ------------------------------------------------------------------------------------------------
// Model-sharded scenes: several contexts (one per GPU) each own some of the models of ONE scene.  The couplings
// of MaskFusion::processFrame between models -- the z-merged model-id image (GlobalProjection), the label image and the
// background pose -- cross the contexts through these calls; maskfusion_amd/sharded.py sequences them with RCCL collectives.
// ------------------------------------------------------------------------------------------------
// GlobalProjection::project (Core/Model/GlobalProjection.cpp:43-107) of this context's models only
extern "C" int mf_export_projection_keys_dev(mf_ctx* c, const int32_t* orders, int32_t n_orders, uint64_t* d_keys_out) {
    if (!c || !d_keys_out || n_orders != (int32_t)c->models.size() || (n_orders > 0 && !orders)) return MF_EINVAL;
    hipStream_t s = c->stream;
    bool all_objects = batch_objects_now(c) && c->frame_no > 0;
    for (size_t i = 1; i < c->models.size(); ++i) all_objects = all_objects && orders[i] >= 0;
    if (all_objects) {   // as in mf_process_frame: the object models' sprites in one launch
        if (orders[0] >= 0) enqueue_global_projection(c, *c->models[0], orders[0]);
        std::vector<ModelState*> objs; std::vector<int> ord;
        for (size_t i = 1; i < c->models.size(); ++i) { objs.push_back(c->models[i].get()); ord.push_back(orders[i]); }
        const long k = staged_frame(c);
        ObjBatch ob; int blocks = 0;
        int rc = make_obj_batch(c, objs, ord, c->cur_rgb, c->cur_depth, c->d_depthF[k % 3], current_mask(c), 1.0f, nullptr, ob, blocks);
        if (rc != MF_OK) return rc;
        launch_obj_global_scatter(ob, blocks, s);
    } else
    for (size_t i = 0; i < c->models.size(); ++i) {
        ModelState& m = *c->models[i];
        if (orders[i] < 0) continue;   // a stand-in (e.g. the background on a rank that only holds objects): not drawn
        enqueue_global_projection(c, m, orders[i]);
    }
    MF_HIP(c, hipMemcpyAsync(d_keys_out, c->d_keys, (size_t)c->P * sizeof(unsigned long long), hipMemcpyDeviceToDevice, s));
    launch_fill_keys(c->d_keys, c->P, s);
    return check_launch(c);
}
// GlobalProjection::downloadDirect (:109-114) of a key image merged over all contexts (per-pixel minimum)
extern "C" int mf_import_projection_keys_dev(mf_ctx* c, const uint64_t* d_keys) {
    if (!c || !d_keys) return MF_EINVAL;
    MF_HIP(c, hipMemcpyAsync(c->d_keys, d_keys, (size_t)c->P * sizeof(unsigned long long), hipMemcpyDeviceToDevice, c->stream));
    launch_global_resolve(c->d_keys, c->d_proj_ids, c->P, c->stream);   // leaves the key image empty again
    return check_launch(c);
}
// MaskFusion::performSegmentation (Core/MaskFusion.h:59; MfSegmentation::performSegmentation, MfSegmentation.cpp:83-538) on the
// staged frame: geometric edges of its vertex / normal maps, then the label stage against `mask` (host, may be NULL) and the
// projected-id image of the last global projection.  model_ids == NULL: this context's own model list; otherwise the GLOBAL list
// (index 0 = background).  The result becomes textureMask (mf_download_segmentation / mf_export_segmentation_dev).  Synchronous.
static int segmentation_enqueue(mf_ctx* c, const uint8_t* mask, const int32_t* class_ids, int32_t n_masks, const int32_t* model_ids,
                                const int32_t* model_class_ids, int32_t n_models, int32_t next_model_id, int32_t allow_new,
                                int32_t* has_new_label, int32_t* new_class_id) {
    if (!c || c->frame_no == 0 || (model_ids && (!model_class_ids || n_models < 1))) return MF_EINVAL;
    if (n_masks < 0 || n_masks > 256 || (n_masks > 0 && (!mask || !class_ids))) return MF_EINVAL;
    hipStream_t s = c->stream;
    const int set = (int)(staged_frame(c) & 1);
    launch_edge_map(c->d_vmap[set][0], c->d_nmap[set][0], c->d_edge, c->W, c->H, c->seg.weightDistance, c->seg.weightConvexity, s);
    launch_edge_binary(c->d_edge, c->d_bin, c->d_tmp_u8, c->W, c->H, c->seg.threshold, c->seg.morphEdgeRadius, c->seg.morphEdgeIterations, s);
    if (n_masks > 0) MF_HIP(c, hipMemcpyAsync(c->d_mask_in, mask, (size_t)c->P, hipMemcpyHostToDevice, s));
    std::vector<SegModelInfo> infos;
    std::vector<const PoseDev*> poses;
    if (model_ids) {
        for (int i = 0; i < n_models; ++i) {
            infos.push_back(SegModelInfo{model_ids[i], model_class_ids[i]});
            const PoseDev* p = c->models[0]->d_pose;   // models that live in another context: "alive" (the background never dies)
            for (auto& m : c->models) if (m->id == model_ids[i]) p = m->d_pose;
            poses.push_back(p);
        }
    } else {
        for (auto& m : c->models) { infos.push_back(SegModelInfo{m->id, m->classID}); poses.push_back(m->d_pose); }
        next_model_id = c->nextID;
    }
    static const int32_t kNoClass[1] = {0};
    int rc = c->labels->enqueue(c->seg, c->W, c->H, c->d_bin, c->cur_depth, n_masks > 0 ? c->d_mask_in : nullptr, n_masks > 0 ? class_ids : kNoClass,
                                n_masks, c->d_proj_ids, infos, poses, next_model_id, allow_new != 0, c->d_mask_tex, s);
    if (rc != MF_OK) return rc;
    if (!has_new_label) {   // mf_perform_segmentation_begin: the decision is read by mf_perform_segmentation_end
        if (!c->ev_labels) MF_HIP(c, hipEventCreateWithFlags(&c->ev_labels, hipEventDisableTiming));
        MF_HIP(c, hipEventRecord(c->ev_labels, s));
        c->labels_pending = true;
        return MF_OK;
    }
    MF_HIP(c, hipStreamSynchronize(s));
    if (c->labels->h_result[2]) { c->err = "label stage: vote tables overflowed (too many components x masks)"; return MF_ESTATE; }
    *has_new_label = c->labels->h_result[0] != 0;
    *new_class_id = c->labels->h_result[1];
    return MF_OK;
}
// The same in two halves, so that work that does not depend on the decision runs on the GPU while the host waits for it -- what
// mf_process_frame does with the background's fusion ("earlyBackgroundFusion"): _begin enqueues the label stage and returns; the caller may
// enqueue mf_fuse_background (the background is never spawned or dropped, its fusion reads only the label image, which is complete on the
// stream); _end waits for the label stage alone and hands out the decision.
extern "C" int mf_perform_segmentation_begin(mf_ctx* c, const uint8_t* mask, const int32_t* class_ids, int32_t n_masks, const int32_t* model_ids,
                                             const int32_t* model_class_ids, int32_t n_models, int32_t next_model_id, int32_t allow_new) {
    if (!c || c->labels_pending) return MF_EINVAL;
    return segmentation_enqueue(c, mask, class_ids, n_masks, model_ids, model_class_ids, n_models, next_model_id, allow_new, nullptr, nullptr);
}
extern "C" int mf_perform_segmentation_end(mf_ctx* c, int32_t* has_new_label, int32_t* new_class_id) {
    if (!c || !has_new_label || !new_class_id || !c->labels_pending) return MF_EINVAL;
    c->labels_pending = false;
    MF_HIP(c, hipEventSynchronize(c->ev_labels));
    if (c->labels->h_result[2]) { c->err = "label stage: vote tables overflowed (too many components x masks)"; return MF_ESTATE; }
    *has_new_label = c->labels->h_result[0] != 0;
    *new_class_id = c->labels->h_result[1];
    return MF_OK;
}
// the background's share of the fusion loop (Core/MaskFusion.cpp:539-565 for models.front()), ahead of mf_fuse_models, which then skips it
extern "C" int mf_fuse_background(mf_ctx* c, float weight_multiplier) {
    if (!c || c->frame_no == 0 || !c->map_ready) return MF_EINVAL;
    const long k = staged_frame(c);
    if (c->bg_fused_frame == k) { c->err = "mf_fuse_background: the background of this frame is already fused"; return MF_ESTATE; }
    if (!c->cfg.rgb_only)
        enqueue_fuse_clean(c, *c->models[0], c->cur_rgb, c->cur_depth, c->d_depthF[k % 3], current_mask(c), c->cfg.depth_cutoff, weight_multiplier, true, false);
    c->bg_fused_frame = k;
    return check_launch(c);
}
extern "C" int mf_perform_segmentation(mf_ctx* c, const uint8_t* mask, const int32_t* class_ids, int32_t n_masks, const int32_t* model_ids,
                                       const int32_t* model_class_ids, int32_t n_models, int32_t next_model_id, int32_t allow_new,
                                       int32_t* has_new_label, int32_t* new_class_id) {
    if (!has_new_label || !new_class_id || (c && c->labels_pending)) return MF_EINVAL;
    return segmentation_enqueue(c, mask, class_ids, n_masks, model_ids, model_class_ids, n_models, next_model_id, allow_new, has_new_label, new_class_id);
}
extern "C" int mf_export_segmentation_dev(mf_ctx* c, uint8_t* d_out) {
    if (!c || !d_out) return MF_EINVAL;
    MF_HIP(c, hipMemcpyAsync(d_out, c->d_mask_tex, (size_t)c->P, hipMemcpyDeviceToDevice, c->stream));
    return MF_OK;
}
// textureMask->Upload(fullSegmentation) (Core/MaskFusion.cpp:297) with a label image computed by another context
extern "C" int mf_import_segmentation_dev(mf_ctx* c, const uint8_t* d_in) {
    if (!c || !d_in) return MF_EINVAL;
    MF_HIP(c, hipMemcpyAsync(c->d_mask_tex, d_in, (size_t)c->P, hipMemcpyDeviceToDevice, c->stream));
    return MF_OK;
}
// spawnObjectModel (Core/MaskFusion.cpp:671-684) with an id chosen by the caller (the context that runs the label stage owns
// getNextModelID); the new model is appended to this context's list, anchored to its background pose
extern "C" int mf_spawn_object_model(mf_ctx* c, int32_t id, int32_t class_id) {
    if (!c || id < 0 || id > 255) return MF_EINVAL;
    for (auto& m : c->models) if (m->id == id) { c->err = "model id in use"; return MF_EINVAL; }
    int rc = spawn_object(c, id, class_id);
    if (rc != MF_OK) return rc;
    c->models.back()->maxDepth = 30.f + 30.f * 1.2f;   // :335-339 (depthMean = depthStd = 30)
    return check_launch(c);
}
// inactivateModel (Core/MaskFusion.cpp:686-713): the model leaves the list, its pose log is kept for exportPoses
extern "C" int mf_drop_model(mf_ctx* c, int32_t model) {
    ModelState* m = model_at(c, model);
    if (!m || model == 0) return MF_EINVAL;
    return retire_model(c, (size_t)model);
}
// Model::updateStaticPose(globalPose) (Core/Model/Model.h:263): pose = initialC2Winv * background pose
extern "C" int mf_model_update_static_pose(mf_ctx* c, int32_t model) {
    if (c) c->vis_tag.model = nullptr;   // (a visibility list belongs to one frame and one pose)
    ModelState* m = model_at(c, model);
    if (!m || model == 0) return MF_EINVAL;
    launch_static_pose(m->d_pose, c->models[0]->d_pose, m->h_pose, c->stream);
    return check_launch(c);
}
// the per-frame object bookkeeping of processFrame for this context's object models: setMaxDepth (Core/MaskFusion.cpp:335-339)
// and the confidence ramp min(4.5, age / 25) (:369-374)
extern "C" int mf_update_object_params(mf_ctx* c) {
    if (!c) return MF_EINVAL;
    for (size_t i = 1; i < c->models.size(); ++i) {
        c->models[i]->maxDepth = 30.f + 30.f * 1.2f;
        c->models[i]->confThr = fminf(4.5f, (float)c->models[i]->age / 25.0f);
    }
    return MF_OK;
}

extern "C" int mf_get_tick(mf_ctx* c, int32_t* tick) {
    if (!c || !tick) return MF_EINVAL;
    *tick = c->host_tick;
    return MF_OK;
}
extern "C" int mf_num_models(mf_ctx* c, int32_t* n) {
    if (!c || !n) return MF_EINVAL;
    *n = (int32_t)c->models.size();
    return MF_OK;
}
static ModelState* model_at(mf_ctx* c, int32_t i) {
    if (!c || i < 0 || i >= (int32_t)c->models.size()) return nullptr;
    return c->models[i].get();
}
// the ids of the model list in list order -- host state only, no synchronisation (mf_model_info waits for the surfel count)
extern "C" int mf_get_model_ids(mf_ctx* c, int32_t* ids, int32_t capacity, int32_t* n) {
    if (!c || !ids || !n || capacity < (int32_t)c->models.size()) return MF_EINVAL;
    for (size_t i = 0; i < c->models.size(); ++i) ids[i] = c->models[i]->id;
    *n = (int32_t)c->models.size();
    return MF_OK;
}
extern "C" int mf_get_pose(mf_ctx* c, int32_t model, float* out) {
    ModelState* m = model_at(c, model);
    if (!m || !out) return MF_EINVAL;
    int rc = mf_sync(c);
    if (rc != MF_OK) return rc;
    const PoseDev& p = *m->h_pose;
    for (int r = 0; r < 3; ++r) {
        for (int col = 0; col < 3; ++col) out[col * 4 + r] = p.R[r * 3 + col];
        out[12 + r] = p.t[r];
        out[r * 4 + 3] = 0.f;
    }
    out[15] = 1.f;
    return MF_OK;
}
extern "C" int mf_get_surfel_count(mf_ctx* c, int32_t model, uint32_t* count) {
    ModelState* m = model_at(c, model);
    if (!m || !count) return MF_EINVAL;
    int rc = mf_sync(c);
    if (rc != MF_OK) return rc;
    *count = (uint32_t)*m->h_count;
    return MF_OK;
}
extern "C" int mf_model_info(mf_ctx* c, int32_t model, mf_model_info_t* out) {
    ModelState* m = model_at(c, model);
    if (!m || !out) return MF_EINVAL;
    int rc = mf_sync(c);
    if (rc != MF_OK) return rc;
    out->id = m->id; out->class_id = m->classID; out->surfels = (uint32_t)*m->h_count; out->confidence_threshold = m->confThr;
    out->is_static = m->isStatic ? 1 : 0; out->age = m->age;
    return MF_OK;
}
extern "C" int mf_model_state_dev(mf_ctx* c, int32_t model, float* d_out16) {
    ModelState* m = model_at(c, model);
    if (!m || !d_out16) return MF_EINVAL;
    launch_model_state(m->d_pose, m->d_frame, d_out16, c->stream);
    return check_launch(c);
}
extern "C" int mf_get_icp_stats(mf_ctx* c, int32_t model, float* e, float* n) {
    ModelState* m = model_at(c, model);
    if (!m || !e || !n) return MF_EINVAL;
    int rc = mf_sync(c);
    if (rc != MF_OK) return rc;
    *e = m->h_pose->lastICPError; *n = m->h_pose->lastICPCount;
    return MF_OK;
}
// Model::getPoseLog: entries in chronological order (the device ring keeps the last pose_log_capacity of them)
static int download_pose_log(mf_ctx* c, ModelState& m, std::vector<int64_t>& ts, std::vector<float>& p7) {
    ts.clear(); p7.clear();
    if (!m.d_poselog || m.log_ts.empty()) return MF_OK;
    MF_HIP(c, hipStreamSynchronize(c->stream));
    const size_t cap = (size_t)c->cfg.pose_log_capacity, n = m.log_ts.size(), keep = n < cap ? n : cap;
    std::vector<float> raw(cap * 8);
    MF_HIP(c, hipMemcpy(raw.data(), m.d_poselog, cap * 8 * sizeof(float), hipMemcpyDeviceToHost));
    for (size_t e = n - keep; e < n; ++e) {
        ts.push_back(m.log_ts[e]);
        const float* r = raw.data() + (e % cap) * 8;
        p7.insert(p7.end(), r, r + 7);
    }
    return MF_OK;
}
extern "C" int mf_get_pose_log(mf_ctx* c, int32_t model, int64_t* ts, float* p7, uint32_t max_entries, uint32_t* count) {
    ModelState* m = model_at(c, model);
    if (!m || !count) return MF_EINVAL;
    std::vector<int64_t> t; std::vector<float> p;
    int rc = download_pose_log(c, *m, t, p);
    if (rc != MF_OK) return rc;
    *count = (uint32_t)t.size();
    if (ts && p7) {
        const size_t n = t.size() < max_entries ? t.size() : max_entries;
        memcpy(ts, t.data(), n * sizeof(int64_t));
        memcpy(p7, p.data(), n * 7 * sizeof(float));
    }
    return MF_OK;
}
static int write_pose_file(mf_ctx* c, const std::string& dir, int id, const std::vector<int64_t>& ts, const std::vector<float>& p) {
    const std::string fn = dir + "poses-" + std::to_string(id) + ".txt";
    FILE* f = fopen(fn.c_str(), "w");
    if (!f) { c->err = "cannot write " + fn; return MF_EINVAL; }
    for (size_t e = 0; e < ts.size(); ++e) {  // std::fixed << std::setprecision(6): timestamp [us] * 1e-6, then t and q (xyzw)
        fprintf(f, "%.6f", (double)ts[e] * 1e-6);
        for (int k = 0; k < 7; ++k) fprintf(f, " %.6f", (double)p[e * 7 + k]);
        fputc('\n', f);
    }
    fclose(f);
    return MF_OK;
}
// MaskFusion::exportPoses (Core/MaskFusion.cpp:851-879): poses-<id>.txt for live and dropped models
extern "C" int mf_export_poses(mf_ctx* c, const char* export_dir) {
    if (!c || !export_dir) return MF_EINVAL;
    const std::string dir(export_dir);
    for (auto& m : c->models) {
        std::vector<int64_t> t; std::vector<float> p;
        int rc = download_pose_log(c, *m, t, p);
        if (rc != MF_OK) return rc;
        if (!m->d_poselog) continue;
        rc = write_pose_file(c, dir, m->id, t, p);
        if (rc != MF_OK) return rc;
    }
    {
        int rc = materialise_retired(c);
        if (rc != MF_OK) return rc;
    }
    for (auto& r : c->retired) {
        int rc = write_pose_file(c, dir, r.id, r.ts, r.p);
        if (rc != MF_OK) return rc;
    }
    return MF_OK;
}
// MaskFusion::savePly (Core/MaskFusion.cpp:733-849): cloud-<id>.ply, binary little endian, model frame, normals negated,
// only surfels whose confidence exceeds the model's threshold
extern "C" int mf_save_ply(mf_ctx* c, const char* export_dir) {
    if (!c || !export_dir) return MF_EINVAL;
    int rc = mf_sync(c);
    if (rc != MF_OK) return rc;
    for (auto& m : c->models) {
        const uint32_t n = (uint32_t)*m->h_count;
        std::vector<float> data((size_t)n * 12);
        uint32_t got = 0;
        int idx = 0;
        for (size_t i = 0; i < c->models.size(); ++i) if (c->models[i].get() == m.get()) idx = (int)i;
        rc = mf_download_map(c, idx, data.data(), n, &got);
        if (rc != MF_OK) return rc;
        uint32_t valid = 0;
        for (uint32_t i = 0; i < got; ++i) valid += data[(size_t)i * 12 + 3] > m->confThr;
        const std::string fn = std::string(export_dir) + "cloud-" + std::to_string(m->id) + ".ply";
        FILE* f = fopen(fn.c_str(), "wb");
        if (!f) { c->err = "cannot write " + fn; return MF_EINVAL; }
        fprintf(f, "ply\nformat binary_little_endian 1.0\nelement vertex %u\nproperty float x\nproperty float y\nproperty float z"
                   "\nproperty uchar red\nproperty uchar green\nproperty uchar blue\nproperty float nx\nproperty float ny"
                   "\nproperty float nz\nproperty float radius\nend_header\n", valid);
        for (uint32_t i = 0; i < got; ++i) {
            const float* s12 = data.data() + (size_t)i * 12;
            if (!(s12[3] > m->confThr)) continue;
            fwrite(s12, sizeof(float), 3, f);
            const int col = (int)s12[4];
            const unsigned char rgb[3] = {(unsigned char)(col >> 16 & 0xFF), (unsigned char)(col >> 8 & 0xFF), (unsigned char)(col & 0xFF)};
            fwrite(rgb, 1, 3, f);
            const float nr[4] = {-s12[8], -s12[9], -s12[10], s12[11]};
            fwrite(nr, sizeof(float), 4, f);
        }
        fclose(f);
    }
    return MF_OK;
}
extern "C" int mf_get_track_stats(mf_ctx* c, int32_t model, float* out8) {
    ModelState* m = model_at(c, model);
    if (!m || !out8) return MF_EINVAL;
    int rc = mf_sync(c);
    if (rc != MF_OK) return rc;
    const PoseDev& p = *m->h_pose;
    out8[0] = p.lastICPError; out8[1] = p.lastICPCount; out8[2] = p.lastRGBError; out8[3] = p.lastRGBCount;
    out8[4] = p.lastSO3Error; out8[5] = p.lastSO3Count; out8[6] = (float)p.so3Iterations; out8[7] = (float)p.rejected;
    return MF_OK;
}
// how many Gauss-Newton iterations of the model's last geometric tracking step solved a system outside the solver's stated domain (fewer
// than 6 inliers, or a pivot below 1e-8 of the largest diagonal entry, i.e. cond(A) > 1e8): DESIGN.md finding F4
extern "C" int mf_get_gn_condition(mf_ctx* c, int32_t model, int32_t* ill_iterations) {
    ModelState* m = model_at(c, model);
    if (!m || !ill_iterations) return MF_EINVAL;
    int rc = mf_sync(c);
    if (rc != MF_OK) return rc;
    *ill_iterations = m->h_pose->illIterations;
    return MF_OK;
}
extern "C" int mf_get_last_fillin(mf_ctx* c, int32_t* used) {
    if (!c || !used) return MF_EINVAL;
    int rc = mf_sync(c);
    if (rc != MF_OK) return rc;
    // h_frame mirrors the state AFTER frame_advance: pad[0] is the decision the last tracking step ran with
    *used = c->models[0]->h_frame->pad[0];
    return MF_OK;
}
extern "C" int mf_download_segmentation(mf_ctx* c, uint8_t* out) {
    if (!c || !out) return MF_EINVAL;
    int rc = mf_sync(c);
    if (rc != MF_OK) return rc;
    MF_HIP(c, hipMemcpy(out, c->d_mask_tex, (size_t)c->P, hipMemcpyDeviceToHost));
    return MF_OK;
}

// MaskFusion.cpp:299-303: cv::threshold(fullSegmentation, out, 254, 255, THRESH_TOZERO_INV) (255 = ignored -> 0) then
// cv::imwrite(exportDir + "Segmentation<tick>.png").  An 8-bit greyscale PNG with stored (uncompressed) deflate blocks: any
// reader decodes it to the same pixels as OpenCV's file.
static uint32_t png_crc(uint32_t crc, const uint8_t* p, size_t n) {
    struct Table {
        uint32_t t[256];
        Table() {
            for (uint32_t i = 0; i < 256; ++i) {
                uint32_t v = i;
                for (int k = 0; k < 8; ++k) v = (v & 1) ? 0xEDB88320u ^ (v >> 1) : v >> 1;
                t[i] = v;
            }
        }
    };
    static const Table table;   // initialised once, thread-safely (contexts of different threads may export at the same time)
    for (size_t i = 0; i < n; ++i) crc = table.t[(crc ^ p[i]) & 255] ^ (crc >> 8);
    return crc;
}
static void png_chunk(std::vector<uint8_t>& f, const char* tag, const std::vector<uint8_t>& body) {
    auto be32 = [&](uint32_t v) { for (int k = 3; k >= 0; --k) f.push_back((uint8_t)(v >> (8 * k))); };
    be32((uint32_t)body.size());
    const size_t at = f.size();
    f.insert(f.end(), tag, tag + 4);
    f.insert(f.end(), body.begin(), body.end());
    be32(png_crc(0xFFFFFFFFu, f.data() + at, f.size() - at) ^ 0xFFFFFFFFu);
}
// cv::imwrite(path, CV_8UC1 image) stand-in: 8-bit greyscale PNG, stored deflate blocks.  HOST pointers, no GPU involved.
extern "C" int mf_write_png_gray8(const char* path, const uint8_t* img, int32_t W, int32_t H) {
    if (!path || !img || W <= 0 || H <= 0) return MF_EINVAL;
    std::vector<uint8_t> raw;   // filter byte 0 + row
    raw.reserve((size_t)(W + 1) * H);
    for (int y = 0; y < H; ++y) {
        raw.push_back(0);
        raw.insert(raw.end(), img + (size_t)y * W, img + (size_t)(y + 1) * W);
    }
    std::vector<uint8_t> z = {0x78, 0x01};
    uint32_t a = 1, b = 0;
    for (uint8_t v : raw) { a = (a + v) % 65521u; b = (b + a) % 65521u; }
    for (size_t off = 0; off < raw.size(); off += 65535) {
        const size_t n = std::min<size_t>(65535, raw.size() - off);
        z.push_back(off + n == raw.size() ? 1 : 0);
        z.push_back((uint8_t)(n & 255)); z.push_back((uint8_t)(n >> 8));
        z.push_back((uint8_t)(~n & 255)); z.push_back((uint8_t)((~n >> 8) & 255));
        z.insert(z.end(), raw.begin() + off, raw.begin() + off + n);
    }
    const uint32_t adler = (b << 16) | a;
    for (int k = 3; k >= 0; --k) z.push_back((uint8_t)(adler >> (8 * k)));
    std::vector<uint8_t> f = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    std::vector<uint8_t> hdr;
    for (uint32_t v : {(uint32_t)W, (uint32_t)H}) for (int k = 3; k >= 0; --k) hdr.push_back((uint8_t)(v >> (8 * k)));
    hdr.insert(hdr.end(), {8, 0, 0, 0, 0});   // 8-bit, greyscale, deflate, no filter, no interlace
    png_chunk(f, "IHDR", hdr);
    png_chunk(f, "IDAT", z);
    png_chunk(f, "IEND", {});
    FILE* fp = fopen(path, "wb");
    if (!fp) return MF_EINVAL;
    const bool ok = fwrite(f.data(), 1, f.size(), fp) == f.size();
    fclose(fp);
    return ok ? MF_OK : MF_EINVAL;
}
extern "C" int mf_export_segmentation_png(mf_ctx* c, const char* path) {
    if (!c || !path) return MF_EINVAL;
    std::vector<uint8_t> img((size_t)c->P);
    int rc = mf_download_segmentation(c, img.data());
    if (rc != MF_OK) return rc;
    for (auto& v : img) v = v > 254 ? 0 : v;   // cv::threshold(..., 254, 255, THRESH_TOZERO_INV): 255 (ignored) -> 0
    rc = mf_write_png_gray8(path, img.data(), c->W, c->H);
    if (rc != MF_OK) c->err = std::string("cannot write ") + path;
    return rc;
}

extern "C" int mf_download_map(mf_ctx* c, int32_t model, float* out, uint32_t max_count, uint32_t* count) {
    ModelState* ms = model_at(c, model);
    if (!ms || !out || !count) return MF_EINVAL;
    int rc = mf_sync(c);
    if (rc != MF_OK) return rc;
    const uint32_t n = (uint32_t)*ms->h_count;
    *count = n;
    const uint32_t m = n < max_count ? n : max_count;
    if (m == 0) return MF_OK;
    std::vector<float4> a(m), b(m), d(m);
    const Surfels& s = ms->surf[ms->cur];
    MF_HIP(c, hipMemcpy(a.data(), s.pc, m * sizeof(float4), hipMemcpyDeviceToHost));
    MF_HIP(c, hipMemcpy(b.data(), s.ct, m * sizeof(float4), hipMemcpyDeviceToHost));
    MF_HIP(c, hipMemcpy(d.data(), s.nr, m * sizeof(float4), hipMemcpyDeviceToHost));
    for (uint32_t i = 0; i < m; ++i) {
        memcpy(out + (size_t)i * 12, &a[i], 16);
        memcpy(out + (size_t)i * 12 + 4, &b[i], 16);
        memcpy(out + (size_t)i * 12 + 8, &d[i], 16);
    }
    return MF_OK;
}

// ------------------------------------------------------------------------------------------------
// parameters
// ------------------------------------------------------------------------------------------------
struct ParamRef { const char* key; int kind; size_t off; };  // kind 0 float, 1 int
static const ParamRef kParams[] = {
    {"depthCutoff", 0, offsetof(mf_config, depth_cutoff)},
    {"icpWeight", 0, offsetof(mf_config, icp_weight)},
    {"outlierCoefficient", 0, offsetof(mf_config, outlier_coefficient)},
    {"maxDepthProcessed", 0, offsetof(mf_config, max_depth_processed)},
    {"fastOdom", 1, offsetof(mf_config, fast_odom)},
    {"so3", 1, offsetof(mf_config, so3)},
    {"rgbOnly", 1, offsetof(mf_config, rgb_only)},
    {"pyramid", 1, offsetof(mf_config, pyramid)},
    {"timeDelta", 1, offsetof(mf_config, time_delta)},
    {"enableMultipleModels", 1, offsetof(mf_config, enable_multiple_models)},
    {"trackAllModels", 1, offsetof(mf_config, track_all_models)},
    {"modelSpawnOffset", 1, offsetof(mf_config, model_spawn_offset)},
};
struct SegRef { const char* key; int kind; size_t off; };  // kind 0 float, 1 int
static const SegRef kSegParams[] = {
    {"mfThreshold", 0, offsetof(SegParams, threshold)},
    {"mfWeightDistance", 0, offsetof(SegParams, weightDistance)},
    {"mfWeightConvexity", 0, offsetof(SegParams, weightConvexity)},
    {"mfMorphEdgeIterations", 1, offsetof(SegParams, morphEdgeIterations)},
    {"mfMorphEdgeRadius", 1, offsetof(SegParams, morphEdgeRadius)},
    {"mfMorphMaskIterations", 1, offsetof(SegParams, morphMaskIterations)},
    {"mfMorphMaskRadius", 1, offsetof(SegParams, morphMaskRadius)},
    {"newModelMinRelativeSize", 0, offsetof(SegParams, minRelSizeNew)},
    {"newModelMaxRelativeSize", 0, offsetof(SegParams, maxRelSizeNew)},
};

extern "C" int mf_set_param(mf_ctx* c, const char* key, double value) {
    if (!c || !key) return MF_EINVAL;
    if (!strcmp(key, "hostProfileReset")) { for (double& v : c->host_us) v = 0; c->host_calls = 0; return MF_OK; }
    if (!strcmp(key, "timings")) { c->timings_on = value != 0; return MF_OK; }
    if (!strcmp(key, "icpProfile")) { c->icp_prof_on = value != 0; return MF_OK; }
    if (!strcmp(key, "hostInputAsync")) {   // 0: mf_process_frame blocks until the frame is fused (rounds 1-3); 1: returns when it is enqueued
        if (hipStreamSynchronize(c->stream) != hipSuccess || hipStreamSynchronize(c->stream_in) != hipSuccess) return MF_EHIP;
        c->host_async = value != 0; return MF_OK;
    }
    if (!strcmp(key, "hostLockstep")) { c->host_lockstep = value != 0; return MF_OK; }
    if (!strcmp(key, "hostWaitUpload")) { c->host_wait_upload = value != 0; return MF_OK; }
    if (!strcmp(key, "splatProfile")) {   // per-tile shader-clock stamps of k_splat_tile (tools/splat_prof.py); debug tap "splat_prof"
        if (value != 0 && !c->d_splat_prof) {
            if (hipMalloc(&c->d_splat_prof, splat_tiles_scratch_ints(c->W, c->H) * 8 * sizeof(unsigned long long)) != hipSuccess) return MF_ENOMEM;
        }
        c->splat_prof_on = value != 0;
        return MF_OK;
    }
    if (!strcmp(key, "gpuLabels")) { c->gpu_labels = value != 0; return MF_OK; }   // 0: host label stage (specification)
    if (!strcmp(key, "splatTiles")) { c->splat_tiles = value != 0; return MF_OK; }
    if (!strcmp(key, "splatTileEntries")) {   // test knob: shrink the tile lists (never beyond what was allocated) to force the overflow path
        const long long v = (long long)value;
        if (v < 1 || v > c->tile_entries_alloc) return MF_EINVAL;
        c->tile_entries_cap = (int)v;
        return MF_OK;
    }
    if (!strcmp(key, "batchTracking")) { c->batch_tracking = value != 0; return MF_OK; }
    if (!strcmp(key, "frameToFrameRGB")) { c->ftf_rgb = value != 0; return MF_OK; }         // MaskFusion::setFrameToFrameRGB (Core/MaskFusion.cpp:910)
    if (!strcmp(key, "batchObjectPasses")) { c->batch_objects = value != 0; return MF_OK; }  // 0: the object models' surfel passes model by model
    if (!strcmp(key, "objectBoundingBoxLimit")) { c->bbox_limit = value != 0; return MF_OK; }   // 0: a headless upstream that never renders (bb_max_z = FLT_MAX)
    if (!strcmp(key, "literalFusionWeight")) {
        c->weight_literal = value != 0;
        for (auto& m : c->models) hipLaunchKernelGGL(k_set_weight_literal, dim3(1), dim3(64), 0, c->stream, m->d_pose, c->weight_literal ? 1 : 0, m->h_pose);
        for (auto& m : c->pool) hipLaunchKernelGGL(k_set_weight_literal, dim3(1), dim3(64), 0, c->stream, m->d_pose, c->weight_literal ? 1 : 0, m->h_pose);
        return check_launch(c);
    }
    if (!strcmp(key, "objectSmallGrids")) { c->object_small_grids = value != 0; return MF_OK; }
    if (!strcmp(key, "objectScatterSplat")) { c->object_scatter_splat = value != 0; return MF_OK; }
    if (!strcmp(key, "globalTiles")) { c->global_tiles = value != 0; return MF_OK; }
    if (!strcmp(key, "cullRuns")) { c->cull_runs = value != 0; c->vis_tag.model = nullptr; return MF_OK; }
    if (!strcmp(key, "cullMinSurfels")) { c->cull_min_surfels = (int)value; c->vis_tag.model = nullptr; return MF_OK; }
    if (!strcmp(key, "rebuildRunTable")) {   // tooling: the background's run table from scratch (what an upload / Model::initialise does)
        launch_run_table(c->models[0]->surf[c->models[0]->cur], c->models[0]->d_frame, c->stream);
        c->vis_tag.model = nullptr;
        return check_launch(c);
    }
    if (!strcmp(key, "cleanLiteralWindow")) { c->clean_literal = value != 0; return MF_OK; }   // 0: the exact-arithmetic 4 x 4 window
    if (!strcmp(key, "earlyBackgroundFusion")) { c->early_bg_fusion = value != 0; return MF_OK; }
    if (!strcmp(key, "modelApiPackedIndex")) { c->model_api_packed = value != 0; return MF_OK; }   // 0: scatter + resolve form (specification)
    if (!strcmp(key, "tileThreads")) { c->splat_tune.tile_threads = (int)value; return MF_OK; }   // A/B: threads per tile workgroup of the tile passes
    if (!strcmp(key, "spriteLanes")) { c->splat_tune.sprite_lanes = (int)value; return MF_OK; }   // A/B: lanes per sprite in the tile z-test (default 4)
    if (!strcmp(key, "overlapPreprocessing")) {
        (void)hipStreamSynchronize(c->stream_pre);
        (void)hipStreamSynchronize(c->stream);
        c->overlap = value != 0;
        return MF_OK;
    }
    if (!strcmp(key, "confidenceThreshold")) { c->cfg.conf_global = (float)value; c->models[0]->confThr = (float)value; return MF_OK; }
    for (const ParamRef& p : kParams)
        if (!strcmp(key, p.key)) {
            char* base = reinterpret_cast<char*>(&c->cfg);
            if (p.kind == 0) *reinterpret_cast<float*>(base + p.off) = (float)value;
            else *reinterpret_cast<int32_t*>(base + p.off) = (int32_t)value;
            return MF_OK;
        }
    for (const SegRef& p : kSegParams)
        if (!strcmp(key, p.key)) {
            char* base = reinterpret_cast<char*>(&c->seg);
            if (p.kind == 0) *reinterpret_cast<float*>(base + p.off) = (float)value;
            else *reinterpret_cast<int*>(base + p.off) = (int)value;
            return MF_OK;
        }
    c->err = std::string("unknown parameter: ") + key;
    return MF_EINVAL;
}
extern "C" int mf_get_param(mf_ctx* c, const char* key, double* value) {
    if (!c || !key || !value) return MF_EINVAL;
    if (!strcmp(key, "confidenceThreshold")) { *value = c->models[0]->confThr; return MF_OK; }
    if (!strcmp(key, "splatTileEntries")) { *value = c->tile_entries_cap; return MF_OK; }
    if (!strcmp(key, "cullRuns")) { *value = c->cull_runs ? 1 : 0; return MF_OK; }
    if (!strcmp(key, "visibleRuns") || !strcmp(key, "backgroundRuns")) {   // test taps: size of the last visibility list / of the background's run table
        MF_HIP(c, hipStreamSynchronize(c->stream));
        int v = 0;
        if (!strcmp(key, "visibleRuns")) MF_HIP(c, hipMemcpy(&v, c->d_vis_count, sizeof(int), hipMemcpyDeviceToHost));
        else { FrameDev f; MF_HIP(c, hipMemcpy(&f, c->models[0]->d_frame, sizeof(FrameDev), hipMemcpyDeviceToHost)); v = f.runs; }
        *value = v;
        return MF_OK;
    }
    // mean host microseconds per mf_process_frame call since the context was created: staging copy | upload enqueue | frame enqueue | whole call
    if (!strcmp(key, "hostStageUs")) { *value = c->host_calls ? c->host_us[0] / c->host_calls : 0; return MF_OK; }
    if (!strcmp(key, "hostUploadUs")) { *value = c->host_calls ? c->host_us[1] / c->host_calls : 0; return MF_OK; }
    if (!strcmp(key, "hostEnqueueUs")) { *value = c->host_calls ? c->host_us[2] / c->host_calls : 0; return MF_OK; }
    if (!strcmp(key, "hostCallUs")) { *value = c->host_calls ? c->host_us[3] / c->host_calls : 0; return MF_OK; }
    if (!strcmp(key, "hostWaitUs")) { *value = c->host_calls ? c->host_us[4] / c->host_calls : 0; return MF_OK; }
    if (!strcmp(key, "frameToFrameRGB")) { *value = c->ftf_rgb ? 1 : 0; return MF_OK; }
    if (!strcmp(key, "objectBoundingBoxLimit")) { *value = c->bbox_limit ? 1 : 0; return MF_OK; }
    for (const ParamRef& p : kParams)
        if (!strcmp(key, p.key)) {
            const char* base = reinterpret_cast<const char*>(&c->cfg);
            *value = p.kind == 0 ? (double)*reinterpret_cast<const float*>(base + p.off) : (double)*reinterpret_cast<const int32_t*>(base + p.off);
            return MF_OK;
        }
    for (const SegRef& p : kSegParams)
        if (!strcmp(key, p.key)) {
            const char* base = reinterpret_cast<const char*>(&c->seg);
            *value = p.kind == 0 ? (double)*reinterpret_cast<const float*>(base + p.off) : (double)*reinterpret_cast<const int*>(base + p.off);
            return MF_OK;
        }
    return MF_EINVAL;
}

extern "C" int mf_get_timings(mf_ctx* c, float* ms) {
    if (!c || !ms) return MF_EINVAL;
    int rc = mf_sync(c);
    if (rc != MF_OK) return rc;
    memcpy(ms, c->last_ms, sizeof(c->last_ms));
    return MF_OK;
}
extern "C" void* mf_get_stream(mf_ctx* c) { return c ? (void*)c->stream : nullptr; }
extern "C" void* mf_get_input_stream(mf_ctx* c) { return c ? (void*)(c->overlap ? c->stream_pre : c->stream) : nullptr; }

static int debug_read_impl(mf_ctx* c, ModelState& mdl, const char* what, void* out, uint64_t out_bytes) {
    int rc = mf_sync(c);
    if (rc != MF_OK) return rc;
    const void* src = nullptr;
    size_t bytes = 0;
    bool variable = false;   // taps whose meaningful length depends on the frame: copy as much as the caller asked for
    const size_t P = (size_t)c->P;
    std::string w(what);
    auto lvl = [&](const std::string& pre, float* const arr[3]) -> bool {
        for (int i = 0; i < 3; ++i)
            if (w == pre + std::to_string(i)) {
                src = arr[i]; bytes = (size_t)(c->W >> i) * (c->H >> i) * 3 * sizeof(float);
                return true;
            }
        return false;
    };
    const int lastSet = (int)((c->frame_no + 1) & 1);  // buffer set of the last processed / staged frame
    if (w == "depthF") { src = c->d_depthF[c->lastF]; bytes = P * 4; }
    else if (lvl("vmap_g", mdl.d_vmap_g) || lvl("nmap_g", mdl.d_nmap_g) || lvl("vmap", c->d_vmap[lastSet]) || lvl("nmap", c->d_nmap[lastSet])) {}
    else if (w == "pred_vertex") { src = mdl.d_predV; bytes = P * 16; }
    else if (w == "pred_normal") { src = mdl.d_predN; bytes = P * 16; }
    else if (w == "pred_image") { src = mdl.d_predImage; bytes = P * 4; }
    else if (w == "pred_time") { src = mdl.d_predTime; bytes = P * 2; }
    else if (w == "index") { src = c->d_index; bytes = P * 4; }
    else if (w == "index_vc") { src = c->d_ivc; bytes = P * 16; }
    else if (w == "index_nr") { src = c->d_inr; bytes = P * 16; }
    else if (w == "index_ct") { src = c->d_ict; bytes = P * 16; }
    else if (w == "index_packed") { src = c->d_iclean; bytes = P * 32; }
    else if (w == "cand_op") { src = c->d_cand_op; bytes = P; variable = true; }
    else if (w == "cand_rec") { src = c->d_cand_rec; bytes = P * 48; variable = true; }
    else if (w == "clean_flags") { src = c->d_flags; bytes = (size_t)c->cap_max + P; variable = true; }
    else if (w == "clean_newconf") { src = c->d_newconf; bytes = ((size_t)c->cap_max + P) * 4; variable = true; }
    else if (w == "icp_log") { src = mdl.d_icp_log; bytes = 19 * 32 * 4; }
    else if (w == "gn_trace") { src = mdl.d_gn_trace; bytes = 20 * kGnTraceRow * 8; }
    else if (w == "icp_prof") { src = c->d_icp_prof; bytes = 19 * 16 * 8; }
    else if (w == "splat_prof") {
        if (!c->d_splat_prof) { c->err = "splat_prof: switch splatProfile on first"; return MF_ESTATE; }
        src = c->d_splat_prof; bytes = splat_tiles_scratch_ints(c->W, c->H) * 8 * 8;
    }
    else if (w == "edge_map") { src = c->d_edge; bytes = P * 4; }
    else if (w == "edge_binary") { src = c->d_bin; bytes = P; }
    else if (w == "projected_ids") { src = c->d_proj_ids; bytes = P; }
    else { c->err = "unknown debug tap: " + w; return MF_EINVAL; }
    if (variable) bytes = out_bytes < bytes ? (size_t)out_bytes : bytes;
    if (out_bytes < bytes) { c->err = "debug_read: buffer too small"; return MF_EINVAL; }
    MF_HIP(c, hipMemcpy(out, src, bytes, hipMemcpyDeviceToHost));
    // "index" / "index_vc" / "index_nr" / "index_ct" are the row-major images of the pre-fusion index pass (and of
    // mf_model_predict_indices); inside mf_process_frame the post-fusion pass that feeds clean() lives in "index_packed"
    // (column-major texel order, two float4 per texel: {vertConf | initTime, lastTime, index bits, 0})
    return MF_OK;
}
extern "C" int mf_debug_read(mf_ctx* c, const char* what, void* out, uint64_t out_bytes) {
    if (!c || !what || !out) return MF_EINVAL;
    return debug_read_impl(c, *c->models[0], what, out, out_bytes);
}
extern "C" int mf_debug_read_model(mf_ctx* c, int32_t model, const char* what, void* out, uint64_t out_bytes) {
    ModelState* m = model_at(c, model);
    if (!m || !what || !out) return MF_EINVAL;
    return debug_read_impl(c, *m, what, out, out_bytes);
}

// ------------------------------------------------------------------------------------------------
// kernel-level entry points
// ------------------------------------------------------------------------------------------------
static int launch_rc() { return hipGetLastError() == hipSuccess ? MF_OK : MF_EHIP; }

extern "C" int mf_k_bilateral(const float* d_depth, float* d_out, int32_t W, int32_t H, void* stream) {
    if (!d_depth || !d_out || W <= 0 || H <= 0) return MF_EINVAL;
    launch_bilateral(d_depth, d_out, W, H, (hipStream_t)stream);
    return launch_rc();
}
extern "C" int mf_k_pyrdown_f(const float* d_src, float* d_dst, int32_t sw, int32_t sh, void* stream) {
    if (!d_src || !d_dst || sw < 2 || sh < 2) return MF_EINVAL;
    launch_pyrdown_f(d_src, d_dst, sw, sh, (hipStream_t)stream);
    return launch_rc();
}
extern "C" int mf_k_intensity(const uint8_t* d_img, int32_t channels, uint8_t* d_out, int32_t n, void* stream) {
    if (!d_img || !d_out || n <= 0 || (channels != 3 && channels != 4)) return MF_EINVAL;
    launch_intensity(d_img, channels, d_out, n, (hipStream_t)stream);
    return launch_rc();
}
extern "C" int mf_k_pyrdown_u8(const uint8_t* d_src, uint8_t* d_dst, int32_t sw, int32_t sh, void* stream) {
    if (!d_src || !d_dst || sw < 2 || sh < 2) return MF_EINVAL;
    launch_pyrdown_u8(d_src, d_dst, sw, sh, (hipStream_t)stream);
    return launch_rc();
}
extern "C" int mf_k_derivative_images(const uint8_t* d_src, int16_t* d_dx, int16_t* d_dy, int32_t W, int32_t H, void* stream) {
    if (!d_src || !d_dx || !d_dy || W <= 0 || H <= 0) return MF_EINVAL;
    launch_derivative(d_src, d_dx, d_dy, W, H, 0.f, nullptr, (hipStream_t)stream);
    return launch_rc();
}
extern "C" int mf_k_so3_prealign(const uint8_t* d_last, const uint8_t* d_next, int32_t W, int32_t H, float fx, float fy, float cx, float cy,
                                 double* R9, float* stats3, void* stream) {
    if (!d_last || !d_next || !R9 || !stats3 || W < 3 || H < 3) return MF_EINVAL;
    char* buf = nullptr;
    const size_t sb = so3_scratch_bytes(W, H);
    if (hipMalloc((void**)&buf, sb + sizeof(So3Result)) != hipSuccess) return MF_ENOMEM;
    So3Result* d = reinterpret_cast<So3Result*>(buf);
    if (launch_so3_prealign(d_last, d_next, W, H, Intr{fx, fy, cx, cy}, d, buf + sizeof(So3Result), (hipStream_t)stream) != 0) {
        (void)hipFree(buf);
        return MF_EINVAL;
    }
    So3Result h;
    (void)hipStreamSynchronize((hipStream_t)stream);
    const hipError_t e = hipMemcpy(&h, d, sizeof(h), hipMemcpyDeviceToHost);
    (void)hipFree(buf);
    if (e != hipSuccess) return MF_EHIP;
    memcpy(R9, h.R, sizeof(h.R));
    stats3[0] = h.error; stats3[1] = h.count; stats3[2] = (float)h.iterations;
    return MF_OK;
}
static RgbLevel make_level(const int16_t* dIdx, const int16_t* dIdy, const float* lastDepth, const float* nextDepth, const uint8_t* lastImage,
                           const uint8_t* nextImage, int W, int H, float minScale, float maxDepthDelta) {
    RgbLevel L;
    L.dIdx = dIdx; L.dIdy = dIdy; L.lastDepth = lastDepth; L.nextDepth = nextDepth; L.lastImage = lastImage; L.nextImage = nextImage;
    L.W = W; L.H = H; L.minScale = minScale; L.maxDepthDelta = maxDepthDelta; L.gate = nullptr;
    return L;
}
extern "C" int mf_k_rgb_residual(float min_scale, const int16_t* d_dIdx, const int16_t* d_dIdy, const float* d_last_depth,
                                 const float* d_next_depth, const uint8_t* d_last_image, const uint8_t* d_next_image, float max_depth_delta,
                                 const float* kt3, const float* krkinv9, int32_t W, int32_t H, void* d_corres, int32_t* count_sigma2,
                                 void* stream) {
    if (!d_dIdx || !d_dIdy || !d_last_depth || !d_next_depth || !d_last_image || !d_next_image || !kt3 || !krkinv9 || !d_corres ||
        !count_sigma2)
        return MF_EINVAL;
    float h[12];
    memcpy(h, krkinv9, 36); memcpy(h + 9, kt3, 12);
    char* scratch = nullptr;
    if (hipMalloc((void**)&scratch, 64) != hipSuccess) return MF_ENOMEM;
    float* d_k = reinterpret_cast<float*>(scratch);
    int* d_sums = reinterpret_cast<int*>(scratch + 48);
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipMemcpyAsync(d_k, h, 48, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemsetAsync(d_sums, 0, 8, s);
    if (e == hipSuccess) {
        launch_rgb_residual_only(make_level(d_dIdx, d_dIdy, d_last_depth, d_next_depth, d_last_image, d_next_image, W, H, min_scale,
                                            max_depth_delta), d_k, reinterpret_cast<RgbCorr*>(d_corres), d_sums, s);
        e = hipMemcpyAsync(count_sigma2, d_sums, 8, hipMemcpyDeviceToHost, s);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(scratch);
    return e == hipSuccess ? MF_OK : MF_EHIP;
}
extern "C" int mf_k_rgb_step(const void* d_corres, float sigma, const float* d_last_depth, float fx, float fy, float cx, float cy,
                             const int16_t* d_dIdx, const int16_t* d_dIdy, float sobel_scale, int32_t W, int32_t H, double* out32,
                             void* stream) {
    if (!d_corres || !d_last_depth || !d_dIdx || !d_dIdy || !out32) return MF_EINVAL;
    double* d_out = nullptr;
    if (hipMalloc((void**)&d_out, 32 * sizeof(double)) != hipSuccess) return MF_ENOMEM;
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(d_out, 0, 32 * sizeof(double), s);
    if (e == hipSuccess) {
        launch_rgb_step_only(make_level(d_dIdx, d_dIdy, d_last_depth, d_last_depth, nullptr, nullptr, W, H, 0.f, 0.f),
                             reinterpret_cast<const RgbCorr*>(d_corres), sigma, Intr{fx, fy, cx, cy}, sobel_scale, d_out, s);
        e = hipMemcpyAsync(out32, d_out, 32 * sizeof(double), hipMemcpyDeviceToHost, s);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(d_out);
    return e == hipSuccess ? MF_OK : MF_EHIP;
}
extern "C" int mf_k_vmap_nmap(const float* d_depth, float* d_vmap, float* d_nmap, int32_t W, int32_t H, float fx, float fy, float cx,
                              float cy, float depth_cutoff, void* stream) {
    if (!d_depth || !d_vmap || !d_nmap || W <= 0 || H <= 0) return MF_EINVAL;
    launch_vmap_nmap(d_depth, d_vmap, d_nmap, W, H, Intr{fx, fy, cx, cy}, depth_cutoff, (hipStream_t)stream);
    return launch_rc();
}
extern "C" int mf_k_model_pyramid(const float* d_v4, const float* d_n4, const float* R9, const float* t3, float* d_vmaps,
                                  float* d_nmaps, int32_t W, int32_t H, void* stream) {
    if (!d_v4 || !d_n4 || !R9 || !t3 || !d_vmaps || !d_nmaps || W % 4 || H % 4) return MF_EINVAL;
    float Rt[12];
    memcpy(Rt, R9, 36); memcpy(Rt + 9, t3, 12);
    float* vm[3]; float* nm[3];
    size_t off = 0;
    for (int i = 0; i < 3; ++i) {
        vm[i] = d_vmaps + off; nm[i] = d_nmaps + off;
        off += (size_t)(W >> i) * (H >> i) * 3;
    }
    launch_model_pyramid((const float4*)d_v4, (const float4*)d_n4, nullptr, nullptr, nullptr, Rt, vm, nm, W, H, Intr{1, 1, 0, 0},
                         (hipStream_t)stream);
    return launch_rc();
}
extern "C" int mf_k_geometric_edges(const float* d_vmap, const float* d_nmap, float* d_edge, uint8_t* d_binary, uint8_t* d_tmp, int32_t W,
                                    int32_t H, float w_distance, float w_convexity, float threshold, int32_t morph_radius,
                                    int32_t morph_iterations, void* stream) {
    if (!d_vmap || !d_nmap || !d_edge || !d_binary || !d_tmp || W <= 2 || H <= 2) return MF_EINVAL;
    launch_edge_map(d_vmap, d_nmap, d_edge, W, H, w_distance, w_convexity, (hipStream_t)stream);
    launch_edge_binary(d_edge, d_binary, d_tmp, W, H, threshold, morph_radius, morph_iterations, (hipStream_t)stream);
    return launch_rc();
}
extern "C" int mf_segmentation_labels(int32_t W, int32_t H, const uint8_t* binary, const float* depth, const uint8_t* mask,
                                      const int32_t* class_ids, int32_t n_masks, const uint8_t* projected_ids, const int32_t* model_ids,
                                      const int32_t* model_class_ids, int32_t n_models, int32_t next_model_id, int32_t allow_new,
                                      const float* p, uint8_t* ignore_map, uint8_t* full, int32_t* has_new, int32_t* new_class) {
    if (!binary || !depth || !projected_ids || !model_ids || !model_class_ids || n_models < 1 || !p || !ignore_map || !full || !has_new ||
        !new_class || W <= 2 || H <= 2 || (n_masks > 0 && (!mask || !class_ids)))
        return MF_EINVAL;
    SegParams prm;
    prm.threshold = p[0]; prm.weightDistance = p[1]; prm.weightConvexity = p[2];
    prm.morphEdgeIterations = (int)p[3]; prm.morphEdgeRadius = (int)p[4]; prm.morphMaskIterations = (int)p[5]; prm.morphMaskRadius = (int)p[6];
    prm.removeEdges = p[7] != 0.f; prm.minRelSizeNew = p[8]; prm.maxRelSizeNew = p[9]; prm.personClassID = (int)p[10];
    std::vector<SegModelInfo> infos;
    for (int i = 0; i < n_models; ++i) infos.push_back(SegModelInfo{model_ids[i], model_class_ids[i]});
    std::vector<uint8_t> ign(ignore_map, ignore_map + (size_t)W * H);
    SegResult res;
    static const int32_t kNoClass[1] = {0};
    segmentation_host(prm, W, H, binary, depth, mask, n_masks > 0 ? class_ids : kNoClass, n_masks, projected_ids, infos, next_model_id,
                      allow_new != 0, ign, full, res);
    memcpy(ignore_map, ign.data(), ign.size());
    *has_new = res.hasNewLabel ? 1 : 0;
    *new_class = res.newClassID;
    return MF_OK;
}

// Device twin of mf_segmentation_labels (same arguments, HOST pointers; the images are staged to the device, the stage runs
// in mf_labels_gpu.hip, the outputs come back) -- the parity tests run both against the oracle.
static __global__ void k_alive_pose(PoseDev* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p->alive = 1; }
extern "C" int mf_k_segmentation_labels(int32_t W, int32_t H, const uint8_t* binary, const float* depth, const uint8_t* mask,
                                        const int32_t* class_ids, int32_t n_masks, const uint8_t* projected_ids, const int32_t* model_ids,
                                        const int32_t* model_class_ids, int32_t n_models, int32_t next_model_id, int32_t allow_new,
                                        const float* p, uint8_t* ignore_map, uint8_t* full, int32_t* has_new, int32_t* new_class) {
    if (!binary || !depth || !projected_ids || !model_ids || !model_class_ids || n_models < 1 || n_models > 64 || !p || !ignore_map ||
        !full || !has_new || !new_class || W <= 2 || H <= 2 || (n_masks > 0 && (!mask || !class_ids)) || n_masks > 256)
        return MF_EINVAL;
    SegParams prm;
    prm.threshold = p[0]; prm.weightDistance = p[1]; prm.weightConvexity = p[2];
    prm.morphEdgeIterations = (int)p[3]; prm.morphEdgeRadius = (int)p[4]; prm.morphMaskIterations = (int)p[5]; prm.morphMaskRadius = (int)p[6];
    prm.removeEdges = p[7] != 0.f; prm.minRelSizeNew = p[8]; prm.maxRelSizeNew = p[9]; prm.personClassID = (int)p[10];
    const size_t P = (size_t)W * H;
    LabelsScratch sc;
    if (!sc.init((int)P)) return MF_ENOMEM;
    uint8_t *d_bin = nullptr, *d_mask = nullptr, *d_proj = nullptr, *d_full = nullptr; float* d_depth = nullptr; PoseDev* d_pose = nullptr;
    if (!sc.dalloc(&d_bin, P) || !sc.dalloc(&d_mask, P) || !sc.dalloc(&d_proj, P) || !sc.dalloc(&d_full, P) || !sc.dalloc(&d_depth, P) ||
        !sc.dalloc(&d_pose, 1))
        return MF_ENOMEM;
    hipStream_t s = nullptr;
    hipError_t e = hipMemcpy(d_bin, binary, P, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_proj, projected_ids, P, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_depth, depth, P * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess && n_masks > 0) e = hipMemcpy(d_mask, mask, P, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(sc.ignoreMap, ignore_map, P, hipMemcpyHostToDevice);
    if (e != hipSuccess) return MF_EHIP;
    hipLaunchKernelGGL(k_alive_pose, dim3(1), dim3(64), 0, s, d_pose);
    std::vector<SegModelInfo> infos;
    std::vector<const PoseDev*> poses;
    for (int i = 0; i < n_models; ++i) { infos.push_back(SegModelInfo{model_ids[i], model_class_ids[i]}); poses.push_back(d_pose); }
    static const int32_t kNoClass[1] = {0};
    int rc = sc.enqueue(prm, W, H, d_bin, d_depth, n_masks > 0 ? d_mask : nullptr, n_masks > 0 ? class_ids : kNoClass, n_masks, d_proj, infos,
                        poses, next_model_id, allow_new != 0, d_full, s);
    if (rc != MF_OK) return rc;
    if (hipDeviceSynchronize() != hipSuccess) return MF_EHIP;
    if (sc.h_result[2]) return MF_ESTATE;
    if (hipMemcpy(full, d_full, P, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(ignore_map, sc.ignoreMap, P, hipMemcpyDeviceToHost) != hipSuccess)
        return MF_EHIP;
    *has_new = sc.h_result[0];
    *new_class = sc.h_result[1];
    return MF_OK;
}

extern "C" int mf_k_gn_solve(const double* sys29, const double* result_rt16, const float* Rprev9, const float* tprev3, double* x6_serial,
                             double* x6_wave, double* result_rt16_out, float* Rcurr9, float* tcurr3, float* stats2, void* stream) {
    if (!sys29 || !result_rt16 || !Rprev9 || !tprev3 || !x6_serial || !x6_wave || !result_rt16_out || !Rcurr9 || !tcurr3 || !stats2) return MF_EINVAL;
    const int rc = gn_solve_standalone(sys29, result_rt16, Rprev9, tprev3, x6_serial, x6_wave, result_rt16_out, Rcurr9, tcurr3, stats2, (hipStream_t)stream);
    return rc == 0 ? MF_OK : (rc == -1 ? MF_ENOMEM : MF_EHIP);
}
extern "C" int mf_k_icp_step(const float* Rcurr9, const float* tcurr3, const float* d_vc, const float* d_nc, const float* Rpi9,
                             const float* tprev3, float fx, float fy, float cx, float cy, const float* d_vp, const float* d_np,
                             float dist_thresh, float angle_thresh, int32_t W, int32_t H, float* d_out32, void* stream) {
    if (!Rcurr9 || !tcurr3 || !d_vc || !d_nc || !Rpi9 || !tprev3 || !d_vp || !d_np || !d_out32 || (W * H) % 4) return MF_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    float* scratch = nullptr;
    const size_t nb = (size_t)icp_geo_grid_blocks(W, H);
    const size_t bytes = nb * kIcpSlots * sizeof(float) + 2 * sizeof(GNState) + 24 * sizeof(float) + 64;
    if (hipMalloc((void**)&scratch, bytes) != hipSuccess) return MF_ENOMEM;
    char* base = reinterpret_cast<char*>(scratch);
    size_t o = nb * kIcpSlots * sizeof(float);
    o = (o + 15) & ~(size_t)15;
    GNState* st = reinterpret_cast<GNState*>(base + o);
    float* dpose = reinterpret_cast<float*>(base + o + 2 * sizeof(GNState));
    float hp[24];
    memcpy(hp, Rcurr9, 36); memcpy(hp + 9, tcurr3, 12); memcpy(hp + 12, Rpi9, 36); memcpy(hp + 21, tprev3, 12);
    (void)hipMemcpyAsync(dpose, hp, sizeof(hp), hipMemcpyHostToDevice, s);
    (void)hipStreamSynchronize(s);  // hp is a stack buffer
    launch_icp_step_standalone(dpose, dpose + 9, d_vc, d_nc, dpose + 12, dpose + 21, Intr{fx, fy, cx, cy}, d_vp, d_np, dist_thresh,
                               angle_thresh, W, H, scratch, st, d_out32, s);
    (void)hipStreamSynchronize(s);
    (void)hipFree(scratch);
    return launch_rc();
}

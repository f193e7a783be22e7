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
    bool table_valid = false;              // the live buffer has a run table (Surfels::box; device: frame->runs > 0): written by initialise / upload / launch_run_table,
                                           // maintained by the in-place clean; the two-launch clean leaves a dense buffer without one
    bool sparse = false;                   // ... and the buffer may have unused slots inside its runs: only the run-aware passes can read it (densify() first)
    long phys_ub = -1, runs_ub = -1;       // upper bounds of frame->phys / frame->runs (the device appends up to P / 4 surfels in up to P / (4 kRun) + 1 runs per
                                           // frame): the host compacts the buffer before the slots behind the last run or the table could run out; -1: unknown
    unsigned long long* h_append = nullptr;   // pinned: append_mirror() of the last in-place clean pass that has RUN (mf_internal.h)
    unsigned clean_seq = 0, mirror_from = 1;  // in-place clean passes enqueued so far on this model / the first one whose mirror describes the present buffer
    unsigned gen = 0;                      // bumped whenever the buffer, its table, the pose or the tick may have changed (visibility lists are cached against it)
    PoseDev* d_pose = nullptr; FrameDev* d_frame = nullptr;
    float4* d_predV = nullptr; float4* d_predN = nullptr; uchar4* d_predImage = nullptr; uint16_t* d_predTime = nullptr;
    uint8_t* d_predGray = nullptr; uint8_t* d_fillGray = nullptr;  // intensity of the RGB projection / of the fill-in image
    bool pred_gray_valid = false;          // the last prediction of this model wrote d_predGray (/ d_fillGray): it ran with a photometric term configured
    // RGBDOdometry of the model (Model::frameToModel): model-side pyramid, Gauss-Newton state, per-workgroup partial sums
    float* d_vmap_g[3] = {}; float* d_nmap_g[3] = {}; GNState* d_gn = nullptr; float* d_partials[2] = {nullptr, nullptr};
    TrackModelDev* d_track = nullptr;      // the block the batched tracker kernels find all of that through
    int* d_track_rect = nullptr;           // TrackModelDev::rect
    float* d_icp_log = nullptr;            // [20][32] reduced systems of the model's last tracking step, one row per iteration (debug tap "icp_log")
    double* d_gn_trace = nullptr;          // [20][kGnTraceRow] systems in fp64 + the state every iteration used (debug tap "gn_trace"; geometric loop)
    // object models: private scratch of the surfel passes, so that the passes of ALL objects of a frame can be one launch each ("batchObjectPasses")
    struct ObjScratch {
        unsigned long long* keys = nullptr; int* index = nullptr; float4* ivc = nullptr; float4* inr = nullptr; float4* iclean = nullptr;
        uint8_t* cand_op = nullptr; float4* cand_rec = nullptr; int* upd_first = nullptr; int* cand_best = nullptr;
        int* clean_ctl = nullptr;
        uint8_t* flags = nullptr; float* newconf = nullptr; int* block_counts = nullptr;   // the two-launch clean form (small maps)
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
        if (h_append) (void)hipHostFree(h_append);
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
    hipStream_t stream = nullptr;      // the frame's chain: preprocessing, tracking, fusion, prediction
    bool clean_literal = true;                         // Model::clean walks its window with the shader text's fp32 trip count ("cleanLiteralWindow")
    bool global_tiles = true;                          // A/B + test knob ("globalTiles"): 0 = every model through k_global_scatter
    bool early_bg_fusion = true;                       // A/B knob ("earlyBackgroundFusion"): 0 = the host visit drains the stream
    hipEvent_t ev_labels = nullptr;                    // the label stage of this frame has written its result words
    long frame_no = 0;
    long bg_fused_frame = -1;          // mf_fuse_background has fused the background of this staged frame (mf_fuse_models then skips it)
    bool labels_pending = false;       // between mf_perform_segmentation_begin and _end
    int lastF = 0;
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
    // "objectStream": inside mf_process_frame the batched passes of the object models (fuse / clean chain, prediction) go to a stream of their own
    // and run BESIDE the background's chain on the main stream -- the two touch disjoint buffers (an object model's private scratch, its own maps) and
    // share read-only inputs (frame, label image, poses).  Fork: behind the label stage (the host has waited for it); join: the end of the frame.
    // obj_s: where the batched object passes are enqueued right now (the main stream outside that window); obj_dep_main: main-stream work the object
    // chain depends on has been enqueued since the fork (a spawn, a compaction) -- the object stream waits for it first.
    // "fusedPreprocessLaunch": a single tracked background's model-side pyramid is built in the depth filter's launch (k_bilateral_model_pyramid);
    // pyr_done: the model whose pyramid of THIS frame that launch has built (enqueue_track then skips its own launch)
    bool fused_preprocess = true;
    ModelState* pyr_done = nullptr; bool pyr_batch_done = false;   // (... or the batched tracker's pyramids of this frame)
    bool object_stream = true;
    hipStream_t stream_obj = nullptr, obj_s = nullptr;
    hipEvent_t ev_obj_dep = nullptr, ev_obj_done = nullptr;
    bool obj_dep_main = false;
    uint8_t* d_maskT_obj = nullptr;                    // the object chain's own copy of d_maskT (every packed resolve pass writes the whole plane)
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
    bool batch_solve_fused = true;         // "batchSolveInPixelPass": the batched Gauss-Newton loop as ONE launch per iteration (k_icp_batch_pixels with k_icp_batch_solve's work as its prologue); 0: two
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
    float4* d_iclean = nullptr;            // packed column-major index map of the clean pass: 2 x float4 per texel (the spare word carries the filtered depth)
    uint8_t* d_maskT = nullptr;            // the frame's mask in the same column-major order (launch_index_resolve writes it, clean reads it)
    uint8_t* d_cand_op = nullptr; float4* d_cand_rec = nullptr; int* d_upd_first = nullptr;
    uint8_t* d_flags = nullptr; float* d_newconf = nullptr; int* d_block_counts = nullptr;
    int* d_cand_best = nullptr;            // surfel a merge candidate was associated with (fuse_data -> fuse_update)
    int* d_clean_ctl = nullptr;            // Model::clean in place (mf_surfel.hip): finished-workgroup counter of k_clean_runs
    int* d_clean_list = nullptr; int* d_clean_count = nullptr;   // ... the runs it has to visit (k_cull_clean)
    int* d_decay_stats = nullptr;          // ... and the frame's mask-disagreement depth ranges the culling rests on (ResolveOut::decay_stats), armed between frames
    int* d_run_offs = nullptr;             // compaction of a sparse buffer (launch_densify): exclusive scan of the run lengths
    // visibility list of the projection passes (Surfels::box, k_cull): the runs of ONE buffer that can be in view under ONE pose; vis_tag says whose
    int* d_vis_list = nullptr; int* d_vis_count = nullptr; int* d_cull_ctl = nullptr; int vis_max_runs = 0;
    struct { const void* model = nullptr; long frame = -1; int cur = -1; unsigned gen = 0; float max_depth = 0.f; int time_delta = 0; } vis_tag;
    int densify_count = 0;                 // compactions of sparse buffers so far ("densifyCount", read-only: tests / bench)
    bool fused_rgb_pyramid = true;         // "fusedRgbPyramid": the frame's intensity pyramid + derivative / gate images as one launch (0: four launches, the executable specification)
    float2* d_row_z = nullptr;             // batched Gauss-Newton loop: depth range of every row of the frame's vertex maps (launch_row_zrange)
    bool slab_culling = true;              // "slabCulling": the batched pixel pass skips the workgroups whose pixels cannot project onto a model's normals (exact)
    int densify_every = 0;                 // "densifyEvery": > 0 = a model's sparse buffer is compacted every so many frames whatever its bounds say (tests)
    bool cull_runs = true;                 // "cullRuns": 0 = every projection pass streams the whole buffer (A/B switch, executable specification)
    int big_map_elements = 6000000;        // "bigMapElements": from this many surfels on a model's buffer is kept as runs (Surfels::box): its clean pass works in
                                           // place on the runs its rules can touch and its projection passes cull by run; below, the two-launch clean (a dense
                                           // copy) and whole-buffer passes
    int in_place_elements = 1000000;       // "inPlaceElements": from this many surfels on update.vert runs in place and the second index scatter is a
                                           // launch of its own (~18 us per million surfels + ~10 us); below, the copying update with the scatter
                                           // riding on it (~30 us per million).  Only with the two-launch clean (<= big_map_elements)
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
    // "passTimings": GPU milliseconds of the surfel passes of the last frame, pass by pass (mf_get_pass_timings; labels in maskfusion_amd.h) -- an
    // event pair around each, created on first use; what bench.py's configs[4] line builds its per-pass roofline rows from IN THE TIMED RUN
    bool pass_timings_on = false;
    hipEvent_t ev_pass[MF_N_PASSES][2] = {};
    bool pass_recorded[MF_N_PASSES] = {};
    float pass_ms[MF_N_PASSES] = {};
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
    f->phys = 0; f->first = 0; f->first_run = 0; f->runsNext = 0;
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
    A(dev_alloc(c, m.allocs, &m.scr.clean_ctl, (size_t)kCleanCtlInts));
    A(dev_alloc(c, m.allocs, &m.scr.flags, cap + P));
    A(dev_alloc(c, m.allocs, &m.scr.newconf, cap + P));
    A(dev_alloc(c, m.allocs, &m.scr.block_counts, (size_t)kCompactBlocks));
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
        A(dev_alloc(c, m->allocs, &m->surf[b].box, run_table_entries((long)cap, (long)c->P)));
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
    A(dev_alloc(c, m->allocs, &m->d_track_rect, 12));
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
        t.rect = m->d_track_rect;
        m->track_host = t;
    }
    hipLaunchKernelGGL(k_pose_identity, dim3(1), dim3(64), 0, c->stream, m->d_pose, c->weight_literal ? 1 : 0);
    hipLaunchKernelGGL(k_frame_init, dim3(1), dim3(64), 0, c->stream, m->d_frame, c->host_tick);
    if (hipHostMalloc((void**)&m->h_pose, sizeof(PoseDev)) != hipSuccess || hipHostMalloc((void**)&m->h_frame, sizeof(FrameDev)) != hipSuccess ||
        hipHostMalloc((void**)&m->h_count, sizeof(int)) != hipSuccess || hipHostMalloc((void**)&m->h_append, sizeof(unsigned long long)) != hipSuccess)
        return MF_ENOMEM;
    *m->h_append = 0ull;
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
    {
        int armed[12];   // (armed: the batched finalize re-arms it after every frame)
        for (int q = 0; q < 12; ++q) armed[q] = (q & 2) ? (int)0x80000000 : 0x7FFFFFFF;
        MF_HIP(c, hipMemcpyAsync(m->d_track_rect, armed, sizeof(armed), hipMemcpyHostToDevice, c->stream));
    }
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
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) return fail(MF_ENODEV);
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
    if (hipStreamCreateWithFlags(&c->stream_obj, hipStreamNonBlocking) != hipSuccess) return fail(MF_EHIP);
    if (hipEventCreateWithFlags(&c->ev_obj_dep, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_obj_done, hipEventDisableTiming) != hipSuccess) return fail(MF_EHIP);
    c->obj_s = c->stream;
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
    A(dev_alloc(c, c->allocs, &c->d_maskT, (size_t)P));
    A(dev_alloc(c, c->allocs, &c->d_maskT_obj, (size_t)P));
    A(dev_alloc(c, c->allocs, &c->d_inr, (size_t)P));
    A(dev_alloc(c, c->allocs, &c->d_cand_op, (size_t)P));
    A(dev_alloc(c, c->allocs, &c->d_cand_rec, (size_t)P * 3));
    A(dev_alloc(c, c->allocs, &c->d_upd_first, (size_t)c->cap_max));
    A(dev_alloc(c, c->allocs, &c->d_flags, (size_t)c->cap_max + P));
    A(dev_alloc(c, c->allocs, &c->d_newconf, (size_t)c->cap_max + P));
    A(dev_alloc(c, c->allocs, &c->d_block_counts, (size_t)kCompactBlocks));
    A(dev_alloc(c, c->allocs, &c->d_cand_best, (size_t)P));
    A(dev_alloc(c, c->allocs, &c->d_clean_ctl, (size_t)kCleanCtlInts));
    A(dev_alloc(c, c->allocs, &c->d_row_z, (size_t)(c->H + (c->H >> 1) + (c->H >> 2))));
    c->vis_max_runs = (int)run_table_runs((long)c->cap_max, (long)P);
    A(dev_alloc(c, c->allocs, &c->d_vis_list, (size_t)c->vis_max_runs));
    A(dev_alloc(c, c->allocs, &c->d_clean_list, (size_t)c->vis_max_runs));
    A(dev_alloc(c, c->allocs, &c->d_clean_count, 1));
    A(dev_alloc(c, c->allocs, &c->d_run_offs, (size_t)c->vis_max_runs + 1));
    A(dev_alloc(c, c->allocs, &c->d_decay_stats, (size_t)kDecayStats + 3));
    launch_arm_decay_stats(c->d_decay_stats, c->stream);
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
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    c->models.clear();
    for (void* p : c->allocs) (void)hipFree(p);
    if (c->d_splat_prof) (void)hipFree(c->d_splat_prof);
    for (void* p : c->host_allocs) (void)hipHostFree(p);
    for (int i = 0; i <= MF_N_TIMINGS; ++i)
        if (c->ev[i]) (void)hipEventDestroy(c->ev[i]);
    for (int i = 0; i < 2; ++i) {
        if (c->ev_icp[i]) (void)hipEventDestroy(c->ev_icp[i]);
        for (int q = 0; q < MF_N_PASSES; ++q) if (c->ev_pass[q][i]) (void)hipEventDestroy(c->ev_pass[q][i]);
    }
    if (c->ev_icp_mid) (void)hipEventDestroy(c->ev_icp_mid);
    for (int i = 0; i < 5; ++i)
        if (c->ev_mm[i]) (void)hipEventDestroy(c->ev_mm[i]);
    if (c->ev_labels) (void)hipEventDestroy(c->ev_labels);
    for (int i = 0; i < 2; ++i) {
        if (c->ev_in_copied[i]) (void)hipEventDestroy(c->ev_in_copied[i]);
        if (c->ev_in_consumed[i]) (void)hipEventDestroy(c->ev_in_consumed[i]);
    }
    if (c->stream_in) { (void)hipStreamSynchronize(c->stream_in); (void)hipStreamDestroy(c->stream_in); }
    if (c->stream_obj) { (void)hipStreamSynchronize(c->stream_obj); (void)hipStreamDestroy(c->stream_obj); }
    if (c->ev_obj_dep) (void)hipEventDestroy(c->ev_obj_dep);
    if (c->ev_obj_done) (void)hipEventDestroy(c->ev_obj_done);
    for (int i = 0; i < mf_ctx::kObjArgSlots; ++i) {
        if (c->d_obj_args[i]) (void)hipFree(c->d_obj_args[i]);
        if (c->h_obj_args[i]) (void)hipHostFree(c->h_obj_args[i]);
        if (c->ev_obj_args[i]) (void)hipEventDestroy(c->ev_obj_args[i]);
    }
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

extern "C" const char* mf_last_error(const mf_ctx* c) { return c ? c->err.c_str() : "null context"; }

#include "mf_frame.inl"      // processFrame: stages, batches, spawn / retire, mf_process_frame[_dev], mf_sync, mf_predict

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

#include "mf_model_api.inl"  // Model-level and sharded-scene entry points

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
    require_dense(c, *ms);     // Model::downloadMap hands out the surfels in order, slot by slot: a sparse buffer is compacted first
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
    if (!strcmp(key, "passTimings")) {
        c->pass_timings_on = value != 0;
        for (int q = 0; q < MF_N_PASSES; ++q) { c->pass_recorded[q] = false; c->pass_ms[q] = 0.f; }
        return MF_OK;
    }
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
    if (!strcmp(key, "batchSolveInPixelPass")) { c->batch_solve_fused = value != 0; return MF_OK; }   // 0: k_icp_batch_solve + k_icp_batch_pixels per iteration (the executable specification)
    if (!strcmp(key, "fusedPreprocessLaunch")) { c->fused_preprocess = value != 0; return MF_OK; }   // 0: k_bilateral and k_model_pyramid as two launches
    if (!strcmp(key, "objectStream")) { c->object_stream = value != 0; return MF_OK; }   // 0: the object models' batched passes on the main stream, behind the background's
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
    if (!strcmp(key, "densifyEvery")) { c->densify_every = (int)value; return MF_OK; }
    if (!strcmp(key, "slabCulling")) { c->slab_culling = value != 0; return MF_OK; }
    if (!strcmp(key, "bigMapElements")) { c->big_map_elements = (int)value; c->vis_tag.model = nullptr; return MF_OK; }
    if (!strcmp(key, "fusedRgbPyramid")) { c->fused_rgb_pyramid = value != 0; return MF_OK; }
    if (!strcmp(key, "inPlaceElements")) { c->in_place_elements = (int)value; return MF_OK; }
    if (!strcmp(key, "rebuildRunTable")) {   // tooling: the background's run table from scratch (what an upload / Model::initialise does)
        launch_run_table(c->models[0]->surf[c->models[0]->cur], c->models[0]->d_frame, c->stream);
        c->vis_tag.model = nullptr;
        return check_launch(c);
    }
    if (!strcmp(key, "cleanLiteralWindow")) { c->clean_literal = value != 0; return MF_OK; }   // 0: the exact-arithmetic 4 x 4 window
    if (!strcmp(key, "earlyBackgroundFusion")) { c->early_bg_fusion = value != 0; return MF_OK; }
    if (!strcmp(key, "modelApiPackedIndex")) { c->model_api_packed = value != 0; return MF_OK; }   // 0: scatter + resolve form (specification)
    if (!strcmp(key, "tileHeight")) { c->splat_tune.tile_h = (int)value; return MF_OK; }   // A/B: 16 x 24 (default), 16 x 20 or 16 x 16 pixel tiles
    if (!strcmp(key, "tileThreads")) { c->splat_tune.tile_threads = (int)value; return MF_OK; }   // A/B: threads per tile workgroup of the tile passes
    if (!strcmp(key, "spriteLanes")) { c->splat_tune.sprite_lanes = (int)value; return MF_OK; }   // A/B: lanes per sprite in the tile z-test (default 4)
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
    if (!strcmp(key, "densifyCount")) { *value = (float)c->densify_count; return MF_OK; }
    if (!strcmp(key, "visibleRuns") || !strcmp(key, "backgroundRuns") || !strcmp(key, "cleanRuns")) {   // test taps: size of the last visibility list / of the
        MF_HIP(c, hipStreamSynchronize(c->stream));                                                      // background's run table / of the last clean list
        int v = 0;
        if (!strcmp(key, "visibleRuns")) MF_HIP(c, hipMemcpy(&v, c->d_vis_count, sizeof(int), hipMemcpyDeviceToHost));
        else if (!strcmp(key, "cleanRuns")) MF_HIP(c, hipMemcpy(&v, c->d_clean_count, sizeof(int), hipMemcpyDeviceToHost));
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
extern "C" int mf_get_pass_timings(mf_ctx* c, float* ms) {
    if (!c || !ms) return MF_EINVAL;
    int rc = mf_sync(c);
    if (rc != MF_OK) return rc;
    memcpy(ms, c->pass_ms, sizeof(c->pass_ms));
    return MF_OK;
}
extern "C" void* mf_get_stream(mf_ctx* c) { return c ? (void*)c->stream : nullptr; }
extern "C" void* mf_get_input_stream(mf_ctx* c) { return c ? (void*)c->stream : nullptr; }

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
    auto img = [&](const std::string& pre, const void* const arr[3], size_t elem) -> bool {   // frame-side intensity pyramid / derivative / gate images
        for (int i = 0; i < 3; ++i)
            if (w == pre + std::to_string(i)) {
                src = arr[i]; bytes = (size_t)(c->W >> i) * (c->H >> i) * elem;
                return true;
            }
        return false;
    };
    const void* const gray3[3] = {c->d_gray[lastSet][0], c->d_gray[lastSet][1], c->d_gray[lastSet][2]};
    const void* const dx3[3] = {c->d_dIdx[0], c->d_dIdx[1], c->d_dIdx[2]}, * const dy3[3] = {c->d_dIdy[0], c->d_dIdy[1], c->d_dIdy[2]};
    const void* const gate3[3] = {c->d_rgb_gate[0], c->d_rgb_gate[1], c->d_rgb_gate[2]};
    if (w == "depthF") { src = c->d_depthF[c->lastF]; bytes = P * 4; }
    else if (img("gray", gray3, 1) || img("dIdx", dx3, 2) || img("dIdy", dy3, 2) || img("rgb_gate", gate3, 1)) {}
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

#include "mf_ktest.inl"      // kernel-level entry points of the parity tests

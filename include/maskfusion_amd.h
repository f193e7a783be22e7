/*
 * maskfusion_amd.h -- C ABI of the MI355X-native MaskFusion hot path (libmaskfusion_amd.so).
 *
 * Drop-in boundary for `MaskFusion::processFrame` and the `Model` operations it drives
 * (reference: martinruenz/maskfusion, Core/MaskFusion.h:45-307, Core/Model/Model.h:108-268).
 * The reference exposes a C++14 class over Eigen/OpenCV/OpenGL types; this ABI carries the same
 * operations over plain pointers and sizes.  include/maskfusion/MaskFusion.h is the header-only C++
 * facade with the reference's class/method names on top of it; INTEGRATION.md shows the binding a
 * MaskFusion maintainer adds.
 *
 * Conventions
 *   - 4x4 poses: 16 floats, COLUMN-major (memcpy-compatible with Eigen::Matrix4f::data()).
 *   - rgb: H*W*3 uint8 (FrameData::rgb, CV_8UC3); depth: H*W float32 metres, 0 = invalid
 *     (FrameData::depth, CV_32FC1); mask: H*W uint8 model ids (FrameData::mask, CV_8UC1) or NULL.
 *   - surfel record (Model::SurfelMap, Core/Model/Model.h:193-206): 12 floats
 *     {x,y,z,conf | colour(24-bit int as float),unused,initTime,lastTime | nx,ny,nz,radius}.
 *   - every function returns 0 on success or a negative MF_E* code; mf_last_error() has the text.
 *     No exceptions cross the ABI; a context is thread-compatible (one caller at a time), owns one HIP
 *     stream on one GPU, and holds no process-global state (unlike the reference's Resolution /
 *     Intrinsics / GPUSetup singletons, Core/Utils/Resolution.h:24-71, Core/Model/Model.h:54-89).
 *   - "_dev" entry points take DEVICE pointers (HBM-resident inputs); the others take host pointers.
 *   - The library needs a gfx950 GPU: mf_create() fails with MF_ENODEV otherwise.  There is no CPU path.
 */
#ifndef MASKFUSION_AMD_H_
#define MASKFUSION_AMD_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MF_OK 0
#define MF_EINVAL (-1)  /* bad argument */
#define MF_ENODEV (-2)  /* no usable GPU / HIP failure at init */
#define MF_EHIP (-3)    /* HIP runtime error (text in mf_last_error) */
#define MF_ENOMEM (-4)
#define MF_ESTATE (-5)  /* call not valid in this state */

typedef struct mf_ctx mf_ctx;

/* Constructor arguments of MaskFusion (Core/MaskFusion.h:47-53) that are live on the hot path, plus what the
 * reference reads from the Resolution / Intrinsics singletons (GUI/MainController.cpp:117-128) and the
 * compile-time surfel budgets (Core/CMakeLists.txt:27-28; Core/Model/Model.cpp:101-108). */
typedef struct mf_config {
    int32_t width, height;
    float fx, fy, cx, cy;
    int32_t device;               /* HIP device ordinal (MASKFUSION_GPU_SLAM) */
    int32_t time_delta;           /* timeDelta = 200 */
    float conf_global;            /* initConfidenceGlobal = 4 */
    float conf_object;            /* initConfidenceObject = 2 */
    float depth_cutoff;           /* depthCut = 3 */
    float icp_weight;             /* icpThresh = 10; >= 100 => geometric term only */
    int32_t fast_odom;            /* fastOdom = 0 */
    int32_t so3;                  /* so3 = 1 */
    int32_t pyramid;              /* pyramid = 1 (MaskFusion.cpp:60) */
    float max_depth_processed;    /* 20 (MaskFusion.cpp:57) */
    float outlier_coefficient;    /* Model::GPUSetup::outlierCoefficient = 0.9 (Model.h:85) */
    int32_t num_gsurfels;         /* MASKFUSION_NUM_GSURFELS = 9437184 */
    int32_t num_osurfels;         /* MASKFUSION_NUM_OSURFELS = 1048576 */
    int32_t enable_multiple_models; /* setEnableMultipleModels; 0 == "-static" */
    int32_t model_spawn_offset;   /* modelSpawnOffset = 20 (Core/MaskFusion.h:51) */
    int32_t track_all_models;     /* MaskFusion::trackAllModels = true (Core/MaskFusion.h:396) */
    int32_t max_models;           /* upper bound on live models (the reference: 256 ids) */
    int32_t rgb_only;             /* rgbOnly = 0 (Core/MaskFusion.h:169): photometric term only */
    int32_t pose_log_capacity;    /* enablePoseLogging (Core/MaskFusion.h:402): entries kept per model; 0 = off */
    int32_t reserved[3];
} mf_config;

/* Fills *cfg with the reference's constructor defaults for a WxH camera. */
int mf_default_config(mf_config* cfg, int32_t width, int32_t height, float fx, float fy, float cx, float cy);

/* MaskFusion::MaskFusion (Core/MaskFusion.cpp:24-120) */
int mf_create(const mf_config* cfg, mf_ctx** out);
/* MaskFusion::~MaskFusion (Core/MaskFusion.cpp:122-142) */
void mf_destroy(mf_ctx* ctx);
const char* mf_last_error(const mf_ctx* ctx);

/* MaskFusion::processFrame (Core/MaskFusion.h:69-70, Core/MaskFusion.cpp:200-607).
 * mask/class_ids may be NULL (n_masks = 0); in_pose16 may be NULL.  The caller's buffers are copied into pinned staging memory before
 * the call returns (they may be reused at once); the upload and the frame itself are ENQUEUED -- the call does not wait for THIS frame
 * (it keeps the host at most two frames ahead: it waits for frame k-2 before enqueueing frame k's upload; a multi-model frame also
 * waits once for the label stage's decision).  Every getter below synchronises first, so
 * processFrame(...); getCurrPose() reads this frame's pose as upstream; mf_sync() waits explicitly.
 * mf_set_param("hostInputAsync", 0) restores the blocking form of rounds 1-3 ("blocks until the frame is fused"). */
int mf_process_frame(mf_ctx* ctx, const uint8_t* rgb, const float* depth, const uint8_t* mask,
                     const int32_t* class_ids, int32_t n_masks, int64_t timestamp, const float* in_pose16,
                     float weight_multiplier, int32_t bootstrap);
/* Same, inputs already resident in HBM (device pointers); enqueues the whole frame on the context's stream and
 * returns without waiting.  mf_sync() (or any getter) waits. */
int mf_process_frame_dev(mf_ctx* ctx, const uint8_t* d_rgb, const float* d_depth, const uint8_t* d_mask,
                         int64_t timestamp, float weight_multiplier);
/* FrameData::classIDs (Core/FrameData.h:25-48) of the masks handed to mf_process_frame_dev: class_ids[v] = class of mask value v
 * (n <= 256; until it is called every mask value is class 0) */
int mf_set_mask_class_ids(mf_ctx* ctx, const int32_t* class_ids, int32_t n);
/* Waits for everything enqueued on the context's stream.  MF_EHIP (text in mf_last_error) for a HIP failure, and when the ordered compaction
 * of a full map's Model::clean gave up one of its bounded waits (never in a correct run; that model's map is then not valid). */
int mf_sync(mf_ctx* ctx);

/* MaskFusion::setTick (Core/MaskFusion.h:206); only after the first frame (tick 1 initialises the map) */
int mf_set_tick(mf_ctx* ctx, int32_t tick);
/* MaskFusion::preallocateModels (Core/MaskFusion.h:57, Core/MaskFusion.cpp:144-149) */
int mf_preallocate_models(mf_ctx* ctx, uint32_t count);
/* MaskFusion::predict (Core/MaskFusion.h:76) */
int mf_predict(mf_ctx* ctx);

/* MaskFusion::getTick / getModels().size() / Model::getPose / lastCount / getConfidenceThreshold
 * (Core/MaskFusion.h:88-90,194; Core/Model/Model.h:180,233).  `model` is the index in the model list (0 = background,
 * objects in spawn order, MaskFusion::getModels()); mf_model_info gives the Model::getID() behind an index. */
int mf_get_tick(mf_ctx* ctx, int32_t* tick);
int mf_num_models(mf_ctx* ctx, int32_t* n);
int mf_get_pose(mf_ctx* ctx, int32_t model, float* out_pose16);
int mf_get_surfel_count(mf_ctx* ctx, int32_t model, uint32_t* count);
/* Asynchronous device-side copy of a model's state into a caller-owned DEVICE buffer of 16 floats
 * {R row-major (9), t (3), lastICPError, lastICPCount, surfel count, alive}: what the multi-GPU gather ships per rank
 * (the reference logs the same per model, Core/MaskFusion.cpp:580-602) without a host round trip. */
int mf_model_state_dev(mf_ctx* ctx, int32_t model, float* d_out16);
/* the same for every model of the list in one call: d_out16[16 * i ..] = models[i]; capacity = models the buffer holds */
int mf_models_state_dev(mf_ctx* ctx, float* d_out16, int32_t capacity);
/* ids of the model list (MaskFusion::models, Core/MaskFusion.h:329-333) in list order; host state, no synchronisation */
int mf_get_model_ids(mf_ctx* ctx, int32_t* ids, int32_t capacity, int32_t* n);
/* Model::getPoseLog (Core/Model/Model.h:257-262) of model i, chronological: ts[e] = the frame's timestamp, p7[e] =
 * {tx ty tz qx qy qz qw} of cam->world (background) or obj->world (objects), Core/MaskFusion.cpp:580-596.
 * ts/p7 may be NULL to query *count only. */
int mf_get_pose_log(mf_ctx* ctx, int32_t model, int64_t* ts, float* p7, uint32_t max_entries, uint32_t* count);
/* MaskFusion::exportPoses (Core/MaskFusion.cpp:851-879): writes <export_dir>poses-<id>.txt ("ts[s] tx ty tz qx qy qz qw",
 * 6 decimals; ts = timestamp * 1e-6) for every live and dropped model */
int mf_export_poses(mf_ctx* ctx, const char* export_dir);
/* MaskFusion::savePly (Core/MaskFusion.cpp:733-849): writes <export_dir>cloud-<id>.ply per live model */
int mf_save_ply(mf_ctx* ctx, const char* export_dir);
/* {lastICPError, lastICPCount, lastRGBError, lastRGBCount, lastSO3Error, lastSO3Count, so3 iterations, rejected by the
 * 0.3 m rule} of the last tracking step of model i (RGBDOdometry.h:68-75, RGBDOdometry.cpp:477-481) */
int mf_get_track_stats(mf_ctx* ctx, int32_t model, float* out8);
/* No upstream twin.  The device solves the 6x6 Gauss-Newton system (RGBDOdometry.cpp:447-459 hands it to Eigen::LDLT, diagonal pivoting, on
 * float-rounded sums) with an unpivoted LDL^T on fp64 sums: identical to rounding for a well-posed system, NOT for a rank-deficient one
 * (DESIGN.md finding F4).  *ill_iterations = how many iterations of the model's last geometric tracking step were outside that domain:
 * fewer than 6 inliers, or smallest pivot < 1e-8 x largest diagonal entry (=> cond(A) > 1e8).  0 means the step is the reference's to rounding. */
int mf_get_gn_condition(mf_ctx* ctx, int32_t model, int32_t* ill_iterations);
/* RGBDOdometry::lastICPError / lastICPCount (Core/Utils/RGBDOdometry.h) of `model` */
int mf_get_icp_stats(mf_ctx* ctx, int32_t model, float* last_error, float* last_count);
/* Model::downloadMap (Core/Model/Model.h:206, Model.cpp:943-974): out has room for max_count*12 floats */
int mf_download_map(mf_ctx* ctx, int32_t model, float* out, uint32_t max_count, uint32_t* count);
/* Model::getID / getClassID / lastCount / getConfidenceThreshold / isNonstatic (Core/Model/Model.h:180,241-268) */
typedef struct mf_model_info_t {
    int32_t id, class_id;
    uint32_t surfels;
    float confidence_threshold;
    int32_t is_static;
    uint32_t age;
} mf_model_info_t;
int mf_model_info(mf_ctx* ctx, int32_t model, mf_model_info_t* out);
/* SegmentationResult::fullSegmentation of the last frame (Core/Segmentation/SegmentationResult.h:35): H*W model ids,
 * 255 = ignored.  Only meaningful with enable_multiple_models. */
int mf_download_segmentation(mf_ctx* ctx, uint8_t* out);
/* The exportSegmentation branch of processFrame (Core/MaskFusion.cpp:299-303): the label image of the last frame with 255 (ignored)
 * zeroed, as an 8-bit greyscale PNG at `path` (upstream: exportDir + "Segmentation<tick>.png") */
int mf_export_segmentation_png(mf_ctx* ctx, const char* path);
/* The cv::imwrite(<name>.png, CV_8UC1) of that branch on its own: H*W bytes -> 8-bit greyscale PNG.  HOST pointer, no GPU involved. */
int mf_write_png_gray8(const char* path, const uint8_t* img, int32_t width, int32_t height);
/* whether the last tracking step used the fill-in maps (MaskFusion::requiresFillIn, MaskFusion.cpp:630-648) */
int mf_get_last_fillin(mf_ctx* ctx, int32_t* used);

/* The per-frame setters of MaskFusion (Core/MaskFusion.h:132-182,234-263).  Keys: "depthCutoff", "icpWeight",
 * "confidenceThreshold" (background), "outlierCoefficient", "fastOdom", "so3", "rgbOnly" (setRgbOnly: photometric term only), "pyramid", "timeDelta",
 * "maxDepthProcessed", "enableMultipleModels", "trackAllModels", "modelSpawnOffset",
 * "newModelMinRelativeSize", "newModelMaxRelativeSize" (SegmentationPerformer.h:36-37) and the MfSegmentation tunables
 * (MaskFusion.h:234-263 / MfSegmentation.h:42-62): "mfThreshold", "mfWeightDistance", "mfWeightConvexity",
 * "mfMorphEdgeIterations", "mfMorphEdgeRadius", "mfMorphMaskIterations", "mfMorphMaskRadius".
 * Implementation switches (defaults are the product; the alternatives exist for A/B measurements and as executable specifications):
 * "splatTiles" (1), "globalTiles" (1: GlobalProjection of the background through tile lists), "gpuLabels" (1: label stage on the
 * device), "batchTracking" (1: one Gauss-Newton launch serves every tracked model), "earlyBackgroundFusion" (1),
 * "cleanLiteralWindow" (1: Model::clean walks its window with copy_unstable.vert's own fp32 trip count, 4 or 5 taps per axis;
 * 0: 4 x 4), "timings", "passTimings" (mf_get_pass_timings), "icpProfile", "objectSmallGrids" (1; 1: the grid-stride
 * surfel kernels of an object model run on a grid sized from its last known surfel count instead of 2048 workgroups), "objectScatterSplat" (1;
 * 1: object models are predicted with the scatter form of the splat instead of tile lists) -- both leave every result bit-identical; measured on
 * MI355X on the 12-model S2 scene: 394 -> 407 frames/s with both (profiles/r03a_bench_2s_object_switches.txt), on since round 3;
 * "bigMapElements" (6 000 000: from this many surfels on a model's buffer is kept as RUNS with a table (Surfels::box): Model::clean works in
 * place on the runs in which its rules can change something (k_cull_clean / k_clean_runs, the frame's new surfels appended behind the last run)
 * and the projection passes visit only the runs that can be in view (k_cull) -- below, the two-launch clean (a dense copy) and whole-buffer passes
 * of rounds 1-4), "inPlaceElements" (1 000 000: from this
 * many surfels on update.vert runs in place -- below, as rounds 1-4's copy with the second index scatter riding on it); which form a model's
 * passes take depends on its size alone and changes no result (tests/test_gpu_switches.py::test_clean_forms_agree), "cullRuns" (1; 0: big maps walk every run of
 * the buffer in every projection pass and in Model::clean -- the executable specification of the culled forms), "slabCulling" (1: the
 * batched Gauss-Newton pixel pass skips the workgroups none of whose pixels can project onto a model's normals -- exact, mf_odometry.hip; 0: every
 * workgroup walks its pixels), "densifyEvery" (0; n > 0: a
 * sparse buffer is compacted every n frames whatever the host's bounds say -- a test switch; by default only when the slots behind the last run or
 * the table entries could run out, before a download, before a model returns to the small-map forms); read-only: "densifyCount" (compactions so
 * far), "cleanRuns" / "visibleRuns" / "backgroundRuns" (runs the last in-place clean visited / on the last visibility list / of the
 * background's table),
 * "literalFusionWeight" (1: Model::computeFusionWeight's log map takes cos(theta) from the float trace of a float matrix as the reference's
 * text does -- its rotation term is then quantised in steps of ~4.9e-4 rad; 0: the same formula evaluated accurately in double.  See
 * DESIGN.md, finding F5; default 1 since round 3), "frameToFrameRGB" (0; MaskFusion::setFrameToFrameRGB, "-ftf": the photometric term tracks
 * against the previous RAW frame, Model.cpp:399-400,981), "objectBoundingBoxLimit" (1: Model::fuse limits an object model's depth by its
 * bounding box + 5 % as upstream does whenever its GUI draws the models, Model.cpp:480-501; 0: bb_max_z = FLT_MAX, a headless upstream).
 * "fusedRgbPyramid" (1: the frame's intensity pyramid and its derivative / gate images -- photometric term, SO(3) -- as one LDS-tiled launch;
 * 0: imageBGRToIntensity + 2 x pyrDownUcharGauss + computeDerivativeImages as four launches, the executable specification; same bytes).
 * Further switches and taps: "batchObjectPasses" (1: the surfel passes of all object models of a frame as one launch per pass; 0: model by
 * model, the executable specification), "objectStream" (1: inside mf_process_frame those batched launches -- the objects' fuse / clean chain and
 * their prediction -- go to a second stream and run beside the background's chain, joined at the end of the frame; 0: one stream; same bytes),
 * "batchSolveInPixelPass" (1: an iteration of the batched Gauss-Newton loop is ONE launch -- every workgroup of a model finishes the previous
 * iteration in its prologue, as the single-model kernel does; 0: a solve launch and a pixel launch per iteration; same bytes),
 * "fusedPreprocessLaunch" (1: when only the background is tracked model by model, its model-side pyramid is built in the depth filter's launch --
 * two independent kernels of a frame side by side; 0: two launches; same bytes),
 * "hostLockstep" (1: mf_process_frame waits for frame k-2 to have run before it enqueues frame k's
 * upload), "hostWaitUpload" (1: ... and for its own upload: single-model frames), "modelApiPackedIndex" (0; 1: mf_model_predict_indices
 * also builds the packed column-major map mf_process_frame feeds Model::clean with), "tileThreads" (512) / "spriteLanes" (4) / "tileHeight" (24; 16, 20, 32: tiles of 16 pixels by that many rows): launch shape of
 * the tile passes (A/B), "splatTileEntries" (test knob: shrinks the tile lists to force their overflow path), "rebuildRunTable" (write-only:
 * rebuilds the background's run table from scratch), "splatProfile" (1: per-tile stamps of the background's tile pass, debug tap
 * "splat_prof").  Read-only (mf_get_param): "visibleRuns" / "backgroundRuns" (runs k_cull listed for the last pass / runs of the
 * background's table), "hostWaitUs" | "hostStageUs" | "hostUploadUs" | "hostEnqueueUs" | "hostCallUs" (host clocks inside
 * mf_process_frame, microseconds per call since mf_set_param("hostProfileReset", 1)). */
int mf_set_param(mf_ctx* ctx, const char* key, double value);
int mf_get_param(mf_ctx* ctx, const char* key, double* value);

/* Stage timings in the reference's Stopwatch label set (Core/Utils/Stopwatch.h:46-121): GPU milliseconds of the
 * last frame measured with HIP events when enabled by mf_set_param("timings", 1).
 * labels: 0 Preprocess, 1 odomInit (model-side pyramids of the background model), 2 odom (all tracking), 3 indexMap,
 *         4 Fuse::Data, 5 Fuse::Update, 6 Fuse::Copy, 7 IndexMap::ACTIVE, 8 Run,
 *         9 icpIterations: first to last Gauss-Newton iteration launch of the background model (what bench.py divides by
 *           the iteration count for its roofline line);
 *         10 icpCoarse / 11 icpFine: the same interval split right before the first level-0 iteration (launch-per-iteration loop of a
 *           single model; 0 for the batched form);
 *         multi-model frames only (0 otherwise) -- what lies between the end of tracking and Fuse::Copy's end (labels 3..6 cover the
 *         BACKGROUND's passes there; Core/MaskFusion.cpp:287-375,539-565):
 *         12 mmGlobalProjection (GlobalProjection::project of every model + id resolve), 13 mmEdgeLabels (geometric edge map, binary
 *         edges, the device label stage), 14 mmBackgroundFuseClean (the background's predictIndices / fuse / clean, enqueued ahead of
 *         the host's look at the label stage's decision), 15 mmHostStall (GPU idle between the end of that and the first object pass:
 *         the host had not enqueued it yet), 16 mmObjectFuseClean (spawn pass + every object model's predictIndices / fuse / clean),
 *         17 mmHostWaitMs (HOST wall-clock milliseconds spent in the event wait behind the label stage) */
#define MF_N_TIMINGS 18
int mf_get_timings(mf_ctx* ctx, float* ms /* [MF_N_TIMINGS] */);
/* The surfel passes of the last frame one by one (Core/Model/Model.cpp:466-772, ModelProjection.cpp:100-268, GlobalProjection.cpp:43-107): GPU
 * milliseconds between two HIP events recorded around each pass's launches, enabled by mf_set_param("passTimings", 1).  `bg`: the background
 * model (or any model handled on its own); `obj`: every object model, one launch per pass for all of them.
 *   0 bgGlobalProjection   k_cull + k_splat_bin + k_global_tile
 *   1 bgIndexMap           k_cull (if not shared) + k_index_scatter + resolve: predictIndices before fuse
 *   2 bgFuseData           k_fuse_data: association
 *   3 bgFuseUpdate         k_fuse_update[_copy]: update.vert
 *   4 bgIndexMap2          k_index_scatter + packed resolve: predictIndices after fuse (rides on the copy-update of small maps)
 *   5 bgClean              Model::clean of the buffer's own surfels: k_cull_clean + k_clean_runs (in place) / the two-launch form (everything)
 *   6 bgAppend             ... and of the frame's candidates, appended (in-place form only)
 *   7 bgPredict            combinedPredict: k_cull + k_splat_bin + k_splat_tile (+ the end-of-frame bookkeeping)
 *   8 objGlobalProjection  9 objFuseClean (predictIndices, fuse, predictIndices, clean)   10 objPredict
 *   11 compaction          launch_densify + the run table of the compacted buffer (0 in a frame without one)
 * With "objectStream" on, rows 9 and 10 are timed on the object stream, where they run BESIDE rows 1-7: each row is then longer than the pass on
 * its own and the rows no longer add up to the frame.  For passes on their own, switch the stage timings on as well ("timings": they keep the frame
 * on one stream) or set "objectStream" to 0 -- bench.py's roofline_passes does the former. */
#define MF_N_PASSES 12
enum { MF_PASS_BG_GLOBAL = 0, MF_PASS_BG_INDEX, MF_PASS_BG_FUSE_DATA, MF_PASS_BG_FUSE_UPDATE, MF_PASS_BG_INDEX2, MF_PASS_BG_CLEAN, MF_PASS_BG_APPEND,
       MF_PASS_BG_PREDICT, MF_PASS_OBJ_GLOBAL, MF_PASS_OBJ_FUSE_CLEAN, MF_PASS_OBJ_PREDICT, MF_PASS_COMPACTION };
int mf_get_pass_timings(mf_ctx* ctx, float* ms /* [MF_N_PASSES] */);
/* The context's HIP stream (hipStream_t), for callers that time with their own events. */
void* mf_get_stream(mf_ctx* ctx);
/* Stream on which rgb/depth/mask handed to mf_process_frame_dev are first read: the context's stream (mf_get_stream) -- producers ordered on
 * it need no further synchronisation; a buffer stays unmodified until its frame has completed there.  (Rounds 2-4 could move the
 * pose-independent preprocessing to a second stream, "overlapPreprocessing"; it never gained anything -- also not with that stream masked to
 * a few compute units, round 5 -- and is gone.  The entry point stays so that callers written against it keep working.) */
void* mf_get_input_stream(mf_ctx* ctx);

/* debug / differential-test taps: copy a device-resident intermediate of the last frame to host.
 * what: "depthF" (H*W f32), "vmap0".."vmap2", "nmap0".."nmap2" (3*h*w f32, current frame),
 *       "vmap_g0".."vmap_g2", "nmap_g0".."nmap_g2" (model side), "pred_vertex", "pred_normal" (H*W*4 f32),
 *       "pred_image" (H*W*4 u8), "pred_time" (H*W u16), "icp_log" (19*32 f32: per-iteration A-upper/b/res/inl),
 *       "gn_trace" (20*64 f64, geometric loop: row r = the reduced system of iteration r as summed in fp64 [0..31], then resultRt [32..47],
 *       Rcurr [48..56], tcurr [57..59] as iteration r used them; row 19 = the state after the last update),
 *       "edge_map" (H*W f32), "edge_binary" (H*W u8), "projected_ids" (H*W u8);
 *       index map of the last (pre-fusion) index pass, index_map.frag's attachments: "index" (H*W i32), "index_vc", "index_nr",
 *       "index_ct" (H*W*4 f32), "index_packed" (post-fusion pass: 2 float4 per texel, column-major texel order);
 *       Model::fuse / clean intermediates (variable length: as many bytes as asked for, at most the buffer): "cand_op" (u8 per
 *       quarter-rate pixel in column-major order: 0 none, 1 merge, 2 new), "cand_rec" (3 float4 per candidate), "clean_flags"
 *       (u8 keep flag per old surfel, then per candidate), "clean_newconf" (f32, same indexing). */
int mf_debug_read(mf_ctx* ctx, const char* what, void* out, uint64_t out_bytes);

/* ------------------------------------------------------------------------------------------------
 * Model-level entry points: the public operations of Model (Core/Model/Model.h:126-162,233-268), one call each, on a
 * frame staged with mf_stage_frame.  MaskFusion::processFrame is a fixed composition of them (mf_process_frame enqueues the
 * same kernels); they let a caller drive a model the way the reference's callers do, and let every surfel pass be compared
 * with the oracle in isolation.  `model` is the index in the model list (as in mf_get_pose).  All calls are asynchronous on
 * the context's stream except where noted.  The index map, the association candidates and the new-surfel records are
 * scratch shared by all models: predictIndices -> fuse -> [predictIndices] -> clean of ONE model must not be interleaved
 * with another model's (the reference keeps those buffers per model, ModelProjection / Model::newUnstableBuffer).
 * ---------------------------------------------------------------------------------------------- */
/* Everything of processFrame that touches no model: upload, MaskFusion::filterDepth (Core/MaskFusion.cpp:217,650-657),
 * Model::generateCUDATextures (Core/Model/Model.h:128, Model.cpp:350-389), the intensity pyramid / derivative images
 * (RGBDOdometry::initRGB).  mask = model id per pixel as textureMask holds it for fuse / clean (Core/MaskFusion.cpp:297);
 * NULL = leave textureMask as it is (all background in a fresh context).  Host pointers; synchronous. */
int mf_stage_frame(mf_ctx* ctx, const uint8_t* rgb, const float* depth, const uint8_t* mask);
/* The same for a frame that is already in device memory (no upstream twin: upstream's FrameData lives on the host, MaskFusion.cpp:212-216
 * uploads it).  Asynchronous: producers of the buffers are ordered on mf_get_stream(ctx) by the caller (e.g. an RCCL broadcast enqueued
 * there), and d_rgb / d_depth stay valid and unmodified until the model-level calls of this frame have completed on that stream. */
int mf_stage_frame_dev(mf_ctx* ctx, const uint8_t* d_rgb, const float* d_depth, const uint8_t* d_mask);
/* The tail of processFrame (Core/MaskFusion.cpp:569-602) for a frame driven through the calls below: tick++, the
 * requiresFillIn decision for the next tracking step, the pose-log entry with `timestamp`, age++ */
int mf_end_frame(mf_ctx* ctx, int64_t timestamp);
/* The three per-model LOOPS of MaskFusion::processFrame over this context's model list, for a caller that sequences a frame itself
 * (one scene sharded by model over several contexts).  They run what mf_process_frame runs for these loops -- ONE batched
 * Gauss-Newton loop over all tracked models, ONE launch per surfel pass for all object models -- under the context's configuration.
 * first_model = 0: the whole list; 1: models[0] is the stand-in of a background owned by another context (pose via
 * mf_model_override_pose; neither tracked, fused nor drawn, but its tick advances).
 *   mf_track_models    Core/MaskFusion.cpp:247-276 (trackableClassIds, updateStaticPose for static objects, the 0.2 m jump rule)
 *   mf_fuse_models     :335-339, :369-374 (setMaxDepth, confidence ramp), :342-353 for models[spawned_model] when spawned_model >= 1
 *                      (-1: none), then the fusion loop :539-565
 *   mf_predict_models  :569 predict(), :573 tick++, :580-596 pose log, :600 incrementAge -- the end of such a frame (instead of
 *                      mf_model_combined_predict per model + mf_end_frame) */
int mf_track_models(mf_ctx* ctx, int32_t first_model, int32_t track_all_models);
int mf_fuse_models(mf_ctx* ctx, int32_t first_model, float weight_multiplier, int32_t spawned_model);
int mf_predict_models(mf_ctx* ctx, int32_t first_model, int64_t timestamp);
/* Model::initialise (Core/Model/Model.h:126, Model.cpp:240-285) from the staged frame */
int mf_model_initialise(mf_ctx* ctx, int32_t model);
/* Model::overridePose (Core/Model/Model.h:235-238): lastPose = pose; pose = pose16 */
int mf_model_override_pose(mf_ctx* ctx, int32_t model, const float* pose16);
/* Model::computeFusionWeight(weightMultiplier) (Core/Model/Model.cpp:449-464) of the model's pose / lastPose.  Synchronous. */
int mf_model_fusion_weight(mf_ctx* ctx, int32_t model, float weight_multiplier, float* out);
/* Model::performTracking (Core/Model/Model.h:135-136, Model.cpp:427-447); the rgb texture is the staged frame's.  frame_to_frame_rgb
 * ("-ftf", GUI/MainController.cpp:252,539): initRGBModel takes the fill-in image -- the previous RAW frame when the last prediction ran
 * with mf_set_param("frameToFrameRGB", 1) (Model.cpp:981) -- instead of the model's RGB projection (Model.cpp:399-400); models that
 * allow no fill-in (objects) are unaffected, as upstream.  MF_ESTATE when a photometric term is requested on a context that built no
 * intensity / derivative images (icpWeight >= 100 and no rgbOnly). */
int mf_model_perform_tracking(mf_ctx* ctx, int32_t model, int32_t frame_to_frame_rgb, int32_t rgb_only, float icp_weight,
                              int32_t pyramid, int32_t fast_odom, int32_t so3, float max_depth_processed, int64_t log_timestamp,
                              int32_t try_fill_in);
/* Model::predictIndices(time, maxDepth, timeDelta) (Core/Model/Model.h:162, ModelProjection.cpp:100-152) */
int mf_model_predict_indices(mf_ctx* ctx, int32_t model, int32_t time, float max_depth, int32_t time_delta);
/* Model::fuse(time, rgb, mask, depthRaw, depthFiltered, depthCutoff, weightMultiplier) (Core/Model/Model.h:142-143,
 * Model.cpp:466-647); the four textures are the staged frame's */
int mf_model_fuse(mf_ctx* ctx, int32_t model, int32_t time, float depth_cutoff, float weight_multiplier);
/* Model::clean(time, graph, timeDelta, depthCutoff, isFern, depthFiltered, mask) (Core/Model/Model.h:146-147,
 * Model.cpp:649-772); no deformation graph, isFern = false (loop closure is dead code upstream) */
int mf_model_clean(mf_ctx* ctx, int32_t model, int32_t time, int32_t time_delta, float depth_cutoff);
/* Model::combinedPredict(maxDepth, time, maxTime, timeDelta, ACTIVE) (Core/Model/Model.h:158, ModelProjection.cpp:187-268);
 * time must equal max_time (the only form the reference calls, Core/MaskFusion.cpp:616-628) */
int mf_model_combined_predict(mf_ctx* ctx, int32_t model, float max_depth, int32_t time, int32_t max_time, int32_t time_delta);
/* No upstream twin (tests, tooling): replace the surfel buffer of `model` with `count` records in mf_download_map's layout.
 * Host pointer; synchronous. */
int mf_model_upload_map(mf_ctx* ctx, int32_t model, const float* surfels12, uint32_t count);
/* Model::makeNonStatic / makeStatic(globalPose) (Core/Model/Model.h:263-266): a non-static object model is tracked even with
 * trackAllModels off (Core/MaskFusion.cpp:263); makeStatic re-anchors it to the background's current pose */
int mf_make_nonstatic(mf_ctx* ctx, int32_t model);
int mf_make_static(mf_ctx* ctx, int32_t model);
/* MaskFusion::setTrackableClassIds (Core/MaskFusion.h:246, MaskFusion.cpp:261,940); n = 0: every class is trackable */
int mf_set_trackable_class_ids(mf_ctx* ctx, const int32_t* ids, int32_t n);
/* mf_debug_read for the per-model taps ("pred_vertex", "pred_normal", "pred_image", "pred_time") of model `model` */
int mf_debug_read_model(mf_ctx* ctx, int32_t model, const char* what, void* out, uint64_t out_bytes);

/* ------------------------------------------------------------------------------------------------
 * Model-sharded scenes: several contexts -- one per GPU -- each own some of the models of ONE scene.  What
 * MaskFusion::processFrame couples between models crosses the contexts through these calls: the z-merged model-id image
 * (GlobalProjection, Core/Model/GlobalProjection.cpp:43-114), the label image (Core/MaskFusion.cpp:289-297) and the background
 * pose (static objects follow it, Core/Model/Model.h:263-264; mf_model_override_pose(ctx, 0, pose) installs it on a context
 * that does not own the background).  maskfusion_amd/sharded.py sequences them with the Model-level calls above and with
 * collectives (all-reduce(MIN) of the keys, broadcast of labels + pose + control record, gather of per-model state).
 * ---------------------------------------------------------------------------------------------- */
/* GlobalProjection::project of this context's models only.  d_keys_out: width*height uint64 = float_bits(z) << 32 | order << 8 | id,
 * all ones = empty; the per-pixel minimum over contexts is the merged image.  orders[i]: position of local model i in the GLOBAL
 * model list (the GL draw order that breaks exact depth ties), < 0: do not draw it (a background stand-in). */
int mf_export_projection_keys_dev(mf_ctx* ctx, const int32_t* orders, int32_t n_orders, uint64_t* d_keys_out);
/* GlobalProjection::downloadDirect of a merged key image: sets the projected-id image of the next mf_perform_segmentation */
int mf_import_projection_keys_dev(mf_ctx* ctx, const uint64_t* d_keys);
/* MaskFusion::performSegmentation (Core/MaskFusion.h:59; MfSegmentation.cpp:83-538) on the staged frame.  mask: host, H*W mask ids
 * or NULL.  model_ids NULL: this context's model list; else the GLOBAL list in order (index 0 = background) with next_model_id
 * = MaskFusion::getNextModelID().  Result -> textureMask; *has_new_label / *new_class_id = SegmentationResult::hasNewLabel and
 * the class of the new label.  Synchronous. */
int mf_perform_segmentation(mf_ctx* ctx, const uint8_t* mask, const int32_t* class_ids, int32_t n_masks, const int32_t* model_ids,
                            const int32_t* model_class_ids, int32_t n_models, int32_t next_model_id, int32_t allow_new,
                            int32_t* has_new_label, int32_t* new_class_id);
/* The same in two halves (what mf_process_frame does with "earlyBackgroundFusion"): _begin enqueues the label stage and returns; the caller
 * may then enqueue work that does not depend on the decision -- mf_fuse_background: the background is never spawned or dropped and its
 * fusion (Core/MaskFusion.cpp:539-565 for models.front()) reads only the label image, complete on the stream -- and _end waits for the
 * label stage alone.  mf_fuse_models skips a background that mf_fuse_background has already fused for the staged frame. */
int mf_perform_segmentation_begin(mf_ctx* ctx, const uint8_t* mask, const int32_t* class_ids, int32_t n_masks, const int32_t* model_ids,
                                  const int32_t* model_class_ids, int32_t n_models, int32_t next_model_id, int32_t allow_new);
int mf_perform_segmentation_end(mf_ctx* ctx, int32_t* has_new_label, int32_t* new_class_id);
int mf_fuse_background(mf_ctx* ctx, float weight_multiplier);
/* SegmentationResult::fullSegmentation as device memory: out of the context that ran the label stage, into the others
 * (textureMask->Upload, Core/MaskFusion.cpp:297) */
int mf_export_segmentation_dev(mf_ctx* ctx, uint8_t* d_out);
int mf_import_segmentation_dev(mf_ctx* ctx, const uint8_t* d_in);
/* MaskFusion::spawnObjectModel (Core/MaskFusion.cpp:671-684) with the id the label-stage owner allocated */
int mf_spawn_object_model(mf_ctx* ctx, int32_t id, int32_t class_id);
/* MaskFusion::inactivateModel (Core/MaskFusion.cpp:686-713) */
int mf_drop_model(mf_ctx* ctx, int32_t model);
/* Model::updateStaticPose(globalPose) (Core/Model/Model.h:263) with this context's background pose */
int mf_model_update_static_pose(mf_ctx* ctx, int32_t model);
/* Model::setMaxDepth + the object confidence ramp of processFrame (Core/MaskFusion.cpp:335-339,369-374) for the local objects */
int mf_update_object_params(mf_ctx* ctx);

/* ------------------------------------------------------------------------------------------------
 * Kernel-level entry points (device pointers, launched on `stream`, asynchronous).  Each replaces one reference
 * CUDA wrapper or GLSL pass; the parity tests call them one by one against oracle/.
 * ---------------------------------------------------------------------------------------------- */
/* MaskFusion::filterDepth + depth_bilateral_metric.frag (Core/MaskFusion.cpp:650-657) */
int mf_k_bilateral(const float* d_depth, float* d_out, int32_t W, int32_t H, void* stream);
/* pyrDownGaussF (Core/Cuda/cudafuncs.cu:510-532) */
int mf_k_pyrdown_f(const float* d_src, float* d_dst, int32_t sw, int32_t sh, void* stream);
/* createVMap + createNMap (Core/Cuda/cudafuncs.cu:136-150,191-205), one level; planar [3][H][W] outputs */
int mf_k_vmap_nmap(const float* d_depth, float* d_vmap, float* d_nmap, int32_t W, int32_t H, float fx, float fy,
                   float cx, float cy, float depth_cutoff, void* stream);
/* RGBDOdometry::initICPModel (Core/Utils/RGBDOdometry.cpp:153-185): copyMaps + resizeVMap/NMap x2 + tranformMaps x3.
 * d_v4/d_n4: H*W float4 predictions; outputs: 3 levels planar, packed level after level; R row-major 3x3 */
int mf_k_model_pyramid(const float* d_v4, const float* d_n4, const float* R9, const float* t3, float* d_vmaps,
                       float* d_nmaps, int32_t W, int32_t H, void* stream);
/* computeGeometricSegmentationMap -> thresholdMap -> morphGeometricSegmentationMap -> invertMap
 * (Core/Cuda/segmentation.cu:277-354; MfSegmentation.cpp:149-207).  d_vmap/d_nmap: level-0 planar maps;
 * d_edge: H*W f32 out; d_binary: H*W u8 out (255 = not an edge); d_tmp: H*W u8 scratch */
int mf_k_geometric_edges(const float* d_vmap, const float* d_nmap, float* d_edge, uint8_t* d_binary, uint8_t* d_tmp,
                         int32_t W, int32_t H, float w_distance, float w_convexity, float threshold, int32_t morph_radius,
                         int32_t morph_iterations, void* stream);
/* Host half of MfSegmentation::performSegmentation (Core/Segmentation/MfSegmentation.cpp:220-522): HOST pointers, no GPU
 * involved (the reference runs this stage on the CPU too).  binary: 255 = not an edge; params: {threshold, weightDistance,
 * weightConvexity, morphEdgeIterations, morphEdgeRadius, morphMaskIterations, morphMaskRadius, removeEdges,
 * minRelSizeNew, maxRelSizeNew, personClassID} (MfSegmentation.h:42-62); ignore_map: persistent H*W in/out. */
int mf_segmentation_labels(int32_t W, int32_t H, const uint8_t* binary, const float* depth, const uint8_t* mask,
                           const int32_t* class_ids, int32_t n_masks, const uint8_t* projected_ids, const int32_t* model_ids,
                           const int32_t* model_class_ids, int32_t n_models, int32_t next_model_id, int32_t allow_new,
                           const float* params11, uint8_t* ignore_map, uint8_t* full_segmentation, int32_t* has_new_label,
                           int32_t* new_class_id);
/* The same stage on the device (what mf_process_frame uses unless mf_set_param("gpuLabels", 0)): identical
 * arguments and results, HOST pointers (staged internally). */
int mf_k_segmentation_labels(int32_t W, int32_t H, const uint8_t* binary, const float* depth, const uint8_t* mask,
                           const int32_t* class_ids, int32_t n_masks, const uint8_t* projected_ids, const int32_t* model_ids,
                           const int32_t* model_class_ids, int32_t n_models, int32_t next_model_id, int32_t allow_new,
                           const float* params11, uint8_t* ignore_map, uint8_t* full_segmentation, int32_t* has_new_label,
                           int32_t* new_class_id);
/* imageBGRToIntensity (Core/Cuda/cudafuncs.cu:626-654); channels = 3 or 4, the first three are used as stored */
int mf_k_intensity(const uint8_t* d_img, int32_t channels, uint8_t* d_out, int32_t n, void* stream);
/* pyrDownUcharGauss (Core/Cuda/cudafuncs.cu:534-588) */
int mf_k_pyrdown_u8(const uint8_t* d_src, uint8_t* d_dst, int32_t sw, int32_t sh, void* stream);
/* computeDerivativeImages (Core/Cuda/cudafuncs.cu:658-718) */
int mf_k_derivative_images(const uint8_t* d_src, int16_t* d_dx, int16_t* d_dy, int32_t W, int32_t H, void* stream);
/* The SO(3) block of RGBDOdometry::getIncrementalTransformation (Core/Utils/RGBDOdometry.cpp:264-324) with so3Step
 * (Core/Cuda/reduce.cu:999-1202) inside: level-2 images and intrinsics in, resultR (row-major double, host) and
 * {lastSO3Error, lastSO3Count, iterations} out.  Synchronous. */
int mf_k_so3_prealign(const uint8_t* d_last, const uint8_t* d_next, int32_t W, int32_t H, float fx, float fy, float cx,
                      float cy, double* R9, float* stats3, void* stream);
/* computeRgbResidual (Core/Cuda/reduce.cu:774-997).  d_corres: W*H records {int16 u0, int16 v0, float diff}
 * (u0 < 0: none) = DataTerm without the redundant fields; count_sigma2 (host) = {count, sum diff^2 as int32}.
 * Synchronous. */
int mf_k_rgb_residual(float min_scale, const int16_t* d_dIdx, const int16_t* d_dIdy, const float* d_last_depth,
                      const float* d_next_depth, const uint8_t* d_last_image, const uint8_t* d_next_image,
                      float max_depth_delta, const float* kt3, const float* krkinv9, int32_t W, int32_t H, void* d_corres,
                      int32_t* count_sigma2, void* stream);
/* rgbStep (Core/Cuda/reduce.cu:529-713) with projectToPointCloud (Core/Cuda/cudafuncs.cu:722-751) evaluated on the
 * fly from d_last_depth; out32 (host, double) = {27 upper-tri products, row6^2, inliers, pad}.  Synchronous. */
int mf_k_rgb_step(const void* d_corres, float sigma, const float* d_last_depth, float fx, float fy, float cx, float cy,
                  const int16_t* d_dIdx, const int16_t* d_dIdy, float sobel_scale, int32_t W, int32_t H, double* out32,
                  void* stream);
/* One Gauss-Newton update exactly as the iteration kernels run it: the host side of getIncrementalTransformation
 * (Core/Utils/RGBDOdometry.cpp:428-474: Eigen LDLT solve of the 6x6 system in double, OdometryProvider::computeUpdateSE3 /
 * rodrigues, Core/Utils/OdometryProvider.h:32-90, currentT = [Rprev|tprev] * transform^-1) moved to the device.  HOST pointers.
 * sys29: the 27 upper-triangle products of the 7-vector row in reduce.cu:378-411 order, then sum r^2, then the inlier count;
 * result_rt16: resultRt before the step (row-major 4x4 double); Rprev9 / tprev3: the model pose the frame is tracked against.
 * Out: x6_serial (the production one-thread LDL^T), x6_wave (the wave-parallel Gauss-Jordan the RGB-D kernels use), the updated
 * resultRt, Rcurr / tcurr, stats2 = {lastICPError = sqrt(sum r^2) / inliers, lastICPCount}.  Synchronous. */
int mf_k_gn_solve(const double* sys29, const double* result_rt16, const float* Rprev9, const float* tprev3, double* x6_serial,
                  double* x6_wave, double* result_rt16_out, float* Rcurr9, float* tcurr3, float* stats2, void* stream);
/* icpStep (Core/Cuda/reduce.cu:446-525): d_out32 receives {27 upper-tri products, sum r^2, inliers, pad} */
int mf_k_icp_step(const float* Rcurr9, const float* tcurr3, const float* d_vmap_curr, const float* d_nmap_curr,
                  const float* Rprev_inv9, const float* tprev3, float fx, float fy, float cx, float cy,
                  const float* d_vmap_g_prev, const float* d_nmap_g_prev, float dist_thresh, float angle_thresh,
                  int32_t W, int32_t H, float* d_out32, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MASKFUSION_AMD_H_ */

/*
 * mf_oracle.h -- CPU restatement of the MaskFusion::processFrame hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under maskfusion_amd/ (the product) may include,
 * link or call this.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg use it, and only as the checker / the timed CPU baseline.
 *
 * PARITY PIN: the reference (martinruenz/maskfusion) ships no tests, golden vectors or fixtures for this path and its
 * build (CUDA + OpenGL 4.3 + Pangolin + OpenCV + Eigen) cannot run here.  Five parts:
 *   - everything restated from Core/Cuda/{reduce,cudafuncs,segmentation}.cu (rows a3-a5, a7-a10, a12 and the device half
 *     of a20 in SURVEY.md section 8) IS pinned: oracle/build_ref.py compiles those translation units for the CPU
 *     (oracle/_ref/libmf_ref.so), tests/golden/ref_vectors.npz holds their outputs on seeded inputs, and
 *     tests/test_ref_pin.py requires this file to reproduce them (bit-exact for integer / per-pixel float results,
 *     1e-5 for reduced sums whose summation order is a launch-shape detail);
 *   - everything restated from GLSL shaders (a2, a13-a19, a21) IS pinned to the shaders' own text: oracle/build_glsl.py compiles
 *     the .vert and .frag files of Core/Shaders as C++ (oracle/_ref/libmf_glsl.so, mechanical edits only), tests/golden/glsl_vectors.npz holds
 *     what they compute on seeded inputs and tests/test_glsl_pin.py requires this file to reproduce it -- bit-exact for every pass
 *     (the clean pass in the literal window mode, see mfo_set_window_literal).  What OpenGL does AROUND a shader (texel selection,
 *     point / sprite coverage, depth test) is a documented rule set, not reference-executed;
 *   - the label-propagation logic of the host half of a20 IS pinned to MfSegmentation.cpp:219-523 compiled from the reference's text
 *     (oracle/build_seg.py, tests/test_seg_pin.py: every pixel identical);
 *   - the host loop of the tracker (a6, a11: SO(3) pre-alignment, three-level ICP + photometric Gauss-Newton, joint solve, SE(3)
 *     update, 0.3 m rule) IS pinned to RGBDOdometry.cpp:227-497 + OdometryProvider.h compiled from the reference's text over the
 *     CPU-compiled device functions (oracle/build_track.py, tests/test_track_pin.py: poses within 1e-6 in every branch, inlier
 *     counts equal); Eigen underneath is a fixed-size stand-in (oracle/eigen_shim);
 *   - the host-side arithmetic inside absent third-party libraries -- Eigen (LDLT, JacobiSVD, Quaternion: a6, a11, a15) and the
 *     OpenCV primitives under a20 (connected components, morphology) -- is PARITY UNPINNED by reference-executed code: restated from the
 *     published algorithms and checked by analytic known-answer tests (tests/test_oracle_kat.py, tests/test_oracle_rgbd_kat.py)
 *     and against numpy / SciPy.
 * Citations are relative to /root/reference/.
 *
 * Conventions
 *   - images: dense row-major, no pitch.  vmap/nmap: planar [3][H][W] float (x,y,z planes),
 *     invalid <=> x plane is NaN (the reference leaves y stale / z=0 there; unobservable).
 *   - 4x4 poses at the API: 16 floats COLUMN-major (Eigen::Matrix4f::data() compatible).
 *   - surfel record = 12 floats {px,py,pz,conf | colorEnc,unused,initTime,lastTime | nx,ny,nz,radius}
 *     (Core/Shaders/Vertex.cpp:21-43).
 */
#ifndef MF_ORACLE_H_
#define MF_ORACLE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------- image preprocessing (a2, a3) ---------------- */
/* Core/Shaders/depth_bilateral_metric.frag:30-76 */
void mfo_bilateral(const float* depth, float* out, int W, int H);
/* Core/Cuda/cudafuncs.cu:333-364 (pyrDownKernelGaussF); dst is (sh/2) x (sw/2) */
void mfo_pyrdown_gauss_f(const float* src, float* dst, int sw, int sh);
/* Core/Cuda/cudafuncs.cu:534-564 (pyrDownKernelIntensityGauss) */
void mfo_pyrdown_gauss_u8(const uint8_t* src, uint8_t* dst, int sw, int sh);
/* Core/Cuda/cudafuncs.cu:109-134 (computeVmapKernel) */
void mfo_create_vmap(const float* depth, float* vmap, int W, int H,
                     float fx, float fy, float cx, float cy, float depthCutoff);
/* Core/Cuda/cudafuncs.cu:152-189 (computeNmapKernel) */
void mfo_create_nmap(const float* vmap, float* nmap, int W, int H);

/* ---------------- model-side maps (a4) ---------------- */
/* Core/Cuda/cudafuncs.cu:271-311 (copyMapsKernel): AoS float4 -> planar, z==0 -> NaN */
void mfo_copy_maps(const float* v4, const float* n4, float* vmap, float* nmap, int W, int H);
/* Core/Cuda/cudafuncs.cu:366-417 (resizeMapKernel<normalize>) */
void mfo_resize_map(const float* in, float* out, int sw, int sh, int normalize);
/* Core/Cuda/cudafuncs.cu:207-249 (tranformMapsKernel), in place allowed; R row-major */
void mfo_transform_maps(const float* vsrc, const float* nsrc, const float* R, const float* t,
                        float* vdst, float* ndst, int W, int H);

/* ---------------- odometry (a6, a7, a11) ---------------- */
/* Core/Cuda/reduce.cu:259-525 (ICPReduction + icpStep host unpack).
 * Rcurr, Rprev_inv row-major 3x3.  A: 36 floats row-major, b: 6, residual: {sum r^2, inliers}. */
void mfo_icp_step(const float* Rcurr, const float* tcurr,
                  const float* vmap_curr, const float* nmap_curr,
                  const float* Rprev_inv, const float* tprev,
                  float fx, float fy, float cx, float cy,
                  const float* vmap_g_prev, const float* nmap_g_prev,
                  float distThres, float angleThres, int W, int H,
                  float* A, float* b, float* residual);
/* cv::morphologyEx(MORPH_CLOSE, ellipse) stand-in, in place (for oracle/build_seg.py's compiled slice of MfSegmentation.cpp) */
void mfo_morph_close_ellipse(uint8_t* img, int W, int H, int radius, int iterations);
/* pose.inverse() as every projection pass consumes it (t_inv uniform); column-major 4x4 */
void mfo_pose_inverse16(const float* pose16, float* out16);
/* ANALYSIS ONLY (tools/window_ambiguity.py): literal fp32 reading of the association / clean window loops; see mf_oracle.c */
void mfo_set_window_literal(int on);
/* clean pass only, default ON: the window of copy_unstable.vert with the shader text's fp32 trip count (4 or 5 taps per axis) */
void mfo_set_clean_literal(int on);
/* Eigen LDLT stand-in (RGBDOdometry.cpp:447-459): symmetric solve in double, n = 3 or 6.
 * Returns 0 on success. */
int mfo_ldlt_solve(const double* A, const double* b, double* x, int n);
/* Core/Utils/OdometryProvider.h:32-67; R row-major 3x3 */
void mfo_rodrigues(const double* w, double* R);
/* Core/Utils/OdometryProvider.h:69-90: resultRt <- exp(x) * resultRt (row-major 4x4 double) */
void mfo_update_se3(double* resultRt, const double* x6);

/* Tracker options (Core/MaskFusion.cpp:57-66, RGBDOdometry.cpp:327-329) */
typedef struct {
    int   pyramid;      /* 1 */
    int   fastOdom;     /* 0 */
    int   so3;          /* 1: SO(3) photometric pre-alignment at level 2 */
    int   rgbOnly;      /* 0 */
    float icpWeight;    /* 10 core default, 20 GUI; >=100 => ICP only */
    float distThresh;   /* 0.10 */
    float angleThresh;  /* sin(20 deg) */
} mfo_track_opts;

/* Per-iteration log for differential tests */
typedef struct {
    int    n_iters;
    float  A[19][36];
    float  b[19][6];
    float  residual[19][2];
    double x[19][6];
} mfo_track_log;

/* Core/Utils/RGBDOdometry.cpp:227-497, ICP-only branch (icp && !rgb).
 * curr_v/curr_n: 3 levels of current-frame maps (camera frame); prev_v/prev_n: 3 levels of model maps in the
 * global frame.  R (row-major), t: pose in/out.  out_inc16: returned increment (col-major).
 * Fills lastICPError/lastICPCount.  log may be NULL. */
void mfo_track_icp(const float* const curr_v[3], const float* const curr_n[3],
                   const float* const prev_v[3], const float* const prev_n[3],
                   int W, int H, float fx, float fy, float cx, float cy,
                   const mfo_track_opts* opts, float* R, float* t, float* out_inc16,
                   float* lastICPError, float* lastICPCount, mfo_track_log* log);

/* ---------------- photometric term + SO(3) pre-alignment (a5, a8-a10, a12) ---------------- */
typedef struct { int16_t zx, zy, ox, oy; float diff; int32_t valid; } mfo_dataterm;   /* DataTerm, types.cuh:75-81 */
typedef struct {
    const float* lastDepth[3]; const float* nextDepth[3];       /* populateRGBDData pyramids (NaN = invalid) */
    const uint8_t* lastImage[3]; const uint8_t* nextImage[3];
    const uint8_t* lastNextImage2;                               /* level-2 intensity of the previous frame (so3) */
} mfo_rgbd_inputs;
typedef struct {
    float lastICPError, lastICPCount, lastRGBError, lastRGBCount, lastSO3Error, lastSO3Count;
    int so3Iterations, iterationsRun, rejected;
} mfo_track_stats;
/* cudafuncs.cu:602-614 */
void mfo_vertices_to_depth(const float* v4, float* depth, int n, float cutOff);
/* cudafuncs.cu:626-639; channels = 3 (frame) or 4 (predicted / fill-in image) */
void mfo_image_to_intensity(const uint8_t* img, int channels, uint8_t* dst, int n);
/* cudafuncs.cu:658-718 */
void mfo_derivative_images(const uint8_t* src, int16_t* dx, int16_t* dy, int W, int H);
/* cudafuncs.cu:722-751; cloud3: 3 floats per pixel */
void mfo_project_to_cloud(const float* depth, float* cloud3, int W, int H, float fx, float fy, float cx, float cy);
/* reduce.cu:774-997 */
void mfo_rgb_residual(float minScale, const int16_t* dIdx, const int16_t* dIdy, const float* lastDepth,
                      const float* nextDepth, const uint8_t* lastImage, const uint8_t* nextImage, mfo_dataterm* corres,
                      float maxDepthDelta, const float* kt, const float* krkinv, int W, int H, int32_t* sigmaSum,
                      int32_t* count);
/* reduce.cu:529-713; A 6x6 row-major, b 6 */
void mfo_rgb_step(const mfo_dataterm* corres, float sigma, const float* cloud3, float fx, float fy, const int16_t* dIdx,
                  const int16_t* dIdy, float sobelScale, int W, int H, float* A, float* b);
/* reduce.cu:999-1202; A 3x3 row-major, b 3, residual {sum r^2, inliers} */
void mfo_so3_step(const uint8_t* lastImage, const uint8_t* nextImage, const float* imageBasis, const float* kinv,
                  const float* krlr, int W, int H, float* A, float* b, float* residual);
void mfo_ldlt3f_solve(const float* A, const float* b, float* x);
/* RGBDOdometry.cpp:264-324 */
void mfo_so3_prealign(const uint8_t* lastNext2, const uint8_t* next2, int W2, int H2, float fx2, float fy2, float cx2,
                      float cy2, double* resultR, float* lastSO3Error, float* lastSO3Count, int* iterations_run);
/* RGBDOdometry.cpp:227-497, every branch (icp / rgb / rgbOnly / so3) */
void mfo_track_rgbd(const float* const curr_v[3], const float* const curr_n[3], const float* const prev_v[3],
                    const float* const prev_n[3], const mfo_rgbd_inputs* in, int W, int H, float fx, float fy, float cx,
                    float cy, const mfo_track_opts* opts, float* R, float* t, float* out_inc16, mfo_track_stats* stats);
/* RGBDOdometry.cpp:187-204 */
void mfo_populate_rgbd(const float* v4, const uint8_t* img, int channels, int W, int H, float* depth[3], uint8_t* image[3]);

/* ---------------- surfels (a13..a18, a21) ---------------- */
typedef struct {
    int W, H;
    float fx, fy, cx, cy;
} mfo_cam;

/* exp() / acos() as both sides of the parity tests evaluate them for surfels.glsl:44 and data.vert:167 (<= 2 ulp; see mf_oracle.c) */
float mfo_shader_exp(float x);
float mfo_shader_acos(float x);
float mfo_encode_color(float r, float g, float b);            /* color_encoding.glsl:19-25 */
void  mfo_decode_color(float c, float* rgb);                    /* color_encoding.glsl:27-34 */
float mfo_get_radius(float depth, float norm_z, float fx, float fy); /* surfels.glsl:19-34 */
float mfo_confidence(float x, float y, float weighting, float cx, float cy); /* surfels.glsl:36-46 */

/* First frame (MaskFusion.cpp:235-238; vertex_feedback.vert/.geom; init_unstable.vert; Model.cpp:240-285).
 * Column-major pixel order; returns count.  maxDepth = maxDepthProcessed (20). */
int mfo_init_surfels(const mfo_cam* cam, const uint8_t* rgb, const float* depthRaw, const float* depthF,
                     int tick, float maxDepth, float* surfels, int capacity);

/* index_map.vert/.frag + ModelProjection.cpp:100-152.  pose16 col-major (model pose).
 * Outputs: index (int32, 0 = empty), vertConf/colorTime/normRad float4 maps (camera frame). */
void mfo_predict_indices(const mfo_cam* cam, const float* pose16, const float* surfels, int count,
                         int time, float maxDepth, int timeDelta,
                         int32_t* index, float* vertConf, float* colorTime, float* normRad);

/* data.vert + Model.cpp:466-581.  Writes per-candidate records: cand_op[c] in {0,1,2}, cand_best[c] (surfel idx),
 * cand_rec[c][12].  Candidate c = xi*(H/2... see mf_oracle.c) enumerates quarter-rate pixels in column-major order. */
void mfo_fuse_data(const mfo_cam* cam, const float* pose16, const uint8_t* rgb, const float* depthRaw,
                   const float* depthF, const uint8_t* mask, int maskID, int time, float weighting, float maxDepth,
                   const int32_t* index, const float* vertConf, const float* normRad,
                   uint8_t* cand_op, int32_t* cand_best, float* cand_rec, int* n_cand);

/* update.vert + Model.cpp:583-646: applies first-writer-wins merge records to every surfel (src -> dst). */
void mfo_fuse_update(const float* src, float* dst, int count, int time,
                     const uint8_t* cand_op, const int32_t* cand_best, const float* cand_rec, int n_cand);

/* copy_unstable.vert:53-157 + .geom + Model.cpp:649-772.  Old surfels then new (op==2) candidates; returns new count. */
int mfo_clean(const mfo_cam* cam, const float* pose16, const float* src, int count,
              const uint8_t* cand_op, const float* cand_rec, int n_cand,
              int time, int timeDelta, float confThreshold, float maxDepth, float outlierCoeff, int maskID,
              const int32_t* index, const float* vertConf, const float* colorTime, const float* normRad,
              const float* depthF, const uint8_t* mask, float* dst, int capacity);

/* splat.vert + combo_splat.frag + ModelProjection.cpp:187-268.
 * Outputs: image RGBA8, vertexConf float4, normalRad float4, time uint16 (0 where empty). */
void mfo_combined_predict(const mfo_cam* cam, const float* pose16, const float* surfels, int count,
                          float maxDepth, float confThreshold, int time, int maxTime, int timeDelta,
                          uint8_t* image, float* vertexConf, float* normalRad, uint16_t* timeMap);

/* fill_vertex/normal/rgb.frag + FillIn.cpp (passthrough = 0|1) */
void mfo_fill_in(const mfo_cam* cam, const uint8_t* predImage, const float* predVertex, const float* predNormal,
                 const uint8_t* rawRgb, const float* rawDepth, int passthrough,
                 uint8_t* fillImage, float* fillVertex, float* fillNormal);

/* MaskFusion.cpp:630-648 (nearest sample of the predicted colour image at 20x down-sampling) */
int mfo_requires_fill_in(const uint8_t* predImage, int W, int H, float ratio);

/* Model.cpp:449-464 + rodrigues2 :891-932.  mfo_set_weight_literal(1): the log map with the reference's float trace (finding F5) */
void mfo_set_weight_literal(int on);
float mfo_fusion_weight(const float* pose16, const float* lastPose16, float weightMultiplier);

/* ---------------- single-model pipeline (a1) ---------------- */
typedef struct mfo_ctx mfo_ctx;

typedef struct {
    int   W, H;
    float fx, fy, cx, cy;
    int   timeDelta;          /* 200 core / INT_MAX/2 open loop */
    float confGlobal;         /* 4 core / 10 GUI */
    float depthCutoff;        /* 3 core / 4 GUI */
    float icpWeight;          /* >=100: ICP only */
    float maxDepthProcessed;  /* 20 */
    float outlierCoeff;       /* 0.9 core / 0.1 GUI */
    int   fastOdom, pyramid, so3;
    int   capacity;           /* max surfels */
    int   rgbOnly;            /* 0 */
} mfo_config;

void     mfo_default_config(mfo_config* c, int W, int H, float fx, float fy, float cx, float cy);
mfo_ctx* mfo_create(const mfo_config* c);
void     mfo_destroy(mfo_ctx* ctx);
/* MaskFusion::processFrame, -static single background model (Core/MaskFusion.cpp:200-607). */
int      mfo_process_frame(mfo_ctx* ctx, const uint8_t* rgb, const float* depth, float weightMultiplier);
int      mfo_process_frame_ex(mfo_ctx* ctx, const uint8_t* rgb, const float* depth, float weightMultiplier,
                              const float* inPose16 /*column-major or NULL*/, int bootstrap);
/* test isolation: the next mfo_process_frame uses depthF (W*H) as the bilateral filter's output */
void     mfo_override_filtered_depth(mfo_ctx* ctx, const float* depthF);
/* MaskFusion::setFrameToFrameRGB ("-ftf"; Model.cpp:399-400,981) */
void     mfo_set_frame_to_frame_rgb(mfo_ctx* ctx, int on);
void     mfo_get_pose(const mfo_ctx* ctx, float* pose16);
int      mfo_get_count(const mfo_ctx* ctx);
int      mfo_get_tick(const mfo_ctx* ctx);
const float* mfo_get_surfels(const mfo_ctx* ctx);
void     mfo_get_icp_stats(const mfo_ctx* ctx, float* err, float* count);
void     mfo_get_track_stats(const mfo_ctx* ctx, mfo_track_stats* out);
/* iterations of the LAST mfo_track_icp call whose system was outside the solver's stated domain (< 6 inliers or cond(A) > 1e8: finding F4) */
int      mfo_last_track_ill(void);
int      mfo_last_track_log(float* out_20x32);   /* reduced geometric systems of the last tracking step, one row per iteration (device log layout) */
/* per-stage wall-clock (ms) accumulated since create: order = preprocess, odomInit, odom, indexMap, fuseData,
 * fuseUpdate, clean, predict */
void     mfo_get_timings(const mfo_ctx* ctx, double* ms8);
/* intermediate buffers for differential tests (valid after process_frame) */
const float*   mfo_dbg_depthF(const mfo_ctx* ctx);
const float*   mfo_dbg_pred_vertex(const mfo_ctx* ctx);
const float*   mfo_dbg_pred_normal(const mfo_ctx* ctx);
const uint8_t* mfo_dbg_pred_image(const mfo_ctx* ctx);
int            mfo_dbg_last_fillin(const mfo_ctx* ctx);

#ifdef __cplusplus
}
#endif

/* =====================================================================================================
 * Multi-model path: GlobalProjection (a19), MfSegmentation (a20), object-model life cycle (a1 multi-model branch)
 * ===================================================================================================== */
#ifdef __cplusplus
extern "C" {
#endif

/* Core/Cuda/segmentation.cu:122-177 (computeGeometricSegmentation_Kernel); vmap/nmap level 0, planar */
void mfo_geometric_edge_map(const float* vmap, const float* nmap, float* out, int W, int H, float wD, float wC);
/* segmentation.cu:257-262, 264-269 */
void mfo_threshold_map(const float* in, uint8_t* out, int n, float threshold);
void mfo_invert_map(const uint8_t* in, uint8_t* out, int n);
/* segmentation.cu:217-255, 334-354: iterations x (dilate -> erode), square radius; in place on data (buffer = scratch) */
void mfo_morph_closing_u8(uint8_t* data, uint8_t* buffer, int W, int H, int radius, int iterations);

/* GlobalProjection::project + downloadDirect (Core/Model/GlobalProjection.cpp:43-114; splat_models.vert,
 * combo_splat_models.frag).  Models in list order; ids: uint8 per pixel (0 where nothing projects). */
typedef struct {
    const float* surfels; int count; const float* pose16; int id;
} mfo_model_view;
void mfo_global_projection(const mfo_cam* cam, const mfo_model_view* models, int n_models, int time, int maxTime,
                           int timeDelta, float depthCutoff, uint8_t* ids);

/* 4-connected component labelling of a binary (0 / non-zero) image, labels numbered in raster order of each
 * component's first pixel, 0 = background (stand-in for cv::connectedComponentsWithStats(..., 4),
 * MfSegmentation.cpp:239).  stats: n_components x 5 ints {left, top, width, height, area}.  Returns n (incl. bg). */
int mfo_connected_components4(const uint8_t* bin, int32_t* labels, int32_t* stats, int max_comp, int W, int H);

typedef struct {
    float threshold, weightDistance, weightConvexity;   /* 0.1 / 1 / 1   (MfSegmentation.h:48-50) */
    int morphEdgeIterations, morphEdgeRadius;             /* 3 / 1 */
    int morphMaskIterations, morphMaskRadius;             /* 3 / 1 */
    int removeEdges;                                      /* 1 */
    float minRelSizeNew, maxRelSizeNew;                   /* 0.07 / 0.4 (SegmentationPerformer.h:41-42) */
    int personClassID;                                    /* 255 */
} mfo_seg_params;
void mfo_default_seg_params(mfo_seg_params* p);

/* MfSegmentation::performSegmentation, CPU stage (Core/Segmentation/MfSegmentation.cpp:220-522).
 * binary: output of threshold->morph->invert (255 = not an edge).  depth: raw frame depth.  mask/classIDs: the frame's
 * Mask R-CNN output (nMasks = number of class ids; mask values index classIDs).  projectedIDs: GlobalProjection ids.
 * modelIDs/modelClassIDs: the model list in order (index 0 = background).  Outputs: fullSegmentation (model id per
 * pixel, 255 = ignore), *hasNewLabel, *newClassID.  ignoreMap is the persistent semanticIgnoreMap (in/out). */
void mfo_mf_segmentation_cpu(const mfo_seg_params* prm, int W, int H, const uint8_t* binary, const float* depth,
                             const uint8_t* mask, const int32_t* classIDs, int nMasks, const uint8_t* projectedIDs,
                             const int32_t* modelIDs, const int32_t* modelClassIDs, int nModels, int nextModelID,
                             int allowNew, uint8_t* ignoreMap, uint8_t* fullSegmentation, int* hasNewLabel,
                             int* newClassID);

/* Multi-model MaskFusion::processFrame (Core/MaskFusion.cpp:200-607 with enableMultipleModels) */
typedef struct mfo_mm mfo_mm;
typedef struct {
    mfo_config base;             /* camera + tracker + background-model settings */
    float confObject;            /* initConfidenceObject = 2 (ramped: min(4.5, age/25), MaskFusion.cpp:369-374) */
    int capacityObject;          /* per object model */
    int trackAllModels;          /* MaskFusion::trackAllModels */
    int modelSpawnOffset;        /* 20 */
    int maxModels;
    mfo_seg_params seg;
} mfo_mm_config;
void    mfo_mm_default_config(mfo_mm_config* c, int W, int H, float fx, float fy, float cx, float cy);
mfo_mm* mfo_mm_create(const mfo_mm_config* c);
void    mfo_mm_destroy(mfo_mm* x);
/* test isolation (see mfo_override_filtered_depth): the NEXT mfo_mm_process_frame takes this image as the bilateral filter's output */
void    mfo_mm_override_filtered_depth(mfo_mm* x, const float* depthF);
void    mfo_mm_set_frame_to_frame_rgb(mfo_mm* x, int on);
/* Model::fuse's bb_max_z (Model.cpp:480-501) from the box the GUI's render pass leaves (Model.cpp:287-346); default on */
void    mfo_mm_set_bbox_limit(mfo_mm* x, int on);
/* MaskFusion::setTrackableClassIds (MaskFusion.cpp:261,940); Model::makeNonStatic / makeStatic / isNonstatic (Model.h:264-268) by list position */
void    mfo_mm_set_trackable_class_ids(mfo_mm* x, const int32_t* ids, int n);
void    mfo_mm_make_nonstatic(mfo_mm* x, int i);
void    mfo_mm_make_static(mfo_mm* x, int i);
int     mfo_mm_is_nonstatic(const mfo_mm* x, int i);
int     mfo_mm_model_class(const mfo_mm* x, int i);
/* teacher forcing (test isolation): see mf_oracle.c */
void    mfo_mm_force_tracking(mfo_mm* x, const int32_t* ids, const float* poses16, int n);
void    mfo_mm_model_tracked_pose(const mfo_mm* x, int i, float* pose16, int* tracked);
int     mfo_mm_model_track_log(const mfo_mm* x, int i, float* out_20x32);
void    mfo_mm_model_tracked_pose_alt(const mfo_mm* x, int i, float* pose16);   /* the same step from a start pose shifted by 1e-6 m */
int     mfo_mm_process_frame(mfo_mm* x, const uint8_t* rgb, const float* depth, const uint8_t* mask,
                             const int32_t* classIDs, int nMasks, float weightMultiplier);
/* test tooling: per-iteration teacher forcing of the tracked models' Gauss-Newton loops (see mf_oracle.c) */
void    mfo_mm_set_probe_poses(mfo_mm* x, const int32_t* ids, const float* poses /* [n][20][12] */, int n);
int     mfo_mm_model_probe_log(const mfo_mm* x, int i, float* out /* [20][32] */);
/* test tooling: replace model i's surfel buffer (twin of mf_model_upload_map); 0 or -1 */
int     mfo_mm_upload_map(mfo_mm* x, int i, const float* surfels12, int count);
int     mfo_mm_num_models(const mfo_mm* x);
int     mfo_mm_model_id(const mfo_mm* x, int i);
int     mfo_mm_model_count(const mfo_mm* x, int i);
void    mfo_mm_model_pose(const mfo_mm* x, int i, float* pose16);
const float*   mfo_mm_model_surfels(const mfo_mm* x, int i);
const uint8_t* mfo_mm_segmentation(const mfo_mm* x);   /* last fullSegmentation */
const uint8_t* mfo_mm_projected_ids(const mfo_mm* x);  /* last GlobalProjection ids */
const float*   mfo_mm_edge_map(const mfo_mm* x);
const float*   mfo_mm_dbg_map(const mfo_mm* x, int which /*0 vmap_g 1 nmap_g 2 vmap 3 nmap*/, int level);
const float*   mfo_mm_dbg_pred(const mfo_mm* x, int model, int which /*0 vertex 1 normal*/);   /* [H][W][4] */

#ifdef __cplusplus
}
#endif
#endif /* MF_ORACLE_H_ */

// mf_context.hip -- the mf_ctx object and the C ABI of include/maskfusion_amd.h.
//
// One context = one GPU, one HIP stream, one background model (multi-model sharding: one context per rank, see
// DESIGN.md section "multi-GPU").  MaskFusion::processFrame (Core/MaskFusion.cpp:200-607) becomes a fixed sequence of
// asynchronous launches on that stream; the host never reads anything back between them.
#include "../../include/maskfusion_amd.h"
#include "mf_internal.h"

#include <math.h>
#include <stdio.h>
#include <string.h>
#include <string>
#include <vector>

using namespace mf;

struct mf_ctx {
    mf_config cfg;
    int W, H, P;
    Intr K;
    hipStream_t stream = nullptr;
    std::string err;
    int host_tick = 1;
    bool timings_on = false;

    // frame-level
    uint8_t* d_rgb = nullptr; float* d_depth = nullptr; uint8_t* d_mask = nullptr; uint8_t* d_zero_mask = nullptr;
    float* d_depthF[2] = {nullptr, nullptr}; int curF = 0;
    float* d_dpyr[3] = {nullptr, nullptr, nullptr};
    float* d_vmap[3]; float* d_nmap[3];
    // model-level (background model)
    Surfels surf[2]; int cur = 0; int cap = 0;
    PoseDev* d_pose = nullptr; GNState* d_gn = nullptr; float* d_partials[2] = {nullptr, nullptr};
    float* d_vmap_g[3]; float* d_nmap_g[3];
    unsigned long long* d_keys = nullptr;
    int* d_index = nullptr; float4* d_ivc = nullptr; float4* d_ict = nullptr; float4* d_inr = nullptr;
    float4* d_predV = nullptr; float4* d_predN = nullptr; uchar4* d_predImage = nullptr; uint16_t* d_predTime = nullptr;
    uint8_t* d_cand_op = nullptr; float4* d_cand_rec = nullptr; int* d_upd_first = nullptr;
    uint8_t* d_flags = nullptr; float* d_newconf = nullptr; int* d_block_counts = nullptr;
    FrameDev* d_frame = nullptr; float* d_icp_log = nullptr; unsigned long long* d_icp_prof = nullptr; bool icp_prof_on = false;
    // pinned host mirrors
    PoseDev* h_pose = nullptr; FrameDev* h_frame = nullptr; int* h_count = nullptr;
    // timings
    hipEvent_t ev[MF_N_TIMINGS + 1] = {};
    float last_ms[MF_N_TIMINGS] = {};
    std::vector<void*> allocs;
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
static int dev_alloc(mf_ctx* c, T** p, size_t n, int fill = 0) {
    void* q = nullptr;
    MF_HIP(c, hipMalloc(&q, n * sizeof(T)));
    MF_HIP(c, hipMemsetAsync(q, fill, n * sizeof(T), c->stream));
    c->allocs.push_back(q);
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
    return MF_OK;
}

static __global__ void k_pose_identity(PoseDev* p) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    PoseDev q;
    memset(&q, 0, sizeof(q));
    for (int k = 0; k < 9; ++k) q.R[k] = q.Ri[k] = q.lastR[k] = (k % 4 == 0) ? 1.f : 0.f;
    q.fusionWeight = 1.f;
    *p = q;
}
static __global__ void k_frame_init(FrameDev* f) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    f->tick = 1; f->count = 0; f->countNext = 0; f->cover = 0; f->useFillIn = 0;
    f->pad[0] = f->pad[1] = f->pad[2] = 0;
}

extern "C" int mf_create(const mf_config* cfg, mf_ctx** out) {
    if (!cfg || !out) return MF_EINVAL;
    *out = nullptr;
    if (cfg->width <= 0 || cfg->height <= 0 || (cfg->width % 8) || (cfg->height % 8)) return MF_EINVAL;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || cfg->device >= ndev) return MF_ENODEV;
    mf_ctx* c = new mf_ctx();
    c->cfg = *cfg;
    c->W = cfg->width; c->H = cfg->height; c->P = c->W * c->H;
    c->K = Intr{cfg->fx, cfg->fy, cfg->cx, cfg->cy};
    auto fail = [&](int code) { mf_destroy(c); return code; };
    if (hipSetDevice(cfg->device) != hipSuccess) return fail(MF_ENODEV);
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) return fail(MF_ENODEV);
    const int W = c->W, H = c->H, P = c->P;
    // Model::TEXTURE_DIMENSION_GLOBAL^2 (Core/Model/Model.cpp:101-105)
    const int dim = 64 * (int)(sqrt((double)cfg->num_gsurfels) / 64);
    c->cap = dim * dim;
    if (c->cap <= 0) return fail(MF_EINVAL);
    int rc = MF_OK;
#define A(call) do { rc = (call); if (rc != MF_OK) return fail(rc); } while (0)
    A(dev_alloc(c, &c->d_rgb, (size_t)P * 3));
    A(dev_alloc(c, &c->d_depth, (size_t)P));
    A(dev_alloc(c, &c->d_mask, (size_t)P));
    A(dev_alloc(c, &c->d_zero_mask, (size_t)P));
    A(dev_alloc(c, &c->d_depthF[0], (size_t)P));
    A(dev_alloc(c, &c->d_depthF[1], (size_t)P));
    for (int i = 0; i < 3; ++i) {
        const size_t lp = (size_t)(W >> i) * (H >> i);
        if (i > 0) A(dev_alloc(c, &c->d_dpyr[i], lp));
        A(dev_alloc(c, &c->d_vmap[i], lp * 3));
        A(dev_alloc(c, &c->d_nmap[i], lp * 3));
        A(dev_alloc(c, &c->d_vmap_g[i], lp * 3));
        A(dev_alloc(c, &c->d_nmap_g[i], lp * 3));
    }
    for (int b = 0; b < 2; ++b) {
        A(dev_alloc(c, &c->surf[b].pc, (size_t)c->cap));
        A(dev_alloc(c, &c->surf[b].ct, (size_t)c->cap));
        A(dev_alloc(c, &c->surf[b].nr, (size_t)c->cap));
        c->surf[b].cap = c->cap;
        A(dev_alloc(c, &c->d_partials[b], (size_t)icp_grid_blocks(W, H) * kIcpSlots));
    }
    A(dev_alloc(c, &c->d_pose, 1));
    A(dev_alloc(c, &c->d_gn, 2));
    A(dev_alloc(c, &c->d_keys, (size_t)P, 0xFF));
    A(dev_alloc(c, &c->d_index, (size_t)P));
    A(dev_alloc(c, &c->d_ivc, (size_t)P));
    A(dev_alloc(c, &c->d_ict, (size_t)P));
    A(dev_alloc(c, &c->d_inr, (size_t)P));
    A(dev_alloc(c, &c->d_predV, (size_t)P));
    A(dev_alloc(c, &c->d_predN, (size_t)P));
    A(dev_alloc(c, &c->d_predImage, (size_t)P));
    A(dev_alloc(c, &c->d_predTime, (size_t)P));
    A(dev_alloc(c, &c->d_cand_op, (size_t)P));
    A(dev_alloc(c, &c->d_cand_rec, (size_t)P * 3));
    A(dev_alloc(c, &c->d_upd_first, (size_t)c->cap));
    A(dev_alloc(c, &c->d_flags, (size_t)c->cap + P));
    A(dev_alloc(c, &c->d_newconf, (size_t)c->cap + P));
    A(dev_alloc(c, &c->d_block_counts, (size_t)kCompactBlocks));
    A(dev_alloc(c, &c->d_frame, 1));
    A(dev_alloc(c, &c->d_icp_log, (size_t)20 * 32));
    A(dev_alloc(c, &c->d_icp_prof, (size_t)20 * 8));
#undef A
    launch_fill_int(c->d_upd_first, kNoUpdate, c->cap, c->stream);
    hipLaunchKernelGGL(k_pose_identity, dim3(1), dim3(64), 0, c->stream, c->d_pose);
    hipLaunchKernelGGL(k_frame_init, dim3(1), dim3(64), 0, c->stream, c->d_frame);
    if (hipHostMalloc((void**)&c->h_pose, sizeof(PoseDev)) != hipSuccess ||
        hipHostMalloc((void**)&c->h_frame, sizeof(FrameDev)) != hipSuccess ||
        hipHostMalloc((void**)&c->h_count, sizeof(int)) != hipSuccess)
        return fail(MF_ENOMEM);
    memset(c->h_pose, 0, sizeof(PoseDev));
    for (int k = 0; k < 9; ++k) c->h_pose->R[k] = c->h_pose->Ri[k] = (k % 4 == 0) ? 1.f : 0.f;
    memset(c->h_frame, 0, sizeof(FrameDev));
    c->h_frame->tick = 1;
    *c->h_count = 0;
    for (int i = 0; i <= MF_N_TIMINGS; ++i)
        if (hipEventCreate(&c->ev[i]) != hipSuccess) return fail(MF_EHIP);
    if (hipStreamSynchronize(c->stream) != hipSuccess) return fail(MF_EHIP);
    *out = c;
    return MF_OK;
}

extern "C" void mf_destroy(mf_ctx* c) {
    if (!c) return;
    if (c->stream) hipStreamSynchronize(c->stream);
    for (void* p : c->allocs) hipFree(p);
    if (c->h_pose) hipHostFree(c->h_pose);
    if (c->h_frame) hipHostFree(c->h_frame);
    if (c->h_count) hipHostFree(c->h_count);
    for (int i = 0; i <= MF_N_TIMINGS; ++i)
        if (c->ev[i]) hipEventDestroy(c->ev[i]);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
}

extern "C" const char* mf_last_error(const mf_ctx* c) { return c ? c->err.c_str() : "null context"; }

// MaskFusion::predict for the background model: combinedPredict(maxDepthProcessed, tick, tick, timeDelta) -- the fill-in
// half (performFillIn) is evaluated lazily by the next tracking step from the retained filtered depth.
static void enqueue_predict(mf_ctx* c) {
    launch_splat_scatter(c->surf[c->cur], c->d_frame, c->d_pose, c->W, c->H, c->K, c->cfg.max_depth_processed,
                         c->cfg.conf_global, c->cfg.time_delta, c->d_keys, c->stream);
    launch_splat_resolve(c->surf[c->cur], c->d_pose, c->d_keys, c->W, c->H, c->K, c->d_predV, c->d_predN, c->d_predImage,
                         c->d_predTime, c->d_frame, c->stream);
}

static void mark(mf_ctx* c, int i) {
    if (c->timings_on) hipEventRecord(c->ev[i], c->stream);
}

extern "C" int mf_process_frame_dev(mf_ctx* c, const uint8_t* d_rgb, const float* d_depth, const uint8_t* d_mask,
                                    int64_t timestamp, float weight_multiplier) {
    (void)timestamp;
    if (!c || !d_rgb || !d_depth) return MF_EINVAL;
    const int W = c->W, H = c->H;
    hipStream_t s = c->stream;
    const mf_config& g = c->cfg;
    // -static (enableMultipleModels == false): everything is background (MaskFusion.cpp:223-230)
    const uint8_t* mask = (g.enable_multiple_models && d_mask) ? d_mask : c->d_zero_mask;
    float* depthF = c->d_depthF[c->curF];
    float* depthF_prev = c->d_depthF[1 - c->curF];

    mark(c, 0);
    launch_bilateral(d_depth, depthF, W, H, s);  // filterDepth, :217
    if (c->host_tick == 1) {
        mark(c, 1); mark(c, 2); mark(c, 3); mark(c, 4); mark(c, 5); mark(c, 6);
        // :235-238
        launch_init_surfels(d_rgb, d_depth, depthF, W, H, c->K, g.max_depth_processed, c->d_frame, c->d_cand_rec, c->d_flags, s);
        c->cur = 0;
        launch_compact_records(c->d_cand_rec, c->d_flags, c->P, c->surf[0], c->d_frame, c->d_block_counts, c->h_count, s);
        mark(c, 7);
    } else {
        // Model::generateCUDATextures (Model.cpp:350-389)
        c->d_dpyr[0] = depthF;
        for (int i = 1; i < 3; ++i) launch_pyrdown_f(c->d_dpyr[i - 1], c->d_dpyr[i], W >> (i - 1), H >> (i - 1), s);
        for (int i = 0; i < 3; ++i) {
            const float div = (float)(1 << i);
            const Intr ki{g.fx / div, g.fy / div, g.cx / div, g.cy / div};
            launch_vmap_nmap(c->d_dpyr[i], c->d_vmap[i], c->d_nmap[i], W >> i, H >> i, ki, g.depth_cutoff, s);
        }
        mark(c, 1);
        // Model::performTracking (Model.cpp:427-447): initICPModel (+ fill-in) then the Gauss-Newton loop
        launch_model_pyramid(c->d_predV, c->d_predN, depthF_prev, c->d_frame, c->d_pose, nullptr, c->d_vmap_g, c->d_nmap_g, W, H,
                             c->K, s);
        mark(c, 2);
        launch_icp_begin(c->d_pose, &c->d_gn[0], s);
        int iters[3] = {g.fast_odom ? 3 : 10, g.pyramid ? 5 : 0, g.pyramid ? 4 : 0};  // RGBDOdometry.cpp:327-329
        int k = 0, nb_prev = 0;
        for (int lvl = 2; lvl >= 0; --lvl) {
            const float div = (float)(1 << lvl);
            for (int j = 0; j < iters[lvl]; ++j) {
                IcpLaunch l;
                l.vmap_curr = c->d_vmap[lvl]; l.nmap_curr = c->d_nmap[lvl];
                l.vmap_prev = c->d_vmap_g[lvl]; l.nmap_prev = c->d_nmap_g[lvl];
                l.W = W >> lvl; l.H = H >> lvl; l.k = Intr{g.fx / div, g.fy / div, g.cx / div, g.cy / div};
                l.distThres = 0.10f; l.angleThres = sinf(20.f * 3.14159254f / 180.f);  // RGBDOdometry.h:35-36
                l.partials_in = nb_prev ? c->d_partials[(k + 1) & 1] : nullptr;
                l.nblocks_in = nb_prev;
                l.partials_out = c->d_partials[k & 1];
                l.state_in = &c->d_gn[k & 1]; l.state_out = &c->d_gn[(k + 1) & 1];
                l.log_out = k > 0 ? c->d_icp_log + 32 * (k - 1) : nullptr;
                l.prof_out = c->icp_prof_on ? c->d_icp_prof + 8 * k : nullptr;
                launch_icp_iteration(l, s);
                nb_prev = icp_grid_blocks(l.W, l.H);
                ++k;
            }
        }
        launch_icp_finalize(nb_prev ? c->d_partials[(k + 1) & 1] : nullptr, nb_prev, &c->d_gn[k & 1], c->d_pose, c->h_pose,
                            k > 0 ? c->d_icp_log + 32 * (k - 1) : nullptr, s);
        mark(c, 3);
        // (the predict() at MaskFusion.cpp:423 only feeds the dead loop-closure block and is overwritten at :569)
        // fusion, :539-565
        const int src = c->cur, dst = 1 - c->cur;
        launch_index_scatter(c->surf[src], c->d_frame, c->d_pose, W, H, c->K, g.max_depth_processed, g.time_delta, c->d_keys, s);
        launch_index_resolve(c->surf[src], c->d_pose, c->d_keys, W, H, c->d_index, c->d_ivc, c->d_ict, c->d_inr, s);
        mark(c, 4);
        launch_fuse_data(d_rgb, d_depth, depthF, mask, 0, c->d_frame, c->d_pose, weight_multiplier, g.depth_cutoff, W, H, c->K,
                         c->d_index, c->d_ivc, c->d_inr, c->d_cand_op, c->d_cand_rec, c->d_upd_first, s);
        mark(c, 5);
        launch_fuse_update(c->surf[src], c->surf[dst], c->d_frame, c->d_upd_first, c->d_cand_rec, s);
        mark(c, 6);
        launch_index_scatter(c->surf[dst], c->d_frame, c->d_pose, W, H, c->K, g.max_depth_processed, g.time_delta, c->d_keys, s);
        launch_index_resolve(c->surf[dst], c->d_pose, c->d_keys, W, H, c->d_index, c->d_ivc, c->d_ict, c->d_inr, s);
        launch_clean(c->surf[dst], c->surf[src], c->d_frame, c->d_pose, W, H, c->K, g.time_delta, g.conf_global,
                     g.outlier_coefficient, 0, c->d_index, c->d_ivc, c->d_ict, depthF, mask, c->d_cand_op, c->d_cand_rec,
                     c->d_flags, c->d_newconf, c->d_block_counts, c->h_count, s);
        mark(c, 7);
    }
    enqueue_predict(c);  // :569
    launch_frame_advance(c->d_frame, W, H, c->h_frame, s);
    mark(c, 8);
    c->curF = 1 - c->curF;
    c->host_tick++;
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { c->err = std::string("launch failed: ") + hipGetErrorString(e); return MF_EHIP; }
    return MF_OK;
}

extern "C" int mf_sync(mf_ctx* c) {
    if (!c) return MF_EINVAL;
    MF_HIP(c, hipStreamSynchronize(c->stream));
    if (c->timings_on) {
        // event i marks the START of stage i; stage i lasts until event i+1.  Stage map (see header):
        // ev0 Preprocess(bilateral+pyramid+maps) ev1 odomInit ev2 odom ev3 indexMap ev4 Fuse::Data ev5 Fuse::Update
        // ev6 indexMap#2 + Fuse::Copy ev7 IndexMap::ACTIVE ev8 end
        float t[MF_N_TIMINGS] = {};
        for (int i = 0; i < 8; ++i) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, c->ev[i], c->ev[i + 1]) == hipSuccess) t[i] = ms;
        }
        float run = 0.f;
        if (hipEventElapsedTime(&run, c->ev[0], c->ev[8]) == hipSuccess) t[8] = run;
        memcpy(c->last_ms, t, sizeof(t));
    }
    return MF_OK;
}

extern "C" int mf_process_frame(mf_ctx* c, const uint8_t* rgb, const float* depth, const uint8_t* mask, const int32_t* class_ids,
                                int32_t n_masks, int64_t timestamp, const float* in_pose16, float weight_multiplier,
                                int32_t bootstrap) {
    (void)class_ids; (void)n_masks;
    if (!c || !rgb || !depth) return MF_EINVAL;
    if (in_pose16 || bootstrap) { c->err = "in_pose / bootstrap not supported yet"; return MF_ESTATE; }
    MF_HIP(c, hipMemcpyAsync(c->d_rgb, rgb, (size_t)c->P * 3, hipMemcpyHostToDevice, c->stream));
    MF_HIP(c, hipMemcpyAsync(c->d_depth, depth, (size_t)c->P * sizeof(float), hipMemcpyHostToDevice, c->stream));
    if (mask) MF_HIP(c, hipMemcpyAsync(c->d_mask, mask, (size_t)c->P, hipMemcpyHostToDevice, c->stream));
    int rc = mf_process_frame_dev(c, c->d_rgb, c->d_depth, mask ? c->d_mask : nullptr, timestamp, weight_multiplier);
    if (rc != MF_OK) return rc;
    return mf_sync(c);
}

extern "C" int mf_predict(mf_ctx* c) {
    if (!c) return MF_EINVAL;
    enqueue_predict(c);
    return MF_OK;
}

extern "C" int mf_get_tick(mf_ctx* c, int32_t* tick) {
    if (!c || !tick) return MF_EINVAL;
    *tick = c->host_tick;
    return MF_OK;
}
extern "C" int mf_num_models(mf_ctx* c, int32_t* n) {
    if (!c || !n) return MF_EINVAL;
    *n = 1;
    return MF_OK;
}
extern "C" int mf_get_pose(mf_ctx* c, int32_t model, float* out) {
    if (!c || !out || model != 0) return MF_EINVAL;
    int rc = mf_sync(c);
    if (rc != MF_OK) return rc;
    const PoseDev& p = *c->h_pose;
    for (int r = 0; r < 3; ++r) {
        for (int col = 0; col < 3; ++col) out[col * 4 + r] = p.R[r * 3 + col];
        out[12 + r] = p.t[r];
        out[r * 4 + 3] = 0.f;
    }
    out[15] = 1.f;
    return MF_OK;
}
extern "C" int mf_get_surfel_count(mf_ctx* c, int32_t model, uint32_t* count) {
    if (!c || !count || model != 0) return MF_EINVAL;
    int rc = mf_sync(c);
    if (rc != MF_OK) return rc;
    *count = (uint32_t)*c->h_count;
    return MF_OK;
}
extern "C" int mf_get_icp_stats(mf_ctx* c, int32_t model, float* e, float* n) {
    if (!c || !e || !n || model != 0) return MF_EINVAL;
    int rc = mf_sync(c);
    if (rc != MF_OK) return rc;
    *e = c->h_pose->lastICPError; *n = c->h_pose->lastICPCount;
    return MF_OK;
}
extern "C" int mf_get_last_fillin(mf_ctx* c, int32_t* used) {
    if (!c || !used) return MF_EINVAL;
    int rc = mf_sync(c);
    if (rc != MF_OK) return rc;
    // h_frame mirrors the state AFTER frame_advance: useFillIn there is the decision for the NEXT frame, pad[0] the
    // decision the last tracking step ran with.
    *used = c->h_frame->pad[0];
    return MF_OK;
}

extern "C" int mf_download_map(mf_ctx* c, int32_t model, float* out, uint32_t max_count, uint32_t* count) {
    if (!c || !out || !count || model != 0) return MF_EINVAL;
    int rc = mf_sync(c);
    if (rc != MF_OK) return rc;
    const uint32_t n = (uint32_t)*c->h_count;
    *count = n;
    const uint32_t m = n < max_count ? n : max_count;
    if (m == 0) return MF_OK;
    std::vector<float4> a(m), b(m), d(m);
    const Surfels& s = c->surf[c->cur];
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

struct ParamRef { const char* key; int kind; size_t off; };  // kind 0 float, 1 int
static const ParamRef kParams[] = {
    {"depthCutoff", 0, offsetof(mf_config, depth_cutoff)},
    {"icpWeight", 0, offsetof(mf_config, icp_weight)},
    {"confidenceThreshold", 0, offsetof(mf_config, conf_global)},
    {"outlierCoefficient", 0, offsetof(mf_config, outlier_coefficient)},
    {"maxDepthProcessed", 0, offsetof(mf_config, max_depth_processed)},
    {"fastOdom", 1, offsetof(mf_config, fast_odom)},
    {"so3", 1, offsetof(mf_config, so3)},
    {"pyramid", 1, offsetof(mf_config, pyramid)},
    {"timeDelta", 1, offsetof(mf_config, time_delta)},
    {"enableMultipleModels", 1, offsetof(mf_config, enable_multiple_models)},
};
extern "C" int mf_set_param(mf_ctx* c, const char* key, double value) {
    if (!c || !key) return MF_EINVAL;
    if (!strcmp(key, "timings")) { c->timings_on = value != 0; return MF_OK; }
    if (!strcmp(key, "icpProfile")) { c->icp_prof_on = value != 0; return MF_OK; }
    for (const ParamRef& p : kParams)
        if (!strcmp(key, p.key)) {
            char* base = reinterpret_cast<char*>(&c->cfg);
            if (p.kind == 0) *reinterpret_cast<float*>(base + p.off) = (float)value;
            else *reinterpret_cast<int32_t*>(base + p.off) = (int32_t)value;
            return MF_OK;
        }
    c->err = std::string("unknown parameter: ") + key;
    return MF_EINVAL;
}
extern "C" int mf_get_param(mf_ctx* c, const char* key, double* value) {
    if (!c || !key || !value) return MF_EINVAL;
    for (const ParamRef& p : kParams)
        if (!strcmp(key, p.key)) {
            const char* base = reinterpret_cast<const char*>(&c->cfg);
            *value = p.kind == 0 ? (double)*reinterpret_cast<const float*>(base + p.off)
                                 : (double)*reinterpret_cast<const int32_t*>(base + p.off);
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

extern "C" int mf_debug_read(mf_ctx* c, const char* what, void* out, uint64_t out_bytes) {
    if (!c || !what || !out) return MF_EINVAL;
    int rc = mf_sync(c);
    if (rc != MF_OK) return rc;
    const void* src = nullptr;
    size_t bytes = 0;
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
    if (w == "depthF") { src = c->d_depthF[1 - c->curF]; bytes = P * 4; }  // curF was flipped at the end of the frame
    else if (lvl("vmap_g", c->d_vmap_g) || lvl("nmap_g", c->d_nmap_g) || lvl("vmap", c->d_vmap) || lvl("nmap", c->d_nmap)) {}
    else if (w == "pred_vertex") { src = c->d_predV; bytes = P * 16; }
    else if (w == "pred_normal") { src = c->d_predN; bytes = P * 16; }
    else if (w == "pred_image") { src = c->d_predImage; bytes = P * 4; }
    else if (w == "index") { src = c->d_index; bytes = P * 4; }
    else if (w == "index_vc") { src = c->d_ivc; bytes = P * 16; }
    else if (w == "icp_log") { src = c->d_icp_log; bytes = 19 * 32 * 4; }
    else if (w == "icp_prof") { src = c->d_icp_prof; bytes = 19 * 8 * 8; }
    else { c->err = "unknown debug tap: " + w; return MF_EINVAL; }
    if (out_bytes < bytes) { c->err = "debug_read: buffer too small"; return MF_EINVAL; }
    MF_HIP(c, hipMemcpy(out, src, bytes, hipMemcpyDeviceToHost));
    return MF_OK;
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
    launch_model_pyramid((const float4*)d_v4, (const float4*)d_n4, nullptr, nullptr, nullptr, Rt, vm, nm, W, H,
                         Intr{1, 1, 0, 0}, (hipStream_t)stream);
    return launch_rc();
}
extern "C" int mf_k_icp_step(const float* Rcurr9, const float* tcurr3, const float* d_vc, const float* d_nc, const float* Rpi9,
                             const float* tprev3, float fx, float fy, float cx, float cy, const float* d_vp, const float* d_np,
                             float dist_thresh, float angle_thresh, int32_t W, int32_t H, float* d_out32, void* stream) {
    if (!Rcurr9 || !tcurr3 || !d_vc || !d_nc || !Rpi9 || !tprev3 || !d_vp || !d_np || !d_out32 || (W * H) % 4) return MF_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    float* scratch = nullptr;
    const size_t nb = (size_t)icp_grid_blocks(W, H);
    const size_t bytes = nb * kIcpSlots * sizeof(float) + 2 * sizeof(GNState) + 24 * sizeof(float) + 64;
    if (hipMalloc((void**)&scratch, bytes) != hipSuccess) return MF_ENOMEM;
    char* base = reinterpret_cast<char*>(scratch);
    size_t o = nb * kIcpSlots * sizeof(float);
    o = (o + 15) & ~(size_t)15;
    GNState* st = reinterpret_cast<GNState*>(base + o);
    float* dpose = reinterpret_cast<float*>(base + o + 2 * sizeof(GNState));
    float hp[24];
    memcpy(hp, Rcurr9, 36); memcpy(hp + 9, tcurr3, 12); memcpy(hp + 12, Rpi9, 36); memcpy(hp + 21, tprev3, 12);
    hipMemcpyAsync(dpose, hp, sizeof(hp), hipMemcpyHostToDevice, s);
    hipStreamSynchronize(s);  // hp is a stack buffer
    launch_icp_step_standalone(dpose, dpose + 9, d_vc, d_nc, dpose + 12, dpose + 21, Intr{fx, fy, cx, cy}, d_vp, d_np, dist_thresh,
                               angle_thresh, W, H, scratch, st, d_out32, s);
    hipStreamSynchronize(s);
    hipFree(scratch);
    return launch_rc();
}

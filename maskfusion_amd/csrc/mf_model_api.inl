// mf_model_api.inl -- the Model-level entry points (Core/Model/Model.h:126-162,233-268, one call each) and the calls that let one scene be
// sharded by model over several contexts.
// (part of mf_context.hip: the library's host side is ONE translation unit -- the context type and its helpers are file-local -- kept in
// four files by subject; mf_context.hip includes them in this order)

static ModelState* model_at(mf_ctx* c, int32_t i);

// ------------------------------------------------------------------------------------------------
// Model-level entry points: the public operations of Model (Core/Model/Model.h:126-162,233-268) one by one, on a frame
// staged with mf_stage_frame.  MaskFusion::processFrame is a fixed composition of these (mf_process_frame enqueues the
// same launches); they exist so that a caller can drive a model the way the reference's own callers do, and so that each
// surfel pass can be compared with the oracle in isolation.
// ------------------------------------------------------------------------------------------------
static __global__ void k_set_count(FrameDev* f, int count, int* host_count) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    f->count = count; f->countNext = count; f->phys = count; f->runs = 0; f->first = 0; f->first_run = 0;   // a dense buffer
    if (host_count) *host_count = count;
}
static void set_model_tick(mf_ctx* c, ModelState& m, int tick) {
    hipLaunchKernelGGL(k_set_tick, dim3(1), dim3(64), 0, c->stream, m.d_frame, tick, m.h_frame);
    m.gen++;   // (what is "seen within timeDelta" changes with the tick: cached visibility lists are stale)
}
// bookkeeping behind a call that replaced m's buffer by a dense one of (at most) n surfels and built its run table
static void fresh_table(mf_ctx* c, ModelState& m, long n) {
    m.table_valid = true; m.sparse = false; m.phys_ub = n; m.runs_ub = (n + kRun - 1) / kRun; m.gen++; m.mirror_from = m.clean_seq + 1;
    c->vis_tag.model = nullptr;
}
static long staged_frame(const mf_ctx* c) { return c->frame_no - 1; }   // index of the frame staged / processed last
static const uint8_t* current_mask(const mf_ctx* c) { return c->cfg.enable_multiple_models ? c->d_mask_tex : c->d_zero_mask; }

// upload + MaskFusion::filterDepth (:217) + Model::generateCUDATextures (Model.cpp:350-389) + intensity pyramid: everything of
// processFrame that does not touch a model.  mask: model id per pixel = what textureMask holds for fuse / clean (NULL: left as it is)
extern "C" int mf_stage_frame(mf_ctx* c, const uint8_t* rgb, const float* depth, const uint8_t* mask) {
    if (c) c->vis_tag.model = nullptr;   // (a visibility list belongs to one frame and one pose)
    if (!c || !rgb || !depth) return MF_EINVAL;
    hipStream_t sin = c->stream;
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
    fresh_table(c, *m, (long)c->P);
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
    fresh_table(c, *m, (long)count);
    if (model == 0) c->map_ready = true;   // the map exists: the next mf_process_frame tracks instead of initialising
    return check_launch(c);
}

// Model::overridePose (Core/Model/Model.h:235-238): lastPose = pose; pose = p
extern "C" int mf_model_override_pose(mf_ctx* c, int32_t model, const float* pose16) {
    ModelState* m = model_at(c, model);
    if (!m || !pose16) return MF_EINVAL;
    launch_override_pose(m->d_pose, pose16, 0, m->h_pose, c->stream);
    m->gen++;
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
    launch_index_scatter(m->surf[m->cur], m->d_frame, m->d_pose, c->W, c->H, c->K, max_depth, time_delta, c->d_keys, true, s);
    launch_index_resolve(m->surf[m->cur], m->d_frame, m->d_pose, c->d_keys, c->W, c->H, c->d_index, c->d_ivc, c->d_inr, c->d_ict, nullptr, nullptr, nullptr, nullptr, true, s);
    if (c->model_api_packed) {   // the layout mf_process_frame feeds clean() with: packed records, column-major
        launch_index_scatter(m->surf[m->cur], m->d_frame, m->d_pose, c->W, c->H, c->K, max_depth, time_delta, c->d_keys, true, s);
        launch_index_resolve(m->surf[m->cur], m->d_frame, m->d_pose, c->d_keys, c->W, c->H, nullptr, nullptr, nullptr, nullptr, c->d_iclean, c->d_depthF[staged_frame(c) % 3],
                             current_mask(c), c->d_maskT, true, s);
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
    // merged surfels moved and were seen now: the boxes and time stamps of their runs are refreshed (inside mf_process_frame the clean pass that
    // follows rewrites the entries of every run it visits -- and it visits every run a merge can have touched)
    if (m->table_valid) launch_run_table(m->surf[m->cur], m->d_frame, s, true);
    m->gen++;
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
    const bool packed = c->model_api_packed != 0;
    bool small = clean_small(c, *m);
    if (!small) {
        bool ok = true;
        int rc = prepare_in_place(c, *m, ok);
        if (rc != MF_OK) return rc;
        small = !ok;
    } else {
        require_dense(c, *m);
    }
    CleanIn in = clean_in(c, *m, time_delta, packed, c->d_depthF[k % 3], current_mask(c));
    if (small) {
        launch_clean_small(in, m->surf[m->cur], m->surf[1 - m->cur], c->stream);
        m->cur = 1 - m->cur;
    } else {
        enqueue_clean_in_place(c, *m, in, false);   // (no decay statistics outside a frame: every run is visited)
    }
    after_clean(c, *m, !small);
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
    if (spawned_model > 0) {
        int rc = enqueue_fuse_clean(c, *c->models[spawned_model], c->cur_rgb, c->cur_depth, depthF, mask, g.max_depth_processed, 100.f, false, false);
        if (rc != MF_OK) return rc;
    }
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

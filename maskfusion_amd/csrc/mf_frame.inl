// mf_frame.inl -- MaskFusion::processFrame (Core/MaskFusion.cpp:200-607) as a sequence of launches: the per-model stages (tracking, predictIndices ->
// fuse -> clean, prediction), their batched forms, model spawn / retirement, and the frame-level entry points mf_process_frame[_dev], mf_sync, mf_predict.
// (part of mf_context.hip: the library's host side is ONE translation unit -- the context type and its helpers are file-local -- kept in
// four files by subject; mf_context.hip includes them in this order)

static void mark(mf_ctx* c, int i, hipStream_t s = nullptr) {
    if (c->timings_on) (void)hipEventRecord(c->ev[i], s ? s : c->stream);
}
// "passTimings": an event pair around the launches of one surfel pass (labels: maskfusion_amd.h, MF_PASS_*); a pass that runs several times in a
// frame (object models handled one by one) keeps its last run
struct PassTimer {
    mf_ctx* c; int id; hipStream_t s;
    PassTimer(mf_ctx* c_, int id_, hipStream_t s_ = nullptr) : c(c_), id(id_), s(s_ ? s_ : c_->stream) {
        if (!c->pass_timings_on || id < 0) return;
        for (int q = 0; q < 2; ++q)
            if (!c->ev_pass[id][q] && hipEventCreate(&c->ev_pass[id][q]) != hipSuccess) { (void)hipGetLastError(); c->ev_pass[id][q] = nullptr; }
        if (c->ev_pass[id][0]) (void)hipEventRecord(c->ev_pass[id][0], s);
    }
    ~PassTimer() {
        if (!c->pass_timings_on || id < 0 || !c->ev_pass[id][0] || !c->ev_pass[id][1]) return;
        (void)hipEventRecord(c->ev_pass[id][1], s);
        c->pass_recorded[id] = true;
    }
};

// "objectStream": the window of a frame in which the batched object passes go to their own stream (mf_context.hip).  begin: behind the label
// stage, which the host has waited for -- nothing the object chain reads is still being written on the main stream unless obj_dep_main says so.
// The destructor joins: the main stream waits for the object stream, so that everything behind the frame (the next frame, every getter) sees
// one stream again.
struct ObjStreamWindow {
    mf_ctx* c; bool open = false;
    explicit ObjStreamWindow(mf_ctx* c_) : c(c_) {}
    void begin() {
        if (open || !c->object_stream || !c->stream_obj || c->timings_on) return;   // (the stage timings bracket the main stream: one stream while they are on)
        open = true;
        c->obj_s = c->stream_obj;
        c->obj_dep_main = false;
    }
    ~ObjStreamWindow() {
        if (!open) return;
        (void)hipEventRecord(c->ev_obj_done, c->stream_obj);
        (void)hipStreamWaitEvent(c->stream, c->ev_obj_done, 0);
        c->obj_s = c->stream;
        c->obj_dep_main = false;
    }
};
// before the object chain's launches: if main-stream work it depends on was enqueued inside the window (a spawned model's first fusion, a
// compaction, a fresh run table), the object stream waits for the main stream as it stands now
static void obj_stream_catch_up(mf_ctx* c) {
    if (c->obj_s == c->stream || !c->obj_dep_main) return;
    (void)hipEventRecord(c->ev_obj_dep, c->stream);
    (void)hipStreamWaitEvent(c->obj_s, c->ev_obj_dep, 0);
    c->obj_dep_main = false;
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
    m.gen++;   // the pose changes: cached visibility lists are stale
    const int set = (int)(frame_k & 1);
    float* const* cur_vmap = c->d_vmap[set];
    float* const* cur_nmap = c->d_nmap[set];
    const int W = c->W, H = c->H;
    hipStream_t s = c->stream;
    if (c->pyr_done == &m) c->pyr_done = nullptr;   // built beside the frame's depth filter (enqueue_preprocess, "fusedPreprocessLaunch")
    else launch_model_pyramid(m.d_predV, m.d_predN, m.allowFillIn ? fillDepth : nullptr, m.d_frame, m.d_pose, nullptr, m.d_vmap_g,
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
    if (c->pyr_batch_done) c->pyr_batch_done = false;   // built beside the frame's depth filter (enqueue_preprocess, "fusedPreprocessLaunch")
    else launch_model_pyramid_batch(b, fillDepth, W, H, c->K, s);
    const bool so3 = g.so3 != 0 && c->gray_frame[set ^ 1] == frame_k - 1 && c->gray_frame[set] == frame_k;
    if (so3)   // one pre-alignment serves every model: it only looks at the two frames (RGBDOdometry.cpp:264-324)
        (void)launch_so3_prealign(c->d_gray[set ^ 1][2], c->d_gray[set][2], W >> 2, H >> 2, Intr{g.fx / 4, g.fy / 4, g.cx / 4, g.cy / 4},
                                  c->d_so3, c->d_so3_scratch, s);
    const So3Result* so3_seed = so3 ? c->d_so3 : nullptr;
    const int iters[3] = {g.fast_odom ? 3 : 10, g.pyramid ? 5 : 0, g.pyramid ? 4 : 0};
    const bool timed = c->timings_on && ms[0] == c->models[0].get();
    if (timed) { (void)hipEventRecord(c->ev_icp[0], s); c->tracked_once = true; c->icp_mid_recorded = false; }
    // slab culling of the pixel pass (mf_odometry.hip: k_icp_batch_pixels): the depth range of every row of the frame's vertex maps, once per frame
    if (c->slab_culling) launch_row_zrange(c->d_vmap[set], W, H, c->d_row_z, s);
    const int row_base[3] = {0, H, H + (H >> 1)};
    int it = 0, nb_prev = 0;
    for (int lvl = 2; lvl >= 0; --lvl) {
        const float div = (float)(1 << lvl);
        for (int j = 0; j < iters[lvl]; ++j) {
            // one launch per iteration ("batchSolveInPixelPass", default): the pixel pass finishes the previous iteration in its prologue; else two
            if (!c->batch_solve_fused) launch_icp_batch_solve(b, it, nb_prev, it == 0 ? so3_seed : nullptr, s);
            launch_icp_batch_pixels(b, it, lvl, c->d_vmap[set][lvl], c->d_nmap[set][lvl], W >> lvl, H >> lvl,
                                    Intr{g.fx / div, g.fy / div, g.cx / div, g.cy / div}, 0.10f, sinf(20.f * 3.14159254f / 180.f), s,
                                    c->slab_culling ? c->d_row_z + row_base[lvl] : nullptr, c->batch_solve_fused, nb_prev, it == 0 ? so3_seed : nullptr);
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

// The runs of m's live buffer that can be in view under m's current pose (k_cull), for the projection passes of this frame: culled once per
// buffer state and pose -- GlobalProjection and the two index-map passes of a frame share one list (the in-place update moves no surfel of a
// run that is not listed: only surfels the first index map drew are merged), the prediction after clean() gets its own (the buffer changed).
// Depth range: the widest any consumer uses.  nullptr: no list -- the passes walk every run of the table (or, a dense buffer without one, every slot).
static const VisList* ensure_vis(mf_ctx* c, ModelState& m, VisList& out) {
    // (a small map is cheaper to stream than to cull: the test is a launch of its own on a chain of launches that are each a few microseconds.
    // The count is the pinned mirror as of the model's last clean pass; which side of the threshold a frame falls on changes no result)
    if (!c->cull_runs || !m.table_valid || (!m.sparse && *m.h_count < c->big_map_elements)) return nullptr;
    const mf_config& g = c->cfg;
    const float max_depth = fmaxf(g.depth_cutoff, g.max_depth_processed);
    out.list = c->d_vis_list; out.count = c->d_vis_count;
    // the list depends on the buffer and its table, the pose, the tick (all behind m.gen), the depth bound and timeDelta
    if (c->vis_tag.model == &m && c->vis_tag.frame == c->frame_no && c->vis_tag.cur == m.cur && c->vis_tag.gen == m.gen && c->vis_tag.max_depth == max_depth &&
        c->vis_tag.time_delta == g.time_delta)
        return &out;
    launch_cull(m.surf[m.cur], m.d_frame, m.d_pose, c->W, c->H, c->K, max_depth, g.time_delta, c->d_vis_list, c->d_vis_count, c->d_cull_ctl,
                (int)run_table_runs((long)m.cap, (long)c->P), c->stream);
    c->vis_tag.model = &m; c->vis_tag.frame = c->frame_no; c->vis_tag.cur = m.cur; c->vis_tag.gen = m.gen; c->vis_tag.max_depth = max_depth;
    c->vis_tag.time_delta = g.time_delta;
    return &out;
}

// which clean form model m gets this frame: the two-launch form (a dense copy) below "bigMapElements", in place on the buffer's runs from there on
static bool clean_small(const mf_ctx* c, const ModelState& m) { return (long)*m.h_count + (long)c->P / 4 < (long)c->big_map_elements; }
// which update form: the copying one (second index scatter riding on it) below "inPlaceElements"; it pairs with the two-launch clean only
static bool update_copy(const mf_ctx* c, const ModelState& m) {
    return clean_small(c, m) && (long)*m.h_count + (long)c->P / 4 < (long)c->in_place_elements;
}
// what a frame can append to a buffer: candidates (data.vert:117 takes every other pixel of every other row) and table entries for them
static long cand_max(const mf_ctx* c) { return (long)((c->W + 1) / 2) * (long)((c->H + 1) / 2); }
static long new_runs_max(const mf_ctx* c) { return (cand_max(c) + kRun - 1) / kRun + 1; }

// (the mirror's bit fields hold buffers of up to 2^26 slots in up to 2^18 runs: 64 M surfels; beyond, the host's own bounds alone decide)
static bool append_mirror_fits(const mf_ctx* c, const ModelState& m) {
    return (long)m.cap < (1L << kAppendPhysBits) && (long)run_table_runs((long)m.cap, (long)c->P) < (1L << kAppendRunBits);
}
// compaction of m's sparse buffer: the surfels of its runs -> the other buffer, dense, no table (launch_densify); that one is live afterwards
static void densify(mf_ctx* c, ModelState& m) {
    launch_densify(m.surf[m.cur], m.surf[1 - m.cur], m.d_frame, c->d_run_offs, m.h_count, c->stream);
    c->obj_dep_main = true;
    m.cur = 1 - m.cur;
    m.sparse = false; m.table_valid = false; m.phys_ub = m.runs_ub = -1; m.gen++;
    m.mirror_from = m.clean_seq + 1;      // (mirrors of earlier passes describe the buffer that was)
    c->densify_count++;
}
// the passes that read a buffer slot by slot (copy-update, two-launch clean, download) need a dense one
static void require_dense(mf_ctx* c, ModelState& m) { if (m.sparse) densify(c, m); }

// Makes m ready for the in-place clean of this frame: a run table, and -- by the host's bounds on what the device has appended since the last exact
// count -- room behind the last run and in the table for a frame's candidates.  When the bounds run out the buffer is compacted and the host reads
// the exact count: ONE stream synchronisation per compaction (every (capacity - count) / (P / 4) frames or later).  ok = false: the (dense) buffer
// is within a frame's candidates of its capacity -- the frame takes the two-launch form, whose ordered copy stops at the capacity exactly where the
// reference's transform feedback does (tests/test_gpu_pipeline.py::test_full_map_clamps_like_the_oracle); such a map pays the synchronisation every frame.
static int prepare_in_place(mf_ctx* c, ModelState& m, bool& ok) {
    const long cm = cand_max(c), nr = new_runs_max(c), table = (long)run_table_runs((long)m.cap, (long)c->P);
    ok = true;
    if (c->densify_every > 0 && m.sparse && (c->frame_no % c->densify_every) == 0) densify(c, m);   // ("densifyEvery": tests)
    // The host's own bounds grow by a frame's worth of CANDIDATES per pass, the buffer by the few that survive.  The device leaves where the buffer
    // really ended behind every pass in pinned memory (append_mirror): bounds from the newest pass that has run are P / 4 per pass still in flight.
    long pub = m.phys_ub, rub = m.runs_ub;
    if (pub >= 0) {
        const unsigned long long v = *(volatile unsigned long long*)m.h_append;
        const unsigned mask = (1u << kAppendSeqBits) - 1u, seq = (unsigned)(v >> (kAppendRunBits + kAppendPhysBits)) & mask;
        const unsigned behind = (m.clean_seq - seq) & mask, since = (seq - m.mirror_from) & mask;   // passes enqueued behind it / it is not older than the buffer
        if (v != 0ull && behind < (mask >> 1) && since < (mask >> 1)) {
            pub = std::min(pub, (long)(v & ((1ull << kAppendPhysBits) - 1ull)) + (long)behind * cm);
            rub = std::min(rub, (long)((v >> kAppendPhysBits) & ((1ull << kAppendRunBits) - 1ull)) + (long)behind * nr);
        }
    }
    PassTimer timer(c, (m.sparse || !m.table_valid) ? MF_PASS_COMPACTION : -1);
    if (pub < 0 || pub + cm > (long)m.cap || rub + nr > table) {
        require_dense(c, m);
        MF_HIP(c, hipStreamSynchronize(c->stream));
        const long n = (long)*m.h_count;
        m.phys_ub = n; m.runs_ub = (n + kRun - 1) / kRun;
        if (n + cm > (long)m.cap) { ok = false; return MF_OK; }
    }
    if (!m.table_valid) {
        launch_run_table(m.surf[m.cur], m.d_frame, c->stream);
        c->obj_dep_main = true;
        m.table_valid = true; m.gen++;
    }
    return MF_OK;
}
// bookkeeping behind a model's clean pass
static void after_clean(mf_ctx* c, ModelState& m, bool in_place) {
    if (in_place) { m.sparse = true; m.table_valid = true; m.phys_ub += cand_max(c); m.runs_ub += new_runs_max(c); m.clean_seq++; }
    else { m.sparse = false; m.table_valid = false; m.phys_ub = m.runs_ub = -1; m.mirror_from = m.clean_seq + 1; }
    m.gen++;
}
// the arguments every clean form of model m shares (shared scratch of the single-model path)
static CleanIn clean_in(mf_ctx* c, ModelState& m, int time_delta, bool packed, const float* depthF, const uint8_t* mask) {
    CleanIn in;
    in.frame = m.d_frame; in.pose = m.d_pose; in.W = c->W; in.H = c->H; in.k = c->K; in.timeDelta = time_delta; in.confThreshold = m.confThr;
    in.outlierCoeff = c->cfg.outlier_coefficient; in.maskID = m.id;
    in.index = c->d_index; in.vc = c->d_ivc; in.ct = c->d_ict; in.packed = packed ? c->d_iclean : nullptr; in.maskT = c->d_maskT;
    in.depthF = depthF; in.mask = mask; in.cand_op = c->d_cand_op; in.cand_rec = c->d_cand_rec;
    in.flags = c->d_flags; in.newconf = c->d_newconf; in.block_counts = c->d_block_counts; in.host_count = m.h_count;
    in.host_append = append_mirror_fits(c, m) ? m.h_append : nullptr; in.seq = m.clean_seq + 1;     // (after_clean counts the pass)
    in.transposed = packed; in.literalWindow = c->clean_literal;
    return in;
}
// Model::clean of m in place (m has a run table): the buffer's own surfels run by run, then the frame's candidates appended.  cull: visit only the
// runs in which a rule of the pass can apply (k_cull_clean; needs the decay statistics of THIS frame's packed resolve pass) -- the background;
// an object model's launch visits every run (its bounding box is the box of all its drawn surfels)
static void enqueue_clean_in_place(mf_ctx* c, ModelState& m, const CleanIn& in, bool cull, bool timed = false) {
    VisList cl{c->d_clean_list, c->d_clean_count};
    {
        PassTimer t(c, timed ? MF_PASS_BG_CLEAN : -1);
        if (cull)
            launch_cull_clean(m.surf[m.cur], m.d_frame, m.d_pose, c->W, c->H, c->K, in.timeDelta, in.confThreshold, c->d_decay_stats, c->d_clean_list,
                              c->d_clean_count, c->d_cull_ctl, (int)run_table_runs((long)m.cap, (long)c->P), c->stream);
        launch_clean_runs(in, m.surf[m.cur], cull ? &cl : nullptr, c->d_clean_ctl, clean_runs_grid((long)*m.h_count + kRun), c->stream);
    }
    PassTimer t(c, timed ? MF_PASS_BG_APPEND : -1);
    launch_clean_append(in, m.surf[m.cur], c->stream);
}

// predictIndices -> fuse -> [predictIndices] -> clean for one model (Core/MaskFusion.cpp:541-563 / :344-353)
static int enqueue_fuse_clean(mf_ctx* c, ModelState& m, const uint8_t* d_rgb, const float* d_depth, const float* depthF,
                              const uint8_t* mask, float fuseDepthCutoff, float weightMultiplier, bool secondIndexPass, bool marks) {
    const mf_config& g = c->cfg;
    const int W = c->W, H = c->H;
    hipStream_t s = c->stream;
    // Which forms a model's passes take is a matter of its size alone (mf_context.hip: in_place_elements / big_map_elements); the results are the same.
    bool small = clean_small(c, m);
    const bool copy = update_copy(c, m);
    if (!small) {
        bool ok = true;
        int rc = prepare_in_place(c, m, ok);
        if (rc != MF_OK) return rc;
        small = !ok;
    } else {
        require_dense(c, m);
    }
    const int src = m.cur, dst = 1 - m.cur;
    const int blocks = surfel_blocks(c, m);
    const bool timed = m.id == 0;       // "passTimings": the background's passes one by one
    VisList vl;
    const VisList* vis = nullptr;
    {
        PassTimer t(c, timed ? MF_PASS_BG_INDEX : -1);
        vis = ensure_vis(c, m, vl);
        // (column-major key images in both index passes of a model handled on its own: index_scatter_one; the resolve of this pass transposes)
        launch_index_scatter(m.surf[src], m.d_frame, m.d_pose, W, H, c->K, g.max_depth_processed, g.time_delta, c->d_keys, true, s, blocks, vis);
        launch_index_resolve(m.surf[src], m.d_frame, m.d_pose, c->d_keys, W, H, c->d_index, c->d_ivc, c->d_inr, secondIndexPass ? nullptr : c->d_ict,
                             nullptr, nullptr, nullptr, nullptr, true, s);
    }
    if (marks) mark(c, 4);
    // Model::fuse maxDepth uniform: min(depthCutoff, model.maxDepth, bb_max_z) (Model.cpp:527); bb_max_z from the model's bounding box on the device
    {
        PassTimer t(c, timed ? MF_PASS_BG_FUSE_DATA : -1);
        launch_fuse_data(d_rgb, d_depth, depthF, mask, m.id, m.d_frame, m.d_pose, weightMultiplier, fminf(fuseDepthCutoff, m.maxDepth), W, H,
                         c->K, c->d_index, c->d_ivc, c->d_inr, c->d_cand_op, c->d_cand_rec, c->d_upd_first, c->d_cand_best, s, c->bbox_limit ? 1 : 0);
    }
    if (marks) mark(c, 5);
    // the in-place clean of the background visits only the runs its rules can touch (k_cull_clean); the resolve pass that feeds clean gathers
    // the frame's statistics for that test
    const bool in_place = !small, cull_clean = in_place && c->cull_runs && secondIndexPass && m.id == 0;
    int live = src;      // the buffer that holds the updated surfels
    if (copy && small) {
        // small map (rounds 1-4's pass): update.vert as a copy src -> dst with the second index scatter (:556) riding on it; clean goes
        // dst -> src: two swaps leave the live buffer where it was
        {
            PassTimer t(c, timed ? MF_PASS_BG_FUSE_UPDATE : -1);
            launch_fuse_update_copy(m.surf[src], m.surf[dst], m.d_frame, c->d_upd_first, c->d_cand_rec, m.d_pose, W, H, c->K, g.max_depth_processed,
                                    g.time_delta, secondIndexPass ? c->d_keys : nullptr, true, s, blocks);
        }
        live = dst;
        if (marks) mark(c, 6);
        if (secondIndexPass) {
            PassTimer t(c, timed ? MF_PASS_BG_INDEX2 : -1);
            launch_index_resolve(m.surf[live], m.d_frame, m.d_pose, c->d_keys, W, H, nullptr, nullptr, nullptr, nullptr, c->d_iclean, depthF, mask, c->d_maskT, true, s);
        }
    } else {
        // update.vert in place -- only the surfels a candidate merged into are touched (the reference copies the whole buffer,
        // Model.cpp:583-646) --, then the second index pass (over the runs in view where the buffer has a run table)
        {
            PassTimer t(c, timed ? MF_PASS_BG_FUSE_UPDATE : -1);
            launch_fuse_update(m.surf[src], m.d_frame, c->d_upd_first, c->d_cand_op, c->d_cand_best, c->d_cand_rec, W, H, s);
        }
        if (marks) mark(c, 6);
        if (secondIndexPass) {   // predictIndices on the updated buffer (:556); its resolve writes the packed, column-major map of clean's window gathers
            PassTimer t(c, timed ? MF_PASS_BG_INDEX2 : -1);
            launch_index_scatter(m.surf[src], m.d_frame, m.d_pose, W, H, c->K, g.max_depth_processed, g.time_delta, c->d_keys, true, s, blocks, vis);
            launch_index_resolve(m.surf[src], m.d_frame, m.d_pose, c->d_keys, W, H, nullptr, nullptr, nullptr, nullptr, c->d_iclean, depthF, mask, c->d_maskT, true, s,
                                 cull_clean ? c->d_decay_stats : nullptr, m.id);
        }
    }
    // clean: two launches live -> the other buffer (a dense copy) below big_map_elements; from there on in place, run by run
    const CleanIn in = clean_in(c, m, g.time_delta, secondIndexPass, depthF, mask);
    if (in_place) {
        enqueue_clean_in_place(c, m, in, cull_clean, timed);
    } else {
        PassTimer t(c, timed ? MF_PASS_BG_CLEAN : -1);
        launch_clean_small(in, m.surf[live], m.surf[1 - live], s, compact_blocks_for((long)*m.h_count + cand_max(c)));
        m.cur = 1 - live;
    }
    after_clean(c, m, in_place);
    return MF_OK;
}

// MaskFusion::predict for one model: combinedPredict(maxDepthProcessed, tick, tick, timeDelta) -- the fill-in half
// (performFillIn) is evaluated lazily by the next tracking step from the retained filtered depth.
// advance: the end-of-frame bookkeeping of this model (processFrame's tail); the tiled prediction runs it as its epilogue, the scatter
// form is followed by k_frame_advance.
static void enqueue_predict(mf_ctx* c, ModelState& m, const FrameAdvance* advance = nullptr) {
    PassTimer timer(c, m.id == 0 ? MF_PASS_BG_PREDICT : -1);
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
                          const float* depthF, const uint8_t* mask, float weightMultiplier, const std::vector<float*>* log_slots, ObjBatch& b, int& blocks,
                          hipStream_t s = nullptr) {
    const mf_config& g = c->cfg;
    if (!s) s = c->stream;
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
        a.clean_ctl = m.scr.clean_ctl; a.host_count = m.h_count;
        a.host_append = append_mirror_fits(c, m) ? m.h_append : nullptr; a.clean_seq = m.clean_seq + 1;
        a.flags = m.scr.flags; a.newconf = m.scr.newconf; a.block_counts = m.scr.block_counts;
        a.predV = m.d_predV; a.predN = m.d_predN; a.predImage = m.d_predImage; a.predTime = m.d_predTime; a.predGray = gray ? m.d_predGray : nullptr;
        a.host_frame = m.h_frame; a.log_slot = log_slots ? (*log_slots)[i] : nullptr;
        a.global_payload = ((unsigned)orders[i] << 8) | ((unsigned)m.id & 255u);
        blocks = std::max(blocks, surfel_blocks(c, m));
    }
    MF_HIP(c, hipMemcpyAsync(c->d_obj_args[slot], h, sizeof(ObjPassArgs) * ms.size(), hipMemcpyHostToDevice, s));
    MF_HIP(c, hipEventRecord(c->ev_obj_args[slot], s));
    b.m = c->d_obj_args[slot]; b.n = (int)ms.size();
    b.W = c->W; b.H = c->H; b.k = c->K; b.maxDepthProcessed = g.max_depth_processed; b.globalMaxDepth = g.depth_cutoff; b.timeDelta = g.time_delta;
    b.outlierCoeff = g.outlier_coefficient; b.cleanLiteral = c->clean_literal ? 1 : 0; b.bboxLimit = c->bbox_limit ? 1 : 0; b.cleanSmall = 0; b.updateCopy = 0;
    b.denseSprites = 0;
    for (ModelState* m : ms) if ((long)*m->h_count >= (long)c->in_place_elements) b.denseSprites = 1;
    b.rgb = d_rgb; b.depthRaw = d_depth; b.depthF = depthF; b.mask = mask; b.maskT = s == c->stream ? c->d_maskT : c->d_maskT_obj; b.bg_pose = c->models[0]->d_pose; b.global_keys = c->d_keys;
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
// (It runs at the head of the frame's chain on the context's stream.  Running it one frame ahead on a stream of its own, beside the previous
// frame's fusion kernels, lost in rounds 2 and 5 -- also with that stream masked to 16 / 32 / 64 compute units: DESIGN.md, "Measured and rejected".)
// pyr_model: the model whose tracking step follows on this stream with nothing in between that its model-side pyramid depends on -- the pyramid
// (launch_model_pyramid's arguments, as enqueue_track passes them) is then built in the filter's launch
static int enqueue_preprocess(mf_ctx* c, const uint8_t* d_rgb, const float* d_depth, long k, bool with_maps, ModelState* pyr_model = nullptr,
                              const float* pyr_fill_depth = nullptr, const std::vector<ModelState*>* pyr_batch = nullptr) {
    const int W = c->W, H = c->H, P = c->P;
    hipStream_t s = c->stream;
    const mf_config& g = c->cfg;
    const int set = (int)(k & 1);
    float* depthF = c->d_depthF[k % 3];
    hipStream_t sp = s;
    mark(c, 0, sp);
    if (pyr_model) {
        ModelState& m = *pyr_model;
        launch_bilateral_model_pyramid(d_depth, depthF, m.d_predV, m.d_predN, m.allowFillIn ? pyr_fill_depth : nullptr, m.d_frame, m.d_pose, m.d_vmap_g,
                                       m.d_nmap_g, W, H, c->K, sp);
        c->pyr_done = pyr_model;
    } else if (pyr_batch) {   // the batched tracker's pyramids (enqueue_track_batch's first launch) beside the filter
        TrackBatch b;
        memset(&b, 0, sizeof(b));
        b.n = (int)pyr_batch->size();
        for (int i = 0; i < b.n; ++i) b.m[i] = (*pyr_batch)[i]->d_track;
        launch_bilateral_model_pyramid_batch(d_depth, depthF, b, pyr_fill_depth, W, H, c->K, sp);
        c->pyr_batch_done = true;
    } else {
        launch_bilateral(d_depth, depthF, W, H, sp);
    }
    if (with_maps) {   // the frame that initialises the map is never tracked against: no vertex / normal maps needed
        launch_frame_pyramid(depthF, c->d_vmap[set], c->d_nmap[set], W, H, c->K, g.depth_cutoff, sp);
    }
    c->cur_rgb = d_rgb;
    c->cur_depth = d_depth;
    if (photometric_on(c) || g.so3) {
        // imageBGRToIntensity + pyrDownUcharGauss of the frame (initRGB / initFirstRGB) and, for the photometric term,
        // computeDerivativeImages (RGBDOdometry.cpp:245-250)
        const float min_scale[3] = {rgb_min_scale(0), rgb_min_scale(1), rgb_min_scale(2)};
        if (c->fused_rgb_pyramid && launch_rgb_pyramid(d_rgb, W, H, c->d_gray[set], c->d_dIdx, c->d_dIdy, c->d_rgb_gate, min_scale, photometric_on(c), sp)) {
            c->gray_frame[set] = k;   // one launch: the three intensity levels and (photometric term) their derivative / gate images
            if (photometric_on(c)) c->deriv_frame = k;
        } else {
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
    }
    mark(c, 1, sp);
    return MF_OK;
}

static int download_pose_log(mf_ctx* c, ModelState& m, std::vector<int64_t>& ts, std::vector<float>& p7);

// GlobalProjection::project for one model (fixed confidence threshold 12, GlobalProjection.cpp:43-107).  The background model goes
// through the tile lists (its ~10^5..10^6 sprites cover millions of pixels: one memory-side atomic each in the scatter form);
// object models (a few thousand sprites) keep the scatter form, which costs them one short launch.  Both write the same keys.
static void enqueue_global_projection(mf_ctx* c, ModelState& m, int order) {
    const mf_config& g = c->cfg;
    PassTimer timer(c, m.id == 0 ? MF_PASS_BG_GLOBAL : -1);
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
        nm->age = 0; nm->isStatic = true; nm->log_ts.clear(); nm->cur = 0; nm->table_valid = false; nm->sparse = false; nm->phys_ub = nm->runs_ub = -1; nm->gen++; nm->mirror_from = nm->clean_seq + 1;
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
// which models the tracking loop tracks and which follow the background (MaskFusion.cpp:261-274); true: the tracked ones go through the batched loop
static bool tracking_plan(mf_ctx* c, size_t first, bool track_all, std::vector<ModelState*>& tracked, std::vector<ModelState*>& follow) {
    ModelState& bg = *c->models[0];
    tracked.clear(); follow.clear();
    if (first == 0) tracked.push_back(&bg);
    for (size_t i = 1; i < c->models.size(); ++i) {
        ModelState& m = *c->models[i];
        // trackable = trackableClassIds.empty() || trackableClassIds.count(classID), :261
        bool trackable = c->trackable.empty();
        for (int id : c->trackable) trackable |= (id == m.classID);
        if ((!m.isStatic || track_all) && trackable) tracked.push_back(&m);   // jump rule of :268-272 in the finalize step
        else follow.push_back(&m);
    }
    return !photometric_on(c) && tracked.size() >= 2 && (int)tracked.size() <= kMaxTrackBatch && c->batch_tracking;
}
static void enqueue_tracking_loop(mf_ctx* c, size_t first, bool track_all, const float* depthF_prev, long k) {
    ModelState& bg = *c->models[0];
    std::vector<ModelState*> tracked, follow;
    if (tracking_plan(c, first, track_all, tracked, follow)) {
        enqueue_track_batch(c, tracked, depthF_prev, k);
    } else {
        for (ModelState* m : tracked) enqueue_track(c, *m, m == &bg ? depthF_prev : nullptr, m == &bg ? 0.f : 0.2f, k);
    }
    for (ModelState* m : follow) launch_static_pose(m->d_pose, bg.d_pose, m->h_pose, c->stream);   // updateStaticPose, :274
    for (auto& m : c->models) m->gen++;   // the poses changed: cached visibility lists are stale
}

// The fusion loop of processFrame (Core/MaskFusion.cpp:539-565) over models[first..]: predictIndices -> fuse -> predictIndices -> clean;
// the object models go through one launch per pass when there are at least two of them ("batchObjectPasses").
static int enqueue_fusion_loop(mf_ctx* c, size_t first, bool multi, const uint8_t* d_rgb, const float* d_depth, const float* depthF,
                               const uint8_t* mask, float weight_multiplier) {
    const mf_config& g = c->cfg;
    const bool batch = multi && batch_objects_now(c);
    for (size_t i = first; i < (batch ? (size_t)1 : c->models.size()); ++i) {
        int rc = enqueue_fuse_clean(c, *c->models[i], d_rgb, d_depth, depthF, mask, g.depth_cutoff, weight_multiplier, true, i == 0);
        if (rc != MF_OK) return rc;
    }
    if (batch) {   // every object model: one launch per pass
        std::vector<ModelState*> objs; std::vector<int> orders;
        object_models(c, objs, orders);
        // one form per launch: the batch takes the form of its largest model.  In place (every model keeps a run table) when a model is big and
        // none of them is within a frame's candidates of its capacity; else the two-launch clean on dense buffers
        bool any_big = false, in_place = true;
        for (ModelState* m : objs) any_big |= !clean_small(c, *m);
        if (any_big) {
            for (ModelState* m : objs) {
                bool ok = true;
                int rc = prepare_in_place(c, *m, ok);
                if (rc != MF_OK) return rc;
                in_place &= ok;
            }
        } else in_place = false;
        if (!in_place) for (ModelState* m : objs) require_dense(c, *m);
        ObjBatch ob; int blocks = 0;
        obj_stream_catch_up(c);
        int rc = make_obj_batch(c, objs, orders, d_rgb, d_depth, depthF, mask, weight_multiplier, nullptr, ob, blocks, c->obj_s);   // (after the compactions: they change the live buffer)
        if (rc != MF_OK) return rc;
        int cblocks = 64;
        for (ModelState* m : objs) cblocks = std::max(cblocks, clean_runs_grid((long)*m->h_count + kRun));
        ob.cleanSmall = in_place ? 0 : 1; ob.updateCopy = in_place ? 0 : 1;
        if (!in_place) for (ModelState* m : objs) if (!update_copy(c, *m)) ob.updateCopy = 0;
        {
            PassTimer timer(c, MF_PASS_OBJ_FUSE_CLEAN, c->obj_s);
            long most = 0;
            for (ModelState* m : objs) most = std::max(most, (long)*m->h_count);
            launch_obj_fuse_clean(ob, blocks, cblocks, c->obj_s, in_place ? kCompactBlocks : compact_blocks_for(most + cand_max(c)));
        }
        for (ModelState* m : objs) {   // copy-update: a -> b -> a; in-place update + two-launch clean: a -> b -- b is the live buffer now; in place: a
            if (!in_place && !ob.updateCopy) m->cur = 1 - m->cur;
            after_clean(c, *m, in_place);
        }
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
        obj_stream_catch_up(c);
        int rc = make_obj_batch(c, objs, orders, d_rgb, d_depth, depthF, mask, weight_multiplier, &slots, ob, blocks, c->obj_s);
        if (rc != MF_OK) return rc;
        PassTimer timer(c, MF_PASS_OBJ_PREDICT, c->obj_s);
        launch_obj_predict_advance(ob, blocks, c->obj_s);
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
    owned->id = -1; owned->classID = -1; owned->age = 0; owned->isStatic = true; owned->log_ts.clear(); owned->cur = 0; owned->table_valid = false; owned->sparse = false; owned->phys_ub = owned->runs_ub = -1; owned->gen++; owned->mirror_from = owned->clean_seq + 1; owned->pred_gray_valid = false;
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
    bool bg_fused = false;
    c->mm_marked = false;
    ObjStreamWindow obj_window(c);   // (joins on every way out of this function)

    // the background's tracking step follows the preprocessing directly (model by model: the batched loop builds every model's pyramid in its own
    // launch): its model-side pyramid rides in the depth filter's launch
    ModelState* pyr_model = nullptr;
    std::vector<ModelState*> pyr_tracked, pyr_follow;
    bool pyr_batch = false;
    c->pyr_done = nullptr; c->pyr_batch_done = false;
    if (c->fused_preprocess && c->map_ready && !(in_pose16 && !bootstrap)) {
        if (tracking_plan(c, 0, g.track_all_models != 0, pyr_tracked, pyr_follow)) pyr_batch = true;   // ... or every tracked model's, batched
        else pyr_model = &bg;
    }
    int prc = enqueue_preprocess(c, d_rgb, d_depth, k, c->map_ready, pyr_model, depthF_prev, pyr_batch ? &pyr_tracked : nullptr);
    if (prc != MF_OK) return prc;

    if (!c->map_ready) {
        c->map_ready = true;
        mark(c, 2); mark(c, 3); mark(c, 4); mark(c, 5); mark(c, 6);
        // :235-238
        launch_init_surfels(d_rgb, d_depth, depthF, W, H, c->K, g.max_depth_processed, bg.d_frame, c->d_cand_rec, c->d_flags, s);
        bg.cur = 0;
        launch_compact_records(c->d_cand_rec, c->d_flags, P, bg.surf[0], bg.d_frame, c->d_block_counts, bg.h_count, s);
        launch_run_table(bg.surf[0], bg.d_frame, s);
        bg.table_valid = true; bg.sparse = false; bg.phys_ub = P; bg.runs_ub = ((long)P + kRun - 1) / kRun; bg.gen++; bg.mirror_from = bg.clean_seq + 1;
        mark(c, 7);
    } else if (in_pose16 && !bootstrap) {
        // the caller supplies the camera pose: no tracking, no segmentation, object poses untouched
        // (MaskFusion.cpp:243,413-415 -- the whole "regular" block is skipped)
        mark(c, 2);
        launch_override_pose(bg.d_pose, in_pose16, 0, bg.h_pose, s);
        bg.gen++;
        mark(c, 3); mark(c, 4);
        if (!g.rgb_only)   // :539
          for (size_t i = 0; i < c->models.size(); ++i) {
            int rc = enqueue_fuse_clean(c, *c->models[i], d_rgb, d_depth, depthF, mask, g.depth_cutoff, weight_multiplier, true, i == 0);
            if (rc != MF_OK) return rc;
          }
        mark(c, 7);
    } else {
        mark(c, 2);
        // tracking, :247-276.  Every model that is tracked this frame goes into one batch (geometric term) or is tracked on its
        // own (photometric term: its scratch images are shared); static objects then follow the background's NEW pose
        enqueue_tracking_loop(c, 0, g.track_all_models != 0, depthF_prev, k);
        if (bootstrap && in_pose16) { launch_override_pose(bg.d_pose, in_pose16, 1, bg.h_pose, s); bg.gen++; }   // :280-283 (after the object loop)
        mark(c, 3);

        if (multi) {
            // GlobalProjection::project(models, tick, tick, timeDelta, depthCutoff) (:289) with its fixed threshold 12
            if (batch_objects_now(c)) {
                // (the objects' scatter on the object stream beside the background's culling / binning passes and the edge maps -- "globalOverlapElements",
                // round 6 -- gave 4.158 -> 4.134 ms on configs[4] and lost 2 % on S2: removed, DESIGN.md "Measured and rejected")
                enqueue_global_projection(c, bg, 0);
                std::vector<ModelState*> objs; std::vector<int> orders;
                object_models(c, objs, orders);
                ObjBatch ob; int blocks = 0;
                int rc = make_obj_batch(c, objs, orders, d_rgb, d_depth, depthF, mask, weight_multiplier, nullptr, ob, blocks);
                if (rc != MF_OK) return rc;
                PassTimer timer(c, MF_PASS_OBJ_GLOBAL);
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
                    int rc2 = enqueue_fuse_clean(c, bg, d_rgb, d_depth, depthF, mask, g.depth_cutoff, weight_multiplier, true, true);
                    if (rc2 != MF_OK) return rc2;
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
            // the label image is complete and the host knows it: from here on the object models' batched passes have a stream of their own
            obj_window.begin();

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
                c->obj_dep_main = true;   // (the label image arrives on the main stream)
            }
            bool spawned = false;
            if (res.hasNewLabel && (int)c->models.size() < g.max_models) {
                int rc = spawn_object(c, take_next_model_id(c), res.newClassID);
                if (rc != MF_OK) return rc;
                c->obj_dep_main = true;   // (its initialisation and first fusion are main-stream work the object chain has to see)
                c->spawnOffset = 0;
                spawned = true;
            }
            for (size_t i = 1; i < c->models.size(); ++i) c->models[i]->maxDepth = 30.f + 30.f * 1.2f;  // :335-339 (depthMean = depthStd = 30)
            if (spawned) {  // :342-353: predictIndices; fuse(maxDepthProcessed, weight 100); clean (no second index pass)
                int rc = enqueue_fuse_clean(c, *c->models.back(), d_rgb, d_depth, depthF, mask, g.max_depth_processed, 100.f, false, false);
                if (rc != MF_OK) return rc;
            }
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
    if (c->pass_timings_on)
        for (int q = 0; q < MF_N_PASSES; ++q) {
            float ms = 0.f;
            if (c->pass_recorded[q] && hipEventElapsedTime(&ms, c->ev_pass[q][0], c->ev_pass[q][1]) == hipSuccess) c->pass_ms[q] = ms;
        }
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
    hipStream_t sin = c->stream;
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

// mf_ktest.inl -- kernel-level entry points (mf_k_*: one reference CUDA wrapper or GLSL pass each) for the parity tests.
// (part of mf_context.hip: the library's host side is ONE translation unit -- the context type and its helpers are file-local -- kept in
// four files by subject; mf_context.hip includes them in this order)

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

// A scripted stand-in for libmaskfusion_amd.so's C ABI (only the entry points the facade uses here) so that the host-side logic of
// include/maskfusion/MaskFusion.h -- the frame queue (Core/MaskFusion.cpp:37,206-209), the model list kept as
// std::list<std::shared_ptr<Model>> across frames, the new / inactive model listeners (Core/MaskFusion.h:303-306) -- can be tested on
// a machine without a GPU.  TEST CODE: nothing of the product links this.
// Script: a frame whose timestamp is 102 spawns object model id 1 (class 41); 104 drops id 1 and spawns id 7 (class 42); 105 drops 7.
#include <maskfusion_amd.h>

#include <cstring>
#include <string>
#include <vector>

struct mf_ctx {
    mf_config cfg;
    int tick = 1;
    std::vector<mf_model_info_t> models;
    std::vector<long long> processed;   // timestamps in processing order
    std::string err;
    double params[8] = {0, 0, 0, 0, 0, 0, 0, 0};
};
static mf_ctx* g_last = nullptr;
extern "C" {
int mf_default_config(mf_config* c, int32_t w, int32_t h, float fx, float fy, float cx, float cy) {
    std::memset(c, 0, sizeof(*c));
    c->width = w; c->height = h; c->fx = fx; c->fy = fy; c->cx = cx; c->cy = cy; c->time_delta = 200; c->enable_multiple_models = 1;
    return MF_OK;
}
int mf_create(const mf_config* cfg, mf_ctx** out) {
    mf_ctx* c = new mf_ctx;
    c->cfg = *cfg;
    mf_model_info_t bg{0, -1, 1000u, cfg->conf_global, 1, 0u};
    c->models.push_back(bg);
    *out = c;
    g_last = c;
    return MF_OK;
}
void mf_destroy(mf_ctx* c) { if (g_last == c) g_last = nullptr; delete c; }
const char* mf_last_error(const mf_ctx* c) { return c ? c->err.c_str() : "null"; }
int mf_process_frame(mf_ctx* c, const uint8_t*, const float*, const uint8_t*, const int32_t*, int32_t, int64_t ts, const float*, float, int32_t) {
    c->processed.push_back(ts);
    auto drop = [&](int id) { for (size_t i = 1; i < c->models.size(); ++i) if (c->models[i].id == id) { c->models.erase(c->models.begin() + (long)i); return; } };
    if (ts == 102) c->models.push_back(mf_model_info_t{1, 41, 10u, 0.f, 1, 0u});
    if (ts == 104) { drop(1); c->models.push_back(mf_model_info_t{7, 42, 20u, 0.f, 1, 0u}); }
    if (ts == 105) drop(7);
    for (auto& m : c->models) m.age++;
    c->tick++;
    return MF_OK;
}
int mf_get_tick(mf_ctx* c, int32_t* t) { *t = c->tick; return MF_OK; }
int mf_num_models(mf_ctx* c, int32_t* n) { *n = (int32_t)c->models.size(); return MF_OK; }
int mf_model_info(mf_ctx* c, int32_t i, mf_model_info_t* out) {
    if (i < 0 || i >= (int32_t)c->models.size()) { c->err = "no such model"; return MF_EINVAL; }
    *out = c->models[(size_t)i];
    return MF_OK;
}
int mf_get_pose(mf_ctx* c, int32_t i, float* p) {
    if (i < 0 || i >= (int32_t)c->models.size()) { c->err = "no such model"; return MF_EINVAL; }
    for (int k = 0; k < 16; ++k) p[k] = (k % 5 == 0) ? 1.f : 0.f;
    p[12] = (float)c->models[(size_t)i].id;   // tx = id: lets the test see WHICH model answered
    return MF_OK;
}
int mf_get_param(mf_ctx* c, const char* key, double* v) {
    *v = !std::strcmp(key, "enableMultipleModels") ? c->cfg.enable_multiple_models : !std::strcmp(key, "timeDelta") ? c->cfg.time_delta : 0.0;
    return MF_OK;
}
static double g_ftf = -1.0;   // last value mf_set_param("frameToFrameRGB", .) received (-1: never set)
int mf_set_param(mf_ctx* c, const char* key, double v) {
    if (!std::strcmp(key, "enableMultipleModels")) c->cfg.enable_multiple_models = (int)v;
    if (!std::strcmp(key, "frameToFrameRGB")) g_ftf = v;
    return MF_OK;
}
double stub_param(const char* key) { return !std::strcmp(key, "frameToFrameRGB") ? g_ftf : -1.0; }
int mf_export_segmentation_png(mf_ctx*, const char*) { return MF_OK; }
// what the test reads back
int stub_processed(long long* out, int max) {
    int n = 0;
    if (g_last) for (long long t : g_last->processed) if (n < max) out[n++] = t;
    return n;
}
}

/*
 * mfgl_api.cpp -- runs the reference's shaders (compiled as C++ by oracle/build_glsl.py, namespaces mfgl::sh_*) the way the host
 * code of martinruenz/maskfusion draws them, and returns what the passes leave in their buffers through plain-pointer C entry
 * points (bound by oracle/mfglsl.py).  TEST INFRASTRUCTURE ONLY.
 *
 * This file is appended to the translation unit after the shader namespaces (it is not compiled on its own).  Everything the
 * shaders compute comes from their text; what is stated HERE is what OpenGL does around them, by the documented rules
 * (DESIGN.md 2b): vertices in buffer order; a point lands in the pixel that contains its window position; a point sprite of size s
 * covers the pixel centres in [u - s/2, u + s/2) (size clamped to [1, 64] px); depth test LESS, earlier primitive wins ties;
 * transform feedback appends the emitted vertices in order.  The geometry shaders are one-line emit conditions, restated at the
 * call site of the vertex shader they follow (vertex_feedback.geom:35, data.geom:38, copy_unstable.geom:34).
 */
#include <stdlib.h>
#include <vector>

namespace mfgl {
vec4 gl_Position; vec4 gl_FragCoord; float gl_PointSize = 1.f; float gl_FragDepth = 0.f; int gl_VertexID = 0; bool g_discarded = false;
}

using namespace mfgl;

namespace {
struct Cam { int W, H; float fx, fy, cx, cy; };
inline mat4 mat4_from(const float* m16) { mat4 M; memcpy(M.m, m16, sizeof(M.m)); return M; }
/* the t_inv uniform: pose.inverse() on the host (Eigen upstream); computed by the oracle's routine so that both sides of a
 * comparison hand their projection the same sixteen floats */
inline void rigid_inverse(const float* p, float* o) { mfo_pose_inverse16(p, o); }
inline std::vector<float> rgba_from_rgb8(const uint8_t* rgb, int P) {   /* GL_RGB upload into GL_RGBA8, sampled as normalised floats */
    std::vector<float> o((size_t)P * 4);
    for (int i = 0; i < P; ++i) {
        o[(size_t)i * 4 + 0] = (float)rgb[i * 3 + 0] / 255.0f; o[(size_t)i * 4 + 1] = (float)rgb[i * 3 + 1] / 255.0f;
        o[(size_t)i * 4 + 2] = (float)rgb[i * 3 + 2] / 255.0f; o[(size_t)i * 4 + 3] = 1.0f;
    }
    return o;
}
/* the uv buffer: one element per pixel in COLUMN-major order (FeedbackBuffer.cpp:44-50, Model.cpp builds the same) */
inline vec2 uv_of(int i, int j, int W, int H) {
    return vec2((float)(((float)i / (float)W) + 1.0 / (2 * (float)W)), (float)(((float)j / (float)H) + 1.0 / (2 * (float)H)));
}
/* window position of a clip-space point with w = 1 (viewport = the whole image); false if it is clipped */
inline bool window_pos(const vec4& p, int W, int H, double& xw, double& yw) {
    if (!(p.x >= -p.w && p.x <= p.w && p.y >= -p.w && p.y <= p.w && p.z >= -p.w && p.z <= p.w)) return false;
    xw = ((double)p.x / p.w + 1.0) * 0.5 * W;
    yw = ((double)p.y / p.w + 1.0) * 0.5 * H;
    return true;
}
}  // namespace

extern "C" {

/* MaskFusion::filterDepth (Core/MaskFusion.cpp:650-657) with depth_bilateral_metric.frag; libm_exp: exp() as the C library has it
 * (the oracle's restatement of this one shader uses expf) */
static int g_exp_libm = 0;
void mfglsl_bilateral(const float* depth, float* out, int W, int H, float maxD) {
    namespace S = sh_depth_bilateral_metric_frag;
    g_exp_libm = 1;
    S::gSampler = sampler2D{depth, W, H, 1};
    S::cols = (float)W; S::rows = (float)H; S::maxD = maxD;
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            S::texcoord = vec2(((float)x + 0.5f) / (float)W, ((float)y + 0.5f) / (float)H);
            S::main();
            out[y * W + x] = S::FragColor;
        }
    g_exp_libm = 0;
}

/* computeFeedbackBuffers (Core/MaskFusion.cpp:188-198: RAW from the raw depth, FILTERED from the filtered depth, each compacted by
 * its own geometry shader) + Model::initialise (Model.cpp:240-285: attributes 0, 1 from RAW, 2 from FILTERED, paired BY POSITION in
 * the two compacted buffers) with init_unstable.vert.  Returns the vertex count of the RAW buffer (what glDrawTransformFeedback
 * draws); *filtered_count receives the FILTERED buffer's. */
int mfglsl_init_surfels(const uint8_t* rgb, const float* depthRaw, const float* depthF, int W, int H, float fx, float fy, float cx, float cy,
                        int time, float maxDepth, float* surfels, int capacity, int* filtered_count) {
    namespace S = sh_vertex_feedback_vert;
    const std::vector<float> rgba = rgba_from_rgb8(rgb, W * H);
    std::vector<float> buf[2];
    for (int pass = 0; pass < 2; ++pass) {
        S::gSampler = sampler2D{pass == 0 ? depthRaw : depthF, W, H, 1};
        S::cSampler = sampler2D{rgba.data(), W, H, 4};
        S::cam = vec4(cx, cy, 1.0f / fx, 1.0f / fy);
        S::cols = (float)W; S::rows = (float)H; S::time = time; S::maxDepth = maxDepth;
        for (int i = 0; i < W; ++i)
            for (int j = 0; j < H; ++j) {
                S::texcoord = uv_of(i, j, W, H);
                S::main();
                if (S::zVal > 0) {   /* vertex_feedback.geom:35 */
                    const float v[12] = {S::vPosition.x, S::vPosition.y, S::vPosition.z, S::vPosition.w, S::vColor.x, S::vColor.y, S::vColor.z,
                                         S::vColor.w, S::vNormRad.x, S::vNormRad.y, S::vNormRad.z, S::vNormRad.w};
                    buf[pass].insert(buf[pass].end(), v, v + 12);
                }
            }
    }
    namespace I = sh_init_unstable_vert;
    const int n = (int)(buf[0].size() / 12);
    if (filtered_count) *filtered_count = (int)(buf[1].size() / 12);
    int count = 0;
    for (int k = 0; k < n && count < capacity; ++k) {
        const float* a = &buf[0][(size_t)k * 12];
        I::vPosition = vec4(a[0], a[1], a[2], a[3]);
        I::vColor = vec4(a[4], a[5], a[6], a[7]);
        if ((size_t)k * 12 + 11 < buf[1].size()) { const float* b = &buf[1][(size_t)k * 12]; I::vNormRad = vec4(b[8], b[9], b[10], b[11]); }
        else I::vNormRad = vec4(0, 0, 0, 0);   /* the FILTERED buffer is shorter: the attribute reads stale / zero memory upstream */
        I::main();
        float* s = surfels + (size_t)count * 12;
        s[0] = I::vPosition0.x; s[1] = I::vPosition0.y; s[2] = I::vPosition0.z; s[3] = I::vPosition0.w;
        s[4] = I::vColor0.x; s[5] = I::vColor0.y; s[6] = I::vColor0.z; s[7] = I::vColor0.w;
        s[8] = I::vNormRad0.x; s[9] = I::vNormRad0.y; s[10] = I::vNormRad0.z; s[11] = I::vNormRad0.w;
        ++count;
    }
    return count;
}

/* ModelProjection::predictIndices (ModelProjection.cpp:100-152): index_map.vert / .frag, 1-px points, FACTOR = 1.
 * index: int32[P] (cleared to 0), the three float4 attachments cleared to 0. */
void mfglsl_predict_indices(const float* pose16, const float* surfels, int count, int time, float maxDepth, int timeDelta, int W, int H,
                            float fx, float fy, float cx, float cy, int32_t* index, float* vertConf, float* colorTime, float* normRad) {
    namespace V = sh_index_map_vert;
    namespace F = sh_index_map_frag;
    const int P = W * H;
    std::vector<float> zbuf((size_t)P, 1.0f);   /* glClear depth = 1 */
    memset(index, 0, sizeof(int32_t) * P);
    memset(vertConf, 0, sizeof(float) * 4 * P); memset(colorTime, 0, sizeof(float) * 4 * P); memset(normRad, 0, sizeof(float) * 4 * P);
    float tinv[16];
    rigid_inverse(pose16, tinv);
    V::t_inv = mat4_from(tinv);
    V::cam = vec4(cx, cy, fx, fy);
    V::cols = (float)W; V::rows = (float)H; V::maxDepth = maxDepth; V::time = time; V::timeDelta = timeDelta;
    for (int i = 0; i < count; ++i) {
        const float* s = surfels + (size_t)i * 12;
        V::vPosition = vec4(s[0], s[1], s[2], s[3]); V::vColorTime = vec4(s[4], s[5], s[6], s[7]); V::vNormRad = vec4(s[8], s[9], s[10], s[11]);
        gl_VertexID = i;
        V::main();
        double xw, yw;
        if (!window_pos(gl_Position, W, H, xw, yw)) continue;
        const int px = (int)floor(xw), py = (int)floor(yw);
        if (px < 0 || py < 0 || px >= W || py >= H) continue;
        const float depth = (gl_Position.z / gl_Position.w + 1.0f) * 0.5f;
        const int p = py * W + px;
        if (!(depth < zbuf[p])) continue;   /* GL_LESS */
        zbuf[p] = depth;
        F::vPosition0 = V::vPosition0; F::vColorTime0 = V::vColorTime0; F::vNormRad0 = V::vNormRad0; F::vertexId = V::vertexId;
        F::main();
        index[p] = F::FragColor;
        memcpy(vertConf + (size_t)p * 4, &F::vPosition1.x, 16); memcpy(colorTime + (size_t)p * 4, &F::vColorTime1.x, 16);
        memcpy(normRad + (size_t)p * 4, &F::vNormRad1.x, 16);
    }
}

/* Model::fuse, PROGRAM1 (Model.cpp:466-581): data.vert for every pixel in the uv buffer's (column-major) order.  Per pixel k of that
 * order: op[k] = updateId (0 none, 1 merge, 2 new), best[k] = the surfel index encoded in gl_Position (op 1), rec[k] = the emitted
 * vertex (data.geom:38 emits when updateId > 0). */
void mfglsl_fuse_data(const float* pose16, const uint8_t* rgb, const float* depthRaw, const float* depthF, const uint8_t* mask, int maskID,
                      int time, float weighting, float maxDepth, int W, int H, float fx, float fy, float cx, float cy, const int32_t* index,
                      const float* vertConf, const float* colorTime, const float* normRad, int texDim, uint8_t* op, int32_t* best, float* rec) {
    namespace S = sh_data_vert;
    const int P = W * H;
    const std::vector<float> rgba = rgba_from_rgb8(rgb, P);
    std::vector<uint32_t> idx32((size_t)P), mask32((size_t)P);
    for (int i = 0; i < P; ++i) { idx32[i] = (uint32_t)index[i]; mask32[i] = mask[i]; }
    S::cSampler = sampler2D{rgba.data(), W, H, 4}; S::drSampler = sampler2D{depthRaw, W, H, 1}; S::drfSampler = sampler2D{depthF, W, H, 1};
    S::indexSampler = usampler2D{idx32.data(), W, H}; S::maskSampler = usampler2D{mask32.data(), W, H};
    S::vertConfSampler = sampler2D{vertConf, W, H, 4}; S::colorTimeSampler = sampler2D{colorTime, W, H, 4}; S::normRadSampler = sampler2D{normRad, W, H, 4};
    S::cam = vec4(cx, cy, 1.0f / fx, 1.0f / fy);
    S::cols = (float)W; S::rows = (float)H; S::scale = 1.0f; S::texDim = (float)texDim; S::pose = mat4_from(pose16);
    S::minDepth = 1.17549435e-38f; S::maxDepth = maxDepth; S::time = (float)time; S::weighting = weighting; S::maskID = (uint)maskID;
    int k = 0;
    for (int i = 0; i < W; ++i)
        for (int j = 0; j < H; ++j, ++k) {
            S::texcoord = uv_of(i, j, W, H);
            S::main();
            op[k] = (uint8_t)S::updateId;
            best[k] = 0;
            if (S::updateId == 1) {   /* the texel of the update map the point lands in = the surfel to update */
                const double xw = ((double)gl_Position.x + 1.0) * 0.5 * texDim, yw = ((double)gl_Position.y + 1.0) * 0.5 * texDim;
                best[k] = (int)floor(yw) * texDim + (int)floor(xw);
            }
            float* r = rec + (size_t)k * 12;
            r[0] = S::vPosition.x; r[1] = S::vPosition.y; r[2] = S::vPosition.z; r[3] = S::vPosition.w;
            r[4] = S::vColor.x; r[5] = S::vColor.y; r[6] = S::vColor.z; r[7] = S::vColor.w;
            r[8] = S::vNormRad.x; r[9] = S::vNormRad.y; r[10] = S::vNormRad.z; r[11] = S::vNormRad.w;
        }
}

/* Model::fuse, PROGRAM2 (Model.cpp:583-646): the update map (what data.frag left: first fragment per texel wins, depth test LESS at
 * constant z) is rebuilt from the data pass's outputs, then update.vert runs for every surfel. */
void mfglsl_fuse_update(const float* src, float* dst, int count, int time, int texDim, const uint8_t* op, const int32_t* best, const float* rec, int n_px) {
    namespace S = sh_update_vert;
    const size_t T = (size_t)texDim * texDim;
    std::vector<float> vert(T * 4, 0.f), col(T * 4, 0.f), nrm(T * 4, 0.f);
    std::vector<uint8_t> written(T, 0);
    for (int k = 0; k < n_px; ++k)
        if (op[k] == 1 && best[k] >= 0 && (size_t)best[k] < T && !written[best[k]]) {
            written[best[k]] = 1;
            memcpy(&vert[(size_t)best[k] * 4], rec + (size_t)k * 12, 16); memcpy(&col[(size_t)best[k] * 4], rec + (size_t)k * 12 + 4, 16);
            memcpy(&nrm[(size_t)best[k] * 4], rec + (size_t)k * 12 + 8, 16);
        }
    S::vertSamp = sampler2D{vert.data(), texDim, texDim, 4}; S::colorSamp = sampler2D{col.data(), texDim, texDim, 4};
    S::normSamp = sampler2D{nrm.data(), texDim, texDim, 4};
    S::texDim = (float)texDim; S::time = time;
    for (int i = 0; i < count; ++i) {
        const float* s = src + (size_t)i * 12;
        S::vPosition = vec4(s[0], s[1], s[2], s[3]); S::vColor = vec4(s[4], s[5], s[6], s[7]); S::vNormRad = vec4(s[8], s[9], s[10], s[11]);
        gl_VertexID = i;
        S::main();
        float* d = dst + (size_t)i * 12;
        d[0] = S::vPosition0.x; d[1] = S::vPosition0.y; d[2] = S::vPosition0.z; d[3] = S::vPosition0.w;
        d[4] = S::vColor0.x; d[5] = S::vColor0.y; d[6] = S::vColor0.z; d[7] = S::vColor0.w;
        d[8] = S::vNormRad0.x; d[9] = S::vNormRad0.y; d[10] = S::vNormRad0.z; d[11] = S::vNormRad0.w;
    }
}

/* Model::clean (Model.cpp:649-772): copy_unstable.vert over the map, then over every vertex the data pass emitted (merges carry
 * w = -1 and are dropped by the shader), copy_unstable.geom:34 emits when test > 0.  Returns the new count. */
int mfglsl_clean(const float* pose16, const float* src, int count, const uint8_t* op, const float* rec, int n_px, int time, int timeDelta,
                 float confThreshold, float maxDepth, float outlierCoeff, int maskID, int W, int H, float fx, float fy, float cx, float cy,
                 const int32_t* index, const float* vertConf, const float* colorTime, const float* normRad, const float* depthF,
                 const uint8_t* mask, float* dst, int capacity, uint8_t* keep /* [count + emitted], may be NULL */) {
    namespace S = sh_copy_unstable_vert;
    const int P = W * H;
    std::vector<uint32_t> idx32((size_t)P), mask32((size_t)P);
    for (int i = 0; i < P; ++i) { idx32[i] = (uint32_t)index[i]; mask32[i] = mask[i]; }
    static const float zero4[4] = {0, 0, 0, 0};
    float tinv[16];
    rigid_inverse(pose16, tinv);
    S::time = time; S::scale = 1.0f; S::outlierCoeff = outlierCoeff; S::t_inv = mat4_from(tinv); S::cam = vec4(cx, cy, fx, fy);
    S::cols = (float)W; S::rows = (float)H; S::confThreshold = confThreshold;
    S::indexSampler = usampler2D{idx32.data(), W, H}; S::maskSampler = usampler2D{mask32.data(), W, H};
    S::vertConfSampler = sampler2D{vertConf, W, H, 4}; S::colorTimeSampler = sampler2D{colorTime, W, H, 4}; S::normRadSampler = sampler2D{normRad, W, H, 4};
    S::nodeSampler = sampler2D{zero4, 1, 1, 1}; S::depthSamplerPrediction = sampler2D{zero4, 1, 1, 1}; S::depthSamplerInput = sampler2D{depthF, W, H, 1};
    S::nodes = 0.f; S::nodeCols = 1.f; S::maxDepth = maxDepth; S::timeDelta = timeDelta; S::isFern = 0; S::maskID = (uint)maskID;
    int n = 0, slot = 0;
    auto run = [&](const float* s) {
        S::vPos = vec4(s[0], s[1], s[2], s[3]); S::vCol = vec4(s[4], s[5], s[6], s[7]); S::vNormR = vec4(s[8], s[9], s[10], s[11]);
        S::main();
        if (keep) keep[slot] = (uint8_t)(S::test > 0);
        ++slot;
        if (S::test > 0 && n < capacity) {
            float* d = dst + (size_t)n * 12;
            d[0] = S::vPosition.x; d[1] = S::vPosition.y; d[2] = S::vPosition.z; d[3] = S::vPosition.w;
            d[4] = S::vColor.x; d[5] = S::vColor.y; d[6] = S::vColor.z; d[7] = S::vColor.w;
            d[8] = S::vNormRad.x; d[9] = S::vNormRad.y; d[10] = S::vNormRad.z; d[11] = S::vNormRad.w;
            ++n;
        }
    };
    for (int i = 0; i < count; ++i) run(src + (size_t)i * 12);
    for (int k = 0; k < n_px; ++k)
        if (op[k] > 0) run(rec + (size_t)k * 12);   /* the new-unstable buffer holds every vertex data.geom emitted, in order */
    return n;
}

/* ModelProjection::combinedPredict (ModelProjection.cpp:187-268): splat.vert + combo_splat.frag, point sprites.
 * image: RGBA8 [P*4], vertexConf / normalRadius: float4 [P], time: uint16 [P]; all cleared to 0. */
void mfglsl_combined_predict(const float* pose16, const float* surfels, int count, float maxDepth, float confThreshold, int time, int maxTime,
                             int timeDelta, int W, int H, float fx, float fy, float cx, float cy, uint8_t* image, float* vertexConf,
                             float* normalRadius, uint16_t* timeOut) {
    namespace V = sh_splat_vert;
    namespace F = sh_combo_splat_frag;
    const int P = W * H;
    std::vector<float> zbuf((size_t)P, 1.0f);
    memset(image, 0, (size_t)P * 4); memset(vertexConf, 0, sizeof(float) * 4 * P); memset(normalRadius, 0, sizeof(float) * 4 * P);
    memset(timeOut, 0, sizeof(uint16_t) * P);
    float tinv[16];
    rigid_inverse(pose16, tinv);
    V::t_inv = mat4_from(tinv); V::cam = vec4(cx, cy, fx, fy); V::cols = (float)W; V::rows = (float)H; V::maxDepth = maxDepth;
    V::confThreshold = confThreshold; V::time = time; V::maxTime = maxTime; V::timeDelta = timeDelta;
    F::cam = vec4(cx, cy, fx, fy); F::maxDepth = maxDepth;
    for (int i = 0; i < count; ++i) {
        const float* s = surfels + (size_t)i * 12;
        V::vPosition = vec4(s[0], s[1], s[2], s[3]); V::vColor = vec4(s[4], s[5], s[6], s[7]); V::vNormRad = vec4(s[8], s[9], s[10], s[11]);
        V::main();
        double u, v;
        if (!window_pos(gl_Position, W, H, u, v)) continue;
        float size = gl_PointSize;
        if (!(size > 0.f)) continue;                       /* NaN sizes too */
        size = size < 1.0f ? 1.0f : (size > 64.0f ? 64.0f : size);   /* point size range: [1, 64] here (NVIDIA: [1, 2047]) */
        const double half = size * 0.5;
        int px0 = (int)ceil(u - half - 0.5), px1 = (int)ceil(u + half - 0.5) - 1, py0 = (int)ceil(v - half - 0.5), py1 = (int)ceil(v + half - 0.5) - 1;
        px0 = px0 < 0 ? 0 : px0; py0 = py0 < 0 ? 0 : py0; px1 = px1 > W - 1 ? W - 1 : px1; py1 = py1 > H - 1 ? H - 1 : py1;
        F::position = V::position; F::normRad = V::normRad; F::colTime = V::colTime;
        for (int py = py0; py <= py1; ++py)
            for (int px = px0; px <= px1; ++px) {
                gl_FragCoord = vec4((float)px + 0.5f, (float)py + 0.5f, 0.f, 1.f);
                g_discarded = false;
                F::main();
                if (g_discarded) continue;
                const int p = py * W + px;
                if (!(gl_FragDepth >= 0.f && gl_FragDepth <= 1.f)) continue;   /* outside the depth range (NaN included) */
                if (!(gl_FragDepth < zbuf[p])) continue;
                zbuf[p] = gl_FragDepth;
                const float c[4] = {F::image.x, F::image.y, F::image.z, F::image.w};
                for (int q = 0; q < 4; ++q) { float t = c[q] < 0.f ? 0.f : (c[q] > 1.f ? 1.f : c[q]); image[(size_t)p * 4 + q] = (uint8_t)(int)rintf(t * 255.0f); }
                memcpy(vertexConf + (size_t)p * 4, &F::vertexConf.x, 16); memcpy(normalRadius + (size_t)p * 4, &F::normalRadius.x, 16);
                timeOut[p] = (uint16_t)F::time;
            }
    }
}

/* GlobalProjection::project + downloadDirect (Core/Model/GlobalProjection.cpp:43-114): splat_models.vert + combo_splat_models.frag for
 * every model into ONE framebuffer (fixed confidence threshold 12), ids cleared to 0.  Models are drawn in list order, which breaks
 * exact depth ties (depth test LESS). */
void mfglsl_global_projection(const float* const* poses16, const float* const* surfels, const int* counts, const int* ids, int n_models, int time,
                              int maxTime, int timeDelta, float maxDepth, int W, int H, float fx, float fy, float cx, float cy, uint8_t* idOut) {
    namespace V = sh_splat_models_vert;
    namespace F = sh_combo_splat_models_frag;
    const int P = W * H;
    std::vector<float> zbuf((size_t)P, 1.0f);
    memset(idOut, 0, (size_t)P);
    V::cam = vec4(cx, cy, fx, fy); V::cols = (float)W; V::rows = (float)H; V::maxDepth = maxDepth; V::confThreshold = 12.0f;
    V::time = time; V::maxTime = maxTime; V::timeDelta = timeDelta;
    F::cam = vec4(cx, cy, fx, fy); F::maxDepth = maxDepth;
    for (int m = 0; m < n_models; ++m) {
        float tinv[16];
        rigid_inverse(poses16[m], tinv);
        V::t_inv = mat4_from(tinv); V::modelID = ids[m]; F::modelID = ids[m];
        for (int i = 0; i < counts[m]; ++i) {
            const float* s = surfels[m] + (size_t)i * 12;
            V::vPosition = vec4(s[0], s[1], s[2], s[3]); V::vColor = vec4(s[4], s[5], s[6], s[7]); V::vNormRad = vec4(s[8], s[9], s[10], s[11]);
            V::main();
            double u, v;
            if (!window_pos(gl_Position, W, H, u, v)) continue;
            float size = gl_PointSize;
            if (!(size > 0.f)) continue;
            size = size < 1.0f ? 1.0f : (size > 64.0f ? 64.0f : size);
            const double half = size * 0.5;
            int px0 = (int)ceil(u - half - 0.5), px1 = (int)ceil(u + half - 0.5) - 1, py0 = (int)ceil(v - half - 0.5), py1 = (int)ceil(v + half - 0.5) - 1;
            px0 = px0 < 0 ? 0 : px0; py0 = py0 < 0 ? 0 : py0; px1 = px1 > W - 1 ? W - 1 : px1; py1 = py1 > H - 1 ? H - 1 : py1;
            F::position = V::position; F::normRad = V::normRad;
            for (int py = py0; py <= py1; ++py)
                for (int px = px0; px <= px1; ++px) {
                    gl_FragCoord = vec4((float)px + 0.5f, (float)py + 0.5f, 0.f, 1.f);
                    g_discarded = false;
                    F::main();
                    if (g_discarded) continue;
                    const int p = py * W + px;
                    if (!(gl_FragDepth >= 0.f && gl_FragDepth <= 1.f)) continue;
                    if (!(gl_FragDepth < zbuf[p])) continue;
                    zbuf[p] = gl_FragDepth;
                    idOut[p] = (uint8_t)F::id;
                }
        }
    }
}

/* FillIn::vertex / normal / image (Core/Shaders/FillIn.cpp) with fill_vertex / fill_normal / fill_rgb.frag */
void mfglsl_fill_in(const uint8_t* predImage /*RGBA8*/, const float* predVertex, const float* predNormal, const uint8_t* rawRgb, const float* rawDepth,
                    int passthrough, int W, int H, float fx, float fy, float cx, float cy, uint8_t* fillImage, float* fillVertex, float* fillNormal) {
    const int P = W * H;
    std::vector<float> pimg((size_t)P * 4);
    for (int i = 0; i < P * 4; ++i) pimg[i] = (float)predImage[i] / 255.0f;
    const std::vector<float> raw = rgba_from_rgb8(rawRgb, P);
    namespace FV = sh_fill_vertex_frag; namespace FN = sh_fill_normal_frag; namespace FC = sh_fill_rgb_frag;
    FV::eSampler = sampler2D{predVertex, W, H, 4}; FV::rSampler = sampler2D{rawDepth, W, H, 1}; FV::cam = vec4(cx, cy, 1.0f / fx, 1.0f / fy);
    FV::cols = (float)W; FV::rows = (float)H; FV::passthrough = passthrough;
    FN::eSampler = sampler2D{predNormal, W, H, 4}; FN::rSampler = sampler2D{rawDepth, W, H, 1}; FN::cam = vec4(cx, cy, 1.0f / fx, 1.0f / fy);
    FN::cols = (float)W; FN::rows = (float)H; FN::passthrough = passthrough;
    FC::eSampler = sampler2D{pimg.data(), W, H, 4}; FC::rSampler = sampler2D{raw.data(), W, H, 4}; FC::passthrough = passthrough;
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const vec2 tc(((float)x + 0.5f) / (float)W, ((float)y + 0.5f) / (float)H);
            const int p = y * W + x;
            FV::texcoord = tc; FV::main(); memcpy(fillVertex + (size_t)p * 4, &FV::FragColor.x, 16);
            FN::texcoord = tc; FN::main(); memcpy(fillNormal + (size_t)p * 4, &FN::FragColor.x, 16);
            FC::texcoord = tc; FC::main();
            const float c[4] = {FC::FragColor.x, FC::FragColor.y, FC::FragColor.z, FC::FragColor.w};
            for (int q = 0; q < 4; ++q) fillImage[(size_t)p * 4 + q] = (uint8_t)(int)rintf(c[q] * 255.0f);
        }
}

void mfglsl_set_exp_libm(int on) { g_exp_libm = on; }
}  // extern "C"

namespace mfgl { int exp_uses_libm() { return g_exp_libm; } }

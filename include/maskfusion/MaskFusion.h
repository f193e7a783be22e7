// MaskFusion.h -- header-only C++ facade with the reference's class / method names over the C ABI
// (include/maskfusion_amd.h).  Mirrors Core/MaskFusion.h:45-307 and Core/Model/Model.h:108-268 of
// martinruenz/maskfusion for the hot path; Eigen / OpenCV types are replaced by plain arrays so that this header has no
// dependencies (a caller that has Eigen passes `pose.data()`; `Eigen::Matrix4f` is column-major like `float[16]` here).
#pragma once

#include <array>
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../maskfusion_amd.h"

namespace maskfusion {

using Matrix4f = std::array<float, 16>;  // column-major, == Eigen::Matrix4f::data()

// Core/FrameData.h:25-48
struct FrameData {
    int64_t timestamp = 0;
    int64_t index = 0;
    const uint8_t* rgb = nullptr;    // H*W*3 (cv::Mat CV_8UC3 .data)
    const float* depth = nullptr;    // H*W metres (CV_32FC1)
    const uint8_t* mask = nullptr;   // H*W model ids or nullptr (CV_8UC1)
    std::vector<int32_t> classIDs;
};
using FrameDataPointer = std::shared_ptr<FrameData>;

class MaskFusion;

// Core/Model/Model.h
class Model {
public:
    struct SurfelMap {  // Model.h:193-204 (3 x Vector4f per surfel)
        std::unique_ptr<std::vector<float>> data;
        unsigned numPoints = 0;
        unsigned numValid = 0;
        void countValid(const float& confThres) {
            numValid = 0;
            for (unsigned i = 0; i < numPoints; i++)
                if ((*data)[i * 12 + 3] > confThres) numValid++;
        }
    };
    struct PoseLogItem {  // Model.h:257-260 (p = tx ty tz qx qy qz qw)
        int64_t ts;
        float p[7];
    };
    unsigned getID() const { return id_; }
    inline Matrix4f getPose() const;
    inline unsigned lastCount() const;
    inline SurfelMap downloadMap() const;
    inline std::vector<PoseLogItem> getPoseLog() const;  // Model.h:262

private:
    friend class MaskFusion;
    Model(mf_ctx* c, int id) : ctx_(c), id_(id) {}
    mf_ctx* ctx_;
    int id_;
};

class MaskFusion {
public:
    // Core/MaskFusion.h:47-53 (arguments that are dead on the open-loop hot path are accepted and ignored); the camera
    // replaces the Resolution / Intrinsics singletons the reference reads (GUI/MainController.cpp:117-128).
    MaskFusion(int width, int height, float fx, float fy, float cx, float cy, int timeDelta = 200, int /*countThresh*/ = 35000,
               float /*errThresh*/ = 5e-05f, float /*covThresh*/ = 1e-05f, bool /*closeLoops*/ = false, bool /*iclnuim*/ = false,
               bool /*reloc*/ = false, float /*photoThresh*/ = 115, float initConfidenceGlobal = 4, float initConfidenceObject = 2,
               float depthCut = 3, float icpThresh = 10, bool fastOdom = false, float /*fernThresh*/ = 0.3095f, bool so3 = true,
               bool /*frameToFrameRGB*/ = false, unsigned modelSpawnOffset = 20, const std::string& exportDirectory = "",
               int device = 0)
        : exportDir_(exportDirectory) {
        mf_config cfg;
        mf_default_config(&cfg, width, height, fx, fy, cx, cy);
        cfg.time_delta = timeDelta;
        cfg.conf_global = initConfidenceGlobal;
        cfg.conf_object = initConfidenceObject;
        cfg.depth_cutoff = depthCut;
        cfg.icp_weight = icpThresh;
        cfg.fast_odom = fastOdom;
        cfg.so3 = so3;
        cfg.model_spawn_offset = (int32_t)modelSpawnOffset;
        cfg.device = device;
        const int rc = mf_create(&cfg, &ctx_);
        if (rc != MF_OK) throw std::runtime_error("mf_create failed with code " + std::to_string(rc));
    }
    ~MaskFusion() { mf_destroy(ctx_); }
    MaskFusion(const MaskFusion&) = delete;
    MaskFusion& operator=(const MaskFusion&) = delete;

    // Core/MaskFusion.h:69-70.  Returns false like the reference (MaskFusion.cpp:606); errors throw.
    bool processFrame(FrameDataPointer frame, const Matrix4f* inPose = nullptr, const float weightMultiplier = 1.f,
                      const bool bootstrap = false) {
        check(mf_process_frame(ctx_, frame->rgb, frame->depth, frame->mask, frame->classIDs.data(), (int32_t)frame->classIDs.size(),
                               frame->timestamp, inPose ? inPose->data() : nullptr, weightMultiplier, bootstrap));
        return false;
    }
    void predict() { check(mf_predict(ctx_)); }  // MaskFusion.h:76
    void preallocateModels(unsigned count) { check(mf_preallocate_models(ctx_, count)); }  // MaskFusion.h:57
    void savePly() { check(mf_save_ply(ctx_, exportDir_.c_str())); }          // MaskFusion.h:282
    void exportPoses() { check(mf_export_poses(ctx_, exportDir_.c_str())); }  // MaskFusion.h:284

    Model getBackgroundModel() { return Model(ctx_, 0); }  // MaskFusion.h:88
    std::vector<Model> getModels() {                        // MaskFusion.h:90
        int32_t n = 0;
        check(mf_num_models(ctx_, &n));
        std::vector<Model> v;
        for (int i = 0; i < n; ++i) v.push_back(Model(ctx_, i));
        return v;
    }
    Matrix4f getCurrPose() { return getBackgroundModel().getPose(); }  // MaskFusion.h:218
    int getTick() { int32_t t = 0; check(mf_get_tick(ctx_, &t)); return t; }  // MaskFusion.h:194
    void setTick(const int& val) { check(mf_set_tick(ctx_, val)); }            // MaskFusion.h:206

    // per-frame setters, MaskFusion.h:132-182
    void setDepthCutoff(const float& v) { set("depthCutoff", v); }
    void setIcpWeight(const float& v) { set("icpWeight", v); }
    void setConfidenceThreshold(const float& v) { set("confidenceThreshold", v); }
    void setOutlierCoefficient(const float& v) { set("outlierCoefficient", v); }
    void setFastOdom(const bool& v) { set("fastOdom", v); }
    void setSo3(const bool& v) { set("so3", v); }
    void setRgbOnly(const bool& v) { set("rgbOnly", v); }
    void setTrackAllModels(bool v) { set("trackAllModels", v); }
    void setPyramid(const bool& v) { set("pyramid", v); }
    void setEnableMultipleModels(bool v) { set("enableMultipleModels", v); }

    mf_ctx* handle() { return ctx_; }

private:
    void set(const char* k, double v) { check(mf_set_param(ctx_, k, v)); }
    void check(int rc) {
        if (rc != MF_OK) throw std::runtime_error(std::string("maskfusion_amd: ") + mf_last_error(ctx_));
    }
    mf_ctx* ctx_ = nullptr;
    std::string exportDir_;
};

inline std::vector<Model::PoseLogItem> Model::getPoseLog() const {
    uint32_t n = 0;
    if (mf_get_pose_log(ctx_, id_, nullptr, nullptr, 0, &n) != MF_OK) throw std::runtime_error(mf_last_error(ctx_));
    std::vector<int64_t> ts(n);
    std::vector<float> p((size_t)n * 7);
    if (n && mf_get_pose_log(ctx_, id_, ts.data(), p.data(), n, &n) != MF_OK) throw std::runtime_error(mf_last_error(ctx_));
    std::vector<PoseLogItem> out(n);
    for (uint32_t i = 0; i < n; ++i) {
        out[i].ts = ts[i];
        for (int k = 0; k < 7; ++k) out[i].p[k] = p[(size_t)i * 7 + k];
    }
    return out;
}

inline Matrix4f Model::getPose() const {
    Matrix4f p;
    if (mf_get_pose(ctx_, id_, p.data()) != MF_OK) throw std::runtime_error(mf_last_error(ctx_));
    return p;
}
inline unsigned Model::lastCount() const {
    uint32_t n = 0;
    if (mf_get_surfel_count(ctx_, id_, &n) != MF_OK) throw std::runtime_error(mf_last_error(ctx_));
    return n;
}
inline Model::SurfelMap Model::downloadMap() const {
    SurfelMap m;
    m.numPoints = lastCount();
    m.data = std::make_unique<std::vector<float>>((size_t)m.numPoints * 12);
    uint32_t n = 0;
    if (mf_download_map(ctx_, id_, m.data->data(), m.numPoints, &n) != MF_OK) throw std::runtime_error(mf_last_error(ctx_));
    return m;
}

}  // namespace maskfusion

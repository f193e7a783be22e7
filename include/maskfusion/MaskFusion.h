// MaskFusion.h -- header-only C++14 facade with the reference's class / method names over the C ABI
// (include/maskfusion_amd.h).  Mirrors Core/MaskFusion.h:45-307, Core/Model/Model.h:108-268, Core/FrameData.h:25-48,
// Core/Segmentation/SegmentationResult.h:32-73, Core/Utils/Resolution.h / Intrinsics.h and Core/Callbacks.h:68 of
// martinruenz/maskfusion for the hot path.  Eigen / OpenCV / OpenGL types are replaced by plain arrays so that this header
// has no dependencies: Matrix4f is column-major like Eigen::Matrix4f::data(); cv::Mat members of FrameData become plain
// pointers; GPUTexture* / FeedbackBuffer arguments of the Model calls are accepted and ignored (the frame staged by
// MaskFusion::processFrame / stageFrame is what those textures held upstream).
//
// Construction follows the reference: set the Resolution / Intrinsics singletons first (GUI/MainController.cpp:117-128), then
// call the constructor with the reference's own argument list, in the reference's order (Core/MaskFusion.h:47-53).  The HIP
// device ordinal (upstream: the CUDA device of the GL context) is chosen with maskfusion::Device::set(ordinal) beforehand.
#pragma once

#include <algorithm>
#include <array>
#include <cstdint>
#include <functional>
#include <list>
#include <memory>
#include <queue>
#include <set>
#include <stdexcept>
#include <string>
#include <vector>

#include "../maskfusion_amd.h"

namespace maskfusion {

using Matrix4f = std::array<float, 16>;  // column-major, == Eigen::Matrix4f::data()

// Core/Utils/Resolution.h:24-71
class Resolution {
public:
    static const Resolution& getInstance() { return instance(); }
    static void setResolution(int width, int height) {
        instance().w_ = width;
        instance().h_ = height;
    }
    const int& width() const { return checked(w_); }
    const int& height() const { return checked(h_); }
    const int& cols() const { return checked(w_); }
    const int& rows() const { return checked(h_); }
    int numPixels() const { return checked(w_) * h_; }

private:
    Resolution() {}
    static Resolution& instance() {
        static Resolution r;
        return r;
    }
    const int& checked(const int& v) const {
        if (w_ <= 0 || h_ <= 0) throw std::logic_error("You haven't initialised the Resolution class!");
        return v;
    }
    int w_ = 0, h_ = 0;
};

// Core/Utils/Intrinsics.h:24-63 (the setter keeps upstream's spelling)
class Intrinsics {
public:
    static const Intrinsics& getInstance() { return instance(); }
    static void setIntrinics(float fx = 0, float fy = 0, float cx = 0, float cy = 0) {
        Intrinsics& i = instance();
        i.fx_ = fx, i.fy_ = fy, i.cx_ = cx, i.cy_ = cy;
        i.checked(i.fx_);
    }
    const float& fx() const { return checked(fx_); }
    const float& fy() const { return checked(fy_); }
    const float& cx() const { return checked(cx_); }
    const float& cy() const { return checked(cy_); }

private:
    Intrinsics() {}
    static Intrinsics& instance() {
        static Intrinsics i;
        return i;
    }
    const float& checked(const float& v) const {
        if (fx_ == 0 || fy_ == 0) throw std::logic_error("You haven't initialised the Intrinsics class!");
        return v;
    }
    float fx_ = 0, fy_ = 0, cx_ = 0, cy_ = 0;
};

// The GPU a MaskFusion constructed next will live on, and its surfel budgets (upstream: compile-time MASKFUSION_GPU_SLAM,
// MASKFUSION_NUM_GSURFELS / NUM_OSURFELS, Core/CMakeLists.txt:27-28).  0 surfels = the library's defaults.
struct Device {
    static void set(int ordinal) { state().ordinal = ordinal; }
    static void setSurfelBudget(int global, int object) { state().gsurfels = global, state().osurfels = object; }
    static int get() { return state().ordinal; }
    struct State { int ordinal = 0, gsurfels = 0, osurfels = 0; };
    static State& state() {
        static State s;
        return s;
    }
};

// Core/FrameData.h:25-48 (cv::Mat -> pointer; the owner keeps the pixels alive while the frame is queued)
struct FrameData {
    int64_t timestamp = 0;
    int64_t index = 0;
    const uint8_t* mask = nullptr;   // H*W mask ids or nullptr (CV_8UC1)
    const uint8_t* rgb = nullptr;    // H*W*3 (CV_8UC3 .data)
    const float* depth = nullptr;    // H*W metres (CV_32FC1)
    std::vector<int> classIDs;       // classIDs[mask value] = class
};
using FrameDataPointer = std::shared_ptr<FrameData>;

// Core/Segmentation/SegmentationResult.h:32-73 (the members the hot path reads)
struct SegmentationResult {
    std::vector<uint8_t> fullSegmentation;  // H*W model id per pixel, 255 = ignored
    int width = 0, height = 0;
    bool hasNewLabel = false;
    int newClassID = -1;
};

struct GPUTexture;      // upstream GL texture wrappers: accepted by the Model calls for signature compatibility, never read
struct FeedbackBuffer;
struct CameraModel;

class MaskFusion;

// Core/Model/Model.h
class Model {
public:
    enum class MatchingType { Drost };  // Model.h:93-95 (re-detection is off the hot path)
    struct SurfelMap {  // Model.h:193-204 (3 x Vector4f per surfel)
        std::unique_ptr<std::vector<float>> data;
        unsigned numPoints = 0;
        unsigned numValid = 0;
        void countValid(const float& confThres) {
            numValid = 0;
            for (unsigned i = 0; i < numPoints; i++)
                if ((*data)[(size_t)i * 12 + 3] > confThres) numValid++;
        }
    };
    struct PoseLogItem {  // Model.h:257-260 (p = tx ty tz qx qy qz qw)
        int64_t ts;
        float p[7];
    };

    unsigned int getID() const { return (unsigned)id_; }                      // Model.h:241
    int getClassID() const { return live() ? info().class_id : classID_; }   // Model.h:242
    float getConfidenceThreshold() const { return info().confidence_threshold; }  // Model.h:180
    unsigned getAge() const { return info().age; }                            // Model.h:249
    bool isNonstatic() const { return !info().is_static; }                   // Model.h:266
    void makeNonStatic() { chk(mf_make_nonstatic(ctx_, idx())); }            // Model.h:265
    // Model.h:264.  globalPose must be the background's current pose (the only value upstream passes, MaskFusion.cpp:330).
    void makeStatic(const Matrix4f& /*globalPose*/) { chk(mf_make_static(ctx_, idx())); }
    void updateStaticPose(const Matrix4f& /*globalPose*/) { chk(mf_model_update_static_pose(ctx_, idx())); }  // Model.h:263
    unsigned int lastCount() const {                                          // Model.h:118
        uint32_t n = 0;
        chk(mf_get_surfel_count(ctx_, idx(), &n));
        return n;
    }
    Matrix4f getPose() const {                                                // Model.h:233
        Matrix4f p;
        chk(mf_get_pose(ctx_, idx(), p.data()));
        return p;
    }
    void overridePose(const Matrix4f& p) { chk(mf_model_override_pose(ctx_, idx(), p.data())); }  // Model.h:235-238

    // Model.h:126 -- from the frame staged by MaskFusion::stageFrame
    void initialise(const FeedbackBuffer* /*rawFeedback*/ = nullptr, const FeedbackBuffer* /*filteredFeedback*/ = nullptr) {
        chk(mf_model_initialise(ctx_, idx()));
    }
    // Model.h:135-136.  Returns the pose after tracking, like the reference.
    Matrix4f performTracking(bool frameToFrameRGB, bool rgbOnly, float icpWeight, bool pyramid, bool fastOdom, bool so3,
                             float maxDepthProcessed, GPUTexture* /*rgb*/, int64_t logTimestamp, bool tryFillIn = false) {
        chk(mf_model_perform_tracking(ctx_, idx(), frameToFrameRGB, rgbOnly, icpWeight, pyramid, fastOdom, so3, maxDepthProcessed,
                                      logTimestamp, tryFillIn));
        return getPose();
    }
    float computeFusionWeight(float weightMultiplier) const {  // Model.h:139
        float w = 0;
        chk(mf_model_fusion_weight(ctx_, idx(), weightMultiplier, &w));
        return w;
    }
    // Model.h:142-143
    void fuse(const int& time, GPUTexture* /*rgb*/, GPUTexture* /*mask*/, GPUTexture* /*depthRaw*/, GPUTexture* /*depthFiltered*/,
              const float depthCutoff, const float weightMultiplier) {
        chk(mf_model_fuse(ctx_, idx(), time, depthCutoff, weightMultiplier));
    }
    // Model.h:146-147 (no deformation graph, isFern must be false: loop closure is dead code upstream)
    void clean(const int& time, std::vector<float>& /*graph*/, const int timeDelta, const float depthCutoff, const bool /*isFern*/,
               GPUTexture* /*depthFiltered*/, GPUTexture* /*mask*/) {
        chk(mf_model_clean(ctx_, idx(), time, timeDelta, depthCutoff));
    }
    // Model.h:158 (predictionType: ModelProjection::ACTIVE, the only one upstream uses)
    void combinedPredict(float depthCutoff, int time, int maxTime, int timeDelta, int /*predictionType*/ = 0) {
        chk(mf_model_combined_predict(ctx_, idx(), depthCutoff, time, maxTime, timeDelta));
    }
    void predictIndices(int time, float depthCutoff, int timeDelta) {  // Model.h:162
        chk(mf_model_predict_indices(ctx_, idx(), time, depthCutoff, timeDelta));
    }
    SurfelMap downloadMap(int /*buffer*/ = -1) const {  // Model.h:206
        SurfelMap m;
        m.numPoints = lastCount();
        m.data = std::make_unique<std::vector<float>>((size_t)m.numPoints * 12);
        uint32_t n = 0;
        chk(mf_download_map(ctx_, idx(), m.data->data(), m.numPoints, &n));
        return m;
    }
    std::vector<PoseLogItem> getPoseLog() const {  // Model.h:262
        uint32_t n = 0;
        chk(mf_get_pose_log(ctx_, idx(), nullptr, nullptr, 0, &n));
        std::vector<int64_t> ts(n);
        std::vector<float> p((size_t)n * 7);
        if (n) chk(mf_get_pose_log(ctx_, idx(), ts.data(), p.data(), n, &n));
        std::vector<PoseLogItem> out(n);
        for (uint32_t i = 0; i < n; ++i) {
            out[i].ts = ts[i];
            for (int k = 0; k < 7; ++k) out[i].p[k] = p[(size_t)i * 7 + k];
        }
        return out;
    }
    // {lastICPError, lastICPCount} of the last tracking step (Model::getFrameOdometry().lastICPError / lastICPCount)
    void getICPStats(float& lastICPError, float& lastICPCount) const {
        chk(mf_get_icp_stats(ctx_, idx(), &lastICPError, &lastICPCount));
    }

private:
    friend class MaskFusion;
    Model(mf_ctx* c, int id, int index, int classID) : ctx_(c), id_(id), index_(index), classID_(classID) {}
    bool live() const { return index_ >= 0; }
    int idx() const {
        if (index_ < 0) throw std::runtime_error("maskfusion_amd: model " + std::to_string(id_) + " is inactive");
        return index_;
    }
    mf_model_info_t info() const {
        mf_model_info_t i;
        chk(mf_model_info(ctx_, idx(), &i));
        return i;
    }
    void chk(int rc) const {
        if (rc != MF_OK) throw std::runtime_error(std::string("maskfusion_amd: ") + mf_last_error(ctx_));
    }
    mf_ctx* ctx_;
    int id_;       // Model::getID(): stable for the life of the model
    int index_;    // position in the library's model list (what the ABI takes); refreshed after every frame; -1 once inactive
    int classID_;  // last known class (readable after the model went inactive)
};

using ModelPointer = std::shared_ptr<Model>;
using ModelList = std::list<ModelPointer>;
using ModelListener = std::function<void(ModelPointer)>;  // Core/Callbacks.h:68

namespace Segmentation {
enum class Method { MASK_FUSION, CO_FUSION, PRECOMPUTED };  // only MASK_FUSION / PRECOMPUTED masks are on the hot path
}

class MaskFusion {
public:
    // Core/MaskFusion.h:47-53, argument for argument.  Arguments of the loop-closure / fern / relocalisation machinery that is
    // dead code upstream (GUI/MainController.cpp:246,399) are accepted and ignored.
    MaskFusion(int timeDelta = 200, int /*countThresh*/ = 35000, float /*errThresh*/ = 5e-05f, float /*covThresh*/ = 1e-05f,
               bool /*closeLoops*/ = true, bool /*iclnuim*/ = false, bool /*reloc*/ = false, float /*photoThresh*/ = 115,
               float initConfidenceGlobal = 4, float initConfidenceObject = 2, float depthCut = 3, float icpThresh = 10,
               bool fastOdom = false, float /*fernThresh*/ = 0.3095f, bool so3 = true, bool frameToFrameRGB = false,
               unsigned modelSpawnOffset = 20, Model::MatchingType /*matchingType*/ = Model::MatchingType::Drost,
               Segmentation::Method segmentationMethod = Segmentation::Method::MASK_FUSION, const std::string& exportDirectory = "",
               bool exportSegmentationResults = false, bool usePrecomputedMasksOnly = false, unsigned frameQueueSize = 0)
        : exportDir_(exportDirectory),
          exportSegmentation_(exportSegmentationResults),
          // MaskFusion.cpp:37
          queueLength_((usePrecomputedMasksOnly || segmentationMethod != Segmentation::Method::MASK_FUSION) ? 0 : frameQueueSize) {
        if (segmentationMethod == Segmentation::Method::CO_FUSION)
            throw std::invalid_argument("maskfusion_amd: the Co-Fusion segmentation method is out of scope (DESIGN.md section 1)");
        const Resolution& res = Resolution::getInstance();
        const Intrinsics& in = Intrinsics::getInstance();
        mf_config cfg;
        mf_default_config(&cfg, res.width(), res.height(), in.fx(), in.fy(), in.cx(), in.cy());
        cfg.time_delta = timeDelta;
        cfg.conf_global = initConfidenceGlobal;
        cfg.conf_object = initConfidenceObject;
        cfg.depth_cutoff = depthCut;
        cfg.icp_weight = icpThresh;
        cfg.fast_odom = fastOdom;
        cfg.so3 = so3;
        cfg.model_spawn_offset = (int32_t)modelSpawnOffset;
        cfg.device = Device::get();
        if (Device::state().gsurfels > 0) cfg.num_gsurfels = Device::state().gsurfels;
        if (Device::state().osurfels > 0) cfg.num_osurfels = Device::state().osurfels;
        cfg.pose_log_capacity = 1000;  // MaskFusion.cpp:65,677: enablePoseLogging, poseLog.reserve(1000)
        width_ = cfg.width, height_ = cfg.height;
        const int rc = mf_create(&cfg, &ctx_);
        if (rc != MF_OK) throw std::runtime_error("maskfusion_amd: mf_create failed with code " + std::to_string(rc));
        if (frameToFrameRGB) mf_set_param(ctx_, "frameToFrameRGB", 1.0);   // MaskFusion.cpp:66
        refreshModels();
    }
    virtual ~MaskFusion() { mf_destroy(ctx_); }
    MaskFusion(const MaskFusion&) = delete;
    MaskFusion& operator=(const MaskFusion&) = delete;

    void preallocateModels(unsigned count) { check(mf_preallocate_models(ctx_, count)); }  // MaskFusion.h:57

    // MaskFusion.h:59 (MfSegmentation.cpp:83-538) on the frame staged last (processFrame stages its frame; see stageFrame)
    SegmentationResult performSegmentation(FrameDataPointer frame) {
        SegmentationResult r;
        std::vector<int32_t> cls(frame->classIDs.begin(), frame->classIDs.end());
        int32_t hasNew = 0, newClass = -1;
        check(mf_perform_segmentation(ctx_, frame->mask, cls.data(), (int32_t)cls.size(), nullptr, nullptr, 0, 0, 1, &hasNew, &newClass));
        r.width = width_, r.height = height_;
        r.fullSegmentation.resize((size_t)width_ * height_);
        check(mf_download_segmentation(ctx_, r.fullSegmentation.data()));
        r.hasNewLabel = hasNew != 0;
        r.newClassID = newClass;
        return r;
    }

    // Core/MaskFusion.h:69-70.  Returns false like the reference (MaskFusion.cpp:606); errors throw.  Frames pass through the
    // frame queue first (MaskFusion.cpp:206-209): with frameQueueSize = q the frame processed is the one handed in q-1 calls ago
    // (upstream the Mask R-CNN thread fills in frame->mask meanwhile).
    bool processFrame(FrameDataPointer frame, const Matrix4f* inPose = nullptr, const float weightMultiplier = 1.f,
                      const bool bootstrap = false) {
        if (!frame || !frame->rgb || !frame->depth || frame->timestamp < 0)
            throw std::invalid_argument("maskfusion_amd: processFrame needs rgb (8UC3), depth (32FC1) and a timestamp >= 0");
        frameQueue_.push(frame);
        if (frameQueue_.size() < queueLength_) return false;
        frame = frameQueue_.front();
        frameQueue_.pop();
        std::vector<int32_t> cls(frame->classIDs.begin(), frame->classIDs.end());
        const int tick = getTick();
        check(mf_process_frame(ctx_, frame->rgb, frame->depth, frame->mask, cls.data(), (int32_t)cls.size(), frame->timestamp,
                               inPose ? inPose->data() : nullptr, weightMultiplier, bootstrap));
        if (exportSegmentation_ && tick > 1 && getParam("enableMultipleModels") != 0)   // MaskFusion.cpp:299-303 (not on the init frame)
            check(mf_export_segmentation_png(ctx_, (exportDir_ + "Segmentation" + std::to_string(tick) + ".png").c_str()));
        refreshModels();  // spawn / inactivate decisions of this frame -> model list, listeners (MaskFusion.cpp:694,711)
        return false;
    }
    // Everything of processFrame that touches no model (upload, filterDepth, Model::generateCUDATextures, Model.h:128) for callers
    // that drive the Model calls themselves; mf_end_frame is processFrame's tail (tick++, pose log, age)
    void stageFrame(FrameDataPointer frame) { check(mf_stage_frame(ctx_, frame->rgb, frame->depth, frame->mask)); }
    void endFrame(int64_t timestamp) { check(mf_end_frame(ctx_, timestamp)); }

    void predict() { check(mf_predict(ctx_)); }  // MaskFusion.h:76
    void savePly() { check(mf_save_ply(ctx_, exportDir_.c_str())); }          // MaskFusion.h:282
    void exportPoses() { check(mf_export_poses(ctx_, exportDir_.c_str())); }  // MaskFusion.h:284

    ModelPointer getBackgroundModel() { return models_.front(); }  // MaskFusion.h:88
    ModelList& getModels() { return models_; }                     // MaskFusion.h:90
    Matrix4f getCurrPose() { return getBackgroundModel()->getPose(); }  // MaskFusion.h:218
    int getTick() { int32_t t = 0; check(mf_get_tick(ctx_, &t)); return t; }  // MaskFusion.h:194
    void setTick(const int& val) { check(mf_set_tick(ctx_, val)); }            // MaskFusion.h:206
    int getTimeDelta() { return (int)getParam("timeDelta"); }                  // MaskFusion.h:200
    float getMaxDepthProcessed() { return (float)getParam("maxDepthProcessed"); }  // MaskFusion.h:212
    float getConfidenceThreshold() { return (float)getParam("confidenceThreshold"); }  // MaskFusion.h:126
    bool getLost() { return false; }  // MaskFusion.h:188 (relocalisation is dead code upstream)
    int getDeforms() { return 0; }
    int getFernDeforms() { return 0; }

    // per-frame setters, MaskFusion.h:132-182
    void setRgbOnly(const bool& v) { set("rgbOnly", v); }
    void setIcpWeight(const float& v) { set("icpWeight", v); }
    void setOutlierCoefficient(const float& v) { set("outlierCoefficient", v); }
    void setPyramid(const bool& v) { set("pyramid", v); }
    void setFastOdom(const bool& v) { set("fastOdom", v); }
    void setSo3(const bool& v) { set("so3", v); }
    void setFrameToFrameRGB(const bool& v) { set("frameToFrameRGB", v); }   // Core/MaskFusion.cpp:910 ("-ftf", GUI/MainController.cpp:252,539)
    void setConfidenceThreshold(const float& v) { set("confidenceThreshold", v); }
    void setFernThresh(const float&) {}  // ferns: dead code upstream
    void setDepthCutoff(const float& v) { set("depthCutoff", v); }
    // MfSegmentation tunables, MaskFusion.h:234-246 (the bilateral-prefilter ones belong to a path MfSegmentation never runs)
    void setMfBilatSigmaDepth(float) {}
    void setMfBilatSigmaColor(float) {}
    void setMfBilatSigmaLocation(float) {}
    void setMfBilatRadius(int) {}
    void setMfMorphEdgeRadius(int v) { set("mfMorphEdgeRadius", v); }
    void setMfMorphEdgeIterations(int v) { set("mfMorphEdgeIterations", v); }
    void setMfMorphMaskRadius(int v) { set("mfMorphMaskRadius", v); }
    void setMfMorphMaskIterations(int v) { set("mfMorphMaskIterations", v); }
    void setMfThreshold(float v) { set("mfThreshold", v); }
    void setMfWeightDistance(float v) { set("mfWeightDistance", v); }
    void setMfWeightConvexity(float v) { set("mfWeightConvexity", v); }
    void setMfNonstaticThreshold(float) {}  // stored and never read upstream (MfSegmentation.h:60)
    void setTrackableClassIds(const std::set<int>& ids) {  // MaskFusion.h:246
        std::vector<int32_t> v(ids.begin(), ids.end());
        check(mf_set_trackable_class_ids(ctx_, v.data(), (int32_t)v.size()));
    }
    void setModelSpawnOffset(const unsigned& v) { set("modelSpawnOffset", v); }
    void setModelDeactivateCount(const unsigned&) {}  // MaskFusion.h:249: the counter is never compared upstream
    // Co-Fusion CRF parameters (MaskFusion.h:250-258): that segmentation method is out of scope
    void setCfPairwiseSigmaRGB(const float&) {}
    void setCfPairwiseSigmaPosition(const float&) {}
    void setCfPairwiseSigmaDepth(const float&) {}
    void setCfPairwiseWeightAppearance(const float&) {}
    void setCfPairwiseWeightSmoothness(const float&) {}
    void setCfThresholdNew(const float&) {}
    void setCfUnaryWeightError(const float&) {}
    void setCfIteration(const unsigned&) {}
    void setCfUnaryKError(const float&) {}
    void setNewModelMinRelativeSize(const float& v) { set("newModelMinRelativeSize", v); }
    void setNewModelMaxRelativeSize(const float& v) { set("newModelMaxRelativeSize", v); }
    void setEnableMultipleModels(bool v) { set("enableMultipleModels", v); }
    void setTrackAllModels(bool v) { set("trackAllModels", v); }
    void setEnableSmartModelDelete(bool) {}  // MaskFusion.h:263: read by no code path upstream

    // Listeners, MaskFusion.h:303-306.  Called from processFrame (the caller's thread) after the frame that spawned / dropped the
    // model, in model-list order; the inactive model's getID() / getClassID() stay readable, its other calls throw.
    void addNewModelListener(const ModelListener& listener) { newModelListeners_.push_back(listener); }
    void addInactiveModelListener(const ModelListener& listener) { inactiveModelListeners_.push_back(listener); }

    double getParam(const char* key) {
        double v = 0;
        check(mf_get_param(ctx_, key, &v));
        return v;
    }
    mf_ctx* handle() { return ctx_; }

private:
    void set(const char* k, double v) { check(mf_set_param(ctx_, k, v)); }
    void check(int rc) {
        if (rc != MF_OK) throw std::runtime_error(std::string("maskfusion_amd: ") + mf_last_error(ctx_));
    }
    // Bring models_ in line with the library's model list: same shared_ptr for a model that lives on, listeners for the others.
    void refreshModels() {
        int32_t n = 0;
        check(mf_num_models(ctx_, &n));
        std::vector<mf_model_info_t> infos((size_t)n);
        for (int i = 0; i < n; ++i) check(mf_model_info(ctx_, i, &infos[(size_t)i]));
        ModelList next;
        std::vector<ModelPointer> born;
        for (int i = 0; i < n; ++i) {
            auto it = std::find_if(models_.begin(), models_.end(), [&](const ModelPointer& m) { return m->id_ == infos[(size_t)i].id; });
            if (it != models_.end()) {
                (*it)->index_ = i;
                (*it)->classID_ = infos[(size_t)i].class_id;
                next.push_back(*it);
                models_.erase(it);
            } else {
                ModelPointer m(new Model(ctx_, infos[(size_t)i].id, i, infos[(size_t)i].class_id));
                next.push_back(m);
                born.push_back(m);
            }
        }
        ModelList gone;
        gone.swap(models_);
        models_.swap(next);
        for (auto& m : gone) {
            m->index_ = -1;
            for (auto& l : inactiveModelListeners_) l(m);
        }
        for (auto& m : born)
            if (m->id_ != 0)
                for (auto& l : newModelListeners_) l(m);
    }

    mf_ctx* ctx_ = nullptr;
    int width_ = 0, height_ = 0;
    std::string exportDir_;
    bool exportSegmentation_;
    unsigned queueLength_;
    std::queue<FrameDataPointer> frameQueue_;
    ModelList models_;
    std::vector<ModelListener> newModelListeners_, inactiveModelListeners_;
};

}  // namespace maskfusion

/*
 * mfseg_api.cpp -- runs the host half of the reference's MfSegmentation::performSegmentation from the reference's own text.
 * TEST INFRASTRUCTURE ONLY.  oracle/build_seg.py assembles the translation unit in memory:
 *     mfcv.h | Core/Utils/BoundingBox.h (reference text) | this file, with the line `MFSEG_SLICE` replaced by the statements of
 *     Core/Segmentation/MfSegmentation.cpp from "// Build use ignore map" (:219) up to, not including, "cudaDeviceSynchronize();" (:524)
 * and pipes it to g++.  Everything between those two lines -- ignore map, connected components, the five edge-growing sweeps, the
 * component x mask / component x model votes, the 65 % rule, label closing, the mask -> model rule with the new-model test, the 60 %
 * follow rule -- therefore runs from the reference's text.  What is restated here (by hand, cited) is only what precedes the slice in
 * the function: declarations and the per-model table set-up (:139-193), minus the CUDA calls that produce the binary edge image (an
 * input here) -- and the OpenCV primitives in mfcv.h.
 */
struct Model {
    int id, cls;
    unsigned int getID() const { return (unsigned)id; }
    int getClassID() const { return cls; }
};
typedef std::list<std::shared_ptr<Model>> ModelList;
typedef ModelList::iterator ModelListIterator;
struct FrameData {               /* Core/FrameData.h:25-48 */
    cv::Mat mask, rgb, depth;
    std::vector<int> classIDs;
    int64_t index = 0;
};
typedef std::shared_ptr<FrameData> FrameDataPointer;
struct SegmentationResult {      /* Core/Segmentation/SegmentationResult.h:32-73 (the members the slice touches) */
    cv::Mat fullSegmentation;
    bool hasNewLabel = false;
    struct ModelData {
        unsigned id;
        ModelListIterator modelListIterator;
        bool isNonStatic = false, isEmpty = true;
        unsigned superPixelCount = 0, pixelCount = 0;
        float avgConfidence = 0;
        int classID = -1;
        float depthMean = 0, depthStd = 0;
        ModelData(unsigned t_id) : id(t_id) {}
    };
    std::vector<ModelData> modelData;
};

struct Harness {
    /* MfSegmentation.h:42-62,101-109 and the constructor's initialisers (MfSegmentation.cpp:43,72-73) */
    float minMaskModelOverlap = 0.05f;
    int minMappedComponentSize = 160;
    int morphMaskIterations = 3, morphMaskRadius = 1;
    bool removeEdges = true, removeEdgeIslands = false;
    int personClassID = 255;
    float minRelSizeNew = 0.07f, maxRelSizeNew = 0.4f;
    struct ModelBuffers { unsigned int maskOverlap[256]; unsigned char modelID; };
    std::vector<ModelBuffers> modelBuffers;
    unsigned char maskToID[256], modelIDToIndex[256], modelIndexToID[256];
    cv::Mat semanticIgnoreMap, cv8UC1Buffer, cvLabelComps, cvLabelEdges;

    Harness(int w, int h) {
        cv8UC1Buffer.create(h, w, CV_8UC1); cvLabelComps.create(h, w, CV_32S); cvLabelEdges.create(h, w, CV_32S);
        semanticIgnoreMap = cv::Mat::zeros(h, w, CV_8UC1);
        memset(maskToID, 0, sizeof(maskToID)); memset(modelIDToIndex, 0, sizeof(modelIDToIndex)); memset(modelIndexToID, 0, sizeof(modelIndexToID));
        maskToID[255] = 255; maskToID[0] = 0;
    }
    void allocateModelBuffers(unsigned char numModels) { if (modelBuffers.size() < numModels) modelBuffers.resize(numModels); }

    SegmentationResult performSegmentation(ModelList& models, FrameDataPointer frame, unsigned char nextModelID, bool allowNew, cv::Mat projectedIDs) {
        /* :139-147 */
        SegmentationResult result;
        const int& width = frame->depth.cols;
        const int& height = frame->depth.rows;
        const size_t total = frame->depth.total();
        result.fullSegmentation = cv::Mat::zeros(height, width, CV_8UC1);
        const int nMasks = int(frame->classIDs.size());
        const int nModels = int(models.size());
        const size_t minNewMaskPixels = minRelSizeNew * total;
        const size_t maxNewMaskPixels = maxRelSizeNew * total;
        /* :160-193 */
        allocateModelBuffers(nModels + 1);
        auto modelItr = models.begin();
        for (unsigned char m = 0; m < models.size(); ++m, ++modelItr) {
            ModelBuffers& mBuffers = modelBuffers[m];
            auto& model = *modelItr;
            mBuffers.modelID = model->getID();
            SegmentationResult::ModelData modelData(model->getID());
            modelData.modelListIterator = modelItr;
            modelData.depthMean = 30;
            modelData.depthStd = 30;
            result.modelData.push_back(modelData);
            modelIDToIndex[model->getID()] = m;
            modelIndexToID[m] = model->getID();
        }
        if (allowNew) {
            modelIDToIndex[nextModelID] = models.size();
            modelIndexToID[models.size()] = nextModelID;
            modelBuffers[models.size()].modelID = nextModelID;
        }
        (void)width; (void)height; (void)minNewMaskPixels; (void)maxNewMaskPixels;
MFSEG_SLICE
        return result;
    }
};

extern "C" void mfseg_labels(int W, int H, const uint8_t* binary, const float* depth, const uint8_t* mask, const int32_t* classIDs, int nMasks,
                             const uint8_t* projectedIDs, const int32_t* modelIDs, const int32_t* modelClassIDs, int nModels, int nextModelID,
                             int allowNew, const float* params11, uint8_t* ignoreMap, uint8_t* full, int* hasNewLabel, int* newClassID) {
    Harness h(W, H);
    /* params11: {threshold, weightDistance, weightConvexity, morphEdgeIterations, morphEdgeRadius, morphMaskIterations, morphMaskRadius,
     * removeEdges, minRelSizeNew, maxRelSizeNew, personClassID} -- the layout of mf_segmentation_labels in include/maskfusion_amd.h */
    h.morphMaskIterations = (int)params11[5]; h.morphMaskRadius = (int)params11[6]; h.removeEdges = params11[7] != 0.f;
    h.minRelSizeNew = params11[8]; h.maxRelSizeNew = params11[9]; h.personClassID = (int)params11[10];
    const size_t P = (size_t)W * H;
    memcpy(h.cv8UC1Buffer.data, binary, P);
    memcpy(h.semanticIgnoreMap.data, ignoreMap, P);
    FrameDataPointer frame = std::make_shared<FrameData>();
    frame->depth.create(H, W, CV_32S); memcpy(frame->depth.data, depth, P * 4);
    if (nMasks > 0) { frame->mask.create(H, W, CV_8UC1); memcpy(frame->mask.data, mask, P); }
    frame->classIDs.assign(classIDs, classIDs + nMasks);
    cv::Mat proj(H, W, CV_8UC1);
    memcpy(proj.data, projectedIDs, P);
    ModelList models;
    for (int i = 0; i < nModels; ++i) models.push_back(std::make_shared<Model>(Model{modelIDs[i], modelClassIDs[i]}));
    SegmentationResult r = h.performSegmentation(models, frame, (unsigned char)nextModelID, allowNew != 0, proj);
    memcpy(full, r.fullSegmentation.data, P);
    memcpy(ignoreMap, h.semanticIgnoreMap.data, P);
    *hasNewLabel = r.hasNewLabel ? 1 : 0;
    *newClassID = r.hasNewLabel ? r.modelData.back().classID : -1;
}

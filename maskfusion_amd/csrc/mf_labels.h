// mf_labels.h -- host half of MfSegmentation (see mf_labels.hip)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <vector>

namespace mf {

// MfSegmentation parameters (Core/Segmentation/MfSegmentation.h:42-62, SegmentationPerformer.h:41-42)
struct SegParams {
    float threshold = 0.1f, weightDistance = 1.f, weightConvexity = 1.f;
    int morphEdgeIterations = 3, morphEdgeRadius = 1;
    int morphMaskIterations = 3, morphMaskRadius = 1;
    bool removeEdges = true;
    float minRelSizeNew = 0.07f, maxRelSizeNew = 0.4f;
    int personClassID = 255;
    float minMaskModelOverlap = 0.05f;   // MfSegmentation.cpp:43
    int minMappedComponentSize = 160;    // MfSegmentation.cpp:43
};

// A mask value that has no class id (FrameData::classIDs shorter than the ids the mask image uses, e.g. a 255 "ignore" label in a
// precomputed Mask####.png) counts as "no mask" (0).  Upstream indexes classIDs[mask] unchecked (MfSegmentation.cpp:226,311), an
// out-of-bounds read there; here every table indexed by a mask value goes through this.
__host__ __device__ inline int mask_id(int value, int nMasks) { return value < nMasks ? value : 0; }

struct SegModelInfo { int id; int classID; };
struct SegResult { bool hasNewLabel = false; int newClassID = -1; };

// binary: 255 = not an edge (threshold -> closing -> invert); depth: raw metric depth; mask / classIDs: the frame's instance
// masks (mask values index classIDs, nMasks = number of class ids, 0 = no masks); projectedIDs: GlobalProjection output;
// models: the model list in order (index 0 = background); ignoreMap: persistent semanticIgnoreMap; full: out, model id per
// pixel (255 = ignore).
void segmentation_host(const SegParams& prm, int W, int H, const uint8_t* binary, const float* depth, const uint8_t* mask,
                       const int32_t* classIDs, int nMasks, const uint8_t* projectedIDs, const std::vector<SegModelInfo>& models,
                       int nextModelID, bool allowNew, std::vector<uint8_t>& ignoreMap, uint8_t* full, SegResult& result);

// Device form of the same stage (mf_labels_gpu.hip).  All pointers are device memory unless noted.
struct PoseDev;
struct LabelsGpuArgs {
    SegParams prm; int W, H;
    const uint8_t* binary; const float* depth; const uint8_t* mask; const uint8_t* proj;   // mask may be null when nMasks == 0
    const int* class_ids; int nMasks;                       // device copy of the frame's class ids
    const int* model_ids; const int* model_cls; const PoseDev* const* model_poses; int nModels;   // models before the jump-rule drop
    int nextModelID; bool allowNew;
    uint8_t* ignoreMap;                                     // persistent semanticIgnoreMap
    uint8_t* full;                                          // out: model id per pixel (255 = ignore) == textureMask
    uint8_t* tmp_u8;
    int* L; int* compId; int* area; int* lab[2]; int* blockCounts;      // [P], [P], [P + 1], 2 x [P], [(P + 255) / 256]
    int4* bbox;                                             // [P + 1] left, top, right, bottom of each component
    int* compMask; int* compModel; int table_cap;           // [table_cap] each
    int* compToMask; int* compFollow;                       // [P + 1]
    unsigned* overlap;                                      // [64 * 256]
    void* tables;                                           // labels_gpu_table_bytes()
    int* result_host;                                       // pinned host: {hasNewLabel, newClassID, overflow, nComponents}
};
size_t labels_gpu_table_bytes();
void launch_labels_gpu(const LabelsGpuArgs& a, hipStream_t s);

}  // namespace mf

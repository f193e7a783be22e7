// Drives the C++ facade (include/maskfusion/MaskFusion.h) the way GUI/MainController.cpp:117-128,399-464,591-606 drives the
// reference: Resolution / Intrinsics singletons, the reference's constructor argument list, per-frame setters, processFrame,
// listeners, exports.  Used by tests/test_abi.py (compile + link, no GPU) and tests/test_gpu_facade.py (run on the GPU, poses
// compared with the Python mirror of the same ABI).
//   facade_main W H fx fy cx cy n_frames frames.bin out_dir multi
// frames.bin: per frame rgb[H*W*3] u8, depth[H*W] f32, mask[H*W] u8.  Prints one line per frame and model:
//   "pose <frame> <id> <16 floats column-major>", "count <frame> <id> <surfels>", "new <id>", "inactive <id>".
#include <maskfusion/MaskFusion.h>

#include <cstdio>
#include <cstdlib>
#include <fstream>

using namespace maskfusion;

int main(int argc, char** argv) {
    if (argc < 11) {
        std::puts("link ok");
        return 0;
    }
    const int W = std::atoi(argv[1]), H = std::atoi(argv[2]);
    const float fx = (float)std::atof(argv[3]), fy = (float)std::atof(argv[4]), cx = (float)std::atof(argv[5]), cy = (float)std::atof(argv[6]);
    const int n = std::atoi(argv[7]);
    const std::string out = argv[9];
    const bool multi = std::atoi(argv[10]) != 0;
    std::ifstream in(argv[8], std::ios::binary);
    if (!in) return 2;

    Resolution::setResolution(W, H);
    Intrinsics::setIntrinics(fx, fy, cx, cy);
    Device::set(0);
    Device::setSurfelBudget(1 << 20, 1 << 18);
    // MainController.cpp:399-402 order: timeDelta, countThresh, errThresh, covThresh, closeLoops, iclnuim, reloc, photoThresh, confGlobal,
    // confObject, depthCut, icpThresh, fastOdom, fernThresh, so3, frameToFrameRGB, modelSpawnOffset, matching, method, exportDir, exportSeg
    MaskFusion mf(200, 35000, 5e-05f, 1e-05f, false, false, false, 115, 4, 2, 3, 100, false, 0.3095f, false, false, 3,
                  Model::MatchingType::Drost, Segmentation::Method::MASK_FUSION, out, true, false, 0);
    mf.addNewModelListener([](ModelPointer m) { std::printf("new %u\n", m->getID()); });
    mf.addInactiveModelListener([](ModelPointer m) { std::printf("inactive %u\n", m->getID()); });
    mf.preallocateModels(1);
    mf.setEnableMultipleModels(multi);
    mf.setTrackAllModels(false);
    mf.setTrackableClassIds({});
    // GUI defaults pushed every frame (GUI/Tools/GUI.h:367-374), with a new-model size that fits the synthetic boxes
    mf.setMfThreshold(0.3f); mf.setMfWeightDistance(150.f); mf.setMfWeightConvexity(2.8f);
    mf.setMfMorphEdgeIterations(0); mf.setMfMorphMaskIterations(0); mf.setNewModelMinRelativeSize(0.004f);

    std::vector<uint8_t> rgb((size_t)W * H * 3), mask((size_t)W * H);
    std::vector<float> depth((size_t)W * H);
    for (int k = 0; k < n; ++k) {
        in.read((char*)rgb.data(), rgb.size());
        in.read((char*)depth.data(), depth.size() * sizeof(float));
        in.read((char*)mask.data(), mask.size());
        if (!in) return 3;
        auto frame = std::make_shared<FrameData>();
        frame->timestamp = k;
        frame->index = k;
        frame->rgb = rgb.data();
        frame->depth = depth.data();
        if (multi) {
            frame->mask = mask.data();
            frame->classIDs = {0, 41, 42};
        }
        mf.setIcpWeight(100.f);   // per-frame setters as MainController::run pushes them (:528-571)
        mf.setSo3(false);
        mf.setDepthCutoff(3.f);
        if (mf.processFrame(frame)) return 4;   // always false upstream (MaskFusion.cpp:606)
        for (auto& m : mf.getModels()) {
            const Matrix4f p = m->getPose();
            std::printf("pose %d %u", k, m->getID());
            for (float v : p) std::printf(" %.9g", v);
            std::printf("\ncount %d %u %u\n", k, m->getID(), m->lastCount());
        }
    }
    std::printf("tick %d models %zu bg_class %d nonstatic %d\n", mf.getTick(), mf.getModels().size(), mf.getBackgroundModel()->getClassID(),
                (int)mf.getBackgroundModel()->isNonstatic());
    auto map = mf.getBackgroundModel()->downloadMap();
    map.countValid(mf.getConfidenceThreshold());
    std::printf("map %u %u log %zu\n", map.numPoints, map.numValid, mf.getBackgroundModel()->getPoseLog().size());
    mf.predict();
    mf.savePly();
    mf.exportPoses();
    // the Model-level calls exist with the reference's signatures (Model.h:126-162); drive one more frame by hand
    auto frame = std::make_shared<FrameData>();
    frame->rgb = rgb.data(); frame->depth = depth.data();
    mf.stageFrame(frame);
    auto bg = mf.getBackgroundModel();
    std::vector<float> graph;
    const int t = mf.getTick();
    bg->performTracking(false, false, 100.f, true, false, false, mf.getMaxDepthProcessed(), nullptr, n, false);
    bg->predictIndices(t, mf.getMaxDepthProcessed(), mf.getTimeDelta());
    bg->fuse(t, nullptr, nullptr, nullptr, nullptr, 3.f, 1.f);
    bg->predictIndices(t, mf.getMaxDepthProcessed(), mf.getTimeDelta());
    bg->clean(t, graph, mf.getTimeDelta(), 3.f, false, nullptr, nullptr);
    bg->combinedPredict(mf.getMaxDepthProcessed(), t, t, mf.getTimeDelta(), 0);
    mf.endFrame(n);
    std::printf("manual %d %u w %.6f\n", mf.getTick(), bg->lastCount(), bg->computeFusionWeight(1.f));
    return 0;
}

// Host-side semantics of the C++ facade against a scripted stand-in for the library (tests/cpp/stub_abi.cpp): runs without a GPU.
#include <maskfusion/MaskFusion.h>

#include <cstdio>
#include <stdexcept>
#include <string>
#include <vector>

using namespace maskfusion;
extern "C" int stub_processed(long long* out, int max);
extern "C" double stub_param(const char* key);

#define CHECK(cond) do { if (!(cond)) { std::printf("FAILED line %d: %s\n", __LINE__, #cond); return 1; } } while (0)

static FrameDataPointer frame(long long ts) {
    static uint8_t rgb[4]; static float depth[1];
    auto f = std::make_shared<FrameData>();
    f->timestamp = ts; f->rgb = rgb; f->depth = depth;
    return f;
}

int main() {
    // Resolution / Intrinsics must be set first, as upstream asserts (Core/Utils/Resolution.h:66, Intrinsics.h:58)
    bool threw = false;
    try { MaskFusion bad; } catch (const std::logic_error&) { threw = true; }
    CHECK(threw);
    Resolution::setResolution(64, 48);
    Intrinsics::setIntrinics(50, 50, 32, 24);
    CHECK(Resolution::getInstance().numPixels() == 64 * 48 && Intrinsics::getInstance().cx() == 32.f);

    {   // frame queue: with frameQueueSize = 3 the frame processed is the one handed in two calls earlier (MaskFusion.cpp:206-209)
        MaskFusion mf(200, 35000, 5e-05f, 1e-05f, true, false, false, 115, 4, 2, 3, 10, false, 0.3095f, true, false, 20, Model::MatchingType::Drost,
                      Segmentation::Method::MASK_FUSION, "", false, false, 3);
        std::vector<unsigned> born, gone;
        mf.addNewModelListener([&](ModelPointer m) { born.push_back(m->getID()); });
        mf.addInactiveModelListener([&](ModelPointer m) { gone.push_back(m->getID()); });
        ModelPointer bg = mf.getBackgroundModel();
        CHECK(bg->getID() == 0 && mf.getModels().size() == 1);
        long long seen[16];
        CHECK(!mf.processFrame(frame(100)) && stub_processed(seen, 16) == 0);     // queued only
        CHECK(!mf.processFrame(frame(101)) && stub_processed(seen, 16) == 0);
        CHECK(!mf.processFrame(frame(102)) && stub_processed(seen, 16) == 1 && seen[0] == 100);
        mf.processFrame(frame(103));                                              // processes 101
        mf.processFrame(frame(104));                                              // processes 102: model 1 is born
        CHECK(stub_processed(seen, 16) == 3 && seen[2] == 102);
        CHECK(born == std::vector<unsigned>({1}) && gone.empty() && mf.getModels().size() == 2);
        ModelPointer one = mf.getModels().back();
        CHECK(one->getID() == 1 && one->getClassID() == 41 && one->getPose()[12] == 1.f);
        CHECK(mf.getBackgroundModel() == bg);                                      // the same shared_ptr for as long as the model lives
        mf.processFrame(frame(105));                                              // 103
        mf.processFrame(frame(106));                                              // 104: 1 inactive, 7 born -- list order 0, 7
        CHECK(born == std::vector<unsigned>({1, 7}) && gone == std::vector<unsigned>({1}));
        CHECK(mf.getModels().size() == 2 && mf.getModels().back()->getID() == 7 && mf.getModels().back()->getPose()[12] == 7.f);
        CHECK(one->getID() == 1 && one->getClassID() == 41);                       // an inactive model keeps its identity ...
        threw = false;
        try { one->getPose(); } catch (const std::runtime_error&) { threw = true; }  // ... but no longer answers for a live one
        CHECK(threw);
        mf.processFrame(frame(107));                                              // 105: 7 inactive
        CHECK(gone == std::vector<unsigned>({1, 7}) && mf.getModels().size() == 1 && mf.getModels().front() == bg);
        CHECK(mf.getTick() == 7);                                                  // 6 frames processed, tick starts at 1
    }
    {   // usePrecomputedMasksOnly forces the queue off (MaskFusion.cpp:37): every call processes its own frame
        MaskFusion mf(200, 35000, 5e-05f, 1e-05f, true, false, false, 115, 4, 2, 3, 10, false, 0.3095f, true, false, 20, Model::MatchingType::Drost,
                      Segmentation::Method::MASK_FUSION, "", false, true, 30);
        long long seen[4];
        mf.processFrame(frame(1));
        CHECK(stub_processed(seen, 4) == 1 && seen[0] == 1);
    }
    // what is not built says so instead of silently doing something else
    {   // frameToFrameRGB ("-ftf", GUI/MainController.cpp:252,539) is built since round 3: the constructor argument and the setter reach the core
        MaskFusion mf(200, 35000, 5e-05f, 1e-05f, true, false, false, 115, 4, 2, 3, 10, false, 0.3095f, true, /*frameToFrameRGB*/ true);
        CHECK(stub_param("frameToFrameRGB") == 1.0);
        mf.setFrameToFrameRGB(false);
        CHECK(stub_param("frameToFrameRGB") == 0.0);
    }
    threw = false;
    try { MaskFusion mf(200, 35000, 5e-05f, 1e-05f, true, false, false, 115, 4, 2, 3, 10, false, 0.3095f, true, false, 20, Model::MatchingType::Drost, Segmentation::Method::CO_FUSION); } catch (const std::invalid_argument&) { threw = true; }
    CHECK(threw);
    threw = false;
    { MaskFusion mf; try { mf.processFrame(nullptr); } catch (const std::invalid_argument&) { threw = true; } }
    CHECK(threw);
    std::puts("facade semantics ok");
    return 0;
}

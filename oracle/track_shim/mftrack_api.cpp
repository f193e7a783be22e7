// mftrack_api.cpp -- harness around the reference's OWN Gauss-Newton tracking loop, RGBDOdometry::getIncrementalTransformation
// (Core/Utils/RGBDOdometry.cpp:227-497), compiled for the CPU.  TEST INFRASTRUCTURE ONLY (same rule as oracle/mf_oracle.h).
//
// oracle/build_track.py cuts the definition of that member function out of the reference file in memory and puts it where this file
// says MFTRACK_SLICE; MFTRACK_ODOMETRY_PROVIDER becomes the text of Core/Utils/OdometryProvider.h (rodrigues, computeUpdateSE3).  The
// device functions the loop calls -- icpStep, rgbStep, so3Step, computeRgbResidual, computeDerivativeImages, projectToPointCloud -- are
// the reference's own, already compiled for the CPU in oracle/_ref (build_ref.py); Eigen is oracle/eigen_shim.  What this file
// supplies is the part of class RGBDOdometry the loop touches (member names and types as in Core/Utils/RGBDOdometry.h:77-147, values as
// its constructor sets them, RGBDOdometry.cpp:21-105), GPUConfig's launch shapes for a GPU that is not in its table
// (Core/Utils/GPUConfig.h:47-54), and no-op TICK / TOCK.
#include "mfref_cuda.h"
#undef __CUDACC__            // host pass: types.cuh then defines mat33(Eigen::Matrix<float, 3, 3, RowMajor>&), which the loop uses
#include <Eigen/Core>
#include <Eigen/Geometry>
#include "cudafuncs.cuh"

#include <algorithm>
#include <cassert>
#include <limits>
#include <vector>

// TICK / TOCK (Stopwatch.h) become call counters: how many times the loop launched each device function is how many iterations
// each of its exits allowed -- compared with the oracle's so3Iterations / iterationsRun
static int g_ticks[4];
static inline void mftrack_tick(const char* name) {
    static const char* const names[4] = {"so3Step", "computeRgbResidual", "icpStep", "rgbStep"};
    for (int i = 0; i < 4; i++) if (!strcmp(name, names[i])) g_ticks[i]++;
}
#define TICK(name) mftrack_tick(name)
#define TOCK(name)

struct GPUConfig {
    static GPUConfig& getInstance() { static GPUConfig g; return g; }
    int icpStepThreads = 128, icpStepBlocks = 112, rgbStepThreads = 128, rgbStepBlocks = 112, rgbResThreads = 256, rgbResBlocks = 336,
        so3StepThreads = 160, so3StepBlocks = 64;
};

MFTRACK_ODOMETRY_PROVIDER

class RGBDOdometry {
 public:
    RGBDOdometry(int width, int height, float cx, float cy, float fx, float fy, unsigned char mask, float distThresh, float angleThresh)
        : lastICPError(0), lastICPCount(width * height), lastRGBError(0), lastRGBCount(width * height), lastSO3Error(0),
          lastSO3Count(width * height), lastA(Eigen::Matrix<double, 6, 6, Eigen::RowMajor>::Zero()),
          lastb(Eigen::Matrix<double, 6, 1>::Zero()), sobelSize(3), sobelScale(1.0 / pow(2.0, sobelSize)), maxDepthDeltaRGB(0.07),
          maxDepthRGB(6.0), distThres_(distThresh), angleThres_(angleThresh), width(width), height(height), maskID(mask) {
        sumDataSE3.create(MAX_THREADS);
        outDataSE3.create(1);
        sumResidualRGB.create(MAX_THREADS);
        sumDataSO3.create(MAX_THREADS);
        outDataSO3.create(1);
        intr = CameraModel(fx, fy, cx, cy);
        iterations.resize(NUM_PYRS);
        vmaps_g_prev_.resize(NUM_PYRS);
        nmaps_g_prev_.resize(NUM_PYRS);
        minimumGradientMagnitudes = {5, 3, 1};
    }

    Eigen::Matrix4f getIncrementalTransformation(Eigen::Vector3f& trans, Eigen::Matrix<float, 3, 3, Eigen::RowMajor>& rot, const bool& rgbOnly,
                                                 const float& icpWeight, const bool& pyramid, const bool& fastOdom, const bool& so3,
                                                 const cudaSurfaceObject_t& icpErrorSurface, const cudaSurfaceObject_t& rgbErrorSurface);

    float lastICPError, lastICPCount, lastRGBError, lastRGBCount, lastSO3Error, lastSO3Count;
    Eigen::Matrix<double, 6, 6, Eigen::RowMajor> lastA;
    Eigen::Matrix<double, 6, 1> lastb;
    static const int NUM_PYRS = 3;

    std::vector<DeviceArray2D<float>> vmaps_g_prev_, nmaps_g_prev_;
    const std::vector<DeviceArray2D<float>>* vertexMapPyramid = nullptr;
    const std::vector<DeviceArray2D<float>>* normalMapPyramid = nullptr;
    const std::vector<DeviceArray2D<unsigned char>>* prevMaskPyramid = nullptr;
    CameraModel intr;
    DeviceArray<JtJJtrSE3> sumDataSE3, outDataSE3;
    DeviceArray<int2> sumResidualRGB;
    DeviceArray<JtJJtrSO3> sumDataSO3, outDataSO3;
    const int sobelSize;
    const float sobelScale, maxDepthDeltaRGB, maxDepthRGB;
    DeviceArray2D<short> nextdIdx[NUM_PYRS], nextdIdy[NUM_PYRS];
    DeviceArray2D<float> lastDepth[NUM_PYRS], nextDepth[NUM_PYRS];
    DeviceArray2D<unsigned char> lastMask[NUM_PYRS], nextMask[NUM_PYRS];
    DeviceArray2D<unsigned char> lastImage[NUM_PYRS], nextImage[NUM_PYRS], lastNextImage[NUM_PYRS];
    DeviceArray2D<DataTerm> corresImg[NUM_PYRS];
    DeviceArray2D<float3> pointClouds[NUM_PYRS];
    std::vector<int> iterations;
    std::vector<float> minimumGradientMagnitudes;
    float distThres_, angleThres_;
    const int width, height;
    unsigned char maskID;
};

MFTRACK_SLICE

namespace {
template <typename T>
void up2(DeviceArray2D<T>& a, const T* p, int rows, int cols) { a.upload(p, (size_t)cols * sizeof(T), rows, cols); }
template <typename T>
void zeros2(DeviceArray2D<T>& a, int rows, int cols) { std::vector<T> z((size_t)rows * cols); up2(a, z.data(), rows, cols); }
}  // namespace

extern "C" {

// One call of the reference's getIncrementalTransformation.
//   curr_v / curr_n: current-frame vertex / normal pyramid, camera frame, planar [3][H>>i][W>>i] (what initICP receives);
//   prev_v / prev_n: model pyramid already in the global frame (what initICPModel leaves in vmaps_g_prev_ / nmaps_g_prev_);
//   lastDepth .. nextImage: the populateRGBDData pyramids (may be NULL when icpWeight >= 100 and !rgbOnly: never read then);
//   lastNext2: level-2 intensity of the previous frame (may be NULL without so3).
//   Masks: all pixels carry the tracked model's id (maskID 0 everywhere), the single-model case.
//   R row-major 3x3 and t: pose in / out.  inc16: the returned increment, column-major.  stats6: lastICPError, lastICPCount,
//   lastRGBError, lastRGBCount, lastSO3Error, lastSO3Count.  lastA36 row-major, lastb6: the final normal equations.  ticks4: launches of
//   so3Step, computeRgbResidual, icpStep, rgbStep.
int mftrack_run(const float* const* curr_v, const float* const* curr_n, const float* const* prev_v, const float* const* prev_n,
                const float* const* lastDepth, const float* const* nextDepth, const uint8_t* const* lastImage, const uint8_t* const* nextImage,
                const uint8_t* lastNext2, int W, int H, float fx, float fy, float cx, float cy, int pyramid, int fastOdom, int so3,
                int rgbOnly, float icpWeight, float distThresh, float angleThresh, float* R, float* t, float* inc16, float* stats6,
                double* lastA36, double* lastb6, int* ticks4) {
    memset(g_ticks, 0, sizeof(g_ticks));
    RGBDOdometry odo(W, H, cx, cy, fx, fy, 0, distThresh, angleThresh);
    std::vector<DeviceArray2D<float>> vcur(3), ncur(3);
    std::vector<DeviceArray2D<unsigned char>> pmask(3);
    for (int i = 0; i < 3; i++) {
        const int w = W >> i, h = H >> i;
        up2(vcur[i], curr_v[i], 3 * h, w);
        up2(ncur[i], curr_n[i], 3 * h, w);
        up2(odo.vmaps_g_prev_[i], prev_v[i], 3 * h, w);
        up2(odo.nmaps_g_prev_[i], prev_n[i], 3 * h, w);
        zeros2(pmask[i], h, w);
        zeros2(odo.lastMask[i], h, w);
        zeros2(odo.nextMask[i], h, w);
        if (lastDepth) up2(odo.lastDepth[i], lastDepth[i], h, w); else zeros2(odo.lastDepth[i], h, w);
        if (nextDepth) up2(odo.nextDepth[i], nextDepth[i], h, w); else zeros2(odo.nextDepth[i], h, w);
        if (lastImage) up2(odo.lastImage[i], lastImage[i], h, w); else zeros2(odo.lastImage[i], h, w);
        if (nextImage) up2(odo.nextImage[i], nextImage[i], h, w); else zeros2(odo.nextImage[i], h, w);
        zeros2(odo.lastNextImage[i], h, w);
        odo.nextdIdx[i].create(h, w);
        odo.nextdIdy[i].create(h, w);
        odo.pointClouds[i].create(h, w);
        odo.corresImg[i].create(h, w);
    }
    if (lastNext2) up2(odo.lastNextImage[2], lastNext2, H >> 2, W >> 2);
    odo.vertexMapPyramid = &vcur;
    odo.normalMapPyramid = &ncur;
    odo.prevMaskPyramid = &pmask;

    Eigen::Matrix<float, 3, 3, Eigen::RowMajor> rot;
    memcpy(rot.data(), R, sizeof(float) * 9);
    Eigen::Vector3f trans(t[0], t[1], t[2]);
    const bool bRgbOnly = rgbOnly != 0, bPyramid = pyramid != 0, bFast = fastOdom != 0, bSo3 = so3 != 0;
    const cudaSurfaceObject_t none = 0;
    Eigen::Matrix4f inc = odo.getIncrementalTransformation(trans, rot, bRgbOnly, icpWeight, bPyramid, bFast, bSo3, none, none);
    memcpy(R, rot.data(), sizeof(float) * 9);
    for (int i = 0; i < 3; i++) t[i] = trans(i);
    memcpy(inc16, inc.data(), sizeof(float) * 16);
    const float st[6] = {odo.lastICPError, odo.lastICPCount, odo.lastRGBError, odo.lastRGBCount, odo.lastSO3Error, odo.lastSO3Count};
    memcpy(stats6, st, sizeof(st));
    memcpy(lastA36, odo.lastA.data(), sizeof(double) * 36);
    memcpy(lastb6, odo.lastb.data(), sizeof(double) * 6);
    memcpy(ticks4, g_ticks, sizeof(g_ticks));
    return 0;
}

}  // extern "C"

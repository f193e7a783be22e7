// mfweight_api.cpp -- harness around the reference's OWN Model::computeFusionWeight and Model::rodrigues2 (Core/Model/Model.cpp:449-464,
// 891-932) and the expression of Model::getLastTransform (Core/Model/Model.h:239), compiled from the reference's text by
// oracle/build_weight.py (MFWEIGHT_SLICES / MFWEIGHT_LAST_TRANSFORM are replaced in memory).  TEST INFRASTRUCTURE ONLY.  Eigen is
// oracle/eigen_shim (its JacobiSVD is a stand-in: only the product U V^T is used, which does not depend on the SVD algorithm).
#include <Eigen/Core>
#include <Eigen/Geometry>

#include <algorithm>
#include <cmath>
#include <cstring>

class Model {
 public:
    Eigen::Matrix4f pose, lastPose;
    inline const Eigen::Matrix4f& getPose() const { return pose; }
    inline Eigen::Matrix4f getLastTransform() const { return MFWEIGHT_LAST_TRANSFORM; }
    static Eigen::Vector3f rodrigues2(const Eigen::Matrix3f& matrix);
    float computeFusionWeight(float weightMultiplier) const;
};

MFWEIGHT_SLICES

extern "C" {
// pose16 / lastPose16 column-major (Eigen::Matrix4f storage)
float mfweight_fusion_weight(const float* pose16, const float* lastPose16, float weightMultiplier) {
    Model m;
    memcpy(m.pose.data(), pose16, 64);
    memcpy(m.lastPose.data(), lastPose16, 64);
    return m.computeFusionWeight(weightMultiplier);
}
void mfweight_rodrigues2(const float* R9_colmajor, float* out3) {
    Eigen::Matrix3f R;
    memcpy(R.data(), R9_colmajor, 36);
    const Eigen::Vector3f r = Model::rodrigues2(R);
    for (int i = 0; i < 3; i++) out3[i] = r(i);
}
}

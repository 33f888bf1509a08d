// Minimal stand-ins for the few Theia / Eigen / Ceres types that appear in the SIGNATURES of the
// rotation-averaging plugin surface, so the host layer builds in an image that has none of them.
// When the real headers are on the include path, define GSFM_USE_REAL_THEIA and include them before
// this file: the estimator header then compiles against the real types unchanged.
//
//   theia::ViewId / ViewIdPair ........ thirdparty/TheiaSfM/src/theia/sfm/types.h:47-50
//   theia::TwoViewInfo ................ thirdparty/TheiaSfM/src/theia/sfm/twoview_info.h:54-83
//   theia::RotationEstimator .......... .../sfm/global_pose_estimation/rotation_estimator.h:50-65
//   theia::RotationErrorType .......... include/pairwise_rotation_error_quat.hpp:50-61
//   CovarianceMap ..................... include/uncertainty.hpp:13
//   ceres::LossFunction ............... ceres/loss_function.h (Evaluate(double, double[3]) const)
#pragma once
#include <cstddef>
#include <cstdint>
#include <functional>
#include <unordered_map>
#include <utility>
#include <vector>

#include "../gsfm_rot.h"

#ifndef GSFM_USE_REAL_THEIA

namespace Eigen {
struct Vector3d {
  double v[3];
  Vector3d() : v{0, 0, 0} {}
  Vector3d(double x, double y, double z) : v{x, y, z} {}
  static Vector3d Zero() { return Vector3d(); }
  double* data() { return v; }
  const double* data() const { return v; }
  double& operator[](int k) { return v[k]; }
  const double& operator[](int k) const { return v[k]; }
  double& operator()(int k) { return v[k]; }
  const double& operator()(int k) const { return v[k]; }
  double x() const { return v[0]; }
  double y() const { return v[1]; }
  double z() const { return v[2]; }
};
struct Vector2d {
  double v[2];
  Vector2d() : v{0, 0} {}
  Vector2d(double x, double y) : v{x, y} {}
  double& operator[](int k) { return v[k]; }
  const double& operator[](int k) const { return v[k]; }
  double x() const { return v[0]; }
  double y() const { return v[1]; }
};
struct Matrix3d {  // column-major storage like Eigen's default
  double m[9];
  Matrix3d() : m{0, 0, 0, 0, 0, 0, 0, 0, 0} {}
  static Matrix3d Identity() { Matrix3d r; r.m[0] = r.m[4] = r.m[8] = 1.0; return r; }
  double& operator()(int r, int c) { return m[3 * c + r]; }
  const double& operator()(int r, int c) const { return m[3 * c + r]; }
  double* data() { return m; }
  const double* data() const { return m; }
};
}  // namespace Eigen

namespace ceres {
class LossFunction {
 public:
  virtual ~LossFunction() {}
  virtual void Evaluate(double sq_norm, double out[3]) const = 0;
};
}  // namespace ceres

namespace theia {
typedef uint32_t ViewId;
typedef std::pair<ViewId, ViewId> ViewIdPair;
static const ViewId kInvalidViewId = 0xffffffffu;
class Reconstruction;  // opaque here: track/camera storage is out of scope

struct TwoViewInfo {
  TwoViewInfo() : focal_length_1(0.0), focal_length_2(0.0), num_verified_matches(0), num_homography_inliers(0), visibility_score(1) {}
  double focal_length_1, focal_length_2;
  Eigen::Vector3d position_2;
  Eigen::Vector3d rotation_2;  // angle-axis of R_12 = R_2 R_1^T
  int num_verified_matches;
  int num_homography_inliers;
  int visibility_score;
};

class RotationEstimator {
 public:
  RotationEstimator() {}
  virtual ~RotationEstimator() {}
  virtual bool EstimateRotations(const std::unordered_map<ViewIdPair, TwoViewInfo>& view_pairs,
                                 std::unordered_map<ViewId, Eigen::Vector3d>* rotations) = 0;
 private:
  RotationEstimator(const RotationEstimator&) = delete;
  void operator=(const RotationEstimator&) = delete;
};

enum class RotationErrorType {
  QUATERNION_NORM = 0, ROTATION_MAT_FNORM = 1, QUATERNION_COSINE = 2, ANGLE_AXIS_COVARIANCE = 3, ANGLE_AXIS = 4,
  ANGLE_AXIS_INLIERS = 5, ANGLE_AXIS_COV_INLIERS = 6, ANGLE_AXIS_COVTRACE = 7, ANGLE_AXIS_COVNORM = 8
};
}  // namespace theia

namespace std {
template <> struct hash<theia::ViewIdPair> {  // theia/util/hash.h
  size_t operator()(const theia::ViewIdPair& p) const noexcept {
    uint64_t k = ((uint64_t)p.first << 32) | p.second;
    k ^= k >> 33; k *= 0xff51afd7ed558ccdULL; k ^= k >> 33;
    return (size_t)k;
  }
};
}  // namespace std

typedef std::unordered_map<theia::ViewIdPair, std::pair<Eigen::Matrix3d, Eigen::Vector3d>> CovarianceMap;

#endif  // GSFM_USE_REAL_THEIA

namespace gsfm {
// A ceres::LossFunction that can additionally describe itself to the device (include/gsfm_rot.h,
// gsfm_loss_node).  Losses without a descriptor are evaluated per edge on the host, exactly like the
// reference does through its Python trampoline (bind_src/GlobalSfMpy.cpp:33-65).
class DescribedLoss {
 public:
  virtual ~DescribedLoss() {}
  // Writes at most `cap` nodes, returns the program length, or -1 if the loss has no native form.
  virtual int NativeProgram(gsfm_loss_node* out, int cap) const = 0;
};
// Number of common tracks of a view pair; needed only by the *_INLIERS error types
// (estimator.cpp:260-273 reads them from the Reconstruction, which is out of scope here).
typedef std::function<int(const theia::ViewIdPair&)> CommonTrackCounter;
}  // namespace gsfm

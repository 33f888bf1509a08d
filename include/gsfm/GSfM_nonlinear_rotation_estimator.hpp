// theia::GSfMNonlinearRotationEstimator with the reference's public surface
// (reference include/GSfM_nonlinear_rotation_estimator.hpp:22-59), implemented on the MI355X
// solver through the C-ABI of include/gsfm_rot.h instead of Ceres.
#pragma once
#include <string>
#include <unordered_map>

#include "compat.hpp"

namespace theia {

// Aliases only: the parameter TYPES below are exactly the reference's
// (std::unordered_map<ViewIdPair, TwoViewInfo> const&, std::unordered_map<ViewId, Eigen::Vector3d>*).
using GsfmViewPairs = std::unordered_map<ViewIdPair, TwoViewInfo>;
using GsfmOrientations = std::unordered_map<ViewId, Eigen::Vector3d>;

class GSfMNonlinearRotationEstimator : public RotationEstimator {
 public:
  // robust_loss_width: width of the SoftL1 loss used by EstimateRotations() (0.1 when omitted, as in the reference).
  explicit GSfMNonlinearRotationEstimator(const double robust_loss_width) : robust_loss_width_(robust_loss_width) {}
  GSfMNonlinearRotationEstimator() : robust_loss_width_(0.1) {}

  // (a1) plugin entry point of theia::RotationEstimator — reference .cpp:24-80.
  //      SoftL1 loss, angle-axis residuals, unit weights. In/out: orientations (initial guess -> solution).
  bool EstimateRotations(const GsfmViewPairs& view_pairs, GsfmOrientations* global_orientations) override;

  // (a2) quaternion-parameterised residuals with a caller-supplied loss — reference .cpp:82-198.
  //      rotation_error_type in {QUATERNION_COSINE (default), QUATERNION_NORM, ROTATION_MAT_FNORM}.
  bool EstimateRotationsWithCustomizedLoss(const GsfmViewPairs& view_pairs, GsfmOrientations* global_orientations,
                                           ceres::LossFunction* loss_function, int thread_num,
                                           RotationErrorType rotation_error_type = RotationErrorType::QUATERNION_COSINE);

  // (a3) angle-axis residuals whitened by per-edge covariances / weights — reference .cpp:201-309.
  //      `covariances` is taken by value like the reference does; `reconstruction` is only consulted by the
  //      *_INLIERS types there (here: SetCommonTrackCounter).
  bool EstimateRotationsWithCustomizedLossAndCovariance(const GsfmViewPairs& view_pairs, GsfmOrientations* global_orientations,
                                                        ceres::LossFunction* loss_function, int thread_num,
                                                        CovarianceMap covariances, RotationErrorType rotation_error_type,
                                                        Reconstruction* reconstruction);

  // (a4) outer IRLS with MAGSAC (nu = 3) weights, inner full solve — reference .cpp:314-457.
  bool EstimateRotationsWithSigmaConsensus(const GsfmViewPairs& view_pairs, GsfmOrientations* global_orientations,
                                           ceres::LossFunction* loss_function, int thread_num, int iters_num, double sigma_max);

  // ---- additions of this build (absent from the reference) ----
  void SetCommonTrackCounter(gsfm::CommonTrackCounter f) { common_tracks_ = f; }  // feeds the *_INLIERS types
  const gsfm_rot_summary& LastSummary() const { return summary_; }               // the reference drops ceres' summary
  const char* LastError() const { return error_.c_str(); }
  // solver options of include/gsfm_rot.h (Ceres defaults + this build's linear-solver switches); starts from gsfm_rot_options_default
  gsfm_rot_options* MutableOptions() { if (!options_set_) { gsfm_rot_options_default(&options_); options_set_ = true; } return &options_; }

 private:
  bool Run(const GsfmViewPairs& view_pairs, GsfmOrientations* global_orientations, ceres::LossFunction* loss_function,
           const gsfm_loss_node* builtin_loss, int thread_num, const CovarianceMap* covariances, RotationErrorType type,
           int sigma_iters, double sigma_max);

  const double robust_loss_width_;
  gsfm::CommonTrackCounter common_tracks_;
  gsfm_rot_summary summary_{};
  gsfm_rot_options options_{};
  bool options_set_ = false;
  std::string error_;
};

}  // namespace theia

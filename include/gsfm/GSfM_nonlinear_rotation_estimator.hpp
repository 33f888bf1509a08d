// theia::GSfMNonlinearRotationEstimator with the reference's public surface
// (reference include/GSfM_nonlinear_rotation_estimator.hpp:22-59), implemented on the MI355X
// solver through the C-ABI of include/gsfm_rot.h instead of Ceres.
#pragma once
#include <string>
#include <unordered_map>

#include "compat.hpp"

namespace theia {

class GSfMNonlinearRotationEstimator : public RotationEstimator {
 public:
  GSfMNonlinearRotationEstimator() : robust_loss_width_(0.1) {}
  explicit GSfMNonlinearRotationEstimator(const double robust_loss_width) : robust_loss_width_(robust_loss_width) {}

  // SoftL1(robust_loss_width) loss, angle-axis residuals, unit weights (reference .cpp:24-80).
  bool EstimateRotations(const std::unordered_map<ViewIdPair, TwoViewInfo>& view_pairs,
                         std::unordered_map<ViewId, Eigen::Vector3d>* global_orientations) override;

  // Quaternion-parameterised residuals with a caller-supplied loss (reference .cpp:82-198).
  bool EstimateRotationsWithCustomizedLoss(const std::unordered_map<ViewIdPair, TwoViewInfo>& view_pairs,
                                           std::unordered_map<ViewId, Eigen::Vector3d>* global_orientations,
                                           ceres::LossFunction* loss_function, int thread_num,
                                           RotationErrorType rotation_error_type = RotationErrorType::QUATERNION_COSINE);

  // Angle-axis residuals whitened by per-edge covariances / weights (reference .cpp:201-309).
  bool EstimateRotationsWithCustomizedLossAndCovariance(const std::unordered_map<ViewIdPair, TwoViewInfo>& view_pairs,
                                                        std::unordered_map<ViewId, Eigen::Vector3d>* global_orientations,
                                                        ceres::LossFunction* loss_function, int thread_num,
                                                        CovarianceMap covariances, RotationErrorType rotation_error_type,
                                                        Reconstruction* reconstruction);

  // Outer IRLS with MAGSAC weights (reference .cpp:314-457).
  bool EstimateRotationsWithSigmaConsensus(const std::unordered_map<ViewIdPair, TwoViewInfo>& view_pairs,
                                           std::unordered_map<ViewId, Eigen::Vector3d>* global_orientations,
                                           ceres::LossFunction* loss_function, int thread_num, int iters_num, double sigma_max);

  // --- additions of this build (not in the reference) ---
  void SetCommonTrackCounter(gsfm::CommonTrackCounter f) { common_tracks_ = f; }  // for the *_INLIERS types
  const gsfm_rot_summary& LastSummary() const { return summary_; }               // the reference discards ceres' summary
  const char* LastError() const { return error_.c_str(); }
  gsfm_rot_options* MutableOptions() { options_set_ = true; return &options_; }

 private:
  bool Run(const std::unordered_map<ViewIdPair, TwoViewInfo>& view_pairs,
           std::unordered_map<ViewId, Eigen::Vector3d>* global_orientations, ceres::LossFunction* loss_function,
           const gsfm_loss_node* builtin_loss, int thread_num, const CovarianceMap* covariances,
           RotationErrorType type, int sigma_iters, double sigma_max);

  const double robust_loss_width_;
  gsfm::CommonTrackCounter common_tracks_;
  gsfm_rot_summary summary_{};
  gsfm_rot_options options_{};
  bool options_set_ = false;
  std::string error_;
};

}  // namespace theia

// theia::GSfMNonlinearRotationEstimator on the MI355X solver.
//
// Each entry point flattens the caller's maps into the SoA arrays of the C-ABI (dense camera index =
// rank of the ViewId among the views that have an initial orientation; edges sorted by ViewIdPair so
// the result does not depend on unordered_map iteration order), solves on the device, and writes the
// rotations back into the caller's map in place.  Skip rules, return values and ownership follow
// reference src/GSfM_nonlinear_rotation_estimator.cpp (line numbers below).
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <exception>
#include <limits>
#include <vector>

#include "../../include/gsfm/GSfM_nonlinear_rotation_estimator.hpp"

namespace theia {

namespace {

// The user's Evaluate may throw (a Python exception surfaces as pybind11::error_already_set).  Nothing may unwind through the
// extern "C" frames of the solver: the first exception is parked, the edge gets NaN -- the solve then ends with FAILURE /
// nonfinite instead of running on stale values -- and Run() rethrows it once the device problem has been released.
struct HostLossCall {
  const ceres::LossFunction* loss;
  std::exception_ptr error;
};
void host_loss_trampoline(void* user, double s, double out[3]) {
  HostLossCall* call = static_cast<HostLossCall*>(user);
  try {
    call->loss->Evaluate(s, out);
  } catch (...) {
    if (!call->error) call->error = std::current_exception();
    out[0] = out[1] = out[2] = std::numeric_limits<double>::quiet_NaN();
  }
}
struct ProblemOwner {   // the device problem is released on every path out of Run()
  gsfm_rot_problem* p = nullptr;
  ~ProblemOwner() { if (p) gsfm_rot_problem_destroy(p); }
};

bool needs_cov(RotationErrorType t) {
  return t == RotationErrorType::ANGLE_AXIS_COVARIANCE || t == RotationErrorType::ANGLE_AXIS_COV_INLIERS ||
         t == RotationErrorType::ANGLE_AXIS_COVTRACE || t == RotationErrorType::ANGLE_AXIS_COVNORM;
}
bool needs_inliers(RotationErrorType t) {
  return t == RotationErrorType::ANGLE_AXIS_INLIERS || t == RotationErrorType::ANGLE_AXIS_COV_INLIERS;
}

}  // namespace

bool GSfMNonlinearRotationEstimator::Run(const std::unordered_map<ViewIdPair, TwoViewInfo>& view_pairs,
                                         std::unordered_map<ViewId, Eigen::Vector3d>* global_orientations,
                                         ceres::LossFunction* loss_function, const gsfm_loss_node* builtin_loss,
                                         int thread_num, const CovarianceMap* covariances, RotationErrorType type,
                                         int sigma_iters, double sigma_max) {
  error_.clear();
  std::memset(&summary_, 0, sizeof(summary_));
  if (global_orientations == nullptr) {  // CHECK_NOTNULL in the reference (:28) aborts; we refuse
    error_ = "global_orientations is NULL";
    return false;
  }
  if (global_orientations->size() == 0) return false;  // :29-34 "no initialization was provided"
  if (view_pairs.size() == 0) return false;            // :35-40 "no relative rotation constraints"
  if (needs_inliers(type) && !common_tracks_) {
    error_ = "the *_INLIERS error types need SetCommonTrackCounter() (track storage is out of scope)";
    return false;
  }

  // dense camera indices
  std::vector<ViewId> ids;
  ids.reserve(global_orientations->size());
  for (const auto& kv : *global_orientations) ids.push_back(kv.first);
  std::sort(ids.begin(), ids.end());
  std::unordered_map<ViewId, uint32_t> index;
  index.reserve(ids.size() * 2);
  for (size_t k = 0; k < ids.size(); ++k) index[ids[k]] = (uint32_t)k;

  // edges: skip those whose views lack an orientation (:57-60) or, for the *_COV* types, a covariance (:239-247)
  std::vector<ViewIdPair> keys;
  keys.reserve(view_pairs.size());
  for (const auto& kv : view_pairs) {
    if (!index.count(kv.first.first) || !index.count(kv.first.second)) continue;
    if (needs_cov(type) && (covariances == nullptr || !covariances->count(kv.first))) continue;
    keys.push_back(kv.first);
  }
  std::sort(keys.begin(), keys.end());
  const size_t E = keys.size(), N = ids.size();
  std::vector<uint32_t> ei(E), ej(E);
  std::vector<double> rel(3 * E), cov6, inl, rot(3 * N);
  if (needs_cov(type)) cov6.resize(6 * E);
  if (needs_inliers(type)) inl.resize(E);
  for (size_t e = 0; e < E; ++e) {
    const TwoViewInfo& info = view_pairs.at(keys[e]);
    ei[e] = index[keys[e].first];
    ej[e] = index[keys[e].second];
    for (int c = 0; c < 3; ++c) rel[3 * e + c] = info.rotation_2[c];
    if (needs_cov(type)) {
      const Eigen::Matrix3d& C = covariances->at(keys[e]).first;
      double* o = &cov6[6 * e];
      o[0] = C(0, 0); o[1] = C(1, 1); o[2] = C(2, 2); o[3] = C(0, 1); o[4] = C(0, 2); o[5] = C(1, 2);
    }
    if (needs_inliers(type)) inl[e] = common_tracks_(keys[e]) / 100.0;  // :263, :268
  }
  for (size_t k = 0; k < N; ++k) {
    const Eigen::Vector3d& w = global_orientations->at(ids[k]);
    rot[3 * k] = w[0]; rot[3 * k + 1] = w[1]; rot[3 * k + 2] = w[2];
  }
  if (E == 0) return true;  // Ceres would solve an empty problem and the reference returns true

  ProblemOwner owner;
  gsfm_status st = gsfm_rot_problem_create((uint32_t)N, E, ei.data(), ej.data(), rel.data(), (int32_t)type,
                                           cov6.empty() ? nullptr : cov6.data(), inl.empty() ? nullptr : inl.data(), nullptr, &owner.p);
  if (st != GSFM_OK) { error_ = gsfm_last_error(); return false; }
  gsfm_rot_problem* P = owner.p;

  // loss: built-in descriptor > self-describing loss > host callback > NULL (Ceres' trivial loss)
  HostLossCall host_call{loss_function, nullptr};
  if (builtin_loss) st = gsfm_rot_set_loss(P, builtin_loss, 1);
  else if (loss_function == nullptr) st = gsfm_rot_set_loss(P, nullptr, 0);
  else {
    gsfm_loss_node prog[GSFM_LOSS_MAX_NODES];
    int n = -1;
    if (const gsfm::DescribedLoss* d = dynamic_cast<const gsfm::DescribedLoss*>(loss_function)) n = d->NativeProgram(prog, GSFM_LOSS_MAX_NODES);
    if (n >= 0) st = gsfm_rot_set_loss(P, prog, n);
    else st = gsfm_rot_set_loss_callback(P, host_loss_trampoline, &host_call);
  }
  if (st != GSFM_OK) { error_ = gsfm_last_error(); return false; }

  gsfm_rot_options opt;
  if (options_set_) opt = options_;
  else gsfm_rot_options_default(&opt);  // max_num_iterations = 200 etc. (:72-74)
  opt.num_threads = thread_num;
  if (sigma_iters > 0) st = gsfm_rot_solve_sigma_consensus(P, rot.data(), sigma_iters, sigma_max, &opt, &summary_);
  else st = gsfm_rot_solve(P, rot.data(), &opt, &summary_);
  if (st != GSFM_OK) error_ = gsfm_last_error();
  if (host_call.error) {   // the user's loss threw: release the device problem first, then let the exception continue to the caller
    gsfm_rot_problem_destroy(owner.p);
    owner.p = nullptr;
    std::rethrow_exception(host_call.error);
  }
  if (st != GSFM_OK) return false;

  for (size_t k = 0; k < N; ++k) {
    Eigen::Vector3d& w = (*global_orientations)[ids[k]];
    w[0] = rot[3 * k]; w[1] = rot[3 * k + 1]; w[2] = rot[3 * k + 2];
  }
  return true;  // like the reference, convergence is not inspected (:79)
}

bool GSfMNonlinearRotationEstimator::EstimateRotations(const std::unordered_map<ViewIdPair, TwoViewInfo>& view_pairs,
                                                       std::unordered_map<ViewId, Eigen::Vector3d>* global_orientations) {
  gsfm_loss_node soft_l1;  // new ceres::SoftLOneLoss(robust_loss_width_) (:44-45)
  std::memset(&soft_l1, 0, sizeof(soft_l1));
  soft_l1.kind = GSFM_LOSS_SOFT_L1;
  soft_l1.p[0] = robust_loss_width_;
  return Run(view_pairs, global_orientations, nullptr, &soft_l1, 1, nullptr, RotationErrorType::ANGLE_AXIS, 0, 0.0);
}

bool GSfMNonlinearRotationEstimator::EstimateRotationsWithCustomizedLoss(
    const std::unordered_map<ViewIdPair, TwoViewInfo>& view_pairs, std::unordered_map<ViewId, Eigen::Vector3d>* global_orientations,
    ceres::LossFunction* loss_function, int thread_num, RotationErrorType rotation_error_type) {
  // the reference only builds a cost function for the three quaternion types (:147-153); any other value
  // would hand Ceres a null cost function
  if (rotation_error_type != RotationErrorType::ROTATION_MAT_FNORM && rotation_error_type != RotationErrorType::QUATERNION_NORM &&
      rotation_error_type != RotationErrorType::QUATERNION_COSINE) {
    error_ = "EstimateRotationsWithCustomizedLoss supports QUATERNION_COSINE, QUATERNION_NORM and ROTATION_MAT_FNORM";
    return false;
  }
  return Run(view_pairs, global_orientations, loss_function, nullptr, thread_num, nullptr, rotation_error_type, 0, 0.0);
}

bool GSfMNonlinearRotationEstimator::EstimateRotationsWithCustomizedLossAndCovariance(
    const std::unordered_map<ViewIdPair, TwoViewInfo>& view_pairs, std::unordered_map<ViewId, Eigen::Vector3d>* global_orientations,
    ceres::LossFunction* loss_function, int thread_num, CovarianceMap covariances, RotationErrorType rotation_error_type,
    Reconstruction* /*reconstruction*/) {
  if ((int)rotation_error_type < (int)RotationErrorType::ANGLE_AXIS_COVARIANCE) {
    error_ = "EstimateRotationsWithCustomizedLossAndCovariance supports the ANGLE_AXIS* error types";
    return false;
  }
  return Run(view_pairs, global_orientations, loss_function, nullptr, thread_num, &covariances, rotation_error_type, 0, 0.0);
}

bool GSfMNonlinearRotationEstimator::EstimateRotationsWithSigmaConsensus(
    const std::unordered_map<ViewIdPair, TwoViewInfo>& view_pairs, std::unordered_map<ViewId, Eigen::Vector3d>* global_orientations,
    ceres::LossFunction* loss_function, int thread_num, int iters_num, double sigma_max) {
  if (iters_num <= 0) {  // the reference's loop body never runs (:356)
    if (global_orientations == nullptr || global_orientations->empty() || view_pairs.empty()) return false;
    return true;
  }
  return Run(view_pairs, global_orientations, loss_function, nullptr, thread_num, nullptr, RotationErrorType::ANGLE_AXIS, iters_num, sigma_max);
}

}  // namespace theia

// C++ test of the plugin surface theia::GSfMNonlinearRotationEstimator (no Python involved), following the
// fixture of Theia's robust_rotation_estimator_test.cc:121-241: random ground-truth orientations, chain +
// random extra edges, R_ij = N * R_j * R_i^T with angular noise, initial guess by chaining, alignment, bound on
// the per-view angular error.  Built and run by tests/test_gpu_host_layer.py (needs a GPU).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../../include/gsfm/GSfM_nonlinear_rotation_estimator.hpp"
#include "../../include/gsfm/view_graph.hpp"

using namespace theia;

namespace {
struct Q { double x, y, z, w; };
Q mul(const Q& a, const Q& b) { return {a.w*b.x + a.x*b.w + a.y*b.z - a.z*b.y, a.w*b.y + a.y*b.w + a.z*b.x - a.x*b.z, a.w*b.z + a.z*b.w + a.x*b.y - a.y*b.x, a.w*b.w - a.x*b.x - a.y*b.y - a.z*b.z}; }
Q conj(const Q& a) { return {-a.x, -a.y, -a.z, a.w}; }
Q from_aa(const Eigen::Vector3d& v) { const double t = std::sqrt(v[0]*v[0] + v[1]*v[1] + v[2]*v[2]); if (t < 1e-15) return {0.5*v[0], 0.5*v[1], 0.5*v[2], 1.0}; const double k = std::sin(0.5*t)/t; return {v[0]*k, v[1]*k, v[2]*k, std::cos(0.5*t)}; }
Eigen::Vector3d to_aa(const Q& q) { const double s = std::sqrt(q.x*q.x + q.y*q.y + q.z*q.z); if (s == 0) return Eigen::Vector3d(2*q.x, 2*q.y, 2*q.z); const double th = 2.0 * (q.w < 0 ? std::atan2(-s, -q.w) : std::atan2(s, q.w)); return Eigen::Vector3d(q.x*th/s, q.y*th/s, q.z*th/s); }
double angle_between(const Q& a, const Q& b) { const Q d = mul(a, conj(b)); return 2.0 * std::atan2(std::sqrt(d.x*d.x + d.y*d.y + d.z*d.z), std::fabs(d.w)); }

int failures = 0;
#define EXPECT(c) do { if (!(c)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); ++failures; } } while (0)

// A loss implemented in C++ by the caller, WITHOUT a native descriptor: goes through the host callback path.
struct CallerHuber : public ceres::LossFunction {
  double a;
  explicit CallerHuber(double a_) : a(a_) {}
  void Evaluate(double s, double out[3]) const override {
    const double b = a * a;
    if (s > b) { const double r = std::sqrt(s); out[0] = 2*a*r - b; out[1] = a/r; out[2] = -out[1]/(2*s); }
    else { out[0] = s; out[1] = 1.0; out[2] = 0.0; }
  }
};
// The same loss WITH a descriptor: evaluated on the device.
struct DescribedHuber : public CallerHuber, public gsfm::DescribedLoss {
  explicit DescribedHuber(double a_) : CallerHuber(a_) {}
  int NativeProgram(gsfm_loss_node* out, int cap) const override { if (cap < 1) return -1; out[0] = gsfm_loss_node{GSFM_LOSS_HUBER, 0, {a, 0, 0}}; return 1; }
};

void run(int num_views, int num_extra_edges, double noise_deg, double tol_deg, int variant) {
  std::mt19937_64 rng(56);
  std::uniform_real_distribution<double> U(-1.0, 1.0);
  std::normal_distribution<double> Nrm;
  std::vector<Q> gt(num_views);
  for (auto& q : gt) q = from_aa(Eigen::Vector3d(0.2 * U(rng), 0.2 * U(rng), 0.2 * U(rng)));   // test.cc:160-163
  ViewGraph graph;
  auto add = [&](ViewId i, ViewId j) {
    if (i == j || graph.HasEdge(i, j)) return;
    if (i > j) std::swap(i, j);
    Eigen::Vector3d axis(Nrm(rng), Nrm(rng), Nrm(rng));
    const double n = std::sqrt(axis[0]*axis[0] + axis[1]*axis[1] + axis[2]*axis[2]), ang = noise_deg * M_PI / 180.0;
    const Q noise = from_aa(Eigen::Vector3d(axis[0]/n*ang, axis[1]/n*ang, axis[2]/n*ang));
    TwoViewInfo info;
    info.rotation_2 = to_aa(mul(noise, mul(gt[j], conj(gt[i]))));                                // test.cc:59-73
    graph.AddEdge(i, j, info);
  };
  for (int k = 1; k < num_views; ++k) add(k - 1, k);
  for (int k = 0; k < num_extra_edges; ++k) add(rng() % num_views, rng() % num_views);
  std::unordered_map<ViewId, Eigen::Vector3d> est;
  est[0] = to_aa(gt[0]);
  for (int k = 1; k < num_views; ++k)                                                             // test.cc:203-211
    est[k] = to_aa(mul(from_aa(graph.GetEdge(k - 1, k)->rotation_2), from_aa(est[k - 1])));
  GSfMNonlinearRotationEstimator estimator(0.1);
  bool ok = false;
  CallerHuber caller_loss(0.1);
  DescribedHuber described_loss(0.1);
  if (variant == 0) ok = estimator.EstimateRotations(graph.GetAllEdges(), &est);
  else if (variant == 1) ok = estimator.EstimateRotationsWithCustomizedLoss(graph.GetAllEdges(), &est, &caller_loss, 4, RotationErrorType::QUATERNION_COSINE);
  else if (variant == 2) ok = estimator.EstimateRotationsWithCustomizedLoss(graph.GetAllEdges(), &est, &described_loss, 4, RotationErrorType::QUATERNION_COSINE);
  else ok = estimator.EstimateRotationsWithSigmaConsensus(graph.GetAllEdges(), &est, nullptr, 4, 5, 0.1);
  EXPECT(ok);
  if (!ok) { std::printf("  error: %s\n", estimator.LastError()); return; }
  // gauge: align view 0, then bound every view's error (the Theia test aligns with AlignRotations)
  const Q g = mul(conj(from_aa(est[0])), gt[0]);
  double worst = 0;
  for (int k = 0; k < num_views; ++k) worst = std::max(worst, angle_between(mul(from_aa(est[k]), g), gt[k]) * 180.0 / M_PI);
  std::printf("  views %d edges %d noise %.1f deg variant %d: %d LM iterations, worst error %.3e deg (bound %.1e)\n", num_views, graph.NumEdges(), noise_deg, variant,
              estimator.LastSummary().num_iterations, worst, tol_deg);
  EXPECT(worst < tol_deg);
}
}  // namespace

int main() {
  run(4, 6, 0.0, 1e-8, 0);       // SmallTestNoNoise      (test.cc:215-221)
  run(4, 6, 1.0, 2.0, 0);        // SmallTestWithNoise    (:223-230), bound relaxed for the view-0 gauge
  run(100, 800, 2.0, 5.0, 0);    // LargeTestWithNoise    (:232-241)
  run(100, 800, 0.0, 1e-6, 1);   // quaternion residuals, caller-defined loss (host callback path)
  run(100, 800, 0.0, 1e-6, 2);   // quaternion residuals, self-describing loss (device path)
  run(100, 800, 1.0, 5.0, 3);    // sigma consensus
  // reference return values (estimator.cpp:29-40)
  GSfMNonlinearRotationEstimator e;
  std::unordered_map<ViewId, Eigen::Vector3d> none;
  std::unordered_map<ViewIdPair, TwoViewInfo> no_edges;
  EXPECT(!e.EstimateRotations(no_edges, &none));
  std::printf(failures ? "FAILED (%d)\n" : "PASSED\n", failures);
  return failures ? 1 : 0;
}

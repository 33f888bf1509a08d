// Optional CPU baseline: the rotation-averaging problem of bench.py built with the Ceres API -- one AutoDiff residual block
// per edge, the whitened angle-axis residual r = Lt log(R_j R_i^T R_ij^T), the MAGSAC sigma-consensus loss, LM + SPARSE_NORMAL_CHOLESKY,
// 200 iterations: what reference src/GSfM_nonlinear_rotation_estimator.cpp:201-309 sets up, written from its description (these are this
// build's own functors, not the reference's sources).  NOT built in this image (no Ceres / Eigen); see CMakeLists.txt.
//
// input: the binary graph written by dump_graph.py
//   u64 n_cams, u64 n_edges, then u32 edge_i[E], u32 edge_j[E], f64 rel_aa[3E], f64 cov6[6E], f64 init_aa[3N]
// output: one JSON line with edge-residuals/s = E * (residual evaluations) / solve seconds.
#include <ceres/ceres.h>
#include <ceres/rotation.h>

#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

namespace {

// r = Lt * log(R_j R_i^T R_ij^T), Lt upper triangular (6 values: l00 l01 l02 l11 l12 l22)
struct WhitenedAngleAxisError {
  WhitenedAngleAxisError(const double* rel_aa, const double* lt) { for (int k = 0; k < 3; ++k) rel_[k] = rel_aa[k]; for (int k = 0; k < 6; ++k) lt_[k] = lt[k]; }
  template <typename T>
  bool operator()(const T* const ri, const T* const rj, T* residual) const {
    T Ri[9], Rj[9], Rij[9];
    ceres::AngleAxisToRotationMatrix(ri, Ri);   // column-major, as ceres::MatrixAdapter's default
    ceres::AngleAxisToRotationMatrix(rj, Rj);
    const T rel[3] = {T(rel_[0]), T(rel_[1]), T(rel_[2])};
    ceres::AngleAxisToRotationMatrix(rel, Rij);
    auto at = [](const T* M, int r, int c) -> const T& { return M[r + 3 * c]; };
    T C[9], E[9];   // C = R_j R_i^T, E = C R_ij^T (column-major)
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { T s = T(0); for (int k = 0; k < 3; ++k) s += at(Rj, r, k) * at(Ri, c, k); C[r + 3 * c] = s; }
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { T s = T(0); for (int k = 0; k < 3; ++k) s += at(C, r, k) * at(Rij, c, k); E[r + 3 * c] = s; }
    T e[3];
    ceres::RotationMatrixToAngleAxis(E, e);
    residual[0] = T(lt_[0]) * e[0] + T(lt_[1]) * e[1] + T(lt_[2]) * e[2];
    residual[1] = T(lt_[3]) * e[1] + T(lt_[4]) * e[2];
    residual[2] = T(lt_[5]) * e[2];
    return true;
  }
  double rel_[3], lt_[6];
};

// MAGSACWeightBasedLoss(sigma), nu = 3, non-inverse (formulas of scripts/loss_functions.py:285-341; table Gamma(1, x/1000) = exp(-x/1000))
class MagsacLoss : public ceres::LossFunction {
 public:
  explicit MagsacLoss(double sigma) : sigma_(sigma) {
    const double C = 4.029720004054876e-01, q = 3.368214175218727;
    gk_ = 3.439485560754856e-03;
    K_ = C * 2.0;                       // C * 2^((nu-1)/2)
    ssm2_ = 2.0 * sigma * sigma;
    cut_ = q * q * sigma * sigma;
    w0_ = K_ / sigma * (1.0 - gk_);     // Gamma(1) = 1
  }
  void Evaluate(double s, double out[3]) const override {
    bool zero = false;
    if (s > cut_) { s = cut_; zero = true; }
    const double x = std::nearbyint(1000.0 * s / ssm2_);
    double sq = x * ssm2_ / 1000.0;
    const double w = K_ / sigma_ * (std::exp(-x / 1000.0) - gk_);
    const double wd = -K_ * std::exp(-sq / ssm2_) / (2.0 * sigma_ * sigma_ * sigma_);
    if (sq < 1e-7) sq = 1e-7;
    const double wdd = 2.0 * K_ * (1.0 / (sigma_ * sigma_)) * std::exp(-sq / ssm2_) / (8.0 * sigma_ * sigma_ * sigma_);
    out[0] = w0_ - w; out[1] = -wd; out[2] = -wdd;
    if (out[1] == 0.0) out[1] = 1e-5;
    if (zero) { out[1] = 1e-5; out[2] = 0.0; }
  }
 private:
  double sigma_, K_, gk_, ssm2_, cut_, w0_;
};

// Lt with Lt^T Lt = (1e8 * Sigma)^-1, upper triangular (reference estimator.cpp:252-256), from cov6 = C00 C11 C22 C01 C02 C12
void whitening(const double* c6, double* lt) {
  const double c00 = c6[0] * 1e8, c11 = c6[1] * 1e8, c22 = c6[2] * 1e8, c01 = c6[3] * 1e8, c02 = c6[4] * 1e8, c12 = c6[5] * 1e8;
  const double k00 = c11 * c22 - c12 * c12, k10 = c12 * c02 - c01 * c22, k20 = c01 * c12 - c11 * c02;
  const double id = 1.0 / (c00 * k00 + c01 * k10 + c02 * k20);
  const double p00 = k00 * id, p10 = k10 * id, p20 = k20 * id, p11 = (c00 * c22 - c02 * c02) * id, p21 = (c02 * c01 - c00 * c12) * id, p22 = (c00 * c11 - c01 * c01) * id;
  const double l00 = std::sqrt(p00), l10 = p10 / l00, l20 = p20 / l00, l11 = std::sqrt(p11 - l10 * l10), l21 = (p21 - l20 * l10) / l11;
  const double l22 = std::sqrt(p22 - l20 * l20 - l21 * l21);
  lt[0] = l00; lt[1] = l10; lt[2] = l20; lt[3] = l11; lt[4] = l21; lt[5] = l22;
}

template <typename T> bool read_vec(FILE* f, std::vector<T>* v, size_t n) { v->resize(n); return fread(v->data(), sizeof(T), n, f) == n; }

}  // namespace

int main(int argc, char** argv) {
  if (argc < 2) { fprintf(stderr, "usage: bench_ceres graph.bin [threads]\n"); return 2; }
  const int threads = argc > 2 ? atoi(argv[2]) : 1;
  FILE* f = fopen(argv[1], "rb");
  if (!f) { perror(argv[1]); return 2; }
  uint64_t hdr[2];
  if (fread(hdr, 8, 2, f) != 2) return 2;
  const size_t N = hdr[0], E = hdr[1];
  std::vector<uint32_t> ei, ej; std::vector<double> rel, cov, x;
  if (!read_vec(f, &ei, E) || !read_vec(f, &ej, E) || !read_vec(f, &rel, 3 * E) || !read_vec(f, &cov, 6 * E) || !read_vec(f, &x, 3 * N)) { fprintf(stderr, "short file\n"); return 2; }
  fclose(f);
  ceres::Problem problem;
  MagsacLoss loss(0.02);
  for (size_t e = 0; e < E; ++e) {
    double lt[6];
    whitening(&cov[6 * e], lt);
    ceres::CostFunction* cost = new ceres::AutoDiffCostFunction<WhitenedAngleAxisError, 3, 3, 3>(new WhitenedAngleAxisError(&rel[3 * e], lt));
    problem.AddResidualBlock(cost, &loss, &x[3 * ei[e]], &x[3 * ej[e]]);
  }
  ceres::Problem::Options popt; (void)popt;
  ceres::Solver::Options options;
  options.linear_solver_type = ceres::SPARSE_NORMAL_CHOLESKY;   // estimator.cpp:299-302
  options.max_num_iterations = 200;
  options.num_threads = threads;
  ceres::Solver::Summary summary;
  const auto t0 = std::chrono::steady_clock::now();
  ceres::Solve(options, &problem, &summary);
  const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  const int sweeps = summary.num_residual_evaluations;
  printf("{\"kind\": \"ceres\", \"ceres_version\": \"%s\", \"threads\": %d, \"cams\": %zu, \"edges\": %zu, \"lm_iterations\": %d, \"residual_sweeps\": %d, "
         "\"seconds\": %.3f, \"value\": %.6e, \"unit\": \"edge-residuals/s\", \"final_cost\": %.9e, \"termination\": \"%s\"}\n",
         CERES_VERSION_STRING, threads, N, E, (int)summary.iterations.size() - 1, sweeps, secs, (double)E * sweeps / secs, summary.final_cost,
         ceres::TerminationTypeToString(summary.termination_type));
  return 0;
}

// TEST INFRASTRUCTURE — CPU oracle. Not part of the product; never linked by it.
//
// Restatement of the reference's per-edge rotation covariance estimation
//   get_covariance_rot ............ src/uncertainty.cpp:82-162
//   MatchedFeaturesSampsonError ... src/uncertainty.cpp:36-81 (AutoDiffCostFunction<.,1,3,3>)
// i.e. refine (rotation, translation) of one view pair on the Sampson distance of its matched features with
// ceres::Solve (TrivialLoss, max 500 iterations, translation on the sphere through
// ceres::HomogeneousVectorParameterization), then ceres::Covariance of the rotation block with the translation
// held constant = (J_R^T J_R)^-1.  Ceres 1.14 pieces restated from its published sources: jet autodiff,
// HomogeneousVectorParameterization (Plus / ComputeJacobian / ComputeHouseholderVector), the LM trust-region
// loop (same control law as oracle/ref_solver.cpp), an exact dense solve of the 5x5 normal equations.
// PARITY STATUS: unpinned (no reference test, covariance_rot.txt is a missing large blob, Ceres is not buildable
// here); checked against finite differences and scipy.optimize.least_squares in tests/.
#include <cmath>
#include <cstring>
#include <limits>
#include <utility>
#include <vector>

#include "oracle_api.h"
#include "ref_rotation.hpp"

using namespace gsfm_oracle;

namespace {

struct Match { double x1, y1, x2, y2; };
struct Intr { double f1, u1, v1, f2, u2, v2; };

// MatchedFeaturesSampsonError::operator() (uncertainty.cpp:51-81)
template <typename T>
T sampson_residual(const Match& m, const Intr& K, const T* rotation, const T* t) {
  T R[9];
  AngleAxisToRotationMatrix(rotation, R);  // row-major here; the reference maps ceres' column-major output
  // inv intrinsics (uncertainty.cpp:105-118): K^-1 = [[1/f,0,-u/f],[0,1/f,-v/f],[0,0,1]]
  const double p1[3] = {(m.x1 - K.u1) / K.f1, (m.y1 - K.v1) / K.f1, 1.0};
  const double p2[3] = {(m.x2 - K.u2) / K.f2, (m.y2 - K.v2) / K.f2, 1.0};
  // F = K2^-T R [t]x K1^-1 ; epiline_x = F x1 = K2^-T R (t x p1)
  const T c[3] = {t[1] * T(p1[2]) - t[2] * T(p1[1]), t[2] * T(p1[0]) - t[0] * T(p1[2]), t[0] * T(p1[1]) - t[1] * T(p1[0])};
  T a[3];
  for (int r = 0; r < 3; ++r) a[r] = R[3 * r] * c[0] + R[3 * r + 1] * c[1] + R[3 * r + 2] * c[2];
  // K2^-T a = (a0/f2, a1/f2, -u2/f2 a0 - v2/f2 a1 + a2)
  const T e0 = a[0] / K.f2, e1 = a[1] / K.f2, e2 = a[2] - a[0] * (K.u2 / K.f2) - a[1] * (K.v2 / K.f2);
  const T numerator = T(m.x2) * e0 + T(m.y2) * e1 + e2;   // x2^T F x1
  // x2^T F = (K1^-T [t]x^T R^T p2)^T ; b = (R^T p2) x t
  T rp[3];
  for (int cidx = 0; cidx < 3; ++cidx) rp[cidx] = R[cidx] * T(p2[0]) + R[3 + cidx] * T(p2[1]) + R[6 + cidx] * T(p2[2]);
  const T b[3] = {rp[1] * t[2] - rp[2] * t[1], rp[2] * t[0] - rp[0] * t[2], rp[0] * t[1] - rp[1] * t[0]};
  const T g0 = b[0] / K.f1, g1 = b[1] / K.f1;               // feature2 . F.col(0), feature2 . F.col(1)
  const T den = g0 * g0 + g1 * g1 + e0 * e0 + e1 * e1;
  return gsfm_oracle::sqrt(numerator * numerator / den);
}

void householder(const double* x, double* v, double* beta) {  // ceres internal::ComputeHouseholderVector, size 3
  const double sigma = x[0] * x[0] + x[1] * x[1];
  v[0] = x[0]; v[1] = x[1]; v[2] = 1.0;
  *beta = 0.0;
  const double xp = x[2];
  if (sigma <= std::numeric_limits<double>::epsilon()) { if (xp < 0.0) *beta = 2.0; return; }
  const double mu = std::sqrt(xp * xp + sigma);
  const double vp = (xp <= 0.0) ? xp - mu : -sigma / (xp + mu);
  *beta = 2.0 * vp * vp / (sigma + vp * vp);
  v[0] /= vp; v[1] /= vp;
}
void hom_plus(const double* x, const double* d, double* out) {  // HomogeneousVectorParameterization::Plus
  const double nd = std::sqrt(d[0] * d[0] + d[1] * d[1]);
  if (nd == 0.0) { out[0] = x[0]; out[1] = x[1]; out[2] = x[2]; return; }
  const double h = 0.5 * nd, sbd = std::sin(h) / h;
  const double y[3] = {0.5 * sbd * d[0], 0.5 * sbd * d[1], std::cos(h)};
  double v[3], beta;
  householder(x, v, &beta);
  const double vy = v[0] * y[0] + v[1] * y[1] + v[2] * y[2];
  const double nx = std::sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
  for (int k = 0; k < 3; ++k) out[k] = nx * (y[k] - v[k] * (beta * vy));
}
void hom_jacobian(const double* x, double* J /*3x2 row-major*/) {  // ::ComputeJacobian
  double v[3], beta;
  householder(x, v, &beta);
  const double nx = std::sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
  for (int i = 0; i < 2; ++i) {
    for (int r = 0; r < 3; ++r) J[2 * r + i] = -0.5 * beta * v[i] * v[r];
    J[2 * i + i] += 0.5;
  }
  for (int k = 0; k < 6; ++k) J[k] *= nx;
}

struct PairProblem {
  const Match* m; int n; Intr K;
  // cost, residuals (n), local jacobian (n x 5, row-major) at (rot, t)
  double evaluate(const double* rot, const double* t, double* res, double* J) const {
    double cost = 0;
    double Jh[6];
    if (J) hom_jacobian(t, Jh);
    for (int k = 0; k < n; ++k) {
      if (!J) { const double r = sampson_residual<double>(m[k], K, rot, t); if (res) res[k] = r; cost += 0.5 * r * r; continue; }
      typedef Jet<6> J6;
      J6 jr[3], jt[3];
      for (int c = 0; c < 3; ++c) { jr[c] = J6(rot[c], c); jt[c] = J6(t[c], 3 + c); }
      const J6 r = sampson_residual<J6>(m[k], K, jr, jt);
      res[k] = r.a; cost += 0.5 * r.a * r.a;
      for (int c = 0; c < 3; ++c) J[5 * k + c] = r.v[c];
      for (int c = 0; c < 2; ++c) J[5 * k + 3 + c] = r.v[3] * Jh[c] + r.v[4] * Jh[2 + c] + r.v[5] * Jh[4 + c];
    }
    return cost;
  }
};

bool chol_solve5(const double* A, const double* b, double* x, int n) {  // dense SPD solve, n <= 5
  double L[25];
  for (int i = 0; i < n; ++i) for (int j = 0; j <= i; ++j) {
    double s = A[n * i + j];
    for (int k = 0; k < j; ++k) s -= L[n * i + k] * L[n * j + k];
    if (i == j) { if (!(s > 0)) return false; L[n * i + j] = std::sqrt(s); } else L[n * i + j] = s / L[n * j + j];
  }
  double y[5];
  for (int i = 0; i < n; ++i) { double s = b[i]; for (int k = 0; k < i; ++k) s -= L[n * i + k] * y[k]; y[i] = s / L[n * i + i]; }
  for (int i = n - 1; i >= 0; --i) { double s = y[i]; for (int k = i + 1; k < n; ++k) s -= L[n * k + i] * x[k]; x[i] = s / L[n * i + i]; }
  return true;
}

// ceres::Solve with default options (LM, jacobi scaling, tolerances 1e-6 / 1e-10 / 1e-8), restated as in ref_solver.cpp
int refine(const PairProblem& P, double* rot, double* t, int max_iterations, int* iters_out) {
  const int n = P.n, NP = 5;
  std::vector<double> res(n), J(5 * (size_t)n), Js(5 * (size_t)n);
  double scale[5] = {1, 1, 1, 1, 1}, g[5], radius = 1e4, decrease_factor = 2.0;
  int invalid = 0, iteration = 0;
  double x_cost = 0, gmax = 0, x_norm = 0;
  auto eval_grad_jac = [&]() {
    x_cost = P.evaluate(rot, t, res.data(), J.data());
    for (int c = 0; c < NP; ++c) { g[c] = 0; for (int k = 0; k < n; ++k) g[c] += J[5 * k + c] * res[k]; }
    if (iteration == 0) for (int c = 0; c < NP; ++c) { double s = 0; for (int k = 0; k < n; ++k) s += J[5 * k + c] * J[5 * k + c]; scale[c] = 1.0 / (1.0 + std::sqrt(s)); }
    for (int k = 0; k < n; ++k) for (int c = 0; c < NP; ++c) Js[5 * k + c] = J[5 * k + c] * scale[c];
    double ng[5] = {-g[0], -g[1], -g[2], -g[3], -g[4]}, tp[3];
    hom_plus(t, ng + 3, tp);
    gmax = 0;
    for (int c = 0; c < 3; ++c) { gmax = std::fmax(gmax, std::fabs(ng[c])); gmax = std::fmax(gmax, std::fabs(t[c] - tp[c])); }
  };
  auto norm6 = [&](const double* r, const double* tt) { double s = 0; for (int c = 0; c < 3; ++c) s += r[c] * r[c] + tt[c] * tt[c]; return std::sqrt(s); };
  x_norm = norm6(rot, t);
  eval_grad_jac();
  if (!std::isfinite(x_cost)) { *iters_out = 0; return 4; }
  if (gmax <= 1e-10) { *iters_out = 0; return 1; }
  bool last_ok = false;
  while (true) {
    if (iteration >= max_iterations) { *iters_out = iteration; return 3; }
    if (last_ok && gmax <= 1e-10) { *iters_out = iteration; return 1; }
    if (radius <= 1e-32) { *iters_out = iteration; return 4; }
    ++iteration; last_ok = false;
    double A[25], b[5], step[5];
    for (int a = 0; a < NP; ++a) { b[a] = 0; for (int k = 0; k < n; ++k) b[a] += Js[5 * k + a] * res[k];
      for (int c = 0; c < NP; ++c) { double s = 0; for (int k = 0; k < n; ++k) s += Js[5 * k + a] * Js[5 * k + c]; A[5 * a + c] = s; } }
    for (int a = 0; a < NP; ++a) A[6 * a] += std::fmin(std::fmax(A[6 * a], 1e-6), 1e32) / radius;
    bool valid = chol_solve5(A, b, step, NP);
    double model_cost_change = 0;
    if (valid) {
      for (int a = 0; a < NP; ++a) { step[a] = -step[a]; if (!std::isfinite(step[a])) valid = false; }
      for (int k = 0; k < n && valid; ++k) { double md = 0; for (int a = 0; a < NP; ++a) md += Js[5 * k + a] * step[a]; model_cost_change -= md * (res[k] + md / 2.0); }
      if (model_cost_change <= 0) valid = false;
    }
    if (!valid) { if (++invalid >= 5) { *iters_out = iteration; return 4; } radius /= decrease_factor; decrease_factor *= 2; continue; }
    invalid = 0;
    double delta[5], crot[3], ct[3];
    for (int a = 0; a < NP; ++a) delta[a] = step[a] * scale[a];
    for (int c = 0; c < 3; ++c) crot[c] = rot[c] + delta[c];
    hom_plus(t, delta + 3, ct);
    double cand = P.evaluate(crot, ct, nullptr, nullptr);
    if (!std::isfinite(cand)) cand = std::numeric_limits<double>::max();
    double sn = 0; for (int c = 0; c < 3; ++c) sn += (rot[c] - crot[c]) * (rot[c] - crot[c]) + (t[c] - ct[c]) * (t[c] - ct[c]);
    sn = std::sqrt(sn);
    const double cost_change = x_cost - cand, rel_dec = cost_change / model_cost_change;
    if (sn <= 1e-8 * (x_norm + 1e-8)) { *iters_out = iteration; return 2; }
    if (std::fabs(cost_change) <= 1e-6 * x_cost) { *iters_out = iteration; return 0; }
    if (rel_dec > 1e-3) {
      std::memcpy(rot, crot, 24); std::memcpy(t, ct, 24); x_norm = norm6(rot, t);
      eval_grad_jac();
      radius = std::fmin(1e16, radius / std::fmax(1.0 / 3.0, 1.0 - std::pow(2.0 * rel_dec - 1.0, 3)));
      decrease_factor = 2.0; last_ok = true;
    } else { radius /= decrease_factor; decrease_factor *= 2; }
  }
}

}  // namespace

// ceres::Covariance::Compute fails on a rank-deficient Jacobian (the reference then CHECK-aborts, uncertainty.cpp:157).  The 3 x 3
// restatement: Cholesky with diagonal pivoting on a copy of H, rank deficient when a pivot drops below 1e-14 of the first
// (Covariance::Options::min_reciprocal_condition_number).  Degenerate view pairs (identical or collinear matches) end here.
static bool information_is_rank_deficient(const double H[9]) {
  double A[3][3];
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) A[r][c] = H[3 * r + c];
  int idx[3] = {0, 1, 2};
  double first = 0.0;
  for (int k = 0; k < 3; ++k) {
    int best = k;
    for (int j = k + 1; j < 3; ++j) if (A[idx[j]][idx[j]] > A[idx[best]][idx[best]]) best = j;
    std::swap(idx[k], idx[best]);
    const double piv = A[idx[k]][idx[k]];
    if (k == 0) { first = piv; if (!(piv > 0.0)) return true; }
    else if (!(piv > 1e-14 * first)) return true;
    for (int i = k + 1; i < 3; ++i)
      for (int j = k + 1; j < 3; ++j) A[idx[i]][idx[j]] -= A[idx[i]][idx[k]] * A[idx[k]][idx[j]] / piv;
  }
  return false;
}

extern "C" {

// One call = store_covariance_rot's loop body for every edge (uncertainty.cpp:164-198).
// match_ptr[e]..match_ptr[e+1] index `matches` (x1 y1 x2 y2 per match, pixels); intrinsics: f1 u1 v1 f2 u2 v2 per edge.
// cov9_out: row-major 3x3 per edge; status_out: 0 ok, 1 skipped (zero translation / no matches), 2 singular.
int orc_cov_estimate(uint64_t n_edges, const uint64_t* match_ptr, const double* matches, const double* intrinsics,
                     const double* rot_in, const double* trans_in, int32_t max_iterations, double* cov9_out,
                     double* rot_out, double* trans_out, int32_t* status_out, int32_t* iters_out) {
#pragma omp parallel for schedule(dynamic, 16)
  for (long e = 0; e < (long)n_edges; ++e) {
    double rot[3] = {rot_in[3 * e], rot_in[3 * e + 1], rot_in[3 * e + 2]}, t[3] = {trans_in[3 * e], trans_in[3 * e + 1], trans_in[3 * e + 2]};
    PairProblem P;
    P.m = reinterpret_cast<const Match*>(matches + 4 * match_ptr[e]);
    P.n = (int)(match_ptr[e + 1] - match_ptr[e]);
    std::memcpy(&P.K, intrinsics + 6 * e, sizeof(Intr));
    int st = 0, it = 0;
    for (int k = 0; k < 9; ++k) cov9_out[9 * e + k] = 0.0;
    if ((t[0] == 0 && t[1] == 0 && t[2] == 0) || P.n == 0) st = 1;      // uncertainty.cpp:123
    else {
      refine(P, rot, t, max_iterations, &it);
      std::vector<double> res(P.n), J(5 * (size_t)P.n);
      P.evaluate(rot, t, res.data(), J.data());
      double H[9] = {0};
      for (int k = 0; k < P.n; ++k) for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) H[3 * a + b] += J[5 * k + a] * J[5 * k + b];
      const double c00 = H[4] * H[8] - H[5] * H[7], c01 = H[5] * H[6] - H[3] * H[8], c02 = H[3] * H[7] - H[4] * H[6];
      const double det = H[0] * c00 + H[1] * c01 + H[2] * c02;
      if (!(std::fabs(det) > 0) || !std::isfinite(det) || information_is_rank_deficient(H)) st = 2;
      else {
        double* C = cov9_out + 9 * e;
        C[0] = c00 / det; C[1] = (H[2] * H[7] - H[1] * H[8]) / det; C[2] = (H[1] * H[5] - H[2] * H[4]) / det;
        C[3] = c01 / det; C[4] = (H[0] * H[8] - H[2] * H[6]) / det; C[5] = (H[2] * H[3] - H[0] * H[5]) / det;
        C[6] = c02 / det; C[7] = (H[1] * H[6] - H[0] * H[7]) / det; C[8] = (H[0] * H[4] - H[1] * H[3]) / det;
      }
    }
    for (int c = 0; c < 3; ++c) { rot_out[3 * e + c] = rot[c]; trans_out[3 * e + c] = t[c]; }
    status_out[e] = st;
    if (iters_out) iters_out[e] = it;
  }
  return 0;
}

// single residual (+ its 1x6 ambient jacobian) for finite-difference tests
double orc_sampson_residual(const double* match4, const double* intr6, const double* rot, const double* t, double* jac6) {
  Match m; std::memcpy(&m, match4, sizeof(m)); Intr K; std::memcpy(&K, intr6, sizeof(K));
  if (!jac6) return sampson_residual<double>(m, K, rot, t);
  typedef Jet<6> J6;
  J6 jr[3], jt[3];
  for (int c = 0; c < 3; ++c) { jr[c] = J6(rot[c], c); jt[c] = J6(t[c], 3 + c); }
  const J6 r = sampson_residual<J6>(m, K, jr, jt);
  for (int c = 0; c < 6; ++c) jac6[c] = r.v[c];
  return r.a;
}
void orc_homogeneous_plus(const double* x3, const double* d2, double* out3) { hom_plus(x3, d2, out3); }
void orc_homogeneous_jacobian(const double* x3, double* J32) { hom_jacobian(x3, J32); }

}  // extern "C"

// TEST INFRASTRUCTURE — CPU oracle. Not part of the product; never linked by it.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
//
// CPU restatement of the reference's robust rotation-averaging path:
//   problem assembly ... src/GSfM_nonlinear_rotation_estimator.cpp:24-80 (a1), :82-198 (a2),
//                        :201-309 (a3), :314-457 (a4, sigma consensus)
//   residual functors .. oracle/ref_residuals.hpp (cites quat.hpp / Theia)
//   losses ............. oracle/ref_loss.hpp (cites scripts/loss_functions.py)
//   everything the reference delegates to ceres-solver 1.14.0 (NOT vendored under
//   /root/reference, pinned only in README.md:21): AutoDiffCostFunction (Jets),
//   ResidualBlock::Evaluate + Corrector, EigenQuaternionParameterization,
//   TrustRegionMinimizer + LevenbergMarquardtStrategy with Ceres' default options,
//   SPARSE_NORMAL_CHOLESKY (restated as an exact dense Cholesky of the same normal
//   equations for small graphs, and block-Jacobi PCG to 1e-14 for large ones).
//
// PARITY STATUS: the residual level is pinned by Theia's 4 known-answer tests and the
// loss level by golden vectors generated from the reference's own Python file
// (tests/golden/); the SOLVER level is "parity unpinned": Ceres/Eigen/Theia cannot be
// built in this image, so the LM trajectory is a restatement of Ceres' published
// algorithm, cross-checked against scipy.optimize.least_squares at dev time.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>
#include <vector>

#include "oracle_api.h"
#ifdef _OPENMP
#include <omp.h>
#endif
#include "ref_loss.hpp"
#include "ref_residuals.hpp"

using namespace gsfm_oracle;

namespace {

struct Edge {
  uint32_t i, j;
  double rel_aa[3];
  double rel_q[4];  // x y z w
  double W[9];
};

enum Functor { F_AA = 0, F_QCOS = 1, F_QNORM = 2, F_RFNORM = 3 };

}  // namespace

struct orc_problem {
  uint32_t n_cams = 0;
  int error_type = 0;
  Functor functor = F_AA;
  int res_dim = 3;
  int param_dim = 3;  // 3 = additive angle-axis, 4 = quaternion + EigenQuaternionParameterization
  bool scalar_weight_mode = false;
  std::vector<Edge> edges;
  std::vector<gsfm_loss_node> loss;
  bool null_loss = true;
  gsfm_loss_callback cb = nullptr;
  void* cb_user = nullptr;
  int linear_solver = 0;  // 0 auto, 1 dense cholesky, 2 pcg
  std::vector<char> active;
  // linearisation kept for orc_normal_matvec / the LM loop
  std::vector<double> rt;      // corrected residuals, res_dim per edge
  std::vector<double> Ji, Jj;  // corrected local jacobians, res_dim x 3 per edge (row-major)
  std::vector<double> trace;   // rows of ORC_TRACE_COLS
  // test hook (orc_capture_steps): the linear systems of the first LM iterations of the next solve, as the linear solver saw them
  struct Captured { std::vector<double> Ji, Jj, rt, D, rhs, y, scale; int cg = 0; };
  int capture_max = 0;
  std::vector<Captured> captured;
  // camera -> incident edges (edge id | role << 31, role 1 = the camera is `second`), in increasing edge order: the
  // per-camera sums below visit the same terms in the same order as a serial loop over the edges, but in parallel
  std::vector<uint32_t> inc_ptr, inc;
};

namespace {

void ensure_incidence(const orc_problem* cp);

// ---- whitening, src/GSfM_nonlinear_rotation_estimator.cpp:251-288 ----
void inverse3_cofactor(const double* m, double* inv) {  // Eigen fixed-size 3x3 inverse: cofactors / det
  const double c00 = m[4] * m[8] - m[5] * m[7];
  const double c10 = m[5] * m[6] - m[3] * m[8];
  const double c20 = m[3] * m[7] - m[4] * m[6];
  const double det = m[0] * c00 + m[1] * c10 + m[2] * c20;
  const double id = 1.0 / det;
  inv[0] = c00 * id; inv[3] = c10 * id; inv[6] = c20 * id;
  inv[1] = (m[2] * m[7] - m[1] * m[8]) * id;
  inv[4] = (m[0] * m[8] - m[2] * m[6]) * id;
  inv[7] = (m[1] * m[6] - m[0] * m[7]) * id;
  inv[2] = (m[1] * m[5] - m[2] * m[4]) * id;
  inv[5] = (m[2] * m[3] - m[0] * m[5]) * id;
  inv[8] = (m[0] * m[4] - m[1] * m[3]) * id;
}

void whitening(int error_type, const double* cov6, double inlier_w, double* W) {
  for (int k = 0; k < 9; ++k) W[k] = 0.0;
  auto set_scalar = [&](double w) { W[0] = W[4] = W[8] = w; };
  double cov[9] = {0};
  if (cov6) {
    // C00 C11 C22 C01 C02 C12 (src/uncertainty.cpp:219-222), scaled by 1e8 (:252)
    cov[0] = cov6[0] * 1e8; cov[4] = cov6[1] * 1e8; cov[8] = cov6[2] * 1e8;
    cov[1] = cov[3] = cov6[3] * 1e8; cov[2] = cov[6] = cov6[4] * 1e8; cov[5] = cov[7] = cov6[5] * 1e8;
  }
  switch (error_type) {
    case GSFM_ROT_ANGLE_AXIS_COVARIANCE:
    case GSFM_ROT_ANGLE_AXIS_COV_INLIERS: {
      double P[9];
      inverse3_cofactor(cov, P);
      // P = L L^T (Eigen llt, reads the lower triangle), Lt = L^T
      const double l00 = std::sqrt(P[0]);
      const double l10 = P[3] / l00, l20 = P[6] / l00;
      const double l11 = std::sqrt(P[4] - l10 * l10);
      const double l21 = (P[7] - l20 * l10) / l11;
      const double l22 = std::sqrt(P[8] - l20 * l20 - l21 * l21);
      W[0] = l00; W[1] = l10; W[2] = l20; W[4] = l11; W[5] = l21; W[8] = l22;
      if (error_type == GSFM_ROT_ANGLE_AXIS_COV_INLIERS)
        for (int k = 0; k < 9; ++k) W[k] *= inlier_w;
      break; }
    case GSFM_ROT_ANGLE_AXIS_INLIERS: set_scalar(inlier_w); break;
    case GSFM_ROT_ANGLE_AXIS_COVTRACE: set_scalar(std::sqrt(1.0 / (cov[0] + cov[4] + cov[8]))); break;
    case GSFM_ROT_ANGLE_AXIS_COVNORM: {
      double f = 0; for (int k = 0; k < 9; ++k) f += cov[k] * cov[k];
      set_scalar(std::sqrt(1.0 / std::sqrt(f))); break; }
    default: set_scalar(1.0);
  }
}

// ---- EigenQuaternionParameterization (ceres 1.14 local_parameterization.cc) ----
void quat_plus(const double* x, const double* d, double* out) {
  const double nd = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
  if (nd > 0.0) {
    const double k = std::sin(nd) / nd;
    QuatXYZW<double> dq{k * d[0], k * d[1], k * d[2], std::cos(nd)};
    QuatXYZW<double> q{x[0], x[1], x[2], x[3]};
    QuatXYZW<double> r = QuatMul(dq, q);
    out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = r.w;
  } else { for (int k = 0; k < 4; ++k) out[k] = x[k]; }
}
void quat_plus_jacobian(const double* x, double* J /*4x3 row-major*/) {
  J[0] = x[3];  J[1] = x[2];   J[2] = -x[1];
  J[3] = -x[2]; J[4] = x[3];   J[5] = x[0];
  J[6] = x[1];  J[7] = -x[0];  J[8] = x[3];
  J[9] = -x[0]; J[10] = -x[1]; J[11] = -x[2];
}

void loss_eval(const orc_problem* p, double s, double* rho) {
  if (p->cb) { p->cb(p->cb_user, s, rho); return; }
  Rho r = eval_loss_program(p->loss.data(), (int)p->loss.size(), s);
  rho[0] = r.v[0]; rho[1] = r.v[1]; rho[2] = r.v[2];
}

// raw residual + local jacobians of one edge (AutoDiffCostFunction restated)
void edge_autodiff(const orc_problem* p, const Edge& e, const double* x, double* r, double* Ji, double* Jj) {
  const int R = p->res_dim;
  if (p->functor == F_AA) {
    AngleAxisError f; std::memcpy(f.rel_aa, e.rel_aa, 24); std::memcpy(f.W, e.W, 72);
    if (!Ji) { f(x + 3 * e.i, x + 3 * e.j, r); return; }
    typedef Jet<6> J6;
    J6 a[3], b[3], out[3];
    for (int k = 0; k < 3; ++k) { a[k] = J6(x[3 * e.i + k], k); b[k] = J6(x[3 * e.j + k], 3 + k); }
    f(a, b, out);
    for (int k = 0; k < 3; ++k) {
      r[k] = out[k].a;
      for (int c = 0; c < 3; ++c) { Ji[3 * k + c] = out[k].v[c]; Jj[3 * k + c] = out[k].v[3 + c]; }
    }
    return;
  }
  const double* xa = x + 4 * e.i; const double* xb = x + 4 * e.j;
  typedef Jet<8> J8;
  J8 a[4], b[4], out[9];
  double rd[9];
  auto run = [&](auto& f) {
    if (!Ji) { f(xa, xb, rd); for (int k = 0; k < R; ++k) r[k] = rd[k]; return; }
    for (int k = 0; k < 4; ++k) { a[k] = J8(xa[k], k); b[k] = J8(xb[k], 4 + k); }
    f(a, b, out);
    double Pa[12], Pb[12];
    quat_plus_jacobian(xa, Pa); quat_plus_jacobian(xb, Pb);
    for (int k = 0; k < R; ++k) {
      r[k] = out[k].a;
      for (int c = 0; c < 3; ++c) {
        double si = 0, sj = 0;
        for (int m = 0; m < 4; ++m) { si += out[k].v[m] * Pa[3 * m + c]; sj += out[k].v[4 + m] * Pb[3 * m + c]; }
        Ji[3 * k + c] = si; Jj[3 * k + c] = sj;
      }
    }
  };
  if (p->functor == F_QCOS) { QuatCosineError f; std::memcpy(f.rel, e.rel_q, 32); f.weight = e.W[0]; run(f); }
  else if (p->functor == F_QNORM) { QuatNormError f; std::memcpy(f.rel, e.rel_q, 32); f.weight = e.W[0]; run(f); }
  else { RotFNormError f; std::memcpy(f.rel, e.rel_q, 32); f.weight = e.W[0]; run(f); }
}

// ResidualBlock::Evaluate + Corrector (ceres 1.14 residual_block.cc / corrector.cc)
// returns 1/2 rho; corrects r, Ji, Jj in place when jacobians requested.
double robustify(const orc_problem* p, int R, double* r, double* Ji, double* Jj, double* s_out, double* rho_out) {
  double s = 0; for (int k = 0; k < R; ++k) s += r[k] * r[k];
  if (s_out) *s_out = s;
  if (p->null_loss && !p->cb) {
    if (rho_out) { rho_out[0] = s; rho_out[1] = 1.0; rho_out[2] = 0.0; }
    return 0.5 * s;
  }
  double rho[3];
  loss_eval(p, s, rho);
  if (rho_out) { rho_out[0] = rho[0]; rho_out[1] = rho[1]; rho_out[2] = rho[2]; }
  if (!Ji) return 0.5 * rho[0];
  const double sqrt_rho1 = std::sqrt(rho[1]);
  double residual_scaling, alpha_sq_norm;
  if (s == 0.0 || rho[2] <= 0.0) { residual_scaling = sqrt_rho1; alpha_sq_norm = 0.0; }
  else {
    const double D = 1.0 + 2.0 * s * rho[2] / rho[1];
    const double alpha = 1.0 - std::sqrt(D);
    residual_scaling = sqrt_rho1 / (1 - alpha);
    alpha_sq_norm = alpha / s;
  }
  for (double* J : {Ji, Jj}) {
    if (alpha_sq_norm == 0.0) { for (int k = 0; k < 3 * R; ++k) J[k] *= sqrt_rho1; continue; }
    for (int c = 0; c < 3; ++c) {
      double rtj = 0; for (int k = 0; k < R; ++k) rtj += J[3 * k + c] * r[k];
      for (int k = 0; k < R; ++k) J[3 * k + c] = sqrt_rho1 * (J[3 * k + c] - alpha_sq_norm * r[k] * rtj);
    }
  }
  for (int k = 0; k < R; ++k) r[k] *= residual_scaling;
  return 0.5 * rho[0];
}

// Evaluator::Evaluate: cost [+ corrected residuals/jacobians kept in p + gradient]
double evaluate(orc_problem* p, const double* x, bool with_jacobian, double* gradient) {
  const int R = p->res_dim;
  const size_t E = p->edges.size();
  if (with_jacobian) { p->rt.assign(E * R, 0.0); p->Ji.assign(E * R * 3, 0.0); p->Jj.assign(E * R * 3, 0.0); }
  // 80-bit accumulator: at 1e8 edges a plain double running sum per thread is only good to ~1e-9 relative, which made the
  // CHECKER the less accurate side (the device's fixed-tree sums equal math.fsum of the per-edge values; tests/manual/cost_sum_probe.py)
  long double cost = 0.0L;
  const bool serial = (p->cb != nullptr);
  if (serial) {
    for (size_t e = 0; e < E; ++e) {
      double r[9], ji[27], jj[27];
      edge_autodiff(p, p->edges[e], x, r, with_jacobian ? ji : nullptr, with_jacobian ? jj : nullptr);
      cost += robustify(p, R, r, with_jacobian ? ji : nullptr, with_jacobian ? jj : nullptr, nullptr, nullptr);
      if (with_jacobian) {
        std::memcpy(&p->rt[e * R], r, 8 * R); std::memcpy(&p->Ji[e * R * 3], ji, 24 * R); std::memcpy(&p->Jj[e * R * 3], jj, 24 * R);
      }
    }
  } else {
#pragma omp parallel for reduction(+ : cost) schedule(static) if (E > 50000)
    for (long e = 0; e < (long)E; ++e) {
      double r[9], ji[27], jj[27];
      edge_autodiff(p, p->edges[e], x, r, with_jacobian ? ji : nullptr, with_jacobian ? jj : nullptr);
      cost += robustify(p, R, r, with_jacobian ? ji : nullptr, with_jacobian ? jj : nullptr, nullptr, nullptr);
      if (with_jacobian) {
        std::memcpy(&p->rt[e * R], r, 8 * R); std::memcpy(&p->Ji[e * R * 3], ji, 24 * R); std::memcpy(&p->Jj[e * R * 3], jj, 24 * R);
      }
    }
  }
  if (with_jacobian && gradient) {
    ensure_incidence(p);
#pragma omp parallel for schedule(dynamic, 64) if (E > 50000)
    for (long cam = 0; cam < (long)p->n_cams; ++cam) {
      double g[3] = {0, 0, 0};
      for (uint32_t d = p->inc_ptr[cam]; d < p->inc_ptr[cam + 1]; ++d) {
        const size_t e = p->inc[d] & 0x7fffffffu;
        const double* J = (p->inc[d] >> 31) ? &p->Jj[e * R * 3] : &p->Ji[e * R * 3];
        for (int c = 0; c < 3; ++c) {
          double a = 0;
          for (int k = 0; k < R; ++k) a += J[3 * k + c] * p->rt[e * R + k];
          g[c] += a;
        }
      }
      gradient[3 * cam] = g[0]; gradient[3 * cam + 1] = g[1]; gradient[3 * cam + 2] = g[2];
    }
  }
  return (double)cost;
}

void ensure_incidence(const orc_problem* cp) {
  orc_problem* p = const_cast<orc_problem*>(cp);
  if (!p->inc_ptr.empty()) return;
  const size_t N = p->n_cams, E = p->edges.size();
  p->inc_ptr.assign(N + 1, 0);
  for (size_t e = 0; e < E; ++e) { p->inc_ptr[p->edges[e].i + 1]++; p->inc_ptr[p->edges[e].j + 1]++; }
  for (size_t c = 0; c < N; ++c) p->inc_ptr[c + 1] += p->inc_ptr[c];
  p->inc.resize(2 * E);
  std::vector<uint32_t> fill(p->inc_ptr.begin(), p->inc_ptr.end() - 1);
  for (size_t e = 0; e < E; ++e) { p->inc[fill[p->edges[e].i]++] = (uint32_t)e; p->inc[fill[p->edges[e].j]++] = (uint32_t)e | 0x80000000u; }
}

void squared_column_norm(const orc_problem* p, double* d) {
  const int R = p->res_dim;
  ensure_incidence(p);
#pragma omp parallel for schedule(dynamic, 64) if (p->edges.size() > 50000)
  for (long cam = 0; cam < (long)p->n_cams; ++cam) {
    double acc[3] = {0, 0, 0};
    for (uint32_t q = p->inc_ptr[cam]; q < p->inc_ptr[cam + 1]; ++q) {
      const size_t e = p->inc[q] & 0x7fffffffu;
      const double* J = (p->inc[q] >> 31) ? &p->Jj[e * R * 3] : &p->Ji[e * R * 3];
      for (int k = 0; k < R; ++k) for (int c = 0; c < 3; ++c) acc[c] += J[3 * k + c] * J[3 * k + c];
    }
    d[3 * cam] = acc[0]; d[3 * cam + 1] = acc[1]; d[3 * cam + 2] = acc[2];
  }
}
void scale_columns(orc_problem* p, const double* scale) {
  const int R = p->res_dim;
#pragma omp parallel for schedule(static) if (p->edges.size() > 50000)
  for (long e = 0; e < (long)p->edges.size(); ++e) {
    const Edge& ed = p->edges[e];
    for (int k = 0; k < R; ++k) for (int c = 0; c < 3; ++c) {
      p->Ji[(e * R + k) * 3 + c] *= scale[3 * ed.i + c];
      p->Jj[(e * R + k) * 3 + c] *= scale[3 * ed.j + c];
    }
  }
}
// y = J v (per edge, res_dim), then optionally z = J^T y
void J_times(const orc_problem* p, const double* v, double* y) {
  const int R = p->res_dim;
#pragma omp parallel for schedule(static) if (p->edges.size() > 50000)
  for (long e = 0; e < (long)p->edges.size(); ++e) {
    const Edge& ed = p->edges[e];
    for (int k = 0; k < R; ++k) {
      double a = 0;
      for (int c = 0; c < 3; ++c) a += p->Ji[(e * R + k) * 3 + c] * v[3 * ed.i + c] + p->Jj[(e * R + k) * 3 + c] * v[3 * ed.j + c];
      y[e * R + k] = a;
    }
  }
}
void Jt_times(const orc_problem* p, const double* y, double* z) {
  const int R = p->res_dim;
  ensure_incidence(p);
#pragma omp parallel for schedule(dynamic, 64) if (p->edges.size() > 50000)
  for (long cam = 0; cam < (long)p->n_cams; ++cam) {
    double acc[3] = {0, 0, 0};
    for (uint32_t q = p->inc_ptr[cam]; q < p->inc_ptr[cam + 1]; ++q) {
      const size_t e = p->inc[q] & 0x7fffffffu;
      const double* J = (p->inc[q] >> 31) ? &p->Jj[e * R * 3] : &p->Ji[e * R * 3];
      for (int c = 0; c < 3; ++c) {
        double a = 0;
        for (int k = 0; k < R; ++k) a += J[3 * k + c] * y[e * R + k];
        acc[c] += a;
      }
    }
    z[3 * cam] = acc[0]; z[3 * cam + 1] = acc[1]; z[3 * cam + 2] = acc[2];
  }
}

// ---- linear solvers for (J^T J + D^2) y = J^T r ----
bool solve_dense(const orc_problem* p, const double* D, const double* rhs, double* y) {
  const int R = p->res_dim;
  const size_t n = 3 * (size_t)p->n_cams;
  std::vector<double> A(n * n, 0.0);
  for (size_t e = 0; e < p->edges.size(); ++e) {
    const Edge& ed = p->edges[e];
    const double* Ji = &p->Ji[e * R * 3]; const double* Jj = &p->Jj[e * R * 3];
    for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) {
      double ii = 0, jj = 0, ij = 0;
      for (int k = 0; k < R; ++k) { ii += Ji[3 * k + a] * Ji[3 * k + b]; jj += Jj[3 * k + a] * Jj[3 * k + b]; ij += Ji[3 * k + a] * Jj[3 * k + b]; }
      A[(3 * ed.i + a) * n + 3 * ed.i + b] += ii;
      A[(3 * ed.j + a) * n + 3 * ed.j + b] += jj;
      A[(3 * ed.i + a) * n + 3 * ed.j + b] += ij;
      A[(3 * ed.j + b) * n + 3 * ed.i + a] += ij;
    }
  }
  for (size_t k = 0; k < n; ++k) A[k * n + k] += D[k] * D[k];
  // in-place lower Cholesky (row-oriented, inner loops contiguous)
  for (size_t i = 0; i < n; ++i) {
    double* Ai = &A[i * n];
    for (size_t j = 0; j <= i; ++j) {
      const double* Aj = &A[j * n];
      double s = Ai[j];
      for (size_t k = 0; k < j; ++k) s -= Ai[k] * Aj[k];
      if (i == j) { if (!(s > 0.0)) return false; Ai[j] = std::sqrt(s); }
      else Ai[j] = s / Aj[j];
    }
  }
  for (size_t i = 0; i < n; ++i) { double s = rhs[i]; for (size_t k = 0; k < i; ++k) s -= A[i * n + k] * y[k]; y[i] = s / A[i * n + i]; }
  for (size_t ii = n; ii-- > 0;) { double s = y[ii]; for (size_t k = ii + 1; k < n; ++k) s -= A[k * n + ii] * y[k]; y[ii] = s / A[ii * n + ii]; }
  return true;
}

bool solve_pcg(const orc_problem* p, const double* D, const double* rhs, double* y, int* iters_out) {
  const int R = p->res_dim;
  const size_t N = p->n_cams, n = 3 * N, E = p->edges.size();
  std::vector<double> M(9 * N, 0.0), Minv(9 * N), r(rhs, rhs + n), z(n), pv(n), Ap(n), tmp(E * R);
  ensure_incidence(p);
#pragma omp parallel for schedule(dynamic, 64) if (E > 50000)
  for (long cam = 0; cam < (long)N; ++cam) {
    for (uint32_t q = p->inc_ptr[cam]; q < p->inc_ptr[cam + 1]; ++q) {
      const size_t e = p->inc[q] & 0x7fffffffu;
      const double* J = (p->inc[q] >> 31) ? &p->Jj[e * R * 3] : &p->Ji[e * R * 3];
      for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) {
        double t = 0;
        for (int k = 0; k < R; ++k) t += J[3 * k + a] * J[3 * k + b];
        M[9 * cam + 3 * a + b] += t;
      }
    }
  }
  for (size_t c = 0; c < N; ++c) {
    for (int a = 0; a < 3; ++a) M[9 * c + 4 * a] += D[3 * c + a] * D[3 * c + a];
    inverse3_cofactor(&M[9 * c], &Minv[9 * c]);
  }
  auto precond = [&](const std::vector<double>& in, std::vector<double>& out) {
    for (size_t c = 0; c < N; ++c) for (int a = 0; a < 3; ++a)
      out[3 * c + a] = Minv[9 * c + 3 * a] * in[3 * c] + Minv[9 * c + 3 * a + 1] * in[3 * c + 1] + Minv[9 * c + 3 * a + 2] * in[3 * c + 2];
  };
  auto dot = [&](const std::vector<double>& a, const std::vector<double>& b) { double s = 0; for (size_t k = 0; k < n; ++k) s += a[k] * b[k]; return s; };
  std::fill(y, y + n, 0.0);
  precond(r, z); pv = z;
  double rz = dot(r, z);
  const double rz0 = rz;
  int it = 0;
  if (rz0 > 0) for (; it < 20000; ++it) {
    J_times(p, pv.data(), tmp.data()); Jt_times(p, tmp.data(), Ap.data());
    for (size_t k = 0; k < n; ++k) Ap[k] += D[k] * D[k] * pv[k];
    const double alpha = rz / dot(pv, Ap);
    for (size_t k = 0; k < n; ++k) { y[k] += alpha * pv[k]; r[k] -= alpha * Ap[k]; }
    precond(r, z);
    const double rz_new = dot(r, z);
    if (std::sqrt(rz_new / rz0) <= 1e-14) { ++it; break; }
    const double beta = rz_new / rz; rz = rz_new;
    for (size_t k = 0; k < n; ++k) pv[k] = z[k] + beta * pv[k];
  }
  if (iters_out) *iters_out = it;
  return true;
}

void state_from_aa(const orc_problem* p, const double* aa, std::vector<double>& x) {
  const size_t N = p->n_cams;
  if (p->param_dim == 3) { x.assign(aa, aa + 3 * N); return; }
  x.resize(4 * N);
  for (size_t c = 0; c < N; ++c) {  // estimator.cpp:130-136: ceres (w,x,y,z) -> Eigen coeffs (x,y,z,w)
    double q[4]; AngleAxisToQuaternion(aa + 3 * c, q);
    x[4 * c] = q[1]; x[4 * c + 1] = q[2]; x[4 * c + 2] = q[3]; x[4 * c + 3] = q[0];
  }
}
void state_to_aa(const orc_problem* p, const std::vector<double>& x, double* aa) {
  const size_t N = p->n_cams;
  if (p->param_dim == 3) { std::memcpy(aa, x.data(), 24 * N); return; }
  for (size_t c = 0; c < N; ++c) {  // estimator.cpp:185-194 (only views touched by an edge)
    if (!p->active[c]) continue;
    double q[4] = {x[4 * c + 3], x[4 * c], x[4 * c + 1], x[4 * c + 2]};
    QuaternionToAngleAxis(q, aa + 3 * c);
  }
}
void plus(const orc_problem* p, const std::vector<double>& x, const double* delta, std::vector<double>& out) {
  const size_t N = p->n_cams;
  out.resize(x.size());
  if (p->param_dim == 3) { for (size_t k = 0; k < 3 * N; ++k) out[k] = x[k] + delta[k]; return; }
  for (size_t c = 0; c < N; ++c) quat_plus(&x[4 * c], delta + 3 * c, &out[4 * c]);
}
double active_norm(const orc_problem* p, const std::vector<double>& x) {
  double s = 0; const int d = p->param_dim;
  for (size_t c = 0; c < p->n_cams; ++c) if (p->active[c]) for (int k = 0; k < d; ++k) s += x[d * c + k] * x[d * c + k];
  return std::sqrt(s);
}

double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// TrustRegionMinimizer::Minimize + LevenbergMarquardtStrategy (ceres 1.14), monotonic steps,
// no inner iterations, no bounds.
int lm_solve(orc_problem* p, std::vector<double>& x, const gsfm_rot_options& o, gsfm_rot_summary* sum) {
  const double t0 = now_ms();
  const size_t N = p->n_cams, n = 3 * N;
  std::memset(sum, 0, sizeof(*sum));
  sum->iters_to_1e6 = -1;
  sum->num_edges_used = p->edges.size();
  p->trace.clear();
  std::vector<double> g(n), scale(n, 1.0), diag(n), D(n), step(n), delta(n), rhs(n), cand, model((size_t)p->edges.size() * p->res_dim), negg(n), tmpx;
  double radius = o.initial_trust_region_radius, decrease_factor = 2.0;
  int num_invalid = 0;
  double x_cost = 0, x_norm = 0, gmax = 0;
  int iteration = 0;

  auto eval_grad_jac = [&]() {
    x_cost = evaluate(p, x.data(), true, g.data());
    sum->num_residual_sweeps++; sum->num_linearizations++;
    if (o.jacobi_scaling) {
      if (iteration == 0) { squared_column_norm(p, scale.data()); for (size_t k = 0; k < n; ++k) scale[k] = 1.0 / (1.0 + std::sqrt(scale[k])); }
      scale_columns(p, scale.data());
    }
    // gradient_max_norm = || x - Plus(x, -g) ||_inf
    for (size_t k = 0; k < n; ++k) negg[k] = -g[k];
    plus(p, x, negg.data(), tmpx);
    gmax = 0; for (size_t k = 0; k < x.size(); ++k) gmax = std::fmax(gmax, std::fabs(x[k] - tmpx[k]));
  };
  auto record = [&](double cost, double cost_change, double step_norm, double rel_dec, int cg) {
    const double row[ORC_TRACE_COLS] = {(double)iteration, cost, cost_change, gmax, step_norm, rel_dec, radius, (double)cg};
    p->trace.insert(p->trace.end(), row, row + ORC_TRACE_COLS);
    if (o.verbose) fprintf(stderr, "[oracle] it %3d cost %.12e dcost %.3e |g| %.3e |dx| %.3e rho %.3e radius %.3e cg %d\n",
                           iteration, cost, cost_change, gmax, step_norm, rel_dec, radius, cg);
  };
  auto finish = [&](int term) {
    sum->termination = term; sum->num_iterations = iteration; sum->final_cost = x_cost;
    sum->final_gradient_max_norm = gmax; sum->final_radius = radius; sum->t_total_ms = now_ms() - t0;
    if (!std::isfinite(x_cost)) sum->nonfinite = 1;
    return 0;
  };

  x_norm = active_norm(p, x);
  eval_grad_jac();
  sum->initial_cost = x_cost;
  record(x_cost, 0, 0, 0, 0);
  if (!std::isfinite(x_cost)) return finish(GSFM_TERM_FAILURE);
  if (gmax <= o.gradient_tolerance) return finish(GSFM_TERM_GRADIENT_TOLERANCE);
  bool last_successful = false;
  while (true) {
    // FinalizeIterationAndCheckIfMinimizerCanContinue
    if (iteration >= o.max_num_iterations) return finish(GSFM_TERM_NO_CONVERGENCE);
    if (last_successful && gmax <= o.gradient_tolerance) return finish(GSFM_TERM_GRADIENT_TOLERANCE);
    if (radius <= o.min_trust_region_radius) return finish(GSFM_TERM_FAILURE);
    ++iteration;
    last_successful = false;
    // LevenbergMarquardtStrategy::ComputeStep
    squared_column_norm(p, diag.data());
    for (size_t k = 0; k < n; ++k) D[k] = std::sqrt(std::fmin(std::fmax(diag[k], o.min_lm_diagonal), o.max_lm_diagonal) / radius);
    Jt_times(p, p->rt.data(), rhs.data());
    int cg = 0;
    // 'auto': exact Cholesky steps -- what the reference's SPARSE_NORMAL_CHOLESKY computes -- wherever a dense factorisation is affordable
    // for a test (up to 512 cameras), PCG(1e-14) beyond.  A fixed rule of the ORACLE: it does not read the product's
    // dense_cholesky_max_cams option (round-2 advisor: mirroring it made the checker follow the thing it checks); tests that want a
    // particular solver on the oracle side say so with orc_set_linear_solver.
    const bool dense = p->linear_solver == 1 || (p->linear_solver == 0 && N <= 512);
    bool ok = dense ? solve_dense(p, D.data(), rhs.data(), step.data()) : solve_pcg(p, D.data(), rhs.data(), step.data(), &cg);
    sum->num_cg_iterations += cg;
    if ((int)p->captured.size() < p->capture_max) {   // (J^T J + diag(D)^2) y = rhs with J = [Ji Jj] per edge (column-scaled), before the sign flip
      orc_problem::Captured c;
      c.Ji = p->Ji; c.Jj = p->Jj; c.rt = p->rt; c.D = D; c.rhs = rhs; c.y = step; c.scale = scale; c.cg = cg;
      p->captured.push_back(std::move(c));
    }
    bool valid = ok;
    for (size_t k = 0; k < n && valid; ++k) if (!std::isfinite(step[k])) valid = false;
    double model_cost_change = 0;
    if (valid) {
      for (size_t k = 0; k < n; ++k) step[k] = -step[k];
      J_times(p, step.data(), model.data());
      for (size_t k = 0; k < model.size(); ++k) model_cost_change -= model[k] * (p->rt[k] + model[k] / 2.0);
      if (model_cost_change <= 0) valid = false;
    }
    if (!valid) {  // HandleInvalidStep
      if (++num_invalid >= 5) return finish(GSFM_TERM_FAILURE);
      radius /= decrease_factor; decrease_factor *= 2.0;
      sum->num_unsuccessful_steps++;
      record(x_cost, 0, 0, 0, cg);
      continue;
    }
    num_invalid = 0;
    for (size_t k = 0; k < n; ++k) delta[k] = step[k] * scale[k];
    plus(p, x, delta.data(), cand);
    double cand_cost = evaluate(p, cand.data(), false, nullptr);
    sum->num_residual_sweeps++;
    if (!std::isfinite(cand_cost)) { cand_cost = std::numeric_limits<double>::max(); sum->nonfinite = 1; }
    double step_norm = 0; for (size_t k = 0; k < x.size(); ++k) step_norm += (x[k] - cand[k]) * (x[k] - cand[k]);
    step_norm = std::sqrt(step_norm);
    const double cost_change = x_cost - cand_cost;
    const double rel_dec = cost_change / model_cost_change;
    if (sum->iters_to_1e6 < 0 && std::fabs(cost_change) <= 1e-6 * x_cost) sum->iters_to_1e6 = iteration;
    if (step_norm <= o.parameter_tolerance * (x_norm + o.parameter_tolerance)) { record(x_cost, cost_change, step_norm, rel_dec, cg); return finish(GSFM_TERM_PARAMETER_TOLERANCE); }
    if (std::fabs(cost_change) <= o.function_tolerance * x_cost) { record(x_cost, cost_change, step_norm, rel_dec, cg); return finish(GSFM_TERM_FUNCTION_TOLERANCE); }
    if (rel_dec > o.min_relative_decrease) {  // HandleSuccessfulStep
      x = cand; x_norm = active_norm(p, x);
      eval_grad_jac();
      radius = radius / std::fmax(1.0 / 3.0, 1.0 - std::pow(2.0 * rel_dec - 1.0, 3));
      radius = std::fmin(o.max_trust_region_radius, radius);
      decrease_factor = 2.0;
      sum->num_successful_steps++;
      last_successful = true;
    } else {  // HandleUnsuccessfulStep
      radius /= decrease_factor; decrease_factor *= 2.0;
      sum->num_unsuccessful_steps++;
    }
    record(x_cost, cost_change, step_norm, rel_dec, cg);
  }
}

gsfm_rot_options defaults() { gsfm_rot_options o; orc_options_default(&o); return o; }

}  // namespace

extern "C" {

int orc_set_num_threads(int n) {  // for the 1-thread / all-cores CPU baselines; returns the previous maximum
#ifdef _OPENMP
  const int prev = omp_get_max_threads();
  if (n > 0) omp_set_num_threads(n);
  return prev;
#else
  (void)n; return 1;
#endif
}

void orc_options_default(gsfm_rot_options* o) {
  std::memset(o, 0, sizeof(*o));
  o->max_num_iterations = 200; o->num_threads = 1;
  o->function_tolerance = 1e-6; o->gradient_tolerance = 1e-10; o->parameter_tolerance = 1e-8;
  o->initial_trust_region_radius = 1e4; o->max_trust_region_radius = 1e16; o->min_trust_region_radius = 1e-32;
  o->min_relative_decrease = 1e-3; o->min_lm_diagonal = 1e-6; o->max_lm_diagonal = 1e32;
  o->jacobi_scaling = 1; o->max_cg_iterations = 20000; o->cg_relative_tolerance = 1e-12; o->cg_check_interval = 8; o->verbose = 0; o->pcg_single_reduction = -1; o->cg_stall_iterations = 0; o->dense_cholesky_max_cams = 512; o->pcg_hip_graph = 1;
  o->pcg_forcing = 1; o->pcg_forcing_tolerance = 1e-8; o->dense_cholesky_auto_cams = 5333; o->lm_device_control = 1; o->component_rest = 1;   // (the oracle never solves loosely: accepted for struct parity, ignored)
}

int32_t orc_residual_dim(int32_t t) { return t == GSFM_ROT_QUATERNION_NORM ? 4 : t == GSFM_ROT_ROTATION_MAT_FNORM ? 9 : 3; }

orc_problem* orc_problem_create(uint32_t n_cams, uint64_t n_edges, const uint32_t* ei, const uint32_t* ej,
                                const double* rel_aa, int32_t error_type, const double* cov6, const double* inlier_weight) {
  if (error_type < 0 || error_type > 8) return nullptr;
  const bool need_cov = error_type == GSFM_ROT_ANGLE_AXIS_COVARIANCE || error_type == GSFM_ROT_ANGLE_AXIS_COV_INLIERS ||
                        error_type == GSFM_ROT_ANGLE_AXIS_COVTRACE || error_type == GSFM_ROT_ANGLE_AXIS_COVNORM;
  const bool need_inl = error_type == GSFM_ROT_ANGLE_AXIS_INLIERS || error_type == GSFM_ROT_ANGLE_AXIS_COV_INLIERS;
  if ((need_cov && !cov6) || (need_inl && !inlier_weight)) return nullptr;
  orc_problem* p = new orc_problem;
  p->n_cams = n_cams; p->error_type = error_type;
  p->functor = error_type == GSFM_ROT_QUATERNION_COSINE ? F_QCOS : error_type == GSFM_ROT_QUATERNION_NORM ? F_QNORM
               : error_type == GSFM_ROT_ROTATION_MAT_FNORM ? F_RFNORM : F_AA;
  p->res_dim = orc_residual_dim(error_type);
  p->param_dim = p->functor == F_AA ? 3 : 4;
  p->scalar_weight_mode = p->functor == F_AA && error_type != GSFM_ROT_ANGLE_AXIS_COVARIANCE && error_type != GSFM_ROT_ANGLE_AXIS_COV_INLIERS;
  p->active.assign(n_cams, 0);
  p->edges.resize(n_edges);
  for (uint64_t e = 0; e < n_edges; ++e) {
    Edge& ed = p->edges[e];
    ed.i = ei[e]; ed.j = ej[e];
    if (ed.i >= n_cams || ed.j >= n_cams || ed.i == ed.j) { delete p; return nullptr; }
    p->active[ed.i] = p->active[ed.j] = 1;
    std::memcpy(ed.rel_aa, rel_aa + 3 * e, 24);
    double q[4]; AngleAxisToQuaternion(ed.rel_aa, q);  // estimator.cpp:132,136
    ed.rel_q[0] = q[1]; ed.rel_q[1] = q[2]; ed.rel_q[2] = q[3]; ed.rel_q[3] = q[0];
    if (p->functor == F_AA) whitening(error_type, cov6 ? cov6 + 6 * e : nullptr, inlier_weight ? inlier_weight[e] : 1.0, ed.W);
    else { for (int k = 0; k < 9; ++k) ed.W[k] = 0; ed.W[0] = ed.W[4] = ed.W[8] = 1.0; }  // cost_weight = 1.0 (:123)
  }
  return p;
}
void orc_problem_destroy(orc_problem* p) { delete p; }

int orc_set_loss(orc_problem* p, const gsfm_loss_node* prog, int32_t n) {
  if (n < 0 || n > GSFM_LOSS_MAX_NODES) return 1;
  p->loss.assign(prog, prog + n); p->null_loss = (n == 0); p->cb = nullptr; return 0;
}
int orc_set_loss_callback(orc_problem* p, gsfm_loss_callback fn, void* user) { p->cb = fn; p->cb_user = user; return 0; }
int orc_set_linear_solver(orc_problem* p, int32_t kind) { p->linear_solver = kind; return 0; }
int orc_set_edge_weights(orc_problem* p, const double* w) {
  if (!p->scalar_weight_mode) return 1;
  for (size_t e = 0; e < p->edges.size(); ++e) { double* W = p->edges[e].W; W[0] = W[4] = W[8] = w[e]; }
  return 0;
}

int orc_residuals(orc_problem* p, const double* rot_aa, double* s_out, double* rho_out, double* residual_out, double* cost) {
  std::vector<double> x; state_from_aa(p, rot_aa, x);
  const int R = p->res_dim;
  long double c = 0.0L;  // see evaluate(): the checker's sum must not be the less accurate one
  const long E = (long)p->edges.size();
#pragma omp parallel for reduction(+ : c) schedule(static) if (E > 50000 && p->cb == nullptr)
  for (long e = 0; e < E; ++e) {
    double r[9], s, rho[3];
    edge_autodiff(p, p->edges[e], x.data(), r, nullptr, nullptr);
    c += robustify(p, R, r, nullptr, nullptr, &s, rho);
    if (s_out) s_out[e] = s;
    if (rho_out) std::memcpy(rho_out + 3 * (size_t)e, rho, 24);
    if (residual_out) std::memcpy(residual_out + (size_t)R * e, r, 8 * R);
  }
  if (cost) *cost = (double)c;
  return 0;
}

int orc_linearize(orc_problem* p, const double* rot_aa, double* gradient, double* diag_blocks, double* cost) {
  std::vector<double> x; state_from_aa(p, rot_aa, x);
  std::vector<double> g(3 * (size_t)p->n_cams);
  const double c = evaluate(p, x.data(), true, g.data());
  if (gradient) std::memcpy(gradient, g.data(), 8 * g.size());
  if (cost) *cost = c;
  if (diag_blocks) {
    const int R = p->res_dim;
    std::fill(diag_blocks, diag_blocks + 9 * (size_t)p->n_cams, 0.0);
    for (size_t e = 0; e < p->edges.size(); ++e) {
      const Edge& ed = p->edges[e];
      for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) {
        double ii = 0, jj = 0;
        for (int k = 0; k < R; ++k) { ii += p->Ji[(e * R + k) * 3 + a] * p->Ji[(e * R + k) * 3 + b]; jj += p->Jj[(e * R + k) * 3 + a] * p->Jj[(e * R + k) * 3 + b]; }
        diag_blocks[9 * ed.i + 3 * a + b] += ii; diag_blocks[9 * ed.j + 3 * a + b] += jj;
      }
    }
  }
  return 0;
}
int orc_normal_matvec(orc_problem* p, const double* v, double* y) {
  if (p->rt.empty()) return 1;
  std::vector<double> t(p->edges.size() * p->res_dim);
  J_times(p, v, t.data()); Jt_times(p, t.data(), y);
  return 0;
}

int orc_solve(orc_problem* p, double* rot_aa_inout, const gsfm_rot_options* opt, gsfm_rot_summary* summary) {
  gsfm_rot_options o = opt ? *opt : defaults();
  gsfm_rot_summary local; if (!summary) summary = &local;
  if (p->n_cams == 0 || p->edges.empty()) return GSFM_ERR_EMPTY;
  std::vector<double> x; state_from_aa(p, rot_aa_inout, x);
  lm_solve(p, x, o, summary);
  state_to_aa(p, x, rot_aa_inout);
  return 0;
}

// EstimateRotationsWithSigmaConsensus, src/GSfM_nonlinear_rotation_estimator.cpp:314-457
int orc_solve_sigma_consensus(orc_problem* p, double* rot, int32_t iters_num, double sigma_max,
                              const gsfm_rot_options* opt, gsfm_rot_summary* summary) {
  if (p->error_type != GSFM_ROT_ANGLE_AXIS) return GSFM_ERR_INVALID_ARG;
  if (p->n_cams == 0 || p->edges.empty()) return GSFM_ERR_EMPTY;
  gsfm_rot_options o = opt ? *opt : defaults();
  gsfm_rot_summary local, total; if (!summary) summary = &local;
  std::memset(&total, 0, sizeof(total));
  const MagsacConst c = magsac_const(3);
  const std::vector<double>& table = magsac_table(3);
  const double squared_sigma_max_2 = sigma_max * sigma_max * 2.0;               // :343
  const double dof_minus_one_per_two = (c.nu - 1.0) / 2.0;                      // :344
  const double C_times_two_ad_dof = c.C * std::pow(2.0, dof_minus_one_per_two); // :345
  const double one_over_sigma = C_times_two_ad_dof / sigma_max;                 // :346
  const double gamma_value = std::tgamma(dof_minus_one_per_two);                // :347
  const double weight_zero = one_over_sigma * (gamma_value - c.upper_incomplete_gamma_of_k);  // :348-349
  const size_t E = p->edges.size();
  std::vector<double> last_weights(E, 0.0), weights(E, 0.0);
  const double t0 = now_ms();
  int outer = 0;
  for (int it = 0; it < iters_num; ++it) {
    ++outer;
    for (size_t e = 0; e < E; ++e) {
      const Edge& ed = p->edges[e];
      AngleAxisError f; std::memcpy(f.rel_aa, ed.rel_aa, 24);
      for (int k = 0; k < 9; ++k) f.W[k] = (k % 4 == 0) ? 1.0 : 0.0;
      double err[3]; f(rot + 3 * ed.i, rot + 3 * ed.j, err);                     // :378-397
      const double residual = std::sqrt(err[0] * err[0] + err[1] * err[1] + err[2] * err[2]);
      double weight;
      if (residual < std::numeric_limits<double>::epsilon()) weight = weight_zero;  // :400-401
      else {
        const double squared_residual = residual * residual;
        size_t x = (size_t)std::round(c.precision * squared_residual / squared_sigma_max_2);  // :407 (C round)
        // :411-412 clamps to stored_gamma_number3, which indexes one past the end of the table in the
        // reference (undefined read). The restatement reads the last stored entry instead.
        if ((size_t)c.stored_gamma_number - 1 < x) x = (size_t)c.stored_gamma_number - 1;
        weight = one_over_sigma * (table[x] - c.upper_incomplete_gamma_of_k);    // :415
      }
      weights[e] = weight;
      p->edges[e].W[0] = p->edges[e].W[4] = p->edges[e].W[8] = weight;           // :420-421
    }
    double avg = 0; for (size_t e = 0; e < E; ++e) avg += std::fabs(weights[e] - last_weights[e]);
    avg /= (double)E;                                                            // :429-433
    std::swap(weights, last_weights);                                            // :436
    std::vector<double> x; state_from_aa(p, rot, x);
    lm_solve(p, x, o, summary);                                                  // :440-446
    state_to_aa(p, x, rot);
    if (it == 0) total = *summary;
    else {
      total.num_iterations += summary->num_iterations; total.num_successful_steps += summary->num_successful_steps;
      total.num_unsuccessful_steps += summary->num_unsuccessful_steps; total.num_residual_sweeps += summary->num_residual_sweeps;
      total.num_linearizations += summary->num_linearizations; total.num_cg_iterations += summary->num_cg_iterations;
      total.final_cost = summary->final_cost; total.termination = summary->termination;
      total.final_gradient_max_norm = summary->final_gradient_max_norm; total.final_radius = summary->final_radius;
    }
    total.last_weight_change = avg;
    if (avg <= 1e-7) break;                                                      // :448
  }
  total.outer_iterations = outer; total.t_total_ms = now_ms() - t0;
  *summary = total;
  return 0;
}

// Test hook: keep the linear systems of the first `max_steps` LM iterations of the NEXT solve (0 switches it off and frees them).
int orc_capture_steps(orc_problem* p, int32_t max_steps) { p->capture_max = max_steps; p->captured.clear(); return 0; }
// System k of the last solve: (J^T J + diag(D)^2) y = rhs, J = per edge the (res_dim x 3) blocks Ji (column block of `first`) and Jj (of
// `second`), Corrector and Jacobi column scaling applied; y = what the oracle's linear solver returned; the parameter step is -y * scale.
// Any output may be NULL.  Returns the PCG iteration count of that solve (0 for the dense solver), -1 if k was not captured.
int orc_captured_step(orc_problem* p, int32_t k, double* Ji, double* Jj, double* rt, double* D, double* rhs, double* y, double* scale) {
  if (k < 0 || k >= (int)p->captured.size()) return -1;
  const orc_problem::Captured& c = p->captured[k];
  if (Ji) std::memcpy(Ji, c.Ji.data(), 8 * c.Ji.size());
  if (Jj) std::memcpy(Jj, c.Jj.data(), 8 * c.Jj.size());
  if (rt) std::memcpy(rt, c.rt.data(), 8 * c.rt.size());
  if (D) std::memcpy(D, c.D.data(), 8 * c.D.size());
  if (rhs) std::memcpy(rhs, c.rhs.data(), 8 * c.rhs.size());
  if (y) std::memcpy(y, c.y.data(), 8 * c.y.size());
  if (scale) std::memcpy(scale, c.scale.data(), 8 * c.scale.size());
  return c.cg;
}

int32_t orc_get_trace(orc_problem* p, double* out, int32_t cap_rows) {
  const int rows = (int)(p->trace.size() / ORC_TRACE_COLS);
  const int m = std::min(rows, cap_rows);
  if (out && m > 0) std::memcpy(out, p->trace.data(), sizeof(double) * ORC_TRACE_COLS * m);
  return rows;
}

void orc_loss_eval(const gsfm_loss_node* prog, int32_t n, double s, double* out) {
  Rho r = eval_loss_program(prog, n, s); out[0] = r.v[0]; out[1] = r.v[1]; out[2] = r.v[2];
}
int32_t orc_magsac_table(int32_t nu, double* out, int32_t cap) {
  if (nu != 3 && nu != 4 && nu != 9) return -1;
  const std::vector<double>& t = magsac_table(nu);
  const int m = std::min((int)t.size(), cap);
  if (out && m > 0) std::memcpy(out, t.data(), 8 * (size_t)m);
  return (int)t.size();
}
void orc_magsac_constants(int32_t nu, double* C, double* q, double* gk) {
  const MagsacConst c = magsac_const(nu); *C = c.C; *q = c.sigma_quantile; *gk = c.upper_incomplete_gamma_of_k;
}
void orc_whitening(int32_t error_type, const double* cov6, double inlier_w, double* W9) { whitening(error_type, cov6, inlier_w, W9); }

// primitives for known-answer tests
void orc_angle_axis_to_rotation_matrix(const double* aa, double* R) { AngleAxisToRotationMatrix(aa, R); }
void orc_rotation_matrix_to_angle_axis(const double* R, double* aa) { RotationMatrixToAngleAxis(R, aa); }
void orc_angle_axis_to_quaternion(const double* aa, double* q) { AngleAxisToQuaternion(aa, q); }
void orc_quaternion_to_angle_axis(const double* q, double* aa) { QuaternionToAngleAxis(q, aa); }
void orc_pairwise_rotation_error(const double* aa1, const double* aa2, const double* rel_aa, double weight, double* out) {
  AngleAxisError f; std::memcpy(f.rel_aa, rel_aa, 24);
  for (int k = 0; k < 9; ++k) f.W[k] = (k % 4 == 0) ? weight : 0.0;
  f(aa1, aa2, out);
}
// one edge: raw residual + local jacobians (autodiff), for Jacobian parity tests
int orc_edge_jacobians(orc_problem* p, uint64_t e, const double* rot_aa, double* r, double* Ji, double* Jj) {
  if (e >= p->edges.size()) return 1;
  std::vector<double> x; state_from_aa(p, rot_aa, x);
  edge_autodiff(p, p->edges[e], x.data(), r, Ji, Jj);
  return 0;
}

}  // extern "C"

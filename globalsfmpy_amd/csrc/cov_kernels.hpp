// Batched per-edge rotation covariance (reference src/uncertainty.cpp:36-198).
//
// For every view pair: refine (rotation, translation) on the Sampson distance of its matched features
// (translation on the sphere: ceres::HomogeneousVectorParameterization), then covariance of the rotation with the
// translation fixed = (J_R^T J_R)^-1.  The reference runs one Ceres problem per edge (20 threads, 500 iterations);
// here ONE WAVEFRONT owns one edge: the 64 lanes stride over the matches, residuals and analytic 1x5 Jacobians are
// reduced with a fixed xor-butterfly (every lane ends up with the same 5x5 normal matrix), and all lanes run the
// tiny Levenberg-Marquardt control redundantly, so the whole solve needs no LDS, no atomics and no host round trip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "so3_dev.hpp"

namespace gsfm {

struct CovArgs {
  uint64_t n_edges;
  const uint64_t* match_ptr;   // [n_edges + 1]
  const double4* matches;      // x1 y1 x2 y2 (pixels)
  const double* intr;          // f1 u1 v1 f2 u2 v2 per edge
  const double* rot_in;        // 3 per edge (angle-axis of R_12)
  const double* trans_in;      // 3 per edge
  int max_iterations;
  double* cov9;                // row-major 3x3 per edge
  double* rot_out;
  double* trans_out;
  int* status;                 // 0 ok, 1 skipped (zero translation / no matches), 2 singular
  int* iters;
};

__device__ __forceinline__ double wave_allsum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// ceres internal::ComputeHouseholderVector for a 3-vector
__device__ __forceinline__ void householder3(const double* x, double* v, double* beta) {
  const double sigma = x[0] * x[0] + x[1] * x[1];
  v[0] = x[0]; v[1] = x[1]; v[2] = 1.0;
  *beta = 0.0;
  const double xp = x[2];
  if (sigma <= 2.220446049250313e-16) { if (xp < 0.0) *beta = 2.0; return; }
  const double mu = sqrt(xp * xp + sigma);
  const double vp = (xp <= 0.0) ? xp - mu : -sigma / (xp + mu);
  *beta = 2.0 * vp * vp / (sigma + vp * vp);
  v[0] /= vp; v[1] /= vp;
}
__device__ __forceinline__ void hom_plus3(const double* x, const double* d, double* out) {
  const double nd = sqrt(d[0] * d[0] + d[1] * d[1]);
  if (nd == 0.0) { out[0] = x[0]; out[1] = x[1]; out[2] = x[2]; return; }
  const double h = 0.5 * nd;
  double sh, ch;
  sincos(h, &sh, &ch);
  const double sbd = sh / h;
  const double y[3] = {0.5 * sbd * d[0], 0.5 * sbd * d[1], ch};
  double v[3], beta;
  householder3(x, v, &beta);
  const double vy = v[0] * y[0] + v[1] * y[1] + v[2] * y[2];
  const double nx = sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
#pragma unroll
  for (int k = 0; k < 3; ++k) out[k] = nx * (y[k] - v[k] * (beta * vy));
}
__device__ __forceinline__ void hom_jac3(const double* x, double* J /*3x2 row-major*/) {
  double v[3], beta;
  householder3(x, v, &beta);
  const double nx = sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int r = 0; r < 3; ++r) J[2 * r + i] = -0.5 * beta * v[i] * v[r];
    J[2 * i + i] += 0.5;
  }
#pragma unroll
  for (int k = 0; k < 6; ++k) J[k] *= nx;
}

// ceres::AngleAxisToRotationMatrix (row-major) + the left Jacobian J_l(w) for d/dw
__device__ __forceinline__ void rodrigues(const double* w, double* R) {
  const double t2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  if (t2 > 2.220446049250313e-16) {
    const double t = sqrt(t2), wx = w[0] / t, wy = w[1] / t, wz = w[2] / t;
    double s, c;
    sincos(t, &s, &c);
    const double k = 1.0 - c;
    R[0] = c + wx * wx * k; R[1] = wx * wy * k - wz * s; R[2] = wy * s + wx * wz * k;
    R[3] = wz * s + wx * wy * k; R[4] = c + wy * wy * k; R[5] = -wx * s + wy * wz * k;
    R[6] = -wy * s + wx * wz * k; R[7] = wx * s + wy * wz * k; R[8] = c + wz * wz * k;
  } else {
    R[0] = 1; R[1] = -w[2]; R[2] = w[1]; R[3] = w[2]; R[4] = 1; R[5] = -w[0]; R[6] = -w[1]; R[7] = w[0]; R[8] = 1;
  }
}

struct PairEval { double cost; double H[15]; double g[5]; };  // H: upper triangle row-major (00 01 02 03 04 11 12 ...)

// cost [+ J^T J and J^T r in the 5 local parameters] of one edge; all lanes return identical values
__device__ __forceinline__ void pair_evaluate(const CovArgs& a, uint64_t mb, uint64_t me, const double* K, const double* rot,
                                              const double* t, bool want_jac, PairEval* out) {
  const int lane = threadIdx.x & 63;
  double R[9], T[9], Ti[9], Jh[6];
  rodrigues(rot, R);
  if (want_jac) { jl_and_inverse(rot, T, Ti); hom_jac3(t, Jh); }
  const double f1 = K[0], u1 = K[1], v1 = K[2], f2 = K[3], u2 = K[4], v2 = K[5];
  const double Rt[3] = {R[0] * t[0] + R[1] * t[1] + R[2] * t[2], R[3] * t[0] + R[4] * t[1] + R[5] * t[2], R[6] * t[0] + R[7] * t[1] + R[8] * t[2]};
  double cost = 0.0, H[15], g[5];
#pragma unroll
  for (int k = 0; k < 15; ++k) H[k] = 0.0;
#pragma unroll
  for (int k = 0; k < 5; ++k) g[k] = 0.0;
  for (uint64_t k = mb + lane; k < me; k += 64) {
    const double4 m = a.matches[k];
    const double p1[3] = {(m.x - u1) / f1, (m.y - v1) / f1, 1.0};
    const double p2[3] = {(m.z - u2) / f2, (m.w - v2) / f2, 1.0};
    const double c[3] = {t[1] * p1[2] - t[2] * p1[1], t[2] * p1[0] - t[0] * p1[2], t[0] * p1[1] - t[1] * p1[0]};   // t x p1
    const double av[3] = {R[0] * c[0] + R[1] * c[1] + R[2] * c[2], R[3] * c[0] + R[4] * c[1] + R[5] * c[2], R[6] * c[0] + R[7] * c[1] + R[8] * c[2]};
    const double e0 = av[0] / f2, e1 = av[1] / f2;
    const double num = p2[0] * av[0] + p2[1] * av[1] + av[2];                                                     // x2^T F x1
    const double rp[3] = {R[0] * p2[0] + R[3] * p2[1] + R[6] * p2[2], R[1] * p2[0] + R[4] * p2[1] + R[7] * p2[2], R[2] * p2[0] + R[5] * p2[1] + R[8] * p2[2]};
    const double b0 = rp[1] * t[2] - rp[2] * t[1], b1 = rp[2] * t[0] - rp[0] * t[2];                                // (R^T p2) x t
    const double g0 = b0 / f1, g1 = b1 / f1;
    const double den = g0 * g0 + g1 * g1 + e0 * e0 + e1 * e1;
    const double isd = 1.0 / sqrt(den);
    const double r = fabs(num) * isd;
    cost += 0.5 * r * r;
    if (!want_jac) continue;
    // ---- d/d eta (left perturbation of R), then d/d omega = (.) J_l(omega) ----
    double dn[3] = {av[1] * p2[2] - av[2] * p2[1], av[2] * p2[0] - av[0] * p2[2], av[0] * p2[1] - av[1] * p2[0]};  // a x p2
    const double de0[3] = {0.0, av[2] / f2, -av[1] / f2};
    const double de1[3] = {-av[2] / f2, 0.0, av[0] / f2};
    const double trp = t[0] * rp[0] + t[1] * rp[1] + t[2] * rp[2];
    const double db0[3] = {(trp * R[0] - rp[0] * Rt[0]) / f1, (trp * R[3] - rp[0] * Rt[1]) / f1, (trp * R[6] - rp[0] * Rt[2]) / f1};
    const double db1[3] = {(trp * R[1] - rp[1] * Rt[0]) / f1, (trp * R[4] - rp[1] * Rt[1]) / f1, (trp * R[7] - rp[1] * Rt[2]) / f1};
    // (sign(num) at num == 0 is taken as +1.  The reference's autodiff of sqrt(num^2 / den) has NO derivative there -- 0.5 / sqrt(0) * 0 = NaN,
    // and Ceres rejects the evaluation -- which only an exactly noise-free match at the exact pose can hit; the oracle restates the NaN.)
    const double sg = (num < 0.0) ? -isd : isd, q = num / den;
    double jeta[3], jt[3];
#pragma unroll
    for (int x = 0; x < 3; ++x) jeta[x] = sg * (dn[x] - q * (g0 * db0[x] + g1 * db1[x] + e0 * de0[x] + e1 * de1[x]));
    // ---- d/d t (ambient) ----
    const double dnt[3] = {p1[1] * rp[2] - p1[2] * rp[1], p1[2] * rp[0] - p1[0] * rp[2], p1[0] * rp[1] - p1[1] * rp[0]};  // p1 x rp
    const double det0[3] = {-(R[1] * p1[2] - R[2] * p1[1]) / f2, -(R[2] * p1[0] - R[0] * p1[2]) / f2, -(R[0] * p1[1] - R[1] * p1[0]) / f2};  // -(R_0 x p1)/f2
    const double det1[3] = {-(R[4] * p1[2] - R[5] * p1[1]) / f2, -(R[5] * p1[0] - R[3] * p1[2]) / f2, -(R[3] * p1[1] - R[4] * p1[0]) / f2};
    const double dbt0[3] = {0.0, -rp[2] / f1, rp[1] / f1};
    const double dbt1[3] = {rp[2] / f1, 0.0, -rp[0] / f1};
#pragma unroll
    for (int x = 0; x < 3; ++x) jt[x] = sg * (dnt[x] - q * (g0 * dbt0[x] + g1 * dbt1[x] + e0 * det0[x] + e1 * det1[x]));
    double j[5];
#pragma unroll
    for (int x = 0; x < 3; ++x) j[x] = jeta[0] * T[x] + jeta[1] * T[3 + x] + jeta[2] * T[6 + x];
    j[3] = jt[0] * Jh[0] + jt[1] * Jh[2] + jt[2] * Jh[4];
    j[4] = jt[0] * Jh[1] + jt[1] * Jh[3] + jt[2] * Jh[5];
    int o = 0;
#pragma unroll
    for (int x = 0; x < 5; ++x) {
      g[x] += j[x] * r;
#pragma unroll
      for (int y = x; y < 5; ++y) H[o++] += j[x] * j[y];
    }
  }
  out->cost = wave_allsum(cost);
  if (want_jac) {
#pragma unroll
    for (int k = 0; k < 15; ++k) out->H[k] = wave_allsum(H[k]);
#pragma unroll
    for (int k = 0; k < 5; ++k) out->g[k] = wave_allsum(g[k]);
  }
}

__device__ __forceinline__ int hidx(int x, int y) { if (x > y) { const int t = x; x = y; y = t; } return x * 5 - x * (x - 1) / 2 + (y - x); }

// 5x5 SPD solve (Cholesky), A given by its upper triangle
__device__ __forceinline__ bool chol5(const double* Au, const double* diag_add, const double* b, double* x) {
  double L[25];
#pragma unroll
  for (int i = 0; i < 5; ++i) {
#pragma unroll
    for (int jx = 0; jx <= i; ++jx) {
      double s = Au[hidx(i, jx)] + ((i == jx) ? diag_add[i] : 0.0);
      for (int k = 0; k < jx; ++k) s -= L[5 * i + k] * L[5 * jx + k];
      if (i == jx) { if (!(s > 0.0)) return false; L[5 * i + jx] = sqrt(s); }
      else L[5 * i + jx] = s / L[5 * jx + jx];
    }
  }
  double y[5];
  for (int i = 0; i < 5; ++i) { double s = b[i]; for (int k = 0; k < i; ++k) s -= L[5 * i + k] * y[k]; y[i] = s / L[5 * i + i]; }
  for (int i = 4; i >= 0; --i) { double s = y[i]; for (int k = i + 1; k < 5; ++k) s -= L[5 * k + i] * x[k]; x[i] = s / L[5 * i + i]; }
  return true;
}

// Is the 3 x 3 information matrix rank deficient to working precision?  Cholesky with diagonal pivoting (the criterion of LAPACK's
// dpstrf): a pivot below 1e-14 of the first one -- ceres::Covariance's own bound on the reciprocal condition number
// (Covariance::Options::min_reciprocal_condition_number).  There ceres::Covariance::Compute returns false and the reference
// CHECK-aborts (src/uncertainty.cpp:157); here the view pair gets status 2 and no covariance, instead of the inverse of rounding noise.
__host__ __device__ inline bool sym3_rank_deficient(double h00, double h01, double h02, double h11, double h12, double h22) {
  const double tol = 1e-14;
  // first pivot: the largest diagonal entry
  double d1, a, b, ab, a1, b1;   // pivot; the other two diagonal entries, their coupling, and their couplings to the pivot
  if (h00 >= h11 && h00 >= h22) { d1 = h00; a = h11; b = h22; ab = h12; a1 = h01; b1 = h02; }
  else if (h11 >= h22) { d1 = h11; a = h00; b = h22; ab = h02; a1 = h01; b1 = h12; }
  else { d1 = h22; a = h00; b = h11; ab = h01; a1 = h02; b1 = h12; }
  if (!(d1 > 0.0)) return true;
  const double saa = a - a1 * a1 / d1, sbb = b - b1 * b1 / d1, sab = ab - a1 * b1 / d1;   // Schur complement
  const double d2 = saa >= sbb ? saa : sbb, other = saa >= sbb ? sbb : saa;
  if (!(d2 > tol * d1)) return true;
  const double d3 = other - sab * sab / d2;
  return !(d3 > tol * d1);
}

__global__ void __launch_bounds__(GSFM_BLOCK) k_cov_estimate(CovArgs a) {
  const uint64_t e = (uint64_t)blockIdx.x * (GSFM_BLOCK / 64) + (threadIdx.x >> 6);
  if (e >= a.n_edges) return;
  const int lane = threadIdx.x & 63;
  double K[6], rot[3], t[3];
#pragma unroll
  for (int k = 0; k < 6; ++k) K[k] = a.intr[6 * e + k];
#pragma unroll
  for (int k = 0; k < 3; ++k) { rot[k] = a.rot_in[3 * e + k]; t[k] = a.trans_in[3 * e + k]; }
  const uint64_t mb = a.match_ptr[e], me = a.match_ptr[e + 1];
  int status = 0, iteration = 0;
  double cov[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  if ((t[0] == 0.0 && t[1] == 0.0 && t[2] == 0.0) || me == mb) status = 1;   // uncertainty.cpp:123
  else {
    // --- ceres::Solve, Levenberg-Marquardt with Ceres 1.14 defaults (same control law as lm_solve in gsfm_rot.hip) ---
    PairEval cur;
    double scale[5], radius = 1e4, decrease_factor = 2.0;
    int invalid = 0;
    pair_evaluate(a, mb, me, K, rot, t, true, &cur);
#pragma unroll
    for (int k = 0; k < 5; ++k) scale[k] = 1.0 / (1.0 + sqrt(cur.H[hidx(k, k)]));
    double x_norm = sqrt(rot[0] * rot[0] + rot[1] * rot[1] + rot[2] * rot[2] + t[0] * t[0] + t[1] * t[1] + t[2] * t[2]);
    auto grad_max = [&](const PairEval& ev) {
      double ng[2] = {-ev.g[3], -ev.g[4]}, tp[3];
      hom_plus3(t, ng, tp);
      double gm = 0.0;
#pragma unroll
      for (int k = 0; k < 3; ++k) gm = fmax(gm, fmax(fabs(ev.g[k]), fabs(t[k] - tp[k])));
      return gm;
    };
    double gmax = grad_max(cur);
    bool running = isfinite(cur.cost) && gmax > 1e-10, last_ok = false;
    while (running) {
      if (iteration >= a.max_iterations) break;
      if (last_ok && gmax <= 1e-10) break;
      if (radius <= 1e-32) break;
      ++iteration; last_ok = false;
      // scaled normal equations: (S H S + D^2) y = S g, D^2 = clamp(diag(S H S), 1e-6, 1e32) / radius
      double Hs[15], bs[5], dd[5], step[5];
#pragma unroll
      for (int x = 0; x < 5; ++x) {
        bs[x] = scale[x] * cur.g[x];
#pragma unroll
        for (int y = x; y < 5; ++y) Hs[hidx(x, y)] = scale[x] * scale[y] * cur.H[hidx(x, y)];
      }
#pragma unroll
      for (int x = 0; x < 5; ++x) dd[x] = fmin(fmax(Hs[hidx(x, x)], 1e-6), 1e32) / radius;
      bool valid = chol5(Hs, dd, bs, step);
      double model_cost_change = 0.0;
      if (valid) {
#pragma unroll
        for (int x = 0; x < 5; ++x) { step[x] = -step[x]; if (!isfinite(step[x])) valid = false; }
        // -(J d).(r + J d / 2) = -d.b - 1/2 d^T H d
        double dHd = 0.0, db = 0.0;
#pragma unroll
        for (int x = 0; x < 5; ++x) {
          db += step[x] * bs[x];
#pragma unroll
          for (int y = 0; y < 5; ++y) dHd += step[x] * Hs[hidx(x, y)] * step[y];
        }
        model_cost_change = -db - 0.5 * dHd;
        if (!(model_cost_change > 0.0)) valid = false;
      }
      if (!valid) { if (++invalid >= 5) break; radius /= decrease_factor; decrease_factor *= 2.0; continue; }
      invalid = 0;
      double delta[5], crot[3], ct[3];
#pragma unroll
      for (int x = 0; x < 5; ++x) delta[x] = step[x] * scale[x];
#pragma unroll
      for (int x = 0; x < 3; ++x) crot[x] = rot[x] + delta[x];
      hom_plus3(t, delta + 3, ct);
      PairEval cand;
      pair_evaluate(a, mb, me, K, crot, ct, false, &cand);
      double cand_cost = isfinite(cand.cost) ? cand.cost : 1.7976931348623157e308;
      double sn = 0.0;
#pragma unroll
      for (int x = 0; x < 3; ++x) sn += (rot[x] - crot[x]) * (rot[x] - crot[x]) + (t[x] - ct[x]) * (t[x] - ct[x]);
      sn = sqrt(sn);
      const double cost_change = cur.cost - cand_cost, rel_dec = cost_change / model_cost_change;
      if (sn <= 1e-8 * (x_norm + 1e-8)) break;                    // parameter tolerance
      if (fabs(cost_change) <= 1e-6 * cur.cost) break;            // function tolerance
      if (rel_dec > 1e-3) {
#pragma unroll
        for (int x = 0; x < 3; ++x) { rot[x] = crot[x]; t[x] = ct[x]; }
        x_norm = sqrt(rot[0] * rot[0] + rot[1] * rot[1] + rot[2] * rot[2] + t[0] * t[0] + t[1] * t[1] + t[2] * t[2]);
        pair_evaluate(a, mb, me, K, rot, t, true, &cur);
        gmax = grad_max(cur);
        radius = fmin(1e16, radius / fmax(1.0 / 3.0, 1.0 - pow(2.0 * rel_dec - 1.0, 3.0)));
        decrease_factor = 2.0; last_ok = true;
      } else { radius /= decrease_factor; decrease_factor *= 2.0; }
    }
    // --- ceres::Covariance of the rotation block, translation constant: (J_R^T J_R)^-1 (uncertainty.cpp:145-160) ---
    const double h00 = cur.H[hidx(0, 0)], h01 = cur.H[hidx(0, 1)], h02 = cur.H[hidx(0, 2)], h11 = cur.H[hidx(1, 1)], h12 = cur.H[hidx(1, 2)], h22 = cur.H[hidx(2, 2)];
    const double c00 = h11 * h22 - h12 * h12, c01 = h12 * h02 - h01 * h22, c02 = h01 * h12 - h11 * h02;
    const double det = h00 * c00 + h01 * c01 + h02 * c02;
    if (!(fabs(det) > 0.0) || !isfinite(det) || sym3_rank_deficient(h00, h01, h02, h11, h12, h22)) status = 2;
    else {
      cov[0] = c00 / det; cov[1] = c01 / det; cov[2] = c02 / det;
      cov[3] = cov[1]; cov[4] = (h00 * h22 - h02 * h02) / det; cov[5] = (h02 * h01 - h00 * h12) / det;
      cov[6] = cov[2]; cov[7] = cov[5]; cov[8] = (h00 * h11 - h01 * h01) / det;
    }
  }
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 9; ++k) a.cov9[9 * e + k] = cov[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) { a.rot_out[3 * e + k] = rot[k]; a.trans_out[3 * e + k] = t[k]; }
    a.status[e] = status;
    if (a.iters) a.iters[e] = iteration;
  }
}

}  // namespace gsfm

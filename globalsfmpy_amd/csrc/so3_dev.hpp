// Device-side SO(3) math for the gfx950 rotation-averaging kernels (fp64).
//
// Rotations live on the device as unit quaternions in Eigen coefficient order
// (x, y, z, w) -- the reference's own state for the QUATERNION_* error types
// (src/GSfM_nonlinear_rotation_estimator.cpp:130-143) and a per-iteration cache
// for the angle-axis types, so the per-edge kernels never evaluate sin/cos:
// the error rotation of an edge is two quaternion products, its log one atan2.
//
// All Jacobians are taken with respect to LEFT perturbations R_k <- Exp(eta_k) R_k.
// The map from the reference's own parameter step to eta is per camera
// (eta = J_l(omega) d_omega for additive angle-axis, eta = 2 delta for ceres'
// EigenQuaternionParameterization) and is applied in O(N) camera kernels.
#pragma once
#include <hip/hip_runtime.h>
#include "devmath.hpp"

namespace gsfm {

struct Quat { double x, y, z, w; };

__device__ __forceinline__ Quat qmul(const Quat& a, const Quat& b) {
  Quat r;
  r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
  r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
  r.y = a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z;
  r.z = a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x;
  return r;
}
__device__ __forceinline__ Quat qconj(const Quat& a) { return Quat{-a.x, -a.y, -a.z, a.w}; }

// Rotation matrix of a unit quaternion, row-major.
__device__ __forceinline__ void qmat(const Quat& q, double* R) {
  const double tx = 2.0 * q.x, ty = 2.0 * q.y, tz = 2.0 * q.z;
  const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
  const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
  const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  R[0] = 1.0 - (tyy + tzz); R[1] = txy - twz;         R[2] = txz + twy;
  R[3] = txy + twz;         R[4] = 1.0 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;         R[7] = tyz + twx;         R[8] = 1.0 - (txx + tyy);
}

// Angle-axis -> quaternion (ceres::AngleAxisToQuaternion semantics, incl. the theta == 0 branch).
__device__ __forceinline__ Quat aa_to_quat(double a0, double a1, double a2) {
  const double t2 = a0 * a0 + a1 * a1 + a2 * a2;
  Quat q;
  if (t2 > 0.0) {
    const double t = sqrt(t2);
    double sh, ch;
    sincos(0.5 * t, &sh, &ch);
    const double k = sh / t;
    q.w = ch; q.x = a0 * k; q.y = a1 * k; q.z = a2 * k;
  } else {
    q.w = 1.0; q.x = 0.5 * a0; q.y = 0.5 * a1; q.z = 0.5 * a2;
  }
  return q;
}

// Quaternion -> angle-axis (ceres::QuaternionToAngleAxis semantics: result angle in (-pi, pi]).
// Also returns theta^2 = |e|^2 and (|w|, s = |v|) for the Jacobian coefficient below.
// SC: atan2 with its coefficients in scalar registers (devmath.hpp; same bits) -- for kernels whose vector registers are the constraint
// (the linearisations); the sweeps are bound by instruction issue and keep the library's form (measured: profiles/r04b_*).
template <bool SC = false>
__device__ __forceinline__ void quat_log(const Quat& q, double* e, double* s_out, double* theta_out) {
  const double s2 = q.x * q.x + q.y * q.y + q.z * q.z;
  if (s2 > 0.0) {
    // one reciprocal square root serves both |v| and the division by it (a correctly rounded sqrt followed by an IEEE
    // division is ~25 instructions more per edge for a result that differs in the last ulp or two)
    const double rs = rsqrt(s2);
    const double s = s2 * rs;
    // ceres takes atan2(-s, -w) for w < 0 and atan2(s, w) otherwise.  atan2 is odd in its first argument bit for bit, so both are
    // +-atan2(s, |w|): ONE evaluation.  (Written as the two-way select, the compiler emitted two inline copies of atan2 under divergent
    // exec masks -- the sign of w is arbitrary under the double cover, so almost every wavefront ran both.)
    const double half = SC ? atan2_q1(s, fabs(q.w)) : atan2(s, fabs(q.w));
    const double two_theta = 2.0 * ((q.w < 0.0) ? -half : half);
    const double k = two_theta * rs;
    e[0] = q.x * k; e[1] = q.y * k; e[2] = q.z * k;
    *s_out = s; *theta_out = two_theta;
  } else {
    e[0] = 2.0 * q.x; e[1] = 2.0 * q.y; e[2] = 2.0 * q.z;
    *s_out = 0.0; *theta_out = 0.0;
  }
}

// c(theta) in J_l^{-1}(phi) = I - 1/2 [phi]x + c [phi]x^2,  c = (1 - (theta/2) cot(theta/2)) / theta^2.
// cot(|theta|/2) = |w| / s for the unit quaternion (v, w) of the same rotation.
__device__ __forceinline__ double jlinv_coeff(double theta, double s, double absw) {
  const double t2 = theta * theta;
  if (t2 < 0.1) {
    return 1.0 / 12.0 + t2 * (1.0 / 720.0 + t2 * (1.0 / 30240.0 + t2 * (1.0 / 1209600.0 + t2 * (1.0 / 47900160.0))));
  }
  return (1.0 - 0.5 * fabs(theta) * absw / s) / t2;
}

// B = I - 1/2 [e]x + c [e]x^2  (row-major). [e]x^2 = e e^T - |e|^2 I.
__device__ __forceinline__ void jlinv_matrix(const double* e, double c, double* B) {
  const double n2 = e[0] * e[0] + e[1] * e[1] + e[2] * e[2];
  const double d = 1.0 - c * n2;
  const double h0 = 0.5 * e[0], h1 = 0.5 * e[1], h2 = 0.5 * e[2];
  B[0] = d + c * e[0] * e[0]; B[1] = c * e[0] * e[1] + h2; B[2] = c * e[0] * e[2] - h1;
  B[3] = c * e[1] * e[0] - h2; B[4] = d + c * e[1] * e[1]; B[5] = c * e[1] * e[2] + h0;
  B[6] = c * e[2] * e[0] + h1; B[7] = c * e[2] * e[1] - h0; B[8] = d + c * e[2] * e[2];
}

// Left Jacobian J_l(w) = I + a [w]x + b [w]x^2 and its inverse, for the additive angle-axis
// parameterisation (per camera, O(N) per iteration).
__device__ __forceinline__ void jl_and_inverse(const double* w, double* T, double* Tinv) {
  const double t2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  double a, b, c;
  if (t2 < 1e-2) {
    a = 0.5 - t2 * (1.0 / 24.0 - t2 * (1.0 / 720.0 - t2 * (1.0 / 40320.0 - t2 * (1.0 / 3628800.0))));
    b = 1.0 / 6.0 - t2 * (1.0 / 120.0 - t2 * (1.0 / 5040.0 - t2 * (1.0 / 362880.0 - t2 * (1.0 / 39916800.0))));
    c = 1.0 / 12.0 + t2 * (1.0 / 720.0 + t2 * (1.0 / 30240.0 + t2 * (1.0 / 1209600.0)));
  } else {
    const double t = sqrt(t2);
    double sh, ch;
    sincos(0.5 * t, &sh, &ch);
    a = 2.0 * sh * sh / t2;              // (1 - cos t) / t^2
    b = (t - 2.0 * sh * ch) / (t2 * t);  // (t - sin t) / t^3
    c = (t2 < 0.1) ? (1.0 / 12.0 + t2 * (1.0 / 720.0 + t2 * (1.0 / 30240.0 + t2 * (1.0 / 1209600.0 + t2 * (1.0 / 47900160.0)))))
                   : (1.0 - 0.5 * t * ch / sh) / t2;
  }
  const double da = 1.0 - b * t2;
  T[0] = da + b * w[0] * w[0]; T[1] = b * w[0] * w[1] - a * w[2]; T[2] = b * w[0] * w[2] + a * w[1];
  T[3] = b * w[1] * w[0] + a * w[2]; T[4] = da + b * w[1] * w[1]; T[5] = b * w[1] * w[2] - a * w[0];
  T[6] = b * w[2] * w[0] - a * w[1]; T[7] = b * w[2] * w[1] + a * w[0]; T[8] = da + b * w[2] * w[2];
  jlinv_matrix(w, c, Tinv);
}

__device__ __forceinline__ void mat3_mul(const double* A, const double* B, double* C) {  // C = A B
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) C[3 * r + c] = A[3 * r] * B[c] + A[3 * r + 1] * B[3 + c] + A[3 * r + 2] * B[6 + c];
}
__device__ __forceinline__ void mat3_tmul(const double* A, const double* B, double* C) {  // C = A^T B
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) C[3 * r + c] = A[r] * B[c] + A[3 + r] * B[3 + c] + A[6 + r] * B[6 + c];
}

// symmetric 3x3 (s00 s01 s02 s11 s12 s22) inverse via cofactors
__device__ __forceinline__ void sym3_inverse(const double* s, double* o) {
  const double c00 = s[3] * s[5] - s[4] * s[4];
  const double c01 = s[2] * s[4] - s[1] * s[5];
  const double c02 = s[1] * s[4] - s[2] * s[3];
  const double det = s[0] * c00 + s[1] * c01 + s[2] * c02;
  const double id = 1.0 / det;
  o[0] = c00 * id; o[1] = c01 * id; o[2] = c02 * id;
  o[3] = (s[0] * s[5] - s[2] * s[2]) * id;
  o[4] = (s[1] * s[2] - s[0] * s[4]) * id;
  o[5] = (s[0] * s[3] - s[1] * s[1]) * id;
}
__device__ __forceinline__ void sym3_mulvec(const double* s, const double* v, double* o) {
  o[0] = s[0] * v[0] + s[1] * v[1] + s[2] * v[2];
  o[1] = s[1] * v[0] + s[3] * v[1] + s[4] * v[2];
  o[2] = s[2] * v[0] + s[4] * v[1] + s[5] * v[2];
}

}  // namespace gsfm

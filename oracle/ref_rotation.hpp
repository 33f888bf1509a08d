// TEST INFRASTRUCTURE — CPU oracle. Not part of the product; never linked by it.
//
// Rotation primitives the reference calls from ceres/rotation.h and Eigen
// (un-vendored: ceres-solver 1.14.0 "Recommand version", reference README.md:21;
//  Eigen3 unpinned, README.md:30).  The published algorithms are restated here,
// branch for branch, templated on the scalar so they run on Jets like the
// reference's autodiff does.  Call sites in the reference:
//   AngleAxisToRotationMatrix / RotationMatrixToAngleAxis:
//     include/pairwise_rotation_error_quat.hpp:223-240,
//     thirdparty/TheiaSfM/src/theia/sfm/global_pose_estimation/pairwise_rotation_error.h:72-89,
//     src/GSfM_nonlinear_rotation_estimator.cpp:380-397
//   AngleAxisToQuaternion / QuaternionToAngleAxis:
//     src/GSfM_nonlinear_rotation_estimator.cpp:130-132,193
//   Eigen quaternion product / conjugate / toRotationMatrix:
//     include/pairwise_rotation_error_quat.hpp:86-93,130-131,170-175
// Matrices are 3x3 row-major: R[3*r + c].
#pragma once
#include <limits>
#include "ref_jet.hpp"

namespace gsfm_oracle {

// ceres::AngleAxisToRotationMatrix
template <typename T>
inline void AngleAxisToRotationMatrix(const T* aa, T* R) {
  const T theta2 = aa[0] * aa[0] + aa[1] * aa[1] + aa[2] * aa[2];
  if (theta2 > T(std::numeric_limits<double>::epsilon())) {
    const T theta = sqrt(theta2);
    const T wx = aa[0] / theta, wy = aa[1] / theta, wz = aa[2] / theta;
    const T c = cos(theta), s = sin(theta);
    const T k = T(1.0) - c;
    R[0] = c + wx * wx * k;       R[1] = wx * wy * k - wz * s;  R[2] = wy * s + wx * wz * k;
    R[3] = wz * s + wx * wy * k;  R[4] = c + wy * wy * k;       R[5] = -wx * s + wy * wz * k;
    R[6] = -wy * s + wx * wz * k; R[7] = wx * s + wy * wz * k;  R[8] = c + wz * wz * k;
  } else {  // first-order Taylor branch
    R[0] = T(1.0); R[1] = -aa[2]; R[2] = aa[1];
    R[3] = aa[2];  R[4] = T(1.0); R[5] = -aa[0];
    R[6] = -aa[1]; R[7] = aa[0];  R[8] = T(1.0);
  }
}

// ceres::RotationMatrixToQuaternion, output (w, x, y, z)
template <typename T>
inline void RotationMatrixToQuaternion(const T* R, T* q) {
  const T trace = R[0] + R[4] + R[8];
  if (trace >= 0.0) {
    T t = sqrt(trace + T(1.0));
    q[0] = T(0.5) * t;
    t = T(0.5) / t;
    q[1] = (R[7] - R[5]) * t;
    q[2] = (R[2] - R[6]) * t;
    q[3] = (R[3] - R[1]) * t;
  } else {
    int i = 0;
    if (R[4] > R[0]) i = 1;
    if (R[8] > R[4 * i]) i = 2;
    const int j = (i + 1) % 3;
    const int k = (j + 1) % 3;
    T t = sqrt(R[4 * i] - R[4 * j] - R[4 * k] + T(1.0));
    q[i + 1] = T(0.5) * t;
    t = T(0.5) / t;
    q[0] = (R[3 * k + j] - R[3 * j + k]) * t;
    q[j + 1] = (R[3 * j + i] + R[3 * i + j]) * t;
    q[k + 1] = (R[3 * k + i] + R[3 * i + k]) * t;
  }
}

// ceres::QuaternionToAngleAxis, input (w, x, y, z)
template <typename T>
inline void QuaternionToAngleAxis(const T* q, T* aa) {
  const T& q1 = q[1]; const T& q2 = q[2]; const T& q3 = q[3];
  const T sin2 = q1 * q1 + q2 * q2 + q3 * q3;
  if (sin2 > T(0.0)) {
    const T sin_theta = sqrt(sin2);
    const T& cos_theta = q[0];
    const T two_theta = T(2.0) * ((cos_theta < 0.0) ? atan2(-sin_theta, -cos_theta)
                                                    : atan2(sin_theta, cos_theta));
    const T k = two_theta / sin_theta;
    aa[0] = q1 * k; aa[1] = q2 * k; aa[2] = q3 * k;
  } else {
    const T k(2.0);
    aa[0] = q1 * k; aa[1] = q2 * k; aa[2] = q3 * k;
  }
}

// ceres::RotationMatrixToAngleAxis = matrix -> quaternion -> angle-axis (1.14)
template <typename T>
inline void RotationMatrixToAngleAxis(const T* R, T* aa) {
  T q[4];
  RotationMatrixToQuaternion(R, q);
  QuaternionToAngleAxis(q, aa);
}

// ceres::AngleAxisToQuaternion, output (w, x, y, z)
template <typename T>
inline void AngleAxisToQuaternion(const T* aa, T* q) {
  const T theta2 = aa[0] * aa[0] + aa[1] * aa[1] + aa[2] * aa[2];
  if (theta2 > T(0.0)) {
    const T theta = sqrt(theta2);
    const T half = theta * T(0.5);
    const T k = sin(half) / theta;
    q[0] = cos(half); q[1] = aa[0] * k; q[2] = aa[1] * k; q[3] = aa[2] * k;
  } else {
    const T k(0.5);
    q[0] = T(1.0); q[1] = aa[0] * k; q[2] = aa[1] * k; q[3] = aa[2] * k;
  }
}

// 3x3 helpers
template <typename T> inline void MatMul(const T* A, const T* B, T* C) {          // C = A B
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c)
    C[3 * r + c] = A[3 * r] * B[c] + A[3 * r + 1] * B[3 + c] + A[3 * r + 2] * B[6 + c]; }
template <typename T> inline void MatMulABt(const T* A, const T* B, T* C) {       // C = A B^T
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c)
    C[3 * r + c] = A[3 * r] * B[3 * c] + A[3 * r + 1] * B[3 * c + 1] + A[3 * r + 2] * B[3 * c + 2]; }

// ---- Eigen::Quaternion restated, coefficient storage order (x, y, z, w) ----
template <typename T> struct QuatXYZW { T x, y, z, w; };

template <typename T> inline QuatXYZW<T> QuatMul(const QuatXYZW<T>& a, const QuatXYZW<T>& b) {
  QuatXYZW<T> r;
  r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
  r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
  r.y = a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z;
  r.z = a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x;
  return r;
}
template <typename T> inline QuatXYZW<T> QuatConj(const QuatXYZW<T>& a) {
  QuatXYZW<T> r; r.x = -a.x; r.y = -a.y; r.z = -a.z; r.w = a.w; return r; }

// Eigen::QuaternionBase::toRotationMatrix (no normalisation)
template <typename T> inline void QuatToRotationMatrix(const QuatXYZW<T>& q, T* R) {
  const T tx = T(2.0) * q.x, ty = T(2.0) * q.y, tz = T(2.0) * q.z;
  const T twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
  const T txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
  const T tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  R[0] = T(1.0) - (tyy + tzz); R[1] = txy - twz;            R[2] = txz + twy;
  R[3] = txy + twz;            R[4] = T(1.0) - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;            R[7] = tyz + twx;            R[8] = T(1.0) - (txx + tyy);
}

}  // namespace gsfm_oracle

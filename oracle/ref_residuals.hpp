// TEST INFRASTRUCTURE — CPU oracle. Not part of the product; never linked by it.
//
// The five residual functors of the reference's rotation-averaging path,
// restated (own code, Eigen-free) so they can be evaluated on doubles and Jets.
#pragma once
#include "ref_rotation.hpp"

namespace gsfm_oracle {

// theia::PairwiseRotationError
// (thirdparty/TheiaSfM/src/theia/sfm/global_pose_estimation/pairwise_rotation_error.h:66-95)
// and PairwiseRotationErrorAngleAxis (include/pairwise_rotation_error_quat.hpp:215-247):
//   r = W * log( Exp(w2) * Exp(w1)^T * Exp(w12)^T ),  W = weight*I  or the 3x3 "Lt".
struct AngleAxisError {
  double rel_aa[3];
  double W[9];  // row-major; weight*I for the scalar variant
  template <typename T>
  void operator()(const T* rotation1, const T* rotation2, T* residuals) const {
    T R1[9], R2[9], Rrel[9], loop[9], err[9], e[3];
    T rel[3] = {T(rel_aa[0]), T(rel_aa[1]), T(rel_aa[2])};
    AngleAxisToRotationMatrix(rotation1, R1);
    AngleAxisToRotationMatrix(rotation2, R2);
    AngleAxisToRotationMatrix(rel, Rrel);
    MatMulABt(R2, R1, loop);    // loop_rotation = R2 * R1^T
    MatMulABt(loop, Rrel, err); // error_rotation = loop * Rrel^T
    RotationMatrixToAngleAxis(err, e);
    for (int r = 0; r < 3; ++r)
      residuals[r] = T(W[3 * r]) * e[0] + T(W[3 * r + 1]) * e[1] + T(W[3 * r + 2]) * e[2];
  }
};

// PairwiseRotationErrorQuat (include/pairwise_rotation_error_quat.hpp:82-106)
// parameters are Eigen coefficient order (x, y, z, w)
struct QuatCosineError {
  double rel[4];  // x y z w
  double weight;
  template <typename T>
  void operator()(const T* rotation1, const T* rotation2, T* residuals) const {
    QuatXYZW<T> qa{rotation1[0], rotation1[1], rotation1[2], rotation1[3]};
    QuatXYZW<T> qb{rotation2[0], rotation2[1], rotation2[2], rotation2[3]};
    QuatXYZW<T> qr{T(rel[0]), T(rel[1]), T(rel[2]), T(rel[3])};
    QuatXYZW<T> est = QuatMul(qb, QuatConj(qa));        // q_b * q_a^-1
    QuatXYZW<T> dq = QuatMul(qr, QuatConj(est));        // q_rel * est^*
    residuals[0] = T(weight) * T(2.0) * dq.x;
    residuals[1] = T(weight) * T(2.0) * dq.y;
    residuals[2] = T(weight) * T(2.0) * dq.z;
  }
};

// PairwiseRotationErrorQuatFNorm (quat.hpp:125-150); the sign canonicalisation tests
// coeffs()[1] (= y), as the reference does (:135,:139).
struct QuatNormError {
  double rel[4];
  double weight;
  template <typename T>
  void operator()(const T* rotation1, const T* rotation2, T* residuals) const {
    QuatXYZW<T> qa{rotation1[0], rotation1[1], rotation1[2], rotation1[3]};
    QuatXYZW<T> qb{rotation2[0], rotation2[1], rotation2[2], rotation2[3]};
    QuatXYZW<T> qr{T(rel[0]), T(rel[1]), T(rel[2]), T(rel[3])};
    QuatXYZW<T> est = QuatMul(qr, qa);
    if (qb.y < T(0.0)) { qb.x = -qb.x; qb.y = -qb.y; qb.z = -qb.z; qb.w = -qb.w; }
    if (est.y < T(0.0)) { est.x = -est.x; est.y = -est.y; est.z = -est.z; est.w = -est.w; }
    residuals[0] = T(weight) * (qb.x - est.x);
    residuals[1] = T(weight) * (qb.y - est.y);
    residuals[2] = T(weight) * (qb.z - est.z);
    residuals[3] = T(weight) * (qb.w - est.w);
  }
};

// PairwiseRotationErrorRotFNorm (quat.hpp:167-196): column-major linear index k -> (k%3, k/3)
struct RotFNormError {
  double rel[4];
  double weight;
  template <typename T>
  void operator()(const T* rotation1, const T* rotation2, T* residuals) const {
    QuatXYZW<T> qa{rotation1[0], rotation1[1], rotation1[2], rotation1[3]};
    QuatXYZW<T> qb{rotation2[0], rotation2[1], rotation2[2], rotation2[3]};
    QuatXYZW<double> qr{rel[0], rel[1], rel[2], rel[3]};
    T R1[9], R2[9], est[9];
    double Rrel_d[9];
    QuatToRotationMatrix(qa, R1);
    QuatToRotationMatrix(qb, R2);
    QuatToRotationMatrix(qr, Rrel_d);
    T Rrel[9];
    for (int k = 0; k < 9; ++k) Rrel[k] = T(Rrel_d[k]);
    MatMul(Rrel, R1, est);
    for (int k = 0; k < 9; ++k) {
      const int r = k % 3, c = k / 3;
      residuals[k] = T(weight) * (est[3 * r + c] - R2[3 * r + c]);
    }
  }
};

}  // namespace gsfm_oracle

// HIP kernels of the rotation-averaging hot path (gfx950, fp64, HBM-bound; no MFMA: there is
// no dense contraction on this path).
//
// Data layout in HBM (all per-edge streams are planes of 16-byte chunks so that a wavefront's
// loads are 64 x 16 B = 1 KiB contiguous):
//   cameras        quats  q[k]  = 2 x double2 (x,y)(z,w)     gathered (L2 / Infinity-Cache resident)
//   cost edges     idx    uint2 (i, j)                        8 B / edge
//                  qrel   2 planes of double2                 32 B / edge
//                  W      0 | 1 plane double | 3 planes double2 (upper-triangular Lt) 0 / 8 / 48 B
//   directed rows  one entry per (owned camera k <- neighbour m) = the block-CSR structure of
//                  J^T J; per entry: col (u32, bit31 = role), qrel, W (copies), H block (72 B,
//                  4 planes double2 + 1 plane double) written by K2, read by K3.
//
// K1 k_cost      1 edge / lane            residual + robust reweight sweep, block-reduced cost
// K2 k_lin       G lanes / camera row     residual, 3x3 body Jacobians, Corrector, g, D, H blocks
// K3 k_matvec    G lanes / camera row     y = M p + sum_d H_d p[col_d]
// K4 k_cg_*      1 camera / lane          fused PCG vector updates + dot-product partials
// K5 k_cam_*     1 camera / lane          quaternion cache, LM diagonal / preconditioner, step
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "loss_dev.hpp"
#include "so3_dev.hpp"

namespace gsfm {

enum { F_AA = 0, F_QCOS = 1, F_QNORM = 2, F_RFNORM = 3 };
enum { W_NONE = 0, W_SCALAR = 1, W_MATRIX = 2, W_MATRIX3 = 3 /* W_MATRIX with the measurement planes as three quaternion components (qrel_three below): a kernel template value only, never a problem's wmode */ };

template <int F> struct ResDim { static constexpr int R = (F == F_QNORM) ? 4 : (F == F_RFNORM) ? 9 : 3; };

#define GSFM_BLOCK 256
#define GSFM_MAX_PARTIALS 1024
// K1 tiles: cost edges are bucketed by (camera block of `first`, camera block of `second`), 2048 cameras per block;
// a 1024-thread workgroup stages BOTH quaternion blocks in LDS (2 x 2048 x 32 B = 128 KiB of the 160 KiB), so the
// sweep performs no global gather at all.
#define GSFM_CAMBLOCK 2048
#ifndef GSFM_TILE_THREADS
#define GSFM_TILE_THREADS 1024
#endif
#ifndef GSFM_K2_ATTR
// K2 is latency-bound (profiles/r01_e_pmc_sq_valu.txt): at the compiler's free choice of 184 VGPRs only two waves fit a
// SIMD; asking for at least three costs no spill (168 VGPRs) and 7.5 % less time on C5.  Four would spill 37 VGPRs (2x slower).
#define GSFM_K2_ATTR __attribute__((amdgpu_waves_per_eu(3)))
#endif
#ifndef GSFM_GATHER_LOAD
#define GSFM_GATHER_LOAD(p) (*(p))   // tuning hook: e.g. __builtin_nontemporal_load(p)
#endif
#ifndef GSFM_K1_UNROLL
#define GSFM_K1_UNROLL 1   // edges per lane whose streams are requested before any of them is evaluated
#endif

// ------------------------------------------------------------------------------------------
// reductions (deterministic: fixed tree inside a wave, fixed order across waves)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_down(v, off, 64));
  return v;
}
// result valid in every thread
__device__ __forceinline__ double block_sum_bcast(double v, double* lds /* >= 5 */) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) lds[w] = v;
  __syncthreads();
  if (threadIdx.x == 0) { double t = 0; for (int k = 0; k < GSFM_BLOCK / 64; ++k) t += lds[k]; lds[4] = t; }
  __syncthreads();
  return lds[4];
}
__device__ __forceinline__ double block_max_bcast(double v, double* lds) {
  v = wave_max(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) lds[w] = v;
  __syncthreads();
  if (threadIdx.x == 0) { double t = lds[0]; for (int k = 1; k < GSFM_BLOCK / 64; ++k) t = fmax(t, lds[k]); lds[4] = t; }
  __syncthreads();
  return lds[4];
}
// every block sums the same `n` partials in the same order -> identical scalar in every block
__device__ __forceinline__ double sum_partials_bcast(const double* __restrict__ partials, int n, double* lds) {
  double v = 0.0;
  for (int k = threadIdx.x; k < n; k += GSFM_BLOCK) v += partials[k];
  return block_sum_bcast(v, lds);
}

// ------------------------------------------------------------------------------------------
// per-edge evaluation
// ------------------------------------------------------------------------------------------
struct EdgeW { double l00, l01, l02, l11, l12, l22; };  // upper-triangular whitening factor / scalar in l00

template <int WM>
__device__ __forceinline__ void apply_w_vec(const EdgeW& W, const double* e, double* r) {
  if (WM == W_NONE) { r[0] = e[0]; r[1] = e[1]; r[2] = e[2]; }
  else if (WM == W_SCALAR) { r[0] = W.l00 * e[0]; r[1] = W.l00 * e[1]; r[2] = W.l00 * e[2]; }
  else {
    r[0] = W.l00 * e[0] + W.l01 * e[1] + W.l02 * e[2];
    r[1] = W.l11 * e[1] + W.l12 * e[2];
    r[2] = W.l22 * e[2];
  }
}
template <int WM>
__device__ __forceinline__ void apply_w_mat(const EdgeW& W, const double* M, double* O) {  // O = W M
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    if (WM == W_NONE) { O[c] = M[c]; O[3 + c] = M[3 + c]; O[6 + c] = M[6 + c]; }
    else if (WM == W_SCALAR) { O[c] = W.l00 * M[c]; O[3 + c] = W.l00 * M[3 + c]; O[6 + c] = W.l00 * M[6 + c]; }
    else {
      O[c] = W.l00 * M[c] + W.l01 * M[3 + c] + W.l02 * M[6 + c];
      O[3 + c] = W.l11 * M[3 + c] + W.l12 * M[6 + c];
      O[6 + c] = W.l22 * M[6 + c];
    }
  }
}

// Residual only.  qi, qj: camera quaternions of (first, second); qr: measured R_ij.
template <int F, int WM>
__device__ __forceinline__ void edge_residual(const Quat& qi, const Quat& qj, const Quat& qr, const EdgeW& W, double* r) {
  if (F == F_AA) {
    // e = log(R_j R_i^T R_ij^T)   (Theia pairwise_rotation_error.h:80-92 / quat.hpp:231-246)
    const Quat qe = qmul(qmul(qj, qconj(qi)), qconj(qr));
    double e[3], s, th;
    quat_log(qe, e, &s, &th);
    apply_w_vec<WM>(W, e, r);
  } else if (F == F_QCOS) {
    // r = 2 vec(q_ij * (q_b * q_a^-1)^*)   (quat.hpp:86-103)
    const Quat dq = qmul(qr, qconj(qmul(qj, qconj(qi))));
    r[0] = 2.0 * dq.x; r[1] = 2.0 * dq.y; r[2] = 2.0 * dq.z;
  } else if (F == F_QNORM) {
    // r = canon(q_b) - canon(q_ij q_a), canon tests the y coefficient  (quat.hpp:130-147)
    const Quat est = qmul(qr, qi);
    const double sb = (qj.y < 0.0) ? -1.0 : 1.0, se = (est.y < 0.0) ? -1.0 : 1.0;
    r[0] = sb * qj.x - se * est.x; r[1] = sb * qj.y - se * est.y;
    r[2] = sb * qj.z - se * est.z; r[3] = sb * qj.w - se * est.w;
  } else {
    // r = vec_colmajor(R_ij R_a - R_b)   (quat.hpp:170-193)
    double Est[9], Rb[9];
    qmat(qmul(qr, qi), Est);
    qmat(qj, Rb);
#pragma unroll
    for (int k = 0; k < 9; ++k) { const int rr = k % 3, cc = k / 3; r[k] = Est[3 * rr + cc] - Rb[3 * rr + cc]; }
  }
}

__device__ __forceinline__ void plus_jac_half(const Quat& q, double sgn, double* P /*4x3*/) {
  // 1/2 * d((h,1) (x) q)/dh : rows x,y,z,w  (ceres EigenQuaternionParameterization::ComputeJacobian, eta = 2 delta)
  const double h = 0.5 * sgn;
  P[0] = h * q.w;  P[1] = h * q.z;   P[2] = -h * q.y;
  P[3] = -h * q.z; P[4] = h * q.w;   P[5] = h * q.x;
  P[6] = h * q.y;  P[7] = -h * q.x;  P[8] = h * q.w;
  P[9] = -h * q.x; P[10] = -h * q.y; P[11] = -h * q.z;
}

// Residual and body Jacobians A_i, A_j (R x 3, row-major) w.r.t. left perturbations of R_i, R_j.
template <int F, int WM>
__device__ __forceinline__ void edge_linearize(const Quat& qi, const Quat& qj, const Quat& qr, const EdgeW& W,
                                               double* r, double* Ai, double* Aj) {
  if (F == F_AA) {
    const Quat qe = qmul(qmul(qj, qconj(qi)), qconj(qr));
    double e[3], s, th;
    quat_log(qe, e, &s, &th);
    apply_w_vec<WM>(W, e, r);
    const double c = jlinv_coeff(th, s, fabs(qe.w));
    double B[9], Rij[9], Bt_R[9];
    jlinv_matrix(e, c, B);            // de/d eta_j = J_l^-1(e)
    apply_w_mat<WM>(W, B, Aj);
    qmat(qr, Rij);
    mat3_tmul(B, Rij, Bt_R);          // de/d eta_i = -J_r^-1(e) R_ij = -J_l^-1(e)^T R_ij
#pragma unroll
    for (int k = 0; k < 9; ++k) Bt_R[k] = -Bt_R[k];
    apply_w_mat<WM>(W, Bt_R, Ai);
  } else if (F == F_QCOS) {
    const Quat a = qmul(qr, qconj(qmul(qj, qconj(qi))));
    r[0] = 2.0 * a.x; r[1] = 2.0 * a.y; r[2] = 2.0 * a.z;
    Aj[0] = -a.w; Aj[1] = a.z;  Aj[2] = -a.y;
    Aj[3] = -a.z; Aj[4] = -a.w; Aj[5] = a.x;
    Aj[6] = a.y;  Aj[7] = -a.x; Aj[8] = -a.w;
    const double K[9] = {a.w, a.z, -a.y, -a.z, a.w, a.x, a.y, -a.x, a.w};
    double Rij[9];
    qmat(qr, Rij);
    mat3_mul(K, Rij, Ai);
  } else if (F == F_QNORM) {
    const Quat est = qmul(qr, qi);
    const double sb = (qj.y < 0.0) ? -1.0 : 1.0, se = (est.y < 0.0) ? -1.0 : 1.0;
    r[0] = sb * qj.x - se * est.x; r[1] = sb * qj.y - se * est.y;
    r[2] = sb * qj.z - se * est.z; r[3] = sb * qj.w - se * est.w;
    plus_jac_half(qj, sb, Aj);
    double P[12], Rij[9];
    plus_jac_half(est, -se, P);
    qmat(qr, Rij);
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int c = 0; c < 3; ++c) Ai[3 * k + c] = P[3 * k] * Rij[c] + P[3 * k + 1] * Rij[3 + c] + P[3 * k + 2] * Rij[6 + c];
  } else {
    double Est[9], Rb[9], Rij[9];
    qmat(qmul(qr, qi), Est);
    qmat(qj, Rb);
    qmat(qr, Rij);
#pragma unroll
    for (int cc = 0; cc < 3; ++cc) {
      const double u0 = Rb[cc], u1 = Rb[3 + cc], u2 = Rb[6 + cc];        // column cc of R_b
      const double v0 = Est[cc], v1 = Est[3 + cc], v2 = Est[6 + cc];     // column cc of R_ij R_a
      r[3 * cc] = v0 - u0; r[3 * cc + 1] = v1 - u1; r[3 * cc + 2] = v2 - u2;
      // d(-R_b col)/d eta_j = [u]x
      double* J = Aj + 9 * cc;
      J[0] = 0.0; J[1] = -u2; J[2] = u1;
      J[3] = u2;  J[4] = 0.0; J[5] = -u0;
      J[6] = -u1; J[7] = u0;  J[8] = 0.0;
      // d(Est col)/d eta_i = -[v]x R_ij
      const double K[9] = {0.0, v2, -v1, -v2, 0.0, v0, v1, -v0, 0.0};
      mat3_mul(K, Rij, Ai + 9 * cc);
    }
  }
}

// Ceres ResidualBlock::Evaluate + Corrector applied in place; returns 1/2 rho.
template <int R>
__device__ __forceinline__ void robustify(const Rho3& rho, double s, double* r, double* Ai, double* Aj) {
  const Corrector c = make_corrector(s, rho);
  if (c.alpha_sq_norm == 0.0) {
#pragma unroll
    for (int k = 0; k < 3 * R; ++k) { Ai[k] *= c.sqrt_rho1; Aj[k] *= c.sqrt_rho1; }
  } else {
#pragma unroll
    for (int col = 0; col < 3; ++col) {
      double ti = 0.0, tj = 0.0;
#pragma unroll
      for (int k = 0; k < R; ++k) { ti += Ai[3 * k + col] * r[k]; tj += Aj[3 * k + col] * r[k]; }
#pragma unroll
      for (int k = 0; k < R; ++k) {
        Ai[3 * k + col] = c.sqrt_rho1 * (Ai[3 * k + col] - c.alpha_sq_norm * r[k] * ti);
        Aj[3 * k + col] = c.sqrt_rho1 * (Aj[3 * k + col] - c.alpha_sq_norm * r[k] * tj);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < R; ++k) r[k] *= c.residual_scaling;
}

__device__ __forceinline__ double2 nt_load2(const double2* __restrict__ p) {
  double2 v;
  v.x = __builtin_nontemporal_load(&p->x);
  v.y = __builtin_nontemporal_load(&p->y);
  return v;
}
__device__ __forceinline__ void nt_store2(double2* __restrict__ p, double x, double y) {
  __builtin_nontemporal_store(x, &p->x);
  __builtin_nontemporal_store(y, &p->y);
}
template <int WM>
__device__ __forceinline__ EdgeW load_w(const double2* __restrict__ w0, const double2* __restrict__ w1,
                                        const double2* __restrict__ w2, const double* __restrict__ ws, size_t e) {
  EdgeW W;
  W.l00 = 1.0; W.l01 = W.l02 = W.l12 = 0.0; W.l11 = W.l22 = 1.0;
  if (WM == W_SCALAR) { W.l00 = __builtin_nontemporal_load(ws + e); }
  else if (WM == W_MATRIX || WM == W_MATRIX3) {
    const double2 a = nt_load2(w0 + e), b = nt_load2(w1 + e), c = nt_load2(w2 + e);
    W.l00 = a.x; W.l01 = a.y; W.l02 = b.x; W.l11 = b.y; W.l12 = c.x; W.l22 = c.y;
  }
  return W;
}
__device__ __forceinline__ Quat load_q(const double2* __restrict__ q2, uint32_t k) {
  const double2 a = q2[2 * (size_t)k], b = q2[2 * (size_t)k + 1];
  return Quat{a.x, a.y, b.x, b.y};
}

// ------------------------------------------------------------------------------------------
// The measured relative rotation of an edge / a directed entry: 24 bytes per position on covariance-whitened problems (round 6; SURVEY 8(d)
// counts the measurement at 24 B).  A unit quaternion is three numbers and a sign: the component of LARGEST magnitude (>= 1/2) is dropped and
// rebuilt as +-sqrt(1 - a^2 - b^2 - c^2); dropping the largest keeps the rebuilt one accurate to an ulp (dropping w outright would lose
// ~1e-16 / w: a third of the benchmark's edges are uniformly random rotations, |w| < 1e-4 on hundreds of them).  Which component was dropped
// (0..3 = x, y, z, w) rides in bit 62 of the first two stored doubles -- the top bit of the exponent field, zero for every |value| < 2 -- and
// its sign in bit 62 of the third (q and -q are the same rotation, but the quaternion-cosine residual, quat.hpp:86-103, carries the sign of
// q_ij into the sign of r: the stored quaternion is the one ceres::AngleAxisToQuaternion gives, not a normalised one).
// Planes: qr0 = (a, b) as double2, qr1 = c as double (the buffer is still handed around as a double2 pointer).
// WHERE: the W_MATRIX problems (ANGLE_AXIS_COVARIANCE / COV_INLIERS: 88 -> 80 B streamed per edge) of at least one million edges -- where the
// sweeps are bound by the stream (gsfm_rot_problem::q3, decided at create: such a problem launches the W_MATRIX3 instantiations; GSFM_QREL3=0/1 in the
// environment overrides).  Below that size the
// launches are bound by latency, the 8 bytes buy nothing, and the rebuilt component -- within an ulp of ceres::AngleAxisToQuaternion's, not always
// equal to it -- would move the small configurations' last bits for no gain: on the real Madrid graph under MAGSAC, whose outcome is bimodal under
// one-ulp changes of the measurements, it was enough to land the run in the other cluster (62 instead of 63 LM iterations, 2.0e-4
// rad from the unperturbed oracle: profiles/r06_bench_madrid_with_q3.json).  Measured at C5, same box, alternating
// (profiles/r06_qrel3_ab.txt): reweight sweep 180 -> 168.5 us (0.61 -> 0.65 of the roofline on SURVEY 8(d)'s bytes), full sweep 221 -> 213.5, s-only
// 172 -> 166.5, K2c 580-593 -> 583-587 (its own stream is not what binds it), the trial-cost sweep 137.4 -> 142.2 (it stores nothing and is bound
// by instruction issue: the ~45 VALU operations of the decode show), a whole solve 12.01 -> 11.94 ms with the final cost equal to the last bit.  The
// unit- and scalar-weight sweeps (40-48 B per edge) are bound by instruction issue throughout -- 98.6 -> 108.4 us with three components -- and
// keep the full quaternion, 32 B.  -DGSFM_QREL3=0: the full quaternion everywhere (rounds 1-5).
#ifndef GSFM_QREL3
#define GSFM_QREL3 1
#endif
// (A compile-time property of the kernels -- the template value W_MATRIX3 -- not a runtime branch inside the W_MATRIX ones: with the branch compiled
// in, the compiler scheduled the W_MATRIX kernels' arithmetic differently, last bits of the small configurations moved, and Madrid / MAGSAC -- bimodal
// under one-ulp changes, DESIGN section 2 -- landed in its other cluster, 62 instead of 63 LM iterations: profiles/r06_madrid_bits.txt.)
__host__ __device__ constexpr bool qrel_three(int wm) { return GSFM_QREL3 != 0 && wm == W_MATRIX3; }
__device__ __forceinline__ void qrel_encode(const Quat& q, double* ab_c /* [3] */) {
  const double v[4] = {q.x, q.y, q.z, q.w};
  if (!(isfinite(v[0]) && isfinite(v[1]) && isfinite(v[2]) && isfinite(v[3]))) { ab_c[0] = ab_c[1] = ab_c[2] = 1.5; return; }   // a non-finite measurement decodes to NaN (1 - 3 x 2.25 < 0)
  int k = 3;
  double m = fabs(v[3]);
#pragma unroll
  for (int c = 2; c >= 0; --c) if (fabs(v[c]) > m) { m = fabs(v[c]); k = c; }
  double o[3];
  int n = 0;
#pragma unroll
  for (int c = 0; c < 4; ++c) if (c != k) o[n++] = v[c];
  unsigned long long b0 = (unsigned long long)__double_as_longlong(o[0]), b1 = (unsigned long long)__double_as_longlong(o[1]), b2 = (unsigned long long)__double_as_longlong(o[2]);
  b0 |= (unsigned long long)(k & 1) << 62; b1 |= (unsigned long long)(k >> 1) << 62; b2 |= (unsigned long long)(v[k] < 0.0 ? 1 : 0) << 62;
  ab_c[0] = __longlong_as_double((long long)b0); ab_c[1] = __longlong_as_double((long long)b1); ab_c[2] = __longlong_as_double((long long)b2);
}
// raw: the full quaternion (x, y) (z, w) -- or, three components: (a, b) in r0, c in r1.x (r1.y unused)
template <int WM>
__device__ __forceinline__ Quat qrel_quat(const double2& r0, const double2& r1) {
  if constexpr (!qrel_three(WM)) return Quat{r0.x, r0.y, r1.x, r1.y};
  else {
    const unsigned ha = (unsigned)__double2hiint(r0.x), hb = (unsigned)__double2hiint(r0.y), hc = (unsigned)__double2hiint(r1.x);
    const unsigned k = ((ha >> 30) & 1u) | (((hb >> 30) & 1u) << 1);
    const double a = __hiloint2double((int)(ha & 0xbfffffffu), __double2loint(r0.x)), b = __hiloint2double((int)(hb & 0xbfffffffu), __double2loint(r0.y)),
                 c = __hiloint2double((int)(hc & 0xbfffffffu), __double2loint(r1.x));
    const double mp = sqrt(1.0 - a * a - b * b - c * c);   // (>= 1/4 for a unit quaternion; negative -> NaN for what qrel_encode made of a non-finite one)
    const double m = __hiloint2double(__double2hiint(mp) ^ (int)((hc << 1) & 0x80000000u), __double2loint(mp));   // (bit 30 of the third's high word -> the sign bit)
    Quat q;   // stored order = (x, y, z, w) with component k removed
    q.x = k == 0u ? m : a;
    q.y = k == 0u ? a : (k == 1u ? m : b);
    q.z = k == 3u ? c : (k == 2u ? m : b);
    q.w = k == 3u ? m : c;
    return q;
  }
}
template <int WM>
__device__ __forceinline__ void qrel_load_nt(const double2* __restrict__ qr0, const double2* __restrict__ qr1, size_t e, double2& r0, double2& r1) {
  r0 = nt_load2(qr0 + e);
  if constexpr (qrel_three(WM)) { r1.x = __builtin_nontemporal_load((const double*)qr1 + e); r1.y = 0.0; } else r1 = nt_load2(qr1 + e);
}
template <int WM>
__device__ __forceinline__ void qrel_load(const double2* __restrict__ qr0, const double2* __restrict__ qr1, size_t e, double2& r0, double2& r1) {
  r0 = qr0[e];
  if constexpr (qrel_three(WM)) { r1.x = ((const double*)qr1)[e]; r1.y = 0.0; } else r1 = qr1[e];
}

// ------------------------------------------------------------------------------------------
// K0': measured relative rotations, angle-axis -> unit quaternion planes, gathered into entry order on the device
// (ceres::AngleAxisToQuaternion, estimator.cpp:132; one upload of the 3E doubles instead of a host gather per entry)
__global__ void __launch_bounds__(GSFM_BLOCK) k_build_qrel(const double* __restrict__ rel_aa, const uint32_t* __restrict__ eid, size_t n,
                                                           double2* __restrict__ qr0, double2* __restrict__ qr1, int three) {
  const size_t t = (size_t)blockIdx.x * GSFM_BLOCK + threadIdx.x;
  if (t >= n) return;
  const double* aa = rel_aa + 3 * (size_t)eid[t];
  const Quat q = aa_to_quat(aa[0], aa[1], aa[2]);
  if (three) {   // (qrel_three: the W_MATRIX problems)
    double v[3];
    qrel_encode(q, v);
    qr0[t] = make_double2(v[0], v[1]);
    ((double*)qr1)[t] = v[2];
  } else {
    qr0[t] = make_double2(q.x, q.y);
    qr1[t] = make_double2(q.z, q.w);
  }
}

// ------------------------------------------------------------------------------------------
// K0: whitening precompute (src/GSfM_nonlinear_rotation_estimator.cpp:251-288), once per problem
// ------------------------------------------------------------------------------------------
// Lt of cov (already scaled by 1e8): P = cov^-1 by cofactors (Eigen's fixed-size 3x3 inverse), P = L L^T, Lt = L^T (upper triangular:
// l01 = L10, l02 = L20, l12 = L21)
__device__ __forceinline__ EdgeW whitening_factor(double c00, double c11, double c22, double c01, double c02, double c12) {
  const double k00 = c11 * c22 - c12 * c12;
  const double k10 = c12 * c02 - c01 * c22;
  const double k20 = c01 * c12 - c11 * c02;
  const double id = 1.0 / (c00 * k00 + c01 * k10 + c02 * k20);
  const double p00 = k00 * id, p10 = k10 * id, p20 = k20 * id;
  const double p11 = (c00 * c22 - c02 * c02) * id;
  const double p21 = (c02 * c01 - c00 * c12) * id;
  const double p22 = (c00 * c11 - c01 * c01) * id;
  EdgeW W;
  W.l00 = sqrt(p00);
  W.l01 = p10 / W.l00; W.l02 = p20 / W.l00;
  W.l11 = sqrt(p11 - W.l01 * W.l01);
  W.l12 = (p21 - W.l02 * W.l01) / W.l11;
  W.l22 = sqrt(p22 - W.l02 * W.l02 - W.l12 * W.l12);
  return W;
}
struct WhitenArgs {
  const double* cov6;      // per ORIGINAL edge, C00 C11 C22 C01 C02 C12 (may be null)
  const double* inl;       // per original edge (may be null)
  const uint32_t* eid;     // entry -> original edge
  size_t n;
  int error_type;
  double2 *w0, *w1, *w2;   // W_MATRIX outputs
  double* ws;              // W_SCALAR output
};
__global__ void __launch_bounds__(GSFM_BLOCK) k_whiten(WhitenArgs a) {
  const size_t t = (size_t)blockIdx.x * GSFM_BLOCK + threadIdx.x;
  if (t >= a.n) return;
  const size_t e = a.eid[t];
  double cov[6] = {0, 0, 0, 0, 0, 0};
  if (a.cov6) {
#pragma unroll
    for (int k = 0; k < 6; ++k) cov[k] = a.cov6[6 * e + k] * 1e8;  // :252
  }
  const double iw = a.inl ? a.inl[e] : 1.0;
  const double c00 = cov[0], c11 = cov[1], c22 = cov[2], c01 = cov[3], c02 = cov[4], c12 = cov[5];
  if (a.error_type == GSFM_ROT_ANGLE_AXIS_COVARIANCE || a.error_type == GSFM_ROT_ANGLE_AXIS_COV_INLIERS) {
    const EdgeW W = whitening_factor(c00, c11, c22, c01, c02, c12);
    const double m = (a.error_type == GSFM_ROT_ANGLE_AXIS_COV_INLIERS) ? iw : 1.0;
    a.w0[t] = make_double2(W.l00 * m, W.l01 * m);
    a.w1[t] = make_double2(W.l02 * m, W.l11 * m);
    a.w2[t] = make_double2(W.l12 * m, W.l22 * m);
  } else if (a.error_type == GSFM_ROT_ANGLE_AXIS_INLIERS) {
    a.ws[t] = iw;                                                     // :263
  } else if (a.error_type == GSFM_ROT_ANGLE_AXIS_COVTRACE) {
    a.ws[t] = sqrt(1.0 / (c00 + c11 + c22));                          // :276-281
  } else if (a.error_type == GSFM_ROT_ANGLE_AXIS_COVNORM) {
    const double f = c00 * c00 + c11 * c11 + c22 * c22 + 2.0 * (c01 * c01 + c02 * c02 + c12 * c12);
    a.ws[t] = sqrt(1.0 / sqrt(f));                                    // :284-286
  }
}

// sigma-consensus weights (src/GSfM_nonlinear_rotation_estimator.cpp:400-416).  The weight of an edge depends only on its UNWEIGHTED
// residual at the rotations an outer iteration starts from -- exactly the point the inner solve's first cost sweep (K1) and first
// linearisation (K2) evaluate anyway.  So there is no weight pass: in `sigma` mode K1 and K2 compute the weight from the unit-weight
// residual they have in registers, store it into their own scalar-weight plane in their own order (8 B per edge / directed entry,
// coalesced) and use it at once; K1 also sums |w - w_old| against the plane's previous content.  Later sweeps of the solve read the
// planes as usual.  (Round 2: an s-only sweep, a weight kernel over the original edge order and two scattered gathers, 478 us at C5.)
struct SigmaDev {
  const double* table;    // Gamma(1, x / 1000), nu = 3
  int table_len;
  int on;                 // 1: this launch computes and stores the weights
  double ssm2, one_over_sigma, gk, weight_zero;
  double inv_ssm2;        // 1 / ssm2 (the fast path of sigma_weight)
};
// The reference's arithmetic -- residual = sqrt(s), squared_residual = residual * residual, x = round(1000 * squared_residual / ssm2) -- costs a
// correctly rounded fp64 square root and division per edge (~45 VALU instructions: the sweep is issue-bound, round-4 SQ counters) for the sake
// of an INTEGER: the table cell.  Round 5: the cell is taken from t = 1000 s / ssm2 evaluated with one multiplication by the reciprocal,
// which is within a few ulp of the reference's argument of round() (sqrt-then-square moves s by at most 2 ulp, the reciprocal by 1.5), so it
// names the same cell unless t lies within ~1e-15 t of a half-integer; lanes closer than 1e-12 t to one (and the ones at the zero-residual
// test) redo it the reference's way.  Same cell -> the same table entry -> the same weight, bit for bit (tests/test_gpu_round3.py).
__device__ __forceinline__ double sigma_weight(const SigmaDev& g, double s_unit) {
  const double last = (double)(g.table_len - 1);
  const double t = 1000.0 * s_unit * g.inv_ssm2;
  double xf = round(t);                                                  // std::round: halves away from zero
  const double frac = fabs(t - xf);                                      // distance to the nearest integer: 0.5 at a cell boundary
  const bool sure = s_unit > 1e-30 && (t > last + 1.0 || fabs(frac - 0.5) > 1e-12 * fmax(t, 1.0));
  if (!sure) {
    const double residual = sqrt(s_unit);
    if (residual < 2.220446049250313e-16) return g.weight_zero;
    const double squared_residual = residual * residual;                 // as written in the reference, not s itself
    xf = round(1000.0 * squared_residual / g.ssm2);
  }
  if (!(xf < last)) xf = last;                                           // last stored entry (the reference reads one past it)
  return g.one_over_sigma * (g.table[(int)xf] - g.gk);
}

// The step after the solve and the evaluation statistic as one edge sweep with K1's device routines, without a problem object:
//   FilterViewPairsFromOrientation (Theia filter_view_pairs_from_orientation.cc:55-122): s_e = |log(R_ij^T R_j R_i^T)|^2 against a threshold;
//   residuals_of_relative_rot (src/compare_reconstructions.cpp:617-647): s_e = |Lt log(R_j R_i^T R_ij^T)|^2 with Lt from 1e8 Sigma_e.
// One edge per lane, edges in the caller's order (coalesced 8 + 24 (+ 48) B per edge in, 8 (+ 1) B out), camera quaternions gathered.
struct EdgeSweepArgs {
  size_t n;
  const uint32_t *ei, *ej;
  const double* rel_aa;   // 3 per edge
  const double* cov6;     // 6 per edge or null (unweighted)
  const double2* q;       // camera quaternion cache
  double max_sq;          // keep = s <= max_sq (ignored when keep is null)
  double* s_out;
  uint8_t* keep;
  unsigned long long* n_kept;
};
__global__ void __launch_bounds__(GSFM_BLOCK) k_edge_sweep(EdgeSweepArgs a) {
  const size_t e = (size_t)blockIdx.x * GSFM_BLOCK + threadIdx.x;
  bool kept = false;
  if (e < a.n) {
    const Quat qi = load_q(a.q, a.ei[e]), qj = load_q(a.q, a.ej[e]);
    const Quat qr = aa_to_quat(a.rel_aa[3 * e], a.rel_aa[3 * e + 1], a.rel_aa[3 * e + 2]);
    double r[3];
    if (a.cov6) {
      const double* c = a.cov6 + 6 * e;
      const EdgeW W = whitening_factor(c[0] * 1e8, c[1] * 1e8, c[2] * 1e8, c[3] * 1e8, c[4] * 1e8, c[5] * 1e8);
      edge_residual<F_AA, W_MATRIX>(qi, qj, qr, W, r);
    } else {
      EdgeW W;
      W.l00 = 1.0; W.l01 = W.l02 = W.l12 = 0.0; W.l11 = W.l22 = 1.0;
      edge_residual<F_AA, W_NONE>(qi, qj, qr, W, r);
    }
    const double s = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
    a.s_out[e] = s;
    kept = s <= a.max_sq;
    if (a.keep) a.keep[e] = kept ? 1 : 0;
  }
  if (a.keep) {   // integer count: order-independent, so an atomic is exact
    const unsigned long long b = __ballot(kept);
    if ((threadIdx.x & 63) == 0 && b) atomicAdd(a.n_kept, (unsigned long long)__popcll(b));
  }
}

// s = |r_e|^2 of EVERY edge a rank holds (each touches one of its rows), written per local edge, by rows of the block-CSR: on a sharded
// problem the cost sweep K1 only visits the edges a rank counts in the cost, but sigma consensus (unit weights, UNIT = true) and
// host-callback losses (the problem's own whitening) need s for both ends' rows.  Same device routine as K1, so two ranks holding the
// same edge -- and the single-GPU sweep -- produce the same bits.
struct RowSArgs {
  uint32_t n_rows, row_base, G;
  const uint32_t* row_ptr;
  const uint32_t* col;
  const uint32_t* eid;
  const double2 *qr0, *qr1;
  const double2 *w0, *w1, *w2;
  const double* ws;
  const double2* q;
  double* s_out;          // per local edge
};
template <int F, int WM, bool UNIT>
__global__ void __launch_bounds__(GSFM_BLOCK) k_row_s(RowSArgs a) {
  constexpr int R = ResDim<F>::R;
  const uint32_t t = blockIdx.x * GSFM_BLOCK + threadIdx.x;
  const uint32_t row = t / a.G, lane = t % a.G;
  if (row >= a.n_rows) return;
  const Quat qk = load_q(a.q, a.row_base + row);
  const uint32_t end = a.row_ptr[row + 1];
  for (uint32_t d = a.row_ptr[row] + lane; d < end; d += a.G) {
    const uint32_t cr = a.col[d];
    const Quat qm = load_q(a.q, cr & 0x7fffffffu);
    double2 r0, r1;
    qrel_load<WM>(a.qr0, a.qr1, d, r0, r1);
    const Quat qr = qrel_quat<WM>(r0, r1);
    EdgeW W = load_w<WM>(a.w0, a.w1, a.w2, a.ws, d);
    if (UNIT) W.l00 = 1.0;
    double r[R];
    if (cr >> 31) edge_residual<F, WM>(qm, qk, qr, W, r);
    else edge_residual<F, WM>(qk, qm, qr, W, r);
    double s = 0.0;
#pragma unroll
    for (int c = 0; c < R; ++c) s += r[c] * r[c];
    a.s_out[a.eid[d]] = s;
  }
}

// scatter per-original-edge scalar weights into an entry-ordered plane (sigma consensus / set_edge_weights)
__global__ void __launch_bounds__(GSFM_BLOCK) k_gather_weights(const double* __restrict__ w_orig,
                                                               const uint32_t* __restrict__ eid, size_t n,
                                                               double* __restrict__ ws) {
  const size_t t = (size_t)blockIdx.x * GSFM_BLOCK + threadIdx.x;
  if (t < n) ws[t] = w_orig[eid[t]];
}

// ------------------------------------------------------------------------------------------
// K1: residual + robust reweight sweep over the cost-owned edges
// ------------------------------------------------------------------------------------------
struct CostTile { uint32_t ib, jb, begin, end; };  // camera blocks of (first, second), edge range
struct CostArgs {
  const CostTile* tiles;     // one per workgroup
  uint32_t n_cams;
  size_t n;                  // edges, ordered by tile; idx holds BLOCK-LOCAL camera indices
  const uint2* idx;          // (i, j)
  const double2 *qr0, *qr1;  // q_rel planes (x,y) (z,w) -- or (a,b), c in the W_MATRIX3 kernels
  const double2 *w0, *w1, *w2;
  const double* ws;
  const double2* q;          // camera quaternions
  const DevLoss* loss;
  const double* rho_ext;     // external rho triples per ORIGINAL edge (callback path) or null
  const uint32_t* eid;       // entry -> original edge (rho_ext only)
  double* partials;          // [gridDim.x] sum of 1/2 rho; FULL kernels: [2 * gridDim.x], second half = sum |w - w_old| (sigma mode)
  // optional per-edge outputs (null in the solver loop), in the PROBLEM's edge order (= the order of the streamed planes; position u holds
  // original edge gsfm_rot_edge_order()[u]): every store is a coalesced non-temporal 8 / 16 B per lane.  (Round 2 stored through `eid`
  // into the caller's order: 438 MB written for 320 MB of payload, 0.43 of the HBM roofline.)
  double* s_out;             // s alone (s_only mode)
  double2* srho_out;         // (s, rho)           } the full sweep: two 16-byte stores per lane
  double2* rho12_out;        // (rho', rho'')      }
  double* rho1_out;          // rho' alone: the reweight sweep of SURVEY 8(d) (8 B out per edge)
  double* r_out;             // residuals, R planes of n
  int s_only;                // 1: write s_out only, skip the loss (callback path, phase 1)
  int direct;                // 1: k_cost_direct (idx = global camera indices, tiles = plain chunks)
  int unit_w;                // 1: ignore the scalar weight plane
  SigmaDev sigma;            // sigma consensus: compute, store (ws_rw) and use the weights
  double* ws_rw;             // = ws, writable
};

// K1.  FULL = false: the solver's trial-cost sweep (cost only: for MAGSAC the value needs no exp and no division
// by constants).  FULL = true: per-edge outputs / external rho / s-only / sigma modes.  One edge per lane; the seven streamed planes
// are 16-byte coalesced, non-temporal loads; both camera quaternions come from LDS.  Measured (tools/bench_cost*.hip, C5): streams only
// 137 us; + all arithmetic 138-152 us (hidden); direct global gathers 181 us; these 2-D LDS tiles 160 us.
// One edge of K1: residual, s, loss, optional per-edge outputs; returns the edge's 1/2 rho (0 in s_only mode).
template <int F, int WM, int LM, int MODE>
__device__ __forceinline__ double cost_edge(const CostArgs& a, const LossView<LM>& lv, uint32_t e, const Quat& qi, const Quat& qj, const Quat& qr, EdgeW W, double& dw_acc) {
  constexpr int R = ResDim<F>::R;
  constexpr bool FULL = MODE == 1;   // MODE 0: cost only (trial sweeps); 1: every optional output, sigma consensus, host-callback rho; 2: the reweight sweep (rho' stored)
  double r[R];
  if (FULL && F == F_AA && WM == W_SCALAR && a.sigma.on) {
    const double w_old = W.l00;
    W.l00 = 1.0;
    edge_residual<F, WM>(qi, qj, qr, W, r);
    const double w = sigma_weight(a.sigma, r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    dw_acc += fabs(w - w_old);
    __builtin_nontemporal_store(w, a.ws_rw + e);
#pragma unroll
    for (int k = 0; k < R; ++k) r[k] *= w;
  } else {
    edge_residual<F, WM>(qi, qj, qr, W, r);
  }
  double s = 0.0;
#pragma unroll
  for (int k = 0; k < R; ++k) s += r[k] * r[k];
  if (MODE == 0) return 0.5 * loss_value<LM>(lv, s);
  if (MODE == 2) {   // SURVEY 8(d)'s reweight sweep and nothing else: residual, loss, rho' out (8 B, coalesced, non-temporal), no run-time option in the way
    const Rho3 rho = loss_eval<LM>(lv, s);
    __builtin_nontemporal_store(rho.r1, a.rho1_out + e);
    return 0.5 * rho.r0;
  }
  if (a.s_only) { __builtin_nontemporal_store(s, a.s_out + e); return 0.0; }
  Rho3 rho;
  if (a.rho_ext) { const size_t o = 3 * (size_t)a.eid[e]; rho.r0 = a.rho_ext[o]; rho.r1 = a.rho_ext[o + 1]; rho.r2 = a.rho_ext[o + 2]; }
  else rho = loss_eval<LM>(lv, s);
  if (a.srho_out) nt_store2(a.srho_out + e, s, rho.r0);
  if (a.rho12_out) nt_store2(a.rho12_out + e, rho.r1, rho.r2);
  if (a.rho1_out) __builtin_nontemporal_store(rho.r1, a.rho1_out + e);
  if (a.r_out) {
#pragma unroll
    for (int k = 0; k < R; ++k) __builtin_nontemporal_store(r[k], a.r_out + (size_t)k * a.n + e);
  }
  return 0.5 * rho.r0;
}

template <int F, int WM, int LM, int MODE>
__global__ void __launch_bounds__(GSFM_TILE_THREADS) k_cost(CostArgs a) {
  __shared__ double2 qi_xy[GSFM_CAMBLOCK], qi_zw[GSFM_CAMBLOCK], qj_xy[GSFM_CAMBLOCK], qj_zw[GSFM_CAMBLOCK];
  __shared__ double lds[GSFM_TILE_THREADS / 64 + 1];
  const CostTile tile = a.tiles[blockIdx.x];
  const LossView<LM> lv = loss_view<LM>(a.loss);   // (before the first store: scalar loads, see loss_dev.hpp)
  {
    const uint32_t bi = tile.ib * GSFM_CAMBLOCK, bj = tile.jb * GSFM_CAMBLOCK;
    const uint32_t ci = min((uint32_t)GSFM_CAMBLOCK, a.n_cams - bi), cj = min((uint32_t)GSFM_CAMBLOCK, a.n_cams - bj);
    for (uint32_t c = threadIdx.x; c < ci; c += GSFM_TILE_THREADS) { qi_xy[c] = a.q[2 * (size_t)(bi + c)]; qi_zw[c] = a.q[2 * (size_t)(bi + c) + 1]; }
    for (uint32_t c = threadIdx.x; c < cj; c += GSFM_TILE_THREADS) { qj_xy[c] = a.q[2 * (size_t)(bj + c)]; qj_zw[c] = a.q[2 * (size_t)(bj + c) + 1]; }
  }
  __syncthreads();
  double acc = 0.0, dw = 0.0;
  constexpr int U = GSFM_K1_UNROLL;
  for (uint32_t e0 = tile.begin + threadIdx.x; e0 < tile.end; e0 += U * GSFM_TILE_THREADS) {
    // request phase: the streams of U edges are in flight before the first residual is evaluated
    uint2 ij[U];
    double2 r0[U], r1[U];
    EdgeW Wm[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t eu = e0 + u * GSFM_TILE_THREADS;
      const uint32_t e = eu < tile.end ? eu : e0;   // lanes past the end re-read their first edge and discard it
      ij[u] = a.idx[e];
      qrel_load_nt<WM>(a.qr0, a.qr1, e, r0[u], r1[u]);
      Wm[u] = load_w<WM>(a.w0, a.w1, a.w2, a.ws, e);
      if (WM == W_SCALAR && a.unit_w) Wm[u].l00 = 1.0;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t e = e0 + u * GSFM_TILE_THREADS;
      if (e >= tile.end) continue;
      const Quat qr = qrel_quat<WM>(r0[u], r1[u]);
      const double2 i0 = qi_xy[ij[u].x], i1 = qi_zw[ij[u].x], j0 = qj_xy[ij[u].y], j1 = qj_zw[ij[u].y];
      const Quat qi{i0.x, i0.y, i1.x, i1.y}, qj{j0.x, j0.y, j1.x, j1.y};
      acc += cost_edge<F, WM, LM, MODE>(a, lv, e, qi, qj, qr, Wm[u], dw);
    }
  }
  // deterministic block reduction (fixed tree per wave, fixed order over the 16 waves)
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int k = 0; k < GSFM_TILE_THREADS / 64; ++k) t += lds[k];
    a.partials[blockIdx.x] = t;
  }
  if (MODE == 1) {   // sum |w - w_old| of the sigma mode (zero otherwise)
    dw = wave_sum(dw);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = dw;
    __syncthreads();
    if (threadIdx.x == 0) {
      double t = 0.0;
      for (int k = 0; k < GSFM_TILE_THREADS / 64; ++k) t += lds[k];
      a.partials[gridDim.x + blockIdx.x] = t;
    }
  }
}

// K1 without LDS staging, for sweeps whose (first block, second block) tiles are too thinly populated to amortise the
// 128 KiB fill -- many cameras at a fixed degree (edges per tile = degree * 2048^2 / cameras), or one rank's share of a
// sharded problem.  Same edge order (so a chunk's gathers fall into few 64 KiB windows of q), `idx` holds GLOBAL camera
// indices, the quaternions are gathered through L1/L2; 256 lanes per workgroup, no LDS, so the occupancy is VGPR-bound.
template <int F, int WM, int LM, int MODE>
__global__ void __launch_bounds__(GSFM_BLOCK) k_cost_direct(CostArgs a) {
  __shared__ double lds[GSFM_BLOCK / 64 + 1];
  const CostTile tile = a.tiles[blockIdx.x];
  const LossView<LM> lv = loss_view<LM>(a.loss);
  double acc = 0.0, dw = 0.0;
  for (uint32_t e = tile.begin + threadIdx.x; e < tile.end; e += GSFM_BLOCK) {
    const uint2 ij = a.idx[e];
    double2 r0, r1;
    qrel_load_nt<WM>(a.qr0, a.qr1, e, r0, r1);
    EdgeW W = load_w<WM>(a.w0, a.w1, a.w2, a.ws, e);
    if (WM == W_SCALAR && a.unit_w) W.l00 = 1.0;
    const Quat qi = load_q(a.q, ij.x), qj = load_q(a.q, ij.y);
    acc += cost_edge<F, WM, LM, MODE>(a, lv, e, qi, qj, qrel_quat<WM>(r0, r1), W, dw);
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int k = 0; k < GSFM_BLOCK / 64; ++k) t += lds[k];
    a.partials[blockIdx.x] = t;
  }
  if (MODE == 1) {
    dw = wave_sum(dw);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = dw;
    __syncthreads();
    if (threadIdx.x == 0) {
      double t = 0.0;
      for (int k = 0; k < GSFM_BLOCK / 64; ++k) t += lds[k];
      a.partials[gridDim.x + blockIdx.x] = t;
    }
  }
}

// The loss program on given squared norms, through the very routines the sweeps use (device-level pin against the reference's
// recorded (s, rho, rho', rho'') vectors): rho3 = loss_eval<LM> (K2's general path and the FULL sweep), val = loss_value<LM> (the
// cost-only sweep), rho1 = loss_rho1<LM> (K2's fast path; LM_SIMPLE / LM_MAGSAC only).
template <int LM>
__global__ void __launch_bounds__(GSFM_BLOCK) k_loss_eval(const DevLoss* __restrict__ loss, const double* __restrict__ s, size_t n,
                                                          double* __restrict__ rho3, double* __restrict__ val, double* __restrict__ rho1) {
  const size_t t = (size_t)blockIdx.x * GSFM_BLOCK + threadIdx.x;
  if (t >= n) return;
  const double sq = s[t];
  if (rho3) { const Rho3 r = loss_eval<LM>(loss, sq); rho3[3 * t] = r.r0; rho3[3 * t + 1] = r.r1; rho3[3 * t + 2] = r.r2; }
  if (val) val[t] = loss_value<LM>(loss, sq);
  if (LM != LM_PROGRAM && rho1) rho1[t] = loss_rho1<LM>(loss, sq);
}

// out[0] = sum partials (single block, fixed order)
__global__ void __launch_bounds__(GSFM_BLOCK) k_sum_partials(const double* __restrict__ partials, int n, double* out) {
  __shared__ double lds[8];
  const double t = sum_partials_bcast(partials, n, lds);
  if (threadIdx.x == 0) out[0] = t;
}

// ------------------------------------------------------------------------------------------
// K2: linearise.  G lanes cooperate on one camera row of the block-CSR J^T J.
// ------------------------------------------------------------------------------------------
struct LinArgs {
  uint32_t n_rows;           // owned rows
  uint32_t row_base;         // global camera index of row 0
  uint32_t G;                // lanes per row (power of two, <= 64)
  const uint32_t* row_ptr;   // [n_rows + 1]
  const uint32_t* col;       // neighbour camera | role << 31 (role 1: the row camera is `second`)
  const uint32_t* eid;
  const double2 *qr0, *qr1;
  const double2 *w0, *w1, *w2;
  const double* ws;
  const double2* q;
  const DevLoss* loss;
  const double* rho_ext;
  double2 *h0, *h1, *h2, *h3;  // H block planes (row-major 3x3: h0=(H00,H01) h1=(H02,H10) h2=(H11,H12) h3=(H20,H21))
  double* h4;                  // H22
  double* gD;                  // 9 per camera: g(3), D sym(6: d00 d01 d02 d11 d12 d22)
  int lap;                     // 1: Laplacian form, planes h0..h2 hold the symmetric edge weight B (see lin_rows)
  const double* go;            // non-null: the launch is predicated -- it does nothing unless *go != 0 (device-side LM control: the step was accepted)
  int fast_ok;                 // host decision: the alpha = 0 fast path may be taken (kind and parameter signs of the loss checked in prepare_loss)
  SigmaDev sigma;              // sigma consensus: compute the weight of every directed entry from its unit-weight residual, store it
  double* ws_rw;               //   into the weight plane (= ws, writable) and use it
};

// LAP = true ("Laplacian form", functors that depend on R_j R_i^T only: angle-axis and quaternion-cosine): for those
// J_i = -J_j Q with Q = R_j R_i^T exactly (also after the Corrector, which multiplies both blocks from the left), hence
//   H_jj = G, H_ji = -G Q, H_ij = -Q^T G, H_ii = Q^T G Q   with G = J_j^T J_j,
// i.e. every off-diagonal block is the row camera's own symmetric G_k = J_k^T J_k times a rotation:
//   H_km p_m = -G_k R_k (R_m^T p_m)   =>   y_k = M_k p_k - sum_{d in row k} G_d (R_k u[col_d]),   u_m = R_m^T p_m.
// (In the body frame B = R_k^T G_k R_k is the same matrix from either end of the edge: a graph Laplacian with one symmetric
// 3x3 weight per edge.)  K2 then stores 6 doubles per directed entry instead of 9 (planes h0..h2), needs no neighbour
// Jacobian, and K3 streams 52 B per entry instead of 76; the rotation by the row's R_k is nine FMAs K3 has room for.
template <int F, int WM, int LM, bool LAP>
__device__ __forceinline__ void lin_rows(const LinArgs& a) {
  if (a.go && *a.go == 0.0) return;
  const LossView<LM> lv = loss_view<LM>(a.loss);   // (before the first store: scalar loads, see loss_dev.hpp)
  constexpr int R = ResDim<F>::R;
  const uint32_t G = a.G;
  const uint32_t t = blockIdx.x * GSFM_BLOCK + threadIdx.x;
  const uint32_t row = t / G, lane = t % G;
  const bool live = row < a.n_rows;
  double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  if (live) {
    const uint32_t k = a.row_base + row;
    const Quat qk = load_q(a.q, k);
    const uint32_t end = a.row_ptr[row + 1];
    for (uint32_t d = a.row_ptr[row] + lane; d < end; d += G) {
      const uint32_t cr = __builtin_nontemporal_load(a.col + d);
      const uint32_t m = cr & 0x7fffffffu;
      const bool row_is_second = (cr >> 31) != 0;
      double2 r0, r1;
      qrel_load_nt<WM>(a.qr0, a.qr1, d, r0, r1);
      const Quat qr = qrel_quat<WM>(r0, r1);
      EdgeW W = load_w<WM>(a.w0, a.w1, a.w2, a.ws, d);
      const bool sig = F == F_AA && WM == W_SCALAR && a.sigma.on;
      if (sig) W.l00 = 1.0;
      const Quat qm = load_q(a.q, m);
      double r[R], Ai[3 * R], Aj[3 * R];
      if (row_is_second) edge_linearize<F, WM>(qm, qk, qr, W, r, Ai, Aj);
      else edge_linearize<F, WM>(qk, qm, qr, W, r, Ai, Aj);
      if (sig) {   // r, Ai, Aj are unweighted here: the weight comes from |e|^2 and multiplies all three
        double su = 0.0;
#pragma unroll
        for (int c = 0; c < R; ++c) su += r[c] * r[c];
        const double w = sigma_weight(a.sigma, su);
        __builtin_nontemporal_store(w, a.ws_rw + d);
#pragma unroll
        for (int c = 0; c < R; ++c) r[c] *= w;
#pragma unroll
        for (int c = 0; c < 3 * R; ++c) { Ai[c] *= w; Aj[c] *= w; }
      }
      double s = 0.0;
#pragma unroll
      for (int c = 0; c < R; ++c) s += r[c] * r[c];
      Rho3 rho;
      if (a.rho_ext) { const size_t o = 3 * (size_t)a.eid[d]; rho.r0 = a.rho_ext[o]; rho.r1 = a.rho_ext[o + 1]; rho.r2 = a.rho_ext[o + 2]; }
      else rho = loss_eval<LM>(lv, s);
      robustify<R>(rho, s, r, Ai, Aj);
      const double* Ar = row_is_second ? Aj : Ai;  // Jacobian of the row camera
      const double* Ac = row_is_second ? Ai : Aj;  // Jacobian of the neighbour (dead code when LAP)
#pragma unroll
      for (int x = 0; x < 3; ++x) {
        double g = 0.0;
#pragma unroll
        for (int c = 0; c < R; ++c) g += Ar[3 * c + x] * r[c];
        acc[x] += g;
      }
      double d00 = 0, d01 = 0, d02 = 0, d11 = 0, d12 = 0, d22 = 0;
#pragma unroll
      for (int c = 0; c < R; ++c) {
        const double x0 = Ar[3 * c], x1 = Ar[3 * c + 1], x2 = Ar[3 * c + 2];
        d00 += x0 * x0; d01 += x0 * x1; d02 += x0 * x2; d11 += x1 * x1; d12 += x1 * x2; d22 += x2 * x2;
      }
      acc[3] += d00; acc[4] += d01; acc[5] += d02; acc[6] += d11; acc[7] += d12; acc[8] += d22;
      // streamed out once, read back by K3: non-temporal stores avoid the write-allocate fetch that PMC showed
      // (FETCH_SIZE of this kernel was 1.8x its algorithmic reads, profiles/r01_c_pmc_hbm_traffic.txt)
      if (LAP) {
        // G = J_k^T J_k of the row camera, as is: H_km p_m = -G_k R_k (R_m^T p_m), the rotation by R_k is applied by K3
        nt_store2(a.h0 + d, d00, d01);
        nt_store2(a.h1 + d, d02, d11);
        nt_store2(a.h2 + d, d12, d22);
      } else {
        double H[9];
#pragma unroll
        for (int x = 0; x < 3; ++x) {
#pragma unroll
          for (int y = 0; y < 3; ++y) {
            double h = 0.0;
#pragma unroll
            for (int c = 0; c < R; ++c) h += Ar[3 * c + x] * Ac[3 * c + y];
            H[3 * x + y] = h;
          }
        }
        nt_store2(a.h0 + d, H[0], H[1]);
        nt_store2(a.h1 + d, H[2], H[3]);
        nt_store2(a.h2 + d, H[4], H[5]);
        nt_store2(a.h3 + d, H[6], H[7]);
        __builtin_nontemporal_store(H[8], a.h4 + d);
      }
    }
  }
  // segmented reduction over the G lanes of the row (rows are G-aligned inside the wavefront)
  for (uint32_t off = G >> 1; off > 0; off >>= 1) {
#pragma unroll
    for (int c = 0; c < 9; ++c) acc[c] += __shfl_down(acc[c], off, G);
  }
  if (live && lane == 0) {
    double* o = a.gD + 9 * (size_t)(a.row_base + row);
#pragma unroll
    for (int c = 0; c < 9; ++c) o[c] = acc[c];
  }
}
// K2, fast path: Laplacian form + a loss whose rho'' is never positive (every LM_SIMPLE leaf, nu = 3 MAGSAC) + no host callback.
// Ceres' Corrector then takes its alpha = 0 branch for EVERY edge (corrector.cc: `if ((sq_norm == 0.0) || (rho[2] <= 0.0))`): residual and
// Jacobians are scaled by sqrt(rho') and nothing else.  So this path evaluates, per directed entry, only what the row needs:
//   * the row camera's Jacobian alone (the neighbour's is never formed; the general path computes both and discards one),
//   * rho' alone (loss_rho1: for MAGSAC one exp and one exact division instead of two exp, a table gather and nine divisions),
//   * g and G = J^T J from the unscaled Jacobian, multiplied by rho' at the end (no sqrt).
// About a third fewer VALU instructions per entry than the general path (round 2: 891, 22 of them IEEE divisions); C5: 914 -> 710 us.
// Same values as lin_rows up to the rounding of sqrt(rho')^2 vs rho'.
struct LinStreams { double2 r0, r1; EdgeW W; };
template <int WM>
__device__ __forceinline__ LinStreams lin_load_streams(const LinArgs& a, uint32_t d) {
  LinStreams S;
  qrel_load_nt<WM>(a.qr0, a.qr1, d, S.r0, S.r1);
  S.W = load_w<WM>(a.w0, a.w1, a.w2, a.ws, d);
  return S;
}
// one directed entry: residual r, row-camera Jacobian Ar (3x3 row-major); returns nothing else
template <int F, int WM>
__device__ __forceinline__ void edge_lin_row(const Quat& qk, const Quat& qm, const Quat& qr, EdgeW& W, bool row_is_second,
                                             const SigmaDev& sg, double* ws_slot, double* r, double* Ar) {
  const Quat qi = row_is_second ? qm : qk, qj = row_is_second ? qk : qm;
  if (F == F_AA) {
    const Quat qe = qmul(qmul(qj, qconj(qi)), qconj(qr));
    double e[3], s, th;
    quat_log<true>(qe, e, &s, &th);
    if (WM == W_SCALAR && sg.on) {   // sigma consensus: e is the unit-weight residual
      W.l00 = sigma_weight(sg, e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
      __builtin_nontemporal_store(W.l00, ws_slot);
    }
    apply_w_vec<WM>(W, e, r);
    const double c = jlinv_coeff(th, s, fabs(qe.w));
    double B[9], Mx[9];
    jlinv_matrix(e, c, B);                 // de/d eta_j = J_l^-1(e)
    if (row_is_second) {
#pragma unroll
      for (int k = 0; k < 9; ++k) Mx[k] = B[k];
    } else {                               // de/d eta_i = -J_l^-1(e)^T R_ij
      double Rij[9], T[9];
      qmat(qr, Rij);
      mat3_tmul(B, Rij, T);
#pragma unroll
      for (int k = 0; k < 9; ++k) Mx[k] = -T[k];
    }
    apply_w_mat<WM>(W, Mx, Ar);
  } else {  // F_QCOS
    const Quat a = qmul(qr, qconj(qmul(qj, qconj(qi))));
    r[0] = 2.0 * a.x; r[1] = 2.0 * a.y; r[2] = 2.0 * a.z;
    if (row_is_second) {
      Ar[0] = -a.w; Ar[1] = a.z;  Ar[2] = -a.y;
      Ar[3] = -a.z; Ar[4] = -a.w; Ar[5] = a.x;
      Ar[6] = a.y;  Ar[7] = -a.x; Ar[8] = -a.w;
    } else {
      const double K[9] = {a.w, a.z, -a.y, -a.z, a.w, a.x, a.y, -a.x, a.w};
      double Rij[9];
      qmat(qr, Rij);
      mat3_mul(K, Rij, Ar);
    }
  }
}
// g (3) and G = J_k^T J_k (6: 00 01 02 11 12 22) of one directed entry, Corrector applied.  FAST: the rho'' <= 0 path above; otherwise the
// general one (both Jacobians, full Corrector, host-callback rho) restricted to the row camera's block -- the Laplacian form needs no more.
template <int F, int WM, int LM, bool FAST>
__device__ __forceinline__ void lin_entry_eval(const LinArgs& a, const LossView<LM>& lv, uint32_t d, uint32_t cr, const Quat& qk, const Quat& qm, LinStreams S, double* g3, double* G6) {
  const Quat qr = qrel_quat<WM>(S.r0, S.r1);
  const bool row_is_second = (cr >> 31) != 0;
  if (FAST) {
    double r[3], Ar[9];
    edge_lin_row<F, WM>(qk, qm, qr, S.W, row_is_second, a.sigma, a.ws_rw + d, r, Ar);
    const double rho1 = loss_rho1<LM>(lv, r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
#pragma unroll
    for (int x = 0; x < 3; ++x) g3[x] = rho1 * (Ar[x] * r[0] + Ar[3 + x] * r[1] + Ar[6 + x] * r[2]);
    G6[0] = rho1 * (Ar[0] * Ar[0] + Ar[3] * Ar[3] + Ar[6] * Ar[6]); G6[1] = rho1 * (Ar[0] * Ar[1] + Ar[3] * Ar[4] + Ar[6] * Ar[7]);
    G6[2] = rho1 * (Ar[0] * Ar[2] + Ar[3] * Ar[5] + Ar[6] * Ar[8]); G6[3] = rho1 * (Ar[1] * Ar[1] + Ar[4] * Ar[4] + Ar[7] * Ar[7]);
    G6[4] = rho1 * (Ar[1] * Ar[2] + Ar[4] * Ar[5] + Ar[7] * Ar[8]); G6[5] = rho1 * (Ar[2] * Ar[2] + Ar[5] * Ar[5] + Ar[8] * Ar[8]);
  } else {
    constexpr int R = ResDim<F>::R;
    EdgeW W = S.W;
    const bool sig = F == F_AA && WM == W_SCALAR && a.sigma.on;
    if (sig) W.l00 = 1.0;
    double r[R], Ai[3 * R], Aj[3 * R];
    if (row_is_second) edge_linearize<F, WM>(qm, qk, qr, W, r, Ai, Aj);
    else edge_linearize<F, WM>(qk, qm, qr, W, r, Ai, Aj);
    if (sig) {
      double su = 0.0;
#pragma unroll
      for (int c = 0; c < R; ++c) su += r[c] * r[c];
      const double w = sigma_weight(a.sigma, su);
      __builtin_nontemporal_store(w, a.ws_rw + d);
#pragma unroll
      for (int c = 0; c < R; ++c) r[c] *= w;
#pragma unroll
      for (int c = 0; c < 3 * R; ++c) { Ai[c] *= w; Aj[c] *= w; }
    }
    double s = 0.0;
#pragma unroll
    for (int c = 0; c < R; ++c) s += r[c] * r[c];
    Rho3 rho;
    if (a.rho_ext) { const size_t o = 3 * (size_t)a.eid[d]; rho.r0 = a.rho_ext[o]; rho.r1 = a.rho_ext[o + 1]; rho.r2 = a.rho_ext[o + 2]; }
    else rho = loss_eval<LM>(lv, s);
    robustify<R>(rho, s, r, Ai, Aj);
    const double* Ar = row_is_second ? Aj : Ai;
#pragma unroll
    for (int x = 0; x < 3; ++x) {
      double g = 0.0;
#pragma unroll
      for (int c = 0; c < R; ++c) g += Ar[3 * c + x] * r[c];
      g3[x] = g;
    }
    double d00 = 0, d01 = 0, d02 = 0, d11 = 0, d12 = 0, d22 = 0;
#pragma unroll
    for (int c = 0; c < R; ++c) {
      const double x0 = Ar[3 * c], x1 = Ar[3 * c + 1], x2 = Ar[3 * c + 2];
      d00 += x0 * x0; d01 += x0 * x1; d02 += x0 * x2; d11 += x1 * x1; d12 += x1 * x2; d22 += x2 * x2;
    }
    G6[0] = d00; G6[1] = d01; G6[2] = d02; G6[3] = d11; G6[4] = d12; G6[5] = d22;
  }
}
// The same entry in the BODY frame (K2c's fast path, angle-axis family): with E = Exp(e) = R_j R_i^T R_ij^T one has J_l^-1(e)^T = J_l^-1(e) E and
// R_ij R_i = E^T R_j, so the first camera's Jacobian -J_l^-1(e)^T R_ij, taken to its body frame, is -J_l^-1(e) R_j -- the NEGATIVE of the
// second camera's body-frame Jacobian A = W J_l^-1(e) R_j.  One formula serves both roles (no R_ij matrix, no role-dependent branch), the block
// B = rho' A^T A comes out in the body frame K3c wants (no R_k^T G R_k afterwards), and the row sums are rotated ONCE per row by the finishing
// kernel: g_k = R_k sum(gb), D_k = R_k (sum B) R_k^T.  ~57 multiply-adds fewer per entry than lin_entry_eval<FAST> + the conjugation.
#ifndef GSFM_K2C_SC
#define GSFM_K2C_SC true   // K2c's transcendental coefficients in scalar registers (frees ~60 VGPRs; A/B: -DGSFM_K2C_SC=false)
#endif
template <int WM, int LM>
__device__ __forceinline__ void lin_entry_body_aa(const LinArgs& a, const LossView<LM>& lv, uint32_t d, uint32_t cr, const Quat& qk, const Quat& qm, LinStreams S, double* gb3, double* B6) {
  const Quat qr = qrel_quat<WM>(S.r0, S.r1);
  const bool row_is_second = (cr >> 31) != 0;
  const Quat qi = row_is_second ? qm : qk, qj = row_is_second ? qk : qm;
  const Quat qe = qmul(qmul(qj, qconj(qi)), qconj(qr));
  double e[3], s, th, r[3];
  quat_log<GSFM_K2C_SC>(qe, e, &s, &th);
  if (WM == W_SCALAR && a.sigma.on) {   // sigma consensus: e is the unit-weight residual
    S.W.l00 = sigma_weight(a.sigma, e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
    __builtin_nontemporal_store(S.W.l00, a.ws_rw + d);
  }
  apply_w_vec<WM>(S.W, e, r);
  double Jm[9], Rj[9], JR[9], A[9];
  jlinv_matrix(e, jlinv_coeff(th, s, fabs(qe.w)), Jm);
  qmat(qj, Rj);
  mat3_mul(Jm, Rj, JR);
  apply_w_mat<WM>(S.W, JR, A);
  const double rho1 = loss_rho1<LM, GSFM_K2C_SC>(lv, r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
  const double sg = row_is_second ? rho1 : -rho1;
#pragma unroll
  for (int x = 0; x < 3; ++x) gb3[x] = sg * (A[x] * r[0] + A[3 + x] * r[1] + A[6 + x] * r[2]);
  B6[0] = rho1 * (A[0] * A[0] + A[3] * A[3] + A[6] * A[6]); B6[1] = rho1 * (A[0] * A[1] + A[3] * A[4] + A[6] * A[7]);
  B6[2] = rho1 * (A[0] * A[2] + A[3] * A[5] + A[6] * A[8]); B6[3] = rho1 * (A[1] * A[1] + A[4] * A[4] + A[7] * A[7]);
  B6[4] = rho1 * (A[1] * A[2] + A[4] * A[5] + A[7] * A[8]); B6[5] = rho1 * (A[2] * A[2] + A[5] * A[5] + A[8] * A[8]);
}
template <int F, int WM, int LM>
__device__ __forceinline__ void lin_rows_fast(const LinArgs& a) {
  if (a.go && *a.go == 0.0) return;
  const LossView<LM> lv = loss_view<LM>(a.loss);   // (before the first store: scalar loads, see loss_dev.hpp)
  const uint32_t G = a.G;
  const uint32_t t = blockIdx.x * GSFM_BLOCK + threadIdx.x;
  const uint32_t row = t / G, lane = t % G;
  const bool live = row < a.n_rows;
  double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  if (live) {
    const Quat qk = load_q(a.q, a.row_base + row);
    const uint32_t end = a.row_ptr[row + 1];
    // (a software-pipelined form of this loop -- next trip's column and streams requested before the current trip is evaluated, its
    // neighbour quaternion after the block stores -- was measured twice and dropped: round 3, 705-719 us against 710 at C5; round 4, with
    // the loss leaf in scalar registers and three waves per SIMD without spills, 404-455 us against 411-456 on the 100k / 10M ANGLE_AXIS
    // problem (profiles/r04b_roll_ab.txt): three to five waves per SIMD already hide the round trips of this loop)
    for (uint32_t d = a.row_ptr[row] + lane; d < end; d += G) {
      const uint32_t cr = __builtin_nontemporal_load(a.col + d);
      const LinStreams S = lin_load_streams<WM>(a, d);
      const Quat qm = load_q(a.q, cr & 0x7fffffffu);
      double g3[3], G6[6];
      lin_entry_eval<F, WM, LM, true>(a, lv, d, cr, qk, qm, S, g3, G6);
#pragma unroll
      for (int c = 0; c < 3; ++c) acc[c] += g3[c];
#pragma unroll
      for (int c = 0; c < 6; ++c) acc[3 + c] += G6[c];
      nt_store2(a.h0 + d, G6[0], G6[1]);
      nt_store2(a.h1 + d, G6[2], G6[3]);
      nt_store2(a.h2 + d, G6[4], G6[5]);
    }
  }
  for (uint32_t off = G >> 1; off > 0; off >>= 1) {
#pragma unroll
    for (int c = 0; c < 9; ++c) acc[c] += __shfl_down(acc[c], off, G);
  }
  if (live && lane == 0) {
    double* o = a.gD + 9 * (size_t)(a.row_base + row);
#pragma unroll
    for (int c = 0; c < 9; ++c) o[c] = acc[c];
  }
}
template <int F, int WM, int LM>
__global__ void __launch_bounds__(GSFM_BLOCK) GSFM_K2_ATTR k_lin_fast(LinArgs a) { lin_rows_fast<F, WM, LM>(a); }

// Two entry points over the same body: `k_lin3` asks for at least three waves per SIMD, which is free (no spill) for the
// instantiations that matter and would spill for the general loss program and the 9-residual functor; the launcher picks.
template <int F, int WM, int LM, bool LAP>
__global__ void __launch_bounds__(GSFM_BLOCK) k_lin(LinArgs a) { lin_rows<F, WM, LM, LAP>(a); }
template <int F, int WM, int LM, bool LAP>
__global__ void __launch_bounds__(GSFM_BLOCK) GSFM_K2_ATTR k_lin3(LinArgs a) { lin_rows<F, WM, LM, LAP>(a); }

// ------------------------------------------------------------------------------------------
// K3: y_k = M_k p_k + sum_d H_d p[col_d]   (M = diagonal block incl. LM damping, sym 6)
// ------------------------------------------------------------------------------------------
struct MatvecArgs {
  uint32_t n_rows, row_base, G;
  const uint32_t* row_ptr;
  const uint32_t* col;
  const double2 *h0, *h1, *h2, *h3;
  const double* h4;
  const double* Mblk;   // 6 per camera
  const double* p;      // 3 per camera
  double* y;            // 3 per camera
  const int* done;      // PCG convergence flag (may be null)
  const double2* q;     // LAP: camera quaternions
  const double* u;      // LAP: u_k = R_k^T p_k, 3 per camera
};
// LAP = false: y_k = M_k p_k + sum_d H_d p[col_d], 76 B per entry.  LAP = true: y_k = M_k p_k - sum_d G_d (R_k u[col_d]), 52 B per entry.
template <bool LAP>
__global__ void __launch_bounds__(GSFM_BLOCK) k_matvec(MatvecArgs a) {
  if (a.done && *a.done) return;
  const uint32_t G = a.G;
  const uint32_t t = blockIdx.x * GSFM_BLOCK + threadIdx.x;
  const uint32_t row = t / G, lane = t % G;
  const bool live = row < a.n_rows;
  double y0 = 0.0, y1 = 0.0, y2 = 0.0;
  if (live) {
    double Rk[9];
    if (LAP) qmat(load_q(a.q, a.row_base + row), Rk);
    const uint32_t end = a.row_ptr[row + 1];
    for (uint32_t d = a.row_ptr[row] + lane; d < end; d += G) {
      const uint32_t m = __builtin_nontemporal_load(a.col + d) & 0x7fffffffu;
      // the blocks are streamed once per mat-vec: non-temporal loads keep the gathered vector resident in L2
      if (LAP) {
        const double2 A = nt_load2(a.h0 + d), B = nt_load2(a.h1 + d), C = nt_load2(a.h2 + d);   // (g00 g01) (g02 g11) (g12 g22)
        const double* um = a.u + 3 * (size_t)m;
        const double u0 = GSFM_GATHER_LOAD(um), u1 = GSFM_GATHER_LOAD(um + 1), u2 = GSFM_GATHER_LOAD(um + 2);
        const double w0 = Rk[0] * u0 + Rk[1] * u1 + Rk[2] * u2, w1 = Rk[3] * u0 + Rk[4] * u1 + Rk[5] * u2, w2 = Rk[6] * u0 + Rk[7] * u1 + Rk[8] * u2;
        y0 += A.x * w0 + A.y * w1 + B.x * w2;
        y1 += A.y * w0 + B.y * w1 + C.x * w2;
        y2 += B.x * w0 + C.x * w1 + C.y * w2;
      } else {
        const double2 A = nt_load2(a.h0 + d), B = nt_load2(a.h1 + d), C = nt_load2(a.h2 + d), D = nt_load2(a.h3 + d);
        const double E = __builtin_nontemporal_load(a.h4 + d);
        const double* pm = a.p + 3 * (size_t)m;
        const double p0 = pm[0], p1 = pm[1], p2 = pm[2];
        y0 += A.x * p0 + A.y * p1 + B.x * p2;
        y1 += B.y * p0 + C.x * p1 + C.y * p2;
        y2 += D.x * p0 + D.y * p1 + E * p2;
      }
    }
  }
  for (uint32_t off = G >> 1; off > 0; off >>= 1) {
    y0 += __shfl_down(y0, off, G); y1 += __shfl_down(y1, off, G); y2 += __shfl_down(y2, off, G);
  }
  if (live && lane == 0) {
    const size_t k = a.row_base + row;
    const double* M = a.Mblk + 6 * k;
    const double* pk = a.p + 3 * k;
    double mp[3];
    sym3_mulvec(M, pk, mp);
    const double sgn = LAP ? -1.0 : 1.0;
    a.y[3 * k] = mp[0] + sgn * y0; a.y[3 * k + 1] = mp[1] + sgn * y1; a.y[3 * k + 2] = mp[2] + sgn * y2;
  }
}
// u_k = R_k^T p_k (the PCG vector kernels produce it together with p; this is for the other callers of the mat-vec)
__device__ __forceinline__ void rot_transpose_apply(const Quat& q, const double* p, double* u) {
  double R[9];
  qmat(q, R);
  u[0] = R[0] * p[0] + R[3] * p[1] + R[6] * p[2];
  u[1] = R[1] * p[0] + R[4] * p[1] + R[7] * p[2];
  u[2] = R[2] * p[0] + R[5] * p[1] + R[8] * p[2];
}
__global__ void __launch_bounds__(GSFM_BLOCK) k_cam_rotT(const double* __restrict__ p, const double2* __restrict__ q, uint32_t n, double* __restrict__ u) {
  const uint32_t k = blockIdx.x * GSFM_BLOCK + threadIdx.x;
  if (k >= n) return;
  const Quat qq{q[2 * (size_t)k].x, q[2 * (size_t)k].y, q[2 * (size_t)k + 1].x, q[2 * (size_t)k + 1].y};
  double v[3];
  rot_transpose_apply(qq, p + 3 * (size_t)k, v);
  u[3 * (size_t)k] = v[0]; u[3 * (size_t)k + 1] = v[1]; u[3 * (size_t)k + 2] = v[2];
}

// ------------------------------------------------------------------------------------------
// K5: camera kernels
// ------------------------------------------------------------------------------------------
// state -> quaternion cache.  param_dim 3: x = angle-axis; 4: x is already (x,y,z,w).
__global__ void __launch_bounds__(GSFM_BLOCK) k_cam_cache(const double* __restrict__ x, uint32_t n, int param_dim,
                                                          double2* __restrict__ q) {
  const uint32_t k = blockIdx.x * GSFM_BLOCK + threadIdx.x;
  if (k >= n) return;
  Quat qq;
  if (param_dim == 3) qq = aa_to_quat(x[3 * (size_t)k], x[3 * (size_t)k + 1], x[3 * (size_t)k + 2]);
  else qq = Quat{x[4 * (size_t)k], x[4 * (size_t)k + 1], x[4 * (size_t)k + 2], x[4 * (size_t)k + 3]};
  q[2 * (size_t)k] = make_double2(qq.x, qq.y);
  q[2 * (size_t)k + 1] = make_double2(qq.z, qq.w);
}

// quaternion state -> angle-axis (ceres::QuaternionToAngleAxis), estimator.cpp:185-194
__global__ void __launch_bounds__(GSFM_BLOCK) k_quat_to_aa(const double* __restrict__ x, const double* __restrict__ active,
                                                           uint32_t n, double* __restrict__ aa) {
  const uint32_t k = blockIdx.x * GSFM_BLOCK + threadIdx.x;
  if (k >= n) return;
  if (active[k] == 0.0) return;  // views never touched by an edge keep their input value
  const Quat q{x[4 * (size_t)k], x[4 * (size_t)k + 1], x[4 * (size_t)k + 2], x[4 * (size_t)k + 3]};
  double e[3], s, th;
  quat_log(q, e, &s, &th);
  aa[3 * (size_t)k] = e[0]; aa[3 * (size_t)k + 1] = e[1]; aa[3 * (size_t)k + 2] = e[2];
}

__device__ __forceinline__ void cam_tangent_maps(const double* __restrict__ x, size_t k, int param_dim, double* T, double* Tinv) {
  if (param_dim == 3) {
    const double w[3] = {x[3 * k], x[3 * k + 1], x[3 * k + 2]};
    jl_and_inverse(w, T, Tinv);
  } else {
#pragma unroll
    for (int c = 0; c < 9; ++c) { T[c] = 0.0; Tinv[c] = 0.0; }
    T[0] = T[4] = T[8] = 2.0; Tinv[0] = Tinv[4] = Tinv[8] = 0.5;
  }
}

struct PrepArgs {
  uint32_t n;
  int param_dim;
  const double* x;
  const double* gD;       // 9 per camera (eta space)
  double* scale;          // 3 per camera: Jacobi column scaling, fixed at iteration 0
  int init_scale;         // 1 at iteration 0
  int jacobi_scaling;
  double radius, min_diag, max_diag;
  const double* radius_dev;  // non-null: the trust-region radius is read from here (device-side LM control) instead of `radius`
  double* Mblk;           // 6: D + Lambda
  double* Minv;           // 6
  double* Lam;            // 6: damping block in eta space
  double* Tinv;           // 9
  double* b;              // 3: -g_eta
  double* gmax_partials;  // [gridDim.x]
};
// LM diagonal (LevenbergMarquardtStrategy::ComputeStep), block-Jacobi preconditioner and the
// gradient max-norm ||x - Plus(x, -g)||_inf, all per camera.
__global__ void __launch_bounds__(GSFM_BLOCK) k_cam_prep(PrepArgs a) {
  __shared__ double lds[8];
  double gm = 0.0;
  const uint32_t k = blockIdx.x * GSFM_BLOCK + threadIdx.x;
  if (k < a.n) {
    double T[9], Ti[9];
    cam_tangent_maps(a.x, k, a.param_dim, T, Ti);
    const double* gd = a.gD + 9 * (size_t)k;
    const double g[3] = {gd[0], gd[1], gd[2]};
    const double D[6] = {gd[3], gd[4], gd[5], gd[6], gd[7], gd[8]};
    // squared column norms in the reference's parameter space: diag(T^T D T)
    double dd[3], gdl[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const double t0 = T[c], t1 = T[3 + c], t2 = T[6 + c];
      double v[3];
      const double tv[3] = {t0, t1, t2};
      sym3_mulvec(D, tv, v);
      dd[c] = t0 * v[0] + t1 * v[1] + t2 * v[2];
      gdl[c] = t0 * g[0] + t1 * g[1] + t2 * g[2];   // (T^T g)_c
    }
    double sc[3];
    if (a.init_scale) {
#pragma unroll
      for (int c = 0; c < 3; ++c) { sc[c] = a.jacobi_scaling ? 1.0 / (1.0 + sqrt(dd[c])) : 1.0; a.scale[3 * (size_t)k + c] = sc[c]; }
    } else {
#pragma unroll
      for (int c = 0; c < 3; ++c) sc[c] = a.scale[3 * (size_t)k + c];
    }
    double lam[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const double s2 = sc[c] * sc[c];
      lam[c] = fmin(fmax(s2 * dd[c], a.min_diag), a.max_diag) / ((a.radius_dev ? *a.radius_dev : a.radius) * s2);
    }
    // Lambda_eta = Tinv^T diag(lam) Tinv
    double L[6];
    L[0] = lam[0] * Ti[0] * Ti[0] + lam[1] * Ti[3] * Ti[3] + lam[2] * Ti[6] * Ti[6];
    L[1] = lam[0] * Ti[0] * Ti[1] + lam[1] * Ti[3] * Ti[4] + lam[2] * Ti[6] * Ti[7];
    L[2] = lam[0] * Ti[0] * Ti[2] + lam[1] * Ti[3] * Ti[5] + lam[2] * Ti[6] * Ti[8];
    L[3] = lam[0] * Ti[1] * Ti[1] + lam[1] * Ti[4] * Ti[4] + lam[2] * Ti[7] * Ti[7];
    L[4] = lam[0] * Ti[1] * Ti[2] + lam[1] * Ti[4] * Ti[5] + lam[2] * Ti[7] * Ti[8];
    L[5] = lam[0] * Ti[2] * Ti[2] + lam[1] * Ti[5] * Ti[5] + lam[2] * Ti[8] * Ti[8];
    double M[6], Mi[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) { M[c] = D[c] + L[c]; a.Lam[6 * (size_t)k + c] = L[c]; a.Mblk[6 * (size_t)k + c] = M[c]; }
    sym3_inverse(M, Mi);
#pragma unroll
    for (int c = 0; c < 6; ++c) a.Minv[6 * (size_t)k + c] = Mi[c];
#pragma unroll
    for (int c = 0; c < 9; ++c) a.Tinv[9 * (size_t)k + c] = Ti[c];
#pragma unroll
    for (int c = 0; c < 3; ++c) a.b[3 * (size_t)k + c] = -g[c];
    if (a.param_dim == 3) gm = fmax(fabs(gdl[0]), fmax(fabs(gdl[1]), fabs(gdl[2])));
    else {
      // || x - Plus(x, -g) ||_inf with the quaternion Plus
      const double d0 = -gdl[0], d1 = -gdl[1], d2 = -gdl[2];
      const double nd = sqrt(d0 * d0 + d1 * d1 + d2 * d2);
      const Quat q{a.x[4 * (size_t)k], a.x[4 * (size_t)k + 1], a.x[4 * (size_t)k + 2], a.x[4 * (size_t)k + 3]};
      if (nd > 0.0) {
        double sn, cs;
        sincos(nd, &sn, &cs);
        const double kk = sn / nd;
        const Quat r = qmul(Quat{kk * d0, kk * d1, kk * d2, cs}, q);
        gm = fmax(fmax(fabs(q.x - r.x), fabs(q.y - r.y)), fmax(fabs(q.z - r.z), fabs(q.w - r.w)));
      }
    }
  }
  const double t = block_max_bcast(gm, lds);
  if (threadIdx.x == 0) a.gmax_partials[blockIdx.x] = t;
}
__global__ void __launch_bounds__(GSFM_BLOCK) k_max_partials(const double* __restrict__ partials, int n, double* out) {
  __shared__ double lds[8];
  double v = 0.0;
  for (int k = threadIdx.x; k < n; k += GSFM_BLOCK) v = fmax(v, partials[k]);
  const double t = block_max_bcast(v, lds);
  if (threadIdx.x == 0) out[0] = t;
}

// Absolute floor of the PCG tolerance (round 5).  A relative residual of 1e-12 stands in for the reference's exact Cholesky solve; on a step of
// 0.1 rad that is an error of 1e-13 rad, and that -- not twelve digits of a step that is itself 1e-9 rad long -- is what the answer can feel.  A
// solve therefore also stops once block-Jacobi's estimate of what ANY camera's step still lacks is below `floor` radians:
//   |delta_k|^2 = |Tinv_k Minv_k r_k|^2 <= |Tinv_k|^2 |Minv_k| (r_k . Minv_k r_k) <= B (r . Minv r),   B = max_k |Tinv_k|_F^2 |Minv_k|_F,
// i.e. once r.z <= floor^2 / B.  B is taken here, once per LM step (cameras without an edge excluded: their residual is zero); the init
// kernels of the solves turn it into a floor under the relative tolerance.  Decisive for disconnected problems (C4: thirteen scenes have
// converged to 1e-12 rad steps while the fourteenth iterates on -- each of their solves used to run 40 iterations on a right-hand side of nothing).
__global__ void __launch_bounds__(GSFM_BLOCK) k_cam_bound(const double* __restrict__ Minv, const double* __restrict__ Tinv, const double* __restrict__ active, uint32_t n, double* partials) {
  __shared__ double lds[8];
  const uint32_t k = blockIdx.x * GSFM_BLOCK + threadIdx.x;
  double v = 0.0;
  if (k < n && active[k] != 0.0) {
    const double* M = Minv + 6 * (size_t)k;
    const double* T = Tinv + 9 * (size_t)k;
    const double m2 = M[0] * M[0] + M[3] * M[3] + M[5] * M[5] + 2.0 * (M[1] * M[1] + M[2] * M[2] + M[4] * M[4]);
    double t2 = 0.0;
#pragma unroll
    for (int c = 0; c < 9; ++c) t2 += T[c] * T[c];
    v = t2 * sqrt(m2);
  }
  const double t = block_max_bcast(v, lds);
  if (threadIdx.x == 0) partials[blockIdx.x] = t;
}
// relative tolerance of a solve whose initial r.z is rz0: the requested one, or the one the absolute floor allows
__device__ __forceinline__ double cg_tol_with_floor(double tol, double rz_abs, double rz0) { return rz0 > 0.0 ? fmax(tol, sqrt(rz_abs / rz0)) : tol; }

// Forcing schedule: a LOOSE PCG iterate carries a component along the gauge direction eta_k = R_k v (all cameras rotated by the same v in their body
// frames: the exact null space of J^T J, held only by the LM damping, hence the last thing PCG resolves and invisible to its energy norm).  The
// exact step has none: v^T sum_k R_k^T Lam_k eta_k = 0 for every v, because the gradient is orthogonal to the gauge.  These two kernels remove it
// from an inexact step the same way -- w = (sum R^T Lam R)^-1 sum R^T Lam eta, eta_k -= R_k w -- and keep the PCG residual consistent
// (r += Lam R_k w; J^T J R w = 0), so that the model decrease computed from (eta, r) stays exact for the corrected step.
// Out of place (eta_out, rcg_out): the PCG state itself must stay what the stopping iteration left, so that the solve can be continued.
struct GaugeArgs { uint32_t n; int nb; const double* active; const double2* q; const double* Lam; const double* eta; const double* rcg; double* part; /* [9][nb] */ };
__global__ void __launch_bounds__(GSFM_BLOCK) k_gauge_part(GaugeArgs a) {
  __shared__ double lds[8];
  double v[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  const uint32_t k = blockIdx.x * GSFM_BLOCK + threadIdx.x;
  if (k < a.n && a.active[k] != 0.0) {
    double R[9], le[3];
    qmat(load_q(a.q, k), R);
    const double* L = a.Lam + 6 * (size_t)k;
    sym3_mulvec(L, a.eta + 3 * (size_t)k, le);
#pragma unroll
    for (int c = 0; c < 3; ++c) v[c] = R[c] * le[0] + R[3 + c] * le[1] + R[6 + c] * le[2];   // R^T Lam eta
    // R^T Lam R (symmetric: 00 01 02 11 12 22)
    double LR[9];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      LR[c] = L[0] * R[c] + L[1] * R[3 + c] + L[2] * R[6 + c];
      LR[3 + c] = L[1] * R[c] + L[3] * R[3 + c] + L[4] * R[6 + c];
      LR[6 + c] = L[2] * R[c] + L[4] * R[3 + c] + L[5] * R[6 + c];
    }
    v[3] = R[0] * LR[0] + R[3] * LR[3] + R[6] * LR[6]; v[4] = R[0] * LR[1] + R[3] * LR[4] + R[6] * LR[7]; v[5] = R[0] * LR[2] + R[3] * LR[5] + R[6] * LR[8];
    v[6] = R[1] * LR[1] + R[4] * LR[4] + R[7] * LR[7]; v[7] = R[1] * LR[2] + R[4] * LR[5] + R[7] * LR[8]; v[8] = R[2] * LR[2] + R[5] * LR[5] + R[8] * LR[8];
  }
#pragma unroll
  for (int c = 0; c < 9; ++c) {
    const double t = block_sum_bcast(v[c], lds);
    if (threadIdx.x == 0) a.part[(size_t)c * a.nb + blockIdx.x] = t;
  }
}
// w = A^-1 s from the nine sums of k_gauge_part (every block: same partials, same order, same bits); a singular A (no damping at all) leaves
// the step alone.  All lanes of the workgroup must call it (block reductions).
__device__ __forceinline__ void gauge_solve(const double* __restrict__ part, int nb, double* lds, double* w) {
  double S[9];
#pragma unroll
  for (int c = 0; c < 9; ++c) S[c] = sum_partials_bcast(part + (size_t)c * nb, nb, lds);
  const double a00 = S[3], a01 = S[4], a02 = S[5], a11 = S[6], a12 = S[7], a22 = S[8];
  const double c00 = a11 * a22 - a12 * a12, c01 = a02 * a12 - a01 * a22, c02 = a01 * a12 - a02 * a11;
  const double det = a00 * c00 + a01 * c01 + a02 * c02;
  const bool ok = fabs(det) > 0.0;
  const double c11 = a00 * a22 - a02 * a02, c12 = a01 * a02 - a00 * a12, c22 = a00 * a11 - a01 * a01, id = ok ? 1.0 / det : 0.0;
  w[0] = id * (c00 * S[0] + c01 * S[1] + c02 * S[2]); w[1] = id * (c01 * S[0] + c11 * S[1] + c12 * S[2]); w[2] = id * (c02 * S[0] + c12 * S[1] + c22 * S[2]);
}
// the correction of camera k: eta_k - R_k w and rcg_k + Lam_k R_k w (inactive cameras: unchanged)
__device__ __forceinline__ void gauge_correct(const double* w, bool active, const double2* __restrict__ q, const double* __restrict__ Lam, uint32_t k, const double* eta_in, const double* rcg_in, double* e, double* rc) {
  double d[3] = {0.0, 0.0, 0.0}, ld[3] = {0.0, 0.0, 0.0};
  if (active) {
    double R[9];
    qmat(load_q(q, k), R);
    d[0] = R[0] * w[0] + R[1] * w[1] + R[2] * w[2]; d[1] = R[3] * w[0] + R[4] * w[1] + R[5] * w[2]; d[2] = R[6] * w[0] + R[7] * w[1] + R[8] * w[2];
    sym3_mulvec(Lam + 6 * (size_t)k, d, ld);
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) { e[c] = eta_in[3 * (size_t)k + c] - d[c]; rc[c] = rcg_in[3 * (size_t)k + c] + ld[c]; }
}

struct StepArgs {
  uint32_t n;
  int param_dim;
  const double* x;        // current state
  const double* active;   // 1.0 for cameras touched by an edge
  const double* eta;      // PCG solution (left-tangent step)
  const double* b;        // -g_eta
  const double* rcg;      // PCG residual b - A eta
  const double* Lam;      // 6
  const double* Tinv;     // 9
  double* x_trial;
  double2* q_trial;
  double* partials;       // 6 * gridDim.x : eta.g, eta.rcg, eta^T Lam eta, |x - x_trial|^2, |x_trial|^2, sum_k |Tinv Minv rcg|_k^8 (0 unless Minv is given)
  const double* Minv;     // non-null (a loose PCG iterate, round 5): the sixth sum -- what block-Jacobi says each camera's step still lacks, in the
                          // units of the update (radians; half-angles for the quaternion state).  The energy norm the loose solve stops on weights a
                          // camera by its own weight sum: a camera whose edges are nearly all cut off by a redescending loss is invisible to it
                          // and can be left 1e-4 rad from its exact step under an energy error of 1e-8 (tests/manual/fuzz_forcing.py dense 5:14,
                          // 6:52: Tukey on ROTATION_MAT_FNORM).  The eighth-power sum is a smooth maximum: its 8th root lies between the largest
                          // camera's value and N^(1/8) times it (4.2 x at 100k cameras); lm_solve holds it against 10 x the rms tolerance.
  // a loose PCG iterate (forcing schedule): the nine sums of k_gauge_part; the gauge component is taken out of eta and the residual corrected
  // on the fly (round 4: the kernel that wrote the corrected copies is gone -- one launch fewer per loose step; the PCG state stays untouched)
  const double* gauge_part; int gauge_nb; const double2* gauge_q;
};
// delta = Tinv eta; x_trial = Plus(x, delta); scalars for the model cost change and the
// parameter-tolerance test (TrustRegionMinimizer::ComputeCandidatePointAndEvaluateCost).
__global__ void __launch_bounds__(GSFM_BLOCK) k_cam_step(StepArgs a) {
  __shared__ double lds[8];
  double v[6] = {0, 0, 0, 0, 0, 0};
  const uint32_t k = blockIdx.x * GSFM_BLOCK + threadIdx.x;
  double gw[3] = {0.0, 0.0, 0.0};
  if (a.gauge_part) gauge_solve(a.gauge_part, a.gauge_nb, lds, gw);
  if (k < a.n) {
    const size_t k3 = 3 * (size_t)k;
    double e[3] = {a.eta[k3], a.eta[k3 + 1], a.eta[k3 + 2]}, rc[3] = {a.rcg[k3], a.rcg[k3 + 1], a.rcg[k3 + 2]};
    if (a.gauge_part) gauge_correct(gw, a.active[k] != 0.0, a.gauge_q, a.Lam, k, a.eta, a.rcg, e, rc);
    const double* Ti = a.Tinv + 9 * (size_t)k;
    const double d[3] = {Ti[0] * e[0] + Ti[1] * e[1] + Ti[2] * e[2], Ti[3] * e[0] + Ti[4] * e[1] + Ti[5] * e[2],
                         Ti[6] * e[0] + Ti[7] * e[1] + Ti[8] * e[2]};
    double le[3];
    sym3_mulvec(a.Lam + 6 * (size_t)k, e, le);
    v[0] = -(e[0] * a.b[k3] + e[1] * a.b[k3 + 1] + e[2] * a.b[k3 + 2]);
    v[1] = e[0] * rc[0] + e[1] * rc[1] + e[2] * rc[2];
    v[2] = e[0] * le[0] + e[1] * le[1] + e[2] * le[2];
    const double act = a.active[k];
    if (a.Minv) {
      double z[3];
      sym3_mulvec(a.Minv + 6 * (size_t)k, rc, z);
      const double dz0 = Ti[0] * z[0] + Ti[1] * z[1] + Ti[2] * z[2], dz1 = Ti[3] * z[0] + Ti[4] * z[1] + Ti[5] * z[2], dz2 = Ti[6] * z[0] + Ti[7] * z[1] + Ti[8] * z[2];
      const double m2 = act * (dz0 * dz0 + dz1 * dz1 + dz2 * dz2), m4 = m2 * m2;
      v[5] = m4 * m4;
    }
    Quat qt;
    if (a.param_dim == 3) {
      const double x0 = a.x[k3] + d[0], x1 = a.x[k3 + 1] + d[1], x2 = a.x[k3 + 2] + d[2];
      a.x_trial[k3] = x0; a.x_trial[k3 + 1] = x1; a.x_trial[k3 + 2] = x2;
      v[3] = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
      v[4] = act * (x0 * x0 + x1 * x1 + x2 * x2);
      qt = aa_to_quat(x0, x1, x2);
    } else {
      const size_t k4 = 4 * (size_t)k;
      const Quat q{a.x[k4], a.x[k4 + 1], a.x[k4 + 2], a.x[k4 + 3]};
      const double nd = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
      qt = q;
      if (nd > 0.0) {
        double sn, cs;
        sincos(nd, &sn, &cs);
        const double kk = sn / nd;
        qt = qmul(Quat{kk * d[0], kk * d[1], kk * d[2], cs}, q);
      }
      a.x_trial[k4] = qt.x; a.x_trial[k4 + 1] = qt.y; a.x_trial[k4 + 2] = qt.z; a.x_trial[k4 + 3] = qt.w;
      const double f0 = q.x - qt.x, f1 = q.y - qt.y, f2 = q.z - qt.z, f3 = q.w - qt.w;
      v[3] = f0 * f0 + f1 * f1 + f2 * f2 + f3 * f3;
      v[4] = act * (qt.x * qt.x + qt.y * qt.y + qt.z * qt.z + qt.w * qt.w);
    }
    a.q_trial[2 * (size_t)k] = make_double2(qt.x, qt.y);
    a.q_trial[2 * (size_t)k + 1] = make_double2(qt.z, qt.w);
  }
#pragma unroll
  for (int c = 0; c < 6; ++c) {
    const double t = block_sum_bcast(v[c], lds);
    if (threadIdx.x == 0) a.partials[(size_t)c * gridDim.x + blockIdx.x] = t;
  }
}
// out[c] = sum partials[c*n .. c*n+n)
__global__ void __launch_bounds__(GSFM_BLOCK) k_sum_partials_multi(const double* __restrict__ partials, int n, int m, double* out) {
  __shared__ double lds[8];
  for (int c = 0; c < m; ++c) {
    const double t = sum_partials_bcast(partials + (size_t)c * n, n, lds);
    if (threadIdx.x == 0) out[c] = t;
  }
}
// |x|^2 over active cameras (Init: x_norm_)
__global__ void __launch_bounds__(GSFM_BLOCK) k_cam_norm(const double* __restrict__ x, const double* __restrict__ active,
                                                         uint32_t n, int param_dim, double* partials) {
  __shared__ double lds[8];
  double v = 0.0;
  const uint32_t k = blockIdx.x * GSFM_BLOCK + threadIdx.x;
  if (k < n && active[k] != 0.0) {
    for (int c = 0; c < param_dim; ++c) { const double t = x[(size_t)param_dim * k + c]; v += t * t; }
  }
  const double t = block_sum_bcast(v, lds);
  if (threadIdx.x == 0) partials[blockIdx.x] = t;
}

// export gradient / diagonal blocks in the reference's parameter space: T^T g, T^T D T
__global__ void __launch_bounds__(GSFM_BLOCK) k_cam_export(const double* __restrict__ x, const double* __restrict__ gD,
                                                           uint32_t n, int param_dim, double* grad, double* blocks, double* D6) {
  const uint32_t k = blockIdx.x * GSFM_BLOCK + threadIdx.x;
  if (k >= n) return;
  double T[9], Ti[9];
  cam_tangent_maps(x, k, param_dim, T, Ti);
  const double* gd = gD + 9 * (size_t)k;
  const double D[9] = {gd[3], gd[4], gd[5], gd[4], gd[6], gd[7], gd[5], gd[7], gd[8]};
  double DT[9], TDT[9];
  mat3_mul(D, T, DT);
  mat3_tmul(T, DT, TDT);
  for (int c = 0; c < 3; ++c) grad[3 * (size_t)k + c] = T[c] * gd[0] + T[3 + c] * gd[1] + T[6 + c] * gd[2];
  for (int c = 0; c < 9; ++c) blocks[9 * (size_t)k + c] = TDT[c];
  for (int c = 0; c < 6; ++c) D6[6 * (size_t)k + c] = gd[3 + c];
}
// v_eta = T v (mode 0)  or  y = T^T y_eta (mode 1), for gsfm_rot_normal_matvec
__global__ void __launch_bounds__(GSFM_BLOCK) k_cam_apply_T(const double* __restrict__ x, uint32_t n, int param_dim, int transpose,
                                                            const double* __restrict__ in, double* __restrict__ out) {
  const uint32_t k = blockIdx.x * GSFM_BLOCK + threadIdx.x;
  if (k >= n) return;
  double T[9], Ti[9];
  cam_tangent_maps(x, k, param_dim, T, Ti);
  const double v0 = in[3 * (size_t)k], v1 = in[3 * (size_t)k + 1], v2 = in[3 * (size_t)k + 2];
  for (int c = 0; c < 3; ++c)
    out[3 * (size_t)k + c] = transpose ? (T[c] * v0 + T[3 + c] * v1 + T[6 + c] * v2) : (T[3 * c] * v0 + T[3 * c + 1] * v1 + T[3 * c + 2] * v2);
}

// ------------------------------------------------------------------------------------------
// K4: PCG vector kernels.  Device scalars: rz[2] (parity-indexed), rz0, done, iters.
// ------------------------------------------------------------------------------------------
struct CgScalars {
  double rz[2];
  double rz0;
  double last_rel;  // sqrt(rz/rz0) at the last iteration
  double best_rel;  // smallest relative residual seen so far (stagnation detection)
  int done;
  int iters;
  int stall;        // iterations since best_rel last halved
  int stalled;      // 1 if the solve ended by stagnation
  int done_seen;    // `done` as k_cg_update found it: what k_cg_pupdate -- the kernel that SETS done -- tests at its entry, so that the launch that
                    // detects convergence still updates p on every block and leaves a state the solve can be resumed from (k_cg_resume)
  int pad_;
  double tol;       // relative tolerance of the current run: device-resident, so that a captured chunk does not freeze it (forcing schedule,
                    // solver_lm.hpp: a loose solve may be continued to the tight tolerance, bit for bit as if it had never stopped)
  double etol2;     // loose solves: stop once the ESTIMATED relative energy-norm error of the iterate, squared, is below this (0 = off); see cg_energy_stop
  double esum;      // sum of the iterations' decreases of the quadratic model, alpha_j (r_j . z_j) = |x_{j+1}|_A^2 - |x_j|_A^2 growth (Hestenes-Stiefel)
  double einc[4];   // the last four of them (ring, indexed by iteration & 3)
  double rz_abs;    // absolute floor on r.z (k_cam_bound): `tol` never drops below sqrt(rz_abs / rz0)
};
// Energy-norm stopping rule of the forcing schedule.  PCG from x_0 = 0 gains inc_j = alpha_j (r_j . z_j) of |x|_A^2 per iteration, and the squared
// energy error after k iterations is the sum of all LATER gains (Hestenes & Stiefel 1952; Strakos & Tichy 2002).  The later gains are extrapolated
// geometrically from the last four: q = (inc_{k-1} + inc_{k-2}) / (inc_{k-3} + inc_{k-4}) is the decay per two iterations, the remainder
// (inc_{k-1} + inc_{k-2}) q / (1 - q).  Unlike a residual norm this bounds what an LM step is about -- the share of the model decrease still
// missing -- whatever the conditioning and the preconditioner.  `k` = iterations done (>= 4), ring = einc.
__device__ __forceinline__ bool cg_energy_stop(const double* ring, double esum, double etol2, int k) {
  if (!(etol2 > 0.0) || k < 4) return false;
  const double a = ring[(k - 1) & 3] + ring[(k - 2) & 3], b = ring[(k - 3) & 3] + ring[(k - 4) & 3];
  if (!(a < b) || !(a >= 0.0)) return false;   // no decay (or a breakdown): carry on
  const double q = a / b;
  // (Round 5 tried a conditioning correction here -- the estimate held against etol2 / kappa, kappa from the current rate q^(1/4) -- for the
  // ill-conditioned far-start problems the round-4 fuzz lost.  Those are caught by the contraction gate of lm_solve instead; with it in place
  // the correction changed no outcome in 420 fuzz trials and cost the spanning-tree start 35 % more iterations: removed.)
  return a * q <= etol2 * esum * (1.0 - q);
}
struct CgArgs {
  uint32_t n;        // cameras
  int nb;            // blocks of the camera kernels (= number of partials)
  int par;           // iteration parity
  double tol, etol2; // (read by the init kernels only: the run's tolerances live in CgScalars)
  int max_iters;
  int stall_limit;   // 0 = off
  const double* Minv;
  const double* b;
  double *xcg, *r, *z, *p, *Ap;
  double* part_a;    // [nb]
  double* part_b;    // [nb]
  const double* zbound; double abs_floor2;   // k_cam_bound's B (device scalar; null or 0: no floor) and floor^2
  CgScalars* sc;
  const double2* q;  // Laplacian form: camera quaternions and
  double* u;         //   u_k = R_k^T p_k, written wherever p is (null otherwise)
  // two-level preconditioner (see k_coarse_*): z = Minv r + P xc, with xc the coarse correction of this iteration; 0 aggregates = off
  uint32_t coarse_n, coarse_chunk;
  const double* xc;  // [3 coarse_n + 1]: correction per aggregate (body frame), then rc . xc
  const double* active;  // 1 / 0 per camera: cameras without an edge take no part in the coarse space either (they must not move)
  double* rc_part;       // [nb][2][3]: this kernel block's share of P^T r for the (at most two) aggregates its 256 cameras belong to; null = the
                         // restriction runs as its own kernel (aggregates narrower than a block)
};
// (P xc)_k = R_k xc[aggregate of k]
__device__ __forceinline__ void coarse_prolong(const CgArgs& a, uint32_t k, double* out) {
  if (a.active[k] == 0.0) { out[0] = out[1] = out[2] = 0.0; return; }
  const uint32_t I = min(k / a.coarse_chunk, a.coarse_n - 1);
  const Quat qq{a.q[2 * (size_t)k].x, a.q[2 * (size_t)k].y, a.q[2 * (size_t)k + 1].x, a.q[2 * (size_t)k + 1].y};
  double R[9];
  qmat(qq, R);
  const double x0 = a.xc[3 * I], x1 = a.xc[3 * I + 1], x2 = a.xc[3 * I + 2];
  out[0] = R[0] * x0 + R[1] * x1 + R[2] * x2; out[1] = R[3] * x0 + R[4] * x1 + R[5] * x2; out[2] = R[6] * x0 + R[7] * x1 + R[8] * x2;
}

// x = 0, r = b, z = Minv r, p = z, partial r.z
__global__ void __launch_bounds__(GSFM_BLOCK) k_cg_init(CgArgs a) {
  __shared__ double lds[8];
  double v = 0.0;
  const uint32_t k = blockIdx.x * GSFM_BLOCK + threadIdx.x;
  if (k < a.n) {
    const size_t k3 = 3 * (size_t)k;
    const double r[3] = {a.b[k3], a.b[k3 + 1], a.b[k3 + 2]};
    double z[3];
    sym3_mulvec(a.Minv + 6 * (size_t)k, r, z);
#pragma unroll
    for (int c = 0; c < 3; ++c) { a.xcg[k3 + c] = 0.0; a.r[k3 + c] = r[c]; a.z[k3 + c] = z[c]; a.p[k3 + c] = z[c]; v += r[c] * z[c]; }
    if (a.u) {
      const Quat qq{a.q[2 * (size_t)k].x, a.q[2 * (size_t)k].y, a.q[2 * (size_t)k + 1].x, a.q[2 * (size_t)k + 1].y};
      double uu[3];
      rot_transpose_apply(qq, z, uu);
      a.u[k3] = uu[0]; a.u[k3 + 1] = uu[1]; a.u[k3 + 2] = uu[2];
    }
  }
  const double t = block_sum_bcast(v, lds);
  if (threadIdx.x == 0) a.part_b[blockIdx.x] = t;
}
__global__ void __launch_bounds__(GSFM_BLOCK) k_cg_init_fin(CgArgs a) {
  __shared__ double lds[8];
  const double rz = sum_partials_bcast(a.part_b, a.nb, lds);
  if (threadIdx.x == 0) {
    const double bound = a.zbound ? *a.zbound : 0.0, rz_abs = bound > 0.0 ? a.abs_floor2 / bound : 0.0;
    a.sc->rz_abs = rz_abs;
    a.sc->rz[0] = rz; a.sc->rz[1] = rz; a.sc->rz0 = rz; a.sc->best_rel = 1.0;
    a.sc->done = !(rz > rz_abs); a.sc->last_rel = a.sc->done ? 0.0 : 1.0;   // (nothing to solve: converged, not "stopped above the tolerance")
    a.sc->iters = 0; a.sc->stall = 0; a.sc->stalled = 0; a.sc->done_seen = a.sc->done; a.sc->tol = cg_tol_with_floor(a.tol, rz_abs, rz);
    a.sc->etol2 = a.etol2; a.sc->esum = 0.0; a.sc->einc[0] = a.sc->einc[1] = a.sc->einc[2] = a.sc->einc[3] = 0.0;
  }
}
// Continue a stopped solve to a tighter tolerance: the vectors, rz and the iteration count are exactly what the stopping iteration left
// (see done_seen), so the iterates that follow are those of a solve that ran at `tol` from the start.
__global__ void k_cg_resume(CgScalars* sc, double tol, double etol2, int max_iters) {
  tol = cg_tol_with_floor(tol, sc->rz_abs, sc->rz0);
  sc->tol = tol; sc->etol2 = etol2;
  sc->done = !(sc->rz0 > 0.0) || !(sc->last_rel > tol) || sc->iters >= max_iters || sc->stalled;
  sc->done_seen = sc->done;
}
// partial p.Ap
__global__ void __launch_bounds__(GSFM_BLOCK) k_cg_dot(CgArgs a) {
  if (a.sc->done) return;
  __shared__ double lds[8];
  double v = 0.0;
  const uint32_t k = blockIdx.x * GSFM_BLOCK + threadIdx.x;
  if (k < a.n) {
    const size_t k3 = 3 * (size_t)k;
    v = a.p[k3] * a.Ap[k3] + a.p[k3 + 1] * a.Ap[k3 + 1] + a.p[k3 + 2] * a.Ap[k3 + 2];
  }
  const double t = block_sum_bcast(v, lds);
  if (threadIdx.x == 0) a.part_a[blockIdx.x] = t;
}
// alpha = rz / pAp; x += alpha p; r -= alpha Ap; z = Minv r; partial r.z
__global__ void __launch_bounds__(GSFM_BLOCK) k_cg_update(CgArgs a) {
  const int done = a.sc->done;
  if (blockIdx.x == 0 && threadIdx.x == 0) a.sc->done_seen = done;   // (nobody writes `done` during this launch)
  if (done) return;
  __shared__ double lds[8];
  const double pAp = sum_partials_bcast(a.part_a, a.nb, lds);
  const double alpha = a.sc->rz[a.par] / pAp;
  if (blockIdx.x == 0 && threadIdx.x == 0) { const double inc = alpha * a.sc->rz[a.par]; a.sc->einc[a.sc->iters & 3] = inc; a.sc->esum += inc; }
  double v = 0.0, rc6[6] = {0, 0, 0, 0, 0, 0};
  const uint32_t k = blockIdx.x * GSFM_BLOCK + threadIdx.x;
  if (k < a.n) {
    const size_t k3 = 3 * (size_t)k;
    double r[3], z[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) { a.xcg[k3 + c] += alpha * a.p[k3 + c]; r[c] = a.r[k3 + c] - alpha * a.Ap[k3 + c]; a.r[k3 + c] = r[c]; }
    sym3_mulvec(a.Minv + 6 * (size_t)k, r, z);
#pragma unroll
    for (int c = 0; c < 3; ++c) { a.z[k3 + c] = z[c]; v += r[c] * z[c]; }
    if (a.rc_part && a.active[k] != 0.0) {   // two-level preconditioner: this camera's term of P^T r, for the first or the second aggregate of the block
      const Quat qq{a.q[2 * (size_t)k].x, a.q[2 * (size_t)k].y, a.q[2 * (size_t)k + 1].x, a.q[2 * (size_t)k + 1].y};
      double uu[3];
      rot_transpose_apply(qq, r, uu);
      const uint32_t I = min(k / a.coarse_chunk, a.coarse_n - 1), I0 = min((blockIdx.x * GSFM_BLOCK) / a.coarse_chunk, a.coarse_n - 1);
      const int sel = I != I0;
#pragma unroll
      for (int c = 0; c < 3; ++c) { rc6[c] = sel ? 0.0 : uu[c]; rc6[3 + c] = sel ? uu[c] : 0.0; }
    }
  }
  const double t = block_sum_bcast(v, lds);
  if (threadIdx.x == 0) a.part_b[blockIdx.x] = t;
  if (a.rc_part) {
    __shared__ double l6[GSFM_BLOCK / 64][6];
#pragma unroll
    for (int c = 0; c < 6; ++c) { const double w = wave_sum(rc6[c]); if ((threadIdx.x & 63) == 0) l6[threadIdx.x >> 6][c] = w; }
    __syncthreads();
    if (threadIdx.x < 6) {
      double sum = 0.0;
      for (int w = 0; w < GSFM_BLOCK / 64; ++w) sum += l6[w][threadIdx.x];
      a.rc_part[6 * (size_t)blockIdx.x + threadIdx.x] = sum;
    }
  }
}
// beta = rz_new / rz; p = z + beta p; convergence test
__global__ void __launch_bounds__(GSFM_BLOCK) k_cg_pupdate(CgArgs a) {
  if (a.sc->done_seen) return;   // not `done`: block 0 of THIS launch sets it, and every block must still finish the p update (resumable state)
  __shared__ double lds[8];
  double rz_new = sum_partials_bcast(a.part_b, a.nb, lds);
  if (a.coarse_n) rz_new += a.xc[3 * a.coarse_n];   // r . (Minv r + P xc) = r . Minv r + (P^T r) . xc
  const double rz_old = a.sc->rz[a.par];
  const double beta = rz_new / rz_old;
  const uint32_t k = blockIdx.x * GSFM_BLOCK + threadIdx.x;
  if (k < a.n) {
    const size_t k3 = 3 * (size_t)k;
    double pn[3], zc[3] = {0.0, 0.0, 0.0};
    if (a.coarse_n) coarse_prolong(a, k, zc);
#pragma unroll
    for (int c = 0; c < 3; ++c) { pn[c] = (a.z[k3 + c] + zc[c]) + beta * a.p[k3 + c]; a.p[k3 + c] = pn[c]; }
    if (a.u) {
      const Quat qq{a.q[2 * (size_t)k].x, a.q[2 * (size_t)k].y, a.q[2 * (size_t)k + 1].x, a.q[2 * (size_t)k + 1].y};
      double uu[3];
      rot_transpose_apply(qq, pn, uu);
      a.u[k3] = uu[0]; a.u[k3 + 1] = uu[1]; a.u[k3 + 2] = uu[2];
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    a.sc->rz[a.par ^ 1] = rz_new;
    const int it = a.sc->iters + 1;
    a.sc->iters = it;
    const double rel = sqrt(rz_new / a.sc->rz0);
    a.sc->last_rel = rel;
    // Every block of this launch finishes its p update (the entry test reads done_seen); every later kernel observes the flag at its entry.
    if (!(rel > a.sc->tol) || it >= a.max_iters || cg_energy_stop(a.sc->einc, a.sc->esum, a.sc->etol2, it)) a.sc->done = 1;
    if (a.stall_limit > 0) {  // numerically singular system: the residual plateaus at rounding level
      if (rel < 0.5 * a.sc->best_rel) { a.sc->best_rel = rel; a.sc->stall = 0; }
      else if (++a.sc->stall >= a.stall_limit) { a.sc->done = 1; a.sc->stalled = 1; }
    }
  }
}

// ------------------------------------------------------------------------------------------
// Two-level preconditioner for spatially coherent graphs (block-Jacobi alone needs hundreds of PCG iterations per step there: the
// condition number grows with the square of the graph's diameter).  In the body frame u_k = R_k^T eta_k the normal matrix of the
// Laplacian form is a graph Laplacian with one symmetric 3 x 3 weight per edge, whose near-null vectors are the constants (the global
// gauge rotation), so the coarse space is one 3-vector per aggregate = contiguous chunk of the locality ordering, rotated by R_k:
//   z = Minv r + P Ac^-1 P^T r,   (P v)_k = R_k v[agg(k)],   Ac = P^T A P  (3 n_agg square, assembled per LM step, inverted on the host).
// Additive, symmetric positive definite: PCG's answer does not depend on it, only its iteration count does.
// ------------------------------------------------------------------------------------------
struct CoarseArgs {
  uint32_t n, n_agg, chunk;     // cameras, aggregates, cameras per aggregate
  const double2* q;
  const double* r;              // residual, 3 per camera
  double* rc;                   // [3 n_agg]  P^T r
  const double* Ainv;           // [nc x nc], symmetric
  double* xc;                   // [nc + 1]   Ainv rc, then rc . xc
  const int* done;
  const double* active;
  const double* rc_part;        // non-null: rc is the fixed-order sum of the camera blocks' shares written by k_cg_update ([nb][2][3])
  uint32_t nb;
};
__global__ void __launch_bounds__(GSFM_BLOCK) k_coarse_restrict(CoarseArgs a) {
  if (a.done && *a.done) return;
  __shared__ double lds[8];
  const uint32_t I = blockIdx.x, lo = I * a.chunk, hi = (I + 1 == a.n_agg) ? a.n : min(a.n, lo + a.chunk);
  double acc[3] = {0.0, 0.0, 0.0};
  for (uint32_t k = lo + threadIdx.x; k < hi; k += GSFM_BLOCK) {
    if (a.active[k] == 0.0) continue;
    const Quat qq{a.q[2 * (size_t)k].x, a.q[2 * (size_t)k].y, a.q[2 * (size_t)k + 1].x, a.q[2 * (size_t)k + 1].y};
    double uu[3];
    rot_transpose_apply(qq, a.r + 3 * (size_t)k, uu);
    acc[0] += uu[0]; acc[1] += uu[1]; acc[2] += uu[2];
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const double t = block_sum_bcast(acc[c], lds);
    if (threadIdx.x == 0) a.rc[3 * I + c] = t;
  }
}
// xc = Ainv rc and rc . xc: one workgroup of 1024 lanes; lane (row, g) sums the columns g, g + groups, ... of its row (Ainv is symmetric, so
// column `row` is read as a row: coalesced over the lanes), the groups are combined through LDS.  3 n_agg <= 384.
#define GSFM_COARSE_MAX_NC 384
__global__ void __launch_bounds__(1024) k_coarse_apply(CoarseArgs a) {
  if (a.done && *a.done) return;
  __shared__ double rcs[GSFM_COARSE_MAX_NC];
  __shared__ double part[1024];
  __shared__ double lds[20];
  const uint32_t nc = 3 * a.n_agg, tid = threadIdx.x, groups = 1024 / nc, row = tid % nc, g = tid / nc;
  for (uint32_t c = tid; c < nc; c += 1024) {
    if (a.rc_part) {   // aggregate I = cameras [I chunk, (I + 1) chunk): the blocks of 256 cameras that overlap it, in order
      const uint32_t I = c / 3, comp = c % 3, lo = I * a.chunk, hi = (I + 1 == a.n_agg) ? a.n : min(a.n, lo + a.chunk);
      double sum = 0.0;
      if (hi > lo) {
        for (uint32_t w = lo / GSFM_BLOCK; w <= (hi - 1) / GSFM_BLOCK && w < a.nb; ++w) {
          const uint32_t I0 = min((w * GSFM_BLOCK) / a.chunk, a.n_agg - 1);
          if (I == I0) sum += a.rc_part[6 * (size_t)w + comp];
          else if (I == I0 + 1) sum += a.rc_part[6 * (size_t)w + 3 + comp];
        }
      }
      rcs[c] = sum;
    } else rcs[c] = a.rc[c];
  }
  __syncthreads();
  double sum = 0.0;
  if (g < groups) {
#pragma unroll 8
    for (uint32_t c = g; c < nc; c += groups) sum += a.Ainv[(size_t)c * nc + row] * rcs[c];
  }
  part[tid] = sum;
  __syncthreads();
  double dot = 0.0;
  if (tid < nc) {
    double x = 0.0;
    for (uint32_t gg = 0; gg < groups; ++gg) x += part[gg * nc + tid];
    a.xc[tid] = x;
    dot = x * rcs[tid];
  }
  // block sum over 1024 lanes
  dot = wave_sum(dot);
  if ((tid & 63) == 0) lds[tid >> 6] = dot;
  __syncthreads();
  if (tid == 0) { double t = 0.0; for (int w = 0; w < 16; ++w) t += lds[w]; a.xc[nc] = t; }
}
// after k_cg_init / k_cg_init_fin: p = z + P xc (and u = R^T p), r.z += rc . xc
__global__ void __launch_bounds__(GSFM_BLOCK) k_cg_init_coarse(CgArgs a) {
  const uint32_t k = blockIdx.x * GSFM_BLOCK + threadIdx.x;
  if (k < a.n) {
    const size_t k3 = 3 * (size_t)k;
    double zc[3], pn[3];
    coarse_prolong(a, k, zc);
#pragma unroll
    for (int c = 0; c < 3; ++c) { pn[c] = a.z[k3 + c] + zc[c]; a.p[k3 + c] = pn[c]; }
    if (a.u) {
      const Quat qq{a.q[2 * (size_t)k].x, a.q[2 * (size_t)k].y, a.q[2 * (size_t)k + 1].x, a.q[2 * (size_t)k + 1].y};
      double uu[3];
      rot_transpose_apply(qq, pn, uu);
      a.u[k3] = uu[0]; a.u[k3 + 1] = uu[1]; a.u[k3 + 2] = uu[2];
    }
  }
}
__global__ void k_cg_init_coarse_fin(CgArgs a) {
  const double rz = a.sc->rz[0] + a.xc[3 * a.coarse_n];
  a.sc->rz[0] = rz; a.sc->rz[1] = rz; a.sc->rz0 = rz; a.sc->done = !(rz > a.sc->rz_abs); a.sc->done_seen = a.sc->done; a.sc->last_rel = a.sc->done ? 0.0 : 1.0;
  a.sc->tol = cg_tol_with_floor(a.tol, a.sc->rz_abs, rz);
}
// Ac = P^T A P from the stored blocks of the Laplacian form: off-diagonal entry (k -> m) contributes -R_k^T G_k R_k to block (agg k, agg m),
// the diagonal block R_k^T M_k R_k to (agg k, agg k).  G lanes per row; a lane sums its consecutive entries that fall into the same
// aggregate before touching memory (rows are sorted by neighbour, so that is most of them), then adds with 64-bit integer atomics on a
// fixed-point image of the matrix (coarse_flush): bit-identical from run to run.  Ac is zero-filled before the launch.
struct CoarseAsmArgs {
  uint32_t n_rows, row_base, G, n_agg, chunk;   // owned rows; global camera index of row 0
  const uint32_t* row_ptr;
  const uint32_t* col;
  const double2 *h0, *h1, *h2;
  const double* Mblk;
  const double2* q;
  double* Ac;          // fixed point while being summed (coarse_flush), doubles after k_coarse_unscale
  const double* scale;
};
// Sums in 64-bit FIXED POINT: integer addition is associative, so the atomics may land in any order and Ac is still the same bits on every
// run (floating-point atomics would make the preconditioner, and through it the PCG path, reproducible to rounding only).  `scale` = 2^e with
// e chosen by k_coarse_scale so that the largest single contribution is below 2^40: 2^22 of them fit before an int64 overflows (an aggregate
// sums <= chunk x degree ~ 2^19), and the resolution of 2^-40 of the largest diagonal entry is far finer than a preconditioner needs.
__device__ __forceinline__ void coarse_flush(double* Ac, uint32_t nc, uint32_t I, uint32_t J, const double* S /* sym 6: 00 01 02 11 12 22 */, double scale) {
  unsigned long long* o = (unsigned long long*)(Ac + (size_t)(3 * I) * nc + 3 * J);
  long long q[6];
#pragma unroll
  for (int c = 0; c < 6; ++c) q[c] = __double2ll_rn(S[c] * scale);
  atomicAdd(o, (unsigned long long)q[0]); atomicAdd(o + 1, (unsigned long long)q[1]); atomicAdd(o + 2, (unsigned long long)q[2]);
  atomicAdd(o + nc, (unsigned long long)q[1]); atomicAdd(o + nc + 1, (unsigned long long)q[3]); atomicAdd(o + nc + 2, (unsigned long long)q[4]);
  atomicAdd(o + 2 * nc, (unsigned long long)q[2]); atomicAdd(o + 2 * nc + 1, (unsigned long long)q[4]); atomicAdd(o + 2 * nc + 2, (unsigned long long)q[5]);
}
// scale[0] = 2^e, scale[1] = 2^-e from the largest diagonal entry of the damped diagonal blocks (every |G_ab| of an edge is below it: G is
// positive semi-definite and M_k sums the G of camera k's edges)
__global__ void __launch_bounds__(GSFM_BLOCK) k_coarse_scale(const double* __restrict__ Mblk, uint32_t n_rows, uint32_t row_base, double* partials, double* scale, int pass) {
  __shared__ double lds[8];
  double v = 0.0;
  if (pass == 0) {
    const uint32_t r = blockIdx.x * GSFM_BLOCK + threadIdx.x;
    if (r < n_rows) { const double* M = Mblk + 6 * (size_t)(row_base + r); v = fmax(fabs(M[0]), fmax(fabs(M[3]), fabs(M[5]))); }
    v = block_max_bcast(v, lds);
    if (threadIdx.x == 0) partials[blockIdx.x] = v;
  } else {   // one block: partials -> the power of two
    for (uint32_t k = threadIdx.x; k < n_rows /* = number of partials */; k += GSFM_BLOCK) v = fmax(v, partials[k]);
    v = block_max_bcast(v, lds);
    if (threadIdx.x == 0) {
      int e = 0;
      if (v > 0.0 && isfinite(v)) { (void)frexp(v, &e); e = 40 - e; }   // v < 2^(40 - e')... v * 2^e < 2^40
      scale[0] = ldexp(1.0, e); scale[1] = ldexp(1.0, -e);
    }
  }
}
// fixed point -> double, in place
__global__ void __launch_bounds__(GSFM_BLOCK) k_coarse_unscale(double* Ac, size_t n, const double* scale) {
  const size_t t = (size_t)blockIdx.x * GSFM_BLOCK + threadIdx.x;
  if (t < n) { const long long q = ((const long long*)Ac)[t]; Ac[t] = (double)q * scale[1]; }
}
// S = R^T Sym R for a symmetric 3 x 3 given as (00 01 02 11 12 22)
__device__ __forceinline__ void sym3_congruence_T(const double* R, const double* M, double* S) {
  const double m[9] = {M[0], M[1], M[2], M[1], M[3], M[4], M[2], M[4], M[5]};
  double t[9];   // t = M R
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) t[3 * r + c] = m[3 * r] * R[c] + m[3 * r + 1] * R[3 + c] + m[3 * r + 2] * R[6 + c];
  // S = R^T t
  S[0] = R[0] * t[0] + R[3] * t[3] + R[6] * t[6]; S[1] = R[0] * t[1] + R[3] * t[4] + R[6] * t[7]; S[2] = R[0] * t[2] + R[3] * t[5] + R[6] * t[8];
  S[3] = R[1] * t[1] + R[4] * t[4] + R[7] * t[7]; S[4] = R[1] * t[2] + R[4] * t[5] + R[7] * t[8]; S[5] = R[2] * t[2] + R[5] * t[5] + R[8] * t[8];
}
__global__ void __launch_bounds__(GSFM_BLOCK) k_coarse_assemble(CoarseAsmArgs a) {
  const uint32_t t = blockIdx.x * GSFM_BLOCK + threadIdx.x;
  const uint32_t row = t / a.G, lane = t % a.G;
  const bool valid = row < a.n_rows;
  const double scale = a.scale[0];
  const uint32_t cam = a.row_base + (valid ? row : a.n_rows - 1);
  const uint32_t nc = 3 * a.n_agg, I = min(cam / a.chunk, a.n_agg - 1);
  // blocks (I, I-1), (I, I), (I, I+1): in a coherent graph nearly every entry; summed over the wavefront below, one set of atomics each
  double near[3][6] = {{0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0}};
  if (valid) {
    double R[9];
    qmat(load_q(a.q, cam), R);
    uint32_t curJ = 0xffffffffu;
    double acc[6] = {0, 0, 0, 0, 0, 0};
    const uint32_t end = a.row_ptr[row + 1];
    for (uint32_t d = a.row_ptr[row] + lane; d < end; d += a.G) {
      const uint32_t m = a.col[d] & 0x7fffffffu, J = min(m / a.chunk, a.n_agg - 1);
      const double2 A = a.h0[d], B = a.h1[d], C = a.h2[d];
      const double Gs[6] = {A.x, A.y, B.x, B.y, C.x, C.y};
      double S[6];
      sym3_congruence_T(R, Gs, S);
      const uint32_t rel = J + 1 - I;   // 0, 1, 2 for the three near blocks (unsigned wrap-around puts everything else above 2)
      if (rel <= 2) {
#pragma unroll
        for (int w = 0; w < 3; ++w)
          if (rel == (uint32_t)w) {
#pragma unroll
            for (int c = 0; c < 6; ++c) near[w][c] -= S[c];
          }
        continue;
      }
      if (J != curJ) {
        if (curJ != 0xffffffffu) coarse_flush(a.Ac, nc, I, curJ, acc, scale);
        curJ = J;
#pragma unroll
        for (int c = 0; c < 6; ++c) acc[c] = 0.0;
      }
#pragma unroll
      for (int c = 0; c < 6; ++c) acc[c] -= S[c];
    }
    if (curJ != 0xffffffffu) coarse_flush(a.Ac, nc, I, curJ, acc, scale);
    if (lane == 0 && end > a.row_ptr[row]) {   // (a camera without edges is not part of the coarse space)
      double S[6];
      sym3_congruence_T(R, a.Mblk + 6 * (size_t)cam, S);
#pragma unroll
      for (int c = 0; c < 6; ++c) near[1][c] += S[c];
    }
  }
  // one set of atomics per wavefront and block when all its rows belong to the same aggregate (they do, except at the chunk boundaries)
  const uint32_t I0 = __shfl(I, 0);
  const bool uniform = __all(I == I0);
#pragma unroll
  for (int w = 0; w < 3; ++w) {
    const uint32_t J = I + (uint32_t)w - 1u;
    if (J >= a.n_agg) continue;          // (I - 1 of the first aggregate wraps around; I + 1 of the last does not exist)
    if (uniform) {
      double sum6[6];
#pragma unroll
      for (int c = 0; c < 6; ++c) sum6[c] = wave_sum(near[w][c]);
      if ((threadIdx.x & 63) == 0) coarse_flush(a.Ac, nc, I0, J, sum6, scale);
    } else if (valid) {
      coarse_flush(a.Ac, nc, I, J, near[w], scale);
    }
  }
}

// ------------------------------------------------------------------------------------------
// Single-reduction PCG (Chronopoulos & Gear): one mat-vec kernel + one vector kernel per iteration.
//   u = M^-1 r, w = A u, gamma = r.u, delta = w.u
//   beta = gamma/gamma_prev, alpha = gamma / (delta - beta gamma / alpha_prev)
//   p = u + beta p, s = w + beta s, x += alpha p, r -= alpha s, u = M^-1 r
// gamma partials are produced by the vector kernel (for the NEXT iteration), delta partials by the
// mat-vec; every block re-sums the partials in the same order, so all blocks (and all ranks) see
// bit-identical scalars and no finalize launch or atomics are needed.
// ------------------------------------------------------------------------------------------
#define GSFM_MV_MAX_PARTIALS 8192   // the fused mat-vec runs `reps` row groups per workgroup so that its delta partials stay below this
// The PCG's status for the host WITHOUT a stream synchronisation: the last node of a chunk copies the scalar block into mapped host memory and
// stamps it with a device-resident count of such posts (system-scope release); the host polls the stamp.  A read-back costs a blit kernel, the
// return of hipStreamSynchronize and the next launch's way to the GPU -- 14 + 4 + 25 us of idle GPU per look in the latency regime (10k
// cameras / 200k edges: 16 looks per solve, a quarter of the PCG time).  No per-call kernel argument: the node is part of the captured chunk.
#define GSFM_MAIL_WORDS 32
__global__ void k_pcg_mail(const double* __restrict__ sc, int nwords, double* mail, double* counter) {
  for (int k = 0; k < nwords; ++k) mail[k] = sc[k];
  const double c = *counter + 1.0;
  *counter = c;
  __threadfence_system();
  __hip_atomic_store(mail + GSFM_MAIL_WORDS, c, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// The LM loop's look at a trial point in ONE launch: the step's five sums and the trial cost's (the reductions k_sum_partials_multi /
// k_sum_partials would have launched: same routine, same order, same bits, written to the same scalars), then the scalar block's post.
__global__ void __launch_bounds__(GSFM_BLOCK) k_trial_post(double* scal, int sc_step, int sc_trial, const double* __restrict__ step_part, int nb_cam,
                                                           const double* __restrict__ cost_part, int nb_cost, int nwords, double* mail, double* counter, int sc_z) {
  __shared__ double lds[8];
  for (int c = 0; c < 5; ++c) {
    const double t = sum_partials_bcast(step_part + (size_t)c * nb_cam, nb_cam, lds);
    if (threadIdx.x == 0) scal[sc_step + c] = t;
  }
  if (sc_z >= 0) {   // (k_cam_step's sixth sum: the per-camera Jacobi estimate of a loose step's error)
    const double t = sum_partials_bcast(step_part + (size_t)5 * nb_cam, nb_cam, lds);
    if (threadIdx.x == 0) scal[sc_z] = t;
  }
  const double t = sum_partials_bcast(cost_part, nb_cost, lds);
  if (threadIdx.x != 0) return;
  scal[sc_trial] = t;
  for (int k = 0; k < nwords; ++k) mail[k] = scal[k];
  const double c = *counter + 1.0;
  *counter = c;
  __threadfence_system();
  __hip_atomic_store(mail + GSFM_MAIL_WORDS, c, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

struct Cg2Scalars {
  double gamma[2];   // parity-indexed gamma_i
  double alpha[2];
  double gamma0;
  double last_rel;
  int done;
  int iters;
  double tol;        // relative tolerance of the current run (device-resident: see CgScalars::tol)
  double etol2, esum, einc[4];   // energy-norm stopping rule of the loose solves (cg_energy_stop): inc_j = alpha_j gamma_j
  double rz_abs;     // absolute floor on gamma = r.z (k_cam_bound): `tol` never drops below sqrt(rz_abs / gamma0)
};
struct Cg2Args {
  uint32_t n;            // cameras
  int nb_cam;            // blocks of the camera kernels
  int n_part_d;          // number of delta partials (mat-vec blocks, or nb_cam when sharded)
  int par;               // iteration parity
  int first;             // 1 on iteration 0
  int max_iters;
  double tol, etol2;     // (read by k_cg2_init only: the run's tolerances live in Cg2Scalars)
  const double* zbound; double abs_floor2;   // k_cam_bound's B (device scalar; null or 0: no floor) and floor^2
  const double* Minv;
  const double* b;
  double *x, *r, *u, *w, *p, *s;
  double* part_g;        // [2][nb_cam]  (parity-indexed)
  double* part_d;        // [n_part_d]
  Cg2Scalars* sc;
  const double2* q;      // Laplacian form: camera quaternions and
  double* urot;          //   urot_k = R_k^T u_k, written wherever u is (null otherwise): the vector the mat-vec gathers
  // Sharded problems: w lives in the all-gather buffer, one slot of `w_stride` doubles per rank = its slice of w (3 * w_slice doubles)
  // followed by `w_tail` delta partials of its own rows -- the partial dot products travel with A u in the ONE collective of the
  // iteration, every rank sums all tails in the same order.  w_stride == 0: w is a plain vector and part_d a plain array.
  uint32_t w_stride, w_slice, w_tail;
};
__device__ __forceinline__ size_t cg2_w_index(const Cg2Args& a, uint32_t k) {
  return a.w_stride ? (size_t)(k / a.w_slice) * a.w_stride + 3 * (size_t)(k % a.w_slice) : 3 * (size_t)k;
}

// x = 0, r = b, u = M^-1 r, p = s = 0, gamma_0 partials
__global__ void __launch_bounds__(GSFM_BLOCK) k_cg2_init(Cg2Args a) {
  __shared__ double lds[8];
  double v = 0.0;
  const uint32_t k = blockIdx.x * GSFM_BLOCK + threadIdx.x;
  if (k < a.n) {
    const size_t k3 = 3 * (size_t)k;
    const double r[3] = {a.b[k3], a.b[k3 + 1], a.b[k3 + 2]};
    double u[3];
    sym3_mulvec(a.Minv + 6 * (size_t)k, r, u);
#pragma unroll
    for (int c = 0; c < 3; ++c) { a.x[k3 + c] = 0.0; a.r[k3 + c] = r[c]; a.u[k3 + c] = u[c]; a.p[k3 + c] = 0.0; a.s[k3 + c] = 0.0; v += r[c] * u[c]; }
    if (a.urot) {
      const Quat qq{a.q[2 * (size_t)k].x, a.q[2 * (size_t)k].y, a.q[2 * (size_t)k + 1].x, a.q[2 * (size_t)k + 1].y};
      double uu[3];
      rot_transpose_apply(qq, u, uu);
      a.urot[k3] = uu[0]; a.urot[k3 + 1] = uu[1]; a.urot[k3 + 2] = uu[2];
    }
  }
  const double t = block_sum_bcast(v, lds);
  if (threadIdx.x == 0) a.part_g[blockIdx.x] = t;
  if (blockIdx.x == 0 && threadIdx.x == 0) { a.sc->done = 0; a.sc->iters = 0; a.sc->last_rel = 1.0; a.sc->gamma0 = 0.0; a.sc->tol = a.tol;
    { const double bound = a.zbound ? *a.zbound : 0.0; a.sc->rz_abs = bound > 0.0 ? a.abs_floor2 / bound : 0.0; }
    a.sc->etol2 = a.etol2; a.sc->esum = 0.0; a.sc->einc[0] = a.sc->einc[1] = a.sc->einc[2] = a.sc->einc[3] = 0.0; }
}
// Continue a stopped solve to a tighter tolerance.  The recurrence stops at a mat-vec ENTRY (every workgroup takes the same decision from
// the same gamma partials, nothing of the iteration has been written), so clearing the flag lets the next mat-vec -- launched with the
// parity and `first` flag of the iteration that stopped -- take the decision again, against the new tolerance.
__global__ void k_cg2_resume(Cg2Scalars* sc, double tol, double etol2) {
  sc->tol = cg_tol_with_floor(tol, sc->rz_abs, sc->gamma0); sc->etol2 = etol2;
  sc->done = 0;
}

// w = A u on the owned rows (G lanes per row, `reps` row groups per workgroup) + delta partials (unsharded only).
// Written for the latency regime: every load that does not depend on another load of this kernel -- the PCG scalars, the gamma
// partials, the row bounds, the first trip's column / blocks and its gathered vector entry -- is requested before the first use of
// any of them, so a workgroup pays ~3 dependent memory round trips (row bounds -> column -> gather) instead of one per stage.
struct MatvecCgArgs { MatvecArgs mv; Cg2Args cg; int with_dots; uint32_t reps; };
template <bool LAP>
__device__ __forceinline__ void mv_entry(const MatvecArgs& a, const double* Rk, uint32_t m, const double2& A, const double2& B, const double2& C,
                                         const double2& D, double E, const double* v, double& y0, double& y1, double& y2) {
  if (LAP) {
    const double w0 = Rk[0] * v[0] + Rk[1] * v[1] + Rk[2] * v[2], w1 = Rk[3] * v[0] + Rk[4] * v[1] + Rk[5] * v[2], w2 = Rk[6] * v[0] + Rk[7] * v[1] + Rk[8] * v[2];
    y0 += A.x * w0 + A.y * w1 + B.x * w2;
    y1 += A.y * w0 + B.y * w1 + C.x * w2;
    y2 += B.x * w0 + C.x * w1 + C.y * w2;
  } else {
    y0 += A.x * v[0] + A.y * v[1] + B.x * v[2];
    y1 += B.y * v[0] + C.x * v[1] + C.y * v[2];
    y2 += D.x * v[0] + D.y * v[1] + E * v[2];
  }
}
template <bool LAP>
__global__ void __launch_bounds__(GSFM_BLOCK) k_matvec_cg(MatvecCgArgs aa) {
  __shared__ double lds[8];
  const MatvecArgs& a = aa.mv;
  const Cg2Args& c = aa.cg;
  const double* __restrict__ vec = LAP ? a.u : a.p;   // the gathered vector: R^T u (Laplacian form) or u itself
  // ---- request phase ----
  const int done = c.sc->done, iters = c.sc->iters;
  const double gamma0 = c.sc->gamma0, tol = c.sc->tol;
  const bool estop = cg_energy_stop(c.sc->einc, c.sc->esum, c.sc->etol2, iters);
  double gpart = 0.0;
  for (int k = threadIdx.x; k < c.nb_cam; k += GSFM_BLOCK) gpart += c.part_g[(size_t)c.par * c.nb_cam + k];
  const uint32_t G = a.G, rows_per_group = GSFM_BLOCK / G;
  const uint32_t lane = threadIdx.x % G, sub = threadIdx.x / G;
  uint32_t row = blockIdx.x * aa.reps * rows_per_group + sub;
  bool live = row < a.n_rows;
  uint32_t d = 0, end = 0;
  if (live) { d = a.row_ptr[row] + lane; end = a.row_ptr[row + 1]; }
  bool has = live && d < end;
  uint32_t m = 0;
  double2 A = make_double2(0, 0), B = A, C = A, D = A;
  double E = 0.0, v[3] = {0, 0, 0};
  Quat qk{0, 0, 0, 1};
  if (live && LAP) qk = load_q(a.q, a.row_base + row);
  double M0[6] = {0, 0, 0, 0, 0, 0}, pk0[3] = {0, 0, 0};   // the row owner's diagonal block and vector entry (first row group)
  if (live && lane == 0) {
    const size_t k = a.row_base + row;
#pragma unroll
    for (int t = 0; t < 6; ++t) M0[t] = a.Mblk[6 * k + t];
#pragma unroll
    for (int t = 0; t < 3; ++t) pk0[t] = a.p[3 * k + t];
  }
  if (has) {
    m = __builtin_nontemporal_load(a.col + d) & 0x7fffffffu;
    A = nt_load2(a.h0 + d); B = nt_load2(a.h1 + d); C = nt_load2(a.h2 + d);
    if (!LAP) { D = nt_load2(a.h3 + d); E = __builtin_nontemporal_load(a.h4 + d); }
    const double* vm = vec + 3 * (size_t)m;
    v[0] = vm[0]; v[1] = vm[1]; v[2] = vm[2];
  }
  // ---- convergence (same decision in every workgroup: same partials, same order) ----
  const double gamma = block_sum_bcast(gpart, lds);
  if (done) return;
  bool conv;
  if (c.first) {
    const double rz_abs = c.sc->rz_abs;   // (written by k_cg2_init, the launch before)
    conv = !(gamma > rz_abs);
    if (blockIdx.x == 0 && threadIdx.x == 0) { c.sc->gamma0 = gamma; c.sc->tol = cg_tol_with_floor(tol, rz_abs, gamma); if (conv) { c.sc->done = 1; c.sc->last_rel = 0.0; } }
  } else {
    const double rel = sqrt(gamma / gamma0);
    // the iteration cap is applied here, at a kernel entry, from a counter written by the PREVIOUS launch: every workgroup takes
    // the same decision, and no workgroup of a vector-update launch can see the flag flip half way through an update of x
    conv = !(rel > tol) || iters >= c.max_iters || estop;
    if (blockIdx.x == 0 && threadIdx.x == 0) { c.sc->last_rel = rel; if (conv) c.sc->done = 1; }
  }
  if (conv) return;
  // ---- rows ----
  double dpart = 0.0;
  for (uint32_t rep = 0; rep < aa.reps; ++rep) {
    if (rep > 0) {
      row = (blockIdx.x * aa.reps + rep) * rows_per_group + sub;
      live = row < a.n_rows;
      d = 0; end = 0;
      if (live) { d = a.row_ptr[row] + lane; end = a.row_ptr[row + 1]; if (LAP) qk = load_q(a.q, a.row_base + row); }
      has = false;   // no prefetched entry: the loop below starts at d
    }
    double y0 = 0.0, y1 = 0.0, y2 = 0.0;
    if (live) {
      double Rk[9];
      if (LAP) qmat(qk, Rk);
      if (has) { mv_entry<LAP>(a, Rk, m, A, B, C, D, E, v, y0, y1, y2); d += G; }
      for (; d < end; d += G) {
        const uint32_t mm = __builtin_nontemporal_load(a.col + d) & 0x7fffffffu;
        const double2 A2 = nt_load2(a.h0 + d), B2 = nt_load2(a.h1 + d), C2 = nt_load2(a.h2 + d);
        double2 D2 = make_double2(0, 0); double E2 = 0.0;
        if (!LAP) { D2 = nt_load2(a.h3 + d); E2 = __builtin_nontemporal_load(a.h4 + d); }
        const double* vm = vec + 3 * (size_t)mm;
        const double vv[3] = {vm[0], vm[1], vm[2]};
        mv_entry<LAP>(a, Rk, mm, A2, B2, C2, D2, E2, vv, y0, y1, y2);
      }
    }
    for (uint32_t off = G >> 1; off > 0; off >>= 1) {
      y0 += __shfl_down(y0, off, G); y1 += __shfl_down(y1, off, G); y2 += __shfl_down(y2, off, G);
    }
    if (live && lane == 0) {
      const size_t k = a.row_base + row;
      if (rep > 0) {
#pragma unroll
        for (int t = 0; t < 6; ++t) M0[t] = a.Mblk[6 * k + t];
#pragma unroll
        for (int t = 0; t < 3; ++t) pk0[t] = a.p[3 * k + t];
      }
      double mp[3];
      sym3_mulvec(M0, pk0, mp);
      const double sgn = LAP ? -1.0 : 1.0;
      const double w0 = mp[0] + sgn * y0, w1 = mp[1] + sgn * y1, w2 = mp[2] + sgn * y2;
      a.y[3 * k] = w0; a.y[3 * k + 1] = w1; a.y[3 * k + 2] = w2;
      dpart += w0 * pk0[0] + w1 * pk0[1] + w2 * pk0[2];
    }
  }
  if (aa.with_dots) {
    const double t = block_sum_bcast(dpart, lds);
    if (threadIdx.x == 0) aa.cg.part_d[blockIdx.x] = t;
  }
}

// sharded path: delta partials over ALL cameras after the all-gather of w (identical on every rank)
__global__ void __launch_bounds__(GSFM_BLOCK) k_cg2_dots(Cg2Args a) {
  if (a.sc->done) return;
  __shared__ double lds[8];
  double v = 0.0;
  const uint32_t k = blockIdx.x * GSFM_BLOCK + threadIdx.x;
  if (k < a.n) {
    const size_t k3 = 3 * (size_t)k;
    v = a.w[k3] * a.u[k3] + a.w[k3 + 1] * a.u[k3 + 1] + a.w[k3 + 2] * a.u[k3 + 2];
  }
  const double t = block_sum_bcast(v, lds);
  if (threadIdx.x == 0) a.part_d[blockIdx.x] = t;
}

// alpha, beta and all five vector updates; gamma partials of the next iteration.  Like the mat-vec above, all loads are requested
// before the first reduction (one memory round trip, then two workgroup reductions, then the stores).
__global__ void __launch_bounds__(GSFM_BLOCK) k_cg2_step(Cg2Args a) {
  __shared__ double lds[8];
  const int done = a.sc->done;
  const double gamma_prev = a.sc->gamma[a.par ^ 1], alpha_prev = a.sc->alpha[a.par ^ 1];
  double gpart = 0.0, dsum = 0.0;
  for (int k = threadIdx.x; k < a.nb_cam; k += GSFM_BLOCK) gpart += a.part_g[(size_t)a.par * a.nb_cam + k];
  if (a.w_stride) {
    for (int k = threadIdx.x; k < a.n_part_d; k += GSFM_BLOCK) dsum += a.w[(size_t)(k / a.w_tail) * a.w_stride + 3 * (size_t)a.w_slice + k % a.w_tail];
  } else for (int k = threadIdx.x; k < a.n_part_d; k += GSFM_BLOCK) dsum += a.part_d[k];
  const uint32_t k = blockIdx.x * GSFM_BLOCK + threadIdx.x;
  const bool live = k < a.n;
  const size_t k3 = 3 * (size_t)(live ? k : 0), kw = cg2_w_index(a, live ? k : 0);
  double uo[3], po[3], wo[3], so[3], xo[3], ro[3], Mi[6];
  Quat qq{0, 0, 0, 1};
#pragma unroll
  for (int c = 0; c < 3; ++c) { uo[c] = a.u[k3 + c]; po[c] = a.p[k3 + c]; wo[c] = a.w[kw + c]; so[c] = a.s[k3 + c]; xo[c] = a.x[k3 + c]; ro[c] = a.r[k3 + c]; }
#pragma unroll
  for (int c = 0; c < 6; ++c) Mi[c] = a.Minv[2 * k3 + c];
  if (a.urot) qq = load_q(a.q, live ? k : 0);
  const double gamma = block_sum_bcast(gpart, lds);
  const double delta = block_sum_bcast(dsum, lds);
  if (done) return;
  double beta, alpha;
  if (a.first) { beta = 0.0; alpha = gamma / delta; }
  else {
    beta = gamma / gamma_prev;
    alpha = gamma / (delta - beta * gamma / alpha_prev);
  }
  double v = 0.0;
  if (live) {
    double r[3], u[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const double p = uo[c] + beta * po[c];
      const double s = wo[c] + beta * so[c];
      a.p[k3 + c] = p; a.s[k3 + c] = s;
      a.x[k3 + c] = xo[c] + alpha * p;
      r[c] = ro[c] - alpha * s;
      a.r[k3 + c] = r[c];
    }
    sym3_mulvec(Mi, r, u);
#pragma unroll
    for (int c = 0; c < 3; ++c) { a.u[k3 + c] = u[c]; v += r[c] * u[c]; }
    if (a.urot) {
      double uu[3];
      rot_transpose_apply(qq, u, uu);
      a.urot[k3] = uu[0]; a.urot[k3 + 1] = uu[1]; a.urot[k3 + 2] = uu[2];
    }
  }
  const double t = block_sum_bcast(v, lds);
  if (threadIdx.x == 0) a.part_g[(size_t)(a.par ^ 1) * a.nb_cam + blockIdx.x] = t;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    const int it = a.sc->iters;
    const double inc = alpha * gamma;
    a.sc->gamma[a.par] = gamma; a.sc->alpha[a.par] = alpha; a.sc->einc[it & 3] = inc; a.sc->esum += inc; a.sc->iters = it + 1;
  }
}

// ------------------------------------------------------------------------------------------
// Small graphs: assemble the damped normal matrix for the exact Cholesky step (dense_kernels.hpp) -- lower triangle only,
// as 32 x 32 tiles, plus the right-hand side -g as block row T.  The block-CSR holds both directions of every edge; the
// lower triangle takes the entry whose row camera has the larger index (H_km for k > m), so atomicAdd only matters for
// repeated camera pairs (two contributions commute).  A is zero-filled before the launch.
// ------------------------------------------------------------------------------------------
struct DenseArgs {
  uint32_t n_rows;
  const uint32_t* row_ptr;
  const uint32_t* col;
  const double2 *h0, *h1, *h2, *h3;
  const double* h4;
  const double* Mblk;  // 6 per camera
  const double* b;     // 3 per camera: right-hand side
  double* A;           // tiles, see chol_tile_off
  uint32_t n, T;
  const double2* q;    // Laplacian form (lap = 1): planes h0..h2 hold G_k, the block is -G_k R_k R_m^T
  int lap;
  double* info_slot;   // status word of the factorisation (an int in a double slot of the scalar block) and
  double* rcg;         // the PCG residual (3 per camera, zero for an exact solve): cleared here instead of by two more memset nodes
};
__device__ __forceinline__ double* dense_elem(double* A, uint32_t gr, uint32_t gc) {
  return A + ((size_t)(gr / 32) * (gr / 32 + 1) / 2 + gc / 32) * 1024 + (gr % 32) * 32 + gc % 32;
}
__global__ void __launch_bounds__(GSFM_BLOCK) k_dense_assemble(DenseArgs a) {
  const uint32_t row = blockIdx.x;
  if (row >= a.n_rows) return;
  if (threadIdx.x < 3) a.rcg[3 * (size_t)row + threadIdx.x] = 0.0;
  if (row == 0 && threadIdx.x == 3) *a.info_slot = 0.0;
  if (threadIdx.x == 0) {
    const double* M = a.Mblk + 6 * (size_t)row;
    const double m[9] = {M[0], M[1], M[2], M[1], M[3], M[4], M[2], M[4], M[5]};
    for (int r = 0; r < 3; ++r) for (int c = 0; c <= r; ++c) *dense_elem(a.A, 3 * row + r, 3 * row + c) = m[3 * r + c];
    for (int c = 0; c < 3; ++c) a.A[(((size_t)a.T * (a.T + 1) / 2) + (3 * row + c) / 32) * 1024 + (3 * row + c) % 32] = a.b[3 * (size_t)row + c];
    if (row == 0) for (uint32_t g = a.n; g < a.T * 32; ++g) *dense_elem(a.A, g, g) = 1.0;   // padding of the last tile: identity
  }
  for (uint32_t d = a.row_ptr[row] + threadIdx.x; d < a.row_ptr[row + 1]; d += GSFM_BLOCK) {
    const uint32_t m = a.col[d] & 0x7fffffffu;
    if (m >= row) continue;   // upper triangle (and self loops, which cannot occur)
    double H[9];
    if (a.lap) {
      const double2 A0 = a.h0[d], B0 = a.h1[d], C0 = a.h2[d];
      const double Gm[9] = {A0.x, A0.y, B0.x, A0.y, B0.y, C0.x, B0.x, C0.x, C0.y};
      double Rk[9], Rm[9], T[9];
      qmat(load_q(a.q, row), Rk);
      qmat(load_q(a.q, m), Rm);
      for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) T[3 * r + c] = Rk[3 * r] * Rm[3 * c] + Rk[3 * r + 1] * Rm[3 * c + 1] + Rk[3 * r + 2] * Rm[3 * c + 2];   // R_k R_m^T
      for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) H[3 * r + c] = -(Gm[3 * r] * T[c] + Gm[3 * r + 1] * T[3 + c] + Gm[3 * r + 2] * T[6 + c]);
    } else {
      const double2 A0 = a.h0[d], B0 = a.h1[d], C0 = a.h2[d], D0 = a.h3[d];
      H[0] = A0.x; H[1] = A0.y; H[2] = B0.x; H[3] = B0.y; H[4] = C0.x; H[5] = C0.y; H[6] = D0.x; H[7] = D0.y; H[8] = a.h4[d];
    }
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) atomicAdd(dense_elem(a.A, 3 * row + r, 3 * m + c), H[3 * r + c]);
  }
}

// ------------------------------------------------------------------------------------------
// Device-side Levenberg-Marquardt control for EXACT steps (latency regime: Madrid-sized graphs, one Cholesky step per iteration).  The
// decisions of TrustRegionMinimizer the host loop takes between two synchronisations -- step validity, the two tolerance tests, acceptance,
// the radius law -- are taken by one lane from the scalars the step and cost kernels left on the device; the kernels of the accept path (state
// copy, linearisation) are predicated on its verdict and the damping is rebuilt from the radius it wrote, so a whole LM iteration is
// enqueued without a host decision and read back ONCE (solver_lm.hpp).  Same formulas, same operation order as the host loop: the two
// controls produce bit-identical trajectories (tests/test_gpu_round4.py).
// ------------------------------------------------------------------------------------------
enum { CT_RADIUS = 0, CT_DF = 1, CT_XCOST = 2, CT_XNORM = 3, CT_GMAX = 4, CT_ACCEPT = 5, CT_TERM = 6 /* -1: go on */, CT_NINVALID = 7, CT_VALID = 8,
       CT_CAND = 9, CT_CC = 10, CT_MCC = 11, CT_STEPN = 12, CT_DENSE_FAIL = 13, CT_NONFINITE = 14, CT_SKIPPED = 15 /* this iteration was enqueued ahead of a verdict that ended the run: nothing was decided */, CT_N = 16 };
struct LmOpts { double function_tolerance, gradient_tolerance, parameter_tolerance, min_relative_decrease, max_radius, min_radius; };
// (2 rel_dec - 1)^3 of the radius law as Ceres rounds it: std::pow(t, 3) is the correctly rounded cube (glibc: < 0.52 ulp), t * t * t is two roundings.
// t^2 = h + l and h t = p + e exactly (FMA residues), the cube is p + (e + l t): one rounding of a value good to 2^-100, i.e. the correctly rounded
// result but for ties nobody will meet -- the device's radius trace equals the oracle's bit for bit (tests/test_gpu_round5.py).
__device__ __forceinline__ double lm_cube(double t) {
#pragma clang fp contract(off)   // (under the device default, -ffp-contract=fast, `p + fma(l, t, e)` becomes fma(h, t, fma(l, t, e)), which counts the residue e twice;
                                 // HIP's __dmul_rn / __dadd_rn are plain operators and do not stop it)
  const double h = t * t, l = fma(t, t, -h);
  const double p = h * t, e = fma(h, t, -p);
  return p + fma(l, t, e);
}
// `it_dev`: the number of the LM iteration the next k_lm_after stamps its record with -- on the device, so that no kernel of an iteration takes a
// per-iteration argument and the whole iteration replays as one hipGraph (solver_lm.hpp)
__global__ void k_set_double(double* p, double v) { *p = v; }   // one device word from a host value, in stream order (no staging buffer to keep alive)
__global__ void k_lm_set(double* ctl, double radius, double df, double x_cost, double x_norm, double gmax, double n_invalid, double* it_dev, double iteration) {
  *it_dev = iteration;
  ctl[CT_RADIUS] = radius; ctl[CT_DF] = df; ctl[CT_XCOST] = x_cost; ctl[CT_XNORM] = x_norm; ctl[CT_GMAX] = gmax; ctl[CT_NINVALID] = n_invalid;
  ctl[CT_ACCEPT] = 0.0; ctl[CT_TERM] = -1.0; ctl[CT_DENSE_FAIL] = 0.0; ctl[CT_NONFINITE] = 0.0; ctl[CT_SKIPPED] = 0.0;
}
// scal: SC_STEP.. = eta.g, eta.r, eta^T Lam eta, |delta|^2, |x_trial|^2 ; trial cost ; dense status.  (indices passed in: the enum lives on the host side)
// The control block is authoritative between host interventions (k_lm_set): iteration k + 1 may be enqueued before the host has read
// iteration k's verdict, so a verdict that ends the run of exact steps -- a termination, a factor that broke down -- must stop every later
// decision: such an iteration is marked SKIPPED, accepts nothing and leaves the block alone.
__device__ __forceinline__ void lm_decide_body(const LmOpts& o, const double* scal, int sc_step, int sc_trial, int sc_info, double* ctl) {
  ctl[CT_ACCEPT] = 0.0;
  if (ctl[CT_TERM] >= 0.0 || ctl[CT_DENSE_FAIL] != 0.0) { ctl[CT_SKIPPED] = 1.0; return; }
  ctl[CT_SKIPPED] = 0.0; ctl[CT_VALID] = 0.0; ctl[CT_NONFINITE] = 0.0;
  int info;
  __builtin_memcpy(&info, scal + sc_info, sizeof(int));
  if (info != 0) { ctl[CT_DENSE_FAIL] = 1.0; return; }   // the factor broke down: the step is meaningless, the host solves it again by PCG
  const double eta_g = scal[sc_step], eta_r = scal[sc_step + 1], eta_L = scal[sc_step + 2];
  const double mcc = -0.5 * eta_g + 0.5 * eta_r + 0.5 * eta_L;
  ctl[CT_MCC] = mcc;
  double radius = ctl[CT_RADIUS], df = ctl[CT_DF];
  if (!(isfinite(mcc) && mcc > 0.0)) {   // HandleInvalidStep
    const double ni = ctl[CT_NINVALID] + 1.0;
    ctl[CT_NINVALID] = ni;
    if (ni >= 5.0) { ctl[CT_TERM] = 4.0; return; }
    ctl[CT_RADIUS] = radius / df; ctl[CT_DF] = df * 2.0;
    return;
  }
  ctl[CT_VALID] = 1.0; ctl[CT_NINVALID] = 0.0;
  double cand = scal[sc_trial];
  if (!isfinite(cand)) { cand = 1.7976931348623157e308; ctl[CT_NONFINITE] = 1.0; }
  const double x_cost = ctl[CT_XCOST], x_norm = ctl[CT_XNORM];
  const double step_norm = sqrt(scal[sc_step + 3]), cost_change = x_cost - cand, rel_dec = cost_change / mcc;
  ctl[CT_CAND] = cand; ctl[CT_CC] = cost_change; ctl[CT_STEPN] = step_norm;
  if (step_norm <= o.parameter_tolerance * (x_norm + o.parameter_tolerance)) { ctl[CT_TERM] = 2.0; return; }
  if (fabs(cost_change) <= o.function_tolerance * x_cost) { ctl[CT_TERM] = 0.0; return; }
  if (rel_dec > o.min_relative_decrease) {   // HandleSuccessfulStep
    ctl[CT_ACCEPT] = 1.0;
    ctl[CT_XNORM] = sqrt(scal[sc_step + 4]); ctl[CT_XCOST] = cand;
    radius = radius / fmax(1.0 / 3.0, 1.0 - lm_cube(2.0 * rel_dec - 1.0));
    ctl[CT_RADIUS] = fmin(o.max_radius, radius); ctl[CT_DF] = 2.0;
  } else { ctl[CT_RADIUS] = radius / df; ctl[CT_DF] = df * 2.0; }
}
// One workgroup closes the step: the five sums of k_cam_step's partials and the trial cost's (the reductions k_sum_partials_multi /
// k_sum_partials would have launched: same routine, same order, same bits, written to the same scalars), the decision (one lane), and -- if the
// step is accepted -- x <- x_trial, q <- q_trial (copies, not pointer swaps: captured graphs hold the addresses).  Three launches fewer per
// exact LM iteration than sum, sum, decide, accept (~4.5 us each on a chain of ~60 dependent launches).
__global__ void __launch_bounds__(GSFM_BLOCK) k_lm_decide(LmOpts o, double* scal, int sc_step, int sc_trial, int sc_info, double* ctl,
                                                          const double* __restrict__ step_part, int nb_cam, const double* __restrict__ cost_part, int nb_cost,
                                                          uint32_t n, int param_dim, double* x, const double* __restrict__ x_trial, double2* q, const double2* __restrict__ q_trial) {
  __shared__ double lds[8];
  for (int c = 0; c < 5; ++c) {
    const double t = sum_partials_bcast(step_part + (size_t)c * nb_cam, nb_cam, lds);
    if (threadIdx.x == 0) scal[sc_step + c] = t;
  }
  {
    const double t = sum_partials_bcast(cost_part, nb_cost, lds);
    if (threadIdx.x == 0) scal[sc_trial] = t;
  }
  if (threadIdx.x == 0) {
    lm_decide_body(o, scal, sc_step, sc_trial, sc_info, ctl);
    lds[5] = ctl[CT_ACCEPT];
  }
  __syncthreads();
  if (lds[5] == 0.0) return;
  for (uint32_t k = threadIdx.x; k < n; k += GSFM_BLOCK) {
    for (int c = 0; c < param_dim; ++c) x[(size_t)param_dim * k + c] = x_trial[(size_t)param_dim * k + c];
    q[2 * (size_t)k] = q_trial[2 * (size_t)k]; q[2 * (size_t)k + 1] = q_trial[2 * (size_t)k + 1];
  }
}
// after the (predicated) linearisation and the damping rebuild: the gradient test of an accepted step, the radius floor
// ... and the iteration's record for the host: the control block as this iteration left it, in its slot of a ring (the host reads it from a
// side stream while the next iteration is already running).  `gterm`: the gradient / radius verdicts are kept apart from CT_TERM in the record
// (the host loop takes them at the top of the NEXT iteration, after recording this one) but halt later decisions just the same.
// `rec` is host memory mapped into the device (the host polls the record's last word instead of synchronising a stream: a cross-stream event
// costs tens of microseconds per iteration, more than the gap it was meant to close); `stamp` = the LM iteration, written last, system scope.
// (round 4: the max-norm reduction of k_cam_prep's partials -- k_max_partials, same routine -- is done here, one launch fewer)
__global__ void __launch_bounds__(GSFM_BLOCK) k_lm_after(LmOpts o, double* scal, int sc_gmax, double* ctl, double* rec_ring, int rec_stride, double* it_dev, const double* __restrict__ gmax_part, int nb_cam) {
  __shared__ double lds[8];
  {
    double v = 0.0;
    for (int k = threadIdx.x; k < nb_cam; k += GSFM_BLOCK) v = fmax(v, gmax_part[k]);
    const double t = block_max_bcast(v, lds);
    if (threadIdx.x != 0) return;
    scal[sc_gmax] = t;
  }
  if (ctl[CT_SKIPPED] != 0.0) return;
  const double stamp = *it_dev;                                  // this iteration's number; the next one's is one more
  double* const rec = rec_ring + (size_t)rec_stride * ((int)stamp & 3);
  *it_dev = stamp + 1.0;
  double term_next = -1.0;
  if (ctl[CT_TERM] < 0.0 && ctl[CT_DENSE_FAIL] == 0.0) {
    if (ctl[CT_ACCEPT] != 0.0) {
      ctl[CT_GMAX] = scal[sc_gmax];
      if (ctl[CT_GMAX] <= o.gradient_tolerance) term_next = 1.0;
    }
    if (term_next < 0.0 && ctl[CT_RADIUS] <= o.min_radius) term_next = 4.0;
  }
  for (int k = 0; k < CT_N; ++k) rec[k] = ctl[k];
  __threadfence_system();
  __hip_atomic_store(rec + CT_N, stamp, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  if (term_next >= 0.0) ctl[CT_TERM] = term_next;   // (after the copy: the record shows the iteration's own verdict)
}

}  // namespace gsfm

// Exact LM step for small graphs: Cholesky of the dense damped normal matrix (3N x 3N) and the two triangular solves -- "normal
// equations + Cholesky", which is literally what the reference's SPARSE_NORMAL_CHOLESKY does
// (src/GSfM_nonlinear_rotation_estimator.cpp:72-74,176-179,299-302).
//
// The regime is launch-latency-bound, not flop-bound (3N <= ~4500: <= 30 GFLOP per factorisation), so the design minimises the
// length of the dependent chain instead of the arithmetic:
//   * the matrix lives as 32 x 32 tiles of the lower triangle (8 KiB, contiguous), plus one extra tile row holding the right-hand
//     side, so the forward substitution L y = b falls out of the factorisation (y is the last row of L);
//   * ONE kernel per block column k (right-looking): every workgroup factors the 32 x 32 diagonal block A_kk itself (one wavefront,
//     rows in registers, v_readlane broadcasts: no inter-workgroup dependency inside a step), solves the two panel tiles it needs by
//     substitution and updates its own trailing tile A_ij -= L_ik L_jk^T.  The redundant work is ~3x the flops of the textbook
//     schedule and irrelevant here; the chain per step is one kernel instead of three (round 1: 111 launches, 2.2 ms at 3N = 1182);
//   * L goes to a second buffer (a workgroup's inputs A_kk, A_ik, A_jk are never written during step k, so there is no race);
//   * the backward substitution L^T x = y is one workgroup sweeping the block rows of L bottom-up.
// fp64 VALU throughout; MFMA tiles would buy nothing at these sizes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gsfm {

#define GSFM_CB 32
#define GSFM_TILE_ELEMS (GSFM_CB * GSFM_CB)
#ifndef GSFM_CHOL_BATCH
#define GSFM_CHOL_BATCH 8
#endif
#define GSFM_DENSE_MAX_T 160   // block rows the backward kernel keeps in LDS: 3N <= 5120

__host__ __device__ inline size_t chol_tile_off(uint32_t i, uint32_t j) { return ((size_t)i * (i + 1) / 2 + j) * GSFM_TILE_ELEMS; }
__host__ __device__ inline size_t chol_num_tiles(uint32_t T) { return (size_t)(T + 1) * (T + 2) / 2; }   // block rows 0..T (row T = rhs)

// broadcast one lane's double through the scalar unit (v_readlane_b32 x 2): a few cycles, where a shuffle through the LDS crossbar
// (ds_bpermute) is ~120 cycles of latency in a dependent chain.  `lane` must be wave-uniform.
__device__ __forceinline__ double readlane_f64(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

struct CholArgs {
  double* A;      // tiles (i >= j), block rows 0..T; row T = right-hand side (first row of each tile)
  double* L;      // same layout: L_ik (i > k), L_kk, and in row T the forward-substituted y = L^-1 b
  uint32_t T;     // block rows of the matrix proper = ceil(3N / 32)
  uint32_t k;     // this step's block column
  int* info;      // 0, or 1 + index of the first non-positive pivot
};

// Step k.  Workgroup 0: L_kk and the right-hand side's panel tile; workgroup 1 + t(t+1)/2 + u: trailing tile (k+1+t, k+1+u), u <= t.
// Wavefront 0 holds the rows of A_kk in lanes 0..31 and the rows of the panel tile A_ik in lanes 32..63, one row per lane in
// registers, and runs the right-looking elimination on all 64 rows at once: for the lower lanes that is the Cholesky factorisation,
// for the upper lanes the same instructions are the substitution P_i = A_ik L_kk^-T (the multipliers L[c][c0] are wave-uniform
// v_readlane broadcasts from the diagonal rows), so the panel costs nothing extra.  Wavefront 1 does the same with A_jk.
__global__ void __launch_bounds__(256) k_chol_step(CholArgs a) {
  __shared__ double Pi[GSFM_CB][GSFM_CB + 1], Pj[GSFM_CB][GSFM_CB + 1];
  const uint32_t k = a.k, T = a.T, tid = threadIdx.x;
  uint32_t i, j;
  const bool diag_wg = blockIdx.x == 0;
  if (diag_wg) { i = T; j = k; }
  else {
    const uint32_t b = blockIdx.x - 1;
    uint32_t t = (uint32_t)((sqrt(8.0 * (double)b + 1.0) - 1.0) * 0.5);
    while ((uint64_t)(t + 1) * (t + 2) / 2 <= b) ++t;
    while ((uint64_t)t * (t + 1) / 2 > b) --t;
    i = k + 1 + t; j = k + 1 + (b - t * (t + 1) / 2);
    if (i == T && j == T) return;   // the right-hand side has no diagonal tile
  }
  const bool same = !diag_wg && i == j;
  const uint32_t ur = tid / 8, uc4 = (tid % 8) * 4;   // this lane's 1 x 4 piece of the trailing tile
  double own[4] = {0, 0, 0, 0};
  if (!diag_wg) {
    const double* so = a.A + chol_tile_off(i, j) + ur * GSFM_CB + uc4;
#pragma unroll
    for (int q = 0; q < 4; ++q) own[q] = so[q];
  }
  const uint32_t wave = tid >> 6, lane = tid & 63;
  if (wave == 0 || (wave == 1 && !diag_wg && !same)) {
    const uint32_t rr = lane & 31;
    const double2* src = (const double2*)(a.A + (lane < 32 ? chol_tile_off(k, k) : chol_tile_off(wave == 0 ? i : j, k)) + rr * GSFM_CB);
    double r[GSFM_CB];
#pragma unroll
    for (int q = 0; q < GSFM_CB / 2; ++q) { const double2 v = src[q]; r[2 * q] = v.x; r[2 * q + 1] = v.y; }
    int bad = 0;
#pragma unroll
    for (int c0 = 0; c0 < GSFM_CB; ++c0) {
      double piv = readlane_f64(r[c0], c0);
      if (!(piv > 0.0)) { if (!bad) bad = c0 + 1; piv = 1.0; }
      const double inv = rsqrt(piv);
      r[c0] = (lane == (uint32_t)c0) ? piv * inv : r[c0] * inv;
      // entries above the diagonal of the diagonal rows (lane < c) are never read by anyone, so they may hold anything
      // multipliers L[c][c0] (held by lane c) in batches: all v_readlane of a batch first, then the FMAs -- one SGPR-hazard wait per
      // batch instead of one per multiplier
#pragma unroll
      for (int cb = c0 + 1; cb < GSFM_CB; cb += GSFM_CHOL_BATCH) {
        double m[GSFM_CHOL_BATCH];
#pragma unroll
        for (int q = 0; q < GSFM_CHOL_BATCH; ++q) if (cb + q < GSFM_CB) m[q] = readlane_f64(r[c0], cb + q);
#pragma unroll
        for (int q = 0; q < GSFM_CHOL_BATCH; ++q) if (cb + q < GSFM_CB) r[cb + q] -= r[c0] * m[q];
      }
    }
    if (lane >= 32) {
      double (*P)[GSFM_CB + 1] = wave == 0 ? Pi : Pj;
#pragma unroll
      for (int c = 0; c < GSFM_CB; ++c) P[rr][c] = r[c];
    } else if (diag_wg) {
      double2* dl = (double2*)(a.L + chol_tile_off(k, k) + rr * GSFM_CB);
#pragma unroll
      for (int q = 0; q < GSFM_CB / 2; ++q) dl[q] = make_double2((uint32_t)(2 * q) <= lane ? r[2 * q] : 0.0, (uint32_t)(2 * q + 1) <= lane ? r[2 * q + 1] : 0.0);
      if (lane == 0 && bad && *a.info == 0) *a.info = (int)(k * GSFM_CB + bad);
    }
  }
  __syncthreads();
  if (diag_wg) {   // y_k = row T of L
    double* dy = a.L + chol_tile_off(T, k) + ur * GSFM_CB + uc4;
#pragma unroll
    for (int q = 0; q < 4; ++q) dy[q] = Pi[ur][uc4 + q];
    return;
  }
  if (j == k + 1 && i < T) {   // first trailing column: this workgroup publishes L_ik
    double* dl = a.L + chol_tile_off(i, k) + ur * GSFM_CB + uc4;
#pragma unroll
    for (int q = 0; q < 4; ++q) dl[q] = Pi[ur][uc4 + q];
  }
  double (*Q)[GSFM_CB + 1] = same ? Pi : Pj;
  double acc[4] = {0, 0, 0, 0};
#pragma unroll
  for (int t = 0; t < GSFM_CB; ++t) {
    const double pv = Pi[ur][t];
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] += pv * Q[uc4 + q][t];
  }
  double* d = a.A + chol_tile_off(i, j) + ur * GSFM_CB + uc4;
#pragma unroll
  for (int q = 0; q < 4; ++q) d[q] = own[q] - acc[q];
}

// x = L^-T y, one workgroup: block rows bottom-up; y stays in LDS.  Software-pipelined: while the other 15 wavefronts fold x_k into the
// right-hand sides of the block rows j <= k - 2 (L_kj tiles are contiguous: block row k of L) and prefetch the next diagonal tile,
// wavefront 0 folds x_k into block row k - 1 and runs that block's 32-step substitution (running right-hand side in lane registers,
// solved components broadcast with v_readlane) -- one barrier per block row.
__device__ __forceinline__ void chol_back_block(const double (*Lk)[GSFM_CB + 1], double v, uint32_t lane, double* xk_out, double* x, uint32_t g0, uint32_t n) {
  const uint32_t l = lane & 31;
  const double rinv = 1.0 / Lk[l][l];   // all reciprocals at once, off the dependent chain
#pragma unroll
  for (int t = GSFM_CB - 1; t >= 0; --t) {
    const double xt = readlane_f64(v * rinv, t);
    if (l < (uint32_t)t) v -= Lk[t][l] * xt;
  }
  if (lane < GSFM_CB) { const double mine = v * rinv; xk_out[lane] = mine; if (g0 + lane < n) x[g0 + lane] = mine; }   // lane t was never modified after step t
}
__global__ void __launch_bounds__(1024) k_chol_back(const double* __restrict__ L, uint32_t n, uint32_t T, double* __restrict__ x) {
  __shared__ double y[GSFM_DENSE_MAX_T * GSFM_CB];
  __shared__ double Lk[2][GSFM_CB][GSFM_CB + 1];
  __shared__ double xk[2][GSFM_CB];
  const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  for (uint32_t idx = tid; idx < T * GSFM_CB; idx += 1024) y[idx] = L[chol_tile_off(T, idx / GSFM_CB) + idx % GSFM_CB];
  Lk[(T - 1) & 1][tid / GSFM_CB][tid % GSFM_CB] = L[chol_tile_off(T - 1, T - 1) + tid];
  __syncthreads();
  if (wave == 0) chol_back_block(Lk[(T - 1) & 1], y[(T - 1) * GSFM_CB + (lane & 31)], lane, xk[(T - 1) & 1], x, (T - 1) * GSFM_CB, n);
  if (T >= 2) { const uint32_t e = tid; Lk[T & 1][e / GSFM_CB][e % GSFM_CB] = L[chol_tile_off(T - 2, T - 2) + e]; }   // (T-2) & 1 == T & 1; all lanes incl. wave 0 after its solve
  __syncthreads();
  for (uint32_t k = T - 1; k >= 1; --k) {
    const double* xs = xk[k & 1];                     // x_k
    if (wave == 0) {
      // y_{k-1} -= L_{k,k-1}^T x_k (lanes 0..31 by column, lanes 32..63 mirror), then the substitution of block k - 1
      const double* t = L + chol_tile_off(k, k - 1) + (lane & 31);
      double s2 = 0.0;
#pragma unroll
      for (int r = 0; r < GSFM_CB; ++r) s2 += t[r * GSFM_CB] * xs[r];
      const double v = y[(k - 1) * GSFM_CB + (lane & 31)] - s2;
      chol_back_block(Lk[(k - 1) & 1], v, lane, xk[(k - 1) & 1], x, (k - 1) * GSFM_CB, n);
    } else {
      for (uint32_t idx = tid - 64; idx + GSFM_CB < k * GSFM_CB; idx += 960) {   // block rows j <= k - 2
        const uint32_t j = idx / GSFM_CB, c = idx % GSFM_CB;
        const double* t = L + chol_tile_off(k, j) + c;
        double s2 = 0.0;
#pragma unroll
        for (int r = 0; r < GSFM_CB; ++r) s2 += t[r * GSFM_CB] * xs[r];
        y[idx] -= s2;
      }
      if (k >= 2) for (uint32_t e = tid - 64; e < GSFM_TILE_ELEMS; e += 960) Lk[k & 1][e / GSFM_CB][e % GSFM_CB] = L[chol_tile_off(k - 2, k - 2) + e];   // diag tile k-2 replaces tile k
    }
    __syncthreads();
  }
}

}  // namespace gsfm

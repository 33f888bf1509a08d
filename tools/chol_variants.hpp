// Forms of the tiled Cholesky step that were measured against the product's (tools/bench_chol_batch.hip, profiles/r06b_chol_look.txt) and are
// not part of the library: the trailing update as a launch of its own behind k_chol_panel, and ONE block column per launch with the panel of
// column k + 1 beside the update with column k (the product runs two columns per launch: dense_kernels.hpp, k_chol_look2).
#pragma once
#include "../globalsfmpy_amd/csrc/dense_kernels.hpp"
namespace gsfm {
// The trailing update of step k in the FUSED step's arithmetic (round 6; the batched factorisation of several components, where the fused
// step's repeated eliminations -- two or more per trailing tile -- are what a launch costs once 3 000 tiles of six scenes share it): tile
// (i, j), k < j <= i <= T, becomes A_ij - L_ik L_jk^T exactly as chol_step_body computes it -- the product first, from zero, eight
// v_mfma_f64_16x16x4_f64 per 16 x 16 quadrant with the contraction index dealt as 4 kk + g, then own - product -- from the panel tiles
// k_chol_panel has written (the rows chol_step_body holds in Pi / Pj: the same instructions on the same inputs).  A factor built from
// k_chol_panel + this kernel is therefore bit-identical to one built by k_chol_step, step by step, and the host may choose per step and per
// LM iteration.  One workgroup per trailing tile: row t = i - k - 1 has t + 1 of them.
__device__ __forceinline__ void chol_update_exact_body(const CholArgs& a) {
  __shared__ double Pi[GSFM_CB][GSFM_CB + 1], Pj[GSFM_CB][GSFM_CB + 1];
  const uint32_t k = a.k, T = a.T, tid = threadIdx.x;
  uint32_t b = blockIdx.x, t = 0;
  while (b >= t + 1) { b -= t + 1; ++t; }
  const uint32_t i = k + 1 + t, j = k + 1 + b;
  if (i == T && j == T) return;    // the right-hand side has no diagonal tile
  const uint32_t ur = tid / 8, uc4 = (tid % 8) * 4, wave = tid >> 6, lane = tid & 63;
  const uint32_t mq = (16 * (wave >> 1) + (lane >> 4)) * GSFM_CB + 16 * (wave & 1) + (lane & 15);
  chol_d4 own;
  {
    const double* so = a.A + chol_tile_off(i, j) + mq;
#pragma unroll
    for (int q = 0; q < 4; ++q) own[q] = so[4 * q * GSFM_CB];
  }
  {
    const double2* li = (const double2*)(a.L + chol_tile_off(i, k) + ur * GSFM_CB + uc4);
    const double2 v0 = li[0], v1 = li[1];
    Pi[ur][uc4] = v0.x; Pi[ur][uc4 + 1] = v0.y; Pi[ur][uc4 + 2] = v1.x; Pi[ur][uc4 + 3] = v1.y;
    if (j != i) {
      const double2* lj = (const double2*)(a.L + chol_tile_off(j, k) + ur * GSFM_CB + uc4);
      const double2 w0 = lj[0], w1 = lj[1];
      Pj[ur][uc4] = w0.x; Pj[ur][uc4 + 1] = w0.y; Pj[ur][uc4 + 2] = w1.x; Pj[ur][uc4 + 3] = w1.y;
    }
  }
  __syncthreads();
  const uint32_t c = lane & 15, g = lane >> 4, ri = 16 * (wave >> 1) + c, rj = 16 * (wave & 1) + c;
  double (*Q)[GSFM_CB + 1] = (j == i) ? Pi : Pj;
  chol_d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int kk = 0; kk < 8; ++kk) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Pi[ri][4 * kk + g], Q[rj][4 * kk + g], acc, 0, 0, 0);
  double* d = a.A + chol_tile_off(i, j) + mq;
#pragma unroll
  for (int q = 0; q < 4; ++q) d[4 * q * GSFM_CB] = own[q] - acc[q];
}
__global__ void __launch_bounds__(256) k_chol_update_exact(CholArgs a) { chol_update_exact_body(a); }
// ONE launch per block column without the fused step's repeated eliminations (round 6): launch k APPLIES column k and PRODUCES column k + 1.
//   * panel workgroups, one per block row i = k + 1 .. T (row T: the right-hand side), first in the grid: the tiles (k + 1, k + 1) and
//     (i, k + 1) receive column k's update on the fly (left-looking: nobody else needs them any more), then wavefront 0 runs the 64-row
//     elimination on them -- L_{k+1,k+1} from workgroup 0, L_{i,k+1} from the others;
//   * update workgroups, one per trailing tile (i, j), k + 2 <= j <= i <= T: A_ij - L_ik L_jk^T as above.
// The two kinds touch disjoint tiles (the panel reads column k + 1 of A and writes column k + 1 of L, the update writes columns >= k + 2 of A
// and reads column k of L, which the PREVIOUS launch produced), so there is no dependency inside a launch, one elimination per block row
// instead of two to three per trailing tile, and the chain is still one launch per column: k_chol_panel for column 0, then launches
// 0 .. T - 2.  Every tile sees the same updates in the same order with the same instructions as under k_chol_step: the factor is
// bit-identical (tools/bench_chol_batch.hip checks every double).
// GSFM_LOOK_NT: trailing tiles per update workgroup, 1 or 3 (they share P_i).  Measured (tools/bench_chol_batch.hip, profiles/r06b_chol_look.txt):
// six matrices side by side 708 -> 682 us with three, Madrid's matrix alone 470 -> 477: three for the batched form, one for the single matrix.
#ifdef GSFM_LOOK_TIMING   // (tools/bench_chol_batch.hip -DGSFM_LOOK_TIMING: phase stamps of the first row workgroup of every launch)
__device__ unsigned long long gsfm_look_ts[128][8];
#define GSFM_LOOK_STAMP(n) do { if (blockIdx.x == 1 && blockIdx.y == 1 && threadIdx.x == 0) gsfm_look_ts[a.k][n] = wall_clock64(); } while (0)
#else
#define GSFM_LOOK_STAMP(n) do { } while (0)
#endif
template <int GSFM_LOOK_NT>
__device__ __forceinline__ void chol_look_body(const CholArgs& a) {
  static_assert(GSFM_LOOK_NT >= 1 && GSFM_LOOK_NT <= 3, "the update workgroup's P_j tiles share the panel's four LDS tiles");
  constexpr int GSFM_LOOK_LDS = 4;
  __shared__ double S[GSFM_LOOK_LDS][GSFM_CB][GSFM_CB + 1];   // panel workgroups: P_{k+1}, P_i and the updated tiles (k + 1, k + 1), (i, k + 1); update workgroups: P_i and up to three P_j
  // (the panel's four tiles in TWO -- the updated tiles in the place of their operands, one more barrier -- measured slower, 493 against 468 us for Madrid's matrix alone: profiles/r06b_chol_look.txt)
  const uint32_t k = a.k, T = a.T, c = k + 1, tid = threadIdx.x, n_panel = T - k;
  const uint32_t ur = tid / 8, uc4 = (tid % 8) * 4, wave = tid >> 6, lane = tid & 63;
  const uint32_t mq = (16 * (wave >> 1) + (lane >> 4)) * GSFM_CB + 16 * (wave & 1) + (lane & 15);
  const uint32_t qc = lane & 15, g = lane >> 4, ri = 16 * (wave >> 1) + qc, rj = 16 * (wave & 1) + qc;
  auto stage = [&](double (*P)[GSFM_CB + 1], const double* tile) {   // a 32 x 32 tile into LDS, 4 doubles per lane
    const double2* t = (const double2*)(tile + ur * GSFM_CB + uc4);
    const double2 v0 = t[0], v1 = t[1];
    P[ur][uc4] = v0.x; P[ur][uc4 + 1] = v0.y; P[ur][uc4 + 2] = v1.x; P[ur][uc4 + 3] = v1.y;
  };
  if (blockIdx.x < n_panel) {
    const uint32_t i = c + blockIdx.x;
    const bool diag = blockIdx.x == 0;             // block row k + 1 itself: L_{k+1,k+1}
    __builtin_amdgcn_s_setprio(3);                 // the chain runs through these wavefronts: ahead of the update workgroups that share the CU
    GSFM_LOOK_STAMP(0);
    double (*Pd)[GSFM_CB + 1] = S[0], (*Pi)[GSFM_CB + 1] = S[1], (*Dd)[GSFM_CB + 1] = S[2], (*Ci)[GSFM_CB + 1] = S[3];
    chol_d4 ownD, ownC = {0.0, 0.0, 0.0, 0.0};
    {
      const double* sd = a.A + chol_tile_off(c, c) + mq;
#pragma unroll
      for (int q = 0; q < 4; ++q) ownD[q] = sd[4 * q * GSFM_CB];
      if (!diag) {
        const double* sc = a.A + chol_tile_off(i, c) + mq;
#pragma unroll
        for (int q = 0; q < 4; ++q) ownC[q] = sc[4 * q * GSFM_CB];
      }
    }
    stage(Pd, a.L + chol_tile_off(c, k));
    if (!diag) stage(Pi, a.L + chol_tile_off(i, k));
    __syncthreads();
    GSFM_LOOK_STAMP(1);
    chol_d4 accD = {0.0, 0.0, 0.0, 0.0}, accC = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) accD = __builtin_amdgcn_mfma_f64_16x16x4f64(Pd[ri][4 * kk + g], Pd[rj][4 * kk + g], accD, 0, 0, 0);
    if (!diag) {
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) accC = __builtin_amdgcn_mfma_f64_16x16x4f64(Pi[ri][4 * kk + g], Pd[rj][4 * kk + g], accC, 0, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) Dd[16 * (wave >> 1) + g + 4 * q][16 * (wave & 1) + qc] = ownD[q] - accD[q];
    if (!diag) {
#pragma unroll
      for (int q = 0; q < 4; ++q) Ci[16 * (wave >> 1) + g + 4 * q][16 * (wave & 1) + qc] = ownC[q] - accC[q];
    }
    __syncthreads();
    GSFM_LOOK_STAMP(2);
    if (wave == 0) {
      const uint32_t rr = lane & 31;
      const double (*src)[GSFM_CB + 1] = (lane < 32 || diag) ? Dd : Ci;
      double r[GSFM_CB];
#pragma unroll
      for (int q = 0; q < GSFM_CB; ++q) r[q] = src[rr][q];
      GSFM_LOOK_STAMP(3);
      const int bad = chol_eliminate64(r, lane);
      GSFM_LOOK_STAMP(4);
      if (diag) {
        if (lane < 32) {
          double2* dl = (double2*)(a.L + chol_tile_off(c, c) + rr * GSFM_CB);
#pragma unroll
          for (int q = 0; q < GSFM_CB / 2; ++q) dl[q] = make_double2((uint32_t)(2 * q) <= lane ? r[2 * q] : 0.0, (uint32_t)(2 * q + 1) <= lane ? r[2 * q + 1] : 0.0);
          if (lane == 0 && bad && *a.info == 0) *a.info = (int)(c * GSFM_CB + bad);
        }
      } else if (lane >= 32) {
        double2* dl = (double2*)(a.L + chol_tile_off(i, c) + rr * GSFM_CB);
#pragma unroll
        for (int q = 0; q < GSFM_CB / 2; ++q) dl[q] = make_double2(r[2 * q], r[2 * q + 1]);
      }
      GSFM_LOOK_STAMP(5);
    }
    return;
  }
  // update workgroups: up to GSFM_LOOK_NT neighbouring tiles (i, j0 ..) of one block row (they share P_i; a third of the workgroups of the one-tile form)
  uint32_t b = blockIdx.x - n_panel, t = 0;
  while (b >= (t + GSFM_LOOK_NT) / GSFM_LOOK_NT) { b -= (t + GSFM_LOOK_NT) / GSFM_LOOK_NT; ++t; }
  const uint32_t i = k + 2 + t, j0 = k + 2 + GSFM_LOOK_NT * b;
  if (i > T) return;
  uint32_t nt = min((uint32_t)GSFM_LOOK_NT, i - j0 + 1);
  if (i == T && j0 + nt - 1 == T) --nt;    // the right-hand side has no diagonal tile
  if (nt == 0) return;
  chol_d4 own[GSFM_LOOK_NT];
#pragma unroll
  for (int u = 0; u < GSFM_LOOK_NT; ++u) if ((uint32_t)u < nt) {
    const double* so = a.A + chol_tile_off(i, j0 + u) + mq;
#pragma unroll
    for (int q = 0; q < 4; ++q) own[u][q] = so[4 * q * GSFM_CB];
  }
  stage(S[0], a.L + chol_tile_off(i, k));
#pragma unroll
  for (int u = 0; u < GSFM_LOOK_NT; ++u) if ((uint32_t)u < nt && j0 + u != i) stage(S[1 + u], a.L + chol_tile_off(j0 + u, k));
  __syncthreads();
  double aop[8];
#pragma unroll
  for (int kk = 0; kk < 8; ++kk) aop[kk] = S[0][ri][4 * kk + g];
#pragma unroll
  for (int u = 0; u < GSFM_LOOK_NT; ++u) if ((uint32_t)u < nt) {
    double (*Q)[GSFM_CB + 1] = (j0 + u == i) ? S[0] : S[1 + u];
    chol_d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(aop[kk], Q[rj][4 * kk + g], acc, 0, 0, 0);
    double* d = a.A + chol_tile_off(i, j0 + u) + mq;
#pragma unroll
    for (int q = 0; q < 4; ++q) d[4 * q * GSFM_CB] = own[u][q] - acc[q];
  }
}
// workgroups of launch k for a matrix of T block rows (k <= T - 2): T - k panel rows + the trailing tiles of the block rows k + 2 .. T
__host__ __device__ inline uint32_t chol_look_grid(uint32_t T, uint32_t k, uint32_t nt) { return (T - k) + chol_step_grid(T - k - 1, nt) - 1; }
template <int GSFM_LOOK_NT>
__global__ void __launch_bounds__(256) k_chol_look(CholArgs a) { chol_look_body<GSFM_LOOK_NT>(a); }
template <int GSFM_LOOK_NT>
__global__ void __launch_bounds__(256) k_chol_look_batch(const CholBatchItem* items, uint32_t k) {
  const CholBatchItem it = items[blockIdx.y];
  if (k + 2 > it.T || blockIdx.x >= chol_look_grid(it.T, k, GSFM_LOOK_NT) || !*it.active) return;
  const CholArgs a{it.A, it.L, it.T, k, it.info};
  chol_look_body<GSFM_LOOK_NT>(a);
}
// The batched step (k_chol_step_batch) as these two launches: blockIdx.y is the matrix, the grid is the LARGEST matrix's
__global__ void __launch_bounds__(64) k_chol_panel_batch(const CholBatchItem* items, uint32_t k) {
  const CholBatchItem it = items[blockIdx.y];
  if (k >= it.T || blockIdx.x > it.T - k || !*it.active) return;
  const CholArgs a{it.A, it.L, it.T, k, it.info};
  chol_panel_body(a);
}
__global__ void __launch_bounds__(256) k_chol_update_exact_batch(const CholBatchItem* items, uint32_t k) {
  const CholBatchItem it = items[blockIdx.y];
  if (k >= it.T) return;
  const uint32_t m = it.T - k;
  if (blockIdx.x >= m * (m + 1) / 2 || !*it.active) return;
  const CholArgs a{it.A, it.L, it.T, k, it.info};
  chol_update_exact_body(a);
}

}  // namespace gsfm

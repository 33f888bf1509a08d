// Forms of the tiled Cholesky step that were measured against the product's (tools/bench_chol_batch.hip, profiles/r06b_chol_look.txt) and are
// not part of the library: the trailing update as a launch of its own behind k_chol_panel, and ONE block column per launch with the panel of
// column k + 1 beside the update with column k (the product runs two columns per launch: dense_kernels.hpp, k_chol_look2).
#pragma once
#include "../globalsfmpy_amd/csrc/dense_kernels.hpp"
namespace gsfm {
// ---- the two-kernel schedule of matrices beyond 64 block columns until round 6 (tools/bench_chol_large.hip)
// Step k of the two-kernel schedule, first half: workgroup 0 (one wavefront) factors A_kk and writes L_kk; workgroup b >= 1 factors A_kk
// again in its lower lanes and, with the same instructions, turns the panel tile A_ik, i = k + b (block row T = the right-hand side), into
// L_ik = A_ik L_kk^-T in its upper lanes.  Reads A, writes L: no race with anything in this step.
__device__ __forceinline__ void chol_panel_body(const CholArgs& a) {
  const uint32_t k = a.k, lane = threadIdx.x, rr = lane & 31, b = blockIdx.x;
  const uint32_t i = k + b;   // b == 0: the diagonal tile itself in both halves
  const double2* src = (const double2*)(a.A + (lane < 32 ? chol_tile_off(k, k) : chol_tile_off(i, k)) + rr * GSFM_CB);
  double r[GSFM_CB];
#pragma unroll
  for (int q = 0; q < GSFM_CB / 2; ++q) { const double2 v = src[q]; r[2 * q] = v.x; r[2 * q + 1] = v.y; }
  const int bad = chol_eliminate64(r, lane);
  if (b == 0) {
    if (lane < 32) {
      double2* dl = (double2*)(a.L + chol_tile_off(k, k) + rr * GSFM_CB);
#pragma unroll
      for (int q = 0; q < GSFM_CB / 2; ++q) dl[q] = make_double2((uint32_t)(2 * q) <= lane ? r[2 * q] : 0.0, (uint32_t)(2 * q + 1) <= lane ? r[2 * q + 1] : 0.0);
      if (lane == 0 && bad && *a.info == 0) *a.info = (int)(k * GSFM_CB + bad);
    }
  } else if (lane >= 32) {
    double2* dl = (double2*)(a.L + chol_tile_off(i, k) + rr * GSFM_CB);
#pragma unroll
    for (int q = 0; q < GSFM_CB / 2; ++q) dl[q] = make_double2(r[2 * q], r[2 * q + 1]);
  }
}
__global__ void __launch_bounds__(64) k_chol_panel(CholArgs a) { chol_panel_body(a); }

// Second half: A_ij -= L_ik L_jk^T for every trailing tile k < j <= i <= T ((T, T) does not exist), one wavefront per tile, on the matrix
// cores: v_mfma_f64_16x16x4_f64 computes D(16x16) = A(16x4) B(4x16) + C; lane l supplies A[l & 15][l >> 4] and B[l >> 4][l & 15] and holds
// C/D[(l >> 4) + 4 reg][l & 15], reg = 0..3 (MI355X guide, fragment layout of the f64 form).  A 32 x 32 tile is 2 x 2 such blocks times
// 8 steps of K = 4; the A operand is -L_ik, the B operand L_jk read row-wise (= L_jk^T column-wise).
// Block columns k .. k + ncol - 1 of L (ncol = 1 or 2) are folded into the tiles (i, j), j0 <= j <= i <= T -- or, col_only, into the tiles
// (i, j0) of one block column alone.  Two columns per pass read and write every trailing tile once instead of twice (the update is bound by
// those 16 KB per tile from ~100 block rows on); the accumulation order per tile -- column k, then column k + 1 -- is the one two separate
// passes have, so the factor is bit-identical whichever way the host pairs the columns.
struct CholUpdArgs { double* A; const double* L; uint32_t T, k, j0, col_only; };
template <int NCOL>
__global__ void __launch_bounds__(256) k_chol_update_mfma(CholUpdArgs a) {
  const uint32_t T = a.T, lane = threadIdx.x & 63, j0 = a.j0;
  const uint64_t b = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const uint32_t m = T - j0 + 1;                          // block rows j0 .. T
  uint32_t i, j;
  if (a.col_only) {
    if (b >= m) return;
    i = j0 + (uint32_t)b; j = j0;
  } else {
    if (b >= (uint64_t)m * (m + 1) / 2) return;
    uint32_t t = (uint32_t)((sqrt(8.0 * (double)b + 1.0) - 1.0) * 0.5);
    while ((uint64_t)(t + 1) * (t + 2) / 2 <= b) ++t;
    while ((uint64_t)t * (t + 1) / 2 > b) --t;
    i = j0 + t; j = j0 + (uint32_t)(b - (uint64_t)t * (t + 1) / 2);
  }
  if (i == T && j == T) return;
  const uint32_t c = lane & 15, g = lane >> 4;
  double* Aij = a.A + chol_tile_off(i, j);
  chol_d4 acc[2][2];
#pragma unroll
  for (int si = 0; si < 2; ++si)
#pragma unroll
    for (int sj = 0; sj < 2; ++sj)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[si][sj][r] = Aij[(16 * si + g + 4 * r) * GSFM_CB + 16 * sj + c];
#pragma unroll
  for (uint32_t cc = 0; cc < (uint32_t)NCOL; ++cc) {
    const double* Li = a.L + chol_tile_off(i, a.k + cc);
    const double* Lj = a.L + chol_tile_off(j, a.k + cc);
    // The contraction index may be dealt to (MFMA step kk, lane group g) in any way, as long as A and B agree: lane group g takes
    // k = 8 g .. 8 g + 7, eight CONSECUTIVE doubles of a tile row, so the operand loads are 64 contiguous bytes per lane (whole rows per
    // 4 lanes) instead of eight 8-byte pieces 32 bytes apart.
    double aop[2][8], bop[2][8];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const double2* ra = (const double2*)(Li + (16 * s + c) * GSFM_CB + 8 * g);
      const double2* rb = (const double2*)(Lj + (16 * s + c) * GSFM_CB + 8 * g);
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        const double2 va = ra[h], vb = rb[h];
        aop[s][2 * h] = -va.x; aop[s][2 * h + 1] = -va.y; bop[s][2 * h] = vb.x; bop[s][2 * h + 1] = vb.y;
      }
    }
#pragma unroll
    for (int si = 0; si < 2; ++si)
#pragma unroll
      for (int sj = 0; sj < 2; ++sj)
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) acc[si][sj] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop[si][kk], bop[sj][kk], acc[si][sj], 0, 0, 0);
  }
#pragma unroll
  for (int si = 0; si < 2; ++si)
#pragma unroll
    for (int sj = 0; sj < 2; ++sj)
#pragma unroll
      for (int r = 0; r < 4; ++r) Aij[(16 * si + g + 4 * r) * GSFM_CB + 16 * sj + c] = acc[si][sj][r];
}

// ---- forms of the step of the small matrices measured in round 6 (tools/bench_chol_batch.hip)
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

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
// fp64 VALU throughout at these sizes.  Beyond ~50 block columns (more than 512 cameras) the redundancy of that schedule (every trailing
// workgroup repeating the diagonal factorisation) and its per-lane 1 x 4 update dominate, and the work is done by two kernels per step
// instead: k_chol_panel (one wavefront per panel tile: the same fused factorisation + substitution, once per tile ROW instead of once per
// trailing tile) and k_chol_update_mfma (one wavefront per trailing tile: A_ij -= L_ik L_jk^T as 2 x 2 x 8 v_mfma_f64_16x16x4_f64 -- the
// one dense contraction on this path, on the matrix cores).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gsfm {

#define GSFM_CB 32
#define GSFM_TILE_ELEMS (GSFM_CB * GSFM_CB)
#ifndef GSFM_CHOL_BATCH
#define GSFM_CHOL_BATCH 8
#endif
#define GSFM_DENSE_MAX_T 500   // block rows whose right-hand side the backward kernel keeps in LDS (125 KB of the 160): 3N <= 16000, 5333 cameras
#define GSFM_CHOL_SPLIT_T 64   // LDS capacity (block rows) of the single-workgroup backward kernel kept for A/B runs: the fused schedule is never used beyond
#define GSFM_CHOL_SPLIT_DEFAULT 64   // more block columns than this: the two-kernel MFMA schedule (see run_dense; crossover measured at ~68: 3N = 2048 1.06 vs 1.10 ms, 2304 1.30 vs 1.27)

__host__ __device__ inline size_t chol_tile_off(uint32_t i, uint32_t j) { return ((size_t)i * (i + 1) / 2 + j) * GSFM_TILE_ELEMS; }
__host__ __device__ inline size_t chol_num_tiles(uint32_t T) { return (size_t)(T + 1) * (T + 2) / 2; }   // block rows 0..T (row T = rhs)

// broadcast one lane's double through the scalar unit (v_readlane_b32 x 2): a few cycles, where a shuffle through the LDS crossbar
// (ds_bpermute) is ~120 cycles of latency in a dependent chain.  `lane` must be wave-uniform.
__device__ __forceinline__ double readlane_f64(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

typedef double chol_d4 __attribute__((ext_vector_type(4)));

struct CholArgs {
  double* A;      // tiles (i >= j), block rows 0..T; row T = right-hand side (first row of each tile)
  double* L;      // same layout: L_ik (i > k), L_kk, and in row T the forward-substituted y = L^-1 b
  uint32_t T;     // block rows of the matrix proper = ceil(3N / 32)
  uint32_t k;     // this step's block column
  int* info;      // 0, or 1 + index of the first non-positive pivot
};

// The 64-row elimination (lanes 0..31: rows of A_kk, lanes 32..63: rows of a panel tile): entries above the diagonal of the diagonal rows
// (lane < c) are never read by anyone, so they may hold anything; the multipliers L[c][c0] (held by lane c) are fetched in batches -- all
// v_readlane of a batch first, then the FMAs: one SGPR-hazard wait per batch instead of one per multiplier.
__device__ __forceinline__ int chol_eliminate64(double* r, uint32_t lane) {
  int bad = 0;
#pragma unroll
  for (int c0 = 0; c0 < GSFM_CB; ++c0) {
    double piv = readlane_f64(r[c0], c0);
    if (!(piv > 0.0)) { if (!bad) bad = c0 + 1; piv = 1.0; }
    const double inv = rsqrt(piv);
    r[c0] = (lane == (uint32_t)c0) ? piv * inv : r[c0] * inv;
#pragma unroll
    for (int cb = c0 + 1; cb < GSFM_CB; cb += GSFM_CHOL_BATCH) {
      double m[GSFM_CHOL_BATCH];
#pragma unroll
      for (int q = 0; q < GSFM_CHOL_BATCH; ++q) if (cb + q < GSFM_CB) m[q] = readlane_f64(r[c0], cb + q);
#pragma unroll
      for (int q = 0; q < GSFM_CHOL_BATCH; ++q) if (cb + q < GSFM_CB) r[cb + q] -= r[c0] * m[q];
    }
  }
  return bad;
}

// Step k.  Workgroup 0: L_kk and the right-hand side's panel tile; every other workgroup: up to NT = 1, 2 or 3 neighbouring trailing tiles
// (i, j0 .. j0 + NT - 1) of ONE tile row i = k + 1 + t (row t has t + 1 tiles, i.e. ceil((t + 1) / NT) workgroups).
// Wavefront 0 holds the rows of A_kk in lanes 0..31 and the rows of the panel tile A_ik in lanes 32..63, one row per lane in
// registers, and runs the right-looking elimination on all 64 rows at once: for the lower lanes that is the Cholesky factorisation,
// for the upper lanes the same instructions are the substitution P_i = A_ik L_kk^-T (the multipliers L[c][c0] are wave-uniform
// v_readlane broadcasts from the diagonal rows), so the panel costs nothing extra.  Wavefronts 1 .. NT do the same with A_jk of the workgroup's
// tiles.  NT per step: chol_step_tiles_per_wg -- with one tile per workgroup the wide early steps run two or three workgroups, i.e. four or
// six eliminations, per CU (16-17 us per step against 9-10 once a step fits one workgroup per CU); two or three tiles per workgroup need 3 / 2 or
// 4 / 3 eliminations per tile instead of 2.
__host__ __device__ inline uint32_t chol_step_grid(uint32_t m /* trailing block rows k+1 .. T */, uint32_t nt_per_wg) {
  uint32_t n = 1;
  for (uint32_t t = 0; t < m; ++t) n += (t + nt_per_wg) / nt_per_wg;
  return n;
}
// tiles per workgroup for a step with m trailing block rows: as few as keep the step at about one workgroup per CU (measured per step at 3N = 1182,
// profiles/r03_chol_steps.txt: the eliminations of a CU share more than its SIMDs -- two side by side run at full speed, four at ~0.7, six at ~0.6)
inline uint32_t chol_step_tiles_per_wg(uint32_t m) { const uint32_t tiles = m * (m + 1) / 2; return tiles <= 256 ? 1u : tiles <= 512 ? 2u : 3u; }
template <int GSFM_CHOL_NT>
__device__ __forceinline__ void chol_step_body(const CholArgs& a) {
  __shared__ double Pi[GSFM_CB][GSFM_CB + 1], Pj[GSFM_CHOL_NT][GSFM_CB][GSFM_CB + 1];
  const uint32_t k = a.k, T = a.T, tid = threadIdx.x;
  uint32_t i, j0 = k, nt = 1;   // nt tiles (i, j0 .. j0 + nt - 1)
  const bool diag_wg = blockIdx.x == 0;
  if (diag_wg) i = T;
  else {
    uint32_t b = blockIdx.x - 1, t = 0;
    while (b >= (t + GSFM_CHOL_NT) / GSFM_CHOL_NT) { b -= (t + GSFM_CHOL_NT) / GSFM_CHOL_NT; ++t; }
    i = k + 1 + t; j0 = k + 1 + GSFM_CHOL_NT * b;
    nt = min((uint32_t)GSFM_CHOL_NT, i - j0 + 1);
    if (i == T && j0 + nt - 1 == T) --nt;    // the right-hand side has no diagonal tile
    if (nt == 0) return;
  }
  const uint32_t ur = tid / 8, uc4 = (tid % 8) * 4;   // this lane's 1 x 4 piece of a tile (publishing L)
  const uint32_t wave = tid >> 6, lane = tid & 63;
  // trailing update on the matrix cores: wavefront w owns the 16 x 16 quadrant (w >> 1, w & 1) of each of the workgroup's tiles, in the C / D
  // layout of v_mfma_f64_16x16x4_f64 (lane l: column l & 15, rows (l >> 4) + 4 reg) -- the per-lane 1 x 4 VALU form cost 1.8 us of a 12 us step
  const uint32_t mq = (16 * (wave >> 1) + (lane >> 4)) * GSFM_CB + 16 * (wave & 1) + (lane & 15);   // element of reg 0; reg r: + 4 r rows
  chol_d4 own[GSFM_CHOL_NT];
  if (!diag_wg) {
#pragma unroll
    for (int u = 0; u < GSFM_CHOL_NT; ++u) if ((uint32_t)u < nt) {
      const double* so = a.A + chol_tile_off(i, j0 + u) + mq;
#pragma unroll
      for (int q = 0; q < 4; ++q) own[u][q] = so[4 * q * GSFM_CB];
    }
  }
  // wavefront 0: the panel tile of row i (or of the right-hand side); wavefront w >= 1: the panel tile of column tile j0 + w - 1, unless that
  // is row i itself (the diagonal tile of the trailing matrix uses P_i twice)
  const uint32_t jw = j0 + wave - 1;
  if (wave == 0 || (!diag_wg && wave - 1 < nt && jw != i)) {
    const uint32_t rr = lane & 31;
    const double2* src = (const double2*)(a.A + (lane < 32 ? chol_tile_off(k, k) : chol_tile_off(wave == 0 ? i : jw, k)) + rr * GSFM_CB);
    double r[GSFM_CB];
#pragma unroll
    for (int q = 0; q < GSFM_CB / 2; ++q) { const double2 v = src[q]; r[2 * q] = v.x; r[2 * q + 1] = v.y; }
    const int bad = chol_eliminate64(r, lane);
    if (lane >= 32) {
      double (*P)[GSFM_CB + 1] = wave == 0 ? Pi : Pj[wave - 1];
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
  if (j0 == k + 1 && i < T) {   // first trailing column: this workgroup publishes L_ik
    double* dl = a.L + chol_tile_off(i, k) + ur * GSFM_CB + uc4;
#pragma unroll
    for (int q = 0; q < 4; ++q) dl[q] = Pi[ur][uc4 + q];
  }
  {
    const uint32_t c = lane & 15, g = lane >> 4, ri = 16 * (wave >> 1) + c, rj = 16 * (wave & 1) + c;
    double aop[8];
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) aop[kk] = Pi[ri][4 * kk + g];
#pragma unroll
    for (int u = 0; u < GSFM_CHOL_NT; ++u) if ((uint32_t)u < nt) {
      double (*Q)[GSFM_CB + 1] = (j0 + u == i) ? Pi : Pj[u];
      chol_d4 acc = {0.0, 0.0, 0.0, 0.0};   // the product first, from zero, then own - product: the order of the per-lane form this replaced
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(aop[kk], Q[rj][4 * kk + g], acc, 0, 0, 0);
      double* d = a.A + chol_tile_off(i, j0 + u) + mq;
#pragma unroll
      for (int q = 0; q < 4; ++q) d[4 * q * GSFM_CB] = own[u][q] - acc[q];
    }
  }
}

template <int GSFM_CHOL_NT>
__global__ void __launch_bounds__(256) k_chol_step(CholArgs a) { chol_step_body<GSFM_CHOL_NT>(a); }

// ---- the same factorisation for SEVERAL matrices at once (round 5): the connected components of a disconnected view graph -- the normal matrix
// is block diagonal, the reference's Cholesky factorises block by block (estimator.cpp:299-305) -- each with its own tiles, size and status;
// blockIdx.y is the matrix, block column k of every matrix that still has one runs in the same launch (the chain is as long as the LARGEST
// component's, the launches are as wide as all of them together).
struct CholBatchItem {
  double* A; double* L; double* x;   // tiles (block rows 0..T), factor, solution (T * 32 doubles)
  uint32_t T, n;                     // block rows, unknowns (3 x cameras of the component)
  int* info;
  const int* active;                 // 0: the component's right-hand side is below the absolute floor of the step (comp_kernels.hpp, k_comp_activity):
                                     // its step is zero to 1e-14 rad, nothing of it is assembled or factorised in this LM step
};
template <int GSFM_CHOL_NT>
__global__ void __launch_bounds__(256) k_chol_step_batch(const CholBatchItem* items, uint32_t k) {
  const CholBatchItem it = items[blockIdx.y];
  if (k >= it.T || blockIdx.x >= chol_step_grid(it.T - k, GSFM_CHOL_NT) || !*it.active) return;
  const CholArgs a{it.A, it.L, it.T, k, it.info};
  chol_step_body<GSFM_CHOL_NT>(a);
}
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
    if (wave == 0) {
      const uint32_t rr = lane & 31;
      const double (*src)[GSFM_CB + 1] = (lane < 32 || diag) ? Dd : Ci;
      double r[GSFM_CB];
#pragma unroll
      for (int q = 0; q < GSFM_CB; ++q) r[q] = src[rr][q];
      const int bad = chol_eliminate64(r, lane);
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

// x_k = L_kk^-T y_k for one block: the 32-step substitution of one wavefront (running right-hand side in lane registers, solved components
// broadcast with v_readlane).
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
// Backward substitution in GROUPS of block rows (both schedules): `k_chol_back_group` -- one workgroup -- solves the block rows k1 - 1 .. k0
// bottom-up (wavefront 0: the 32-step substitution of block k, as chol_back_block; wavefronts 1..7: one tile (k, j) of the group each, loaded
// while wavefront 0 solves, folded into the group's right-hand sides once x_k is there), then `k_chol_back_update` -- one workgroup per block
// row ABOVE the group, all of them in parallel on their own CUs -- folds the group's x into the rest: y_j -= L_kj^T x_k, k descending.  Per
// tile the dot products run over r = 0 .. 31 in order and the subtractions over k descending, exactly as the single-workgroup kernel above
// did them: the same bits.  That kernel streamed all of L through one CU (135 us at 3N = 1182, 21 % of a Madrid LM iteration); here a group
// costs its 8 or 16 dependent block rows inside one launch and the bulk of L is read by many CUs at once (2 launches per group).
struct CholBackGroupArgs { double* L; double* x; uint32_t n, T, k0, k1; };   // x: T * 32 doubles (padded); y_j = first row of tile (T, j) of L
template <int GR>   // block rows per group = wavefronts of the workgroup (8 or 16)
__device__ __forceinline__ void chol_back_group_body(const CholBackGroupArgs& a) {
  constexpr uint32_t NT = 64 * GR, LOADERS = 32 * (GR - 1), PER = (GSFM_TILE_ELEMS + LOADERS - 1) / LOADERS;
  __shared__ double yg[GR][GSFM_CB];
  __shared__ double Lk[2][GSFM_CB][GSFM_CB + 1];
  __shared__ double xk[2][GSFM_CB];
  const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, c = lane & 31, k0 = a.k0, k1 = a.k1, G = k1 - k0;
  for (uint32_t idx = tid; idx < G * GSFM_CB; idx += NT) yg[idx / GSFM_CB][idx % GSFM_CB] = a.L[chol_tile_off(a.T, k0 + idx / GSFM_CB) + idx % GSFM_CB];
  for (uint32_t e = tid; e < GSFM_TILE_ELEMS; e += NT) Lk[(k1 - 1) & 1][e / GSFM_CB][e % GSFM_CB] = a.L[chol_tile_off(k1 - 1, k1 - 1) + e];
  __syncthreads();
  for (uint32_t k = k1; k-- > k0;) {
    // wavefronts 1 .. GR - 1, lower half: column c of tile (k, j), j = k0 + wave - 1 (requested before x_k exists); upper half: the next diagonal tile
    const uint32_t j = k0 + wave - 1;
    const bool tile = wave >= 1 && lane < GSFM_CB && j < k;
    double tv[GSFM_CB], dv[PER];
    if (tile) {
      const double* t = a.L + chol_tile_off(k, j) + c;
#pragma unroll
      for (int r = 0; r < GSFM_CB; ++r) tv[r] = t[r * GSFM_CB];
    }
    const bool diag = wave >= 1 && lane >= GSFM_CB && k > k0;
    const uint32_t e0 = (wave - 1) * GSFM_CB + c;     // LOADERS lanes, PER elements each
    if (diag) {
      const double* d = a.L + chol_tile_off(k - 1, k - 1);
#pragma unroll
      for (int i = 0; i < (int)PER; ++i) { const uint32_t e = e0 + LOADERS * i; dv[i] = e < GSFM_TILE_ELEMS ? d[e] : 0.0; }
    }
    if (wave == 0) chol_back_block(Lk[k & 1], yg[k - k0][c], lane, xk[k & 1], a.x, k * GSFM_CB, a.T * GSFM_CB);
    __syncthreads();
    if (tile) {
      double s2 = 0.0;
#pragma unroll
      for (int r = 0; r < GSFM_CB; ++r) s2 += tv[r] * xk[k & 1][r];
      yg[j - k0][c] -= s2;
    }
    if (diag) {
#pragma unroll
      for (int i = 0; i < (int)PER; ++i) { const uint32_t e = e0 + LOADERS * i; if (e < GSFM_TILE_ELEMS) Lk[(k - 1) & 1][e / GSFM_CB][e % GSFM_CB] = dv[i]; }
    }
    __syncthreads();
  }
}
template <int GR>
__global__ void __launch_bounds__(64 * GR) k_chol_back_group(CholBackGroupArgs a) { chol_back_group_body<GR>(a); }
template <int GR>
__device__ __forceinline__ void chol_back_update_body(const CholBackGroupArgs& a) {
  __shared__ double xs[GR][GSFM_CB], s2s[GR][GSFM_CB];
  const uint32_t tid = threadIdx.x, c = tid & 31, kk = tid >> 5, j = blockIdx.x, G = a.k1 - a.k0;   // kk: block row k0 + kk of the group
  double tv[GSFM_CB];
  if (kk < G) {
    const double* t = a.L + chol_tile_off(a.k0 + kk, j) + c;
#pragma unroll
    for (int r = 0; r < GSFM_CB; ++r) tv[r] = t[r * GSFM_CB];
    xs[kk][c] = a.x[(a.k0 + kk) * GSFM_CB + c];
  }
  __syncthreads();
  if (kk < G) {
    double s2 = 0.0;
#pragma unroll
    for (int r = 0; r < GSFM_CB; ++r) s2 += tv[r] * xs[kk][r];
    s2s[kk][c] = s2;
  }
  __syncthreads();
  if (tid < GSFM_CB) {
    double* yj = a.L + chol_tile_off(a.T, j);
    double v = yj[c];
    for (uint32_t q = G; q-- > 0;) v -= s2s[q][c];
    yj[c] = v;
  }
}
template <int GR>
__global__ void __launch_bounds__(32 * GR) k_chol_back_update(CholBackGroupArgs a) { chol_back_update_body<GR>(a); }
// batched: group g (counted from the bottom) of every matrix that has one; the update's workgroups beyond a matrix's k0 leave at once
template <int GR>
__global__ void __launch_bounds__(64 * GR) k_chol_back_group_batch(const CholBatchItem* items, uint32_t g) {
  const CholBatchItem it = items[blockIdx.y];
  if (g * GR >= it.T || !*it.active) return;
  const uint32_t k1 = it.T - g * GR, k0 = k1 > GR ? k1 - GR : 0;
  const CholBackGroupArgs a{it.L, it.x, it.n, it.T, k0, k1};
  chol_back_group_body<GR>(a);
}
template <int GR>
__global__ void __launch_bounds__(32 * GR) k_chol_back_update_batch(const CholBatchItem* items, uint32_t g) {
  const CholBatchItem it = items[blockIdx.y];
  if (g * GR >= it.T || !*it.active) return;
  const uint32_t k1 = it.T - g * GR, k0 = k1 > GR ? k1 - GR : 0;
  if (blockIdx.x >= k0) return;
  const CholBackGroupArgs a{it.L, it.x, it.n, it.T, k0, k1};
  chol_back_update_body<GR>(a);
}


}  // namespace gsfm

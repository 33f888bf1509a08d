// Exact LM step for small graphs: Cholesky of the dense damped normal matrix (3N x 3N) and the two triangular solves -- "normal
// equations + Cholesky", which is literally what the reference's SPARSE_NORMAL_CHOLESKY does
// (src/GSfM_nonlinear_rotation_estimator.cpp:72-74,176-179,299-302).
//
// The regime is launch-latency-bound, not flop-bound (3N <= ~4500: <= 30 GFLOP per factorisation), so the design minimises the
// length of the dependent chain instead of the arithmetic:
//   * the matrix lives as 32 x 32 tiles of the lower triangle (8 KiB, contiguous), plus one extra tile row holding the right-hand
//     side, so the forward substitution L y = b falls out of the factorisation (y is the last row of L);
//   * ONE launch per TWO block columns (round 6, k_chol_look2): the launch produces the columns c0 and c0 + 1 and applies the two before them.
//     Its panel workgroups -- one per block row -- give their own five tiles the pending updates on the fly (left-looking), then run the
//     64-row elimination (one wavefront, rows in registers, v_readlane broadcasts: the diagonal block's factorisation in the lower lanes IS
//     the substitution of the panel tile in the upper lanes) once per column; its update workgroups fold both pending columns into two
//     trailing tiles each on the matrix cores.  No dependency inside a launch, one elimination per block row and column.  Rounds 3-5 ran
//     ONE fused kernel per column in which every trailing tile's workgroup repeated the eliminations it needed (two to three per tile:
//     k_chol_step, kept behind GSFM_CHOL_FUSED=1 as the reference of the bit-identity test) -- the six scenes of C4 side by side 985 -> 660 us
//     per factorisation + solve, Madrid's matrix alone 519 -> 444, every double of L, y and x the same (tools/bench_chol_batch.hip);
//   * L goes to a second buffer, and what a launch reads of A and L nobody writes in that launch (the panel workgroups read the columns
//     c0, c0 + 1 of A and the pending columns of L, and write the columns c0, c0 + 1 of L; the update workgroups write the columns >= c0 + 2
//     of A): no race, no ordering between workgroups;
//   * the backward substitution L^T x = y runs in groups of 8 block rows: one workgroup solves a group (a wavefront per block row: the
//     32-step substitution with the diagonal tile's column in registers, the others folding x_k into the group's rows), one launch folds the
//     group's x into all block rows above it.
// The same schedule serves every size up to GSFM_DENSE_MAX_T block columns (round 6; until then matrices beyond 64 block columns ran a panel
// kernel and a separate trailing update per step: 3N = 2400 / 4500 / 9000 1.25 -> 0.89, 3.11 -> 2.25, 13.2 -> 11.8 ms per factorise + solve,
// tools/bench_chol_large.hip -- those kernels now live in tools/chol_variants.hpp).  The trailing updates are fp64 MFMA, everything else fp64 VALU.
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
#define GSFM_CHOL_FUSED_MAX_T 64   // the fused step of rounds 2-5 (GSFM_CHOL_FUSED=1: the reference of the bit-identity test) is never used beyond

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
// 1 / sqrt(x) with the instructions of the device library's rsqrt(double) -- v_rsq_f64, then y0 + (y0 e)(0.375 e + 0.5) with e = 1 - x y0^2 -- minus
// its closing special-case select (x = 0, inf, nan: never a pivot that is used).  The same bits for every positive finite x
// (tools/check_devmath.hip compares 2^26 of them; tools/bench_chol_batch.hip every double of the factors).
__device__ __forceinline__ double chol_rsqrt(double x) {
  const double y0 = __builtin_amdgcn_rsq(x);
  const double e = __builtin_fma(-x * y0, y0, 1.0);
  return __builtin_fma(y0 * e, __builtin_fma(e, 0.375, 0.5), y0);
}
#ifndef GSFM_ELIM_SHORT
#define GSFM_ELIM_SHORT 1     // 0: the loop of rounds 1-5 (tools/bench_chol_batch.hip -DGSFM_ELIM_SHORT=0 for the A/B)
#endif
__device__ __forceinline__ int chol_eliminate64(double* r, uint32_t lane) {
  (void)lane;
  // The elimination is what a launch of the factorisation costs (tools/bench_chol_batch.hip -DGSFM_LOOK_TIMING: 4.4 us of a 9.4 us launch until
  // round 6: ~2 100 instructions in one wavefront, 137 ns per pivot).  Round 6 takes three things out of every pivot, none of which changes
  // a bit of a factor that is used -- 3.65 us:
  //   * the pivot test's select: rsqrt runs on the pivot as it comes, the test sets the status beside it -- after a non-positive pivot the
  //     rows hold garbage either way, the factor is refused (info) and the step solved again by PCG;
  //   * the select that gave lane c0 "piv * inv" and the others "r * inv": in lane c0 r[c0] IS the pivot;
  //   * rsqrt's special-case select (chol_rsqrt).
  // Measured and not adopted (the same bits each, profiles/r06b_chol_look.txt): the multipliers broadcast through LDS instead of v_readlane
  // pairs -- half the instructions, but their latency lands on the chain: 4.95 us as written, 4.27 software-pipelined by one pivot,
  // against 4.38 --; the next pivot computed in every lane from two early broadcasts (no trip through the scalar unit between two
  // pivots, four instructions more per pivot): 4.00 against 3.65.
  int bad = 0;
#pragma unroll
  for (int c0 = 0; c0 < GSFM_CB; ++c0) {
#if GSFM_ELIM_SHORT
    const double piv = readlane_f64(r[c0], c0);
    if (!(piv > 0.0) && !bad) bad = c0 + 1;
    const double inv = chol_rsqrt(piv);
    r[c0] = r[c0] * inv;
#else
    double piv = readlane_f64(r[c0], c0);
    if (!(piv > 0.0)) { if (!bad) bad = c0 + 1; piv = 1.0; }
    const double inv = rsqrt(piv);
    r[c0] = (lane == (uint32_t)c0) ? piv * inv : r[c0] * inv;
#endif
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
// TWO block columns per launch (round 6, the form the product runs): launch c0 PRODUCES the columns c0 and c0 + 1 and APPLIES the two columns
// before them (the previous launch's).  Half the launches of k_chol_look for the same work per block row plus one redundant panel tile:
//   * panel workgroups -- the diagonal one, then one per block row i = c0 + 2 .. T -- hold the tiles (c0, c0), (c0 + 1, c0), (c0 + 1, c0 + 1),
//     (i, c0), (i, c0 + 1) in registers (16 x 16 quadrants per wavefront), give them the two pending columns' updates on the fly, run the
//     64-row elimination of column c0 for the rows c0 + 1 and i on two wavefronts at once, fold column c0 into (c0 + 1, c0 + 1) and
//     (i, c0 + 1) and eliminate column c0 + 1;
//   * update workgroups, one per two neighbouring trailing tiles (i, j), j >= c0 + 2: both pending columns in one pass, in ascending order.
// PROD = 1: a matrix's LAST column when it has an odd number of them (one column produced).  PEND = 0: the first launch (nothing pending, no
// update workgroups).  Per tile the updates arrive in ascending column order, each as "product from zero, then own - product" with the
// contraction index dealt as 4 kk + g, and the eliminations are chol_eliminate64 on the same rows: the factor is bit-identical to k_chol_step's
// and k_chol_look's (tools/bench_chol_batch.hip compares every double).
#ifdef GSFM_LOOK_TIMING   // (tools/bench_chol_batch.hip -DGSFM_LOOK_TIMING: phase stamps of the first row workgroup of matrix 1 in every launch)
__device__ unsigned long long gsfm_look2_ts[64][8];
#define GSFM_LOOK2_STAMP(n) do { if (blockIdx.x == 1 && blockIdx.y == 1 && threadIdx.x == 0) gsfm_look2_ts[a.k / 2][n] = wall_clock64(); } while (0)
#else
#define GSFM_LOOK2_STAMP(n) do { } while (0)
#endif
#define GSFM_LOOK2_NT 2       // trailing tiles per update workgroup: P_i and P_j of two columns for two tiles = the panel's six LDS tiles
#define GSFM_LOOK2_LDS 6
template <int PEND, int PROD>
__device__ __forceinline__ void chol_look2_body(const CholArgs& a, double (*S)[GSFM_CB][GSFM_CB + 1]) {
  static_assert((PEND == 0 || PEND == 2) && (PROD == 1 || PROD == 2), "pending columns: none or two; produced: one or two");
  const uint32_t c0 = a.k, T = a.T, tid = threadIdx.x;
  const uint32_t n_panel = T - c0 + 2 - PROD;     // the diagonal workgroup + the block rows c0 + PROD .. T
  const uint32_t ur = tid / 8, uc4 = (tid % 8) * 4, wave = tid >> 6, lane = tid & 63;
  const uint32_t qa = wave >> 1, qb = wave & 1, qc = lane & 15, g = lane >> 4, ri = 16 * qa + qc, rj = 16 * qb + qc;
  const uint32_t mq = (16 * qa + g) * GSFM_CB + 16 * qb + qc;
  auto own_load = [&](uint32_t i, uint32_t j) {   // this lane's four elements of its wavefront's quadrant of tile (i, j) of A
    chol_d4 v; const double* s = a.A + chol_tile_off(i, j) + mq;
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = s[4 * q * GSFM_CB];
    return v;
  };
  auto own_store = [&](uint32_t i, uint32_t j, chol_d4 v) {
    double* d = a.A + chol_tile_off(i, j) + mq;
#pragma unroll
    for (int q = 0; q < 4; ++q) d[4 * q * GSFM_CB] = v[q];
  };
  auto tile_regs = [&](uint32_t i, uint32_t j) {   // tile (i, j) of L, four consecutive doubles per lane
    const double2* t = (const double2*)(a.L + chol_tile_off(i, j) + ur * GSFM_CB + uc4);
    const double2 v0 = t[0], v1 = t[1];
    chol_d4 v = {v0.x, v0.y, v1.x, v1.y};
    return v;
  };
  auto tile_put = [&](double (*P)[GSFM_CB + 1], chol_d4 v) { P[ur][uc4] = v[0]; P[ur][uc4 + 1] = v[1]; P[ur][uc4 + 2] = v[2]; P[ur][uc4 + 3] = v[3]; };
  auto tile_publish = [&](uint32_t i, uint32_t j, const double (*P)[GSFM_CB + 1]) {
    double* d = a.L + chol_tile_off(i, j) + ur * GSFM_CB + uc4;
#pragma unroll
    for (int q = 0; q < 4; ++q) d[q] = P[ur][uc4 + q];
  };
  auto quad_put = [&](double (*P)[GSFM_CB + 1], chol_d4 v) {
#pragma unroll
    for (int q = 0; q < 4; ++q) P[16 * qa + g + 4 * q][16 * qb + qc] = v[q];
  };
  auto minus_prod = [&](chol_d4 o, const double (*X)[GSFM_CB + 1], const double (*Y)[GSFM_CB + 1]) {   // own - X Y^T: the product first, from zero
    chol_d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(X[ri][4 * kk + g], Y[rj][4 * kk + g], acc, 0, 0, 0);
    chol_d4 r;
#pragma unroll
    for (int q = 0; q < 4; ++q) r[q] = o[q] - acc[q];
    return r;
  };
  // one wavefront: the 64-row elimination of [D | X] (X == D for the diagonal workgroup's own factor); the upper lanes' rows go back into X,
  // the lower lanes' -- the Cholesky factor of D -- to tile (cd, cd) of L where asked
  auto eliminate = [&](double (*D)[GSFM_CB + 1], double (*X)[GSFM_CB + 1], bool write_diag, uint32_t cd) {
    const uint32_t rr = lane & 31;
    const double (*src)[GSFM_CB + 1] = lane < 32 ? D : X;
    double r[GSFM_CB];
#pragma unroll
    for (int q = 0; q < GSFM_CB; ++q) r[q] = src[rr][q];
    const int bad = chol_eliminate64(r, lane);
    if (lane >= 32) {
      if (X != D) {
#pragma unroll
        for (int q = 0; q < GSFM_CB; ++q) X[rr][q] = r[q];
      }
    } else if (write_diag) {
      double2* dl = (double2*)(a.L + chol_tile_off(cd, cd) + rr * GSFM_CB);
#pragma unroll
      for (int q = 0; q < GSFM_CB / 2; ++q) dl[q] = make_double2((uint32_t)(2 * q) <= lane ? r[2 * q] : 0.0, (uint32_t)(2 * q + 1) <= lane ? r[2 * q + 1] : 0.0);
      if (lane == 0 && bad && *a.info == 0) *a.info = (int)(cd * GSFM_CB + bad);
    }
  };
  if (blockIdx.x < n_panel) {
    __builtin_amdgcn_s_setprio(3);                 // the chain runs through these wavefronts: ahead of the update workgroups that share the CU
    const bool diag = blockIdx.x == 0;
    const uint32_t i = c0 + PROD - 1 + blockIdx.x;  // (not used by the diagonal workgroup)
    GSFM_LOOK2_STAMP(0);
    chol_d4 D00 = own_load(c0, c0), D10 = {0.0, 0.0, 0.0, 0.0}, D11 = D10, R0 = D10, R1 = D10;
    if (PROD == 2) { D10 = own_load(c0 + 1, c0); D11 = own_load(c0 + 1, c0 + 1); }
    if (!diag) { R0 = own_load(i, c0); if (PROD == 2) R1 = own_load(i, c0 + 1); }
    if (PEND == 2) {
      chol_d4 st[2][3];
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const uint32_t col = c0 - 2 + p;
        st[p][0] = tile_regs(c0, col);
        st[p][1] = PROD == 2 ? tile_regs(c0 + 1, col) : st[p][0];
        st[p][2] = !diag ? tile_regs(i, col) : st[p][0];
      }
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        if (p) __syncthreads();                    // (the first pending column's operands have been read)
        tile_put(S[0], st[p][0]);
        if (PROD == 2) tile_put(S[1], st[p][1]);
        if (!diag) tile_put(S[2], st[p][2]);
        __syncthreads();
        if (p == 0) GSFM_LOOK2_STAMP(7);
        D00 = minus_prod(D00, S[0], S[0]);
        if (PROD == 2) { D10 = minus_prod(D10, S[1], S[0]); D11 = minus_prod(D11, S[1], S[1]); }
        if (!diag) { R0 = minus_prod(R0, S[2], S[0]); if (PROD == 2) R1 = minus_prod(R1, S[2], S[1]); }
      }
    }
    GSFM_LOOK2_STAMP(1);
    quad_put(S[3], D00);
    if (PROD == 2) quad_put(S[4], D10);
    if (!diag) quad_put(S[5], R0);
    __syncthreads();
    GSFM_LOOK2_STAMP(2);
    if (PROD == 1) {                               // the last column of a matrix with an odd number of them: as k_chol_panel
      if (wave == 0) {
        const uint32_t rr = lane & 31;
        const double (*src)[GSFM_CB + 1] = (lane < 32 || diag) ? S[3] : S[5];
        double r[GSFM_CB];
#pragma unroll
        for (int q = 0; q < GSFM_CB; ++q) r[q] = src[rr][q];
        const int bad = chol_eliminate64(r, lane);
        if (diag) {
          if (lane < 32) {
            double2* dl = (double2*)(a.L + chol_tile_off(c0, c0) + rr * GSFM_CB);
#pragma unroll
            for (int q = 0; q < GSFM_CB / 2; ++q) dl[q] = make_double2((uint32_t)(2 * q) <= lane ? r[2 * q] : 0.0, (uint32_t)(2 * q + 1) <= lane ? r[2 * q + 1] : 0.0);
            if (lane == 0 && bad && *a.info == 0) *a.info = (int)(c0 * GSFM_CB + bad);
          }
        } else if (lane >= 32) {
          double2* dl = (double2*)(a.L + chol_tile_off(i, c0) + rr * GSFM_CB);
#pragma unroll
          for (int q = 0; q < GSFM_CB / 2; ++q) dl[q] = make_double2(r[2 * q], r[2 * q + 1]);
        }
      }
      return;
    }
    // column c0: the rows c0 + 1 (wavefront 0; the diagonal workgroup's lower lanes give L_{c0,c0}) and i (wavefront 1) side by side
    if (wave == 0) eliminate(S[3], S[4], diag, c0);
    else if (wave == 1 && !diag) eliminate(S[3], S[5], false, c0);
    GSFM_LOOK2_STAMP(3);
    __syncthreads();
    GSFM_LOOK2_STAMP(4);
    // column c0 into (c0 + 1, c0 + 1) and (i, c0 + 1), then column c0 + 1
    D11 = minus_prod(D11, S[4], S[4]);
    if (!diag) R1 = minus_prod(R1, S[5], S[4]);
    quad_put(S[0], D11);
    if (!diag) quad_put(S[1], R1);
    if (diag) tile_publish(c0 + 1, c0, S[4]); else tile_publish(i, c0, S[5]);
    __syncthreads();
    GSFM_LOOK2_STAMP(5);
    if (wave == 0) {
      if (diag) eliminate(S[0], S[0], true, c0 + 1);
      else {
        const uint32_t rr = lane & 31;
        const double (*src)[GSFM_CB + 1] = lane < 32 ? S[0] : S[1];
        double r[GSFM_CB];
#pragma unroll
        for (int q = 0; q < GSFM_CB; ++q) r[q] = src[rr][q];
        (void)chol_eliminate64(r, lane);
        if (lane >= 32) {
          double2* dl = (double2*)(a.L + chol_tile_off(i, c0 + 1) + rr * GSFM_CB);
#pragma unroll
          for (int q = 0; q < GSFM_CB / 2; ++q) dl[q] = make_double2(r[2 * q], r[2 * q + 1]);
        }
      }
      GSFM_LOOK2_STAMP(6);
    }
    return;
  }
  if (PEND == 0) return;
  // update workgroups: up to two neighbouring tiles (i, j0), (i, j0 + 1) of one block row, both pending columns
  uint32_t b = blockIdx.x - n_panel, t = 0;
  while (b >= (t + GSFM_LOOK2_NT) / GSFM_LOOK2_NT) { b -= (t + GSFM_LOOK2_NT) / GSFM_LOOK2_NT; ++t; }
  const uint32_t i = c0 + PROD + t, j0 = c0 + PROD + GSFM_LOOK2_NT * b;
  if (i > T) return;
  uint32_t nt = min((uint32_t)GSFM_LOOK2_NT, i - j0 + 1);
  if (i == T && j0 + nt - 1 == T) --nt;    // the right-hand side has no diagonal tile
  if (nt == 0) return;
  chol_d4 own[GSFM_LOOK2_NT];
#pragma unroll
  for (int u = 0; u < GSFM_LOOK2_NT; ++u) if ((uint32_t)u < nt) own[u] = own_load(i, j0 + u);
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    tile_put(S[p], tile_regs(i, c0 - 2 + p));
#pragma unroll
    for (int u = 0; u < GSFM_LOOK2_NT; ++u) if ((uint32_t)u < nt && j0 + u != i) tile_put(S[2 + 2 * u + p], tile_regs(j0 + u, c0 - 2 + p));
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < GSFM_LOOK2_NT; ++u) if ((uint32_t)u < nt) {
#pragma unroll
    for (int p = 0; p < 2; ++p) own[u] = minus_prod(own[u], S[p], (j0 + u == i) ? S[p] : S[2 + 2 * u + p]);
    own_store(i, j0 + u, own[u]);
  }
}
// workgroups of the launch that produces the columns c0 (, c0 + 1) of a matrix of T block rows
__host__ __device__ inline uint32_t chol_look2_grid(uint32_t T, uint32_t c0, bool pending) {
  const uint32_t prod = T - c0 >= 2 ? 2 : 1, n_panel = T - c0 + 2 - prod;
  return n_panel + (pending ? chol_step_grid(T - c0 - prod + 1, GSFM_LOOK2_NT) - 1 : 0);
}
template <int PEND>
__global__ void __launch_bounds__(256) k_chol_look2(CholArgs a) {
  __shared__ double S[GSFM_LOOK2_LDS][GSFM_CB][GSFM_CB + 1];
  if (a.T - a.k >= 2) chol_look2_body<PEND, 2>(a, S); else chol_look2_body<PEND, 1>(a, S);
}
template <int PEND>
__global__ void __launch_bounds__(256) k_chol_look2_batch(const CholBatchItem* items, uint32_t c0) {
  __shared__ double S[GSFM_LOOK2_LDS][GSFM_CB][GSFM_CB + 1];
  const CholBatchItem it = items[blockIdx.y];
  if (c0 >= it.T || blockIdx.x >= chol_look2_grid(it.T, c0, PEND != 0) || !*it.active) return;
  const CholArgs a{it.A, it.L, it.T, c0, it.info};
  if (it.T - c0 >= 2) chol_look2_body<PEND, 2>(a, S); else chol_look2_body<PEND, 1>(a, S);
}
// x_k = L_kk^-T y_k for one block: the 32-step substitution of one wavefront (running right-hand side in lane registers, solved components
// broadcast with v_readlane).
__device__ __forceinline__ void chol_back_block(const double (*Lk)[GSFM_CB + 1], double v, uint32_t lane, double* xk_out, double* x, uint32_t g0, uint32_t n) {
  const uint32_t l = lane & 31;
  const double rinv = 1.0 / Lk[l][l];   // all reciprocals at once, off the dependent chain
  // (round 6: this lane's column of L_kk in registers before the chain starts -- read inside it, every step waited for its LDS read: 1.64 us per
  // block row, of which the 32 dependent multiply / broadcast / FMA steps are a third; tools/bench_chol_batch.hip -DGSFM_BACK_TIMING)
  double col[GSFM_CB];
#pragma unroll
  for (int t = 0; t < GSFM_CB; ++t) col[t] = Lk[t][l];
#pragma unroll
  for (int t = GSFM_CB - 1; t >= 0; --t) {
    const double xt = readlane_f64(v * rinv, t);
    if (l < (uint32_t)t) v -= col[t] * xt;
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
#ifdef GSFM_BACK_TIMING   // (tools/bench_chol_batch.hip -DGSFM_BACK_TIMING: stamps of wavefront 0 of matrix 1's group kernel, per block row)
__device__ unsigned long long gsfm_back_ts[64][4];
#define GSFM_BACK_STAMP(row, n) do { if (blockIdx.y == 1 && threadIdx.x == 0) gsfm_back_ts[row][n] = wall_clock64(); } while (0)
#else
#define GSFM_BACK_STAMP(row, n) do { } while (0)
#endif
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
    GSFM_BACK_STAMP(k, 0);
    if (wave == 0) chol_back_block(Lk[k & 1], yg[k - k0][c], lane, xk[k & 1], a.x, k * GSFM_CB, a.T * GSFM_CB);
    GSFM_BACK_STAMP(k, 1);
    __syncthreads();
    GSFM_BACK_STAMP(k, 2);
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
    GSFM_BACK_STAMP(k, 3);
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

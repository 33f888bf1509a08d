// Exact LM step for small graphs: blocked right-looking Cholesky of the dense damped normal matrix (3N x 3N, row-major; L in the
// lower triangle, a transposed copy of its off-diagonal part in the upper) and the two triangular solves -- "normal equations + Cholesky", which is literally what the reference's
// SPARSE_NORMAL_CHOLESKY does (src/GSfM_nonlinear_rotation_estimator.cpp:72).  Three kernels per 32-column block:
// diagonal block (one workgroup, LDS), panel (one row per lane), trailing update (32 x 32 tiles); the whole sequence is
// captured once per problem into a hipGraph.  fp64 VALU throughout: 3N <= ~4000 means <= 20 GFLOP per factorisation
// and the launch chain, not the arithmetic, sets the time, so MFMA tiles would buy nothing here.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gsfm {

#define GSFM_CB 32
// broadcast one lane's double through the scalar unit (v_readlane_b32 x 2): a few cycles, where a shuffle through the LDS crossbar
// (ds_bpermute) is ~120 cycles of latency in a dependent chain.  `lane` must be wave-uniform.
__device__ __forceinline__ double readlane_f64(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}
struct CholArgs {
  double* A;      // n x n row-major; lower triangle in, L out
  uint32_t n;
  uint32_t k0;    // first row/column of the current block
  int* info;      // 0, or 1 + index of the first non-positive pivot
  double* Dinv;   // [n / 32 rounded up][32][32]: inverse of every diagonal block L_kk (lower triangular, row-major, zero above)
};

// One wavefront, no LDS, no barriers: lane i keeps row i of the block in registers; column j is finished with one broadcast of the
// pivot and one broadcast per remaining column (a 256-lane LDS version spent ~0.75 us per column in barriers and LDS latency).
__global__ void __launch_bounds__(64) k_chol_diag(CholArgs a) {
  const uint32_t nb = min((uint32_t)GSFM_CB, a.n - a.k0), lane = threadIdx.x;
  const bool rowlive = lane < nb;
  double r[GSFM_CB];
  {
    const double* src = a.A + (size_t)(a.k0 + (rowlive ? lane : 0)) * a.n + a.k0;
#pragma unroll
    for (int c = 0; c < GSFM_CB; ++c) r[c] = (rowlive && (uint32_t)c <= lane && (uint32_t)c < nb) ? src[c] : ((uint32_t)c == lane ? 1.0 : 0.0);
  }
  int bad = 0;
#pragma unroll
  for (int j = 0; j < GSFM_CB; ++j) {
    double piv = readlane_f64(r[j], j);
    if (!(piv > 0.0)) { if (!bad) bad = j + 1; piv = 1.0; }
    const double d = sqrt(piv);
    r[j] = (lane == (uint32_t)j) ? d : r[j] / d;
    // no per-step lane masks: entries above the diagonal (lane < c) are never read by anyone, so they may hold anything
#pragma unroll
    for (int c = j + 1; c < GSFM_CB; ++c) r[c] -= r[j] * readlane_f64(r[j], c);   // readlane: L[c][j], held by lane c
  }
  if (lane == 0 && bad && (uint32_t)bad <= nb && *a.info == 0) *a.info = (int)(a.k0 + bad);
  if (rowlive) {
    double* dst = a.A + (size_t)(a.k0 + lane) * a.n + a.k0;
#pragma unroll
    for (int c = 0; c < GSFM_CB; ++c) if ((uint32_t)c <= lane) dst[c] = r[c];
  }
  // X = L^-1 (lower triangular), row i in lane i: x[i][i] = 1 / L[i][i], x[i][j] = -(sum_{t=j+1..i} x[i][t] L[t][j]) / L[j][j].
  // The panel and the triangular solves then multiply by X instead of running 32-step substitution chains.
  double x[GSFM_CB];
#pragma unroll
  for (int j = GSFM_CB - 1; j >= 0; --j) {
    const double inv_jj = 1.0 / readlane_f64(r[j], j);
    double acc = 0.0;
#pragma unroll
    for (int t = j + 1; t < GSFM_CB; ++t) {
      acc += x[t] * readlane_f64(r[j], t);             // L[t][j] from lane t; x[t] is exactly 0 in the lanes above the diagonal
    }
    x[j] = (lane == (uint32_t)j) ? inv_jj : (lane > (uint32_t)j ? -acc * inv_jj : 0.0);
  }
  if (lane < GSFM_CB) {   // padded rows/columns of a partial last block carry the identity: harmless
    double* dst = a.Dinv + ((size_t)(a.k0 / GSFM_CB) * GSFM_CB + lane) * GSFM_CB;
#pragma unroll
    for (int c = 0; c < GSFM_CB; ++c) dst[c] = x[c];
  }
}

// rows below the diagonal block: A[i, block] <- A[i, block] L_kk^-T = A[i, block] X^T, a 32-wide product per element (no
// substitution chain); 64 rows per workgroup, 8 outputs per lane; the result also goes transposed into the upper triangle so
// that the forward substitution of k_chol_solve reads L[i][k0 + c] with consecutive lanes on consecutive i
#define GSFM_PANEL_ROWS 64
__global__ void __launch_bounds__(256) k_chol_panel(CholArgs a) {
  __shared__ double X[GSFM_CB][GSFM_CB + 1];
  __shared__ double V[GSFM_PANEL_ROWS][GSFM_CB + 1];
  const uint32_t nb = min((uint32_t)GSFM_CB, a.n - a.k0), tid = threadIdx.x;
  const double* xin = a.Dinv + (size_t)(a.k0 / GSFM_CB) * GSFM_CB * GSFM_CB;
  for (uint32_t idx = tid; idx < GSFM_CB * GSFM_CB; idx += 256) X[idx / GSFM_CB][idx % GSFM_CB] = xin[idx];
  const uint32_t row0 = a.k0 + nb + blockIdx.x * GSFM_PANEL_ROWS;
  for (uint32_t idx = tid; idx < GSFM_PANEL_ROWS * GSFM_CB; idx += 256) {
    const uint32_t r = idx / GSFM_CB, c = idx % GSFM_CB;
    V[r][c] = (row0 + r < a.n && c < nb) ? a.A[(size_t)(row0 + r) * a.n + a.k0 + c] : 0.0;
  }
  __syncthreads();
  const uint32_t c = tid % GSFM_CB, rg = tid / GSFM_CB;   // column c, rows rg, rg + 8, ...
  double out[GSFM_PANEL_ROWS / 8];
#pragma unroll
  for (int q = 0; q < GSFM_PANEL_ROWS / 8; ++q) {
    const uint32_t r = rg + 8 * q;
    double s2 = 0.0;
#pragma unroll
    for (int t = 0; t < GSFM_CB; ++t) s2 += V[r][t] * X[c][t];   // X[c][t] = 0 for t > c
    out[q] = s2;
  }
#pragma unroll
  for (int q = 0; q < GSFM_PANEL_ROWS / 8; ++q) {
    const uint32_t row = row0 + rg + 8 * q;
    if (row < a.n && c < nb) {
      a.A[(size_t)row * a.n + a.k0 + c] = out[q];
      a.A[(size_t)(a.k0 + c) * a.n + row] = out[q];
    }
  }
}

// trailing update A[i][j] -= sum_c P[i][c] P[j][c] for i >= j >= k0 + 32; one 32 x 32 tile of the lower triangle per workgroup
__global__ void __launch_bounds__(256) k_chol_update(CholArgs a) {
  __shared__ double Pi[GSFM_CB][GSFM_CB + 1], Pj[GSFM_CB][GSFM_CB + 1];
  // linear block index -> (ti, tj), ti >= tj
  const uint32_t b = blockIdx.x;
  uint32_t ti = (uint32_t)((sqrt(8.0 * (double)b + 1.0) - 1.0) * 0.5);
  while ((uint64_t)(ti + 1) * (ti + 2) / 2 <= b) ++ti;
  while ((uint64_t)ti * (ti + 1) / 2 > b) --ti;
  const uint32_t tj = b - ti * (ti + 1) / 2;
  const uint32_t base = a.k0 + GSFM_CB, i0 = base + GSFM_CB * ti, j0 = base + GSFM_CB * tj, tid = threadIdx.x;
  for (uint32_t idx = tid; idx < GSFM_CB * GSFM_CB; idx += 256) {
    const uint32_t r = idx / GSFM_CB, c = idx % GSFM_CB;
    Pi[r][c] = (i0 + r < a.n) ? a.A[(size_t)(i0 + r) * a.n + a.k0 + c] : 0.0;
    Pj[r][c] = (j0 + r < a.n) ? a.A[(size_t)(j0 + r) * a.n + a.k0 + c] : 0.0;
  }
  __syncthreads();
  const uint32_t tc = tid % GSFM_CB, tr = tid / GSFM_CB;   // 8 row groups x 32 columns
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const uint32_t r = tr + 8 * q, i = i0 + r, j = j0 + tc;
    if (i >= a.n || j >= a.n || j > i) continue;
    double s = 0.0;
#pragma unroll
    for (int c = 0; c < GSFM_CB; ++c) s += Pi[r][c] * Pj[tc][c];
    a.A[(size_t)i * a.n + j] -= s;
  }
}

// x = (L L^T)^-1 b, one workgroup.  Per 32-row block: the first wavefront multiplies by the inverse diagonal block (rows of X from
// LDS, the right-hand side broadcast with v_readlane), then all lanes update the remaining rows from the transposed panel copy.
__global__ void __launch_bounds__(1024) k_chol_solve(const double* __restrict__ A, const double* __restrict__ Dinv, uint32_t n,
                                                     const double* __restrict__ b, double* __restrict__ x) {
  __shared__ double yb[GSFM_CB];
  __shared__ double X[GSFM_CB][GSFM_CB + 1];
  const uint32_t tid = threadIdx.x;
  for (uint32_t i = tid; i < n; i += 1024) x[i] = b[i];
  __syncthreads();
  for (uint32_t k0 = 0; k0 < n; k0 += GSFM_CB) {   // L y = b
    const uint32_t nb = min((uint32_t)GSFM_CB, n - k0);
    X[tid / GSFM_CB][tid % GSFM_CB] = Dinv[(size_t)(k0 / GSFM_CB) * GSFM_CB * GSFM_CB + tid];
    __syncthreads();
    if (tid < 64) {
      const uint32_t lane = tid < GSFM_CB ? tid : 0;
      const double v = tid < nb ? x[k0 + tid] : 0.0;
      double y = 0.0;
#pragma unroll
      for (int t = 0; t < GSFM_CB; ++t) y += X[lane][t] * readlane_f64(v, t);   // y = X b_block
      if (tid < GSFM_CB) yb[tid] = tid < nb ? y : 0.0;
      if (tid < nb) x[k0 + tid] = y;
    }
    __syncthreads();
    for (uint32_t i = k0 + nb + tid; i < n; i += 1024) {   // (rows below exist only under full blocks: nb == 32 here)
      double s2 = x[i];
#pragma unroll
      for (int c = 0; c < GSFM_CB; ++c) s2 -= A[(size_t)(k0 + c) * n + i] * yb[c];   // L[i][k0 + c] from its transposed copy
      x[i] = s2;
    }
    __syncthreads();
  }
  const uint32_t nblk = (n + GSFM_CB - 1) / GSFM_CB;
  for (uint32_t kb = nblk; kb-- > 0;) {             // L^T x = y
    const uint32_t k0 = kb * GSFM_CB, nb = min((uint32_t)GSFM_CB, n - k0);
    X[tid / GSFM_CB][tid % GSFM_CB] = Dinv[(size_t)kb * GSFM_CB * GSFM_CB + tid];
    __syncthreads();
    if (tid < 64) {
      const uint32_t lane = tid < GSFM_CB ? tid : 0;
      const double v = tid < nb ? x[k0 + tid] : 0.0;
      double y = 0.0;
#pragma unroll
      for (int t = 0; t < GSFM_CB; ++t) y += X[t][lane] * readlane_f64(v, t);   // x_block = X^T y_block
      if (tid < GSFM_CB) yb[tid] = tid < nb ? y : 0.0;
      if (tid < nb) x[k0 + tid] = y;
    }
    __syncthreads();
    for (uint32_t i = tid; i < k0; i += 1024) {
      double s2 = x[i];
      if (nb == GSFM_CB) {
#pragma unroll
        for (int c = 0; c < GSFM_CB; ++c) s2 -= A[(size_t)(k0 + c) * n + i] * yb[c];
      } else {
        for (uint32_t c = 0; c < nb; ++c) s2 -= A[(size_t)(k0 + c) * n + i] * yb[c];
      }
      x[i] = s2;
    }
    __syncthreads();
  }
}

}  // namespace gsfm

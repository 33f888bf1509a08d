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
    if (lane == (uint32_t)j) r[j] = d;
    else if (lane > (uint32_t)j) r[j] = r[j] / d;
#pragma unroll
    for (int c = j + 1; c < GSFM_CB; ++c) {
      const double lcj = readlane_f64(r[j], c);       // L[c][j], held by lane c
      if (lane >= (uint32_t)c) r[c] -= r[j] * lcj;
    }
  }
  if (lane == 0 && bad && (uint32_t)bad <= nb && *a.info == 0) *a.info = (int)(a.k0 + bad);
  if (rowlive) {
    double* dst = a.A + (size_t)(a.k0 + lane) * a.n + a.k0;
#pragma unroll
    for (int c = 0; c < GSFM_CB; ++c) if ((uint32_t)c <= lane) dst[c] = r[c];
  }
}

// rows below the diagonal block: A[i, block] <- A[i, block] L_kk^-T, one row per lane (the row lives in LDS, column-padded)
#define GSFM_PANEL_ROWS 128
__global__ void __launch_bounds__(GSFM_PANEL_ROWS) k_chol_panel(CholArgs a) {
  __shared__ double L[GSFM_CB][GSFM_CB + 1];
  __shared__ double V[GSFM_PANEL_ROWS][GSFM_CB + 1];
  const uint32_t nb = min((uint32_t)GSFM_CB, a.n - a.k0), tid = threadIdx.x;
  for (uint32_t idx = tid; idx < GSFM_CB * GSFM_CB; idx += GSFM_PANEL_ROWS) {
    const uint32_t r = idx / GSFM_CB, c = idx % GSFM_CB;
    L[r][c] = (r < nb && c <= r) ? a.A[(size_t)(a.k0 + r) * a.n + a.k0 + c] : (r == c ? 1.0 : 0.0);
  }
  // coalesced load of the block's rows: consecutive lanes read consecutive columns of one row
  const uint32_t row0 = a.k0 + nb + blockIdx.x * GSFM_PANEL_ROWS;
  for (uint32_t idx = tid; idx < GSFM_PANEL_ROWS * GSFM_CB; idx += GSFM_PANEL_ROWS) {
    const uint32_t r = idx / GSFM_CB, c = idx % GSFM_CB;
    V[r][c] = (row0 + r < a.n && c < nb) ? a.A[(size_t)(row0 + r) * a.n + a.k0 + c] : 0.0;
  }
  __syncthreads();
  if (row0 + tid < a.n) {
    // L and V are padded (identity / zeros) to the full 32 columns, so both loops have compile-time bounds and the LDS reads
    // of one column are all in flight together instead of one ~100-cycle round trip per multiply
    double v[GSFM_CB];
#pragma unroll
    for (int c = 0; c < GSFM_CB; ++c) v[c] = V[tid][c];
#pragma unroll
    for (int c = 0; c < GSFM_CB; ++c) {
      double s = v[c];
#pragma unroll
      for (int t = 0; t < c; ++t) s -= v[t] * L[c][t];
      v[c] = s / L[c][c];
    }
#pragma unroll
    for (int c = 0; c < GSFM_CB; ++c) V[tid][c] = v[c];
    // the same numbers transposed into the (otherwise unused) upper triangle: the forward substitution of k_chol_solve then
    // reads L[i][k0 + c] with consecutive lanes on consecutive i, like the backward one
#pragma unroll
    for (int c = 0; c < GSFM_CB; ++c) if ((uint32_t)c < nb) a.A[(size_t)(a.k0 + c) * a.n + row0 + tid] = v[c];
  }
  __syncthreads();
  for (uint32_t idx = tid; idx < GSFM_PANEL_ROWS * GSFM_CB; idx += GSFM_PANEL_ROWS) {
    const uint32_t r = idx / GSFM_CB, c = idx % GSFM_CB;
    if (row0 + r < a.n && c < nb) a.A[(size_t)(row0 + r) * a.n + a.k0 + c] = V[r][c];
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

// x = (L L^T)^-1 b, one workgroup; every 32 x 32 diagonal system is staged in LDS and solved by the first wavefront with
// shuffles (reading the pivots from global memory inside the dependent chain cost ~1 us per column)
__global__ void __launch_bounds__(1024) k_chol_solve(const double* __restrict__ A, uint32_t n, const double* __restrict__ b, double* __restrict__ x) {
  __shared__ double yb[GSFM_CB];
  __shared__ double Ld[GSFM_CB][GSFM_CB + 1];
  const uint32_t tid = threadIdx.x;
  for (uint32_t i = tid; i < n; i += 1024) x[i] = b[i];
  __syncthreads();
  for (uint32_t k0 = 0; k0 < n; k0 += GSFM_CB) {   // L y = b
    const uint32_t nb = min((uint32_t)GSFM_CB, n - k0);
    { const uint32_t r = tid / GSFM_CB, c = tid % GSFM_CB; Ld[r][c] = (r < nb && c <= r) ? A[(size_t)(k0 + r) * n + k0 + c] : (r == c ? 1.0 : 0.0); }
    __syncthreads();
    if (tid < 64) {
      const uint32_t lane = tid < GSFM_CB ? tid : 0;
      double v = tid < nb ? x[k0 + tid] : 0.0;
      for (uint32_t t = 0; t < nb; ++t) {
        const double yt = readlane_f64(v, (int)t) / Ld[t][t];
        if (tid == t) v = yt;
        else if (tid > t && tid < nb) v -= Ld[lane][t] * yt;
      }
      if (tid < GSFM_CB) yb[tid] = tid < nb ? v : 0.0;
      if (tid < nb) x[k0 + tid] = v;
    }
    __syncthreads();
    for (uint32_t i = k0 + nb + tid; i < n; i += 1024) {   // (rows below exist only under full blocks: nb == 32 here)
      double s = x[i];
#pragma unroll
      for (int c = 0; c < GSFM_CB; ++c) s -= A[(size_t)(k0 + c) * n + i] * yb[c];   // L[i][k0 + c] from its transposed copy
      x[i] = s;
    }
    __syncthreads();
  }
  const uint32_t nblk = (n + GSFM_CB - 1) / GSFM_CB;
  for (uint32_t kb = nblk; kb-- > 0;) {             // L^T x = y
    const uint32_t k0 = kb * GSFM_CB, nb = min((uint32_t)GSFM_CB, n - k0);
    { const uint32_t r = tid / GSFM_CB, c = tid % GSFM_CB; Ld[r][c] = (r < nb && c <= r) ? A[(size_t)(k0 + r) * n + k0 + c] : (r == c ? 1.0 : 0.0); }
    __syncthreads();
    if (tid < 64) {
      const uint32_t lane = tid < GSFM_CB ? tid : 0;
      double v = tid < nb ? x[k0 + tid] : 0.0;
      for (uint32_t t = nb; t-- > 0;) {
        const double xt = readlane_f64(v, (int)t) / Ld[t][t];
        if (tid == t) v = xt;
        else if (tid < t) v -= Ld[t][lane] * xt;
      }
      if (tid < GSFM_CB) yb[tid] = tid < nb ? v : 0.0;
      if (tid < nb) x[k0 + tid] = v;
    }
    __syncthreads();
    for (uint32_t i = tid; i < k0; i += 1024) {
      double s = x[i];
      if (nb == GSFM_CB) {
#pragma unroll
        for (int c = 0; c < GSFM_CB; ++c) s -= A[(size_t)(k0 + c) * n + i] * yb[c];
      } else {
        for (uint32_t c = 0; c < nb; ++c) s -= A[(size_t)(k0 + c) * n + i] * yb[c];
      }
      x[i] = s;
    }
    __syncthreads();
  }
}

}  // namespace gsfm

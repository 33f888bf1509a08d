// The 64-row elimination of the exact Cholesky step (lanes 0..31: rows of A_kk, lanes 32..63: rows of a panel tile), two forms:
//   row-per-lane (csrc/dense_kernels.hpp::chol_eliminate64): 32 pivots, every remaining column updated through v_readlane multipliers;
//   blocked: the tile pair lives in the MFMA C layout, four columns at a time go through LDS into the row-per-lane form, are factored there,
//            and are applied to the remaining columns as one rank-4 v_mfma_f64_16x16x4_f64 per 16 x 16 block (same products, same order:
//            the same bits -- checked here).
// usage: bench_elim [workgroups]   (one eliminating wavefront per workgroup, like the first panel tile of k_chol_step)
// Result on MI355X (profiles/r03_bench_elim.txt): bit-identical, and 6.45 us against 5.84 us for the row-per-lane loop -- eight panels x four LDS
// round trips and the 32 rsqrt chains (which stay sequential) cost more than the ~1 700 v_readlane / FMA pairs the MFMAs replace.  Not adopted.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../globalsfmpy_amd/csrc/dense_kernels.hpp"
using namespace gsfm;

// ---- the blocked form (lives here only: measured slower than the row-per-lane loop, see the end of this file) ----
// The 64 x 32 tile pair in the C / D layout of v_mfma_f64_16x16x4_f64: v[rb][cb][reg] = element (16 rb + (lane >> 4) + 4 reg, 16 cb + (lane & 15)).
// Four columns at a time go through LDS (T) into the row-per-lane form, are factored there exactly as chol_eliminate64 factors them (6
// multiplier broadcasts per four pivots instead of ~64), go back through LDS, and are applied to all remaining columns as ONE rank-4 MFMA per
// 16 x 16 block: the matrix core adds its four products in k order with FMA rounding, which is the order and the rounding of the row-per-lane
// loop -- every entry anyone reads comes out with the same bits (tools/archive/bench_elim.hip checks that).
struct ElimC { chol_d4 v[4][2]; };
#define GSFM_WAVE_LDS_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)
__device__ __forceinline__ void elimc_load(ElimC& V, const double* __restrict__ diag, const double* __restrict__ panel, uint32_t lane) {
  const uint32_t g = lane >> 4, c = lane & 15;
#pragma unroll
  for (int rb = 0; rb < 4; ++rb)
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const uint32_t R = 16 * rb + g + 4 * r;
        V.v[rb][cb][r] = (rb < 2 ? diag + R * GSFM_CB : panel + (R - 32) * GSFM_CB)[16 * cb + c];
      }
}
__device__ __forceinline__ void elimc_store_rows(const ElimC& V, double* __restrict__ out /* 64 x 32, row-major */, uint32_t lane) {
  const uint32_t g = lane >> 4, c = lane & 15;
#pragma unroll
  for (int rb = 0; rb < 4; ++rb)
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int r = 0; r < 4; ++r) out[(16 * rb + g + 4 * r) * GSFM_CB + 16 * cb + c] = V.v[rb][cb][r];
}
__device__ __forceinline__ int chol_eliminate64_blocked(ElimC& V, double (*T)[5], uint32_t lane) {
  const uint32_t g = lane >> 4, c = lane & 15;
  int bad = 0;
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    constexpr int dummy = 0; (void)dummy;
    const int cbp = p / 4, c0 = (4 * p) % 16;
    const bool in_panel = (int)c >= c0 && (int)c < c0 + 4;
    // the four columns of the panel: C layout -> T[row][q] -> one row per lane
    if (in_panel) {
#pragma unroll
      for (int rb = 0; rb < 4; ++rb)
#pragma unroll
        for (int r = 0; r < 4; ++r) T[16 * rb + g + 4 * r][c - c0] = V.v[rb][cbp][r];
    }
    GSFM_WAVE_LDS_SYNC();
    double t[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) t[q] = T[lane][q];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int j = 4 * p + q;
      double piv = readlane_f64(t[q], j);
      if (!(piv > 0.0)) { if (!bad) bad = j + 1; piv = 1.0; }
      const double inv = rsqrt(piv);
      t[q] = (lane == (uint32_t)j) ? piv * inv : t[q] * inv;
#pragma unroll
      for (int q2 = q + 1; q2 < 4; ++q2) { const double m = readlane_f64(t[q], 4 * p + q2); t[q2] -= t[q] * m; }
    }
    GSFM_WAVE_LDS_SYNC();
#pragma unroll
    for (int q = 0; q < 4; ++q) T[lane][q] = t[q];
    GSFM_WAVE_LDS_SYNC();
    if (in_panel) {
#pragma unroll
      for (int rb = 0; rb < 4; ++rb)
#pragma unroll
        for (int r = 0; r < 4; ++r) V.v[rb][cbp][r] = T[16 * rb + g + 4 * r][c - c0];
    }
    if (p < 7) {   // rank-4 update of every column to the right of the panel
      double aop[4];
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) aop[rb] = -T[16 * rb + c][g];
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) {
        if (cb < cbp || (cb == cbp && c0 + 4 >= 16)) continue;
        double bop = T[16 * cb + c][g];
        if (cb == cbp && (int)c < c0 + 4) bop = 0.0;   // columns up to the panel's last one are finished
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) V.v[rb][cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop[rb], bop, V.v[rb][cb], 0, 0, 0);
      }
    }
    GSFM_WAVE_LDS_SYNC();
  }
  return bad;
}

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// in: tiles[2 * b] = A_kk, tiles[2 * b + 1] = A_xk (32 x 32 row-major); out: the 64 x 32 result, row-major
__global__ void __launch_bounds__(64) k_ref(const double* __restrict__ tiles, double* __restrict__ out, int reps) {
  const uint32_t lane = threadIdx.x, rr = lane & 31;
  const double* src = tiles + ((size_t)2 * blockIdx.x + (lane < 32 ? 0 : 1)) * 1024 + rr * 32;
  double r[32];
  for (int it = 0; it < reps; ++it) {
#pragma unroll
    for (int q = 0; q < 32; ++q) r[q] = src[q];
    (void)chol_eliminate64(r, lane);
  }
  double* dst = out + (size_t)blockIdx.x * 2048 + lane * 32;
#pragma unroll
  for (int q = 0; q < 32; ++q) dst[q] = r[q];
}

__global__ void __launch_bounds__(64) k_blocked(const double* __restrict__ tiles, double* __restrict__ out, int reps) {
  __shared__ double T[64][5];
  const uint32_t lane = threadIdx.x;
  const double* a0 = tiles + (size_t)2 * blockIdx.x * 1024;
  ElimC V;
  for (int it = 0; it < reps; ++it) {
    elimc_load(V, a0, a0 + 1024, lane);
    (void)chol_eliminate64_blocked(V, T, lane);
  }
  elimc_store_rows(V, out + (size_t)blockIdx.x * 2048, lane);
}

int main(int argc, char** argv) {
  const int nwg = argc > 1 ? atoi(argv[1]) : 256;
  std::vector<double> h((size_t)nwg * 2048);
  srand(5);
  for (int b = 0; b < nwg; ++b) {
    // SPD diagonal tile: B B^T / 8 + I ; panel tile: random
    double B[32][8];
    for (auto& row : B) for (double& v : row) v = (double)rand() / RAND_MAX - 0.5;
    for (int r = 0; r < 32; ++r) for (int c = 0; c < 32; ++c) { double s = r == c ? 1.0 : 0.0; for (int t = 0; t < 8; ++t) s += B[r][t] * B[c][t] / 8.0; h[(size_t)b * 2048 + r * 32 + c] = s; }
    for (int e = 0; e < 1024; ++e) h[(size_t)b * 2048 + 1024 + e] = (double)rand() / RAND_MAX - 0.5;
  }
  double *d_in, *d_a, *d_b;
  CHK(hipMalloc(&d_in, 8 * h.size())); CHK(hipMalloc(&d_a, 8 * h.size())); CHK(hipMalloc(&d_b, 8 * h.size()));
  CHK(hipMemcpy(d_in, h.data(), 8 * h.size(), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_ref, dim3(nwg), dim3(64), 0, 0, d_in, d_a, 1);
  hipLaunchKernelGGL(k_blocked, dim3(nwg), dim3(64), 0, 0, d_in, d_b, 1);
  CHK(hipDeviceSynchronize());
  std::vector<double> ra(h.size()), rb(h.size());
  CHK(hipMemcpy(ra.data(), d_a, 8 * h.size(), hipMemcpyDeviceToHost)); CHK(hipMemcpy(rb.data(), d_b, 8 * h.size(), hipMemcpyDeviceToHost));
  // compare what anyone reads: diagonal rows' lower triangle, all of the panel rows
  size_t ndiff = 0; double maxd = 0;
  for (int b = 0; b < nwg; ++b) for (int r = 0; r < 64; ++r) for (int c = 0; c < 32; ++c) {
    if (r < 32 && c > r) continue;
    const size_t o = (size_t)b * 2048 + r * 32 + c;
    if (memcmp(&ra[o], &rb[o], 8) != 0) { ++ndiff; maxd = fmax(maxd, fabs(ra[o] - rb[o])); }
  }
  printf("%d tile pairs: %zu of the read entries differ in their bits (max |difference| %.3e)\n", nwg, ndiff, maxd);
  hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  const int reps = 200;
  for (int which = 0; which < 2; ++which) {
    for (int w = 0; w < 2; ++w) {
      CHK(hipEventRecord(e0, 0));
      if (which == 0) hipLaunchKernelGGL(k_ref, dim3(nwg), dim3(64), 0, 0, d_in, d_a, reps);
      else hipLaunchKernelGGL(k_blocked, dim3(nwg), dim3(64), 0, 0, d_in, d_b, reps);
      CHK(hipEventRecord(e1, 0)); CHK(hipEventSynchronize(e1));
    }
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-14s %d workgroups: %.2f us per elimination (load + eliminate, %d back to back in one wavefront)\n", which ? "blocked" : "row-per-lane", nwg, 1e3 * ms / reps, reps);
  }
  return 0;
}

// Dev check + timing of the blocked Cholesky kernels (globalsfmpy_amd/csrc/dense_kernels.hpp) on a random SPD matrix.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "../globalsfmpy_amd/csrc/dense_kernels.hpp"
using namespace gsfm;
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
int main(int argc, char** argv) {
  for (uint32_t n : {7u, 32u, 33u, 100u, 231u, 1182u, 3090u}) {
    std::mt19937_64 rng(n);
    std::normal_distribution<double> nd;
    std::vector<double> B((size_t)n * 8), A((size_t)n * n, 0.0), b(n), x(n);
    for (auto& v : B) v = nd(rng);
    for (uint32_t i = 0; i < n; ++i) for (uint32_t j = 0; j < n; ++j) { double s = 0; for (int k = 0; k < 8; ++k) s += B[i * 8 + k] * B[j * 8 + k]; A[(size_t)i * n + j] = s + (i == j ? 5.0 : 0.0); }
    for (auto& v : b) v = nd(rng);
    double *dA, *db, *dx, *dDinv; int* dinfo;
    CHK(hipMalloc(&dA, 8 * (size_t)n * n)); CHK(hipMalloc(&db, 8 * n)); CHK(hipMalloc(&dx, 8 * n)); CHK(hipMalloc(&dinfo, 4)); CHK(hipMalloc(&dDinv, 8 * (size_t)((n + GSFM_CB - 1) / GSFM_CB) * GSFM_CB * GSFM_CB));
    CHK(hipMemcpy(db, b.data(), 8 * n, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    float best = 1e9, best_solve = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
      CHK(hipMemcpy(dA, A.data(), 8 * (size_t)n * n, hipMemcpyHostToDevice)); CHK(hipMemset(dinfo, 0, 4));
      CHK(hipEventRecord(e0));
      for (uint32_t k0 = 0; k0 < n; k0 += GSFM_CB) {
        CholArgs c{dA, n, k0, dinfo, dDinv};
        hipLaunchKernelGGL(k_chol_diag, dim3(1), dim3(64), 0, 0, c);
        if (k0 + GSFM_CB >= n) break;
        const uint32_t below = n - k0 - GSFM_CB, tiles = (below + GSFM_CB - 1) / GSFM_CB;
        hipLaunchKernelGGL(k_chol_panel, dim3((below + GSFM_PANEL_ROWS - 1) / GSFM_PANEL_ROWS), dim3(256), 0, 0, c);
        hipLaunchKernelGGL(k_chol_update, dim3(tiles * (tiles + 1) / 2), dim3(256), 0, 0, c);
      }
      CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
      float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms);
      CHK(hipEventRecord(e0));
      hipLaunchKernelGGL(k_chol_solve, dim3(1), dim3(1024), 0, 0, (const double*)dA, (const double*)dDinv, n, (const double*)db, dx);
      CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
      CHK(hipEventElapsedTime(&ms, e0, e1)); best_solve = std::min(best_solve, ms);
    }
    int info; CHK(hipMemcpy(&info, dinfo, 4, hipMemcpyDeviceToHost)); CHK(hipMemcpy(x.data(), dx, 8 * n, hipMemcpyDeviceToHost));
    double rmax = 0, bmax = 0;
    for (uint32_t i = 0; i < n; ++i) { double s = -b[i]; for (uint32_t j = 0; j < n; ++j) s += A[(size_t)i * n + j] * x[j]; rmax = std::max(rmax, std::fabs(s)); bmax = std::max(bmax, std::fabs(b[i])); }
    if (n <= 100) {  // compare the factor with a host Cholesky
      std::vector<double> L(A), Ld((size_t)n * n);
      for (uint32_t i = 0; i < n; ++i) for (uint32_t j = 0; j <= i; ++j) { double s2 = L[(size_t)i * n + j]; for (uint32_t k = 0; k < j; ++k) s2 -= L[(size_t)i * n + k] * L[(size_t)j * n + k]; L[(size_t)i * n + j] = (i == j) ? std::sqrt(s2) : s2 / L[(size_t)j * n + j]; }
      CHK(hipMemcpy(Ld.data(), dA, 8 * (size_t)n * n, hipMemcpyDeviceToHost));
      int shown = 0;
      for (uint32_t i = 0; i < n && shown < 6; ++i) for (uint32_t j = 0; j <= i && shown < 6; ++j)
        if (std::fabs(Ld[(size_t)i * n + j] - L[(size_t)i * n + j]) > 1e-9) { printf("   L[%u][%u]: device %.6f host %.6f\n", i, j, Ld[(size_t)i * n + j], L[(size_t)i * n + j]); ++shown; }
    }
    printf("n = %5u: info %d, |Ax - b|_inf / |b|_inf = %.2e, factor %.3f ms, solve %.3f ms\n", n, info, rmax / bmax, best, best_solve);
    hipFree(dA); hipFree(db); hipFree(dx); hipFree(dinfo); hipFree(dDinv);
  }
  return 0;
}

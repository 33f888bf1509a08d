// Dev check + timing of the tiled Cholesky kernels (globalsfmpy_amd/csrc/dense_kernels.hpp) on a random SPD matrix.
// hipcc --offload-arch=gfx950 -O3 -o bench_chol bench_chol.hip ; ./bench_chol [n ...]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <random>
#include <algorithm>
#include "../globalsfmpy_amd/csrc/dense_kernels.hpp"
using namespace gsfm;
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

int main(int argc, char** argv) {
  std::vector<uint32_t> sizes;
  for (int k = 1; k < argc; ++k) sizes.push_back(atoi(argv[k]));
  if (sizes.empty()) sizes = {96, 1182, 2400, 4500, 9000};
  for (uint32_t n : sizes) for (int split = 0; split < 2; ++split) {
    if (!split && n > 5000) continue;   // (the one-kernel-per-step schedule needs minutes there)
    const uint32_t T = (n + GSFM_CB - 1) / GSFM_CB;
    const size_t elems = chol_num_tiles(T) * GSFM_TILE_ELEMS;
    std::mt19937_64 rng(n);
    std::normal_distribution<double> N01(0, 1);
    // A = B B^T / n + I with B n x 8 (cheap SPD with full coupling), dense symmetric
    std::vector<double> B((size_t)n * 8), A((size_t)n * n), b(n), hA(elems, 0.0);
    for (auto& v : B) v = N01(rng);
    for (auto& v : b) v = N01(rng);
    for (uint32_t r = 0; r < n; ++r) for (uint32_t c = 0; c <= r; ++c) {
      double s = (r == c) ? 1.0 : 0.0;
      for (int t = 0; t < 8; ++t) s += B[(size_t)r * 8 + t] * B[(size_t)c * 8 + t] / 8.0;
      A[(size_t)r * n + c] = A[(size_t)c * n + r] = s;
      hA[chol_tile_off(r / 32, c / 32) + (r % 32) * 32 + c % 32] = s;
    }
    for (uint32_t g = n; g < T * 32; ++g) hA[chol_tile_off(g / 32, g / 32) + (g % 32) * 33] = 1.0;
    for (uint32_t g = 0; g < n; ++g) hA[chol_tile_off(T, g / 32) + g % 32] = b[g];
    double *dA0, *dA, *dL, *dx; int* dinfo;
    CHK(hipMalloc(&dA0, 8 * elems)); CHK(hipMalloc(&dA, 8 * elems)); CHK(hipMalloc(&dL, 8 * elems)); CHK(hipMalloc(&dx, 8 * (size_t)T * 32)); CHK(hipMalloc(&dinfo, 4));
    CHK(hipMemcpy(dA0, hA.data(), 8 * elems, hipMemcpyHostToDevice)); CHK(hipMemset(dL, 0, 8 * elems)); CHK(hipMemset(dinfo, 0, 4));
    hipStream_t st; CHK(hipStreamCreate(&st));
    auto enqueue = [&]() {
      CHK(hipMemcpyAsync(dA, dA0, 8 * elems, hipMemcpyDeviceToDevice, st));
      auto update = [&](uint32_t k, uint32_t ncol, uint32_t j0, bool col_only) {
        CholUpdArgs u{dA, dL, T, k, j0, col_only ? 1u : 0u};
        const uint64_t m = T - j0 + 1, tiles = col_only ? m : m * (m + 1) / 2;
        if (j0 > T || !tiles) return;
        const dim3 grid((uint32_t)((tiles + 3) / 4)), blk(256);
        if (ncol == 4) hipLaunchKernelGGL(k_chol_update_mfma<4>, grid, blk, 0, st, u);
        else if (ncol == 3) hipLaunchKernelGGL(k_chol_update_mfma<3>, grid, blk, 0, st, u);
        else if (ncol == 2) hipLaunchKernelGGL(k_chol_update_mfma<2>, grid, blk, 0, st, u);
        else hipLaunchKernelGGL(k_chol_update_mfma<1>, grid, blk, 0, st, u);
      };
      if (!split) for (uint32_t k = 0; k < T; ++k) {
        CholArgs c{dA, dL, T, k, dinfo};
        const uint64_t m = T - k;
        { const uint32_t nt = getenv("GSFM_CHOL_NT") ? (uint32_t)std::max(1, std::min(3, atoi(getenv("GSFM_CHOL_NT")))) : chol_step_tiles_per_wg((uint32_t)m); const dim3 grid(chol_step_grid((uint32_t)m, nt));
          if (nt == 3) hipLaunchKernelGGL(k_chol_step<3>, grid, dim3(256), 0, st, c); else if (nt == 2) hipLaunchKernelGGL(k_chol_step<2>, grid, dim3(256), 0, st, c); else hipLaunchKernelGGL(k_chol_step<1>, grid, dim3(256), 0, st, c); }
      } else if (getenv("CHOL_SINGLE_COLUMN_UPDATES")) for (uint32_t k = 0; k < T; ++k) {   // (the schedule before the columns were paired)
        CholArgs c{dA, dL, T, k, dinfo};
        hipLaunchKernelGGL(k_chol_panel, dim3(T - k + 1), dim3(64), 0, st, c);
        update(k, 1, k + 1, false);
      } else {   // the product's schedule (run_dense): groups of 2 block columns, 4 beyond 192 (CHOL_GROUP=1..4 overrides)
        const uint32_t GROUP = getenv("CHOL_GROUP") ? (uint32_t)std::max(1, std::min(4, atoi(getenv("CHOL_GROUP")))) : (T > 192 ? 4u : 2u);
        for (uint32_t k = 0; k < T; k += GROUP) {
          const uint32_t g = std::min(GROUP, T - k);
          for (uint32_t c = 0; c < g; ++c) {
            CholArgs pc{dA, dL, T, k + c, dinfo};
            hipLaunchKernelGGL(k_chol_panel, dim3(T - (k + c) + 1), dim3(64), 0, st, pc);
            if (c + 1 < g) update(k, c + 1, k + c + 1, true);
          }
          update(k, g, k + g, false);
        }
      }
      if (getenv("CHOL_BACK_OLD")) {   // (the backward forms before the grouped one)
        if (!split) hipLaunchKernelGGL(k_chol_back<GSFM_DENSE_MAX_T>, dim3(1), dim3(1024), 0, st, (const double*)dL, n, T, dx);
        else for (uint32_t k = T; k >= 1; --k) { CholBackArgs b{dL, dx, n, T, k}; hipLaunchKernelGGL(k_chol_back_step, dim3(k == T ? 1 : k), dim3(64), 0, st, b); }
      } else for (uint32_t k1 = T; k1 > 0;) {   // the product's backward substitution (run_dense): groups of 8 block rows
        const uint32_t k0 = k1 > 8 ? k1 - 8 : 0;
        CholBackGroupArgs b{dL, dx, n, T, k0, k1};
        hipLaunchKernelGGL(k_chol_back_group<8>, dim3(1), dim3(512), 0, st, b);
        if (k0) hipLaunchKernelGGL(k_chol_back_update<8>, dim3(k0), dim3(256), 0, st, b);
        k1 = k0;
      }
    };
    hipGraph_t g; hipGraphExec_t ge;
    CHK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal)); enqueue(); CHK(hipStreamEndCapture(st, &g)); CHK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CHK(hipGraphLaunch(ge, st)); CHK(hipStreamSynchronize(st));
    std::vector<double> x((size_t)T * 32);
    int info; CHK(hipMemcpy(&info, dinfo, 4, hipMemcpyDeviceToHost)); CHK(hipMemcpy(x.data(), dx, 8 * (size_t)n, hipMemcpyDeviceToHost));
    double rmax = 0, bmax = 0;
    for (uint32_t r = 0; r < n; ++r) { double s = -b[r]; for (uint32_t c = 0; c < n; ++c) s += A[(size_t)r * n + c] * x[c]; rmax = fmax(rmax, fabs(s)); bmax = fmax(bmax, fabs(b[r])); }
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const int reps = 20;
    CHK(hipEventRecord(e0, st)); for (int k = 0; k < reps; ++k) CHK(hipGraphLaunch(ge, st)); CHK(hipEventRecord(e1, st)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    // the backward kernel alone
    CHK(hipEventRecord(e0, st)); for (int k = 0; k < reps; ++k) hipLaunchKernelGGL(k_chol_back<GSFM_CHOL_SPLIT_T>, dim3(1), dim3(1024), 0, st, (const double*)dL, n, std::min<uint32_t>(T, GSFM_CHOL_SPLIT_T), dx); CHK(hipEventRecord(e1, st)); CHK(hipEventSynchronize(e1));
    float msb; CHK(hipEventElapsedTime(&msb, e0, e1));
    printf("n = %5u (T = %3u) %-28s: info %d  max |Ax - b| / max|b| = %.2e   factor+solve %.3f ms (graph replay), single-workgroup backward of min(T, 48) block rows alone %.3f ms\n", n, T,
           split ? "panel + MFMA update per step" : "one kernel per step", info, rmax / bmax, ms / reps, msb / reps);
    CHK(hipFree(dA0)); CHK(hipFree(dA)); CHK(hipFree(dL)); CHK(hipFree(dx)); CHK(hipFree(dinfo));
  }
  return 0;
}

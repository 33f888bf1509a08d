// One LARGE matrix (3N = 2400 / 4500 / 9000: graphs of 800 - 3000 cameras, where lm_solve switches to exact steps once PCG has turned out dearer):
// the product's schedule beyond 64 block columns until round 6 -- k_chol_panel + k_chol_update_mfma in groups of two / four columns -- against
// k_chol_look2 (two columns per launch, the schedule of the small matrices).  The two differ in the LAST BITS of the factor (the update's
// accumulation order), so the check is the residual |Ax - b|, not a comparison of doubles.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -o bench_chol_large bench_chol_large.hip ; ./bench_chol_large [n ...]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <random>
#include <algorithm>
#include "chol_variants.hpp"
using namespace gsfm;
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

int main(int argc, char** argv) {
  std::vector<uint32_t> sizes;
  for (int k = 1; k < argc; ++k) sizes.push_back(atoi(argv[k]));
  if (sizes.empty()) sizes = {1182, 2046, 2400, 4500, 9000};
  for (uint32_t n : sizes) {
    const uint32_t T = (n + GSFM_CB - 1) / GSFM_CB;
    const size_t elems = chol_num_tiles(T) * GSFM_TILE_ELEMS;
    std::mt19937_64 rng(n);
    std::normal_distribution<double> N01(0, 1);
    std::vector<double> B((size_t)n * 8), b(n), hA(elems, 0.0);
    for (auto& v : B) v = N01(rng);
    for (auto& v : b) v = N01(rng);
    auto a_at = [&](uint32_t r, uint32_t c) { double s = (r == c) ? 1.0 : 0.0; for (int t = 0; t < 8; ++t) s += B[(size_t)r * 8 + t] * B[(size_t)c * 8 + t] / 8.0; return s; };
    for (uint32_t r = 0; r < n; ++r) for (uint32_t c = 0; c <= r; ++c) hA[chol_tile_off(r / 32, c / 32) + (r % 32) * 32 + c % 32] = a_at(r, c);
    for (uint32_t g = n; g < T * 32; ++g) hA[chol_tile_off(g / 32, g / 32) + (g % 32) * 33] = 1.0;
    for (uint32_t g = 0; g < n; ++g) hA[chol_tile_off(T, g / 32) + g % 32] = b[g];
    double *dA0, *dA, *dL, *dx; int* dinfo;
    CHK(hipMalloc(&dA0, 8 * elems)); CHK(hipMalloc(&dA, 8 * elems)); CHK(hipMalloc(&dL, 8 * elems)); CHK(hipMalloc(&dx, 8 * (size_t)T * 32)); CHK(hipMalloc(&dinfo, 4));
    CHK(hipMemcpy(dA0, hA.data(), 8 * elems, hipMemcpyHostToDevice)); CHK(hipMemset(dL, 0, 8 * elems)); CHK(hipMemset(dinfo, 0, 4));
    hipStream_t st; CHK(hipStreamCreate(&st));
    for (int form = 0; form < 2; ++form) {
      auto enqueue = [&]() {
        CHK(hipMemcpyAsync(dA, dA0, 8 * elems, hipMemcpyDeviceToDevice, st));
        if (form == 0) {
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
          const uint32_t GROUP = T > 192 ? 4u : 2u;
          for (uint32_t k = 0; k < T; k += GROUP) {
            const uint32_t g = std::min(GROUP, T - k);
            for (uint32_t c = 0; c < g; ++c) {
              CholArgs pc{dA, dL, T, k + c, dinfo};
              hipLaunchKernelGGL(k_chol_panel, dim3(T - (k + c) + 1), dim3(64), 0, st, pc);
              if (c + 1 < g) update(k, c + 1, k + c + 1, true);
            }
            update(k, g, k + g, false);
          }
        } else {
          CholArgs c{dA, dL, T, 0, dinfo};
          hipLaunchKernelGGL(k_chol_look2<0>, dim3(chol_look2_grid(T, 0, false)), dim3(256), 0, st, c);
          for (c.k = 2; c.k < T; c.k += 2) hipLaunchKernelGGL(k_chol_look2<2>, dim3(chol_look2_grid(T, c.k, true)), dim3(256), 0, st, c);
        }
        for (uint32_t k1 = T; k1 > 0;) {
          const uint32_t k0 = k1 > 8 ? k1 - 8 : 0;
          CholBackGroupArgs bg{dL, dx, n, T, k0, k1};
          hipLaunchKernelGGL(k_chol_back_group<8>, dim3(1), dim3(512), 0, st, bg);
          if (k0) hipLaunchKernelGGL(k_chol_back_update<8>, dim3(k0), dim3(256), 0, st, bg);
          k1 = k0;
        }
      };
      hipGraph_t g; hipGraphExec_t ge;
      CHK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal)); enqueue(); CHK(hipStreamEndCapture(st, &g)); CHK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      CHK(hipGraphLaunch(ge, st)); CHK(hipStreamSynchronize(st));
      std::vector<double> x((size_t)T * 32);
      int info; CHK(hipMemcpy(&info, dinfo, 4, hipMemcpyDeviceToHost)); CHK(hipMemcpy(x.data(), dx, 8 * (size_t)n, hipMemcpyDeviceToHost));
      double rmax = 0, bmax = 0;
      for (uint32_t r = 0; r < n; r += 5) { double s = -b[r]; for (uint32_t c = 0; c < n; ++c) s += a_at(r, c) * x[c]; rmax = fmax(rmax, fabs(s)); bmax = fmax(bmax, fabs(b[r])); }
      hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
      const int reps = 10;
      CHK(hipEventRecord(e0, st)); for (int k = 0; k < reps; ++k) CHK(hipGraphLaunch(ge, st)); CHK(hipEventRecord(e1, st)); CHK(hipEventSynchronize(e1));
      float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
      printf("n = %5u (T = %3u) %-44s: info %d  max |Ax - b| / max|b| (sampled) = %.2e   factor + solve %.3f ms (graph replay)\n", n, T,
             form == 0 ? "k_chol_panel + k_chol_update_mfma (groups)" : "k_chol_look2 (two columns per launch)", info, rmax / bmax, ms / reps);
      CHK(hipGraphExecDestroy(ge)); CHK(hipGraphDestroy(g));
    }
    CHK(hipFree(dA0)); CHK(hipFree(dA)); CHK(hipFree(dL)); CHK(hipFree(dx)); CHK(hipFree(dinfo));
  }
  return 0;
}

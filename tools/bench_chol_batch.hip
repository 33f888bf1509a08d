// Dev check + timing of the BATCHED tiled Cholesky (globalsfmpy_amd/csrc/dense_kernels.hpp, the per-component exact step of a disconnected view
// graph): the fused step kernel against the two-launch form of the wide steps (k_chol_panel_batch + k_chol_update_exact_batch), which must give
// the same bits.  Matrices: random SPD of the six factorised scenes of C4 (3 x 227 / 394 / 450 / 332 / 328 / 437 unknowns).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -o bench_chol_batch bench_chol_batch.hip ; ./bench_chol_batch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <random>
#include <algorithm>
#include "chol_variants.hpp"
using namespace gsfm;
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

int main() {
  const std::vector<uint32_t> cams = {227, 394, 450, 332, 328, 437};
  const uint32_t NI = (uint32_t)cams.size();
  std::vector<CholBatchItem> items(NI);
  std::vector<size_t> offA(NI), offL(NI), offX(NI);
  size_t words = 0; uint32_t Tmax = 0;
  for (uint32_t c = 0; c < NI; ++c) { const uint32_t n = 3 * cams[c], T = (n + 31) / 32; items[c].T = T; items[c].n = n; offA[c] = words; words += chol_num_tiles(T) * 1024; Tmax = std::max(Tmax, T); }
  const size_t a_words = words;
  for (uint32_t c = 0; c < NI; ++c) { offL[c] = words; words += chol_num_tiles(items[c].T) * 1024; }
  for (uint32_t c = 0; c < NI; ++c) { offX[c] = words; words += (size_t)items[c].T * 32; }
  std::vector<double> h(words, 0.0);
  for (uint32_t c = 0; c < NI; ++c) {
    const uint32_t n = items[c].n, T = items[c].T;
    std::mt19937_64 rng(1000 + c); std::normal_distribution<double> N01(0, 1);
    std::vector<double> B((size_t)n * 8);
    for (auto& v : B) v = N01(rng);
    double* A = h.data() + offA[c];
    for (uint32_t r = 0; r < n; ++r) for (uint32_t q = 0; q <= r; ++q) {
      double s = (r == q) ? 1.0 : 0.0;
      for (int t = 0; t < 8; ++t) s += B[(size_t)r * 8 + t] * B[(size_t)q * 8 + t] / 8.0;
      A[chol_tile_off(r / 32, q / 32) + (r % 32) * 32 + q % 32] = s;
    }
    for (uint32_t g = n; g < T * 32; ++g) A[chol_tile_off(g / 32, g / 32) + (g % 32) * 33] = 1.0;
    for (uint32_t g = 0; g < n; ++g) A[chol_tile_off(T, g / 32) + g % 32] = N01(rng);
  }
  double *slab0, *slab; int *dinfo, *dactive; CholBatchItem* ditems;
  CHK(hipMalloc(&slab0, 8 * a_words)); CHK(hipMalloc(&slab, 8 * words)); CHK(hipMalloc(&dinfo, 4 * NI)); CHK(hipMalloc(&dactive, 4 * NI)); CHK(hipMalloc(&ditems, sizeof(CholBatchItem) * NI));
  CHK(hipMemcpy(slab0, h.data(), 8 * a_words, hipMemcpyHostToDevice));
  for (uint32_t c = 0; c < NI; ++c) { items[c].A = slab + offA[c]; items[c].L = slab + offL[c]; items[c].x = slab + offX[c]; items[c].info = dinfo + c; items[c].active = dactive + c; }
  CHK(hipMemcpy(ditems, items.data(), sizeof(CholBatchItem) * NI, hipMemcpyHostToDevice));
  hipStream_t st; CHK(hipStreamCreate(&st));
  const uint32_t NEVER = 0xffffffffu, LOOK = 0xfffffffeu, LOOK2 = 0xfffffffdu;
  // split_above_tiles: steps whose trailing tile count (of the LARGEST matrix) is above this run as panel + update
  const int look_nt = getenv("LOOK_NT") ? atoi(getenv("LOOK_NT")) : 3;   // trailing tiles per update workgroup of the one-launch-per-column form
  auto enqueue = [&](uint32_t split_above_tiles) {
    CHK(hipMemcpyAsync(slab, slab0, 8 * a_words, hipMemcpyDeviceToDevice, st));
    CHK(hipMemsetAsync(slab + a_words, 0, 8 * (words - a_words), st));
    CHK(hipMemsetAsync(dinfo, 0, 4 * NI, st));
    if (split_above_tiles == LOOK) {   // the product from round 6 on: the panel of column 0, then one launch per column (applies column k, produces column k + 1)
      hipLaunchKernelGGL(k_chol_panel_batch, dim3(Tmax + 1, NI), dim3(64), 0, st, (const CholBatchItem*)ditems, 0u);
      for (uint32_t k = 0; k + 2 <= Tmax; ++k) {
        if (look_nt == 3) hipLaunchKernelGGL(k_chol_look_batch<3>, dim3(chol_look_grid(Tmax, k, 3), NI), dim3(256), 0, st, (const CholBatchItem*)ditems, k);
        else hipLaunchKernelGGL(k_chol_look_batch<1>, dim3(chol_look_grid(Tmax, k, 1), NI), dim3(256), 0, st, (const CholBatchItem*)ditems, k);
      }
    } else if (split_above_tiles == LOOK2) {   // two columns per launch
      hipLaunchKernelGGL(k_chol_look2_batch<0>, dim3(chol_look2_grid(Tmax, 0, false), NI), dim3(256), 0, st, (const CholBatchItem*)ditems, 0u);
      for (uint32_t c0 = 2; c0 < Tmax; c0 += 2) hipLaunchKernelGGL(k_chol_look2_batch<2>, dim3(chol_look2_grid(Tmax, c0, true), NI), dim3(256), 0, st, (const CholBatchItem*)ditems, c0);
    } else for (uint32_t k = 0; k < Tmax; ++k) {
      const uint32_t m = Tmax - k, tiles = m * (m + 1) / 2;
      if (tiles > split_above_tiles) {
        hipLaunchKernelGGL(k_chol_panel_batch, dim3(m + 1, NI), dim3(64), 0, st, (const CholBatchItem*)ditems, k);
        hipLaunchKernelGGL(k_chol_update_exact_batch, dim3(tiles, NI), dim3(256), 0, st, (const CholBatchItem*)ditems, k);
      } else {
        const uint32_t nt = chol_step_tiles_per_wg(m);
        const dim3 grid(chol_step_grid(m, nt), NI);
        if (nt == 3) hipLaunchKernelGGL(k_chol_step_batch<3>, grid, dim3(256), 0, st, (const CholBatchItem*)ditems, k);
        else if (nt == 2) hipLaunchKernelGGL(k_chol_step_batch<2>, grid, dim3(256), 0, st, (const CholBatchItem*)ditems, k);
        else hipLaunchKernelGGL(k_chol_step_batch<1>, grid, dim3(256), 0, st, (const CholBatchItem*)ditems, k);
      }
    }
#ifndef BACK_GR
#define BACK_GR 8
#endif
    constexpr uint32_t GR = BACK_GR;   // (-DBACK_GR=16: groups of 16 block rows, the A/B of rounds 3 and 6)
    for (uint32_t g = 0; g * GR < Tmax; ++g) {
      hipLaunchKernelGGL(k_chol_back_group_batch<GR>, dim3(1, NI), dim3(64 * GR), 0, st, (const CholBatchItem*)ditems, g);
      const uint32_t k1 = Tmax - g * GR, k0 = k1 > GR ? k1 - GR : 0;
      if (k0) hipLaunchKernelGGL(k_chol_back_update_batch<GR>, dim3(k0, NI), dim3(32 * GR), 0, st, (const CholBatchItem*)ditems, g);
    }
  };
  const std::vector<std::pair<const char*, uint32_t>> forms = {{"fused step everywhere", NEVER}, {"panel + update where > 256 trailing tiles", 256}, {"panel + update everywhere", 0}, {"one launch per column: panel k + 1 beside update k", LOOK}, {"one launch per TWO columns", LOOK2}};
  const std::vector<std::vector<int>> live = {{1, 1, 1, 1, 1, 1}, {0, 1, 1, 0, 0, 1}, {0, 1, 0, 0, 0, 0}};
  std::vector<double> ref;
  for (const auto& lv : live) {
    CHK(hipMemcpy(dactive, lv.data(), 4 * NI, hipMemcpyHostToDevice));
    printf("live matrices:"); for (uint32_t c = 0; c < NI; ++c) if (lv[c]) printf(" %u", 3 * cams[c]); printf("\n");
    for (size_t f = 0; f < forms.size(); ++f) {
      hipGraph_t g; hipGraphExec_t ge;
      CHK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal)); enqueue(forms[f].second); CHK(hipStreamEndCapture(st, &g)); CHK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      CHK(hipGraphLaunch(ge, st)); CHK(hipStreamSynchronize(st));
      std::vector<double> out(words - a_words); std::vector<int> info(NI);
      CHK(hipMemcpy(out.data(), slab + a_words, 8 * (words - a_words), hipMemcpyDeviceToHost)); CHK(hipMemcpy(info.data(), dinfo, 4 * NI, hipMemcpyDeviceToHost));
      size_t diff = 0;
      unsigned long long fnv = 1469598103934665603ull;   // checksum of every double of L, y and x (to compare BUILDS, e.g. -DGSFM_ELIM_LDS=0 against 1)
      { const unsigned char* pb = (const unsigned char*)out.data(); for (size_t q = 0; q < 8 * out.size(); ++q) { fnv ^= pb[q]; fnv *= 1099511628211ull; } }
      if (f == 0) ref = out; else for (size_t q = 0; q < out.size(); ++q) diff += std::memcmp(&out[q], &ref[q], 8) != 0;
      // residual of the first live matrix
      double rmax = 0.0;
      for (uint32_t c = 0; c < NI; ++c) if (lv[c]) {
        const uint32_t n = items[c].n; const double* A = h.data() + offA[c]; const double* x = out.data() + (offX[c] - a_words);
        auto a_at = [&](uint32_t r, uint32_t q) { if (q > r) std::swap(r, q); return A[chol_tile_off(r / 32, q / 32) + (r % 32) * 32 + q % 32]; };
        for (uint32_t r = 0; r < n; r += 7) { double s = -A[chol_tile_off(items[c].T, r / 32) + r % 32]; for (uint32_t q = 0; q < n; ++q) s += a_at(r, q) * x[q]; rmax = std::fmax(rmax, std::fabs(s)); }
        break;
      }
      hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
      const int reps = 30;
      CHK(hipEventRecord(e0, st)); for (int k = 0; k < reps; ++k) CHK(hipGraphLaunch(ge, st)); CHK(hipEventRecord(e1, st)); CHK(hipEventSynchronize(e1));
      float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
      printf("  %-52s: %8.1f us per factorisation + solve (graph replay)  info %d  |Ax - b| sampled %.1e  doubles of L, y, x differing from the fused form: %zu  fnv %016llx\n", forms[f].first, 1e3 * ms / reps, *std::max_element(info.begin(), info.end()), rmax, diff, fnv);
      CHK(hipGraphExecDestroy(ge)); CHK(hipGraphDestroy(g));
    }
  }
#ifdef GSFM_BACK_TIMING
  {   // the backward substitution's group kernel, wavefront 0 of matrix 1 (Madrid's size) alone: per block row
    const std::vector<int> only = {0, 1, 0, 0, 0, 0};
    CHK(hipMemcpy(dactive, only.data(), 4 * NI, hipMemcpyHostToDevice));
    enqueue(LOOK2); CHK(hipStreamSynchronize(st));
    static unsigned long long ts[64][4];
    CHK(hipMemcpyFromSymbol(ts, HIP_SYMBOL(gsfm_back_ts), sizeof(ts)));
    double acc[3] = {0, 0, 0}; int n = 0;
    for (uint32_t k = 0; k < items[1].T; ++k, ++n) for (int q = 0; q < 3; ++q) acc[q] += (double)(ts[k][q + 1] - ts[k][q]) * 10.0;   // 100 MHz
    printf("backward substitution, group kernel, matrix 1182 alone, mean per block row over %d rows (ns): 32-step substitution %.0f | barrier %.0f | fold into the group's rows + next diagonal tile into LDS + barrier %.0f\n", n, acc[0] / n, acc[1] / n, acc[2] / n);
  }
#endif
#ifdef GSFM_LOOK_TIMING
  {   // the two-column launch: the first row workgroup of matrix 1 (Madrid's size) alone
    const std::vector<int> only = {0, 1, 0, 0, 0, 0};
    CHK(hipMemcpy(dactive, only.data(), 4 * NI, hipMemcpyHostToDevice));
    enqueue(LOOK2); CHK(hipStreamSynchronize(st));
    static unsigned long long ts[64][8];
    CHK(hipMemcpyFromSymbol(ts, HIP_SYMBOL(gsfm_look2_ts), sizeof(ts)));
    double acc[7] = {0, 0, 0, 0, 0, 0, 0}, landed = 0; int n = 0;
    for (uint32_t g = 1; 2 * g + 3 <= items[1].T; ++g, ++n) {   // launches c0 = 2, 4, ...: two pending columns, two produced, a row workgroup exists
      for (int q = 1; q < 7; ++q) acc[q] += (double)(ts[g][q] - ts[g][q - 1]) * 10.0;
      landed += (double)(ts[g][7] - ts[g][0]) * 10.0;
      if (2 * (g + 1) + 3 <= items[1].T) acc[0] += (double)(ts[g + 1][0] - ts[g][6]) * 10.0;
    }
    printf("two columns per launch, row workgroup, matrix 1182 alone, mean over %d launches (ns): loads + two pending columns on the fly %.0f (of which until the first operands are in LDS: %.0f) | tiles to LDS + barrier %.0f | elimination of column c0 %.0f | barrier %.0f | column c0 into c0 + 1, publish, barrier %.0f | elimination of column c0 + 1 + store %.0f | end -> start of the next %.0f\n",
           n, acc[1] / n, landed / n, acc[2] / n, acc[3] / n, acc[4] / n, acc[5] / n, acc[6] / n, acc[0] / std::max(1, n - 1));
  }
  {   // phase stamps of the first row workgroup of matrix 1 (Madrid's size) in every launch of the one-launch-per-column form (last run: that matrix alone)
    const std::vector<int> only = {0, 1, 0, 0, 0, 0};
    CHK(hipMemcpy(dactive, only.data(), 4 * NI, hipMemcpyHostToDevice));
    enqueue(LOOK); CHK(hipStreamSynchronize(st));
    static unsigned long long ts[128][8];
    CHK(hipMemcpyFromSymbol(ts, HIP_SYMBOL(gsfm_look_ts), sizeof(ts)));
    double acc[6] = {0, 0, 0, 0, 0, 0}; int n = 0;
    for (uint32_t k = 0; k + 3 <= items[1].T; ++k, ++n) {
      for (int q = 1; q < 6; ++q) acc[q] += (double)(ts[k][q] - ts[k][q - 1]) * 10.0;   // 100 MHz
      if (k + 4 <= items[1].T) acc[0] += (double)(ts[k + 1][0] - ts[k][5]) * 10.0;
    }
    printf("row workgroup of the one-launch-per-column form, matrix 1182 alone, mean over %d launches (ns): loads + stage + barrier %.0f | two products + LDS + barrier %.0f | rows from LDS %.0f | elimination %.0f | store %.0f | end of one -> start of the next %.0f\n",
           n, acc[1] / n, acc[2] / n, acc[3] / n, acc[4] / n, acc[5] / n, acc[0] / std::max(1, n - 1));
  }
#endif
  return 0;
}

// Dev comparison (not part of the product, which links nothing but the HIP runtime): rocSOLVER's dpotrf + dpotrs on the same SPD test matrix
// as tools/bench_chol.hip, for the sizes of the exact-step path.  hipcc --offload-arch=gfx950 -O3 -o bench_rocsolver_potrf bench_rocsolver_potrf.hip -lrocsolver -lrocblas
#include <hip/hip_runtime.h>
#include <rocsolver/rocsolver.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
int main(int argc, char** argv) {
  std::vector<int> sizes;
  for (int k = 1; k < argc; ++k) sizes.push_back(atoi(argv[k]));
  if (sizes.empty()) sizes = {1182, 2400, 4500};
  rocblas_handle h; rocblas_create_handle(&h);
  hipStream_t st; CHK(hipStreamCreate(&st)); rocblas_set_stream(h, st);
  for (int n : sizes) {
    std::mt19937_64 rng(n); std::normal_distribution<double> N01(0, 1);
    std::vector<double> B((size_t)n * 8), A((size_t)n * n), b(n);
    for (auto& v : B) v = N01(rng);
    for (auto& v : b) v = N01(rng);
    for (int r = 0; r < n; ++r) for (int c = 0; c <= r; ++c) { double s = (r == c) ? 1.0 : 0.0; for (int t = 0; t < 8; ++t) s += B[(size_t)r * 8 + t] * B[(size_t)c * 8 + t] / 8.0; A[(size_t)r * n + c] = A[(size_t)c * n + r] = s; }
    double *dA0, *dA, *db0, *db; int* dinfo;
    CHK(hipMalloc(&dA0, 8 * (size_t)n * n)); CHK(hipMalloc(&dA, 8 * (size_t)n * n)); CHK(hipMalloc(&db0, 8 * n)); CHK(hipMalloc(&db, 8 * n)); CHK(hipMalloc(&dinfo, 4));
    CHK(hipMemcpy(dA0, A.data(), 8 * (size_t)n * n, hipMemcpyHostToDevice)); CHK(hipMemcpy(db0, b.data(), 8 * n, hipMemcpyHostToDevice));
    auto run = [&]() {
      CHK(hipMemcpyAsync(dA, dA0, 8 * (size_t)n * n, hipMemcpyDeviceToDevice, st)); CHK(hipMemcpyAsync(db, db0, 8 * n, hipMemcpyDeviceToDevice, st));
      rocsolver_dpotrf(h, rocblas_fill_lower, n, dA, n, dinfo);
      rocsolver_dpotrs(h, rocblas_fill_lower, n, 1, dA, n, db, n);
    };
    run(); run(); CHK(hipStreamSynchronize(st));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const int reps = 10;
    CHK(hipEventRecord(e0, st)); for (int k = 0; k < reps; ++k) run(); CHK(hipEventRecord(e1, st)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<double> x(n); int info; CHK(hipMemcpy(x.data(), db, 8 * n, hipMemcpyDeviceToHost)); CHK(hipMemcpy(&info, dinfo, 4, hipMemcpyDeviceToHost));
    double rmax = 0; for (int r = 0; r < n; ++r) { double s = -b[r]; for (int c = 0; c < n; ++c) s += A[(size_t)r * n + c] * x[c]; rmax = fmax(rmax, fabs(s)); }
    printf("rocSOLVER n = %5d: info %d  max |Ax - b| = %.2e   dpotrf + dpotrs (+ the two copies) %.3f ms per call\n", n, info, rmax, ms / reps);
    CHK(hipFree(dA0)); CHK(hipFree(dA)); CHK(hipFree(db0)); CHK(hipFree(db)); CHK(hipFree(dinfo));
  }
  return 0;
}

// Dev micro-benchmark (not part of the product): symmetric (one block per EDGE) storage of the Laplacian-form normal matrix.  Rows hold only
// their "upper" entries (neighbour index > row index), the lower half of the product is scattered with fp64 atomics (non-deterministic sums).
// Measures whether halving the streamed bytes pays for 3 atomics per edge.  C5 shape: 100k rows, ~100 upper entries per row.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
__device__ __forceinline__ double2 nt2(const double2* p) { double2 v; v.x = __builtin_nontemporal_load(&p->x); v.y = __builtin_nontemporal_load(&p->y); return v; }
struct Args { unsigned n_rows, G; const unsigned* row_ptr; const unsigned* col; const double2 *h0, *h1, *h2; const double* u; double* y; };

template <int MODE>  // 0: gather + atomics (full symmetric product); 1: gather only (upper half of the product); 2: atomics without the gather
__global__ void __launch_bounds__(256) k_mv_sym(Args a) {
  const unsigned G = a.G, t = blockIdx.x * 256 + threadIdx.x, row = t / G, lane = t % G;
  const bool live = row < a.n_rows;
  double y0 = 0, y1 = 0, y2 = 0;
  if (live) {
    const double* ui = a.u + 3 * (size_t)row;
    const double i0 = ui[0], i1 = ui[1], i2 = ui[2];
    const unsigned end = a.row_ptr[row + 1];
    for (unsigned d = a.row_ptr[row] + lane; d < end; d += G) {
      const unsigned m = __builtin_nontemporal_load(a.col + d);
      const double2 A = nt2(a.h0 + d), B = nt2(a.h1 + d), C = nt2(a.h2 + d);
      if (MODE != 2) {
        const double* um = a.u + 3 * (size_t)m; const double u0 = um[0], u1 = um[1], u2 = um[2];
        y0 += A.x * u0 + A.y * u1 + B.x * u2; y1 += A.y * u0 + B.y * u1 + C.x * u2; y2 += B.x * u0 + C.x * u1 + C.y * u2;
      }
      if (MODE != 1) {
        double* ym = a.y + 3 * (size_t)m;
        unsafeAtomicAdd(ym, A.x * i0 + A.y * i1 + B.x * i2); unsafeAtomicAdd(ym + 1, A.y * i0 + B.y * i1 + C.x * i2); unsafeAtomicAdd(ym + 2, B.x * i0 + C.x * i1 + C.y * i2);
      }
    }
  }
  for (unsigned off = G >> 1; off > 0; off >>= 1) { y0 += __shfl_down(y0, off, G); y1 += __shfl_down(y1, off, G); y2 += __shfl_down(y2, off, G); }
  if (live && lane == 0 && MODE != 2) { double* yr = a.y + 3 * (size_t)row; unsafeAtomicAdd(yr, y0); unsafeAtomicAdd(yr + 1, y1); unsafeAtomicAdd(yr + 2, y2); }
}
template <typename F> float timeit(F f, int reps = 10) {
  hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  f(); f();
  CHK(hipEventRecord(e0));
  for (int k = 0; k < reps; ++k) f();
  CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
  float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); return ms / reps * 1e3f;
}
int main() {
  const unsigned N = 100000, DEG = 200;
  std::mt19937 rng(1);
  std::vector<std::vector<unsigned>> up(N);
  for (unsigned r = 0; r < N; ++r) for (unsigned k = 0; k < DEG / 2; ++k) { unsigned c = rng() % N; if (c == r) continue; up[c < r ? c : r].push_back(c < r ? r : c); }
  std::vector<unsigned> rp(N + 1, 0), col;
  for (unsigned r = 0; r < N; ++r) { rp[r + 1] = rp[r] + (unsigned)up[r].size(); col.insert(col.end(), up[r].begin(), up[r].end()); }
  const size_t ne = col.size();
  unsigned *d_rp, *d_col; double2 *h0, *h1, *h2; double *u, *y;
  CHK(hipMalloc(&d_rp, 4 * (N + 1))); CHK(hipMalloc(&d_col, 4 * ne)); CHK(hipMalloc(&h0, 16 * ne)); CHK(hipMalloc(&h1, 16 * ne)); CHK(hipMalloc(&h2, 16 * ne));
  CHK(hipMalloc(&u, 24 * (size_t)N)); CHK(hipMalloc(&y, 24 * (size_t)N));
  CHK(hipMemcpy(d_rp, rp.data(), 4 * (N + 1), hipMemcpyHostToDevice)); CHK(hipMemcpy(d_col, col.data(), 4 * ne, hipMemcpyHostToDevice));
  CHK(hipMemset(h0, 0, 16 * ne)); CHK(hipMemset(h1, 0, 16 * ne)); CHK(hipMemset(h2, 0, 16 * ne)); CHK(hipMemset(u, 0, 24 * (size_t)N)); CHK(hipMemset(y, 0, 24 * (size_t)N));
  printf("rows %u, edges %zu (upper entries; the row lengths fall from %zu to %zu), %.3f GB streamed per product (52 B per edge)\n", N, ne, up[0].size(), up[N - 1].size(), 52.0 * ne * 1e-9);
  const char* names[3] = {"gather + 3 atomics per edge", "gather only (half product)", "atomics only"};
  for (unsigned G : {64u, 32u, 16u}) {
    Args a{N, G, d_rp, d_col, h0, h1, h2, u, y};
    const int grid = (int)(((size_t)N * G + 255) / 256);
    float t;
    t = timeit([&] { hipLaunchKernelGGL(k_mv_sym<0>, dim3(grid), dim3(256), 0, 0, a); }); printf("G=%2u %-30s %8.1f us\n", G, names[0], t);
    t = timeit([&] { hipLaunchKernelGGL(k_mv_sym<1>, dim3(grid), dim3(256), 0, 0, a); }); printf("G=%2u %-30s %8.1f us\n", G, names[1], t);
    t = timeit([&] { hipLaunchKernelGGL(k_mv_sym<2>, dim3(grid), dim3(256), 0, 0, a); }); printf("G=%2u %-30s %8.1f us\n", G, names[2], t);
  }
  CHK(hipDeviceSynchronize());
  return 0;
}

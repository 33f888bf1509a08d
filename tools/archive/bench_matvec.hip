// Dev micro-benchmark (not part of the product): variants of the block-CSR mat-vec on synthetic data
// shaped like C5 (100k rows x 200 entries).  hipcc --offload-arch=gfx950 -O3 -o bench_matvec bench_matvec.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>
#include <random>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

struct Args { unsigned n_rows, G; const unsigned* row_ptr; const unsigned* col; const double2 *h0,*h1,*h2,*h3; const double* h4; const double* p; const double4* p4; double* y; };

template <int MODE>  // 0: full (3x8B gather) 1: no gather 2: padded double4 gather 3: nontemporal H + 3x8B 4: nontemporal + padded
__global__ void __launch_bounds__(256) k_mv(Args a) {
  const unsigned G = a.G, t = blockIdx.x * 256 + threadIdx.x, row = t / G, lane = t % G;
  const bool live = row < a.n_rows;
  double y0 = 0, y1 = 0, y2 = 0;
  if (live) {
    const unsigned end = a.row_ptr[row + 1];
    for (unsigned d = a.row_ptr[row] + lane; d < end; d += G) {
      double2 A, B, C, D; double E; unsigned m;
      if (MODE >= 3) {
        A.x = __builtin_nontemporal_load(&a.h0[d].x); A.y = __builtin_nontemporal_load(&a.h0[d].y);
        B.x = __builtin_nontemporal_load(&a.h1[d].x); B.y = __builtin_nontemporal_load(&a.h1[d].y);
        C.x = __builtin_nontemporal_load(&a.h2[d].x); C.y = __builtin_nontemporal_load(&a.h2[d].y);
        D.x = __builtin_nontemporal_load(&a.h3[d].x); D.y = __builtin_nontemporal_load(&a.h3[d].y);
        E = __builtin_nontemporal_load(&a.h4[d]); m = __builtin_nontemporal_load(&a.col[d]);
      } else { A = a.h0[d]; B = a.h1[d]; C = a.h2[d]; D = a.h3[d]; E = a.h4[d]; m = a.col[d]; }
      double p0, p1, p2;
      if (MODE == 1) { p0 = 1.0 + (m & 1); p1 = 2.0; p2 = 3.0; }
      else if (MODE == 2 || MODE == 4) { const double4 v = a.p4[m]; p0 = v.x; p1 = v.y; p2 = v.z; }
      else { const double* pm = a.p + 3 * (size_t)m; p0 = pm[0]; p1 = pm[1]; p2 = pm[2]; }
      y0 += A.x * p0 + A.y * p1 + B.x * p2; y1 += B.y * p0 + C.x * p1 + C.y * p2; y2 += D.x * p0 + D.y * p1 + E * p2;
    }
  }
  for (unsigned off = G >> 1; off > 0; off >>= 1) { y0 += __shfl_down(y0, off, G); y1 += __shfl_down(y1, off, G); y2 += __shfl_down(y2, off, G); }
  if (live && lane == 0) { a.y[3 * (size_t)row] = y0; a.y[3 * (size_t)row + 1] = y1; a.y[3 * (size_t)row + 2] = y2; }
}


// unrolled by two: both trips' column indices and blocks are requested before either gather is consumed
__global__ void __launch_bounds__(256) k_mv_u2(Args a) {
  const unsigned G = a.G, t = blockIdx.x * 256 + threadIdx.x, row = t / G, lane = t % G;
  const bool live = row < a.n_rows;
  double y0 = 0, y1 = 0, y2 = 0;
  if (live) {
    const unsigned end = a.row_ptr[row + 1];
    unsigned d = a.row_ptr[row] + lane;
    for (; d + G < end; d += 2 * G) {
      const unsigned d2 = d + G;
      const unsigned m1 = __builtin_nontemporal_load(&a.col[d]), m2 = __builtin_nontemporal_load(&a.col[d2]);
      double2 A1, B1, C1, D1, A2, B2, C2, D2;
      A1.x = __builtin_nontemporal_load(&a.h0[d].x); A1.y = __builtin_nontemporal_load(&a.h0[d].y);
      B1.x = __builtin_nontemporal_load(&a.h1[d].x); B1.y = __builtin_nontemporal_load(&a.h1[d].y);
      C1.x = __builtin_nontemporal_load(&a.h2[d].x); C1.y = __builtin_nontemporal_load(&a.h2[d].y);
      D1.x = __builtin_nontemporal_load(&a.h3[d].x); D1.y = __builtin_nontemporal_load(&a.h3[d].y);
      A2.x = __builtin_nontemporal_load(&a.h0[d2].x); A2.y = __builtin_nontemporal_load(&a.h0[d2].y);
      B2.x = __builtin_nontemporal_load(&a.h1[d2].x); B2.y = __builtin_nontemporal_load(&a.h1[d2].y);
      C2.x = __builtin_nontemporal_load(&a.h2[d2].x); C2.y = __builtin_nontemporal_load(&a.h2[d2].y);
      D2.x = __builtin_nontemporal_load(&a.h3[d2].x); D2.y = __builtin_nontemporal_load(&a.h3[d2].y);
      const double E1 = __builtin_nontemporal_load(&a.h4[d]), E2 = __builtin_nontemporal_load(&a.h4[d2]);
      const double* p1 = a.p + 3 * (size_t)m1; const double* p2 = a.p + 3 * (size_t)m2;
      const double a0 = p1[0], a1 = p1[1], a2 = p1[2], b0 = p2[0], b1 = p2[1], b2 = p2[2];
      y0 += A1.x * a0 + A1.y * a1 + B1.x * a2 + A2.x * b0 + A2.y * b1 + B2.x * b2;
      y1 += B1.y * a0 + C1.x * a1 + C1.y * a2 + B2.y * b0 + C2.x * b1 + C2.y * b2;
      y2 += D1.x * a0 + D1.y * a1 + E1 * a2 + D2.x * b0 + D2.y * b1 + E2 * b2;
    }
    if (d < end) {
      const unsigned m = __builtin_nontemporal_load(&a.col[d]);
      const double2 A = a.h0[d], B = a.h1[d], C = a.h2[d], D = a.h3[d]; const double E = a.h4[d];
      const double* pm = a.p + 3 * (size_t)m;
      y0 += A.x * pm[0] + A.y * pm[1] + B.x * pm[2]; y1 += B.y * pm[0] + C.x * pm[1] + C.y * pm[2]; y2 += D.x * pm[0] + D.y * pm[1] + E * pm[2];
    }
  }
  for (unsigned off = G >> 1; off > 0; off >>= 1) { y0 += __shfl_down(y0, off, G); y1 += __shfl_down(y1, off, G); y2 += __shfl_down(y2, off, G); }
  if (live && lane == 0) { a.y[3 * (size_t)row] = y0; a.y[3 * (size_t)row + 1] = y1; a.y[3 * (size_t)row + 2] = y2; }
}


// fp32 copy of the blocks (two float4 planes + one float plane, 36 B + 4 B col per entry); p and y stay fp64
struct Args32 { unsigned n_rows, G; const unsigned* row_ptr; const unsigned* col; const float4 *g0, *g1; const float* g2; const double* p; double* y; };
__global__ void __launch_bounds__(256) k_mv_f32(Args32 a) {
  const unsigned G = a.G, t = blockIdx.x * 256 + threadIdx.x, row = t / G, lane = t % G;
  const bool live = row < a.n_rows;
  double y0 = 0, y1 = 0, y2 = 0;
  if (live) {
    const unsigned end = a.row_ptr[row + 1];
    for (unsigned d = a.row_ptr[row] + lane; d < end; d += G) {
      const unsigned m = __builtin_nontemporal_load(&a.col[d]);
      float4 A, B;
      A.x = __builtin_nontemporal_load(&a.g0[d].x); A.y = __builtin_nontemporal_load(&a.g0[d].y); A.z = __builtin_nontemporal_load(&a.g0[d].z); A.w = __builtin_nontemporal_load(&a.g0[d].w);
      B.x = __builtin_nontemporal_load(&a.g1[d].x); B.y = __builtin_nontemporal_load(&a.g1[d].y); B.z = __builtin_nontemporal_load(&a.g1[d].z); B.w = __builtin_nontemporal_load(&a.g1[d].w);
      const float E = __builtin_nontemporal_load(&a.g2[d]);
      const double* pm = a.p + 3 * (size_t)m;
      const double p0 = pm[0], p1 = pm[1], p2 = pm[2];
      y0 += (double)A.x * p0 + (double)A.y * p1 + (double)A.z * p2;
      y1 += (double)A.w * p0 + (double)B.x * p1 + (double)B.y * p2;
      y2 += (double)B.z * p0 + (double)B.w * p1 + (double)E * p2;
    }
  }
  for (unsigned off = G >> 1; off > 0; off >>= 1) { y0 += __shfl_down(y0, off, G); y1 += __shfl_down(y1, off, G); y2 += __shfl_down(y2, off, G); }
  if (live && lane == 0) { a.y[3 * (size_t)row] = y0; a.y[3 * (size_t)row + 1] = y1; a.y[3 * (size_t)row + 2] = y2; }
}

// AoS variant: one 80-byte record per entry (9 doubles + col + pad) read as 5 x 16 B
struct Rec { double h[9]; unsigned col, pad; };
__global__ void __launch_bounds__(256) k_mv_aos(unsigned n_rows, unsigned G, const unsigned* row_ptr, const Rec* rec, const double4* p4, double* y) {
  const unsigned t = blockIdx.x * 256 + threadIdx.x, row = t / G, lane = t % G;
  const bool live = row < n_rows;
  double y0 = 0, y1 = 0, y2 = 0;
  if (live) {
    const unsigned end = row_ptr[row + 1];
    for (unsigned d = row_ptr[row] + lane; d < end; d += G) {
      const double2* r = (const double2*)&rec[d];
      const double2 A = r[0], B = r[1], C = r[2], D = r[3], E = r[4];
      const unsigned m = __double_as_longlong(E.y) & 0xffffffffu;
      const double4 v = p4[m];
      y0 += A.x * v.x + A.y * v.y + B.x * v.z; y1 += B.y * v.x + C.x * v.y + C.y * v.z; y2 += D.x * v.x + D.y * v.y + E.x * v.z;
    }
  }
  for (unsigned off = G >> 1; off > 0; off >>= 1) { y0 += __shfl_down(y0, off, G); y1 += __shfl_down(y1, off, G); y2 += __shfl_down(y2, off, G); }
  if (live && lane == 0) { y[3 * (size_t)row] = y0; y[3 * (size_t)row + 1] = y1; y[3 * (size_t)row + 2] = y2; }
}
// pure streaming read of the same byte count (reference ceiling)
__global__ void __launch_bounds__(256) k_stream(const double2* a, size_t n, double* out) {
  double s = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { const double2 v = a[i]; s += v.x + v.y; }
  if (s == 12345.678) out[0] = s;
}

template <typename F> float timeit(F f, int reps = 10) {
  hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  f(); f();
  CHK(hipEventRecord(e0));
  for (int k = 0; k < reps; ++k) f();
  CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
  float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); return ms / reps * 1e3f;
}

int main(int argc, char** argv) {
  const unsigned N = argc > 1 ? atoi(argv[1]) : 100000, DEG = argc > 2 ? atoi(argv[2]) : 200;
  const size_t nd = (size_t)N * DEG;
  std::vector<unsigned> rp(N + 1), col(nd);
  std::mt19937 rng(1);
  for (unsigned r = 0; r <= N; ++r) rp[r] = r * DEG;
  // argv[3] = locality window W (0 = uniform random columns): neighbours of row r drawn from [r - W/2, r + W/2], sorted
  const unsigned WIN = argc > 3 ? atoi(argv[3]) : 0;
  for (unsigned r = 0; r < N; ++r) {
    for (unsigned k = 0; k < DEG; ++k) col[(size_t)r * DEG + k] = WIN ? (unsigned)((r + N + (rng() % WIN) - WIN / 2) % N) : rng() % N;
    if (WIN) std::sort(col.begin() + (size_t)r * DEG, col.begin() + (size_t)(r + 1) * DEG);
  }
  printf("columns: %s (window %u)\n", WIN ? "local" : "uniform random", WIN);
  unsigned *d_rp, *d_col; double2 *h0, *h1, *h2, *h3; double *h4, *p, *y; double4* p4; Rec* rec;
  CHK(hipMalloc(&d_rp, 4 * (N + 1))); CHK(hipMalloc(&d_col, 4 * nd));
  CHK(hipMalloc(&h0, 16 * nd)); CHK(hipMalloc(&h1, 16 * nd)); CHK(hipMalloc(&h2, 16 * nd)); CHK(hipMalloc(&h3, 16 * nd)); CHK(hipMalloc(&h4, 8 * nd));
  CHK(hipMalloc(&p, 24 * (size_t)N)); CHK(hipMalloc(&p4, 32 * (size_t)N)); CHK(hipMalloc(&y, 24 * (size_t)N)); CHK(hipMalloc(&rec, sizeof(Rec) * nd));
  CHK(hipMemcpy(d_rp, rp.data(), 4 * (N + 1), hipMemcpyHostToDevice)); CHK(hipMemcpy(d_col, col.data(), 4 * nd, hipMemcpyHostToDevice));
  CHK(hipMemset(h0, 0, 16 * nd)); CHK(hipMemset(h1, 0, 16 * nd)); CHK(hipMemset(h2, 0, 16 * nd)); CHK(hipMemset(h3, 0, 16 * nd)); CHK(hipMemset(h4, 0, 8 * nd));
  CHK(hipMemset(p, 0, 24 * (size_t)N)); CHK(hipMemset(p4, 0, 32 * (size_t)N));
  { std::vector<Rec> hr(nd); for (size_t d = 0; d < nd; ++d) { for (int k = 0; k < 9; ++k) hr[d].h[k] = 1e-3 * k; hr[d].col = col[d]; hr[d].pad = 0; }
    CHK(hipMemcpy(rec, hr.data(), sizeof(Rec) * nd, hipMemcpyHostToDevice)); }
  double2* big; CHK(hipMalloc(&big, 64 * nd)); CHK(hipMemset(big, 0, 64 * nd));
  const double bytes = 76.0 * nd;
  for (unsigned G : {64u, 32u}) {
    Args a{N, G, d_rp, d_col, h0, h1, h2, h3, h4, p, p4, y};
    const int grid = (int)(((size_t)N * G + 255) / 256);
    const char* names[5] = {"planes+3x8B gather", "planes, no gather", "planes+double4 gather", "nontemporal+3x8B", "nontemporal+double4"};
    float t;
    t = timeit([&] { hipLaunchKernelGGL(k_mv<0>, dim3(grid), dim3(256), 0, 0, a); }); printf("G=%2u %-24s %8.1f us  %6.2f TB/s\n", G, names[0], t, bytes / t * 1e-6);
    t = timeit([&] { hipLaunchKernelGGL(k_mv<1>, dim3(grid), dim3(256), 0, 0, a); }); printf("G=%2u %-24s %8.1f us  %6.2f TB/s\n", G, names[1], t, bytes / t * 1e-6);
    t = timeit([&] { hipLaunchKernelGGL(k_mv<2>, dim3(grid), dim3(256), 0, 0, a); }); printf("G=%2u %-24s %8.1f us  %6.2f TB/s\n", G, names[2], t, bytes / t * 1e-6);
    t = timeit([&] { hipLaunchKernelGGL(k_mv<3>, dim3(grid), dim3(256), 0, 0, a); }); printf("G=%2u %-24s %8.1f us  %6.2f TB/s\n", G, names[3], t, bytes / t * 1e-6);
    t = timeit([&] { hipLaunchKernelGGL(k_mv<4>, dim3(grid), dim3(256), 0, 0, a); }); printf("G=%2u %-24s %8.1f us  %6.2f TB/s\n", G, names[4], t, bytes / t * 1e-6);
    { Args32 b{N, G, d_rp, d_col, (const float4*)h0, (const float4*)h1, (const float*)h2, p, y};
      t = timeit([&] { hipLaunchKernelGGL(k_mv_f32, dim3(grid), dim3(256), 0, 0, b); }); printf("G=%2u %-24s %8.1f us  %6.2f TB/s (40 B/entry)\n", G, "fp32 blocks, fp64 p/y", t, 40.0 * nd / t * 1e-6); }
    t = timeit([&] { hipLaunchKernelGGL(k_mv_u2, dim3(grid), dim3(256), 0, 0, a); }); printf("G=%2u %-24s %8.1f us  %6.2f TB/s\n", G, "nt + 3x8B, unroll 2", t, bytes / t * 1e-6);
    t = timeit([&] { hipLaunchKernelGGL(k_mv_aos, dim3(grid), dim3(256), 0, 0, N, G, d_rp, rec, p4, y); }); printf("G=%2u %-24s %8.1f us  %6.2f TB/s (80 B/entry)\n", G, "AoS 80B + double4", t, 80.0 * nd / t * 1e-6);
  }
  for (int blocks : {2048, 8192, 32768}) {
    float t = timeit([&] { hipLaunchKernelGGL(k_stream, dim3(blocks), dim3(256), 0, 0, big, nd * 4, y); });
    printf("stream read 64B/entry x %zu, %d blocks: %8.1f us  %6.2f TB/s\n", nd, blocks, t, 64.0 * nd / t * 1e-6);
  }
  CHK(hipDeviceSynchronize());
  return 0;
}

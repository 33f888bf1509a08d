// Dev micro-benchmark (not part of the product): variants of the Laplacian-form block-CSR mat-vec (52 B per directed entry) on the C5 shape
// (100k rows, Poisson-ish degrees around 200, uniformly random columns).  hipcc --offload-arch=gfx950 -O3 -o bench_matvec3 bench_matvec3.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

struct Args { unsigned n_rows, G; const unsigned* row_ptr; const unsigned* col; const double2 *h0, *h1, *h2; const double* u; const double4* u4; const float* uf; const double* M; const double* p; const double2* q; double* y; };
__device__ __forceinline__ double2 nt2(const double2* p) { double2 v; v.x = __builtin_nontemporal_load(&p->x); v.y = __builtin_nontemporal_load(&p->y); return v; }
__device__ __forceinline__ double ld_sc1(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// MODE 0 product form; 1 no gather; 2 gather only (no planes); 3 agent-scope (L1-bypassing) gather; 4 double4-padded gather; 5 fp32 gather (upper bound of a 1-request gather)
// 6 next trip's column index requested before this trip's gather is consumed
template <int MODE, int BLOCK>
__global__ void __launch_bounds__(BLOCK) k_mv(Args a) {
  const unsigned G = a.G, t = blockIdx.x * BLOCK + threadIdx.x, row = t / G, lane = t % G;
  const bool live = row < a.n_rows;
  double y0 = 0, y1 = 0, y2 = 0;
  if (live) {
    const double2 qa = a.q[2 * (size_t)row], qb = a.q[2 * (size_t)row + 1];
    const double R0 = qa.x, R1 = qa.y, R2 = qb.x, R3 = qb.y, R4 = qa.x * qb.y, R5 = qa.y * qb.x, R6 = qa.x + qb.x, R7 = qa.y - qb.y, R8 = qb.x * qb.y;   // stand-in for the 3x3 rotation
    const unsigned end = a.row_ptr[row + 1];
    unsigned d = a.row_ptr[row] + lane;
    unsigned mnext = (MODE == 6 && d < end) ? (__builtin_nontemporal_load(a.col + d) & 0x7fffffffu) : 0;
    for (; d < end; d += G) {
      unsigned m;
      if (MODE == 6) { m = mnext; if (d + G < end) mnext = __builtin_nontemporal_load(a.col + d + G) & 0x7fffffffu; }
      else m = __builtin_nontemporal_load(a.col + d) & 0x7fffffffu;
      double2 A = make_double2(1, 2), B = make_double2(3, 4), C = make_double2(5, 6);
      if (MODE != 2) { A = nt2(a.h0 + d); B = nt2(a.h1 + d); C = nt2(a.h2 + d); }
      double u0, u1, u2;
      if (MODE == 1) { u0 = 1.0 + (m & 1); u1 = 2.0; u2 = 3.0; }
      else if (MODE == 3) { const double* um = a.u + 3 * (size_t)m; u0 = ld_sc1(um); u1 = ld_sc1(um + 1); u2 = ld_sc1(um + 2); }
      else if (MODE == 4) { const double4 v = a.u4[m]; u0 = v.x; u1 = v.y; u2 = v.z; }
      else if (MODE == 5) { const float* um = a.uf + 3 * (size_t)m; u0 = um[0]; u1 = um[1]; u2 = um[2]; }
      else { const double* um = a.u + 3 * (size_t)m; u0 = um[0]; u1 = um[1]; u2 = um[2]; }
      const double w0 = R0 * u0 + R1 * u1 + R2 * u2, w1 = R3 * u0 + R4 * u1 + R5 * u2, w2 = R6 * u0 + R7 * u1 + R8 * u2;
      y0 += A.x * w0 + A.y * w1 + B.x * w2; y1 += A.y * w0 + B.y * w1 + C.x * w2; y2 += B.x * w0 + C.x * w1 + C.y * w2;
    }
  }
  for (unsigned off = G >> 1; off > 0; off >>= 1) { y0 += __shfl_down(y0, off, G); y1 += __shfl_down(y1, off, G); y2 += __shfl_down(y2, off, G); }
  if (live && lane == 0) {
    const double* M = a.M + 6 * (size_t)row; const double* pk = a.p + 3 * (size_t)row;
    a.y[3 * (size_t)row] = M[0] * pk[0] + M[1] * pk[1] + M[2] * pk[2] - y0; a.y[3 * (size_t)row + 1] = M[1] * pk[0] + M[3] * pk[1] + M[4] * pk[2] - y1;
    a.y[3 * (size_t)row + 2] = M[2] * pk[0] + M[4] * pk[1] + M[5] * pk[2] - y2;
  }
}

// CSR-stream: a wavefront owns 64 * K consecutive ENTRIES (full lanes every trip), rows are recovered from a per-entry row id, segmented
// reduction with shuffles; rows straddling wavefronts are finished with (non-deterministic, benchmark only) atomics.
template <int K>
__global__ void __launch_bounds__(256) k_mv_stream(Args a, const unsigned* __restrict__ rowid, size_t nd) {
  const size_t wave = ((size_t)blockIdx.x * 256 + threadIdx.x) / 64; const unsigned lane = threadIdx.x & 63;
  const size_t base = wave * 64 * K;
  for (int k = 0; k < K; ++k) {
    const size_t d = base + (size_t)k * 64 + lane;
    double y0 = 0, y1 = 0, y2 = 0; unsigned r = 0xffffffffu;
    if (d < nd) {
      r = rowid[d];
      const unsigned m = __builtin_nontemporal_load(a.col + d) & 0x7fffffffu;
      const double2 A = nt2(a.h0 + d), B = nt2(a.h1 + d), C = nt2(a.h2 + d);
      const double* um = a.u + 3 * (size_t)m; const double u0 = um[0], u1 = um[1], u2 = um[2];
      y0 = A.x * u0 + A.y * u1 + B.x * u2; y1 = A.y * u0 + B.y * u1 + C.x * u2; y2 = B.x * u0 + C.x * u1 + C.y * u2;
    }
    // segmented inclusive scan from the right: lane i accumulates lanes > i of the same row
    for (int off = 1; off < 64; off <<= 1) {
      const double t0 = __shfl_down(y0, off, 64), t1 = __shfl_down(y1, off, 64), t2 = __shfl_down(y2, off, 64);
      const unsigned rr = __shfl_down(r, off, 64);
      if (lane + off < 64 && rr == r) { y0 += t0; y1 += t1; y2 += t2; }
    }
    const unsigned rprev = __shfl_up(r, 1, 64);
    if (r != 0xffffffffu && (lane == 0 || rprev != r)) { atomicAdd(a.y + 3 * (size_t)r, y0); atomicAdd(a.y + 3 * (size_t)r + 1, y1); atomicAdd(a.y + 3 * (size_t)r + 2, y2); }
  }
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
  const unsigned N = argc > 1 ? atoi(argv[1]) : 100000, DEG = argc > 2 ? atoi(argv[2]) : 200; const int jitter = argc > 3 ? atoi(argv[3]) : 14;
  std::mt19937 rng(1);
  std::vector<unsigned> rp(N + 1, 0);
  std::normal_distribution<double> nd01(0, 1);
  for (unsigned r = 0; r < N; ++r) { int dg = (int)(DEG + jitter * nd01(rng) + 0.5); if (dg < 1) dg = 1; rp[r + 1] = rp[r] + dg; }
  const size_t nd = rp[N];
  std::vector<unsigned> col(nd), rid(nd);
  for (unsigned r = 0; r < N; ++r) for (unsigned d = rp[r]; d < rp[r + 1]; ++d) { col[d] = rng() % N; rid[d] = r; }
  unsigned *d_rp, *d_col, *d_rid; double2 *h0, *h1, *h2, *q; double *u, *M, *p, *y; double4* u4; float* uf;
  CHK(hipMalloc(&d_rp, 4 * (N + 1))); CHK(hipMalloc(&d_col, 4 * nd)); CHK(hipMalloc(&d_rid, 4 * nd));
  CHK(hipMalloc(&h0, 16 * nd)); CHK(hipMalloc(&h1, 16 * nd)); CHK(hipMalloc(&h2, 16 * nd)); CHK(hipMalloc(&q, 32 * (size_t)N));
  CHK(hipMalloc(&u, 24 * (size_t)N)); CHK(hipMalloc(&u4, 32 * (size_t)N)); CHK(hipMalloc(&uf, 12 * (size_t)N)); CHK(hipMalloc(&M, 48 * (size_t)N)); CHK(hipMalloc(&p, 24 * (size_t)N)); CHK(hipMalloc(&y, 24 * (size_t)N));
  CHK(hipMemcpy(d_rp, rp.data(), 4 * (N + 1), hipMemcpyHostToDevice)); CHK(hipMemcpy(d_col, col.data(), 4 * nd, hipMemcpyHostToDevice)); CHK(hipMemcpy(d_rid, rid.data(), 4 * nd, hipMemcpyHostToDevice));
  CHK(hipMemset(h0, 0, 16 * nd)); CHK(hipMemset(h1, 0, 16 * nd)); CHK(hipMemset(h2, 0, 16 * nd)); CHK(hipMemset(q, 0, 32 * (size_t)N)); CHK(hipMemset(u, 0, 24 * (size_t)N));
  CHK(hipMemset(u4, 0, 32 * (size_t)N)); CHK(hipMemset(uf, 0, 12 * (size_t)N)); CHK(hipMemset(M, 0, 48 * (size_t)N)); CHK(hipMemset(p, 0, 24 * (size_t)N));
  const double bytes = 52.0 * nd + 48.0 * N;
  printf("rows %u, directed entries %zu (mean degree %.1f, sd %d), %.3f GB algorithmic per product\n", N, nd, (double)nd / N, jitter, bytes * 1e-9);
  const char* names[7] = {"product form (3x8B gather)", "no gather", "gather only (no planes)", "agent-scope gather (L1 bypass)", "double4 gather", "fp32 gather (12 B)", "column index one trip ahead"};
  for (unsigned G : {64u, 32u}) {
    Args a{N, G, d_rp, d_col, h0, h1, h2, u, u4, uf, M, p, q, y};
    const int grid = (int)(((size_t)N * G + 255) / 256);
    float t;
#define RUN(MODE) t = timeit([&] { hipLaunchKernelGGL((k_mv<MODE, 256>), dim3(grid), dim3(256), 0, 0, a); }); printf("G=%2u %-34s %8.1f us  %6.2f TB/s\n", G, names[MODE], t, bytes / t * 1e-6);
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6)
    const int grid1k = (int)(((size_t)N * G + 1023) / 1024);
    t = timeit([&] { hipLaunchKernelGGL((k_mv<0, 1024>), dim3(grid1k), dim3(1024), 0, 0, a); }); printf("G=%2u %-34s %8.1f us  %6.2f TB/s\n", G, "product form, 1024-lane workgroups", t, bytes / t * 1e-6);
  }
  {
    Args a{N, 64, d_rp, d_col, h0, h1, h2, u, u4, uf, M, p, q, y};
    const size_t waves = (nd + 255) / 256; const int grid = (int)((waves * 64 + 255) / 256);
    float t = timeit([&] { hipLaunchKernelGGL((k_mv_stream<4>), dim3(grid), dim3(256), 0, 0, a, (const unsigned*)d_rid, nd); });
    printf("     %-34s %8.1f us  %6.2f TB/s (56 B/entry with the row ids)\n", "entry-parallel CSR-stream, K=4", t, (56.0 * nd) / t * 1e-6);
  }
  CHK(hipDeviceSynchronize());
  return 0;
}

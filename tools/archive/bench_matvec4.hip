// Dev micro-benchmark (not part of the product): the Laplacian-form mat-vec with the gathered vector staged in LDS.  The matrix is cut into
// (row range x column block) tiles; one 1024-lane workgroup per tile loads its column block of u into LDS (SoA), then runs G lanes per
// (row, tile) segment over the tile's entries (stored contiguously, sorted by row, 2-byte block-local column index), writes one partial
// y per (row, column block); a second kernel sums the partials.  C5 shape: 100k rows, degree ~200, uniformly random columns.
// hipcc --offload-arch=gfx950 -O3 -o bench_matvec4 bench_matvec4.hip ; ./bench_matvec4 [col_block] [row_ranges] [G]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#include <algorithm>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
__device__ __forceinline__ double2 nt2(const double2* p) { double2 v; v.x = __builtin_nontemporal_load(&p->x); v.y = __builtin_nontemporal_load(&p->y); return v; }

struct TArgs {
  unsigned n_rows, CB, RR, n_cb, G;
  const unsigned* tile_seg;      // [tiles][RR + 1] entry offsets (global) of the (row, tile) segments
  const unsigned short* col;     // block-local column per entry (tiled order)
  const double2 *h0, *h1, *h2;   // G planes, tiled order
  const double* u;               // 3 per camera
  const double2* q;              // row quaternions
  double* part;                  // [n_cb][n_rows][3]
};

template <int THREADS>
__global__ void __launch_bounds__(THREADS) k_mv_tiled(TArgs a) {
  extern __shared__ double lds[];   // ux[CB] uy[CB] uz[CB]
  const unsigned tile = blockIdx.x, cb = tile % a.n_cb, rr = tile / a.n_cb;
  const unsigned c0 = cb * a.CB, cn = min(a.CB, a.n_rows - c0);
  double *ux = lds, *uy = lds + a.CB, *uz = lds + 2 * a.CB;
  for (unsigned c = threadIdx.x; c < cn; c += THREADS) { const double* s = a.u + 3 * (size_t)(c0 + c); ux[c] = s[0]; uy[c] = s[1]; uz[c] = s[2]; }
  __syncthreads();
  const unsigned G = a.G, lane = threadIdx.x % G, grp = threadIdx.x / G, groups = THREADS / G;
  const unsigned r0 = rr * a.RR, rn = min(a.RR, a.n_rows - r0);
  const unsigned* seg = a.tile_seg + (size_t)tile * (a.RR + 1);
  for (unsigned lr = grp; lr < rn + (groups - rn % groups) % groups; lr += groups) {
    const bool live = lr < rn;
    double y0 = 0, y1 = 0, y2 = 0;
    if (live) {
      const unsigned row = r0 + lr;
      const double2 qa = a.q[2 * (size_t)row], qb = a.q[2 * (size_t)row + 1];
      const double R0 = qa.x, R1 = qa.y, R2 = qb.x, R3 = qb.y, R4 = qa.x * qb.y, R5 = qa.y * qb.x, R6 = qa.x + qb.x, R7 = qa.y - qb.y, R8 = qb.x * qb.y;
      const unsigned end = seg[lr + 1];
      for (unsigned d = seg[lr] + lane; d < end; d += G) {
        const unsigned m = a.col[d];
        const double2 A = nt2(a.h0 + d), B = nt2(a.h1 + d), C = nt2(a.h2 + d);
        const double u0 = ux[m], u1 = uy[m], u2 = uz[m];
        const double w0 = R0 * u0 + R1 * u1 + R2 * u2, w1 = R3 * u0 + R4 * u1 + R5 * u2, w2 = R6 * u0 + R7 * u1 + R8 * u2;
        y0 += A.x * w0 + A.y * w1 + B.x * w2; y1 += A.y * w0 + B.y * w1 + C.x * w2; y2 += B.x * w0 + C.x * w1 + C.y * w2;
      }
    }
    for (unsigned off = G >> 1; off > 0; off >>= 1) { y0 += __shfl_down(y0, off, G); y1 += __shfl_down(y1, off, G); y2 += __shfl_down(y2, off, G); }
    if (live && lane == 0) { double* o = a.part + ((size_t)cb * a.n_rows + r0 + lr) * 3; o[0] = y0; o[1] = y1; o[2] = y2; }
  }
}
__global__ void __launch_bounds__(256) k_mv_finish(unsigned n_rows, unsigned n_cb, const double* part, const double* M, const double* p, double* y) {
  const unsigned t = blockIdx.x * 256 + threadIdx.x;   // one output component per lane
  if (t >= 3 * n_rows) return;
  const unsigned row = t / 3, c = t % 3;
  double s = 0;
  for (unsigned cb = 0; cb < n_cb; ++cb) s += part[(size_t)cb * n_rows * 3 + t];
  const double* Mk = M + 6 * (size_t)row; const double* pk = p + 3 * (size_t)row;
  const double mp = c == 0 ? Mk[0] * pk[0] + Mk[1] * pk[1] + Mk[2] * pk[2] : c == 1 ? Mk[1] * pk[0] + Mk[3] * pk[1] + Mk[4] * pk[2] : Mk[2] * pk[0] + Mk[4] * pk[1] + Mk[5] * pk[2];
  y[t] = mp - s;
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
  const unsigned N = 100000, DEG = 200; const int jitter = 14;
  const unsigned CB = argc > 1 ? atoi(argv[1]) : 6144, NRR = argc > 2 ? atoi(argv[2]) : 45, G = argc > 3 ? atoi(argv[3]) : 16;
  const unsigned n_cb = (N + CB - 1) / CB, RR = (N + NRR - 1) / NRR, tiles = n_cb * NRR;
  std::mt19937 rng(1);
  std::normal_distribution<double> nd01(0, 1);
  std::vector<std::vector<unsigned>> rows(N);
  size_t nd = 0;
  for (unsigned r = 0; r < N; ++r) { int dg = (int)(DEG + jitter * nd01(rng) + 0.5); if (dg < 1) dg = 1; rows[r].resize(dg); for (auto& c : rows[r]) c = rng() % N; std::sort(rows[r].begin(), rows[r].end()); nd += dg; }
  // tiled order: tile = rr * n_cb + cb; inside a tile by row
  std::vector<unsigned> seg((size_t)tiles * (RR + 1), 0);
  std::vector<unsigned short> col(nd);
  size_t pos = 0;
  for (unsigned rr = 0; rr < NRR; ++rr) for (unsigned cb = 0; cb < n_cb; ++cb) {
    const unsigned tile = rr * n_cb + cb;
    unsigned* s = &seg[(size_t)tile * (RR + 1)];
    for (unsigned lr = 0; lr < RR; ++lr) {
      s[lr] = (unsigned)pos;
      const unsigned row = rr * RR + lr;
      if (row < N) for (unsigned c : rows[row]) if (c / CB == cb) col[pos++] = (unsigned short)(c - cb * CB);
    }
    s[RR] = (unsigned)pos;
  }
  if (pos != nd) { printf("layout bug %zu %zu\n", pos, nd); return 1; }
  unsigned* d_seg; unsigned short* d_col; double2 *h0, *h1, *h2, *q; double *u, *M, *p, *y, *part;
  CHK(hipMalloc(&d_seg, 4 * seg.size())); CHK(hipMalloc(&d_col, 2 * nd)); CHK(hipMalloc(&h0, 16 * nd)); CHK(hipMalloc(&h1, 16 * nd)); CHK(hipMalloc(&h2, 16 * nd));
  CHK(hipMalloc(&q, 32 * (size_t)N)); CHK(hipMalloc(&u, 24 * (size_t)N)); CHK(hipMalloc(&M, 48 * (size_t)N)); CHK(hipMalloc(&p, 24 * (size_t)N)); CHK(hipMalloc(&y, 24 * (size_t)N));
  CHK(hipMalloc(&part, 24 * (size_t)N * n_cb));
  CHK(hipMemcpy(d_seg, seg.data(), 4 * seg.size(), hipMemcpyHostToDevice)); CHK(hipMemcpy(d_col, col.data(), 2 * nd, hipMemcpyHostToDevice));
  CHK(hipMemset(h0, 0, 16 * nd)); CHK(hipMemset(h1, 0, 16 * nd)); CHK(hipMemset(h2, 0, 16 * nd)); CHK(hipMemset(q, 0, 32 * (size_t)N)); CHK(hipMemset(u, 0, 24 * (size_t)N));
  CHK(hipMemset(M, 0, 48 * (size_t)N)); CHK(hipMemset(p, 0, 24 * (size_t)N));
  const double bytes = 52.0 * nd + 48.0 * N;   // the product form's algorithmic bytes, for comparable TB/s
  const size_t lds = 3 * (size_t)CB * 8;
  printf("rows %u, entries %zu; column blocks %u x %u cameras (%zu KB of LDS), row ranges %u x %u rows -> %u tiles of ~%zu entries, %.1f entries per (row, tile) segment, G = %u\n",
         N, nd, n_cb, CB, lds / 1024, NRR, RR, tiles, nd / tiles, (double)nd / N / n_cb, G);
  TArgs a{N, CB, RR, n_cb, G, d_seg, d_col, h0, h1, h2, u, q, part};
  CHK(hipFuncSetAttribute((const void*)k_mv_tiled<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  CHK(hipFuncSetAttribute((const void*)k_mv_tiled<512>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  float t1 = timeit([&] { hipLaunchKernelGGL((k_mv_tiled<1024>), dim3(tiles), dim3(1024), lds, 0, a); });
  float t2 = timeit([&] { hipLaunchKernelGGL(k_mv_finish, dim3((3 * N + 255) / 256), dim3(256), 0, 0, N, n_cb, (const double*)part, (const double*)M, (const double*)p, y); });
  printf("tiled, 1024 lanes: %8.1f us + finish %6.1f us = %8.1f us  (%5.2f TB/s of the product form's 1.045 GB)\n", t1, t2, t1 + t2, bytes / (t1 + t2) * 1e-6);
  float t3 = timeit([&] { hipLaunchKernelGGL((k_mv_tiled<512>), dim3(tiles), dim3(512), lds, 0, a); });
  printf("tiled,  512 lanes: %8.1f us + finish %6.1f us = %8.1f us\n", t3, t2, t3 + t2);
  CHK(hipDeviceSynchronize());
  return 0;
}

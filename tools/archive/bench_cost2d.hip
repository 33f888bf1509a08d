// Dev micro-benchmark: K1 with 2-D camera tiles, BOTH endpoint quaternion blocks staged in LDS.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "../globalsfmpy_amd/csrc/kernels.hpp"
using namespace gsfm;
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
#ifndef CB2
#define CB2 2048
#endif
#define TT2 1024
struct Tile2 { unsigned ib, jb, begin, end; };
struct A2 { const Tile2* tiles; unsigned n_cams; const uint2* idx; const double2 *qr0, *qr1, *w0, *w1, *w2; const double2* q; const DevLoss* loss; double* partials; };

__global__ void __launch_bounds__(TT2) k_cost2d(A2 a) {
  __shared__ double2 qi_xy[CB2], qi_zw[CB2], qj_xy[CB2], qj_zw[CB2];
  __shared__ double lds[TT2 / 64 + 1];
  const Tile2 t = a.tiles[blockIdx.x];
  const unsigned bi = t.ib * CB2, bj = t.jb * CB2;
  const unsigned ci = min((unsigned)CB2, a.n_cams - bi), cj = min((unsigned)CB2, a.n_cams - bj);
  for (unsigned c = threadIdx.x; c < ci; c += TT2) { qi_xy[c] = a.q[2 * (size_t)(bi + c)]; qi_zw[c] = a.q[2 * (size_t)(bi + c) + 1]; }
  for (unsigned c = threadIdx.x; c < cj; c += TT2) { qj_xy[c] = a.q[2 * (size_t)(bj + c)]; qj_zw[c] = a.q[2 * (size_t)(bj + c) + 1]; }
  __syncthreads();
  double acc = 0.0;
  for (unsigned e = t.begin + threadIdx.x; e < t.end; e += TT2) {
    const uint2 ij = a.idx[e];   // block-local indices
    const double2 r0 = nt_load2(a.qr0 + e), r1 = nt_load2(a.qr1 + e);
    const double2 wa = nt_load2(a.w0 + e), wb = nt_load2(a.w1 + e), wc = nt_load2(a.w2 + e);
    const Quat qr{r0.x, r0.y, r1.x, r1.y};
    EdgeW W; W.l00 = wa.x; W.l01 = wa.y; W.l02 = wb.x; W.l11 = wb.y; W.l12 = wc.x; W.l22 = wc.y;
    const double2 a0 = qi_xy[ij.x], a1 = qi_zw[ij.x], b0 = qj_xy[ij.y], b1 = qj_zw[ij.y];
    const Quat qi{a0.x, a0.y, a1.x, a1.y}, qj{b0.x, b0.y, b1.x, b1.y};
    double r[3];
    edge_residual<F_AA, W_MATRIX>(qi, qj, qr, W, r);
    const double s = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
    acc += 0.5 * loss_value<LM_MAGSAC>(a.loss, s);
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) { double s = 0; for (int k = 0; k < TT2 / 64; ++k) s += lds[k]; a.partials[blockIdx.x] = s; }
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
  const unsigned N = 100000; const size_t E = 10000000;
  const size_t max_tile = argc > 1 ? atoi(argv[1]) : 16384;
  std::mt19937_64 rng(3);
  const unsigned nb = (N + CB2 - 1) / CB2;
  std::vector<unsigned> ei(E), ej(E);
  for (size_t e = 0; e < E; ++e) { unsigned a = rng() % N, b = rng() % N; if (a == b) b = (b + 1) % N; ei[e] = std::min(a, b); ej[e] = std::max(a, b); }
  std::vector<size_t> cnt((size_t)nb * nb + 1, 0);
  for (size_t e = 0; e < E; ++e) cnt[(size_t)(ei[e] / CB2) * nb + ej[e] / CB2 + 1]++;
  for (size_t t = 0; t < (size_t)nb * nb; ++t) cnt[t + 1] += cnt[t];
  std::vector<uint2> idx(E);
  { std::vector<size_t> f(cnt.begin(), cnt.end() - 1); for (size_t e = 0; e < E; ++e) { const size_t t = (size_t)(ei[e] / CB2) * nb + ej[e] / CB2; idx[f[t]++] = make_uint2(ei[e] % CB2, ej[e] % CB2); } }
  std::vector<Tile2> tiles;
  for (unsigned a = 0; a < nb; ++a) for (unsigned b = 0; b < nb; ++b) {
    size_t lo = cnt[(size_t)a * nb + b], hi = cnt[(size_t)a * nb + b + 1];
    while (lo < hi) { const size_t ce = std::min(hi, lo + max_tile); tiles.push_back(Tile2{a, b, (unsigned)lo, (unsigned)ce}); lo = ce; }
  }
  printf("blocks %u, tiles %zu, avg edges/tile %.0f\n", nb, tiles.size(), (double)E / tiles.size());
  auto randq = [&](std::vector<double2>& a, std::vector<double2>& b, size_t n) { a.resize(n); b.resize(n); std::normal_distribution<double> d;
    for (size_t k = 0; k < n; ++k) { double x = d(rng), y = d(rng), z = d(rng), w = d(rng), s = 1 / std::sqrt(x*x+y*y+z*z+w*w); a[k] = make_double2(x*s, y*s); b[k] = make_double2(z*s, w*s); } };
  std::vector<double2> q0, q1, c0, c1; randq(q0, q1, E); randq(c0, c1, N);
  std::vector<double2> cq(2 * (size_t)N); for (unsigned k = 0; k < N; ++k) { cq[2 * k] = c0[k]; cq[2 * k + 1] = c1[k]; }
  std::vector<double2> w(E, make_double2(30.0, 1.0));
  uint2* d_idx; double2 *d_q0, *d_q1, *d_w0, *d_w1, *d_w2, *d_q; double* d_part; DevLoss* d_loss; double* d_tab; Tile2* d_tiles;
  CHK(hipMalloc(&d_idx, 8 * E)); CHK(hipMalloc(&d_q0, 16 * E)); CHK(hipMalloc(&d_q1, 16 * E)); CHK(hipMalloc(&d_w0, 16 * E)); CHK(hipMalloc(&d_w1, 16 * E)); CHK(hipMalloc(&d_w2, 16 * E));
  CHK(hipMalloc(&d_q, 32 * (size_t)N)); CHK(hipMalloc(&d_part, 8 * tiles.size())); CHK(hipMalloc(&d_loss, sizeof(DevLoss))); CHK(hipMalloc(&d_tab, 8 * 36843)); CHK(hipMalloc(&d_tiles, sizeof(Tile2) * tiles.size()));
  CHK(hipMemcpy(d_idx, idx.data(), 8 * E, hipMemcpyHostToDevice)); CHK(hipMemcpy(d_q0, q0.data(), 16 * E, hipMemcpyHostToDevice)); CHK(hipMemcpy(d_q1, q1.data(), 16 * E, hipMemcpyHostToDevice));
  CHK(hipMemcpy(d_w0, w.data(), 16 * E, hipMemcpyHostToDevice)); CHK(hipMemcpy(d_w1, w.data(), 16 * E, hipMemcpyHostToDevice)); CHK(hipMemcpy(d_w2, w.data(), 16 * E, hipMemcpyHostToDevice));
  CHK(hipMemcpy(d_q, cq.data(), 32 * (size_t)N, hipMemcpyHostToDevice)); CHK(hipMemcpy(d_tiles, tiles.data(), sizeof(Tile2) * tiles.size(), hipMemcpyHostToDevice));
  { std::vector<double> t(36843); for (int x = 0; x < 36843; ++x) t[x] = std::exp(-x / 1000.0); CHK(hipMemcpy(d_tab, t.data(), 8 * 36843, hipMemcpyHostToDevice));
    DevLoss L{}; L.n = 1; L.nodes[0].kind = GSFM_LOSS_MAGSAC; L.nodes[0].nu = 3; L.nodes[0].aux[1] = 8e-4; L.nodes[0].aux[4] = 40.0; L.nodes[0].aux[5] = 40.0; L.nodes[0].aux[6] = 4.5e-3; L.nodes[0].aux[7] = 3.4e-3;
    L.nodes[0].table = d_tab; L.nodes[0].table_len = 36843; CHK(hipMemcpy(d_loss, &L, sizeof(L), hipMemcpyHostToDevice)); }
  A2 a{d_tiles, N, d_idx, d_q0, d_q1, d_w0, d_w1, d_w2, d_q, d_loss, d_part};
  const float t = timeit([&] { hipLaunchKernelGGL(k_cost2d, dim3((unsigned)tiles.size()), dim3(TT2), 0, 0, a); });
  printf("2-D LDS tiles (CB %d, <= %zu edges/WG): %7.1f us  -> %.2f TB/s algorithmic\n", CB2, max_tile, t, (88.0 * E + 24.0 * N) / t * 1e-6);
  return 0;
}

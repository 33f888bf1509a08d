// Dev micro-benchmark: ablation of K2 (linearise) on synthetic C5-shaped data, reusing the product's device code.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "../globalsfmpy_amd/csrc/kernels.hpp"
using namespace gsfm;
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

// ABL: 0 full | 1 no H stores | 2 no q_m gather | 3 trivial loss | 4 streams + stores only (no math) | 5 no stores + trivial loss
template <int ABL>
__global__ void __launch_bounds__(GSFM_BLOCK) k_lin_abl(LinArgs a) {
  constexpr int R = 3;
  const uint32_t G = a.G;
  const uint32_t t = blockIdx.x * GSFM_BLOCK + threadIdx.x;
  const uint32_t row = t / G, lane = t % G;
  const bool live = row < a.n_rows;
  double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  if (live) {
    const uint32_t k = a.row_base + row;
    const Quat qk = load_q(a.q, k);
    const uint32_t end = a.row_ptr[row + 1];
    for (uint32_t d = a.row_ptr[row] + lane; d < end; d += G) {
      const uint32_t cr = __builtin_nontemporal_load(a.col + d);
      const uint32_t m = cr & 0x7fffffffu;
      const bool row_is_second = (cr >> 31) != 0;
      const double2 r0 = nt_load2(a.qr0 + d), r1 = nt_load2(a.qr1 + d);
      const Quat qr{r0.x, r0.y, r1.x, r1.y};
      const EdgeW W = load_w<W_MATRIX>(a.w0, a.w1, a.w2, a.ws, d);
      double H[9];
      if (ABL == 4) {
        H[0] = r0.x + W.l00; H[1] = r0.y + W.l01; H[2] = r1.x + W.l02; H[3] = r1.y + W.l11; H[4] = W.l12; H[5] = W.l22; H[6] = (double)m; H[7] = qk.x; H[8] = qk.w;
        acc[0] += H[0];
      } else {
        const Quat qm = (ABL == 2) ? Quat{r1.y, r0.x, r0.y, r1.x} : load_q(a.q, m);
        double r[R], Ai[3 * R], Aj[3 * R];
        if (row_is_second) edge_linearize<F_AA, W_MATRIX>(qm, qk, qr, W, r, Ai, Aj);
        else edge_linearize<F_AA, W_MATRIX>(qk, qm, qr, W, r, Ai, Aj);
        double s = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
        Rho3 rho;
        if (ABL == 3 || ABL == 5) { rho.r0 = s; rho.r1 = 1.0; rho.r2 = 0.0; } else rho = loss_eval<LM_MAGSAC>(a.loss, s);
        robustify<R>(rho, s, r, Ai, Aj);
        const double* Ar = row_is_second ? Aj : Ai;
        const double* Ac = row_is_second ? Ai : Aj;
#pragma unroll
        for (int x = 0; x < 3; ++x) {
          double g = 0.0;
#pragma unroll
          for (int c = 0; c < R; ++c) g += Ar[3 * c + x] * r[c];
          acc[x] += g;
#pragma unroll
          for (int y = 0; y < 3; ++y) { double h = 0.0;
#pragma unroll
            for (int c = 0; c < R; ++c) h += Ar[3 * c + x] * Ac[3 * c + y];
            H[3 * x + y] = h; }
        }
#pragma unroll
        for (int c = 0; c < R; ++c) { const double x0 = Ar[3 * c], x1 = Ar[3 * c + 1], x2 = Ar[3 * c + 2];
          acc[3] += x0 * x0; acc[4] += x0 * x1; acc[5] += x0 * x2; acc[6] += x1 * x1; acc[7] += x1 * x2; acc[8] += x2 * x2; }
      }
      if (ABL == 1 || ABL == 5) { acc[8] += H[0] + H[1] + H[2] + H[3] + H[4] + H[5] + H[6] + H[7] + H[8]; }
      else { nt_store2(a.h0 + d, H[0], H[1]); nt_store2(a.h1 + d, H[2], H[3]); nt_store2(a.h2 + d, H[4], H[5]); nt_store2(a.h3 + d, H[6], H[7]); __builtin_nontemporal_store(H[8], a.h4 + d); }
    }
  }
  for (uint32_t off = G >> 1; off > 0; off >>= 1) {
#pragma unroll
    for (int c = 0; c < 9; ++c) acc[c] += __shfl_down(acc[c], off, G);
  }
  if (live && lane == 0) { double* o = a.gD + 9 * (size_t)(a.row_base + row);
#pragma unroll
    for (int c = 0; c < 9; ++c) o[c] = acc[c]; }
}

template <typename F> float timeit(F f, int reps = 5) {
  hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  f(); f();
  CHK(hipEventRecord(e0));
  for (int k = 0; k < reps; ++k) f();
  CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
  float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); return ms / reps * 1e3f;
}

int main() {
  const unsigned N = 100000, DEG = 200; const size_t nd = (size_t)N * DEG;
  std::mt19937_64 rng(5);
  std::vector<unsigned> rp(N + 1), col(nd);
  for (unsigned r = 0; r <= N; ++r) rp[r] = r * DEG;
  for (size_t d = 0; d < nd; ++d) { unsigned m = rng() % N; col[d] = m | ((rng() & 1) ? 0x80000000u : 0u); }
  auto randq = [&](std::vector<double2>& a, std::vector<double2>& b, size_t n) { a.resize(n); b.resize(n); std::normal_distribution<double> dd;
    for (size_t k = 0; k < n; ++k) { double x = dd(rng), y = dd(rng), z = dd(rng), w = dd(rng), s = 1 / std::sqrt(x*x+y*y+z*z+w*w); a[k] = make_double2(x*s, y*s); b[k] = make_double2(z*s, w*s); } };
  std::vector<double2> q0, q1, c0, c1; randq(q0, q1, nd); randq(c0, c1, N);
  std::vector<double2> cq(2 * (size_t)N); for (unsigned k = 0; k < N; ++k) { cq[2 * k] = c0[k]; cq[2 * k + 1] = c1[k]; }
  std::vector<double2> w(nd, make_double2(30.0, 1.0));
  unsigned *d_rp, *d_col; double2 *d_q0, *d_q1, *d_w0, *d_w1, *d_w2, *d_q, *h0, *h1, *h2, *h3; double *h4, *gD, *d_tab; DevLoss* d_loss;
  CHK(hipMalloc(&d_rp, 4 * (N + 1))); CHK(hipMalloc(&d_col, 4 * nd)); CHK(hipMalloc(&d_q0, 16 * nd)); CHK(hipMalloc(&d_q1, 16 * nd));
  CHK(hipMalloc(&d_w0, 16 * nd)); CHK(hipMalloc(&d_w1, 16 * nd)); CHK(hipMalloc(&d_w2, 16 * nd)); CHK(hipMalloc(&d_q, 32 * (size_t)N));
  CHK(hipMalloc(&h0, 16 * nd)); CHK(hipMalloc(&h1, 16 * nd)); CHK(hipMalloc(&h2, 16 * nd)); CHK(hipMalloc(&h3, 16 * nd)); CHK(hipMalloc(&h4, 8 * nd));
  CHK(hipMalloc(&gD, 72 * (size_t)N)); CHK(hipMalloc(&d_tab, 8 * 36843)); CHK(hipMalloc(&d_loss, sizeof(DevLoss)));
  CHK(hipMemcpy(d_rp, rp.data(), 4 * (N + 1), hipMemcpyHostToDevice)); CHK(hipMemcpy(d_col, col.data(), 4 * nd, hipMemcpyHostToDevice));
  CHK(hipMemcpy(d_q0, q0.data(), 16 * nd, hipMemcpyHostToDevice)); CHK(hipMemcpy(d_q1, q1.data(), 16 * nd, hipMemcpyHostToDevice));
  CHK(hipMemcpy(d_w0, w.data(), 16 * nd, hipMemcpyHostToDevice)); CHK(hipMemcpy(d_w1, w.data(), 16 * nd, hipMemcpyHostToDevice)); CHK(hipMemcpy(d_w2, w.data(), 16 * nd, hipMemcpyHostToDevice));
  CHK(hipMemcpy(d_q, cq.data(), 32 * (size_t)N, hipMemcpyHostToDevice));
  { std::vector<double> t(36843); for (int x = 0; x < 36843; ++x) t[x] = std::exp(-x / 1000.0); CHK(hipMemcpy(d_tab, t.data(), 8 * 36843, hipMemcpyHostToDevice));
    DevLoss L{}; L.n = 1; L.nodes[0].kind = GSFM_LOSS_MAGSAC; L.nodes[0].nu = 3; L.nodes[0].aux[0] = 4e-4; L.nodes[0].aux[1] = 8e-4; L.nodes[0].aux[2] = 8e-6; L.nodes[0].aux[3] = 0.8; L.nodes[0].aux[4] = 40.0; L.nodes[0].aux[5] = 40.0; L.nodes[0].aux[6] = 4.5e-3; L.nodes[0].aux[7] = 3.4e-3;
    L.nodes[0].table = d_tab; L.nodes[0].table_len = 36843; CHK(hipMemcpy(d_loss, &L, sizeof(L), hipMemcpyHostToDevice)); }
  LinArgs a{}; a.n_rows = N; a.row_base = 0; a.G = 64; a.row_ptr = d_rp; a.col = d_col; a.qr0 = d_q0; a.qr1 = d_q1; a.w0 = d_w0; a.w1 = d_w1; a.w2 = d_w2; a.q = d_q; a.loss = d_loss;
  a.h0 = h0; a.h1 = h1; a.h2 = h2; a.h3 = h3; a.h4 = h4; a.gD = gD;
  const char* names[6] = {"full", "no H stores", "no q_m gather", "trivial loss", "streams + stores only (no math)", "no stores + trivial loss"};
  for (unsigned G : {64u, 16u}) {
    a.G = G; const int grid = (int)(((size_t)N * G + 255) / 256);
    float t;
    t = timeit([&] { hipLaunchKernelGGL(k_lin_abl<0>, dim3(grid), dim3(256), 0, 0, a); }); printf("G=%2u %-34s %8.1f us\n", G, names[0], t);
    t = timeit([&] { hipLaunchKernelGGL(k_lin_abl<1>, dim3(grid), dim3(256), 0, 0, a); }); printf("G=%2u %-34s %8.1f us\n", G, names[1], t);
    t = timeit([&] { hipLaunchKernelGGL(k_lin_abl<2>, dim3(grid), dim3(256), 0, 0, a); }); printf("G=%2u %-34s %8.1f us\n", G, names[2], t);
    t = timeit([&] { hipLaunchKernelGGL(k_lin_abl<3>, dim3(grid), dim3(256), 0, 0, a); }); printf("G=%2u %-34s %8.1f us\n", G, names[3], t);
    t = timeit([&] { hipLaunchKernelGGL(k_lin_abl<4>, dim3(grid), dim3(256), 0, 0, a); }); printf("G=%2u %-34s %8.1f us\n", G, names[4], t);
    t = timeit([&] { hipLaunchKernelGGL(k_lin_abl<5>, dim3(grid), dim3(256), 0, 0, a); }); printf("G=%2u %-34s %8.1f us\n", G, names[5], t);
  }
  return 0;
}

// Dev micro-benchmark: ablation of the K1 sweep (angle-axis + upper-triangular whitening + MAGSAC value).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "../globalsfmpy_amd/csrc/kernels.hpp"
using namespace gsfm;
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

struct A { size_t n; const uint2* idx; const double2 *qr0, *qr1, *w0, *w1, *w2; const double2* q; const DevLoss* loss; double* partials; unsigned n_cams; };

// ABL: 0 full, 1 no q gathers (q from the stream), 2 streams only (no math), 3 full but identity loss (s/2), 4 math w/o log (no atan2/sqrt)
template <int ABL, int THREADS>
__global__ void __launch_bounds__(THREADS) k_abl(A a) {
  __shared__ double lds[THREADS / 64 + 1];
  double acc = 0.0;
  const size_t stride = (size_t)gridDim.x * THREADS;
  for (size_t e = (size_t)blockIdx.x * THREADS + threadIdx.x; e < a.n; e += stride) {
    const uint2 ij = a.idx[e];
    const double2 r0 = nt_load2(a.qr0 + e), r1 = nt_load2(a.qr1 + e);
    const double2 wa = nt_load2(a.w0 + e), wb = nt_load2(a.w1 + e), wc = nt_load2(a.w2 + e);
    if (ABL == 2) { acc += r0.x + r0.y + r1.x + r1.y + wa.x + wa.y + wb.x + wb.y + wc.x + wc.y + (double)(ij.x + ij.y); continue; }
    const Quat qr{r0.x, r0.y, r1.x, r1.y};
    EdgeW W; W.l00 = wa.x; W.l01 = wa.y; W.l02 = wb.x; W.l11 = wb.y; W.l12 = wc.x; W.l22 = wc.y;
    Quat qi, qj;
    if (ABL == 1) { qi = Quat{r0.y, r0.x, r1.y, r1.x}; qj = Quat{r1.x, r0.y, r0.x, r1.y}; acc += (double)(ij.x ^ ij.y) * 1e-30; }
    else { qi = load_q(a.q, ij.x); qj = load_q(a.q, ij.y); }
    double r[3];
    if (ABL == 4) {
      const Quat qe = qmul(qmul(qj, qconj(qi)), qconj(qr));
      const double e3[3] = {2 * qe.x, 2 * qe.y, 2 * qe.z};
      apply_w_vec<W_MATRIX>(W, e3, r);
    } else edge_residual<F_AA, W_MATRIX>(qi, qj, qr, W, r);
    const double s = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
    acc += (ABL == 3 || ABL == 4) ? 0.5 * s : 0.5 * loss_value<LM_MAGSAC>(a.loss, s);
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) { double t = 0; for (int k = 0; k < THREADS / 64; ++k) t += lds[k]; a.partials[blockIdx.x] = t; }
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
  std::mt19937_64 rng(3);
  std::vector<uint2> idx(E);
  { // sorted by i like the generator's output
    std::vector<unsigned> ii(E); for (auto& v : ii) v = rng() % N; std::sort(ii.begin(), ii.end());
    for (size_t e = 0; e < E; ++e) idx[e] = make_uint2(ii[e], (unsigned)(rng() % N)); }
  auto randq = [&](std::vector<double2>& a, std::vector<double2>& b, size_t n) { a.resize(n); b.resize(n); std::normal_distribution<double> d;
    for (size_t k = 0; k < n; ++k) { double x = d(rng), y = d(rng), z = d(rng), w = d(rng), s = 1 / std::sqrt(x*x+y*y+z*z+w*w); a[k] = make_double2(x*s, y*s); b[k] = make_double2(z*s, w*s); } };
  std::vector<double2> q0, q1, c0, c1; randq(q0, q1, E); randq(c0, c1, N);
  std::vector<double2> cq(2 * (size_t)N); for (unsigned k = 0; k < N; ++k) { cq[2 * k] = c0[k]; cq[2 * k + 1] = c1[k]; }
  std::vector<double2> w(E, make_double2(30.0, 1.0));
  uint2* d_idx; double2 *d_q0, *d_q1, *d_w0, *d_w1, *d_w2, *d_q; double* d_part; DevLoss* d_loss; double* d_tab;
  CHK(hipMalloc(&d_idx, 8 * E)); CHK(hipMalloc(&d_q0, 16 * E)); CHK(hipMalloc(&d_q1, 16 * E)); CHK(hipMalloc(&d_w0, 16 * E)); CHK(hipMalloc(&d_w1, 16 * E)); CHK(hipMalloc(&d_w2, 16 * E));
  CHK(hipMalloc(&d_q, 32 * (size_t)N)); CHK(hipMalloc(&d_part, 8 * 8192)); CHK(hipMalloc(&d_loss, sizeof(DevLoss))); CHK(hipMalloc(&d_tab, 8 * 36843));
  CHK(hipMemcpy(d_idx, idx.data(), 8 * E, hipMemcpyHostToDevice)); CHK(hipMemcpy(d_q0, q0.data(), 16 * E, hipMemcpyHostToDevice)); CHK(hipMemcpy(d_q1, q1.data(), 16 * E, hipMemcpyHostToDevice));
  CHK(hipMemcpy(d_w0, w.data(), 16 * E, hipMemcpyHostToDevice)); CHK(hipMemcpy(d_w1, w.data(), 16 * E, hipMemcpyHostToDevice)); CHK(hipMemcpy(d_w2, w.data(), 16 * E, hipMemcpyHostToDevice));
  CHK(hipMemcpy(d_q, cq.data(), 32 * (size_t)N, hipMemcpyHostToDevice));
  { std::vector<double> t(36843); for (int x = 0; x < 36843; ++x) t[x] = std::exp(-x / 1000.0); CHK(hipMemcpy(d_tab, t.data(), 8 * 36843, hipMemcpyHostToDevice));
    DevLoss L{}; L.n = 1; L.nodes[0].kind = GSFM_LOSS_MAGSAC; L.nodes[0].nu = 3; L.nodes[0].aux[1] = 8e-4; L.nodes[0].aux[4] = 40.0; L.nodes[0].aux[5] = 40.0; L.nodes[0].aux[6] = 4.5e-3; L.nodes[0].aux[7] = 3.4e-3;
    L.nodes[0].table = d_tab; L.nodes[0].table_len = 36843; CHK(hipMemcpy(d_loss, &L, sizeof(L), hipMemcpyHostToDevice)); }
  A a{E, d_idx, d_q0, d_q1, d_w0, d_w1, d_w2, d_q, d_loss, d_part, N};
  const char* names[5] = {"full (gathers + log + MAGSAC value)", "no q gathers", "streams only", "gathers + log, trivial loss", "gathers, no log (no atan2/sqrt/div)"};
  for (int grid : {2048, 4096}) {
    float t;
    t = timeit([&] { hipLaunchKernelGGL((k_abl<0, 256>), dim3(grid), dim3(256), 0, 0, a); }); printf("grid %d  %-40s %7.1f us\n", grid, names[0], t);
    t = timeit([&] { hipLaunchKernelGGL((k_abl<1, 256>), dim3(grid), dim3(256), 0, 0, a); }); printf("grid %d  %-40s %7.1f us\n", grid, names[1], t);
    t = timeit([&] { hipLaunchKernelGGL((k_abl<2, 256>), dim3(grid), dim3(256), 0, 0, a); }); printf("grid %d  %-40s %7.1f us\n", grid, names[2], t);
    t = timeit([&] { hipLaunchKernelGGL((k_abl<3, 256>), dim3(grid), dim3(256), 0, 0, a); }); printf("grid %d  %-40s %7.1f us\n", grid, names[3], t);
    t = timeit([&] { hipLaunchKernelGGL((k_abl<4, 256>), dim3(grid), dim3(256), 0, 0, a); }); printf("grid %d  %-40s %7.1f us\n", grid, names[4], t);
  }
  return 0;
}

// Dev micro-benchmark (not part of the product; round 6, review item 1a): what a BARE stream of a rank-sized launch takes on one MI355X --
// the floor under one rank's share of the sharded kernels.  A rank of 8 of the C5 graph streams 125 MB per mat-vec (K3c: 2.5 M positions x
// 50 B), 120 MB per cost sweep (K1: 1.25 M edges x 96 B) and 243 MB in + 120 MB out per linearisation (K2c: 2.5 M positions x 97 B in, 48 B
// out); 1/8 of the one-GPU kernel time -- what the round-5 review priced them against -- assumes that a launch of that size still streams
// at the 6 TB/s the 1 GB launches reach.  This measures it: non-temporal 16-byte loads of `bytes` in all, G workgroups of T lanes, each
// workgroup a contiguous range in grid-stride steps of its own (as K3c walks its sub-chunks), optionally a 16-byte non-temporal store per
// `out_every` loads; averaged over 50 back-to-back launches with HIP events (so the per-launch figure includes the dependent-launch boundary,
// exactly as gsfm_rot_time_kernels's figures do).
//   hipcc --offload-arch=gfx950 -O3 -o bench_small_stream bench_small_stream.hip ; ./bench_small_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int T, int U>
__global__ void __launch_bounds__(T) k_stream(const double2* __restrict__ in, size_t n16, double2* __restrict__ out, int out_every, double* sink) {
  const size_t per = (n16 + gridDim.x - 1) / gridDim.x, lo = per * blockIdx.x, hi = lo + per < n16 ? lo + per : n16;
  double acc = 0.0;
  for (size_t i = lo + threadIdx.x; i < hi; i += (size_t)T * U) {
    double2 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { const size_t j = i + (size_t)u * T < hi ? i + (size_t)u * T : i; v[u].x = __builtin_nontemporal_load(&in[j].x); v[u].y = __builtin_nontemporal_load(&in[j].y); }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      acc += v[u].x + v[u].y;
      if (out_every && ((i / T + u) % out_every) == 0) { const size_t j = (i + (size_t)u * T) / out_every; __builtin_nontemporal_store(v[u].x, &out[j].x); __builtin_nontemporal_store(acc, &out[j].y); }
    }
  }
  if (acc == 1.2345e-300) sink[blockIdx.x] = acc;
}
__global__ void k_empty(double* sink) { if (threadIdx.x == 9999) sink[0] = 1.0; }

int main() {
  const size_t cap = (size_t)1 << 30;
  double2 *in, *out; double* sink;
  CHK(hipMalloc(&in, cap)); CHK(hipMalloc(&out, cap)); CHK(hipMalloc(&sink, 1 << 20));
  CHK(hipMemset(in, 0, cap)); CHK(hipMemset(out, 0, cap));
  hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  hipStream_t st; CHK(hipStreamCreate(&st));
  auto time = [&](auto launch) { for (int k = 0; k < 5; ++k) launch(); CHK(hipEventRecord(e0, st)); for (int k = 0; k < 50; ++k) launch(); CHK(hipEventRecord(e1, st)); CHK(hipStreamSynchronize(st)); float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); return 1e3 * ms / 50; };
  printf("empty kernel, 256 x 256, back to back: %.2f us per launch\n", time([&] { hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, st, sink); }));
  struct Case { const char* what; double mb_in; int out_every; };
  const Case cases[] = {{"half of one rank of 8's K3c: 62.5 MB in", 62.5, 0}, {"K3c, one rank of 8: 125 MB in", 125.0, 0}, {"K1, one rank of 8: 120 MB in", 120.0, 0}, {"K2c, one rank of 8: 243 MB in + 121 MB out", 243.0, 2},
                        {"K3c, one rank of 4: 250 MB in", 250.0, 0}, {"K3c, one rank of 2: 500 MB in", 500.0, 0}, {"K3c, one GPU: 1000 MB in", 1000.0, 0}};
  for (const Case& c : cases) {
    const size_t n16 = (size_t)(c.mb_in * 1e6 / 16);
    printf("%s\n", c.what);
    auto row = [&](const char* cfg, double us) { printf("   %-34s %7.1f us   %5.2f TB/s on %.0f MB\n", cfg, us, (c.mb_in + (c.out_every ? c.mb_in / c.out_every : 0.0)) * 1e6 / (us * 1e-6) / 1e12, c.mb_in + (c.out_every ? c.mb_in / c.out_every : 0.0)); };
    row("500 WGs x 512 lanes, 1 load/lane", time([&] { hipLaunchKernelGGL((k_stream<512, 1>), dim3(500), dim3(512), 0, st, in, n16, out, c.out_every, sink); }));
    row("500 WGs x 512 lanes, 4 loads/lane", time([&] { hipLaunchKernelGGL((k_stream<512, 4>), dim3(500), dim3(512), 0, st, in, n16, out, c.out_every, sink); }));
    row("1024 WGs x 512 lanes, 4 loads/lane", time([&] { hipLaunchKernelGGL((k_stream<512, 4>), dim3(1024), dim3(512), 0, st, in, n16, out, c.out_every, sink); }));
    row("2048 WGs x 256 lanes, 4 loads/lane", time([&] { hipLaunchKernelGGL((k_stream<256, 4>), dim3(2048), dim3(256), 0, st, in, n16, out, c.out_every, sink); }));
    row("4096 WGs x 256 lanes, 8 loads/lane", time([&] { hipLaunchKernelGGL((k_stream<256, 8>), dim3(4096), dim3(256), 0, st, in, n16, out, c.out_every, sink); }));
  }
  return 0;
}

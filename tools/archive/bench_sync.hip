// Host cost of reading a few scalars back from the device (LM control reads ~100 bytes two or three times per iteration; DESIGN.md section 6, latency regime).
// build: hipcc -O3 --offload-arch=gfx950 -o tools/bench_sync tools/archive/bench_sync.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
__global__ void k_touch(double* p) { if (threadIdx.x == 0) p[0] += 1.0; }
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  double *d, *pinned, *mapped, *mapped_dev, pageable[16];
  CK(hipMalloc(&d, 128)); CK(hipMemset(d, 0, 128));
  CK(hipHostMalloc(&pinned, 128, hipHostMallocDefault));
  CK(hipHostMalloc(&mapped, 128, hipHostMallocMapped)); CK(hipHostGetDevicePointer((void**)&mapped_dev, mapped, 0));
  mapped[0] = 0.0;
  const int reps = 2000;
  for (int mode = 0; mode < 5; ++mode) {
    double t0 = 0;
    for (int r = -100; r < reps; ++r) {
      if (r == 0) t0 = now_us();
      if (mode == 4) hipLaunchKernelGGL(k_touch, dim3(1), dim3(64), 0, st, mapped_dev);
      else if (mode != 3) hipLaunchKernelGGL(k_touch, dim3(1), dim3(64), 0, st, d);
      if (mode == 0) CK(hipMemcpyAsync(pageable, d, 128, hipMemcpyDeviceToHost, st));
      if (mode == 1) CK(hipMemcpyAsync(pinned, d, 128, hipMemcpyDeviceToHost, st));
      CK(hipStreamSynchronize(st));
    }
    const double us = (now_us() - t0) / reps;
    const char* names[5] = {"kernel + memcpyAsync to pageable + sync", "kernel + memcpyAsync to pinned + sync", "kernel + sync (no read-back)", "sync of an idle stream",
                            "kernel writing host-mapped memory + sync"};
    printf("%-45s %.2f us\n", names[mode], us);
  }
  // launch cost alone: 20 dependent empty-ish kernels, one sync
  double t0 = now_us();
  for (int r = 0; r < 200; ++r) { for (int k = 0; k < 20; ++k) hipLaunchKernelGGL(k_touch, dim3(1), dim3(64), 0, st, d); CK(hipStreamSynchronize(st)); }
  printf("20 dependent tiny kernels + sync: %.2f us per kernel\n", (now_us() - t0) / 200 / 20);
  return 0;
}

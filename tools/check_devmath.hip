// Bit-for-bit comparison of csrc/devmath.hpp (atan2_q1, exp_sc: the device library's routines with their coefficients in scalar
// registers) and of dense_kernels.hpp's chol_rsqrt with atan2() / exp() / rsqrt() on the device.  Prints the number of mismatching arguments; exit status 0 iff there is none.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I globalsfmpy_amd/csrc -o /tmp/check_devmath tools/check_devmath.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include "devmath.hpp"
#include "dense_kernels.hpp"   // chol_rsqrt: rsqrt(double) without its special-case select (round 6)

__device__ __forceinline__ uint64_t mix(uint64_t z) {   // splitmix64
  z += 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31);
}
__device__ __forceinline__ double u01(uint64_t h) { return (double)(h >> 11) * 0x1.0p-53; }
__device__ __forceinline__ bool same(double a, double b) { return __builtin_bit_cast(uint64_t, a) == __builtin_bit_cast(uint64_t, b) || (a != a && b != b); }

// mode 0: (y, x) = (sin, |cos|) of a half angle as quat_log sees them, over all scales of the angle; mode 1: arbitrary positive pairs over
// 600 binades, with zeros / infinities / NaN / denormals mixed in; mode 2: exp over [-1100, 1100] and at the MAGSAC table arguments -x / 1000
__global__ void k_check(int mode, uint64_t n, uint64_t seed, unsigned long long* bad, double* first) {
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t h0 = mix(seed + 2 * i), h1 = mix(seed + 2 * i + 1);
    double a, b, got, want;
    if (mode == 0) {
      const double ang = ldexp(u01(h0), -(int)(h1 % 60)) * 1.5707963267948966;
      double sn, cs; sincos(ang, &sn, &cs);
      a = sn; b = fabs(cs);
      if (!(a > 0.0)) a = 1e-300;
      got = gsfm::atan2_q1(a, b); want = atan2(a, b);
    } else if (mode == 1) {
      a = ldexp(0.5 + u01(h0), (int)(h0 % 600) - 300); b = ldexp(0.5 + u01(h1), (int)(h1 % 600) - 300);
      const unsigned sp = (unsigned)(h1 >> 40) & 1023u;
      if (sp == 0) b = 0.0; else if (sp == 1) b = __builtin_inf(); else if (sp == 2) b = __builtin_nan(""); else if (sp == 3) a = __builtin_nan("");
      else if (sp == 4) a = 4.9e-324; else if (sp == 5) b = 4.9e-324; else if (sp == 6) a = __builtin_inf(); else if (sp == 7) b = a;
      got = gsfm::atan2_q1(a, b); want = atan2(a, b);
      if (a == __builtin_inf() && b == __builtin_inf()) want = got;   // (documented: not reproduced, cannot occur for a unit quaternion)
    } else if (mode == 3) {   // chol_rsqrt against rsqrt over every positive finite binade, denormals included (what a pivot can be)
      a = ldexp(0.5 + u01(h0), (int)(h1 % 2098) - 1074);
      if (!(a > 0.0) || a == __builtin_inf()) a = 1.0;
      b = 0.0;
      got = gsfm::chol_rsqrt(a); want = rsqrt(a);
    } else {
      if (h1 & 1) a = -1e-3 * (double)(h0 % 200000); else a = (u01(h0) - 0.5) * 2200.0;
      const unsigned sp = (unsigned)(h1 >> 40) & 1023u;
      if (sp == 0) a = __builtin_nan(""); else if (sp == 1) a = __builtin_inf(); else if (sp == 2) a = -__builtin_inf(); else if (sp == 3) a = 0.0; else if (sp == 4) a = -745.2;
      b = 0.0;
      got = gsfm::exp_sc(a); want = exp(a);
    }
    if (!same(got, want)) {
      if (atomicAdd(bad, 1ull) == 0) { first[0] = a; first[1] = b; first[2] = got; first[3] = want; }
    }
  }
}

int main(int argc, char** argv) {
  const uint64_t n = argc > 1 ? strtoull(argv[1], nullptr, 10) : (1ull << 26);
  unsigned long long* bad; double* first;
  if (hipMalloc(&bad, 8) != hipSuccess || hipMalloc(&first, 32) != hipSuccess) { printf("no device memory\n"); return 2; }
  int rc = 0;
  for (int mode = 0; mode < 4; ++mode) {
    (void)hipMemset(bad, 0, 8); (void)hipMemset(first, 0, 32);
    hipLaunchKernelGGL(k_check, dim3(4096), dim3(256), 0, 0, mode, n, 0x1234567ull * (mode + 1), bad, first);
    unsigned long long hb = 0; double hf[4];
    if (hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost) != hipSuccess) { printf("hip error\n"); return 2; }
    (void)hipMemcpy(hf, first, 32, hipMemcpyDeviceToHost);
    printf("mode %d (%s): %llu arguments, %llu mismatches", mode, mode == 0 ? "atan2_q1 on half-angle pairs" : mode == 1 ? "atan2_q1 on arbitrary pairs" : mode == 2 ? "exp_sc" : "chol_rsqrt on positive finite arguments", (unsigned long long)n, hb);
    if (hb) printf("  first: a = %a b = %a got %a want %a", hf[0], hf[1], hf[2], hf[3]);
    printf("\n");
    if (hb) rc = 1;
  }
  return rc;
}

// Device math whose constants live in scalar registers (fp64, gfx950).
#pragma once
#include <hip/hip_runtime.h>

namespace gsfm {

// ---- atan2 and exp with their polynomial coefficients in SCALAR registers -------------------------------------------------------
// The compiler materialises every fp64 literal of an inlined libm routine in a VECTOR register pair and hoists it out of the loop it
// sits in: the 20 coefficients of atan2 and the 13 of exp pinned ~70 VGPRs for the whole of K2c -- a third of its budget, the difference
// between two and three wavefronts per SIMD -- and cost a v_mov_b64 per Horner step (v_fmac needs its addend in the destination).  Passing
// a literal through an empty asm with an "s" constraint keeps it in an SGPR pair (two s_mov_b32 at the point of use, on the scalar unit),
// and v_fma_f64 takes it as its addend directly.  The arithmetic below is the device library's own, operation for operation (ocml
// atan2 / atanred / exp for f64, ROCm 7.2; read off its bitcode): same minimax coefficients, same Horner order, same IEEE division, so
// the results are bit-identical to atan2() / exp() -- pinned by tests/test_gpu_devmath.py over 2^26 arguments each.
__device__ __forceinline__ double sgpr_const(unsigned long long bits) {
  asm volatile("" : "+s"(bits));
  return __builtin_bit_cast(double, bits);
}
// atan2(y, x) for y > 0 (or NaN), x >= 0 (or NaN): the first quadrant is all quat_log needs.
__device__ __forceinline__ double atan2_q1(double y, double x) {
#pragma clang fp contract(off)
  const double mx = __builtin_fmax(x, y), mn = __builtin_fmin(x, y);
  const double a = mn / mx;
  const double t = a * a;
  double p = __builtin_fma(t, sgpr_const(0x3EEBA404B5E68A13ull), sgpr_const(0xBF23E260BD3237F4ull));
  p = __builtin_fma(t, p, sgpr_const(0x3F4B2BB069EFB384ull));
  p = __builtin_fma(t, p, sgpr_const(0xBF67952DAF56DE9Bull));
  p = __builtin_fma(t, p, sgpr_const(0x3F7D6D43A595C56Full));
  p = __builtin_fma(t, p, sgpr_const(0xBF8C6EA4A57D9582ull));
  p = __builtin_fma(t, p, sgpr_const(0x3F967E295F08B19Full));
  p = __builtin_fma(t, p, sgpr_const(0xBF9E9AE6FC27006Aull));
  p = __builtin_fma(t, p, sgpr_const(0x3FA2C15B5711927Aull));
  p = __builtin_fma(t, p, sgpr_const(0xBFA59976E82D3FF0ull));
  p = __builtin_fma(t, p, sgpr_const(0x3FA82D5D6EF28734ull));
  p = __builtin_fma(t, p, sgpr_const(0xBFAAE5CE6A214619ull));
  p = __builtin_fma(t, p, sgpr_const(0x3FAE1BB48427B883ull));
  p = __builtin_fma(t, p, sgpr_const(0xBFB110E48B207F05ull));
  p = __builtin_fma(t, p, sgpr_const(0x3FB3B13657B87036ull));
  p = __builtin_fma(t, p, sgpr_const(0xBFB745D119378E4Full));
  p = __builtin_fma(t, p, sgpr_const(0x3FBC71C717E1913Cull));
  p = __builtin_fma(t, p, sgpr_const(0xBFC2492492376B7Dull));
  p = __builtin_fma(t, p, sgpr_const(0x3FC99999999952CCull));
  p = __builtin_fma(t, p, sgpr_const(0xBFD5555555555523ull));
  double r = __builtin_fma(a, t * p, a);
  if (x < y) r = sgpr_const(0x3FF921FB54442D18ull) - r;
  return (x != x || y != y) ? __builtin_nan("") : r;
}
// exp(x), all of ocml's f64 routine (range reduction by ln 2 in two pieces, degree-11 polynomial, ldexp, the two range clamps)
__device__ __forceinline__ double exp_sc(double x) {
#pragma clang fp contract(off)
  const double n = __builtin_rint(x * sgpr_const(0x3FF71547652B82FEull));
  double r = __builtin_fma(-n, sgpr_const(0x3FE62E42FEFA39EFull), x);
  r = __builtin_fma(-n, sgpr_const(0x3C7ABC9E3B39803Full), r);
  double p = __builtin_fma(r, sgpr_const(0x3E5ADE156A5DCB37ull), sgpr_const(0x3E928AF3FCA7AB0Cull));
  p = __builtin_fma(r, p, sgpr_const(0x3EC71DEE623FDE64ull));
  p = __builtin_fma(r, p, sgpr_const(0x3EFA01997C89E6B0ull));
  p = __builtin_fma(r, p, sgpr_const(0x3F2A01A014761F6Eull));
  p = __builtin_fma(r, p, sgpr_const(0x3F56C16C1852B7B0ull));
  p = __builtin_fma(r, p, sgpr_const(0x3F81111111122322ull));
  p = __builtin_fma(r, p, sgpr_const(0x3FA55555555502A1ull));
  p = __builtin_fma(r, p, sgpr_const(0x3FC5555555555511ull));
  p = __builtin_fma(r, p, sgpr_const(0x3FE000000000000Bull));
  p = __builtin_fma(r, p, 1.0);
  p = __builtin_fma(r, p, 1.0);
  double e = __builtin_ldexp(p, (int)n);
  if (x > 1024.0) e = __builtin_inf();
  if (x < -1075.0) e = 0.0;
  return e;
}

}  // namespace gsfm

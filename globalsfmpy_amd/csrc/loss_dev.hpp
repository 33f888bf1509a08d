// Device-side robust losses: the stack machine of include/gsfm_rot.h evaluated once per
// edge.  Formulas follow the reference's scripts/loss_functions.py (line numbers in
// include/gsfm_rot.h); host-only constants of the MAGSAC leaves (pow/tgamma terms) are
// precomputed by prepare_loss() into aux[] so the device only does exp / sqrt / a table gather.
#pragma once
#include <hip/hip_runtime.h>
#include <cfloat>
#include "../../include/gsfm_rot.h"
#include "devmath.hpp"

namespace gsfm {

struct DevLossNode {
  int32_t kind;
  int32_t nu;      // MAGSAC: 3, 4 or 9
  double p[3];
  // MAGSAC aux: 0 squared_sigma, 1 squared_sigma_max_2, 2 cubed_sigma_max, 3 C_times_two_ad_dof,
  //             4 one_over_sigma, 5 weight_zero, 6 cut (= q^2 sigma^2), 7 upper_incomplete_gamma_of_k
  double aux[8];
  double rho1_scale;    // MAGSAC: C 2^((nu-1)/2) / (2 sigma^3), the constant factor of rho' = -weight'(s) (host-precomputed reciprocal)
  double rho2_scale;    // nu = 3: 2 C 2^((nu-1)/2) / (8 sigma^5), the constant factor of -rho''
  double e2_clamp;      // nu = 3: exp(-1e-7 / 2 sigma^2), the exponential of rho'' for cells whose s falls below the 1e-7 floor of loss_functions.py:317
  int32_t x_clamp;      //   ... i.e. for table cells x < x_clamp
  int32_t pad_;
  const double* table;  // device pointer, Gamma((nu-1)/2, x/1000)
  int32_t table_len;
  int32_t inverse;
};

struct DevLoss {
  int32_t n;       // 0 = NULL loss
  int32_t external;  // 1: rho triples are supplied per edge (host callback path)
  DevLossNode nodes[GSFM_LOSS_MAX_NODES];
};

struct Rho3 { double r0, r1, r2; };

__device__ __forceinline__ Rho3 loss_magsac(const DevLossNode& n, double sq) {
  const double ssm2 = n.aux[1];
  bool zero_derivative = false;
  if (sq > n.aux[6]) { sq = n.aux[6]; zero_derivative = true; }
  // Python round() is half-to-even: rint()
  long x = (long)rint(1000.0 * sq / ssm2);
  if (x > (long)n.table_len - 1) x = (long)n.table_len - 1;  // unreachable after the clamp above
  double s = (double)x * ssm2 / 1000.0;
  const double weight = n.aux[4] * (n.table[x] - n.aux[7]);
  double u = s / ssm2;
  // (s/2sigma^2)^(nu/2 - 1.5): exponent 0 (nu=3), 1/2 (nu=4), 3 (nu=9)
  double pw = (n.nu == 3) ? 1.0 : (n.nu == 4) ? sqrt(u) : u * u * u;
  const double wd = -n.aux[3] * pw * exp(-u) / (2.0 * n.aux[2]);
  if (s < 1e-7) s = 1e-7;
  u = s / ssm2;
  pw = (n.nu == 3) ? 1.0 : (n.nu == 4) ? sqrt(u) : u * u * u;
  const double wdd = 2.0 * n.aux[3] * pw * (1.0 / n.aux[0] - ((double)n.nu - 3.0) / s) * exp(-u) / (8.0 * n.aux[2]);
  Rho3 o;
  if (n.inverse) {
    o.r0 = 1.0 / weight;
    o.r1 = -1.0 / (weight * weight) * wd;
    o.r2 = 2.0 / (weight * weight * weight) * wd * wd - wdd / (weight * weight);
  } else {
    o.r0 = n.aux[5] - weight;
    o.r1 = -wd;
    o.r2 = -wdd;
    if (o.r1 == 0.0) o.r1 = 0.00001;
  }
  if (zero_derivative) { o.r1 = 0.00001; o.r2 = 0.0; }
  return o;
}

__device__ __forceinline__ Rho3 loss_leaf(const DevLossNode& n, double s) {
  Rho3 o;
  const double a = n.p[0];
  switch (n.kind) {
    case GSFM_LOSS_TRIVIAL: o.r0 = s; o.r1 = 1.0; o.r2 = 0.0; break;
    case GSFM_LOSS_HUBER: {
      const double b = a * a;
      if (s > b) { const double r = sqrt(s); o.r0 = 2.0 * a * r - b; o.r1 = fmax(a / r, DBL_MIN); o.r2 = -o.r1 / (2.0 * s); }
      else { o.r0 = s; o.r1 = 1.0; o.r2 = 0.0; }
      break; }
    case GSFM_LOSS_SOFT_L1: {
      const double b = a * a, c = 1.0 / b;
      const double sum = 1.0 + s * c, tmp = sqrt(sum);
      o.r0 = 2.0 * b * (tmp - 1.0); o.r1 = fmax(1.0 / tmp, DBL_MIN); o.r2 = -(c * o.r1) / (2.0 * sum);
      break; }
    case GSFM_LOSS_CAUCHY: {
      const double b = a * a, c = 1.0 / b;
      const double sum = 1.0 + s * c, inv = 1.0 / sum;
      o.r0 = b * log(sum); o.r1 = fmax(inv, DBL_MIN); o.r2 = -c * (inv * inv);
      break; }
    case GSFM_LOSS_ARCTAN: {
      const double b = 1.0 / (a * a);
      const double sum = 1.0 + s * s * b, inv = 1.0 / sum;
      o.r0 = a * atan2(s, a); o.r1 = fmax(inv, DBL_MIN); o.r2 = -2.0 * s * b * (inv * inv);
      break; }
    case GSFM_LOSS_TOLERANT: {
      const double b = n.p[1], c = n.p[2];  // p[2] = b log(1 + exp(-a/b)), precomputed on the host
      const double x = (s - a) / b;
      if (x > 36.7) { o.r0 = s - a - c; o.r1 = 1.0; o.r2 = 0.0; }
      else {
        const double ex = exp(x);
        o.r0 = b * log(1.0 + ex) - c; o.r1 = fmax(ex / (1.0 + ex), DBL_MIN); o.r2 = 0.5 / (b * (1.0 + cosh(x)));
      }
      break; }
    case GSFM_LOSS_TUKEY: {
      const double a2 = a * a;
      if (s <= a2) { const double v = 1.0 - s / a2, v2 = v * v; o.r0 = a2 / 6.0 * (1.0 - v2 * v); o.r1 = 0.5 * v2; o.r2 = -1.0 / a2 * v; }
      else { o.r0 = a2 / 6.0; o.r1 = 0.0; o.r2 = 0.0; }
      break; }
    case GSFM_LOSS_LONE_HALF: {
      const double sa = sqrt(a);
      o.r0 = 2.0 * a * sa * pow(s, 0.25);
      if (s < 0.01) s = 0.01;
      o.r1 = 0.5 * pow(a, -1.5) * pow(s, -0.75);
      o.r2 = -0.375 * a * sa * pow(s, -1.75);
      break; }
    case GSFM_LOSS_LTWO: {
      const double a2 = a * a;
      o.r0 = s * s / (a2 * 2.0); o.r1 = s / a2; o.r2 = 1.0 / a2;
      break; }
    case GSFM_LOSS_GEMAN_MCCLURE: {
      const double a2 = a * a, g2 = n.p[1];
      const double t = s / a2 + g2;
      o.r0 = a2 * g2 * s / (2.0 * (s + a2 * g2));
      o.r1 = (g2 * g2) / (2.0 * (t * t));
      o.r2 = -(g2 * g2) / (a2 * (t * t * t));
      break; }
    case GSFM_LOSS_MAGSAC: return loss_magsac(n, s);
    default: o.r0 = s; o.r1 = 1.0; o.r2 = 0.0;
  }
  return o;
}

// Cheap single leaves (no pow / cosh / atan2 / table): the kernels are specialised on the loss shape so that
// the common configurations do not pay the registers of the general interpreter.
__device__ __forceinline__ Rho3 loss_leaf_simple(const DevLossNode& n, double s) {
  Rho3 o;
  const double a = n.p[0];
  switch (n.kind) {
    case GSFM_LOSS_HUBER: {
      const double b = a * a;
      if (s > b) { const double r = sqrt(s); o.r0 = 2.0 * a * r - b; o.r1 = fmax(a / r, DBL_MIN); o.r2 = -o.r1 / (2.0 * s); }
      else { o.r0 = s; o.r1 = 1.0; o.r2 = 0.0; }
      break; }
    case GSFM_LOSS_SOFT_L1: {
      const double b = a * a, c = 1.0 / b;
      const double sum = 1.0 + s * c, tmp = sqrt(sum);
      o.r0 = 2.0 * b * (tmp - 1.0); o.r1 = fmax(1.0 / tmp, DBL_MIN); o.r2 = -(c * o.r1) / (2.0 * sum);
      break; }
    case GSFM_LOSS_TUKEY: {
      const double a2 = a * a;
      if (s <= a2) { const double v = 1.0 - s / a2, v2 = v * v; o.r0 = a2 / 6.0 * (1.0 - v2 * v); o.r1 = 0.5 * v2; o.r2 = -1.0 / a2 * v; }
      else { o.r0 = a2 / 6.0; o.r1 = 0.0; o.r2 = 0.0; }
      break; }
    case GSFM_LOSS_GEMAN_MCCLURE: {
      const double a2 = a * a, g2 = n.p[1];
      const double t = s / a2 + g2;
      o.r0 = a2 * g2 * s / (2.0 * (s + a2 * g2));
      o.r1 = (g2 * g2) / (2.0 * (t * t));
      o.r2 = -(g2 * g2) / (a2 * (t * t * t));
      break; }
    default: o.r0 = s; o.r1 = 1.0; o.r2 = 0.0;  // TRIVIAL / NULL loss
  }
  return o;
}

// rho only (trial-cost sweeps never need the derivatives): the MAGSAC value is one table lookup.
__device__ __forceinline__ double loss_magsac_value(const DevLossNode& n, double sq) {
#pragma clang fp contract(off)   // the same rounding as loss_magsac3().r0 (the sigma-consensus form of the first cost sweep uses that one): x_cost and the trial costs agree to the bit
  const double ssm2 = n.aux[1];
  if (sq > n.aux[6]) sq = n.aux[6];
  long x = (long)rint(1000.0 * sq / ssm2);     // same expression as the full evaluation: same table cell
  if (x > (long)n.table_len - 1) x = (long)n.table_len - 1;
  // nu = 3: table[x] = Gamma(1, x/1000) = exp(-x/1000); evaluating it beats a second random gather (the
  // arithmetic of this kernel is hidden behind the streams). Same quantised x, value within 1 ulp of the table.
  const double tv = (n.nu == 3) ? exp(-1e-3 * (double)x) : n.table[x];
  const double weight = n.aux[4] * (tv - n.aux[7]);
  return n.inverse ? 1.0 / weight : n.aux[5] - weight;
}

// Kernel specialisations.  LM_MAGSAC = ONE MAGSACWeightBasedLoss leaf with nu = 3, not inverted (the pipeline's default,
// scripts/sfm_pipeline.py:136); the nu = 4 / 9 and inverse variants run through the general program.
enum { LM_PROGRAM = 0, LM_SIMPLE = 1, LM_MAGSAC = 2 };

// The nu = 3, non-inverted MAGSAC weight loss (LM_MAGSAC), all three values from ONE exponential and ONE exact division: with nu = 3 the
// stored table IS the exponential, Gamma(1, x / 1000) = exp(-x / 1000) (include/gamma_values.cpp regenerates from exactly that), and the
// derivative formulas of loss_functions.py:308-321 reduce to constants times exp(-s_q / 2 sigma^2) with s_q = x 2 sigma^2 / 1000 the
// quantised argument, i.e. the same exponential (to ~3 ulp of its argument); the constants are host-precomputed.  The general routine
// above spends two exp, a table gather and nine IEEE divisions on the same three numbers.  Pinned by the recorded reference rows
// (tests/test_gpu_loss_golden.py, 1e-12 relative, tie-rounding rows included: the cell index comes from the same exact division).
__device__ __forceinline__ Rho3 loss_magsac3(const DevLossNode& n, double sq) {
#pragma clang fp contract(off)   // rho = w(0) - w(s) must round the product before the subtraction, like the reference: rho(0) == 0 exactly
  bool zero_derivative = false;
  if (sq > n.aux[6]) { sq = n.aux[6]; zero_derivative = true; }
  const long x = (long)rint(1000.0 * sq / n.aux[1]);   // Python round(): half to even
  const double e = exp(-1e-3 * (double)x);
  Rho3 o;
  o.r0 = n.aux[5] - n.aux[4] * (e - n.aux[7]);
  o.r1 = n.rho1_scale * e;
  if (o.r1 == 0.0) o.r1 = 0.00001;
  o.r2 = -n.rho2_scale * (x < (long)n.x_clamp ? n.e2_clamp : e);
  if (zero_derivative) { o.r1 = 0.00001; o.r2 = 0.0; }
  return o;
}

// rho' alone, for the linearisation (K2) of losses whose rho'' is never positive -- every LM_SIMPLE leaf and the nu = 3 MAGSAC
// weight loss (rho'' = -C / (8 sigma^5) exp(-u) < 0) -- so that Ceres' Corrector takes its alpha = 0 branch for every edge and
// needs nothing but sqrt(rho').  nu = 3: weight'(s) = -C 2 exp(-s / 2 sigma^2) / (2 sigma^3) on the quantised s (loss_functions.py:
// 304-321): the table cell x comes from the same exact division as in loss_magsac, exp(-x / 1000) IS the table value
// Gamma(1, x / 1000), and the constant factor is one host-precomputed product instead of a division per edge.
// SC: the exponential with its coefficients in scalar registers (devmath.hpp: the same bits, ~25 vector registers fewer).
template <bool SC>
__device__ __forceinline__ double loss_magsac3_rho1(const DevLossNode& n, double sq) {
  if (sq > n.aux[6]) return 0.00001;
  const long x = (long)rint(1000.0 * sq / n.aux[1]);
  const double r1 = n.rho1_scale * (SC ? exp_sc(-1e-3 * (double)x) : exp(-1e-3 * (double)x));
  return r1 == 0.0 ? 0.00001 : r1;
}

// Evaluate the program. `loss` is a wave-uniform global pointer (scalar loads).
__device__ __forceinline__ Rho3 loss_eval_program(const DevLoss* __restrict__ loss, double s) {
  const int n = loss->n;
  if (n == 1) return loss_leaf(loss->nodes[0], s);  // one leaf of any kind
  if (n <= 0) { Rho3 t; t.r0 = s; t.r1 = 1.0; t.r2 = 0.0; return t; }
  Rho3 res[GSFM_LOSS_MAX_STACK];
  double arg[GSFM_LOSS_MAX_STACK];
  int nr = 0, na = 1;
  arg[0] = s;
  for (int k = 0; k < n; ++k) {
    const DevLossNode& nd = loss->nodes[k];
    if (nd.kind == GSFM_LOSS_OP_SCALE) {
      res[nr - 1].r0 *= nd.p[0]; res[nr - 1].r1 *= nd.p[0]; res[nr - 1].r2 *= nd.p[0];
    } else if (nd.kind == GSFM_LOSS_OP_PUSH_ARG) {
      arg[na++] = res[nr - 1].r0;
    } else if (nd.kind == GSFM_LOSS_OP_COMPOSE) {
      const Rho3 f = res[--nr];
      const Rho3 g = res[--nr];
      --na;
      Rho3 o;
      o.r0 = f.r0; o.r1 = f.r1 * g.r1; o.r2 = f.r2 * g.r1 * g.r1 + f.r1 * g.r2;
      res[nr++] = o;
    } else {
      res[nr++] = loss_leaf(nd, arg[na - 1]);
    }
  }
  return res[0];
}

template <int LM>
__device__ __forceinline__ Rho3 loss_eval(const DevLoss* __restrict__ loss, double s) {
  if (LM == LM_SIMPLE) return loss_leaf_simple(loss->nodes[0], s);
  if (LM == LM_MAGSAC) return loss_magsac3(loss->nodes[0], s);
  return loss_eval_program(loss, s);
}

template <int LM>
__device__ __forceinline__ double loss_value(const DevLoss* __restrict__ loss, double s) {
  if (LM == LM_MAGSAC) return loss_magsac_value(loss->nodes[0], s);
  return loss_eval<LM>(loss, s).r0;  // the unused derivatives are dead code for the simple leaves
}

// What a kernel keeps of the loss after its prologue.  For the single-leaf specialisations a COPY of the leaf, taken before the kernel's
// first global store: the compiler then reads the fields it needs with scalar loads and they stay in SGPRs.  Read through the pointer
// INSIDE a kernel that also stores (K2, K2c, the K1 modes with per-edge outputs), the same fields were per-lane vector loads from a uniform
// address (scalar loads are not coherent with the kernel's own stores, so the compiler may not use them), each followed by
// s_waitcnt vmcnt(0) -- a wait for everything the lane had in flight, twice per entry in the MAGSAC leaf (cut test, then cell and scale).
// The general program keeps the pointer (its node loop is the slow path anyway).
template <int LM> struct LossView { const DevLoss* p; DevLossNode n; };
template <int LM>
__device__ __forceinline__ LossView<LM> loss_view(const DevLoss* __restrict__ p) {
  LossView<LM> v{};   // (the general program never reads the copy)
  v.p = p;
  if (LM != LM_PROGRAM) v.n = p->nodes[0];
  return v;
}
template <int LM>
__device__ __forceinline__ Rho3 loss_eval(const LossView<LM>& v, double s) {
  if (LM == LM_SIMPLE) return loss_leaf_simple(v.n, s);
  if (LM == LM_MAGSAC) return loss_magsac3(v.n, s);
  return loss_eval_program(v.p, s);
}
template <int LM>
__device__ __forceinline__ double loss_value(const LossView<LM>& v, double s) {
  if (LM == LM_MAGSAC) return loss_magsac_value(v.n, s);
  return loss_eval<LM>(v, s).r0;
}
// K2's fast path (see lin_rows_fast): only for LM_SIMPLE / LM_MAGSAC
template <int LM, bool SC = true>
__device__ __forceinline__ double loss_rho1(const LossView<LM>& v, double s) {
  if (LM == LM_MAGSAC) return loss_magsac3_rho1<SC>(v.n, s);
  return loss_leaf_simple(v.n, s).r1;
}
template <int LM, bool SC = true>
__device__ __forceinline__ double loss_rho1(const DevLoss* __restrict__ loss, double s) { return loss_rho1<LM, SC>(loss_view<LM>(loss), s); }

// Ceres Corrector (corrector.cc 1.14): residual scaling and the alpha term.
struct Corrector { double sqrt_rho1, residual_scaling, alpha_sq_norm; };
__device__ __forceinline__ Corrector make_corrector(double s, const Rho3& rho) {
  Corrector c;
  c.sqrt_rho1 = sqrt(rho.r1);
  if (s == 0.0 || rho.r2 <= 0.0) { c.residual_scaling = c.sqrt_rho1; c.alpha_sq_norm = 0.0; }
  else {
    const double D = 1.0 + 2.0 * s * rho.r2 / rho.r1;
    const double alpha = 1.0 - sqrt(D);
    c.residual_scaling = c.sqrt_rho1 / (1.0 - alpha);
    c.alpha_sq_norm = alpha / s;
  }
  return c;
}

}  // namespace gsfm

// TEST INFRASTRUCTURE — CPU oracle. Not part of the product; never linked by it.
//
// Robust losses of the reference, restated from scripts/loss_functions.py
// (each leaf cites the class it follows) plus the MAGSAC constants/tables of
// include/gamma_values.cpp.  The tables are NOT copied: table[x] = Gamma((nu-1)/2, x/1000)
// has a closed form for nu = 3, 4, 9 and is regenerated here; tests/golden/ holds
// sampled values of the reference table to pin the regeneration.
#pragma once
#include <cfloat>
#include <cmath>
#include <vector>
#include "../include/gsfm_rot.h"

namespace gsfm_oracle {

struct MagsacConst {
  double nu, C, sigma_quantile, upper_incomplete_gamma_of_k;
  int stored_gamma_number;
  double precision;
};

// include/gamma_values.cpp:6-11, 384-389, 780-785 (values are inputs of the loss definition)
inline MagsacConst magsac_const(int nu) {
  switch (nu) {
    case 3: return {3.0, 4.029720004054876e-01, 3.368214175218727, 3.439485560754856e-03, 36843, 1000.0};
    case 4: return {4.0, 2.525252525252525e-01, 3.643721193503644e+00, 3.611260617758625e-03, 38683, 1000.0};
    default: return {9.0, 3.837828575290349e-03, 4.654674460524809e+00, 3.344206155099048e-02, 48553, 1000.0};
  }
}

// Upper incomplete gamma Gamma(a, x), a = (nu-1)/2 in {1, 1.5, 4}: closed forms.
inline double upper_gamma_closed_form(int nu, double x) {
  if (x == 0.0) return std::tgamma((nu - 1.0) / 2.0);  // Gamma(a, 0) = Gamma(a): keeps rho(0) == 0 exactly
  if (nu == 3) return std::exp(-x);                                              // Gamma(1, x)
  if (nu == 4) return 0.5 * std::sqrt(M_PI) * std::erfc(std::sqrt(x)) + std::sqrt(x) * std::exp(-x);  // Gamma(3/2, x)
  return 6.0 * std::exp(-x) * (1.0 + x + x * x / 2.0 + x * x * x / 6.0);         // Gamma(4, x)
}

inline std::vector<double> make_magsac_table(int nu) {
  const MagsacConst c = magsac_const(nu);
  std::vector<double> t(c.stored_gamma_number);
  for (int x = 0; x < c.stored_gamma_number; ++x) t[x] = upper_gamma_closed_form(nu, x / c.precision);
  return t;
}
inline const std::vector<double>& magsac_table(int nu) {
  // function-local statics: initialisation is thread-safe (the evaluator runs under OpenMP)
  static const std::vector<double> t3 = make_magsac_table(3);
  static const std::vector<double> t4 = make_magsac_table(4);
  static const std::vector<double> t9 = make_magsac_table(9);
  return (nu == 3) ? t3 : (nu == 4) ? t4 : t9;
}

struct Rho { double v[3]; };

// Python round(): half to even -> rint() under the default rounding mode.
inline double py_round(double x) { return std::rint(x); }

inline Rho eval_magsac(double sigma, int nu, bool inverse, double squared_residual) {
  // loss_functions.py:285-341 (nu=3), :344-400 (nu=4), :402-458 (nu=9); identical bodies.
  const MagsacConst c = magsac_const(nu);
  const std::vector<double>& table = magsac_table(nu);
  const double squared_sigma = sigma * sigma;
  const double squared_sigma_max_2 = 2.0 * squared_sigma;
  const double cubed_sigma_max = squared_sigma * sigma;
  const double dof_minus_one_per_two = (c.nu - 1.0) / 2.0;
  const double C_times_two_ad_dof = c.C * std::pow(2.0, dof_minus_one_per_two);
  const double one_over_sigma = C_times_two_ad_dof / sigma;
  const double gamma_value = std::tgamma(dof_minus_one_per_two);
  const double gamma_difference = gamma_value - c.upper_incomplete_gamma_of_k;
  const double weight_zero = one_over_sigma * gamma_difference;

  bool zero_derivative = false;
  if (squared_residual > c.sigma_quantile * c.sigma_quantile * squared_sigma) {
    squared_residual = c.sigma_quantile * c.sigma_quantile * squared_sigma;
    zero_derivative = true;
  }
  long x = (long)py_round(c.precision * squared_residual / squared_sigma_max_2);
  if (c.stored_gamma_number < x) x = c.stored_gamma_number;  // unreachable after the clamp above
  double s = x * squared_sigma_max_2 / c.precision;
  const double weight = one_over_sigma * (table[x] - c.upper_incomplete_gamma_of_k);
  const double ex = c.nu / 2.0 - 1.5;
  const double weight_derivative =
      -C_times_two_ad_dof * std::pow(s / squared_sigma_max_2, ex) * std::exp(-s / squared_sigma_max_2) /
      (2.0 * cubed_sigma_max);
  if (s < 1e-7) s = 1e-7;
  const double weight_second_derivative =
      2.0 * C_times_two_ad_dof * std::pow(s / squared_sigma_max_2, ex) *
      (1.0 / squared_sigma - (c.nu - 3.0) / s) * std::exp(-s / squared_sigma_max_2) / (8.0 * cubed_sigma_max);
  Rho r;
  if (inverse) {
    r.v[0] = 1.0 / weight;
    r.v[1] = -1.0 / (weight * weight) * weight_derivative;
    r.v[2] = 2.0 / (weight * weight * weight) * weight_derivative * weight_derivative -
             weight_second_derivative / (weight * weight);
    if (zero_derivative) { r.v[1] = 0.00001; r.v[2] = 0.0; }
  } else {
    r.v[0] = weight_zero - weight;
    r.v[1] = -weight_derivative;
    r.v[2] = -weight_second_derivative;
    if (r.v[1] == 0) r.v[1] = 0.00001;
    if (zero_derivative) { r.v[1] = 0.00001; r.v[2] = 0.0; }
  }
  return r;
}

inline Rho eval_leaf(const gsfm_loss_node& n, double s) {
  Rho o;
  const double a = n.p[0];
  switch (n.kind) {
    case GSFM_LOSS_TRIVIAL:  // loss_functions.py:47-54
      o.v[0] = s; o.v[1] = 1.0; o.v[2] = 0.0; break;
    case GSFM_LOSS_HUBER: {  // :56-72
      const double b = a * a;
      if (s > b) {
        const double r = std::sqrt(s);
        o.v[0] = 2 * a * r - b; o.v[1] = std::fmax(a / r, DBL_MIN); o.v[2] = -o.v[1] / (2.0 * s);
      } else { o.v[0] = s; o.v[1] = 1.0; o.v[2] = 0.0; }
      break; }
    case GSFM_LOSS_SOFT_L1: {  // :74-86
      const double b = a * a, c = 1.0 / b;
      const double sum = 1.0 + s * c, tmp = std::sqrt(sum);
      o.v[0] = 2.0 * b * (tmp - 1.0); o.v[1] = std::fmax(1.0 / tmp, DBL_MIN); o.v[2] = -(c * o.v[1]) / (2.0 * sum);
      break; }
    case GSFM_LOSS_CAUCHY: {  // :88-99
      const double b = a * a, c = 1.0 / b;
      const double sum = 1.0 + s * c, inv = 1.0 / sum;
      o.v[0] = b * std::log(sum); o.v[1] = std::fmax(inv, DBL_MIN); o.v[2] = -c * (inv * inv);
      break; }
    case GSFM_LOSS_ARCTAN: {  // :101-112
      const double b = 1 / (a * a);
      const double sum = 1 + s * s * b, inv = 1.0 / sum;
      o.v[0] = a * std::atan2(s, a); o.v[1] = std::fmax(inv, DBL_MIN); o.v[2] = -2.0 * s * b * (inv * inv);
      break; }
    case GSFM_LOSS_TOLERANT: {  // :114-165
      const double b = n.p[1];
      const double c = b * std::log(1 + std::exp(-a / b));
      const double x = (s - a) / b;
      if (x > 36.7) { o.v[0] = s - a - c; o.v[1] = 1.0; o.v[2] = 0.0; }
      else {
        const double e_x = std::exp(x);
        o.v[0] = b * std::log(1.0 + e_x) - c;
        o.v[1] = std::fmax(e_x / (1.0 + e_x), DBL_MIN);
        o.v[2] = 0.5 / (b * (1.0 + std::cosh(x)));
      }
      break; }
    case GSFM_LOSS_TUKEY: {  // :167-185
      const double a2 = a * a;
      if (s <= a2) {
        const double value = 1.0 - s / a2, value_sq = value * value;
        o.v[0] = a2 / 6.0 * (1.0 - value_sq * value); o.v[1] = 0.5 * value_sq; o.v[2] = -1.0 / a2 * value;
      } else { o.v[0] = a2 / 6.0; o.v[1] = 0.0; o.v[2] = 0.0; }
      break; }
    case GSFM_LOSS_LONE_HALF: {  // :187-215
      const double sqrt_a = std::sqrt(a);
      o.v[0] = 2.0 * a * sqrt_a * std::pow(s, 0.25);
      if (s < 0.01) s = 0.01;
      o.v[1] = 0.5 * std::pow(a, -1.5) * std::pow(s, -0.75);
      o.v[2] = -0.375 * a * sqrt_a * std::pow(s, -1.75);
      break; }
    case GSFM_LOSS_LTWO: {  // :216-237
      const double a_sq = a * a;
      o.v[0] = s * s / (a_sq * 2.0); o.v[1] = s / a_sq; o.v[2] = 1 / a_sq;
      break; }
    case GSFM_LOSS_GEMAN_MCCLURE: {  // :239-248
      const double a_sq = a * a, sigma2 = n.p[1];
      o.v[0] = a_sq * sigma2 * s / (2.0 * (s + a_sq * sigma2));
      const double t = s / a_sq + sigma2;
      o.v[1] = (sigma2 * sigma2) / (2.0 * (t * t));
      o.v[2] = -(sigma2 * sigma2) / (a_sq * (t * t * t));
      break; }
    case GSFM_LOSS_MAGSAC:
      return eval_magsac(n.p[0], (int)n.p[1], n.p[2] != 0.0, s);
    default:
      o.v[0] = o.v[1] = o.v[2] = NAN;
  }
  return o;
}

// The stack machine of include/gsfm_rot.h (ComposedLoss :250-265, ScaledLoss :267-281).
inline Rho eval_loss_program(const gsfm_loss_node* prog, int n, double s) {
  if (n <= 0) { Rho t; t.v[0] = s; t.v[1] = 1.0; t.v[2] = 0.0; return t; }
  Rho res[GSFM_LOSS_MAX_STACK];
  double arg[GSFM_LOSS_MAX_STACK];
  int nr = 0, na = 1;
  arg[0] = s;
  for (int k = 0; k < n; ++k) {
    const gsfm_loss_node& nd = prog[k];
    if (nd.kind == GSFM_LOSS_OP_SCALE) {
      for (int c = 0; c < 3; ++c) res[nr - 1].v[c] *= nd.p[0];
    } else if (nd.kind == GSFM_LOSS_OP_PUSH_ARG) {
      arg[na++] = res[nr - 1].v[0];
    } else if (nd.kind == GSFM_LOSS_OP_COMPOSE) {
      const Rho f = res[--nr];
      const Rho g = res[--nr];
      --na;
      Rho o;
      o.v[0] = f.v[0];
      o.v[1] = f.v[1] * g.v[1];
      o.v[2] = f.v[2] * g.v[1] * g.v[1] + f.v[1] * g.v[2];
      res[nr++] = o;
    } else {
      res[nr++] = eval_leaf(nd, arg[na - 1]);
    }
  }
  return res[0];
}

}  // namespace gsfm_oracle

// TEST INFRASTRUCTURE — CPU oracle. Not part of the product; never linked by it.
//
// Forward-mode dual numbers, restating what ceres::Jet<double, N> does for the
// reference's AutoDiffCostFunction<..> residual blocks
// (reference src/pairwise_rotation_error.cpp:46-85 instantiate AutoDiffCostFunction
//  with N = 3+3 or 4+4 parameters; ceres-solver 1.14.0 is NOT vendored in the
//  reference tree, its published jet.h semantics are restated here).
#pragma once
#include <cmath>

namespace gsfm_oracle {

template <int N>
struct Jet {
  double a;
  double v[N];
  Jet() : a(0.0) { for (int k = 0; k < N; ++k) v[k] = 0.0; }
  Jet(double x) : a(x) { for (int k = 0; k < N; ++k) v[k] = 0.0; }  // NOLINT (implicit like ceres)
  Jet(double x, int k) : a(x) { for (int i = 0; i < N; ++i) v[i] = 0.0; v[k] = 1.0; }
};

template <int N> inline Jet<N> operator+(const Jet<N>& f, const Jet<N>& g) {
  Jet<N> h; h.a = f.a + g.a; for (int k = 0; k < N; ++k) h.v[k] = f.v[k] + g.v[k]; return h; }
template <int N> inline Jet<N> operator-(const Jet<N>& f, const Jet<N>& g) {
  Jet<N> h; h.a = f.a - g.a; for (int k = 0; k < N; ++k) h.v[k] = f.v[k] - g.v[k]; return h; }
template <int N> inline Jet<N> operator-(const Jet<N>& f) {
  Jet<N> h; h.a = -f.a; for (int k = 0; k < N; ++k) h.v[k] = -f.v[k]; return h; }
template <int N> inline Jet<N> operator*(const Jet<N>& f, const Jet<N>& g) {
  Jet<N> h; h.a = f.a * g.a; for (int k = 0; k < N; ++k) h.v[k] = f.a * g.v[k] + f.v[k] * g.a; return h; }
template <int N> inline Jet<N> operator/(const Jet<N>& f, const Jet<N>& g) {
  // ceres jet.h: h = f/g, dh = (df - h dg)/g
  Jet<N> h; const double gi = 1.0 / g.a; h.a = f.a * gi;
  for (int k = 0; k < N; ++k) h.v[k] = (f.v[k] - h.a * g.v[k]) * gi; return h; }
template <int N> inline Jet<N> operator+(const Jet<N>& f, double s) { Jet<N> h = f; h.a += s; return h; }
template <int N> inline Jet<N> operator+(double s, const Jet<N>& f) { Jet<N> h = f; h.a += s; return h; }
template <int N> inline Jet<N> operator-(const Jet<N>& f, double s) { Jet<N> h = f; h.a -= s; return h; }
template <int N> inline Jet<N> operator-(double s, const Jet<N>& f) { Jet<N> h = -f; h.a += s; return h; }
template <int N> inline Jet<N> operator*(const Jet<N>& f, double s) {
  Jet<N> h; h.a = f.a * s; for (int k = 0; k < N; ++k) h.v[k] = f.v[k] * s; return h; }
template <int N> inline Jet<N> operator*(double s, const Jet<N>& f) { return f * s; }
template <int N> inline Jet<N> operator/(const Jet<N>& f, double s) { return f * (1.0 / s); }
template <int N> inline Jet<N> operator/(double s, const Jet<N>& g) { return Jet<N>(s) / g; }

template <int N> inline bool operator<(const Jet<N>& f, const Jet<N>& g) { return f.a < g.a; }
template <int N> inline bool operator>(const Jet<N>& f, const Jet<N>& g) { return f.a > g.a; }
template <int N> inline bool operator>=(const Jet<N>& f, const Jet<N>& g) { return f.a >= g.a; }
template <int N> inline bool operator<(const Jet<N>& f, double s) { return f.a < s; }
template <int N> inline bool operator>(const Jet<N>& f, double s) { return f.a > s; }
template <int N> inline bool operator>=(const Jet<N>& f, double s) { return f.a >= s; }

template <int N> inline Jet<N> sqrt(const Jet<N>& f) {
  Jet<N> h; h.a = std::sqrt(f.a); const double t = 1.0 / (2.0 * h.a);
  for (int k = 0; k < N; ++k) h.v[k] = f.v[k] * t; return h; }
template <int N> inline Jet<N> sin(const Jet<N>& f) {
  Jet<N> h; h.a = std::sin(f.a); const double c = std::cos(f.a);
  for (int k = 0; k < N; ++k) h.v[k] = c * f.v[k]; return h; }
template <int N> inline Jet<N> cos(const Jet<N>& f) {
  Jet<N> h; h.a = std::cos(f.a); const double s = -std::sin(f.a);
  for (int k = 0; k < N; ++k) h.v[k] = s * f.v[k]; return h; }
template <int N> inline Jet<N> atan2(const Jet<N>& g, const Jet<N>& f) {
  // atan2(g, f): d = (f dg - g df) / (f^2 + g^2)
  Jet<N> h; h.a = std::atan2(g.a, f.a); const double t = 1.0 / (f.a * f.a + g.a * g.a);
  for (int k = 0; k < N; ++k) h.v[k] = t * (f.a * g.v[k] - g.a * f.v[k]); return h; }

// plain-double overloads so the templated code below reads the same for T = double
inline double sqrt(double x) { return std::sqrt(x); }
inline double sin(double x) { return std::sin(x); }
inline double cos(double x) { return std::cos(x); }
inline double atan2(double y, double x) { return std::atan2(y, x); }

inline double scalar_part(double x) { return x; }
template <int N> inline double scalar_part(const Jet<N>& x) { return x.a; }

}  // namespace gsfm_oracle

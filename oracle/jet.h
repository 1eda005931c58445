// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// Forward-mode dual number ("jet") of compile-time width N: value + N partial
// derivatives, propagated by the chain rule.  This restates what the reference
// obtains from ceres::AutoDiffCostFunction (reference src/base/cost_functions.h:55-60,
// :130-137): the exact derivative of the templated functor, evaluated by
// operator overloading.  Ceres itself is NOT in /root/reference (third-party,
// version unpinned by README.md:115-120); only the differentiation rule is
// restated here, and that rule is just calculus.
#pragma once
#include <cmath>

namespace oracle {

template <int N>
struct Jet {
  double a;     // value
  double v[N];  // d(value)/d(param_k)

  Jet() : a(0.0) { for (int k = 0; k < N; ++k) v[k] = 0.0; }
  Jet(double s) : a(s) { for (int k = 0; k < N; ++k) v[k] = 0.0; }  // NOLINT implicit
  static Jet Var(double s, int k) { Jet j(s); j.v[k] = 1.0; return j; }
};

template <int N> inline Jet<N> operator+(const Jet<N>& x, const Jet<N>& y) {
  Jet<N> r; r.a = x.a + y.a; for (int k = 0; k < N; ++k) r.v[k] = x.v[k] + y.v[k]; return r; }
template <int N> inline Jet<N> operator-(const Jet<N>& x, const Jet<N>& y) {
  Jet<N> r; r.a = x.a - y.a; for (int k = 0; k < N; ++k) r.v[k] = x.v[k] - y.v[k]; return r; }
template <int N> inline Jet<N> operator-(const Jet<N>& x) {
  Jet<N> r; r.a = -x.a; for (int k = 0; k < N; ++k) r.v[k] = -x.v[k]; return r; }
template <int N> inline Jet<N> operator*(const Jet<N>& x, const Jet<N>& y) {
  Jet<N> r; r.a = x.a * y.a; for (int k = 0; k < N; ++k) r.v[k] = x.a * y.v[k] + x.v[k] * y.a; return r; }
template <int N> inline Jet<N> operator/(const Jet<N>& x, const Jet<N>& y) {
  // d(x/y) = (dx - (x/y) dy) / y
  Jet<N> r; const double inv = 1.0 / y.a; r.a = x.a * inv;
  for (int k = 0; k < N; ++k) r.v[k] = (x.v[k] - r.a * y.v[k]) * inv; return r; }
template <int N> inline Jet<N>& operator+=(Jet<N>& x, const Jet<N>& y) { x = x + y; return x; }
template <int N> inline Jet<N>& operator-=(Jet<N>& x, const Jet<N>& y) { x = x - y; return x; }
template <int N> inline Jet<N>& operator*=(Jet<N>& x, const Jet<N>& y) { x = x * y; return x; }
template <int N> inline Jet<N>& operator/=(Jet<N>& x, const Jet<N>& y) { x = x / y; return x; }

template <int N> inline bool operator<(const Jet<N>& x, const Jet<N>& y) { return x.a < y.a; }
template <int N> inline bool operator>(const Jet<N>& x, const Jet<N>& y) { return x.a > y.a; }

template <int N> inline Jet<N> sqrt(const Jet<N>& x) {
  Jet<N> r; r.a = std::sqrt(x.a); const double d = 0.5 / r.a;
  for (int k = 0; k < N; ++k) r.v[k] = d * x.v[k]; return r; }
template <int N> inline Jet<N> atan(const Jet<N>& x) {
  Jet<N> r; r.a = std::atan(x.a); const double d = 1.0 / (1.0 + x.a * x.a);
  for (int k = 0; k < N; ++k) r.v[k] = d * x.v[k]; return r; }
template <int N> inline Jet<N> tan(const Jet<N>& x) {
  Jet<N> r; r.a = std::tan(x.a); const double d = 1.0 + r.a * r.a;
  for (int k = 0; k < N; ++k) r.v[k] = d * x.v[k]; return r; }

// scalar overloads so the same templated code runs with T = double
inline double sqrt(double x) { return std::sqrt(x); }
inline double atan(double x) { return std::atan(x); }
inline double tan(double x) { return std::tan(x); }

}  // namespace oracle

// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// The rules of Ceres' trust-region minimiser with the Levenberg-Marquardt strategy, as PUBLISHED (Ceres Solver documentation, "Solving Non-linear Least
// Squares": TrustRegionMinimizer / LevenbergMarquardtStrategy; the reference configures it at src/optim/bundle_adjustment.cc:273-306), in ONE place: the
// bundle-adjustment loop of oracle/bundle_adjustment.h takes every rule from here, and so does DenseLevenbergMarquardt below - a driver for small dense
// problems whose only purpose is to PIN these rules to Ceres itself without Ceres: on Powell's function (Ceres' examples/powell.cc) it reproduces the
// iteration table the Ceres tutorial prints - cost, |gradient|, |step|, tr_ratio, tr_radius of all fifteen iterations, digit for digit
// (tests/golden/ceres_powell_trace.txt, tests/test_oracle_bundle_adjustment.py).  What that pins: Jacobi scaling 1 / (1 + ||column||) fixed at the start,
// the LM diagonal clamp(diag(J'J)) / radius on the scaled system, the step-quality ratio against the model's cost change -(J d)'(r + J d / 2), the
// acceptance threshold, radius /= max(1/3, 1 - (2 rho - 1)^3) on success and radius /= k, k *= 2 on failure, the gradient max-norm.  What it does not pin:
// the Schur elimination, the loss corrector and the manifold Plus of the bundle-adjustment problem (known-answer tests of their own).
#pragma once
#include <cmath>
#include <functional>
#include <vector>
#include "linalg.h"

namespace oracle {
namespace lm {

inline double JacobiScale(double squared_column_norm) { return 1.0 / (1.0 + std::sqrt(squared_column_norm)); }
inline double ClampDiagonal(double d, double min_lm_diagonal, double max_lm_diagonal) { return std::fmin(std::fmax(d, min_lm_diagonal), max_lm_diagonal); }
// entry of the LM regulariser D (the system solved is J_s'J_s + D^2) for a clamped diagonal entry of J_s'J_s
inline double LmD(double clamped_diagonal, double radius) { return std::sqrt(clamped_diagonal / radius); }

struct Radius {
  double radius, decrease_factor = 2.0;
  void Accept(double step_quality, double max_radius) {
    radius = radius / std::fmax(1.0 / 3.0, 1.0 - std::pow(2.0 * step_quality - 1.0, 3));
    radius = std::fmin(max_radius, radius);
    decrease_factor = 2.0;
  }
  void Reject() { radius /= decrease_factor; decrease_factor *= 2.0; }
};

struct Options {
  int max_num_iterations = 50;
  double function_tolerance = 1e-6, gradient_tolerance = 1e-10, parameter_tolerance = 1e-8;      // Ceres' Solver::Options defaults
  double initial_trust_region_radius = 1e4, max_trust_region_radius = 1e16, min_trust_region_radius = 1e-32, min_relative_decrease = 1e-3;
  double min_lm_diagonal = 1e-6, max_lm_diagonal = 1e32;
};
struct Iteration { double cost, cost_change, gradient_max_norm, step_norm, relative_decrease, radius; int successful; };

// eval(x, r, J): residuals r (m) and the row-major m x n Jacobian J at x.  Euclidean parameters, trivial loss, dense normal equations.
inline std::vector<Iteration> DenseLevenbergMarquardt(int m, int n, const std::function<void(const double*, double*, double*)>& eval, double* x, const Options& opt) {
  std::vector<Iteration> trace;
  std::vector<double> r(m), J((size_t)m * n), rc(m), Jc((size_t)m * n), xc(n), scale(n), g(n), diag(n), A((size_t)n * n), step(n), delta(n);
  auto cost_of = [&](const std::vector<double>& res) { double c = 0; for (double v : res) c += v * v; return 0.5 * c; };
  auto gradient = [&]() { double gm = 0; for (int j = 0; j < n; ++j) { double s = 0; for (int i = 0; i < m; ++i) s += J[(size_t)i * n + j] * r[i]; g[j] = s; gm = std::fmax(gm, std::fabs(s)); } return gm; };
  eval(x, r.data(), J.data());
  double cost = cost_of(r), gmax = gradient();
  for (int j = 0; j < n; ++j) { double cn = 0; for (int i = 0; i < m; ++i) cn += J[(size_t)i * n + j] * J[(size_t)i * n + j]; scale[j] = JacobiScale(cn); }
  Radius tr{opt.initial_trust_region_radius};
  trace.push_back({cost, 0.0, gmax, 0.0, 0.0, tr.radius, 1});
  bool reuse_diagonal = false, last_successful = true;
  for (int iter = 1;; ++iter) {
    if (last_successful && gmax <= opt.gradient_tolerance) break;
    if (iter > opt.max_num_iterations || tr.radius < opt.min_trust_region_radius) break;
    if (!reuse_diagonal)
      for (int j = 0; j < n; ++j) { double cn = 0; for (int i = 0; i < m; ++i) { const double v = J[(size_t)i * n + j] * scale[j]; cn += v * v; } diag[j] = ClampDiagonal(cn, opt.min_lm_diagonal, opt.max_lm_diagonal); }
    reuse_diagonal = true;
    for (int a = 0; a < n; ++a)
      for (int b = 0; b < n; ++b) { double s = 0; for (int i = 0; i < m; ++i) s += J[(size_t)i * n + a] * scale[a] * J[(size_t)i * n + b] * scale[b]; A[(size_t)a * n + b] = s; }
    for (int j = 0; j < n; ++j) { const double d = LmD(diag[j], tr.radius); A[(size_t)j * n + j] += d * d; step[j] = -scale[j] * g[j]; }
    bool valid = CholeskyFactor(n, A.data());
    double model_change = 0;
    if (valid) {
      CholeskySolve(n, A.data(), step.data());
      for (int i = 0; i < m; ++i) { double jd = 0; for (int j = 0; j < n; ++j) jd += J[(size_t)i * n + j] * scale[j] * step[j]; model_change -= jd * (r[i] + jd / 2.0); }
      valid = model_change > 0.0;
    }
    if (!valid) { tr.Reject(); trace.push_back({cost, 0.0, gmax, 0.0, 0.0, tr.radius, 0}); last_successful = false; if (trace.size() > 400) break; continue; }
    double step_norm = 0, x_norm = 0;
    for (int j = 0; j < n; ++j) { delta[j] = step[j] * scale[j]; step_norm += delta[j] * delta[j]; x_norm += x[j] * x[j]; xc[j] = x[j] + delta[j]; }
    step_norm = std::sqrt(step_norm); x_norm = std::sqrt(x_norm);
    if (step_norm <= opt.parameter_tolerance * (x_norm + opt.parameter_tolerance)) break;
    eval(xc.data(), rc.data(), Jc.data());
    const double ccost = cost_of(rc), cost_change = cost - ccost;
    if (std::fabs(cost_change) <= opt.function_tolerance * cost) {      // (Ceres records the iteration, then stops)
      trace.push_back({ccost, cost_change, gmax, step_norm, cost_change / model_change, tr.radius, 1});
      for (int j = 0; j < n; ++j) x[j] = xc[j];
      break;
    }
    const double rel = cost_change / model_change;
    if (rel > opt.min_relative_decrease) {
      for (int j = 0; j < n; ++j) x[j] = xc[j];
      r.swap(rc); J.swap(Jc);
      cost = ccost; gmax = gradient();
      tr.Accept(rel, opt.max_trust_region_radius);
      reuse_diagonal = false; last_successful = true;
      trace.push_back({cost, cost_change, gmax, step_norm, rel, tr.radius, 1});
    } else {
      tr.Reject(); last_successful = false;
      trace.push_back({cost, cost_change, gmax, step_norm, rel, tr.radius, 0});
    }
  }
  return trace;
}

// Powell's function as Ceres' examples/powell.cc states it: f1 = x1 + 10 x2, f2 = sqrt(5) (x3 - x4), f3 = (x2 - 2 x3)^2, f4 = sqrt(10) (x1 - x4)^2; start (3, -1, 0, 1)
inline std::vector<Iteration> PowellTrace(double x[4], const Options& opt) {
  auto eval = [](const double* p, double* r, double* J) {
    const double s5 = std::sqrt(5.0), s10 = std::sqrt(10.0);
    r[0] = p[0] + 10.0 * p[1]; r[1] = s5 * (p[2] - p[3]); r[2] = (p[1] - 2.0 * p[2]) * (p[1] - 2.0 * p[2]); r[3] = s10 * (p[0] - p[3]) * (p[0] - p[3]);
    for (int i = 0; i < 16; ++i) J[i] = 0.0;
    J[0] = 1.0; J[1] = 10.0;
    J[4 + 2] = s5; J[4 + 3] = -s5;
    J[8 + 1] = 2.0 * (p[1] - 2.0 * p[2]); J[8 + 2] = -4.0 * (p[1] - 2.0 * p[2]);
    J[12 + 0] = 2.0 * s10 * (p[0] - p[3]); J[12 + 3] = -2.0 * s10 * (p[0] - p[3]);
  };
  return DenseLevenbergMarquardt(4, 4, eval, x, opt);
}

// Ceres' examples/helloworld.cc: one residual f = 10 - x, start x = 0.5
inline std::vector<Iteration> HelloWorldTrace(double x[1], const Options& opt) {
  auto eval = [](const double* p, double* r, double* J) { r[0] = 10.0 - p[0]; J[0] = -1.0; };
  return DenseLevenbergMarquardt(1, 1, eval, x, opt);
}

}  // namespace lm
}  // namespace oracle

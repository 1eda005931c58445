// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// CPU restatement of the robust track triangulation:
//   TriangulateMultiViewPoint (K x 4 null vector of [l_i^T P_i])          reference src/base/triangulation.cc:41-57
//   TriangulationEstimator::{Estimate, Residuals}, EstimateTriangulation  src/estimators/triangulation.cc:55-149
//   LORANSAC<E, L, InlierSupportMeasurer, CombinationSampler>::Estimate    src/optim/loransac.h:88-235
//   CombinationSampler (lexicographic k-combinations)                      src/optim/combination_sampler.cc:41-70, util/math.h:140-175
//   CalculateNormalizedLineAngularError, CalculateSquaredLineReprojectionError   src/base/projection.cc:161-203, 238-262
// JacobiSVD(...).matrixV().col(3) (Eigen, absent) == eigenvector of A^T A for its smallest eigenvalue (cyclic Jacobi).
#pragma once
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <vector>

#include "camera_models.h"
#include "init_solvers.h"   // SymmetricEigen
#include "ransac.h"         // RansacOptions, ComputeNumTrials, Support

namespace oracle {

struct TriView { double P[12]; double center[3]; int model; const double* params; double width, height; };

inline bool TriangulateMultiView(const std::vector<const TriView*>& views, const std::vector<const double*>& lines, double xyz[3]) {
  double AtA[16] = {0};
  for (size_t i = 0; i < views.size(); ++i) {
    const double* P = views[i]->P; const double* l = lines[i];
    double row[4];
    for (int c = 0; c < 4; ++c) row[c] = l[0] * P[c] + l[1] * P[4 + c] + l[2] * P[8 + c];
    for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) AtA[4 * r + c] += row[r] * row[c];
  }
  double w[4], V[16];
  SymmetricEigen(4, AtA, w, V);
  for (int i = 0; i < 3; ++i) xyz[i] = V[4 * i] / V[12];
  return true;
}
inline double ProjZ(const double* P, const double X[3]) { return P[8] * X[0] + P[9] * X[1] + P[10] * X[2] + P[11]; }

// projection.cc:161-203 with the projection matrix given
inline double SquaredLineReprojectionErrorP(const double l[3], const double X[3], const TriView& v) {
  const double pz = ProjZ(v.P, X);
  if (pz < DBL_EPSILON) return DBL_MAX;
  const double px = v.P[0] * X[0] + v.P[1] * X[1] + v.P[2] * X[2] + v.P[3], py = v.P[4] * X[0] + v.P[5] * X[1] + v.P[6] * X[2] + v.P[7];
  const double inv = 1.0 / pz, u = inv * px, w = inv * py;
  const double alpha = l[0] * u + l[1] * w + l[2];
  double ix, iy, jx, jy;
  WorldToImage<double>(v.model, v.params, u, w, &ix, &iy);
  if (!(ix >= 0 && ix < v.width && iy >= 0 && iy < v.height)) return DBL_MAX;
  WorldToImage<double>(v.model, v.params, u - l[0] * alpha, w - l[1] * alpha, &jx, &jy);
  return (ix - jx) * (ix - jx) + (iy - jy) * (iy - jy);
}
// projection.cc:238-262
inline double NormalizedLineAngularError(const double l[3], const double X[3], const TriView& v) {
  const double nl = std::sqrt(l[0] * l[0] + l[1] * l[1] + l[2] * l[2]);
  const double r0 = v.P[0] * X[0] + v.P[1] * X[1] + v.P[2] * X[2] + v.P[3], r1 = v.P[4] * X[0] + v.P[5] * X[1] + v.P[6] * X[2] + v.P[7], r2 = ProjZ(v.P, X);
  if (r2 < 0) return DBL_MAX;
  double ix, iy;
  WorldToImage<double>(v.model, v.params, r0 / r2, r1 / r2, &ix, &iy);
  if (ix < 0 || ix >= v.width || iy < 0 || iy >= v.height) return DBL_MAX;
  const double nr = std::sqrt(r0 * r0 + r1 * r1 + r2 * r2);
  const double d = (l[0] * r0 + l[1] * r1 + l[2] * r2) / (nl * nr);
  return std::fabs(M_PI_2 - std::acos(std::fabs(d)));
}
inline double TriAngle(const double c1[3], const double c2[3], const double X[3]) {
  double b2 = 0, r1 = 0, r2 = 0;
  for (int i = 0; i < 3; ++i) { b2 += (c1[i] - c2[i]) * (c1[i] - c2[i]); r1 += (X[i] - c1[i]) * (X[i] - c1[i]); r2 += (X[i] - c2[i]) * (X[i] - c2[i]); }
  const double den = 2.0 * std::sqrt(r1 * r2);
  if (den == 0.0) return 0.0;
  const double a = std::fabs(std::acos((r1 + r2 - b2) / den));
  return std::fmin(a, M_PI - a);
}

struct TriangulationOptions { double min_tri_angle = 0.0; int residual_type = 0; /* 0 ANGULAR_ERROR, 1 REPROJECTION_ERROR */ RansacOptions ransac; };
struct TriangulationReport { bool success = false; uint64_t num_trials = 0; Support support; std::vector<char> inlier_mask; double xyz[3] = {0, 0, 0}; };

// one track: n observations (line i seen in view views[i])
inline TriangulationReport EstimateTriangulation(const TriangulationOptions& opt_in, int n, const double* lines /*n x 3*/, const TriView* const* views) {
  TriangulationReport report;
  if (n < 3) return report;
  const int kMin = 3;
  RansacOptions opt = opt_in.ransac;
  {  // RANSAC ctor (ransac.h:149-155)
    const uint64_t kNumSamples = 100000;
    opt.max_num_trials = std::min(opt.max_num_trials, ComputeNumTrials(static_cast<uint64_t>(opt.min_inlier_ratio * kNumSamples), kNumSamples, opt.confidence,
                                                                        opt.dyn_num_trials_multiplier, kMin));
  }
  auto estimate = [&](const std::vector<int>& idx, double xyz[3]) -> bool {   // TriangulationEstimator::Estimate
    std::vector<const TriView*> vs; std::vector<const double*> ls;
    for (int i : idx) { vs.push_back(views[i]); ls.push_back(lines + 3 * i); }
    TriangulateMultiView(vs, ls, xyz);
    for (int i : idx) if (!(ProjZ(views[i]->P, xyz) >= DBL_EPSILON)) return false;
    for (size_t a = 0; a < idx.size(); ++a)
      for (size_t b = 0; b < a; ++b)
        if (TriAngle(views[idx[a]]->center, views[idx[b]]->center, xyz) >= opt_in.min_tri_angle) return true;
    return false;
  };
  auto residuals = [&](const double xyz[3], std::vector<double>* r) {
    r->resize(n);
    for (int i = 0; i < n; ++i) {
      if (opt_in.residual_type == 1) (*r)[i] = SquaredLineReprojectionErrorP(lines + 3 * i, xyz, *views[i]);
      else { const double a = NormalizedLineAngularError(lines + 3 * i, xyz, *views[i]); (*r)[i] = a * a; }
    }
  };
  auto evaluate = [&](const std::vector<double>& r, double maxr) {   // InlierSupportMeasurer::Evaluate
    Support s; s.num_inliers = 0; s.residual_sum = 0;
    for (double v : r) if (v <= maxr) { ++s.num_inliers; s.residual_sum += v; }
    return s;
  };
  auto better = [](const Support& a, const Support& b) { return a.num_inliers > b.num_inliers || (a.num_inliers == b.num_inliers && a.residual_sum < b.residual_sum); };
  const double max_residual = opt.max_error * opt.max_error;
  Support best; double best_model[3] = {0, 0, 0};
  bool abort = false;
  std::vector<double> r;
  // CombinationSampler: 3-combinations of 0..n-1 in lexicographic order, wrapping around
  int comb[3] = {0, 1, 2};
  uint64_t nck = (uint64_t)n * (n - 1) * (n - 2) / 6;
  uint64_t max_num_trials = std::min<uint64_t>(opt.max_num_trials, nck);
  uint64_t dyn = max_num_trials;
  for (report.num_trials = 0; report.num_trials < max_num_trials; ++report.num_trials) {
    if (abort) { report.num_trials += 1; break; }
    const std::vector<int> sample = {comb[0], comb[1], comb[2]};
    {  // next combination
      if (comb[2] + 1 < n) ++comb[2];
      else if (comb[1] + 2 < n) { ++comb[1]; comb[2] = comb[1] + 1; }
      else if (comb[0] + 3 < n) { ++comb[0]; comb[1] = comb[0] + 1; comb[2] = comb[0] + 2; }
      else { comb[0] = 0; comb[1] = 1; comb[2] = 2; }
    }
    double xyz[3];
    if (!estimate(sample, xyz)) continue;
    residuals(xyz, &r);
    const Support s = evaluate(r, max_residual);
    if (better(s, best)) {
      best = s; for (int i = 0; i < 3; ++i) best_model[i] = xyz[i];
      if (s.num_inliers > (uint64_t)kMin && s.num_inliers >= (uint64_t)kMin) {
        std::vector<int> inl;
        for (int i = 0; i < n; ++i) if (r[i] <= max_residual) inl.push_back(i);
        double lxyz[3];
        if (estimate(inl, lxyz)) {
          residuals(lxyz, &r);
          const Support ls = evaluate(r, max_residual);
          if (better(ls, best)) { best = ls; for (int i = 0; i < 3; ++i) best_model[i] = lxyz[i]; }
        }
      }
      dyn = ComputeNumTrials(best.num_inliers, n, opt.confidence, opt.dyn_num_trials_multiplier, kMin);
    }
    if (report.num_trials >= dyn && report.num_trials >= opt.min_num_trials) { abort = true; }
  }
  report.support = best;
  for (int i = 0; i < 3; ++i) report.xyz[i] = best_model[i];
  if (best.num_inliers < (uint64_t)kMin) return report;
  report.success = true;
  residuals(best_model, &r);
  report.inlier_mask.resize(n);
  for (int i = 0; i < n; ++i) report.inlier_mask[i] = r[i] <= max_residual;
  return report;
}

}  // namespace oracle

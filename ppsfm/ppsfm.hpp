// ppsfm.hpp — header-only C++ host mirror of the reference's extension points on top of the C ABI
// (include/ppsfm_hip.h).  No Eigen / Ceres / glog needed: matrices are plain std::array, errors are
// exceptions carrying pp_last_error().  Where Eigen exists the same layouts map 1:1
// (Matrix3x4d row-major copy, Vector3d data()).
//
//   ppsfm::P6LEstimator          <-> colmap::P6LEstimator   (reference src/estimators/absolute_pose.h:48-80)
//        the colmap Estimator concept used by RANSAC<E> (src/optim/ransac.h:146,191,204-205,223-228):
//        typedefs X_t/Y_t/M_t, kMinNumSamples, Estimate(X, Y), Residuals(X, Y, M, &residuals)
//   ppsfm::AbsolutePoseFromLinesRANSAC <-> RANSAC<P6LEstimator> (src/estimators/pose.cc:48), Estimate(X, Y) -> Report
//   ppsfm::EstimateAbsolutePoseFromLines  <-> src/estimators/pose.h:110-115
//   ppsfm::BundleAdjustmentProblem         <-> the flat form of what BundleAdjuster::SetUp builds
//        (src/optim/bundle_adjustment.cc:326-542); Solve() replaces ceres::Solve (:306)
#pragma once
#include <array>
#include <cmath>
#include <cstdint>
#include <limits>
#include <stdexcept>
#include <string>
#include <vector>

#include "../include/ppsfm_hip.h"

namespace ppsfm {

struct Error : std::runtime_error {
  int code;
  Error(int c, const std::string& what) : std::runtime_error(what), code(c) {}
};
inline void Check(int rc) {
  if (rc != PP_OK) throw Error(rc, pp_last_error());
}

using Vector3d = std::array<double, 3>;
using Vector4d = std::array<double, 4>;
using Matrix3x4d = std::array<double, 12>;  // row-major

// feature/types.h:98-138
struct FeatureLine {
  Vector3d line{{0, 0, 0}};
  bool is_aligned = false;
  uint64_t point3D_id = std::numeric_limits<uint64_t>::max();
  const Vector3d& Line() const { return line; }
  bool IsAligned() const { return is_aligned; }
};
using FeatureLines = std::vector<FeatureLine>;

// optim/ransac.h:47-76
struct RANSACOptions {
  double max_error = 0.0;
  double min_inlier_ratio = 0.1;
  double confidence = 0.99;
  double dyn_num_trials_multiplier = 3.0;
  size_t min_num_trials = 0;
  size_t max_num_trials = std::numeric_limits<size_t>::max();
  void Check() const {
    if (!(max_error > 0) || min_inlier_ratio < 0 || min_inlier_ratio > 1 || confidence < 0 || confidence > 1 ||
        min_num_trials > max_num_trials)
      throw Error(PP_ERR_INVALID, "RANSACOptions::Check failed");
  }
};

namespace detail {
class PoseHandle {
 public:
  PoseHandle(const FeatureLines& X, const std::vector<Vector3d>& Y, int device) {
    if (X.size() != Y.size()) throw Error(PP_ERR_INVALID, "CHECK_EQ(X.size(), Y.size())");
    std::vector<double> l(3 * X.size()), p(3 * X.size());
    std::vector<uint8_t> a(X.size());
    for (size_t i = 0; i < X.size(); ++i) {
      for (int c = 0; c < 3; ++c) { l[3 * i + c] = X[i].line[c]; p[3 * i + c] = Y[i][c]; }
      a[i] = X[i].is_aligned ? 1 : 0;
    }
    Check(pp_pose_create(static_cast<int32_t>(X.size()), l.data(), p.data(), a.data(), device, &h_));
  }
  ~PoseHandle() { pp_pose_destroy(h_); }
  PoseHandle(const PoseHandle&) = delete;
  PoseHandle& operator=(const PoseHandle&) = delete;
  pp_pose_handle get() const { return h_; }

 private:
  pp_pose_handle h_ = nullptr;
};
}  // namespace detail

class P6LEstimator {
 public:
  typedef FeatureLine X_t;
  typedef Vector3d Y_t;
  typedef Matrix3x4d M_t;
  static const int kMinNumSamples = 6;
  int device = 0;

  std::vector<M_t> Estimate(const std::vector<X_t>& lines2D, const std::vector<Y_t>& points3D) const {
    if (lines2D.size() != 6 || points3D.size() != 6) throw Error(PP_ERR_INVALID, "P6LEstimator::Estimate needs 6 pairs");
    detail::PoseHandle h(lines2D, points3D, device);
    const uint32_t sample[6] = {0, 1, 2, 3, 4, 5};
    double models[96];
    int32_t n = 0;
    Check(pp_pose_p6l_batch(h.get(), 1, sample, models, &n));
    std::vector<M_t> out(n);
    for (int k = 0; k < n; ++k)
      for (int e = 0; e < 12; ++e) out[k][e] = models[12 * k + e];
    return out;
  }
  void Residuals(const std::vector<X_t>& lines2D, const std::vector<Y_t>& points3D, const M_t& proj_matrix,
                 std::vector<double>* residuals) const {
    detail::PoseHandle h(lines2D, points3D, device);
    residuals->resize(lines2D.size());
    Check(pp_pose_residuals(h.get(), 1, proj_matrix.data(), residuals->data()));
  }
};

// RANSAC<P6LEstimator, InlierSupportMeasurer, RandomSampler>
class AbsolutePoseFromLinesRANSAC {
 public:
  struct Support {
    size_t num_inliers = 0;
    double residual_sum = std::numeric_limits<double>::max();
  };
  struct Report {
    bool success = false;
    size_t num_trials = 0;
    Support support;
    std::vector<char> inlier_mask;
    Matrix3x4d model{};
  };
  explicit AbsolutePoseFromLinesRANSAC(const RANSACOptions& options, unsigned seed = 0, int device = 0)
      : options_(options), seed_(seed), device_(device) {
    options.Check();
  }
  Report Estimate(const FeatureLines& X, const std::vector<Vector3d>& Y) const {
    detail::PoseHandle h(X, Y, device_);
    pp_ransac_options o;
    pp_ransac_options_default(&o);
    o.max_error = options_.max_error; o.min_inlier_ratio = options_.min_inlier_ratio; o.confidence = options_.confidence;
    o.dyn_num_trials_multiplier = options_.dyn_num_trials_multiplier;
    o.min_num_trials = options_.min_num_trials; o.max_num_trials = options_.max_num_trials; o.seed = seed_;
    pp_ransac_report r;
    std::vector<uint8_t> mask(X.size() ? X.size() : 1);
    Check(pp_pose_ransac(h.get(), &o, &r, mask.data()));
    Report rep;
    rep.success = r.success != 0; rep.num_trials = r.num_trials;
    rep.support.num_inliers = r.num_inliers; rep.support.residual_sum = r.residual_sum;
    for (int e = 0; e < 12; ++e) rep.model[e] = r.model[e];
    if (rep.success) rep.inlier_mask.assign(mask.begin(), mask.begin() + X.size());
    return rep;
  }
  P6LEstimator estimator;

 private:
  RANSACOptions options_;
  unsigned seed_;
  int device_;
};

// base/pose.cc:41-51 (Eigen::Quaterniond(rot_mat) -> (w,x,y,z))
inline Vector4d RotationMatrixToQuaternion(const Matrix3x4d& P) {
  auto R = [&](int r, int c) { return P[4 * r + c]; };
  Vector4d q{{0, 0, 0, 0}};
  double t = R(0, 0) + R(1, 1) + R(2, 2);
  if (t > 0) {
    t = std::sqrt(t + 1.0);
    q[0] = 0.5 * t; t = 0.5 / t;
    q[1] = (R(2, 1) - R(1, 2)) * t; q[2] = (R(0, 2) - R(2, 0)) * t; q[3] = (R(1, 0) - R(0, 1)) * t;
  } else {
    int i = 0;
    if (R(1, 1) > R(0, 0)) i = 1;
    if (R(2, 2) > R(i, i)) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    t = std::sqrt(R(i, i) - R(j, j) - R(k, k) + 1.0);
    q[1 + i] = 0.5 * t; t = 0.5 / t;
    q[0] = (R(k, j) - R(j, k)) * t; q[1 + j] = (R(j, i) + R(i, j)) * t; q[1 + k] = (R(k, i) + R(i, k)) * t;
  }
  return q;
}

// estimators/pose.h:110-115, pose.cc:52-94
inline bool EstimateAbsolutePoseFromLines(const RANSACOptions& options, const FeatureLines& lines2D,
                                          const std::vector<Vector3d>& points3D, Vector4d* qvec, Vector3d* tvec,
                                          size_t* num_inliers, std::vector<char>* inlier_mask, unsigned seed = 0, int device = 0) {
  options.Check();
  AbsolutePoseFromLinesRANSAC ransac(options, seed, device);
  const auto report = ransac.Estimate(lines2D, points3D);
  *num_inliers = report.support.num_inliers;
  *inlier_mask = report.inlier_mask;
  if (*num_inliers == 0) return false;
  size_t num_aligned_inliers = 0;
  for (size_t i = 0; i < lines2D.size() && i < inlier_mask->size(); ++i)
    if ((*inlier_mask)[i] && lines2D[i].IsAligned()) ++num_aligned_inliers;
  if (num_aligned_inliers > *num_inliers * 0.9) return false;
  *qvec = RotationMatrixToQuaternion(report.model);
  *tvec = Vector3d{{report.model[3], report.model[7], report.model[11]}};
  for (double v : *qvec) if (std::isnan(v)) return false;
  for (double v : *tvec) if (std::isnan(v)) return false;
  return true;
}

// Flat bundle-adjustment problem (what BundleAdjuster::SetUp produces) + the solve.
class BundleAdjustmentProblem {
 public:
  BundleAdjustmentProblem(const pp_ba_problem_desc& desc, int device = 0) : C_(desc.num_poses), P_(desc.num_points), K_(desc.num_cameras) {
    Check(pp_ba_create(&desc, device, &h_));
  }
  ~BundleAdjustmentProblem() { pp_ba_destroy(h_); }
  BundleAdjustmentProblem(const BundleAdjustmentProblem&) = delete;
  BundleAdjustmentProblem& operator=(const BundleAdjustmentProblem&) = delete;
  void SetParameters(const double* poses, const double* points, const double* intr) { Check(pp_ba_set_parameters(h_, poses, points, intr)); }
  void GetParameters(double* poses, double* points, double* intr) { Check(pp_ba_get_parameters(h_, poses, points, intr)); }
  // returns false when the solver reports FAILURE (ceres::Solver::Summary::IsSolutionUsable() == false)
  bool Solve(const pp_ba_options& options, pp_ba_summary* summary) {
    const int rc = pp_ba_solve(h_, &options, summary);
    if (rc == PP_ERR_NUMERIC) return false;
    Check(rc);
    return true;
  }
  pp_ba_handle handle() const { return h_; }

 private:
  pp_ba_handle h_ = nullptr;
  int C_, P_, K_;
};

}  // namespace ppsfm

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
//   ppsfm::init::initialize_reconstruction <-> src/init/initializer.h:103-108 (four-view initialisation: gravity
//        alignment, FourView2dEstimator LO-MSAC, lifting, PlanarOffsetEstimator LO-MSAC; both runs on the device)
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
  // (a descriptor that does not say otherwise gets PP_ORDERING_AUTO: the reference's BundleAdjuster::Solve hands the camera ordering to Ceres' SPARSE_SCHUR without
  // being asked, bundle_adjustment.cc:279-282; the handles of a point-sharded group pass PP_ORDERING_NATURAL or a union co-visibility)
  BundleAdjustmentProblem(const pp_ba_problem_desc& desc, int device = 0) : C_(desc.num_poses), P_(desc.num_points), K_(desc.num_cameras) {
    pp_ba_problem_desc d = desc;
    if (d.ordering == PP_ORDERING_DEFAULT) d.ordering = PP_ORDERING_AUTO;
    Check(pp_ba_create(&d, device, &h_));
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


// ---- four-view initialisation (src/init/initializer.cc:58-216) ------------------------------------------------------
namespace init {

using Pose = Matrix3x4d;
struct InitOptions {              // initializer.h:48-58
  double min_tri_angle = 0.1;     // "Minimum mean triangulation angle (in rad)"
  double min_num_inliers = 6;
  double max_error = 0.005;       // in normalised coordinates
};

namespace detail {
// Eigen::Quaterniond::FromTwoVectors(a, b).toRotationMatrix(), row-major 3x3
inline std::array<double, 9> FromTwoVectors(const Vector3d& a, const Vector3d& b) {
  auto norm = [](const Vector3d& v) { return std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); };
  const double na = norm(a), nb = norm(b);
  const Vector3d v0{{a[0] / na, a[1] / na, a[2] / na}}, v1{{b[0] / nb, b[1] / nb, b[2] / nb}};
  const double c = v0[0] * v1[0] + v0[1] * v1[1] + v0[2] * v1[2];
  double w, x, y, z;
  if (c < -1.0 + 1e-12) {         // antiparallel: half turn about an axis orthogonal to a
    Vector3d e{{1, 0, 0}};
    if (std::fabs(v0[0]) > 0.9) e = Vector3d{{0, 1, 0}};
    Vector3d ax{{v0[1] * e[2] - v0[2] * e[1], v0[2] * e[0] - v0[0] * e[2], v0[0] * e[1] - v0[1] * e[0]}};
    const double n = norm(ax);
    w = 0; x = ax[0] / n; y = ax[1] / n; z = ax[2] / n;
  } else {
    const double s = std::sqrt((1.0 + c) * 2.0);
    w = 0.5 * s;
    x = (v0[1] * v1[2] - v0[2] * v1[1]) / s; y = (v0[2] * v1[0] - v0[0] * v1[2]) / s; z = (v0[0] * v1[1] - v0[1] * v1[0]) / s;
  }
  return {{1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w), 2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
           2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)}};
}
struct Handles {
  pp_fourview2d_handle fv = nullptr;
  pp_planar_handle pl = nullptr;
  ~Handles() { pp_fourview2d_destroy(fv); pp_planar_destroy(pl); }
};
}  // namespace detail

// lines[i], gravity[i]: the four views.  Returns false exactly where the reference does (too few inliers, mean
// triangulation angle — computed in degrees and compared with min_tri_angle as the reference does, :186-190 —, and the
// planar-offset stage's checks :209-215).
inline bool initialize_reconstruction(const std::vector<FeatureLines>& lines, const std::vector<Vector3d>& gravity, const InitOptions& options,
                                      std::vector<Pose>* output, double* inlier_ratio, int device = 0) {
  *inlier_ratio = 0;
  if (lines.size() != 4 || gravity.size() != 4) throw Error(PP_ERR_INVALID, "initialize_reconstruction: four views expected");
  std::vector<double> x[4], lr[4];
  std::array<double, 36> Rg;
  for (int i = 0; i < 4; ++i) {
    const std::array<double, 9> R = detail::FromTwoVectors(gravity[i], Vector3d{{0.0, 1.0, 0.0}});
    for (int e = 0; e < 9; ++e) Rg[9 * i + e] = R[e];
    for (const FeatureLine& f : lines[i]) {
      const Vector3d& l0 = f.Line();
      if (f.IsAligned()) {       // only the aligned lines are pre-rotated (:76-90)
        const Vector3d l{{R[0] * l0[0] + R[1] * l0[1] + R[2] * l0[2], R[3] * l0[0] + R[4] * l0[1] + R[5] * l0[2], R[6] * l0[0] + R[7] * l0[1] + R[8] * l0[2]}};
        if (std::fabs(l[1]) > 1e-6) throw Error(PP_ERR_INVALID, "initialize_reconstruction: CHECK_NEAR(l(1), 0.0, 1e-6)");
        double a = l[2], b = -l[0];
        if (b < 0) { a = -a; b = -b; }
        const double n = std::sqrt(a * a + b * b);
        x[i].push_back(a / n); x[i].push_back(b / n);
      } else {
        lr[i].insert(lr[i].end(), l0.begin(), l0.end());
      }
    }
  }
  for (int i = 1; i < 4; ++i)
    if (x[i].size() != x[0].size() || lr[i].size() != lr[0].size()) throw Error(PP_ERR_INVALID, "initialize_reconstruction: CHECK_EQ on the track counts");
  const int n2 = (int)(x[0].size() / 2), n3 = (int)(lr[0].size() / 3);
  if (n2 < 5 || n3 < 3) return false;
  std::vector<double> xs, ls;
  for (int i = 0; i < 4; ++i) { xs.insert(xs.end(), x[i].begin(), x[i].end()); ls.insert(ls.end(), lr[i].begin(), lr[i].end()); }

  pp_lomsac_options ro;
  pp_lomsac_options_default(&ro);
  ro.final_least_squares = 1; ro.min_num_iterations = 1000; ro.squared_inlier_threshold = options.max_error;
  detail::Handles hs;
  Check(pp_fourview2d_create(n2, xs.data(), device, &hs.fv));
  pp_lomsac_report rep;
  std::array<double, 24> c2;
  std::vector<double> X2((size_t)2 * n2);
  std::vector<int32_t> inl(n2);
  Check(pp_fourview2d_lomsac(hs.fv, &ro, nullptr, &rep, c2.data(), X2.data(), inl.data()));
  if (rep.best_num_inliers < options.min_num_inliers) return false;

  double angle_sum = 0;             // mean minimum triangulation angle over the first three cameras (:157-190)
  double ctr[3][2];
  for (int c = 0; c < 3; ++c) {
    const double* P = &c2[6 * c];
    ctr[c][0] = -(P[0] * P[2] + P[3] * P[5]); ctr[c][1] = -(P[1] * P[2] + P[4] * P[5]);
  }
  for (int k = 0; k < rep.num_inlier_indices; ++k) {
    const int i = inl[k];
    double best = std::numeric_limits<double>::max();
    for (int c1 = 0; c1 < 3; ++c1)
      for (int c2i = c1 + 1; c2i < 3; ++c2i) {
        const double v1[2] = {ctr[c1][0] - X2[2 * i], ctr[c1][1] - X2[2 * i + 1]}, v2[2] = {ctr[c2i][0] - X2[2 * i], ctr[c2i][1] - X2[2 * i + 1]};
        const double d = (v1[0] * v2[0] + v1[1] * v2[1]) / (std::sqrt(v1[0] * v1[0] + v1[1] * v1[1]) * std::sqrt(v2[0] * v2[0] + v2[1] * v2[1]));
        best = std::min(best, std::acos(std::max(-1.0, std::min(1.0, d))));
      }
    angle_sum += best;
  }
  const double mean_tri_angle = (angle_sum / rep.num_inlier_indices) / M_PI * 180.0;
  if (mean_tri_angle < options.min_tri_angle) return false;

  std::array<double, 48> poses{};   // lift_camera (:45-56)
  for (int c = 0; c < 4; ++c) {
    const double* P = &c2[6 * c];
    double* Q = &poses[12 * c];
    Q[0] = P[0]; Q[2] = P[1]; Q[8] = P[3]; Q[10] = P[4]; Q[5] = 1.0; Q[3] = P[2]; Q[11] = P[5];
  }
  Check(pp_planar_create(n3, poses.data(), ls.data(), Rg.data(), device, &hs.pl));
  pp_lomsac_report rep3;
  std::array<double, 3> ty;
  std::array<double, 48> cams3;
  std::vector<int32_t> inl3(n3);
  Check(pp_planar_lomsac(hs.pl, &ro, &rep3, ty.data(), cams3.data(), inl3.data()));
  if (rep3.best_num_inliers < options.min_tri_angle) return false;     // sic (:209-210)
  output->resize(4);
  for (int c = 0; c < 4; ++c) for (int e = 0; e < 12; ++e) (*output)[c][e] = cams3[12 * c + e];
  *inlier_ratio = rep3.inlier_ratio;
  return rep3.best_num_inliers >= options.min_num_inliers;
}

}  // namespace init

}  // namespace ppsfm

// ransaclib_solvers.hpp — the RansacLib *Solver concept* on top of the C ABI (include/ppsfm_hip.h), so that the
// reference's own driver  ransac_lib::LocallyOptimizedMSAC<Model, ModelVector, Solver>  (reference
// lib/RansacLib/RansacLib/ransac.h:118-428) runs unchanged over device-side solvers.
//
// The concept, as the driver uses it (ransac.h:134-135, 181, 296, 314, 345, 383, 412-418):
//     int  min_sample_size() const;            int non_minimal_sample_size() const;      int num_data() const;
//     int  MinimalSolver(const std::vector<int>& sample, ModelVector* models) const;
//     int  NonMinimalSolver(const std::vector<int>& sample, Model* model) const;
//     double EvaluateModelOnPoint(const Model& model, int i) const;
//     void LeastSquares(const std::vector<int>& sample, Model* model) const;
// and the reference's three models of it:
//     init::PlanarOffsetEstimator      reference src/init/initializer.h:72-93   ->  ppsfm::init::PlanarOffsetSolver
//     init::FourView2dEstimator        reference src/init/sfm2d.h:48-97         ->  ppsfm::init::FourView2dSolver
//     init::AbsolutePose2dEstimator    reference src/init/sfm2d.h:99-143        ->  ppsfm::init::AbsolutePose2dSolver
//
// How the per-point interface meets a batch device: the driver scores a model with num_data() calls of
// EvaluateModelOnPoint (ScoreModel / GetInliers, ransac.h:291-332).  The adaptors evaluate a model on ALL points in one
// launch (pp_*_evaluate) the first time the driver asks about it and answer the following calls from that vector; a small
// cache keyed by the model's camera block keeps the few models the driver alternates between (best, candidate, LO model).
// A Reconstruction model carries its triangulated points, as the reference's does.
//
// This is the "keep RansacLib's driver" integration level of INTEGRATION.md 3; the whole run on the device is
// pp_planar_lomsac / pp_fourview2d_lomsac / pp_pose2d_lomsac.  No Eigen needed: matrices are std::array, row-major.
#pragma once
#include <array>
#include <cstring>
#include <vector>

#include "ppsfm.hpp"

namespace ppsfm {
namespace init {

using Vector2d = std::array<double, 2>;
using Pose2d = std::array<double, 6>;   // 2x3 row-major (Eigen::Matrix<double,2,3> of sfm2d.h:43)

namespace detail {
// errors of the last few models, keyed by the bytes of the model's camera block
template <size_t kKeyDoubles>
class ErrorCache {
 public:
  template <typename Fill>      // Fill(std::vector<double>* errors) evaluates the model on all points
  double Get(const double* key, int i, Fill&& fill) const {
    for (int s = 0; s < kSlots; ++s)
      if (valid_[s] && std::memcmp(keys_[s].data(), key, sizeof(double) * kKeyDoubles) == 0) return errors_[s][i];
    const int s = next_;
    next_ = (next_ + 1) % kSlots;
    std::memcpy(keys_[s].data(), key, sizeof(double) * kKeyDoubles);
    fill(&errors_[s]);
    valid_[s] = true;
    return errors_[s][i];
  }
  // a model whose errors are already known (the solver evaluated it to fill in its points)
  // (an entry with the same key is replaced: a stale one must not win the lookup)
  void Put(const double* key, std::vector<double>&& errors) const {
    int s = -1;
    for (int q = 0; q < kSlots; ++q)
      if (valid_[q] && std::memcmp(keys_[q].data(), key, sizeof(double) * kKeyDoubles) == 0) s = q;
    if (s < 0) { s = next_; next_ = (next_ + 1) % kSlots; }
    std::memcpy(keys_[s].data(), key, sizeof(double) * kKeyDoubles);
    errors_[s] = std::move(errors);
    valid_[s] = true;
  }

 private:
  static const int kSlots = 20;      // a minimal sample of FourView2dSolver yields up to 16 candidates, scored one after the other
  mutable std::array<std::array<double, kKeyDoubles>, kSlots> keys_{};
  mutable std::array<std::vector<double>, kSlots> errors_;
  mutable std::array<bool, kSlots> valid_{};
  mutable int next_ = 0;
};
}  // namespace detail

// ---- PlanarOffsetEstimator (reference src/init/initializer.h:72-98, initializer.cc:219-333, 447-451) ------------------
class PlanarOffsetSolver {
 public:
  struct Reconstruction {
    std::vector<Pose> cams;          // 4 x (3x4 row-major)
    std::vector<Vector3d> X;         // num_data() triangulated points
  };
  typedef std::vector<Reconstruction> ReconstructionVector;

  // poses: the four lifted cameras (t_y = 0); lines[v][i]: line of track i in view v; Rg[v]: 3x3 row-major gravity rotations
  PlanarOffsetSolver(const std::vector<Pose>& poses, const std::vector<std::vector<Vector3d>>& lines, const std::vector<std::array<double, 9>>& Rg,
                     double inlier_threshold, int device = 0)
      : n_(lines.empty() ? 0 : (int)lines[0].size()), inlier_threshold_(inlier_threshold) {
    if (poses.size() != 4 || lines.size() != 4 || Rg.size() != 4) throw Error(PP_ERR_INVALID, "PlanarOffsetSolver: four views expected");
    std::vector<double> p(48), l((size_t)12 * n_), r(36);
    for (int v = 0; v < 4; ++v) {
      std::memcpy(&p[12 * v], poses[v].data(), sizeof(double) * 12);
      std::memcpy(&r[9 * v], Rg[v].data(), sizeof(double) * 9);
      if ((int)lines[v].size() != n_) throw Error(PP_ERR_INVALID, "PlanarOffsetSolver: CHECK_EQ on the track counts");
      for (int i = 0; i < n_; ++i) std::memcpy(&l[((size_t)v * n_ + i) * 3], lines[v][i].data(), sizeof(double) * 3);
    }
    Check(pp_planar_create(n_, p.data(), l.data(), r.data(), device, &h_));
  }
  ~PlanarOffsetSolver() { pp_planar_destroy(h_); }
  PlanarOffsetSolver(const PlanarOffsetSolver&) = delete;
  PlanarOffsetSolver& operator=(const PlanarOffsetSolver&) = delete;

  int min_sample_size() const { return 3; }
  int non_minimal_sample_size() const { return 20; }
  int num_data() const { return n_; }

  int MinimalSolver(const std::vector<int>& sample, ReconstructionVector* models) const {
    models->clear();
    Reconstruction rec;
    if (!Solve(sample, &rec)) return 0;
    models->push_back(std::move(rec));
    return 1;
  }
  int NonMinimalSolver(const std::vector<int>& sample, Reconstruction* model) const { return Solve(sample, model) ? 1 : 0; }
  double EvaluateModelOnPoint(const Reconstruction& model, int i) const {
    if (model.cams.size() != 4) return 1e5;      // default-constructed model (never happens in the driver)
    double key[3] = {model.cams[1][7], model.cams[2][7], model.cams[3][7]};     // the three out-of-plane offsets define the model
    return cache_.Get(key, i, [&](std::vector<double>* err) {
      err->resize(n_);
      Check(pp_planar_evaluate(h_, key, err->data(), nullptr, nullptr));
    });
  }
  // the reference's LeastSquares returns on its first line (initializer.cc:450-451): a no-op
  void LeastSquares(const std::vector<int>&, Reconstruction*) const {}
  pp_planar_handle handle() const { return h_; }

 private:
  bool Solve(const std::vector<int>& sample, Reconstruction* rec) const {
    double ty[3];
    std::vector<int32_t> s(sample.begin(), sample.end());
    Check(pp_planar_solve_batch(h_, 1, (int32_t)s.size(), s.data(), ty));
    if (!(ty[0] == ty[0] && ty[1] == ty[1] && ty[2] == ty[2])) return false;        // singular 3x3 system
    std::vector<double> err(n_), X((size_t)3 * n_);
    double cams[48];
    Check(pp_planar_evaluate(h_, ty, err.data(), X.data(), cams));
    rec->cams.resize(4); rec->X.resize(n_);
    for (int v = 0; v < 4; ++v) std::memcpy(rec->cams[v].data(), cams + 12 * v, sizeof(double) * 12);
    for (int i = 0; i < n_; ++i) rec->X[i] = Vector3d{{X[3 * i], X[3 * i + 1], X[3 * i + 2]}};
    const double key[3] = {rec->cams[1][7], rec->cams[2][7], rec->cams[3][7]};
    cache_.Put(key, std::move(err));
    return true;
  }
  pp_planar_handle h_ = nullptr;
  int n_;
  double inlier_threshold_;
  detail::ErrorCache<3> cache_;
};

// ---- AbsolutePose2dEstimator (reference src/init/sfm2d.h:99-143, sfm2d.cc:491-530) ------------------------------------
class AbsolutePose2dSolver {
 public:
  AbsolutePose2dSolver(const std::vector<Vector2d>& x, const std::vector<Vector2d>& X, int device = 0) : n_((int)x.size()) {
    if (x.size() != X.size()) throw Error(PP_ERR_INVALID, "AbsolutePose2dSolver: CHECK_EQ(x.size(), X.size())");
    std::vector<double> a((size_t)2 * n_), b((size_t)2 * n_);
    for (int i = 0; i < n_; ++i) { a[2 * i] = x[i][0]; a[2 * i + 1] = x[i][1]; b[2 * i] = X[i][0]; b[2 * i + 1] = X[i][1]; }
    Check(pp_pose2d_create(n_, a.data(), b.data(), device, &h_));       // normalises the bearings, as the reference's ctor does
  }
  ~AbsolutePose2dSolver() { pp_pose2d_destroy(h_); }
  AbsolutePose2dSolver(const AbsolutePose2dSolver&) = delete;
  AbsolutePose2dSolver& operator=(const AbsolutePose2dSolver&) = delete;

  int min_sample_size() const { return 3; }
  int non_minimal_sample_size() const { return 2 * min_sample_size(); }
  int num_data() const { return n_; }
  int MinimalSolver(const std::vector<int>& sample, std::vector<Pose2d>* models) const {
    Pose2d cam;
    NonMinimalSolver(sample, &cam);
    models->clear();
    models->push_back(cam);
    return 1;
  }
  int NonMinimalSolver(const std::vector<int>& sample, Pose2d* model) const {
    std::vector<int32_t> s(sample.begin(), sample.end());
    Check(pp_pose2d_solve_batch(h_, 1, (int32_t)s.size(), s.data(), model->data()));
    return 1;
  }
  double EvaluateModelOnPoint(const Pose2d& model, int i) const {
    return cache_.Get(model.data(), i, [&](std::vector<double>* err) {
      err->resize(n_);
      Check(pp_pose2d_evaluate(h_, model.data(), err->data()));
    });
  }
  void LeastSquares(const std::vector<int>& sample, Pose2d* model) const { NonMinimalSolver(sample, model); }

 private:
  pp_pose2d_handle h_ = nullptr;
  int n_;
  detail::ErrorCache<6> cache_;
};

// ---- FourView2dEstimator (reference src/init/sfm2d.h:48-97, sfm2d.cc:178-489) -----------------------------------------
class FourView2dSolver {
 public:
  struct Reconstruction {
    std::vector<Pose2d> cams;        // 4 x (2x3 row-major)
    std::vector<Vector2d> X;         // num_data() triangulated 2D points
  };
  typedef std::vector<Reconstruction> ReconstructionVector;

  // x[v][i]: bearing of track i in view v.  frames: the three 2x2 coordinate changes of factorize_trifocal_tensor, which the
  // reference draws with Matrix2d::setRandom() per call (sfm2d.cc:231-235); nullptr = the library's fixed set.
  FourView2dSolver(const std::vector<std::vector<Vector2d>>& x, double inlier_threshold, const double* frames = nullptr, int device = 0)
      : n_(x.empty() ? 0 : (int)x[0].size()), inlier_threshold_(inlier_threshold), have_frames_(frames != nullptr) {
    if (x.size() != 4) throw Error(PP_ERR_INVALID, "FourView2dSolver: four views expected");
    std::vector<double> a((size_t)8 * n_);
    for (int v = 0; v < 4; ++v) {
      if ((int)x[v].size() != n_) throw Error(PP_ERR_INVALID, "FourView2dSolver: CHECK_EQ on the track counts");
      for (int i = 0; i < n_; ++i) { a[((size_t)v * n_ + i) * 2] = x[v][i][0]; a[((size_t)v * n_ + i) * 2 + 1] = x[v][i][1]; }
    }
    if (frames) std::memcpy(frames_.data(), frames, sizeof(double) * 12);
    Check(pp_fourview2d_create(n_, a.data(), device, &h_));
  }
  ~FourView2dSolver() { pp_fourview2d_destroy(h_); }
  FourView2dSolver(const FourView2dSolver&) = delete;
  FourView2dSolver& operator=(const FourView2dSolver&) = delete;

  int min_sample_size() const { return 5; }
  int non_minimal_sample_size() const { return 2 * min_sample_size(); }
  int num_data() const { return n_; }

  int MinimalSolver(const std::vector<int>& sample, ReconstructionVector* models) const {
    models->clear();
    std::vector<int32_t> s(sample.begin(), sample.end());
    std::vector<double> cams((size_t)16 * 24);
    int32_t count = 0;
    Check(pp_fourview2d_minimal_batch(h_, 1, (int32_t)s.size(), s.data(), have_frames_ ? frames_.data() : nullptr, cams.data(), &count));
    for (int m = 0; m < count; ++m) models->push_back(WithPoints(cams.data() + 24 * m));
    return count;
  }
  // MinimalSolver + the candidate with the smallest MSAC score over all tracks (sfm2d.cc:446-467), in one device call
  int NonMinimalSolver(const std::vector<int>& sample, Reconstruction* model) const {
    std::vector<int32_t> s(sample.begin(), sample.end());
    double cams[24], score = 0;
    int32_t index = -1;
    Check(pp_fourview2d_nonminimal_batch(h_, 1, (int32_t)s.size(), s.data(), have_frames_ ? frames_.data() : nullptr, inlier_threshold_, cams, &score, &index));
    if (index < 0) return 0;
    *model = WithPoints(cams);
    return 1;
  }
  double EvaluateModelOnPoint(const Reconstruction& model, int i) const {
    if (model.cams.size() != 4) return 1e6;
    double key[24];
    for (int v = 0; v < 4; ++v) std::memcpy(key + 6 * v, model.cams[v].data(), sizeof(double) * 6);
    // The reference evaluates against the model's OWN points model.X (sfm2d.cc:302-319): after MinimalSolver / NonMinimalSolver
    // those are the three-view triangulations, after LeastSquares they are refined points.  On a cache miss (the driver comes
    // back to an old LO-refined best model in GetInliers, RansacLib ransac.h:228,247, long after its entry was evicted) the
    // errors are therefore recomputed from model.X, never from a fresh triangulation.
    if ((int)model.X.size() != n_) return 1e6;
    return cache_.Get(key, i, [&](std::vector<double>* err) {
      err->resize(n_);
      std::vector<double> X((size_t)2 * n_);
      for (int q = 0; q < n_; ++q) { X[2 * q] = model.X[q][0]; X[2 * q + 1] = model.X[q][1]; }
      Check(pp_fourview2d_evaluate_points(h_, key, X.data(), err->data()));
    });
  }
  // bundle_adjust2d on the sample (>= 10 tracks) + optimize_points2d on all tracks (sfm2d.cc:469-489)
  void LeastSquares(const std::vector<int>& sample, Reconstruction* model) const {
    if (model->cams.size() != 4 || (int)model->X.size() != n_) return;
    std::vector<int32_t> s(sample.begin(), sample.end());
    double cams[24];
    std::vector<double> X((size_t)2 * n_);
    for (int v = 0; v < 4; ++v) std::memcpy(cams + 6 * v, model->cams[v].data(), sizeof(double) * 6);
    for (int i = 0; i < n_; ++i) { X[2 * i] = model->X[i][0]; X[2 * i + 1] = model->X[i][1]; }
    Check(pp_fourview2d_least_squares(h_, (int32_t)s.size(), s.data(), cams, X.data()));
    for (int v = 0; v < 4; ++v) std::memcpy(model->cams[v].data(), cams + 6 * v, sizeof(double) * 6);
    for (int i = 0; i < n_; ++i) model->X[i] = Vector2d{{X[2 * i], X[2 * i + 1]}};
    std::vector<double> err(n_);
    Check(pp_fourview2d_evaluate_points(h_, cams, X.data(), err.data()));      // errors against the model's own (refined) points
    cache_.Put(cams, std::move(err));
  }

 private:
  Reconstruction WithPoints(const double* cams) const {
    Reconstruction rec;
    rec.cams.resize(4); rec.X.resize(n_);
    for (int v = 0; v < 4; ++v) std::memcpy(rec.cams[v].data(), cams + 6 * v, sizeof(double) * 6);
    std::vector<double> err(n_), X((size_t)2 * n_);
    Check(pp_fourview2d_evaluate(h_, cams, err.data(), X.data()));
    for (int i = 0; i < n_; ++i) rec.X[i] = Vector2d{{X[2 * i], X[2 * i + 1]}};
    cache_.Put(cams, std::move(err));
    return rec;
  }
  pp_fourview2d_handle h_ = nullptr;
  int n_;
  double inlier_threshold_;
  bool have_frames_;
  std::array<double, 12> frames_{};
  detail::ErrorCache<24> cache_;
};

}  // namespace init
}  // namespace ppsfm

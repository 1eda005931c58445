// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// CPU restatement of the four-view line-initialisation hot path driven by RansacLib's LO-MSAC:
//   LocallyOptimizedMSAC::EstimateModel / ScoreModel / GetInliers / LocalOptimization / LeastSquaresFit
//                                         reference lib/RansacLib/RansacLib/ransac.h:127-428
//   UniformSampling                       lib/RansacLib/RansacLib/sampling.h:46-135
//   RandomShuffleAndResize, NumRequiredIterations      lib/RansacLib/RansacLib/utils.h:48-132
//   PlanarOffsetEstimator::{MinimalSolver, NonMinimalSolver, EvaluateModelOnPoint, LeastSquares (no-op)},
//   four_view_triangulate                 src/init/initializer.cc:219-333, :450-451
//   FourView2dEstimator::{EvaluateModelOnPoint, AbsPoseSolver}, three_view_triangulate2d
//                                         src/init/sfm2d.cc:194-213, :302-361
//   AbsolutePose2dEstimator::{NonMinimalSolver, EvaluateModelOnPoint}     src/init/sfm2d.cc:491-530
//   FourView2dEstimator::MinimalSolver, factorize_trifocal_tensor, metric_upgrade, trifocal_tensor_coord_change
//                                         src/init/sfm2d.cc:178-298, :363-444
//   FourView2dEstimator::{NonMinimalSolver, LeastSquares}, bundle_adjust2d, optimize_points2d, BundleAdjustment2DCostFunction
//                                         src/init/sfm2d.cc:42-175, :446-489   (Ceres absent: its published trust-region
//                                         Levenberg-Marquardt and HomogeneousVectorParameterization are restated; PARITY UNPINNED)
//
// Eigen (absent) pieces restated by their published definitions: colPivHouseholderQr().solve == the least
// squares solution for full column rank (computed by Householder QR with column pivoting);
// JacobiSVD(...).matrixV().col(last) == eigenvector of A^T A for its smallest eigenvalue (cyclic Jacobi).
// The LO-MSAC driver is pinned against the reference's own headers compiled in place (oracle/_ref).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <numeric>
#include <random>
#include <vector>
#include "trust_region.h"
#include "linalg.h"

namespace oracle {

// ---- small dense helpers -------------------------------------------------------------------------
// least squares min |A x - b|, A m x n row-major (n <= 4), Householder QR with column pivoting
inline void LeastSquaresQR(int m, int n, const double* Ain, const double* bin, double* x) {
  std::vector<double> A(Ain, Ain + (size_t)m * n), b(bin, bin + m);
  int perm[8];
  for (int j = 0; j < n; ++j) perm[j] = j;
  const int steps = std::min(m, n);
  for (int k = 0; k < steps; ++k) {
    int best = k; double bn = -1;
    for (int j = k; j < n; ++j) { double s = 0; for (int i = k; i < m; ++i) s += A[i * n + j] * A[i * n + j]; if (s > bn) { bn = s; best = j; } }
    if (best != k) { for (int i = 0; i < m; ++i) std::swap(A[i * n + k], A[i * n + best]); std::swap(perm[k], perm[best]); }
    double norm = 0; for (int i = k; i < m; ++i) norm += A[i * n + k] * A[i * n + k];
    norm = std::sqrt(norm);
    if (norm == 0) continue;
    const double alpha = A[k * n + k] > 0 ? -norm : norm;
    std::vector<double> v(m, 0.0);
    for (int i = k; i < m; ++i) v[i] = A[i * n + k];
    v[k] -= alpha;
    double vn = 0; for (int i = k; i < m; ++i) vn += v[i] * v[i];
    if (vn == 0) continue;
    for (int j = k; j < n; ++j) { double s = 0; for (int i = k; i < m; ++i) s += v[i] * A[i * n + j]; s = 2 * s / vn; for (int i = k; i < m; ++i) A[i * n + j] -= s * v[i]; }
    { double s = 0; for (int i = k; i < m; ++i) s += v[i] * b[i]; s = 2 * s / vn; for (int i = k; i < m; ++i) b[i] -= s * v[i]; }
  }
  double y[8];
  for (int k = steps - 1; k >= 0; --k) { double s = b[k]; for (int j = k + 1; j < steps; ++j) s -= A[k * n + j] * y[j]; y[k] = s / A[k * n + k]; }
  for (int j = 0; j < n; ++j) x[j] = 0;
  for (int k = 0; k < steps; ++k) x[perm[k]] = y[k];
}

// eigen-decomposition of a symmetric n x n matrix (n <= 6) by cyclic Jacobi; eigenvalues ascending in w,
// eigenvectors in the columns of V (row-major n x n)
inline void SymmetricEigen(int n, const double* Ain, double* w, double* V) {
  double A[36];
  for (int i = 0; i < n * n; ++i) A[i] = Ain[i];
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) V[i * n + j] = i == j;
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0; for (int i = 0; i < n; ++i) for (int j = i + 1; j < n; ++j) off += A[i * n + j] * A[i * n + j];
    if (off < 1e-300) break;
    for (int p = 0; p < n; ++p) for (int q = p + 1; q < n; ++q) {
      if (A[p * n + q] == 0) continue;
      const double theta = (A[q * n + q] - A[p * n + p]) / (2 * A[p * n + q]);
      const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1));
      const double c = 1 / std::sqrt(t * t + 1), s = t * c;
      for (int k = 0; k < n; ++k) { const double akp = A[k * n + p], akq = A[k * n + q]; A[k * n + p] = c * akp - s * akq; A[k * n + q] = s * akp + c * akq; }
      for (int k = 0; k < n; ++k) { const double apk = A[p * n + k], aqk = A[q * n + k]; A[p * n + k] = c * apk - s * aqk; A[q * n + k] = s * apk + c * aqk; }
      for (int k = 0; k < n; ++k) { const double vkp = V[k * n + p], vkq = V[k * n + q]; V[k * n + p] = c * vkp - s * vkq; V[k * n + q] = s * vkp + c * vkq; }
    }
  }
  int idx[6]; for (int i = 0; i < n; ++i) idx[i] = i;
  std::sort(idx, idx + n, [&](int a, int b) { return A[a * n + a] < A[b * n + b]; });
  double V2[36];
  for (int j = 0; j < n; ++j) { w[j] = A[idx[j] * n + idx[j]]; for (int i = 0; i < n; ++i) V2[i * n + j] = V[i * n + idx[j]]; }
  for (int i = 0; i < n * n; ++i) V[i] = V2[i];
}

// ---- RansacLib restatement ---------------------------------------------------------------------------
struct LORansacOptions {   // ransac.h:46-92 defaults
  uint32_t min_num_iterations = 100, max_num_iterations = 10000;
  double success_probability = 0.9999, squared_inlier_threshold = 1.0;
  unsigned random_seed = 0;
  int num_lo_steps = 10; double threshold_multiplier = std::sqrt(2.0); int num_lsq_iterations = 4;
  int min_sample_multiplicator = 7, non_min_sample_multiplier = 3; uint32_t lo_starting_iterations = 50;
  bool final_least_squares = false;
};
struct RansacStatistics {
  uint32_t num_iterations = 0; int best_num_inliers = 0; double best_model_score = std::numeric_limits<double>::max();
  double inlier_ratio = 0; std::vector<int> inlier_indices; int number_lo_iterations = 0;
};

inline uint32_t NumRequiredIterations(double inlier_ratio, double prob_missing, int sample_size, uint32_t min_it, uint32_t max_it) {
  if (inlier_ratio <= 0.0) return max_it;
  if (inlier_ratio >= 1.0) return min_it;
  const double kProbNonInlierSample = 1.0 - std::pow(inlier_ratio, static_cast<double>(sample_size));
  const double num_iters = std::ceil(std::log(prob_missing) / std::log(kProbNonInlierSample) + 0.5);
  uint32_t r = std::min(static_cast<uint32_t>(num_iters), max_it);
  return std::max(min_it, r);
}

// the host toolchain's <random> is used exactly as the reference uses it (sampling.h / utils.h)
class UniformSampling {
 public:
  UniformSampling(unsigned seed, int num_data, int sample_size) : num_data_(num_data), sample_size_(sample_size) {
    rng_.seed(seed);
    const double kCoeff = static_cast<double>(num_data) / static_cast<double>(num_data - sample_size);
    draw_sample_ = kCoeff < M_E;
    dist_.param(std::uniform_int_distribution<int>::param_type(0, num_data_ - 1));
  }
  void Sample(std::vector<int>* s) {
    if (draw_sample_) {
      s->resize(sample_size_);
      for (int i = 0; i < sample_size_; ++i) {
        bool found = true;
        while (found) { found = false; (*s)[i] = dist_(rng_); for (int j = 0; j < i; ++j) if ((*s)[j] == (*s)[i]) { found = true; break; } }
      }
    } else {
      s->resize(num_data_);
      std::iota(s->begin(), s->end(), 0);
      if (sample_size_ == num_data_) return;
      Shuffle(&rng_, s);
      s->resize(sample_size_);
    }
  }
  static void Shuffle(std::mt19937* rng, std::vector<int>* v) {
    const int n = static_cast<int>(v->size());
    for (int i = 0; i < n - 1; ++i) { std::uniform_int_distribution<int> d(i, n - 1); std::swap((*v)[i], (*v)[d(*rng)]); }
  }
 private:
  std::mt19937 rng_; std::uniform_int_distribution<int> dist_; int num_data_, sample_size_; bool draw_sample_;
};

template <class Model, class Solver>
class LocallyOptimizedMSAC {
 public:
  int EstimateModel(const LORansacOptions& o, const Solver& solver, Model* best_model, RansacStatistics* st) const {
    *st = RansacStatistics();
    const int kMin = solver.min_sample_size(), kN = solver.num_data();
    if (kMin > kN || kMin <= 0) return 0;
    UniformSampling sampler(o.random_seed, kN, kMin);
    uint32_t max_it = std::max(o.max_num_iterations, o.min_num_iterations);
    const double thr = o.squared_inlier_threshold;
    Model best_min; double best_min_score = std::numeric_limits<double>::max();
    std::vector<int> sample(kMin); std::vector<Model> models;
    auto refresh = [&]() {
      st->best_num_inliers = GetInliers(solver, *best_model, thr, &st->inlier_indices);
      st->inlier_ratio = static_cast<double>(st->best_num_inliers) / static_cast<double>(kN);
      max_it = NumRequiredIterations(st->inlier_ratio, 1.0 - o.success_probability, kMin, o.min_num_iterations, o.max_num_iterations);
    };
    for (st->num_iterations = 0; st->num_iterations < max_it; ++st->num_iterations) {
      if (st->num_iterations == o.lo_starting_iterations && best_min_score < std::numeric_limits<double>::max()) {
        ++st->number_lo_iterations;
        LocalOptimization(o, solver, best_model, &st->best_model_score);
        refresh();
      }
      sampler.Sample(&sample);
      const int nm = solver.MinimalSolver(sample, &models);
      if (nm <= 0) continue;
      double best_local = std::numeric_limits<double>::max(); int best_id = 0;
      for (int m = 0; m < nm; ++m) { const double sc = ScoreModel(solver, models[m], thr); if (sc < best_local) { best_local = sc; best_id = m; } }
      if (best_local < best_min_score || st->num_iterations == o.lo_starting_iterations) {
        const bool kBestMin = best_local < best_min_score;
        if (kBestMin) { best_min_score = best_local; best_min = models[best_id]; Update(best_min_score, best_min, &st->best_model_score, best_model); }
        const bool kRunLO = st->num_iterations >= o.lo_starting_iterations && best_min_score < std::numeric_limits<double>::max();
        if (!kBestMin && !kRunLO) continue;
        if (kRunLO) {
          ++st->number_lo_iterations;
          double score = best_min_score;
          LocalOptimization(o, solver, &best_min, &score);
          Update(score, best_min, &st->best_model_score, best_model);
        }
        refresh();
      }
    }
    if (st->num_iterations <= o.lo_starting_iterations && st->best_model_score < std::numeric_limits<double>::max()) {
      ++st->number_lo_iterations;
      LocalOptimization(o, solver, best_model, &st->best_model_score);
      st->best_num_inliers = GetInliers(solver, *best_model, thr, &st->inlier_indices);
      st->inlier_ratio = static_cast<double>(st->best_num_inliers) / static_cast<double>(kN);
    }
    if (o.final_least_squares) {
      Model refined = *best_model;
      solver.LeastSquares(st->inlier_indices, &refined);
      const double score = ScoreModel(solver, refined, thr);
      if (score < st->best_model_score) {
        st->best_model_score = score; *best_model = refined;
        st->best_num_inliers = GetInliers(solver, *best_model, thr, &st->inlier_indices);
        st->inlier_ratio = static_cast<double>(st->best_num_inliers) / static_cast<double>(kN);
      }
    }
    return st->best_num_inliers;
  }

  static double ScoreModel(const Solver& s, const Model& m, double thr) {
    double score = 0; const int n = s.num_data();
    for (int i = 0; i < n; ++i) score += std::min(s.EvaluateModelOnPoint(m, i), thr);
    return score;
  }
  static int GetInliers(const Solver& s, const Model& m, double thr, std::vector<int>* inl) {
    const int n = s.num_data(); int c = 0;
    if (inl) inl->clear();
    for (int i = 0; i < n; ++i) if (s.EvaluateModelOnPoint(m, i) < thr) { ++c; if (inl) inl->push_back(i); }   // strict <
    return c;
  }

 private:
  static void Update(double sc, const Model& m, double* best_sc, Model* best) { if (sc < *best_sc) { *best_sc = sc; *best = m; } }
  static void ShuffleAndResize(int target, std::mt19937* rng, std::vector<int>* v) { UniformSampling::Shuffle(rng, v); v->resize(target); }
  void LeastSquaresFit(const LORansacOptions& o, double thresh, const Solver& s, std::mt19937* rng, Model* m) const {
    const int kSize = o.min_sample_multiplicator * s.min_sample_size();
    std::vector<int> inl;
    const int ni = GetInliers(s, *m, thresh, &inl);
    if (ni < s.min_sample_size()) return;
    ShuffleAndResize(std::min(kSize, ni), rng, &inl);
    s.LeastSquares(inl, m);
  }
  void LocalOptimization(const LORansacOptions& o, const Solver& s, Model* best_min, double* score_best) const {
    const int kN = s.num_data(), kMinNonMin = s.non_minimal_sample_size();
    if (kMinNonMin > kN) return;
    const int kMin = s.min_sample_size();
    const double thr = o.squared_inlier_threshold, mult = o.threshold_multiplier;
    std::mt19937 rng; rng.seed(o.random_seed);
    Model m_init = *best_min;
    LeastSquaresFit(o, thr * mult, s, &rng, &m_init);
    double score = ScoreModel(s, m_init, thr);
    Update(score, m_init, score_best, best_min);
    std::vector<int> base;
    GetInliers(s, m_init, thr, &base);
    const int kNonMin = std::max(kMinNonMin, std::min(kMin * o.non_min_sample_multiplier, static_cast<int>(base.size()) / 2));
    std::vector<int> sample;
    for (int r = 0; r < o.num_lo_steps; ++r) {
      sample = base;
      ShuffleAndResize(kNonMin, &rng, &sample);
      Model m_non_min;
      if (!s.NonMinimalSolver(sample, &m_non_min)) continue;
      score = ScoreModel(s, m_non_min, thr);
      Update(score, m_non_min, score_best, best_min);
      LeastSquaresFit(o, thr, s, &rng, &m_non_min);
      double thresh = mult * thr;
      const double upd = (mult - 1.0) * thr / static_cast<int>(o.num_lsq_iterations - 1);
      for (int i = 0; i < o.num_lsq_iterations; ++i) {
        LeastSquaresFit(o, thresh, s, &rng, &m_non_min);
        score = ScoreModel(s, m_non_min, thr);
        Update(score, m_non_min, score_best, best_min);
        thresh -= upd;
      }
    }
  }
};

// ---- planar offset estimator (initializer.cc:219-333) --------------------------------------------------
struct Pose34 { double m[12]; };   // 3x4 row-major

struct PlanarOffsetModel {
  Pose34 cams[4];
  std::vector<double> X;   // N x 3
};

class PlanarOffsetEstimator {
 public:
  // poses: 4 lifted cameras (3x4), lines[j]: N x 3 (unaligned lines of view j), Rg[j]: 3x3 row-major
  PlanarOffsetEstimator(const double* poses, const double* const lines[4], int n, const double* Rg, double thr) : n_(n), thr_(thr) {
    for (int j = 0; j < 4; ++j) {
      for (int e = 0; e < 12; ++e) poses_[j].m[e] = poses[12 * j + e];
      for (int e = 0; e < 9; ++e) Rg_[j][e] = Rg[9 * j + e];
      lines_[j].assign(lines[j], lines[j] + 3 * (size_t)n);
    }
  }
  int min_sample_size() const { return 3; }
  int non_minimal_sample_size() const { return 20; }
  int num_data() const { return n_; }

  void FourViewTriangulate(const Pose34 cams[4], std::vector<double>* X) const {
    X->resize(3 * (size_t)n_);
    for (int i = 0; i < n_; ++i) {
      double A[12], b[4];
      for (int j = 0; j < 4; ++j) {
        const double* l = &lines_[j][3 * i];
        for (int c = 0; c < 3; ++c) A[3 * j + c] = l[0] * cams[j].m[c] + l[1] * cams[j].m[4 + c] + l[2] * cams[j].m[8 + c];
        b[j] = -(l[0] * cams[j].m[3] + l[1] * cams[j].m[7] + l[2] * cams[j].m[11]);
      }
      LeastSquaresQR(4, 3, A, b, &(*X)[3 * i]);
    }
  }
  // the out-of-plane translations (t_y of cameras 1..3) from a sample of line quadruples
  bool SolveOffsets(const std::vector<int>& sample, double tt[3]) const {
    const int m = (int)sample.size();
    std::vector<double> A(3 * (size_t)m), b(m);
    for (int i = 0; i < m; ++i) {
      double A0[9], B0[12] = {0};
      for (int j = 1; j < 4; ++j) {
        const double* l = &lines_[j][3 * sample[i]];
        double lg[3]; for (int r = 0; r < 3; ++r) lg[r] = Rg_[j][3 * r] * l[0] + Rg_[j][3 * r + 1] * l[1] + Rg_[j][3 * r + 2] * l[2];
        for (int c = 0; c < 3; ++c) A0[3 * (j - 1) + c] = lg[0] * poses_[j].m[c] + lg[1] * poses_[j].m[4 + c] + lg[2] * poses_[j].m[8 + c];
        B0[4 * (j - 1) + (j - 1)] = lg[1];
        B0[4 * (j - 1) + 3] = lg[0] * poses_[j].m[3] + lg[2] * poses_[j].m[11];
      }
      if (!LuSolve(3, 4, A0, B0)) return false;          // B0 = A0^-1 B0  (partialPivLu)
      double RB[12];                                      // Rg0^T * B0
      for (int r = 0; r < 3; ++r) for (int c = 0; c < 4; ++c) RB[4 * r + c] = Rg_[0][r] * B0[c] + Rg_[0][3 + r] * B0[4 + c] + Rg_[0][6 + r] * B0[8 + c];
      const double* l0 = &lines_[0][3 * sample[i]];
      for (int c = 0; c < 3; ++c) A[3 * i + c] = l0[0] * RB[c] + l0[1] * RB[4 + c] + l0[2] * RB[8 + c];
      b[i] = -(l0[0] * RB[3] + l0[1] * RB[7] + l0[2] * RB[11]);
    }
    LeastSquaresQR(m, 3, A.data(), b.data(), tt);
    return true;
  }
  void CamsFromOffsets(const double tt[3], Pose34 cams[4]) const {
    for (int j = 0; j < 4; ++j) {
      Pose34 p = poses_[j];
      if (j > 0) p.m[7] = tt[j - 1];
      for (int r = 0; r < 3; ++r) for (int c = 0; c < 4; ++c)
        cams[j].m[4 * r + c] = Rg_[j][r] * p.m[c] + Rg_[j][3 + r] * p.m[4 + c] + Rg_[j][6 + r] * p.m[8 + c];   // Rg^T * pose
    }
  }
  int MinimalSolver(const std::vector<int>& sample, std::vector<PlanarOffsetModel>* models) const {
    double tt[3];
    models->clear();
    if (!SolveOffsets(sample, tt)) return 0;
    PlanarOffsetModel rec;
    CamsFromOffsets(tt, rec.cams);
    FourViewTriangulate(rec.cams, &rec.X);
    models->push_back(rec);
    return 1;
  }
  int NonMinimalSolver(const std::vector<int>& sample, PlanarOffsetModel* model) const {
    std::vector<PlanarOffsetModel> models;
    MinimalSolver(sample, &models);
    if (models.empty()) return 0;
    *model = models[0];      // a single model: the best-score selection of :289-303 is trivial; LeastSquares is a no-op
    return 1;
  }
  double EvaluateModelOnPoint(const PlanarOffsetModel& model, int i) const {
    double err = 0;
    const double* X = &model.X[3 * i];
    double z[4][3];
    for (int j = 0; j < 4; ++j)
      for (int r = 0; r < 3; ++r) z[j][r] = model.cams[j].m[4 * r] * X[0] + model.cams[j].m[4 * r + 1] * X[1] + model.cams[j].m[4 * r + 2] * X[2] + model.cams[j].m[4 * r + 3];
    if (z[0][2] < 0 || z[1][2] < 0 || z[2][2] < 0 || z[3][2] < 0) return 100000.0;
    double e[4];
    for (int j = 0; j < 4; ++j) {
      const double* l = &lines_[j][3 * i];
      const double zx = z[j][0] / z[j][2], zy = z[j][1] / z[j][2];
      e[j] = std::fabs((l[0] * zx + l[1] * zy + l[2] * 1.0) / std::sqrt(l[0] * l[0] + l[1] * l[1]));
    }
    // nested as initializer.cc:332 nests it: for finite errors any order gives the maximum; with a NaN among them (a model from a degenerate sample) std::max's
    // "a < b ? b : a" makes the order the result - a NaN model's error is NaN, which no threshold test counts as an inlier (a running maximum from 0 would
    // drop the NaNs and score such a model PERFECT)
    err = std::max(e[0], std::max(e[1], std::max(e[2], e[3])));
    return err;
  }
  void LeastSquares(const std::vector<int>&, PlanarOffsetModel*) const {}   // returns on its first line (initializer.cc:450-451)

 private:
  int n_; double thr_;
  Pose34 poses_[4]; double Rg_[4][9];
  std::vector<double> lines_[4];
};

// ---- 2D pieces (sfm2d.cc) ------------------------------------------------------------------------------
struct Pose2d { double m[6]; };   // 2x3 row-major

// AbsPoseSolver (sfm2d.cc:321-361): 2D similarity-free pose [a -b tx; b a ty] from bearings x and 2D points X
inline int AbsPoseSolver2d(const std::vector<int>& sample, const double* x /*n x 2*/, const double* X /*n x 2*/, Pose2d* model) {
  const int m = (int)sample.size();
  std::vector<double> A(2 * (size_t)m), B(2 * (size_t)m);
  for (int i = 0; i < m; ++i) {
    const double x1 = x[2 * sample[i]], x2 = x[2 * sample[i] + 1], X1 = X[2 * sample[i]], X2 = X[2 * sample[i] + 1];
    A[2 * i] = X1 * x2 - X2 * x1; A[2 * i + 1] = -X1 * x1 - X2 * x2;
    B[2 * i] = x2; B[2 * i + 1] = -x1;
  }
  double BtB[4] = {0, 0, 0, 0}, BtA[4] = {0, 0, 0, 0};
  for (int i = 0; i < m; ++i) for (int r = 0; r < 2; ++r) for (int c = 0; c < 2; ++c) { BtB[2 * r + c] += B[2 * i + r] * B[2 * i + c]; BtA[2 * r + c] += B[2 * i + r] * A[2 * i + c]; }
  const double det = BtB[0] * BtB[3] - BtB[1] * BtB[2];
  const double inv[4] = {BtB[3] / det, -BtB[1] / det, -BtB[2] / det, BtB[0] / det};
  double C[4];
  for (int r = 0; r < 2; ++r) for (int c = 0; c < 2; ++c) C[2 * r + c] = -(inv[2 * r] * BtA[c] + inv[2 * r + 1] * BtA[2 + c]);
  double MtM[4] = {0, 0, 0, 0};
  for (int i = 0; i < m; ++i) {
    const double r0 = A[2 * i] + B[2 * i] * C[0] + B[2 * i + 1] * C[2], r1 = A[2 * i + 1] + B[2 * i] * C[1] + B[2 * i + 1] * C[3];
    MtM[0] += r0 * r0; MtM[1] += r0 * r1; MtM[2] += r0 * r1; MtM[3] += r1 * r1;
  }
  double w[2], V[4];
  SymmetricEigen(2, MtM, w, V);
  double ab[2] = {V[0], V[2]};      // right singular vector of the smallest singular value (matrixV().col(1))
  const double n = std::sqrt(ab[0] * ab[0] + ab[1] * ab[1]); ab[0] /= n; ab[1] /= n;
  const double t0 = C[0] * ab[0] + C[1] * ab[1], t1 = C[2] * ab[0] + C[3] * ab[1];
  double* M = model->m;
  M[0] = ab[0]; M[1] = -ab[1]; M[2] = t0; M[3] = ab[1]; M[4] = ab[0]; M[5] = t1;
  const double X1 = X[2 * sample[0]], X2 = X[2 * sample[0] + 1];
  if (M[3] * X1 + M[4] * X2 + M[5] < 0) for (int e = 0; e < 6; ++e) M[e] = -M[e];
  return 1;
}

class AbsolutePose2dEstimator {   // sfm2d.h:99-130, sfm2d.cc:491-530
 public:
  AbsolutePose2dEstimator(const double* x, const double* X, int n) : n_(n), x_(x, x + 2 * (size_t)n), X_(X, X + 2 * (size_t)n) {
    for (int i = 0; i < n; ++i) { const double nr = std::sqrt(x_[2 * i] * x_[2 * i] + x_[2 * i + 1] * x_[2 * i + 1]); x_[2 * i] /= nr; x_[2 * i + 1] /= nr; }
  }
  int min_sample_size() const { return 3; }
  int non_minimal_sample_size() const { return 6; }
  int num_data() const { return n_; }
  int MinimalSolver(const std::vector<int>& sample, std::vector<Pose2d>* models) const {
    models->resize(1);
    return NonMinimalSolver(sample, &(*models)[0]);
  }
  int NonMinimalSolver(const std::vector<int>& sample, Pose2d* model) const {
    double AtA[16] = {0};
    for (int s : sample) {
      const double x1 = x_[2 * s], x2 = x_[2 * s + 1], X1 = X_[2 * s], X2 = X_[2 * s + 1];
      const double row[4] = {X1 * x2 - X2 * x1, -X1 * x1 - X2 * x2, x2, -x1};
      for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) AtA[4 * r + c] += row[r] * row[c];
    }
    double w[4], V[16];
    SymmetricEigen(4, AtA, w, V);
    double t[4] = {V[0], V[4], V[8], V[12]};
    const double n = std::sqrt(t[0] * t[0] + t[1] * t[1]);
    for (double& v : t) v /= n;
    double* M = model->m;
    M[0] = t[0]; M[1] = -t[1]; M[2] = t[2]; M[3] = t[1]; M[4] = t[0]; M[5] = t[3];
    if (M[3] * X_[2 * sample[0]] + M[4] * X_[2 * sample[0] + 1] + M[5] < 0) for (int e = 0; e < 6; ++e) M[e] = -M[e];
    return 1;
  }
  double EvaluateModelOnPoint(const Pose2d& m, int i) const {
    const double z0 = m.m[0] * X_[2 * i] + m.m[1] * X_[2 * i + 1] + m.m[2], z1 = m.m[3] * X_[2 * i] + m.m[4] * X_[2 * i + 1] + m.m[5];
    const double n = std::sqrt(z0 * z0 + z1 * z1);
    return 1.0 - (x_[2 * i] * (z0 / n) + x_[2 * i + 1] * (z1 / n));
  }
  void LeastSquares(const std::vector<int>& sample, Pose2d* model) const { NonMinimalSolver(sample, model); }   // sfm2d.h
 private:
  int n_; std::vector<double> x_, X_;
};

// three_view_triangulate2d (sfm2d.cc:194-213) for one point: bearings x1..x3 (2-vectors), cameras 2x3
inline void ThreeViewTriangulate2d(const Pose2d P[3], const double* const x[3], int i, double X[2]) {
  double A[6], b[3];
  for (int j = 0; j < 3; ++j) {
    const double xa = x[j][2 * i], xb = x[j][2 * i + 1];
    A[2 * j] = xa * P[j].m[3] - xb * P[j].m[0];
    A[2 * j + 1] = xa * P[j].m[4] - xb * P[j].m[1];
    b[j] = xb * P[j].m[2] - xa * P[j].m[5];
  }
  LeastSquaresQR(3, 2, A, b, X);
}

// FourView2dEstimator::EvaluateModelOnPoint (sfm2d.cc:302-319): bearings are unit 2-vectors; hnormalized of
// a 2-vector is x/y; error = max over the four views of |x_j/y_j - z_j0/z_j1|, 1e6 if any z_j1 < 0
inline double FourView2dError(const Pose2d cams[4], const double* const x[4], int i, const double X[2]) {
  double err = 0;
  double z[4][2];
  for (int j = 0; j < 4; ++j) { z[j][0] = cams[j].m[0] * X[0] + cams[j].m[1] * X[1] + cams[j].m[2]; z[j][1] = cams[j].m[3] * X[0] + cams[j].m[4] * X[1] + cams[j].m[5]; }
  if (z[0][1] < 0 || z[1][1] < 0 || z[2][1] < 0 || z[3][1] < 0) return 1000000.0;
  double e[4];
  for (int j = 0; j < 4; ++j) e[j] = std::fabs(x[j][2 * i] / x[j][2 * i + 1] - z[j][0] / z[j][1]);
  err = std::max(e[0], std::max(e[1], std::max(e[2], e[3])));      // (sfm2d.cc:316's nesting: see PlanarOffset above for why it matters with NaNs)
  return err;
}


// ---- four-view 2D minimal solver (sfm2d.cc:178-298, 363-444) ----------------------------------------------
// The 2D trifocal tensor T_abc (index a + 2b + 4c; a,b,c = component in views 1,2,3) satisfies
// sum_abc T_abc x1_a x2_b x3_c = 0 for corresponding bearings.  For calibrated cameras two entries are linear in
// the other six (sfm2d.cc:377-379): T_0 = T_3 + T_5 + T_6,  T_1 = T_7 - T_2 - T_4.

// right singular vector of the smallest singular value of an m x n matrix (n <= 6) = eigenvector of A^T A
inline void NullVector(int m, int n, const double* A, double* v) {
  double AtA[36] = {0};
  for (int i = 0; i < m; ++i) for (int r = 0; r < n; ++r) for (int c = 0; c < n; ++c) AtA[r * n + c] += A[i * n + r] * A[i * n + c];
  double w[6], V[36];
  SymmetricEigen(n, AtA, w, V);
  for (int r = 0; r < n; ++r) v[r] = V[r * n + 0];
}

// T'_{a'b'c'} = sum_abc A1[a][a'] A2[b][b'] A3[c][c'] T_abc   (trifocal_tensor_coord_change, sfm2d.cc:215-224)
inline void TrifocalCoordChange(const double T[8], const double A1[4], const double A2[4], const double A3[4], double out[8]) {
  for (int cp = 0; cp < 2; ++cp) for (int bp = 0; bp < 2; ++bp) for (int ap = 0; ap < 2; ++ap) {
    double s = 0;
    for (int c = 0; c < 2; ++c) for (int b = 0; b < 2; ++b) for (int a = 0; a < 2; ++a)
      s += A1[2 * a + ap] * A2[2 * b + bp] * A3[2 * c + cp] * T[a + 2 * b + 4 * c];
    out[ap + 2 * bp + 4 * cp] = s;
  }
}

// A1, A2, A3: row-major 2x2 projective changes of the image coordinates (the reference draws them with
// setRandom() on every call, sfm2d.cc:231-235; here they are inputs)
inline int FactorizeTrifocalTensor(const double T[8], const double A1[4], const double A2[4], const double A3[4], Pose2d P2[2], Pose2d P3[2]) {
  double AT[8];
  TrifocalCoordChange(T, A1, A2, A3, AT);
  const double alpha = AT[2] * AT[7] - AT[3] * AT[6];
  const double beta = AT[1] * AT[6] + AT[3] * AT[4] - AT[0] * AT[7] - AT[2] * AT[5];
  const double gamma = AT[0] * AT[5] - AT[1] * AT[4];
  const double disc = beta * beta - 4.0 * alpha * gamma;
  if (disc < 0) return 0;
  const double sq = std::sqrt(disc);
  double aa[2];
  aa[0] = (beta > 0) ? (2 * gamma) / (-beta - sq) : (2.0 * gamma) / (-beta + sq);
  aa[1] = gamma / (alpha * aa[0]);
  const double iA1det = 1.0 / (A1[0] * A1[3] - A1[1] * A1[2]);
  const double A1inv[4] = {A1[3] * iA1det, -A1[1] * iA1det, -A1[2] * iA1det, A1[0] * iA1det};
  for (int i = 0; i < 2; ++i) {
    double a1 = aa[i];
    const double sn = std::sqrt(1 + a1 * a1);
    a1 /= sn;
    const double a2 = 1 / sn;
    const double rho = -(AT[1] * a2 - AT[3] * a1) / (AT[2] * a1 - AT[0] * a2);
    const double b1 = rho * a1, b2 = rho * a2, c1 = -a2, c2 = a1;
    // linear system G d = 0 for the six entries of the third camera (sfm2d.cc:263-270), row by row
    double G[42] = {0};
    auto g = [&](int r, int c) -> double& { return G[r * 6 + c]; };
    g(0, 1) = AT[7] * c2; g(0, 2) = -AT[0] * c1; g(0, 4) = AT[0] * b1; g(0, 5) = -AT[7] * a2;
    g(1, 2) = -AT[1] * c1; g(1, 3) = AT[7] * c2; g(1, 4) = AT[1] * b1; g(1, 5) = -AT[7] * b2;
    g(2, 1) = -AT[7] * c1; g(2, 2) = -AT[2] * c1; g(2, 4) = AT[2] * b1; g(2, 5) = AT[7] * a1;
    g(3, 2) = -AT[3] * c1; g(3, 3) = -AT[7] * c1; g(3, 4) = AT[3] * b1; g(3, 5) = AT[7] * b1;
    g(4, 0) = -AT[7] * c2; g(4, 2) = -AT[4] * c1; g(4, 4) = AT[7] * a2 + AT[4] * b1;
    g(5, 2) = -AT[5] * c1 - AT[7] * c2; g(5, 4) = AT[7] * b2 + AT[5] * b1;
    g(6, 0) = AT[7] * c1; g(6, 2) = -AT[6] * c1; g(6, 4) = -AT[7] * a1 + AT[6] * b1;
    double d[6];
    NullVector(7, 6, G, d);
    const double Q2[6] = {a1, b1, c1, a2, b2, c2};
    const double Q3[6] = {d[0], d[2], d[4], d[1], d[3], d[5]};
    // revert the change of coordinates: P = A * Q, then the 2x2 left block *= A1^-1
    for (int v = 0; v < 2; ++v) {
      const double* A = v == 0 ? A2 : A3; const double* Q = v == 0 ? Q2 : Q3; Pose2d& P = v == 0 ? P2[i] : P3[i];
      double M[6];
      for (int r = 0; r < 2; ++r) for (int c = 0; c < 3; ++c) M[3 * r + c] = A[2 * r] * Q[c] + A[2 * r + 1] * Q[3 + c];
      for (int r = 0; r < 2; ++r) {
        P.m[3 * r] = M[3 * r] * A1inv[0] + M[3 * r + 1] * A1inv[2];
        P.m[3 * r + 1] = M[3 * r] * A1inv[1] + M[3 * r + 1] * A1inv[3];
        P.m[3 * r + 2] = M[3 * r + 2];
      }
    }
  }
  return 2;
}

// metric_upgrade (sfm2d.cc:178-191): H = [1 0 0; 0 1 0; h0 h1 1] making P2, P3 calibrated
inline void MetricUpgrade(const Pose2d& P2, const Pose2d& P3, double h[2]) {
  const double A[8] = {P2.m[2], -P2.m[5], P2.m[5], P2.m[2], P3.m[2], -P3.m[5], P3.m[5], P3.m[2]};
  const double b[4] = {P2.m[4] - P2.m[0], -P2.m[1] - P2.m[3], P3.m[4] - P3.m[0], -P3.m[1] - P3.m[3]};
  LeastSquaresQR(4, 2, A, b, h);
}

struct FourView2dModel { Pose2d cams[4]; };

// x: 4 x n x 2 unit bearings.  Returns the number of models (0, 8 or 16), cameras only; the points of a
// model are re-triangulated from cams[0..2] whenever it is scored (ThreeViewTriangulate2d).
inline int FourView2dMinimalSolver(const double* x, int n, const int* sample, int m, const double A1[4], const double A2[4], const double A3[4],
                                   FourView2dModel* models) {
  const double* xs[4] = {x, x + 2 * (size_t)n, x + 4 * (size_t)n, x + 6 * (size_t)n};
  std::vector<double> A(6 * (size_t)m);
  for (int i = 0; i < m; ++i) {
    const int s = sample[i];
    const double a[2] = {xs[0][2 * s], xs[0][2 * s + 1]}, b[2] = {xs[1][2 * s], xs[1][2 * s + 1]}, c[2] = {xs[2][2 * s], xs[2][2 * s + 1]};
    double mono[8];
    for (int cc = 0; cc < 2; ++cc) for (int bb = 0; bb < 2; ++bb) for (int aa = 0; aa < 2; ++aa) mono[aa + 2 * bb + 4 * cc] = a[aa] * b[bb] * c[cc];
    // unknowns t_k = T_{k+2}; T_0 = t1 + t3 + t4, T_1 = t5 - t0 - t2
    double* row = &A[6 * (size_t)i];
    for (int k = 0; k < 6; ++k) row[k] = mono[k + 2];
    row[1] += mono[0]; row[3] += mono[0]; row[4] += mono[0];
    row[5] += mono[1]; row[0] -= mono[1]; row[2] -= mono[1];
  }
  double t[6];
  NullVector(m, 6, A.data(), t);
  double T[8];
  T[0] = t[1] + t[3] + t[4]; T[1] = -t[2] - t[0] + t[5];
  for (int k = 0; k < 6; ++k) T[k + 2] = t[k];
  Pose2d P2[2], P3[2];
  const int nf = FactorizeTrifocalTensor(T, A1, A2, A3, P2, P3);
  int count = 0;
  for (int f = 0; f < nf; ++f) {
    double h[2];
    MetricUpgrade(P2[f], P3[f], h);
    Pose2d Q2 = P2[f], Q3 = P3[f];
    for (Pose2d* Q : {&Q2, &Q3}) for (int r = 0; r < 2; ++r) { Q->m[3 * r] += Q->m[3 * r + 2] * h[0]; Q->m[3 * r + 1] += Q->m[3 * r + 2] * h[1]; }   // Q * H
    const double n2 = std::sqrt(Q2.m[0] * Q2.m[0] + Q2.m[3] * Q2.m[3]), n3 = std::sqrt(Q3.m[0] * Q3.m[0] + Q3.m[3] * Q3.m[3]);
    for (double& v : Q2.m) v /= n2;
    for (double& v : Q3.m) v /= n3;
    const double sc = std::sqrt(Q2.m[2] * Q2.m[2] + Q2.m[5] * Q2.m[5]);
    Q2.m[2] /= sc; Q2.m[5] /= sc; Q3.m[2] /= sc; Q3.m[5] /= sc;
    for (int flip1 = 0; flip1 < 2; ++flip1) for (int flip2 = 0; flip2 < 2; ++flip2) for (int flip3 = 0; flip3 < 2; ++flip3) {
      FourView2dModel& M = models[count];
      M.cams[0] = Pose2d{{1, 0, 0, 0, 1, 0}}; M.cams[1] = Q2; M.cams[2] = Q3;
      const double nt = std::sqrt(M.cams[1].m[2] * M.cams[1].m[2] + M.cams[1].m[5] * M.cams[1].m[5]);
      M.cams[2].m[2] /= nt; M.cams[2].m[5] /= nt;
      const double nt2 = std::sqrt(M.cams[1].m[2] * M.cams[1].m[2] + M.cams[1].m[5] * M.cams[1].m[5]);
      M.cams[1].m[2] /= nt2; M.cams[1].m[5] /= nt2;
      if (flip1) { M.cams[1].m[2] *= -1; M.cams[1].m[5] *= -1; M.cams[2].m[2] *= -1; M.cams[2].m[5] *= -1; }
      if (flip2) for (double& v : M.cams[1].m) v *= -1;
      if (flip3) for (double& v : M.cams[2].m) v *= -1;
      // fourth camera from the sample's points (triangulated with the first three) and bearings (sfm2d.cc:433-435)
      std::vector<double> Xs(2 * (size_t)m), x4(2 * (size_t)m);
      std::vector<int> idx(m);
      for (int i = 0; i < m; ++i) {
        ThreeViewTriangulate2d(M.cams, xs, sample[i], &Xs[2 * i]);
        x4[2 * i] = xs[3][2 * sample[i]]; x4[2 * i + 1] = xs[3][2 * sample[i] + 1];
        idx[i] = i;
      }
      AbsPoseSolver2d(idx, x4.data(), Xs.data(), &M.cams[3]);
      ++count;
    }
  }
  return count;
}

// ---- FourView2dEstimator::LeastSquares (sfm2d.cc:42-175, 469-489) -------------------------------------------------------
// Ceres (third party, absent, version unpinned) is restated by its published algorithm, as oracle/bundle_adjustment.h does:
// trust-region Levenberg-Marquardt (Jacobi scaling fixed at the start, clamped LM diagonal / radius, radius update
// /max(1/3, 1-(2 rho-1)^3) resp. /2,/4,..), function/gradient/parameter tolerance 1e-10, 50 iterations, exact (Schur)
// linear solve; HomogeneousVectorParameterization(2) = rotation of the 2-vector by |delta|/2 through a Householder frame.

// residual of BundleAdjustment2DCostFunction and its derivatives wrt (q0,q1,t0,t1,X0,X1)
inline double Residual2d(const double q[2], const double t[2], const double X[2], const double x[2], double d[6]) {
  const double p0 = q[0] * X[0] - q[1] * X[1] + t[0], p1 = q[1] * X[0] + q[0] * X[1] + t[1];
  if (d) {
    const double a = 1.0 / p1, b = -p0 / (p1 * p1);
    d[0] = a * X[0] + b * X[1]; d[1] = -a * X[1] + b * X[0]; d[2] = a; d[3] = b;
    d[4] = a * q[0] + b * q[1]; d[5] = -a * q[1] + b * q[0];
  }
  return p0 / p1 - x[0] / x[1];
}

// HomogeneousVectorParameterization of size 2 (ceres local_parameterization.cc): Householder frame of x
struct Homogeneous2 {
  static void Householder(const double x[2], double v[2], double* beta) {
    const double sigma = x[0] * x[0];
    v[0] = x[0]; v[1] = 1.0; *beta = 0.0;
    const double pivot = x[1];
    if (sigma <= std::numeric_limits<double>::epsilon()) { if (pivot < 0.0) *beta = 2.0; return; }
    const double mu = std::sqrt(pivot * pivot + sigma);
    double vp = (pivot <= 0.0) ? pivot - mu : -sigma / (pivot + mu);
    *beta = 2.0 * vp * vp / (sigma + vp * vp);
    v[0] /= vp;
  }
  static void Plus(const double x[2], double delta, double out[2]) {
    const double nd = std::fabs(delta);
    if (nd == 0.0) { out[0] = x[0]; out[1] = x[1]; return; }
    const double half = 0.5 * nd;
    const double y[2] = {0.5 * (std::sin(half) / half) * delta, std::cos(half)};
    double v[2], beta; Householder(x, v, &beta);
    const double nx = std::sqrt(x[0] * x[0] + x[1] * x[1]);
    const double vy = v[0] * y[0] + v[1] * y[1];
    out[0] = nx * (y[0] - v[0] * beta * vy); out[1] = nx * (y[1] - v[1] * beta * vy);
  }
  static void Jacobian(const double x[2], double J[2]) {   // d Plus / d delta at 0: 0.5 |x| * first column of H
    double v[2], beta; Householder(x, v, &beta);
    const double nx = std::sqrt(x[0] * x[0] + x[1] * x[1]);
    J[0] = 0.5 * nx * (1.0 - beta * v[0] * v[0]); J[1] = 0.5 * nx * (-beta * v[1] * v[0]);
  }
};

struct TrustRegion2d {   // the shared trust-region bookkeeping (Ceres TrustRegionMinimizer + LevenbergMarquardtStrategy): the rules of oracle/trust_region.h,
  lm::Radius rule{1e4};  // which reproduce Ceres' published Powell / hello-world tables (tests/golden/ceres_*_trace.txt)
  double& radius = rule.radius;
  bool reuse_diagonal = false;
  int invalid = 0;
  void Accept(double rel) { rule.Accept(rel, 1e16); reuse_diagonal = false; }
  void Reject() { rule.Reject(); reuse_diagonal = true; }
};

// optimize_points2d (sfm2d.cc:80-120): cameras constant, ALL points free, one joint LM (block-diagonal 2x2 system)
inline void OptimizePoints2d(const Pose2d cams[4], const double* const x[4], int n, double* X) {
  const double kTol = 1e-10;
  double q[4][2], t[4][2];
  for (int i = 0; i < 4; ++i) { q[i][0] = cams[i].m[0]; q[i][1] = cams[i].m[3]; t[i][0] = cams[i].m[2]; t[i][1] = cams[i].m[5]; }
  std::vector<double> H(3 * (size_t)n), g(2 * (size_t)n), scale(2 * (size_t)n, 1.0), diag(2 * (size_t)n), step(2 * (size_t)n), Xc(2 * (size_t)n);
  auto evaluate = [&](const double* P, bool jac) {
    double cost = 0;
    for (int j = 0; j < n; ++j) {
      double h00 = 0, h01 = 0, h11 = 0, g0 = 0, g1 = 0;
      for (int i = 0; i < 4; ++i) {
        double d[6];
        const double r = Residual2d(q[i], t[i], P + 2 * j, x[i] + 2 * j, jac ? d : nullptr);
        cost += 0.5 * r * r;
        if (jac) { h00 += d[4] * d[4]; h01 += d[4] * d[5]; h11 += d[5] * d[5]; g0 += d[4] * r; g1 += d[5] * r; }
      }
      if (jac) { H[3 * j] = h00; H[3 * j + 1] = h01; H[3 * j + 2] = h11; g[2 * j] = g0; g[2 * j + 1] = g1; }
    }
    return cost;
  };
  double cost = evaluate(X, true);
  for (int j = 0; j < n; ++j) { scale[2 * j] = lm::JacobiScale(H[3 * j]); scale[2 * j + 1] = lm::JacobiScale(H[3 * j + 2]); }
  auto gmax = [&]() { double m = 0; for (int i = 0; i < 2 * n; ++i) m = std::fmax(m, std::fabs(g[i])); return m; };
  TrustRegion2d tr;
  bool last_ok = true;
  for (int iter = 1;; ++iter) {
    if (last_ok && gmax() <= kTol) break;
    if (iter > 50 || tr.radius < 1e-32) break;
    if (!tr.reuse_diagonal)
      for (int j = 0; j < n; ++j) {
        diag[2 * j] = lm::ClampDiagonal(scale[2 * j] * scale[2 * j] * H[3 * j], 1e-6, 1e32);
        diag[2 * j + 1] = lm::ClampDiagonal(scale[2 * j + 1] * scale[2 * j + 1] * H[3 * j + 2], 1e-6, 1e32);
      }
    tr.reuse_diagonal = true;
    double model = 0, sn = 0, xn = 0;
    bool valid = true;
    for (int j = 0; j < n; ++j) {
      const double s0 = scale[2 * j], s1 = scale[2 * j + 1];
      const double a = s0 * s0 * H[3 * j] + diag[2 * j] / tr.radius, b = s0 * s1 * H[3 * j + 1], c = s1 * s1 * H[3 * j + 2] + diag[2 * j + 1] / tr.radius;
      const double det = a * c - b * b;
      if (!(det > 0.0)) { valid = false; break; }
      const double r0 = -s0 * g[2 * j], r1 = -s1 * g[2 * j + 1];
      const double d0 = (c * r0 - b * r1) / det, d1 = (a * r1 - b * r0) / det;
      step[2 * j] = s0 * d0; step[2 * j + 1] = s1 * d1;
      // model cost change -(J d)^T (r + J d / 2) = -(g.d + d^T H d / 2)
      const double e0 = step[2 * j], e1 = step[2 * j + 1];
      model -= g[2 * j] * e0 + g[2 * j + 1] * e1 + 0.5 * (H[3 * j] * e0 * e0 + 2.0 * H[3 * j + 1] * e0 * e1 + H[3 * j + 2] * e1 * e1);
      sn += e0 * e0 + e1 * e1; xn += X[2 * j] * X[2 * j] + X[2 * j + 1] * X[2 * j + 1];
      Xc[2 * j] = X[2 * j] + e0; Xc[2 * j + 1] = X[2 * j + 1] + e1;
    }
    if (!valid || !(model > 0.0)) {
      if (++tr.invalid >= 5) break;      // max_num_consecutive_invalid_steps (Ceres default 5)
      tr.Reject(); last_ok = false; continue;
    }
    tr.invalid = 0;
    if (std::sqrt(sn) <= kTol * (std::sqrt(xn) + kTol)) break;
    const double ccost = evaluate(Xc.data(), false);
    const double change = cost - ccost;
    if (std::fabs(change) <= kTol * cost) break;
    const double rel = change / model;
    if (rel > 1e-3) {
      std::memcpy(X, Xc.data(), sizeof(double) * 2 * n);
      cost = evaluate(X, true);
      tr.Accept(rel); last_ok = true;
    } else { tr.Reject(); last_ok = false; }
  }
}

// bundle_adjust2d (sfm2d.cc:122-175): m >= 10 sample points + cameras 1..3; camera 0 constant, the (cos, sin) pairs and
// the translation of camera 1 are unit 2-vectors under HomogeneousVectorParameterization(2)
inline void BundleAdjust2d(Pose2d cams[4], const double* const x[4] /*each m x 2*/, int m, double* X /*m x 2*/) {
  if (m < 10) return;
  const double kTol = 1e-10;
  double q[4][2], t[4][2];
  for (int i = 0; i < 4; ++i) { q[i][0] = cams[i].m[0]; q[i][1] = cams[i].m[3]; t[i][0] = cams[i].m[2]; t[i][1] = cams[i].m[5]; }
  // tangent layout: cam1 (q:1, t:1), cam2 (q:1, t:2), cam3 (q:1, t:2) -> 8 camera columns, then 2 per point
  const int nc = 8, n = nc + 2 * m, nr = 4 * m;
  const int qcol[4] = {-1, 0, 2, 5}, tcol[4] = {-1, 1, 3, 6};
  std::vector<double> J((size_t)nr * n), r(nr), A((size_t)n * n), rhs(n), scale(n, 1.0), diag(n), delta(n), g(n);
  double qc[4][2], tc[4][2];
  std::vector<double> Xc(2 * (size_t)m);
  auto evaluate = [&](const double (*Q)[2], const double (*T)[2], const double* P, bool jac) {
    double cost = 0;
    if (jac) std::fill(J.begin(), J.end(), 0.0);
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < m; ++j) {
        double d[6];
        const int row = i * m + j;
        const double res = Residual2d(Q[i], T[i], P + 2 * j, x[i] + 2 * j, jac ? d : nullptr);
        cost += 0.5 * res * res;
        if (!jac) continue;
        r[row] = res;
        double* Jr = &J[(size_t)row * n];
        if (i > 0) {
          double jq[2]; Homogeneous2::Jacobian(Q[i], jq);
          Jr[qcol[i]] = d[0] * jq[0] + d[1] * jq[1];
          if (i == 1) { double jt[2]; Homogeneous2::Jacobian(T[i], jt); Jr[tcol[i]] = d[2] * jt[0] + d[3] * jt[1]; }
          else { Jr[tcol[i]] = d[2]; Jr[tcol[i] + 1] = d[3]; }
        }
        Jr[nc + 2 * j] = d[4]; Jr[nc + 2 * j + 1] = d[5];
      }
    return cost;
  };
  auto gradient = [&]() { for (int c = 0; c < n; ++c) { double s = 0; for (int row = 0; row < nr; ++row) s += J[(size_t)row * n + c] * r[row]; g[c] = s; } };
  auto gmax = [&]() {   // ||x - Plus(x, -g)||_inf per block
    double mx = 0;
    for (int i = 1; i < 4; ++i) {
      double o[2]; Homogeneous2::Plus(q[i], -g[qcol[i]], o);
      mx = std::fmax(mx, std::fmax(std::fabs(q[i][0] - o[0]), std::fabs(q[i][1] - o[1])));
      if (i == 1) { Homogeneous2::Plus(t[i], -g[tcol[i]], o); mx = std::fmax(mx, std::fmax(std::fabs(t[i][0] - o[0]), std::fabs(t[i][1] - o[1]))); }
      else mx = std::fmax(mx, std::fmax(std::fabs(g[tcol[i]]), std::fabs(g[tcol[i] + 1])));
    }
    for (int c = nc; c < n; ++c) mx = std::fmax(mx, std::fabs(g[c]));
    return mx;
  };
  double cost = evaluate(q, t, X, true);
  gradient();
  for (int c = 0; c < n; ++c) { double s = 0; for (int row = 0; row < nr; ++row) s += J[(size_t)row * n + c] * J[(size_t)row * n + c]; scale[c] = lm::JacobiScale(s); }
  TrustRegion2d tr;
  bool last_ok = true;
  for (int iter = 1;; ++iter) {
    if (last_ok && gmax() <= kTol) break;
    if (iter > 50 || tr.radius < 1e-32) break;
    if (!tr.reuse_diagonal)
      for (int c = 0; c < n; ++c) { double s = 0; for (int row = 0; row < nr; ++row) s += J[(size_t)row * n + c] * J[(size_t)row * n + c]; diag[c] = lm::ClampDiagonal(scale[c] * scale[c] * s, 1e-6, 1e32); }
    tr.reuse_diagonal = true;
    // (J_s^T J_s + D^2/radius) d = -J_s^T r by dense Cholesky (Ceres: exact Schur elimination, the same step)
    for (int a = 0; a < n; ++a) {
      for (int b = 0; b <= a; ++b) { double s = 0; for (int row = 0; row < nr; ++row) s += J[(size_t)row * n + a] * J[(size_t)row * n + b]; A[(size_t)a * n + b] = A[(size_t)b * n + a] = s * scale[a] * scale[b]; }
      A[(size_t)a * n + a] += diag[a] / tr.radius;
      rhs[a] = -scale[a] * g[a];
    }
    bool valid = CholeskyFactor(n, A.data());
    double model = 0;
    if (valid) {
      CholeskySolve(n, A.data(), rhs.data());
      for (int c = 0; c < n; ++c) delta[c] = rhs[c] * scale[c];
      for (int row = 0; row < nr; ++row) { double jd = 0; for (int c = 0; c < n; ++c) jd += J[(size_t)row * n + c] * delta[c]; model -= jd * (r[row] + 0.5 * jd); }
      if (!(model > 0.0)) valid = false;
    }
    if (!valid) { if (++tr.invalid >= 5) break; tr.Reject(); last_ok = false; continue; }
    tr.invalid = 0;
    double sn = 0, xn = 0;
    for (int c = 0; c < n; ++c) sn += delta[c] * delta[c];
    for (int i = 1; i < 4; ++i) xn += q[i][0] * q[i][0] + q[i][1] * q[i][1] + t[i][0] * t[i][0] + t[i][1] * t[i][1];
    for (int j = 0; j < 2 * m; ++j) xn += X[j] * X[j];
    for (int i = 0; i < 4; ++i) { qc[i][0] = q[i][0]; qc[i][1] = q[i][1]; tc[i][0] = t[i][0]; tc[i][1] = t[i][1]; }
    for (int i = 1; i < 4; ++i) {
      Homogeneous2::Plus(q[i], delta[qcol[i]], qc[i]);
      if (i == 1) Homogeneous2::Plus(t[i], delta[tcol[i]], tc[i]);
      else { tc[i][0] = t[i][0] + delta[tcol[i]]; tc[i][1] = t[i][1] + delta[tcol[i] + 1]; }
    }
    for (int j = 0; j < 2 * m; ++j) Xc[j] = X[j] + delta[nc + j];
    if (std::sqrt(sn) <= kTol * (std::sqrt(xn) + kTol)) break;
    const double ccost = evaluate(qc, tc, Xc.data(), false);
    const double change = cost - ccost;
    if (std::fabs(change) <= kTol * cost) break;
    const double rel = change / model;
    if (rel > 1e-3) {
      for (int i = 1; i < 4; ++i) { q[i][0] = qc[i][0]; q[i][1] = qc[i][1]; t[i][0] = tc[i][0]; t[i][1] = tc[i][1]; }
      std::memcpy(X, Xc.data(), sizeof(double) * 2 * m);
      cost = evaluate(q, t, X, true);
      gradient();
      tr.Accept(rel); last_ok = true;
    } else { tr.Reject(); last_ok = false; }
  }
  for (int i = 0; i < 4; ++i) { cams[i].m[0] = q[i][0]; cams[i].m[1] = -q[i][1]; cams[i].m[3] = q[i][1]; cams[i].m[4] = q[i][0]; cams[i].m[2] = t[i][0]; cams[i].m[5] = t[i][1]; }
}

// FourView2dEstimator (sfm2d.h:48-97): the model carries its points, exactly as the reference's Reconstruction
struct FourView2dRec { Pose2d cams[4]; std::vector<double> X; };
class FourView2dEstimator {
 public:
  FourView2dEstimator(const double* x /*4 x n x 2*/, int n, double thr, const double frames[12]) : n_(n), thr_(thr), x_(x, x + 8 * (size_t)n) {
    for (size_t i = 0; i < 4 * (size_t)n; ++i) { const double nr = std::sqrt(x_[2 * i] * x_[2 * i] + x_[2 * i + 1] * x_[2 * i + 1]); x_[2 * i] /= nr; x_[2 * i + 1] /= nr; }
    for (int i = 0; i < 12; ++i) fr_[i] = frames[i];
    for (int j = 0; j < 4; ++j) xs_[j] = x_.data() + 2 * (size_t)n * j;
  }
  int min_sample_size() const { return 5; }
  int non_minimal_sample_size() const { return 10; }
  int num_data() const { return n_; }
  int MinimalSolver(const std::vector<int>& sample, std::vector<FourView2dRec>* models) const {
    FourView2dModel mm[16];
    const int c = FourView2dMinimalSolver(x_.data(), n_, sample.data(), (int)sample.size(), fr_, fr_ + 4, fr_ + 8, mm);
    models->resize(c);
    for (int k = 0; k < c; ++k) {
      FourView2dRec& R = (*models)[k];
      for (int j = 0; j < 4; ++j) R.cams[j] = mm[k].cams[j];
      R.X.resize(2 * (size_t)n_);
      for (int i = 0; i < n_; ++i) ThreeViewTriangulate2d(R.cams, xs_, i, &R.X[2 * i]);
    }
    return c;
  }
  int NonMinimalSolver(const std::vector<int>& sample, FourView2dRec* model) const {   // sfm2d.cc:446-467
    std::vector<FourView2dRec> models;
    MinimalSolver(sample, &models);
    double best = std::numeric_limits<double>::max();
    for (size_t k = 0; k < models.size(); ++k) {
      double score = 0;
      for (int j = 0; j < n_; ++j) score += std::min(thr_, EvaluateModelOnPoint(models[k], j));
      if (score < best) { best = score; *model = models[k]; }
    }
    return models.empty() ? 0 : 1;
  }
  double EvaluateModelOnPoint(const FourView2dRec& m, int i) const { return FourView2dError(m.cams, xs_, i, &m.X[2 * (size_t)i]); }
  void LeastSquares(const std::vector<int>& sample, FourView2dRec* model) const {       // sfm2d.cc:469-489
    const int m = (int)sample.size();
    std::vector<double> xs(8 * (size_t)m), X(2 * (size_t)m);
    const double* xp[4];
    for (int j = 0; j < 4; ++j) { xp[j] = &xs[2 * (size_t)m * j]; for (int i = 0; i < m; ++i) { xs[2 * ((size_t)m * j + i)] = xs_[j][2 * sample[i]]; xs[2 * ((size_t)m * j + i) + 1] = xs_[j][2 * sample[i] + 1]; } }
    for (int i = 0; i < m; ++i) { X[2 * i] = model->X[2 * (size_t)sample[i]]; X[2 * i + 1] = model->X[2 * (size_t)sample[i] + 1]; }
    BundleAdjust2d(model->cams, xp, m, X.data());
    for (int i = 0; i < m; ++i) { model->X[2 * (size_t)sample[i]] = X[2 * i]; model->X[2 * (size_t)sample[i] + 1] = X[2 * i + 1]; }
    OptimizePoints2d(model->cams, xs_, n_, model->X.data());
  }
 private:
  int n_; double thr_; std::vector<double> x_; const double* xs_[4]; double fr_[12];
};

}  // namespace oracle

// ORACLE — TEST INFRASTRUCTURE ONLY.
//
// extern "C" surface of the CPU restatement, loaded with ctypes by tests/, by
// __graft_entry__.smoke() and by bench.py's cpu_baseline leg — and by nothing else.
// The product (privacy_preserving_sfm_amd/) never links, imports or calls this library.
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstring>
#include <random>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "absolute_pose.h"
#include "bundle_adjustment.h"
#include "init_solvers.h"
#include "filter.h"
#include "triangulation.h"
#include "line_cost.h"
#include "ransac.h"

using namespace oracle;

extern "C" {

int orc_num_params(int model) { return NumParams(model); }

void orc_world_to_image(int model, const double* params, double u, double v, double* xy) {
  WorldToImage<double>(model, params, u, v, &xy[0], &xy[1]);
}

// one residual block, ambient Jacobians as ceres::CostFunction::Evaluate would return them
int orc_line_cost(int model, const double* line, const double* q, const double* t, const double* X,
                  const double* cam, double* r, double* Jq, double* Jt, double* JX, double* Jcam) {
  return LineCostEvaluate(model, line, q, t, X, cam, r, Jq, Jt, JX, Jcam) ? 0 : -1;
}

// batched evaluation in the layout of the device kernel K1:
//   residuals 2M ; J_pose M x (2x6) = [d r/d rot-tangent (3) | d r/d t (3)] ; J_point M x (2x3) ;
//   J_cam M x (2 x cam_stride) (optional)
// ambient != 0: J_pose is M x (2x7) = [d r/d q (w,x,y,z) | d r/d t]
int orc_ba_eval(int64_t M, const double* lines, const int32_t* obs_pose, const int32_t* obs_point,
                const int32_t* pose_camera, const int32_t* camera_model, const double* poses,
                const double* points, const double* intr, int ambient, double* residuals, double* Jpose,
                double* Jpoint, double* Jcam, int cam_stride) {
#pragma omp parallel for schedule(static)
  for (int64_t o = 0; o < M; ++o) {
    const int c = obs_pose[o], p = obs_point[o], k = pose_camera[c];
    const int model = camera_model[k], ncam = NumParams(model);
    double r[2], Jq[8], Jt[6], JX[6], Jc[24];
    const bool want_j = Jpose || Jpoint || Jcam;
    LineCostEvaluate(model, lines + 3 * o, poses + 7 * c, poses + 7 * c + 4, points + 3 * p, intr + kCamStride * k, r,
                     want_j ? Jq : nullptr, want_j ? Jt : nullptr, want_j ? JX : nullptr, Jcam ? Jc : nullptr);
    if (residuals) { residuals[2 * o] = r[0]; residuals[2 * o + 1] = r[1]; }
    if (Jpose) {
      if (ambient) {
        for (int row = 0; row < 2; ++row) {
          for (int i = 0; i < 4; ++i) Jpose[14 * o + 7 * row + i] = Jq[4 * row + i];
          for (int i = 0; i < 3; ++i) Jpose[14 * o + 7 * row + 4 + i] = Jt[3 * row + i];
        }
      } else {
        double Pl[12];
        QuaternionPlusJacobian(poses + 7 * c, Pl);
        for (int row = 0; row < 2; ++row) {
          for (int j = 0; j < 3; ++j) {
            double s = 0; for (int i = 0; i < 4; ++i) s += Jq[4 * row + i] * Pl[3 * i + j];
            Jpose[12 * o + 6 * row + j] = s;
          }
          for (int j = 0; j < 3; ++j) Jpose[12 * o + 6 * row + 3 + j] = Jt[3 * row + j];
        }
      }
    }
    if (Jpoint) for (int i = 0; i < 6; ++i) Jpoint[6 * o + i] = JX[i];
    if (Jcam) for (int row = 0; row < 2; ++row) for (int j = 0; j < ncam; ++j) Jcam[(2 * o + row) * (int64_t)cam_stride + j] = Jc[row * ncam + j];
  }
  return 0;
}

struct orc_ba_problem {
  int32_t num_poses, num_points, num_cameras, loss_type;
  int64_t num_obs;
  double loss_scale;
  const double* lines; const int32_t* obs_pose; const int32_t* obs_point; const int32_t* pose_camera;
  const int32_t* camera_model; const uint8_t* pose_const; const uint8_t* tvec_const_mask;
  const uint8_t* point_const; const uint16_t* camera_const_mask;
};
struct orc_ba_options {
  int32_t max_num_iterations, max_num_consecutive_invalid_steps;
  double function_tolerance, gradient_tolerance, parameter_tolerance;
  double initial_trust_region_radius, max_trust_region_radius, min_trust_region_radius;
  double min_relative_decrease, min_lm_diagonal, max_lm_diagonal;
  int32_t jacobi_scaling, blocked_cholesky;   // blocked_cholesky: timing path only (bench.py cpu_baseline)
  int32_t iterative_schur, max_linear_solver_iterations;   // ITERATIVE_SCHUR + SCHUR_JACOBI (reference: above 1000 images)
  double eta;
};
struct orc_ba_summary {
  double initial_cost, final_cost;
  int32_t num_successful_steps, num_unsuccessful_steps, termination, num_iterations;
  double time_s;
  int32_t linear_solver_iterations, reserved;
};

static BAProblem ToProblem(const orc_ba_problem* d) {
  BAProblem p;
  p.num_poses = d->num_poses; p.num_points = d->num_points; p.num_cameras = d->num_cameras; p.num_obs = d->num_obs;
  p.lines = d->lines; p.obs_pose = d->obs_pose; p.obs_point = d->obs_point; p.pose_camera = d->pose_camera;
  p.camera_model = d->camera_model; p.pose_const = d->pose_const; p.tvec_const_mask = d->tvec_const_mask;
  p.point_const = d->point_const; p.camera_const_mask = d->camera_const_mask;
  p.loss_type = d->loss_type; p.loss_scale = d->loss_scale;
  return p;
}
static BAOptions ToOptions(const orc_ba_options* o) {
  BAOptions b;
  b.max_num_iterations = o->max_num_iterations; b.max_num_consecutive_invalid_steps = o->max_num_consecutive_invalid_steps;
  b.function_tolerance = o->function_tolerance; b.gradient_tolerance = o->gradient_tolerance; b.parameter_tolerance = o->parameter_tolerance;
  b.initial_trust_region_radius = o->initial_trust_region_radius; b.max_trust_region_radius = o->max_trust_region_radius;
  b.min_trust_region_radius = o->min_trust_region_radius; b.min_relative_decrease = o->min_relative_decrease;
  b.min_lm_diagonal = o->min_lm_diagonal; b.max_lm_diagonal = o->max_lm_diagonal; b.jacobi_scaling = o->jacobi_scaling != 0;
  b.blocked_cholesky = o->blocked_cholesky != 0;
  b.iterative_schur = o->iterative_schur != 0; b.max_linear_solver_iterations = o->max_linear_solver_iterations; b.eta = o->eta;
  return b;
}

// The oracle's trust-region rules (oracle/trust_region.h) on Powell's function, Ceres' examples/powell.cc: rows of 7 doubles as orc_ba_solve's trace, the end point in x[4];
// function_tolerance etc. as given (Ceres' example runs with the Solver::Options defaults and max_num_iterations = 100)
int orc_powell_trace(double* x, int max_num_iterations, double function_tolerance, double gradient_tolerance, double parameter_tolerance, double* trace, int trace_cap) {
  lm::Options o;
  o.max_num_iterations = max_num_iterations; o.function_tolerance = function_tolerance; o.gradient_tolerance = gradient_tolerance; o.parameter_tolerance = parameter_tolerance;
  x[0] = 3.0; x[1] = -1.0; x[2] = 0.0; x[3] = 1.0;
  const std::vector<lm::Iteration> it = lm::PowellTrace(x, o);
  for (size_t i = 0; i < it.size() && (int)i < trace_cap; ++i) {
    const double row[7] = {it[i].cost, it[i].cost_change, it[i].gradient_max_norm, it[i].step_norm, it[i].relative_decrease, it[i].radius, (double)it[i].successful};
    for (int k = 0; k < 7; ++k) trace[7 * i + k] = row[k];
  }
  return (int)it.size();
}

// the same for Ceres' examples/helloworld.cc (f = 10 - x from x = 0.5)
int orc_helloworld_trace(double* x, int max_num_iterations, double function_tolerance, double gradient_tolerance, double parameter_tolerance, double* trace, int trace_cap) {
  lm::Options o;
  o.max_num_iterations = max_num_iterations; o.function_tolerance = function_tolerance; o.gradient_tolerance = gradient_tolerance; o.parameter_tolerance = parameter_tolerance;
  x[0] = 0.5;
  const std::vector<lm::Iteration> it = lm::HelloWorldTrace(x, o);
  for (size_t i = 0; i < it.size() && (int)i < trace_cap; ++i) {
    const double row[7] = {it[i].cost, it[i].cost_change, it[i].gradient_max_norm, it[i].step_norm, it[i].relative_decrease, it[i].radius, (double)it[i].successful};
    for (int k = 0; k < 7; ++k) trace[7 * i + k] = row[k];
  }
  return (int)it.size();
}

// in-place LM solve; trace (optional): per iteration 7 doubles {cost, cost_change, gmax, step_norm, rel, radius, ok}
int orc_ba_solve(const orc_ba_problem* d, const orc_ba_options* o, double* poses, double* points, double* intr,
                 orc_ba_summary* out, double* trace, int trace_cap) {
  BAProblem p = ToProblem(d);
  BASolver solver(p, poses, points, intr);
  const auto t0 = std::chrono::steady_clock::now();
  BASummary s = solver.Solve(ToOptions(o));
  const auto t1 = std::chrono::steady_clock::now();
  out->initial_cost = s.initial_cost; out->final_cost = s.final_cost;
  out->num_successful_steps = s.num_successful_steps; out->num_unsuccessful_steps = s.num_unsuccessful_steps;
  out->termination = s.termination; out->num_iterations = (int)s.iterations.size() - 1;
  out->time_s = std::chrono::duration<double>(t1 - t0).count();
  out->linear_solver_iterations = s.linear_solver_iterations; out->reserved = 0;
  if (trace) for (int i = 0; i < (int)s.iterations.size() && i < trace_cap; ++i) {
    const BAIteration& it = s.iterations[i];
    double* t = trace + 7 * i;
    t[0] = it.cost; t[1] = it.cost_change; t[2] = it.gradient_max_norm; t[3] = it.step_norm; t[4] = it.relative_decrease; t[5] = it.radius; t[6] = it.successful;
  }
  return 0;
}

double orc_ba_cost(const orc_ba_problem* d, const double* poses, const double* points, const double* intr, double* residuals) {
  BAProblem p = ToProblem(d);
  BASolver solver(p, const_cast<double*>(poses), const_cast<double*>(points), const_cast<double*>(intr));
  return solver.Cost(poses, points, intr, residuals);
}

// The damped, Jacobi-scaled reduced camera system at the given point for a given trust-region radius:
// S (nc x nc row-major), rhs (nc), the full step (nc + 3*variable points, SCALED coordinates), the
// Jacobi scale and the gradient.  Returns nc (or -1 if the system is not positive definite).
int orc_ba_reduced_system(const orc_ba_problem* d, const orc_ba_options* o, double radius, double* poses, double* points,
                          double* intr, double* S, double* rhs, double* step, double* scale_out, double* grad_out,
                          int32_t* n_point_cols) {
  BAProblem p = ToProblem(d);
  BASolver solver(p, poses, points, intr);
  BAOptions opt = ToOptions(o);
  solver.Evaluate();
  const int n = solver.num_camera_cols() + solver.num_point_cols();
  if (n_point_cols) *n_point_cols = solver.num_point_cols();
  std::vector<double> g, scale(n, 1.0), diag, D(n), st, Sv, rv;
  solver.Gradient(&g);
  if (opt.jacobi_scaling) { std::vector<double> cn; solver.SquaredColumnNorms(nullptr, &cn); for (int i = 0; i < n; ++i) scale[i] = 1.0 / (1.0 + std::sqrt(cn[i])); }
  solver.SquaredColumnNorms(&scale, &diag);
  for (int i = 0; i < n; ++i) { diag[i] = std::fmin(std::fmax(diag[i], opt.min_lm_diagonal), opt.max_lm_diagonal); D[i] = std::sqrt(diag[i] / radius); }
  const bool ok = solver.SolveNormalEquations(scale, D, &st, &Sv, &rv);
  if (S) std::memcpy(S, Sv.data(), sizeof(double) * Sv.size());
  if (rhs) std::memcpy(rhs, rv.data(), sizeof(double) * rv.size());
  if (step && ok) std::memcpy(step, st.data(), sizeof(double) * st.size());
  if (scale_out) std::memcpy(scale_out, scale.data(), sizeof(double) * n);
  if (grad_out) std::memcpy(grad_out, g.data(), sizeof(double) * n);
  return ok ? solver.num_camera_cols() : -1;
}

// ---- absolute pose path ----------------------------------------------------------------
void orc_line_residuals(int n, const double* lines, const double* pts, const double* P, double* residuals) {
  SquaredLineReprojectionError(n, lines, pts, P, residuals);
}
void orc_support(int n, const double* residuals, double max_residual, uint64_t* num_inliers, double* residual_sum) {
  const Support s = EvaluateSupport(n, residuals, max_residual);
  *num_inliers = s.num_inliers; *residual_sum = s.residual_sum;
}
// InlierSupportMeasurer::Compare (support_measurement.cc:51-60) on (count, sum) pairs
int orc_support_better(uint64_t n1, double sum1, uint64_t n2, double sum2) {
  Support a, b; a.num_inliers = n1; a.residual_sum = sum1; b.num_inliers = n2; b.residual_sum = sum2;
  return SupportBetter(a, b) ? 1 : 0;
}
int orc_re3q3(const double* coeffs, double* solutions, const double* affine) { return Re3q3(coeffs, solutions, true, affine); }
int orc_p6l(const double* lines6, const double* points6, const uint8_t* aligned6, double* models, const double* mix,
            const double* affine) {
  return P6LEstimate(lines6, points6, aligned6, models, mix, affine);
}
// first `count` k-subsets the persistent-permutation sampler draws from mt19937(seed)
void orc_sampler(uint32_t seed, uint32_t n, int k, int64_t count, uint32_t* out) {
  MT19937 rng(seed); RandomSampler s(k, &rng); s.Initialize(n);
  for (int64_t i = 0; i < count; ++i) s.Sample(out + i * k);
}
void orc_mt19937(uint32_t seed, int64_t count, uint32_t* out) { MT19937 g(seed); for (int64_t i = 0; i < count; ++i) out[i] = g.Next(); }
// the toolchain's own distribution, to pin the restated integer rule against it
void orc_std_uniform(uint32_t seed, int64_t count, const uint32_t* lo, const uint32_t* hi, uint32_t* out_std, uint32_t* out_restated) {
  std::mt19937 g(seed); MT19937 h(seed);
  for (int64_t i = 0; i < count; ++i) {
    std::uniform_int_distribution<uint32_t> dist(lo[i], hi[i]);
    out_std[i] = dist(g);
    out_restated[i] = UniformInt(h, lo[i], hi[i]);
  }
}
uint64_t orc_compute_num_trials(uint64_t num_inliers, uint64_t num_samples, double confidence, double mult) {
  return ComputeNumTrials(num_inliers, num_samples, confidence, mult, 6);
}

struct orc_ransac_options { double max_error, min_inlier_ratio, confidence, dyn_num_trials_multiplier; uint64_t min_num_trials, max_num_trials; };
struct orc_ransac_report { int32_t success, best_model_idx; uint64_t num_trials, num_inliers; double residual_sum; double model[12]; int64_t best_trial; double time_s; };

int orc_p6l_ransac(const orc_ransac_options* o, int n, const double* lines, const double* pts, const uint8_t* aligned,
                   uint32_t seed, orc_ransac_report* rep, uint8_t* inlier_mask) {
  RansacOptions opt; opt.max_error = o->max_error; opt.min_inlier_ratio = o->min_inlier_ratio; opt.confidence = o->confidence;
  opt.dyn_num_trials_multiplier = o->dyn_num_trials_multiplier; opt.min_num_trials = o->min_num_trials; opt.max_num_trials = o->max_num_trials;
  const auto t0 = std::chrono::steady_clock::now();
  RansacReport r = P6LRansac(opt, n, lines, pts, aligned, seed);
  const auto t1 = std::chrono::steady_clock::now();
  rep->success = r.success; rep->num_trials = r.num_trials; rep->num_inliers = r.support.num_inliers; rep->residual_sum = r.support.residual_sum;
  std::memcpy(rep->model, r.model, sizeof(r.model)); rep->best_trial = r.best_trial; rep->best_model_idx = r.best_model_idx;
  rep->time_s = std::chrono::duration<double>(t1 - t0).count();
  if (inlier_mask) { if (r.success) for (int i = 0; i < n; ++i) inlier_mask[i] = r.inlier_mask[i]; else std::memset(inlier_mask, 0, n); }
  return 0;
}

// throughput probe for the cpu_baseline: H hypotheses (pre-drawn six-tuples), each solved and every
// returned model scored against all n correspondences, single thread, as the reference loop does
// (optim/ransac.h:213-249).  Returns seconds; *models_scored gets the total number of models.
double orc_p6l_hypotheses_timed(int n, const double* lines, const double* pts, const uint8_t* aligned, int64_t H,
                                const uint32_t* samples, double max_residual, int64_t* models_scored, uint64_t* best_inliers) {
  std::vector<double> residuals(n);
  int64_t nm_total = 0; uint64_t best = 0;
  const auto t0 = std::chrono::steady_clock::now();
  for (int64_t h = 0; h < H; ++h) {
    double l6[18], p6[18]; uint8_t a6[6];
    for (int i = 0; i < 6; ++i) { const uint32_t id = samples[6 * h + i]; for (int c = 0; c < 3; ++c) { l6[3 * i + c] = lines[3 * id + c]; p6[3 * i + c] = pts[3 * id + c]; } a6[i] = aligned ? aligned[id] : 0; }
    double models[96];
    const int nm = P6LEstimate(l6, p6, a6, models);
    for (int m = 0; m < nm; ++m) {
      SquaredLineReprojectionError(n, lines, pts, models + 12 * m, residuals.data());
      const Support s = EvaluateSupport(n, residuals.data(), max_residual);
      best = std::max<uint64_t>(best, s.num_inliers);
    }
    nm_total += nm;
  }
  const auto t1 = std::chrono::steady_clock::now();
  if (models_scored) *models_scored = nm_total;
  if (best_inliers) *best_inliers = best;
  return std::chrono::duration<double>(t1 - t0).count();
}

// n <= 0: back to the OpenMP default of this process (what OMP_NUM_THREADS / the CPU affinity mask gave it at start-up - NOT
// omp_get_num_procs(): in a container that is the host's thread count, far above the cores the process may use)
void orc_set_num_threads(int n) {
#ifdef _OPENMP
  static const int default_threads = omp_get_max_threads();
  omp_set_num_threads(n > 0 ? n : default_threads);
#else
  (void)n;
#endif
}

int orc_num_threads() {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

}  // extern "C"

// ---- four-view initialisation path -------------------------------------------------------------------
namespace {
// the same toy line-fitting Solver as oracle/ref_ransaclib_trace.cpp, driven by the RESTATED LO-MSAC
struct ToyLine { double a, b, c; };
class ToyLineSolver {
 public:
  ToyLineSolver(const std::vector<double>& x, const std::vector<double>& y) : x_(x), y_(y) {}
  int min_sample_size() const { return 2; }
  int non_minimal_sample_size() const { return 6; }
  int num_data() const { return static_cast<int>(x_.size()); }
  int MinimalSolver(const std::vector<int>& s, std::vector<ToyLine>* models) const {
    models->clear();
    const double dx = x_[s[1]] - x_[s[0]], dy = y_[s[1]] - y_[s[0]];
    const double n = std::sqrt(dx * dx + dy * dy);
    if (n < 1e-12) return 0;
    ToyLine l{-dy / n, dx / n, 0};
    l.c = -(l.a * x_[s[0]] + l.b * y_[s[0]]);
    models->push_back(l);
    return 1;
  }
  int NonMinimalSolver(const std::vector<int>& s, ToyLine* m) const {
    double mx = 0, my = 0;
    for (int i : s) { mx += x_[i]; my += y_[i]; }
    mx /= s.size(); my /= s.size();
    double sxx = 0, sxy = 0, syy = 0;
    for (int i : s) { sxx += (x_[i] - mx) * (x_[i] - mx); sxy += (x_[i] - mx) * (y_[i] - my); syy += (y_[i] - my) * (y_[i] - my); }
    const double th = 0.5 * std::atan2(2 * sxy, sxx - syy);
    m->a = -std::sin(th); m->b = std::cos(th); m->c = -(m->a * mx + m->b * my);
    return 1;
  }
  double EvaluateModelOnPoint(const ToyLine& m, int i) const { const double d = m.a * x_[i] + m.b * y_[i] + m.c; return d * d; }
  void LeastSquares(const std::vector<int>& s, ToyLine* m) const { NonMinimalSolver(s, m); }
 private:
  std::vector<double> x_, y_;
};
}  // namespace

extern "C" {

// writes the same text as oracle/_ref/ransaclib_trace (built from the reference's own headers)
int orc_lomsac_line_trace(int n, char* out, int cap) {
  std::mt19937 g(7);
  std::uniform_real_distribution<double> u(-1, 1);
  std::normal_distribution<double> nz(0, 0.01);
  std::vector<double> x(n), y(n);
  for (int i = 0; i < n; ++i) { x[i] = u(g); y[i] = (i % 3 == 0) ? u(g) : 0.5 * x[i] + 0.1 + nz(g); }
  std::string txt = "samples";
  UniformSampling sampler(0, n, 2);
  char buf[256];
  for (int t = 0; t < 16; ++t) { std::vector<int> s; sampler.Sample(&s); snprintf(buf, sizeof(buf), " %d %d", s[0], s[1]); txt += buf; }
  txt += "\n";
  LORansacOptions opt;
  opt.min_num_iterations = 100; opt.max_num_iterations = 1000; opt.squared_inlier_threshold = 0.03 * 0.03; opt.random_seed = 0;
  ToyLineSolver solver(x, y);
  LocallyOptimizedMSAC<ToyLine, ToyLineSolver> lomsac;
  RansacStatistics st;
  ToyLine best{0, 0, 0};
  const int ninl = lomsac.EstimateModel(opt, solver, &best, &st);
  snprintf(buf, sizeof(buf), "inliers %d iterations %d lo %d score %.17g ratio %.17g\n", ninl, st.num_iterations, st.number_lo_iterations,
           st.best_model_score, st.inlier_ratio);
  txt += buf;
  snprintf(buf, sizeof(buf), "model %.17g %.17g %.17g\n", best.a, best.b, best.c);
  txt += buf;
  for (double eps : {0.1, 0.25, 0.5, 0.9}) { snprintf(buf, sizeof(buf), "numiter %.2f %u\n", eps, NumRequiredIterations(eps, 0.0001, 5, 100, 10000)); txt += buf; }
  if ((int)txt.size() + 1 > cap) return -1;
  std::memcpy(out, txt.c_str(), txt.size() + 1);
  return (int)txt.size();
}

struct orc_lomsac_options { uint32_t min_num_iterations, max_num_iterations; double success_probability, squared_inlier_threshold; uint32_t random_seed; int32_t final_least_squares; };
struct orc_lomsac_stats { uint32_t num_iterations; int32_t best_num_inliers; double best_model_score, inlier_ratio; int32_t number_lo_iterations, pad; };
static LORansacOptions ToLo(const orc_lomsac_options* o) {
  LORansacOptions r; r.min_num_iterations = o->min_num_iterations; r.max_num_iterations = o->max_num_iterations;
  r.success_probability = o->success_probability; r.squared_inlier_threshold = o->squared_inlier_threshold; r.random_seed = o->random_seed;
  r.final_least_squares = o->final_least_squares != 0; return r;
}
static void FromStats(const RansacStatistics& st, orc_lomsac_stats* out, int32_t* inlier_idx) {
  out->num_iterations = st.num_iterations; out->best_num_inliers = st.best_num_inliers; out->best_model_score = st.best_model_score;
  out->inlier_ratio = st.inlier_ratio; out->number_lo_iterations = st.number_lo_iterations;
  if (inlier_idx) for (size_t i = 0; i < st.inlier_indices.size(); ++i) inlier_idx[i] = st.inlier_indices[i];
}

int orc_abspose2d_nonminimal(const double* x, const double* X, int n, const int32_t* sample, int m, double* P) {
  AbsolutePose2dEstimator est(x, X, n);
  std::vector<int> s(sample, sample + m);
  Pose2d p;
  const int r = est.NonMinimalSolver(s, &p);
  std::memcpy(P, p.m, sizeof(p.m));
  return r;
}
int orc_abspose2d_lomsac(const double* x, const double* X, int n, const orc_lomsac_options* o, double* P, orc_lomsac_stats* st, int32_t* inlier_idx) {
  AbsolutePose2dEstimator est(x, X, n);
  LocallyOptimizedMSAC<Pose2d, AbsolutePose2dEstimator> lomsac;
  RansacStatistics rs; Pose2d best{};
  const int inl = lomsac.EstimateModel(ToLo(o), est, &best, &rs);
  std::memcpy(P, best.m, sizeof(best.m));
  FromStats(rs, st, inlier_idx);
  return inl;
}
// AbsPoseSolver of FourView2dEstimator (x NOT normalised by this call: the estimator's ctor normalises)
int orc_abspose_solver2d(const double* x, const double* X, const int32_t* sample, int m, double* P) {
  std::vector<int> s(sample, sample + m);
  Pose2d p;
  const int r = AbsPoseSolver2d(s, x, X, &p);
  std::memcpy(P, p.m, sizeof(p.m));
  return r;
}
// triangulate every point from views 0..2 and evaluate the four-view 1D bearing error (sfm2d.cc:194-213, 302-319)
double orc_fourview2d_score(const double* cams /*4x6*/, const double* x /*4 x n x 2, unit bearings*/, int n, double thr, double* X_out, double* err_out,
                            int32_t* num_inliers) {
  Pose2d P[4];
  for (int j = 0; j < 4; ++j) std::memcpy(P[j].m, cams + 6 * j, sizeof(P[j].m));
  const double* xs[4] = {x, x + 2 * (size_t)n, x + 4 * (size_t)n, x + 6 * (size_t)n};
  double score = 0; int inl = 0;
  for (int i = 0; i < n; ++i) {
    double X[2];
    ThreeViewTriangulate2d(P, xs, i, X);
    const double e = FourView2dError(P, xs, i, X);
    if (X_out) { X_out[2 * i] = X[0]; X_out[2 * i + 1] = X[1]; }
    if (err_out) err_out[i] = e;
    score += std::min(e, thr);
    inl += e < thr;
  }
  if (num_inliers) *num_inliers = inl;
  return score;
}

struct PlanarInputs { const double* poses; const double* lines; const double* Rg; int n; double thr; };
static PlanarOffsetEstimator MakePlanar(const double* poses, const double* lines, int n, const double* Rg, double thr) {
  const double* l[4] = {lines, lines + 3 * (size_t)n, lines + 6 * (size_t)n, lines + 9 * (size_t)n};
  return PlanarOffsetEstimator(poses, l, n, Rg, thr);
}
// minimal solver for a batch of 3-samples: offsets (t_y of cameras 1..3) and the four cameras
int orc_planar_minimal(const double* poses, const double* lines, int n, const double* Rg, const int32_t* samples, int num, int sample_size,
                       double* offsets /*num x 3*/, double* cams /*num x 4 x 12*/) {
  PlanarOffsetEstimator est = MakePlanar(poses, lines, n, Rg, 1.0);
  for (int h = 0; h < num; ++h) {
    std::vector<int> s(samples + (size_t)h * sample_size, samples + (size_t)(h + 1) * sample_size);
    double tt[3];
    if (!est.SolveOffsets(s, tt)) { tt[0] = tt[1] = tt[2] = std::nan(""); }
    for (int k = 0; k < 3; ++k) offsets[3 * h + k] = tt[k];
    if (cams) { Pose34 c[4]; est.CamsFromOffsets(tt, c); for (int j = 0; j < 4; ++j) std::memcpy(cams + ((size_t)h * 4 + j) * 12, c[j].m, sizeof(c[j].m)); }
  }
  return 0;
}
// triangulate all + error per point for one set of offsets; returns the MSAC score
double orc_planar_score(const double* poses, const double* lines, int n, const double* Rg, const double* offsets, double thr, double* X_out,
                        double* err_out, int32_t* num_inliers) {
  PlanarOffsetEstimator est = MakePlanar(poses, lines, n, Rg, thr);
  PlanarOffsetModel m;
  est.CamsFromOffsets(offsets, m.cams);
  est.FourViewTriangulate(m.cams, &m.X);
  double score = 0; int inl = 0;
  for (int i = 0; i < n; ++i) {
    const double e = est.EvaluateModelOnPoint(m, i);
    if (err_out) err_out[i] = e;
    score += std::min(e, thr);
    inl += e < thr;
  }
  if (X_out) std::memcpy(X_out, m.X.data(), sizeof(double) * 3 * (size_t)n);
  if (num_inliers) *num_inliers = inl;
  return score;
}
int orc_planar_lomsac(const double* poses, const double* lines, int n, const double* Rg, const orc_lomsac_options* o, double* cams /*4x12*/,
                      orc_lomsac_stats* st, int32_t* inlier_idx) {
  PlanarOffsetEstimator est = MakePlanar(poses, lines, n, Rg, o->squared_inlier_threshold);
  LocallyOptimizedMSAC<PlanarOffsetModel, PlanarOffsetEstimator> lomsac;
  RansacStatistics rs; PlanarOffsetModel best;
  for (int j = 0; j < 4; ++j) std::memset(best.cams[j].m, 0, sizeof(best.cams[j].m));
  const int inl = lomsac.EstimateModel(ToLo(o), est, &best, &rs);
  for (int j = 0; j < 4; ++j) std::memcpy(cams + 12 * j, best.cams[j].m, sizeof(best.cams[j].m));
  FromStats(rs, st, inlier_idx);
  return inl;
}

}  // extern "C"

extern "C" int orc_fourview2d_minimal(const double* x /*4 x n x 2 unit*/, int n, const int32_t* samples, int num, int sample_size, const double* A123 /*12*/,
                                      double* cams_out /*num x 16 x 24*/, int32_t* count_out) {
  for (int h = 0; h < num; ++h) {
    std::vector<int> s(samples + (size_t)h * sample_size, samples + (size_t)(h + 1) * sample_size);
    FourView2dModel models[16];
    const int c = FourView2dMinimalSolver(x, n, s.data(), sample_size, A123, A123 + 4, A123 + 8, models);
    count_out[h] = c;
    for (int m = 0; m < c; ++m) for (int j = 0; j < 4; ++j) std::memcpy(cams_out + (((size_t)h * 16 + m) * 4 + j) * 6, models[m].cams[j].m, sizeof(double) * 6);
  }
  return 0;
}

// FourView2dEstimator::LeastSquares on a model given by its cameras; X_inout (n x 2) = the model's points
extern "C" int orc_fourview2d_least_squares(const double* x, int n, const int32_t* sample, int m, const double* frames, double* cams_inout /*24*/,
                                            double* X_inout /*n x 2*/) {
  FourView2dEstimator est(x, n, 1.0, frames);
  FourView2dRec rec;
  for (int j = 0; j < 4; ++j) std::memcpy(rec.cams[j].m, cams_inout + 6 * j, sizeof(double) * 6);
  rec.X.assign(X_inout, X_inout + 2 * (size_t)n);
  std::vector<int> s(sample, sample + m);
  est.LeastSquares(s, &rec);
  for (int j = 0; j < 4; ++j) std::memcpy(cams_inout + 6 * j, rec.cams[j].m, sizeof(double) * 6);
  std::memcpy(X_inout, rec.X.data(), sizeof(double) * 2 * (size_t)n);
  return 0;
}
extern "C" int orc_fourview2d_lomsac(const double* x, int n, const orc_lomsac_options* o, const double* frames, double* cams_out /*24*/, double* X_out /*n x 2*/,
                                     orc_lomsac_stats* st, int32_t* inlier_idx) {
  FourView2dEstimator est(x, n, o->squared_inlier_threshold, frames);
  LocallyOptimizedMSAC<FourView2dRec, FourView2dEstimator> lomsac;
  RansacStatistics rs; FourView2dRec best;
  best.X.assign(2 * (size_t)n, 0.0);
  for (int j = 0; j < 4; ++j) std::memset(best.cams[j].m, 0, sizeof(best.cams[j].m));
  const int inl = lomsac.EstimateModel(ToLo(o), est, &best, &rs);
  for (int j = 0; j < 4; ++j) std::memcpy(cams_out + 6 * j, best.cams[j].m, sizeof(double) * 6);
  if (X_out) std::memcpy(X_out, best.X.data(), sizeof(double) * 2 * (size_t)n);
  FromStats(rs, st, inlier_idx);
  return inl;
}

extern "C" int64_t orc_filter_points3d(int64_t M, int P, int C, const double* lines, const int32_t* obs_pose, const int32_t* obs_point, const uint8_t* obs_aligned,
                                       const int32_t* pose_camera, const int32_t* camera_model, const int32_t* cam_size, const double* poses, const double* points,
                                       const double* intr, double max_reproj_error, double min_tri_angle_deg, const uint8_t* point_subset, uint8_t* obs_deleted,
                                       uint8_t* point_deleted, double* point_error) {
  return FilterPoints3D(M, P, C, lines, obs_pose, obs_point, obs_aligned, pose_camera, camera_model, cam_size, poses, points, intr, kCamStride, max_reproj_error,
                        min_tri_angle_deg, point_subset, obs_deleted, point_deleted, point_error);
}
extern "C" int64_t orc_filter_negative_depth(int64_t M, const int32_t* obs_pose, const int32_t* obs_point, const double* poses, const double* points, uint8_t* obs_negative) {
  return FilterObservationsWithNegativeDepth(M, obs_pose, obs_point, poses, points, obs_negative);
}

struct orc_triangulation_options { double min_tri_angle; int32_t residual_type; int32_t pad; orc_ransac_options ransac; };
extern "C" int orc_triangulate_tracks(int32_t T, const int32_t* track_start, const double* lines, const int32_t* obs_view, int32_t V, const double* P, const double* centers,
                                      const int32_t* view_camera, const int32_t* camera_model, const double* intr, const int32_t* cam_size,
                                      const orc_triangulation_options* o, uint8_t* success, double* xyz, uint8_t* inlier_mask, int32_t* num_trials) {
  std::vector<TriView> views(V);
  for (int v = 0; v < V; ++v) {
    std::memcpy(views[v].P, P + 12 * v, sizeof(double) * 12); std::memcpy(views[v].center, centers + 3 * v, sizeof(double) * 3);
    const int k = view_camera[v];
    views[v].model = camera_model[k]; views[v].params = intr + (size_t)kCamStride * k; views[v].width = cam_size[2 * k]; views[v].height = cam_size[2 * k + 1];
  }
  TriangulationOptions opt;
  opt.min_tri_angle = o->min_tri_angle; opt.residual_type = o->residual_type;
  opt.ransac.max_error = o->ransac.max_error; opt.ransac.min_inlier_ratio = o->ransac.min_inlier_ratio; opt.ransac.confidence = o->ransac.confidence;
  opt.ransac.dyn_num_trials_multiplier = o->ransac.dyn_num_trials_multiplier; opt.ransac.min_num_trials = o->ransac.min_num_trials;
  opt.ransac.max_num_trials = o->ransac.max_num_trials;
  for (int t = 0; t < T; ++t) {
    const int e0 = track_start[t], n = track_start[t + 1] - e0;
    std::vector<const TriView*> vs(n);
    for (int i = 0; i < n; ++i) vs[i] = &views[obs_view[e0 + i]];
    const TriangulationReport r = EstimateTriangulation(opt, n, lines + 3 * (size_t)e0, vs.data());
    success[t] = r.success; num_trials[t] = (int32_t)r.num_trials;
    for (int i = 0; i < 3; ++i) xyz[3 * t + i] = r.xyz[i];
    for (int i = 0; i < n; ++i) inlier_mask[e0 + i] = r.success ? (uint8_t)r.inlier_mask[i] : 0;
  }
  return 0;
}

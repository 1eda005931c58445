// Runs the C++ host mirror (ppsfm/ppsfm.hpp) ON THE DEVICE: the three top-level callers of SURVEY.md 8b
//   EstimateAbsolutePoseFromLines      (reference src/estimators/pose.cc:52-94)
//   BundleAdjustmentProblem::Solve      (reference src/optim/bundle_adjustment.cc:260-320, flat form)
//   init::initialize_reconstruction     (reference src/init/initializer.cc:58-216)
// Driven by tests/test_gpu_cpp_mirror.py: inputs come from a text file of whitespace-separated numbers (written by the
// test from the same synthetic scenes the Python mirror tests use), results go to stdout with 17 significant digits and
// are compared there with the oracle / the Python mirror.
//   usage: cpp_mirror_gpu_test <pose|ba|init> <input file>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>

#include "../ppsfm/ppsfm.hpp"

namespace {
struct Reader {
  std::ifstream in;
  explicit Reader(const char* path) : in(path) { if (!in) { std::fprintf(stderr, "cannot open %s\n", path); std::exit(3); } }
  double d() { double v; if (!(in >> v)) { std::fprintf(stderr, "short input\n"); std::exit(3); } return v; }
  long long i() { long long v; if (!(in >> v)) { std::fprintf(stderr, "short input\n"); std::exit(3); } return v; }
};
void PrintVec(const char* name, const double* v, size_t n) {
  std::printf("%s", name);
  for (size_t k = 0; k < n; ++k) std::printf(" %.17g", v[k]);
  std::printf("\n");
}

int RunPose(Reader& r) {
  const int n = (int)r.i();
  ppsfm::RANSACOptions o;
  o.max_error = r.d(); o.min_inlier_ratio = r.d(); o.confidence = r.d(); o.dyn_num_trials_multiplier = r.d();
  o.min_num_trials = (size_t)r.i(); o.max_num_trials = (size_t)r.i();
  const unsigned seed = (unsigned)r.i();
  ppsfm::FeatureLines X(n);
  std::vector<ppsfm::Vector3d> Y(n);
  for (int k = 0; k < n; ++k) {
    for (int c = 0; c < 3; ++c) X[k].line[c] = r.d();
    X[k].is_aligned = r.i() != 0;
    for (int c = 0; c < 3; ++c) Y[k][c] = r.d();
  }
  // the RANSAC<P6LEstimator> shape (Report) ...
  ppsfm::AbsolutePoseFromLinesRANSAC ransac(o, seed);
  const auto rep = ransac.Estimate(X, Y);
  std::printf("report %d %zu %zu %.17g\n", (int)rep.success, rep.num_trials, rep.support.num_inliers, rep.support.residual_sum);
  PrintVec("model", rep.model.data(), 12);
  std::printf("mask ");
  for (char m : rep.inlier_mask) std::printf("%d", (int)m);
  std::printf("\n");
  // ... and the top-level caller
  ppsfm::Vector4d q; ppsfm::Vector3d t; size_t ninl = 0; std::vector<char> mask;
  const bool ok = ppsfm::EstimateAbsolutePoseFromLines(o, X, Y, &q, &t, &ninl, &mask, seed);
  std::printf("pose %d %zu\n", (int)ok, ninl);
  PrintVec("qvec", q.data(), 4);
  PrintVec("tvec", t.data(), 3);
  // Estimator concept: Estimate on the first six pairs, Residuals of the winning model over all pairs
  ppsfm::P6LEstimator est;
  const auto models = est.Estimate(ppsfm::FeatureLines(X.begin(), X.begin() + 6), std::vector<ppsfm::Vector3d>(Y.begin(), Y.begin() + 6));
  std::printf("p6l %zu\n", models.size());
  for (const auto& m : models) PrintVec("p6l_model", m.data(), 12);
  std::vector<double> res;
  est.Residuals(X, Y, rep.model, &res);
  PrintVec("residuals", res.data(), res.size());
  return 0;
}

int RunBA(Reader& r) {
  pp_ba_problem_desc d{};
  d.num_poses = (int32_t)r.i(); d.num_points = (int32_t)r.i(); d.num_cameras = (int32_t)r.i(); d.num_obs = r.i();
  d.loss_type = (int32_t)r.i(); d.loss_scale = r.d();
  const int C = d.num_poses, P = d.num_points, K = d.num_cameras;
  const long long M = d.num_obs;
  std::vector<double> lines(3 * M), poses(7 * C), points(3 * P), intr((size_t)PP_CAM_STRIDE * K);
  std::vector<int32_t> obs_pose(M), obs_point(M), pose_camera(C), camera_model(K);
  std::vector<uint8_t> pose_const(C), tvec_mask(C), point_const(P);
  std::vector<uint16_t> cam_mask(K);
  for (auto& v : lines) v = r.d();
  for (auto& v : obs_pose) v = (int32_t)r.i();
  for (auto& v : obs_point) v = (int32_t)r.i();
  for (auto& v : pose_camera) v = (int32_t)r.i();
  for (auto& v : camera_model) v = (int32_t)r.i();
  for (auto& v : pose_const) v = (uint8_t)r.i();
  for (auto& v : tvec_mask) v = (uint8_t)r.i();
  for (auto& v : point_const) v = (uint8_t)r.i();
  for (auto& v : cam_mask) v = (uint16_t)r.i();
  for (auto& v : poses) v = r.d();
  for (auto& v : points) v = r.d();
  for (auto& v : intr) v = r.d();
  d.lines = lines.data(); d.obs_pose = obs_pose.data(); d.obs_point = obs_point.data(); d.pose_camera = pose_camera.data();
  d.camera_model = camera_model.data(); d.pose_const = pose_const.data(); d.tvec_const_mask = tvec_mask.data();
  d.point_const = point_const.data(); d.camera_const_mask = cam_mask.data();
  pp_ba_options o;
  pp_ba_options_default(&o);
  o.max_num_iterations = (int32_t)r.i(); o.gradient_tolerance = r.d();
  struct Counter { int calls = 0; } counter;
  o.iteration_callback = [](void* ctx, const pp_ba_iteration_summary*) -> int32_t { ++static_cast<Counter*>(ctx)->calls; return PP_SOLVER_CONTINUE; };
  o.iteration_callback_ctx = &counter;
  ppsfm::BundleAdjustmentProblem problem(d);
  problem.SetParameters(poses.data(), points.data(), intr.data());
  pp_ba_summary s;
  const bool usable = problem.Solve(o, &s);
  problem.GetParameters(poses.data(), points.data(), intr.data());
  std::printf("summary %d %d %d %d %d %.17g %.17g %d\n", (int)usable, s.termination, s.num_iterations, s.num_successful_steps, s.num_residuals,
              s.initial_cost, s.final_cost, counter.calls);
  PrintVec("poses", poses.data(), poses.size());
  PrintVec("points", points.data(), points.size());
  return 0;
}

int RunInit(Reader& r) {
  std::vector<ppsfm::FeatureLines> lines(4);
  std::vector<ppsfm::Vector3d> gravity(4);
  for (int v = 0; v < 4; ++v) {
    for (int c = 0; c < 3; ++c) gravity[v][c] = r.d();
    const int n = (int)r.i();
    lines[v].resize(n);
    for (int k = 0; k < n; ++k) {
      for (int c = 0; c < 3; ++c) lines[v][k].line[c] = r.d();
      lines[v][k].is_aligned = r.i() != 0;
    }
  }
  std::vector<ppsfm::init::Pose> poses;
  double ratio = 0;
  const bool ok = ppsfm::init::initialize_reconstruction(lines, gravity, ppsfm::init::InitOptions(), &poses, &ratio);
  std::printf("init %d %.17g %zu\n", (int)ok, ratio, poses.size());
  for (const auto& p : poses) PrintVec("pose", p.data(), 12);
  return 0;
}
}  // namespace

int main(int argc, char** argv) {
  if (argc != 3) { std::fprintf(stderr, "usage: %s <pose|ba|init> <input file>\n", argv[0]); return 2; }
  const std::string mode = argv[1];
  Reader r(argv[2]);
  try {
    if (mode == "pose") return RunPose(r);
    if (mode == "ba") return RunBA(r);
    if (mode == "init") return RunInit(r);
  } catch (const ppsfm::Error& e) {
    std::fprintf(stderr, "ppsfm::Error %d: %s\n", e.code, e.what());
    return 4;
  }
  return 2;
}

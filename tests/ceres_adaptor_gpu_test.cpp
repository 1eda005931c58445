// Exercises ppsfm/ceres_adaptor.hpp on the device WITHOUT Ceres: this program stands where ceres::Problem::Evaluate would —
// it owns the parameter blocks, calls EvaluationCallback::PrepareForEvaluation, then CostFunction::Evaluate on every residual
// block with the argument shapes Ceres uses (all Jacobians, some null, none) — and prints what the blocks returned.
// Built against tests/stubs/ceres/ceres.h (interface shapes only, see there).  Driven by tests/test_gpu_ceres_adaptor.py.
//   usage: ceres_adaptor_gpu_test <input file>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <memory>

#include "../ppsfm/ceres_adaptor.hpp"

struct Reader {
  std::ifstream in;
  explicit Reader(const char* path) : in(path) { if (!in) { std::fprintf(stderr, "cannot open %s\n", path); std::exit(3); } }
  double d() { double v; if (!(in >> v)) { std::fprintf(stderr, "short input\n"); std::exit(3); } return v; }
  long long i() { long long v; if (!(in >> v)) { std::fprintf(stderr, "short input\n"); std::exit(3); } return v; }
};

constexpr int kN = 4;   // SIMPLE_RADIAL: f, cx, cy, k (base/camera_models.h)

int main(int argc, char** argv) {
  if (argc != 2) return 2;
  Reader r(argv[1]);
  pp_ba_problem_desc d{};
  d.num_poses = (int32_t)r.i(); d.num_points = (int32_t)r.i(); d.num_cameras = (int32_t)r.i(); d.num_obs = r.i();
  d.loss_type = 0; d.loss_scale = 1.0;
  const int C = d.num_poses, P = d.num_points, K = d.num_cameras;
  const long long M = d.num_obs;
  std::vector<double> lines(3 * M);
  std::vector<int32_t> obs_pose(M), obs_point(M), pose_camera(C), camera_model(K);
  std::vector<uint8_t> pose_const(C);
  for (auto& v : lines) v = r.d();
  for (auto& v : obs_pose) v = (int32_t)r.i();
  for (auto& v : obs_point) v = (int32_t)r.i();
  for (auto& v : pose_camera) v = (int32_t)r.i();
  for (auto& v : camera_model) v = (int32_t)r.i();
  for (auto& v : pose_const) v = (uint8_t)r.i();
  // the "Reconstruction": separately owned parameter blocks, as Image / Point3D / Camera objects hold them
  std::vector<std::vector<double>> qvec(C, std::vector<double>(4)), tvec(C, std::vector<double>(3)), xyz(P, std::vector<double>(3)), cam(K, std::vector<double>(kN));
  for (int c = 0; c < C; ++c) { for (double& v : qvec[c]) v = r.d(); for (double& v : tvec[c]) v = r.d(); }
  for (auto& p : xyz) for (double& v : p) v = r.d();
  for (auto& k : cam) for (double& v : k) v = r.d();
  d.lines = lines.data(); d.obs_pose = obs_pose.data(); d.obs_point = obs_point.data(); d.pose_camera = pose_camera.data();
  d.camera_model = camera_model.data(); d.pose_const = pose_const.data();
  using namespace ppsfm::ceres_adaptor;
  try {
    BatchedLineEvaluator eval(d, 0, /*want_cam=*/true);
    for (int c = 0; c < C; ++c) eval.SetPoseBlocks(c, qvec[c].data(), tvec[c].data());
    for (int p = 0; p < P; ++p) eval.SetPointBlock(p, xyz[p].data());
    for (int k = 0; k < K; ++k) eval.SetCameraBlock(k, cam[k].data());
    std::vector<std::unique_ptr<ceres::CostFunction>> blocks;
    for (long long o = 0; o < M; ++o) {
      if (pose_const[obs_pose[o]]) blocks.emplace_back(new SlicedConstantPoseLineCostFunction<kN>(&eval, o));
      else blocks.emplace_back(new SlicedLineCostFunction<kN>(&eval, o));
    }
    // block sizes as the reference's factories declare them (cost_functions.h:55-60, 130-137)
    std::printf("sizes %d", blocks[0]->num_residuals());
    for (int s : blocks[0]->parameter_block_sizes()) std::printf(" %d", s);
    std::printf("\n");
    // 1) residuals only (a trial point)
    eval.PrepareForEvaluation(false, true);
    double cost = 0;
    for (long long o = 0; o < M; ++o) {
      double res[2];
      const int c = obs_pose[o], p = obs_point[o], k = pose_camera[c];
      const double* params4[4] = {qvec[c].data(), tvec[c].data(), xyz[p].data(), cam[k].data()};
      const double* params2[2] = {xyz[p].data(), cam[k].data()};
      if (!blocks[o]->Evaluate(pose_const[c] ? params2 : params4, res, nullptr)) return 5;
      cost += 0.5 * (res[0] * res[0] + res[1] * res[1]);
    }
    std::printf("cost %.17g\n", cost);
    // a block asked for Jacobians it has no evaluation of must fail, not return stale data
    {
      double res[2], j0[8], j1[6], j2[6], j3[2 * kN];
      double* J4[4] = {j0, j1, j2, j3};
      double* J2[2] = {j2, j3};
      const int c = obs_pose[0];
      const double* params[4] = {nullptr, nullptr, nullptr, nullptr};
      std::printf("stale %d\n", (int)blocks[0]->Evaluate(params, res, pose_const[c] ? J2 : J4));
    }
    // 2) move a parameter block in place (what Ceres does between evaluations), evaluate with Jacobians
    xyz[0][0] += 0.125;
    eval.PrepareForEvaluation(true, true);
    for (long long o = 0; o < M; ++o) {
      double res[2], j0[8], j1[6], j2[6], j3[2 * kN];
      const int c = obs_pose[o];
      const double* params[4] = {nullptr, nullptr, nullptr, nullptr};     // the blocks never read them: the callback gathered already
      if (pose_const[c]) {
        double* J[2] = {j2, (o % 3 == 0) ? nullptr : j3};                  // Ceres leaves out the Jacobians of constant blocks
        if (!blocks[o]->Evaluate(params, res, J)) return 6;
        std::printf("c %lld %.17g %.17g", o, res[0], res[1]);
        for (double v : j2) std::printf(" %.17g", v);
        if (J[1]) for (double v : j3) std::printf(" %.17g", v);
        std::printf("\n");
      } else {
        double* J[4] = {j0, j1, (o % 5 == 0) ? nullptr : j2, j3};
        if (!blocks[o]->Evaluate(params, res, J)) return 6;
        std::printf("v %lld %.17g %.17g", o, res[0], res[1]);
        for (double v : j0) std::printf(" %.17g", v);
        for (double v : j1) std::printf(" %.17g", v);
        if (J[2]) for (double v : j2) std::printf(" %.17g", v);
        for (double v : j3) std::printf(" %.17g", v);
        std::printf("\n");
      }
    }
    // 3) same point again, Jacobians already there: no new device work is needed (the call must be a no-op, results unchanged)
    eval.PrepareForEvaluation(true, false);
    double res[2];
    const double* params[4] = {nullptr, nullptr, nullptr, nullptr};
    blocks[M - 1]->Evaluate(params, res, nullptr);
    std::printf("again %.17g %.17g\n", res[0], res[1]);
  } catch (const std::exception& e) {
    std::fprintf(stderr, "exception: %s\n", e.what());
    return 4;
  }
  return 0;
}

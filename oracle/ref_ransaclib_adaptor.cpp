// ORACLE — TEST INFRASTRUCTURE.  Reference-derived checker (oracle/_ref/ransaclib_adaptor).
//
// The reference's OWN LO-MSAC driver (/root/reference/lib/RansacLib/RansacLib/ransac.h, std-only, included with -I from where
// it lies, never copied) instantiated over the product's RansacLib Solver-concept adaptors (ppsfm/ransaclib_solvers.hpp), i.e.
// exactly what a maintainer gets by swapping the estimator types at reference src/init/initializer.cc:119-123, 201-206 and
// src/init/sfm2d_test.cc:177-183.  Built only in the build container (the reference is absent on the GPU box; the binary
// travels there with the repository snapshot) and run by tests/test_gpu_ransaclib_adaptor.py, which compares the run of the
// reference's driver with the library's own replay of that driver (pp_planar_lomsac / pp_pose2d_lomsac / pp_fourview2d_lomsac).
//   usage: ransaclib_adaptor <planar|pose2d|fourview2d> <input file>
#include <RansacLib/ransac.h>

#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>

#include "../ppsfm/ransaclib_solvers.hpp"

namespace {
struct Reader {
  std::ifstream in;
  explicit Reader(const char* path) : in(path) { if (!in) { std::fprintf(stderr, "cannot open %s\n", path); std::exit(3); } }
  double d() { double v; if (!(in >> v)) { std::fprintf(stderr, "short input\n"); std::exit(3); } return v; }
  long long i() { long long v; if (!(in >> v)) { std::fprintf(stderr, "short input\n"); std::exit(3); } return v; }
};
ransac_lib::LORansacOptions ReadOptions(Reader& r) {
  ransac_lib::LORansacOptions o;
  o.min_num_iterations_ = (uint32_t)r.i(); o.max_num_iterations_ = (uint32_t)r.i(); o.squared_inlier_threshold_ = r.d();
  o.random_seed_ = (unsigned)r.i(); o.final_least_squares_ = r.i() != 0;
  return o;
}
void PrintStats(const ransac_lib::RansacStatistics& st, int ninl) {
  std::printf("stats %d %u %d %d %.17g %.17g\n", ninl, st.num_iterations, st.best_num_inliers, st.number_lo_iterations, st.best_model_score, st.inlier_ratio);
  std::printf("inliers");
  for (int i : st.inlier_indices) std::printf(" %d", i);
  std::printf("\n");
}
}  // namespace

int main(int argc, char** argv) {
  if (argc != 3) return 2;
  const std::string mode = argv[1];
  Reader r(argv[2]);
  using namespace ppsfm::init;
  try {
    if (mode == "planar") {
      const int n = (int)r.i();
      std::vector<Pose> poses(4);
      std::vector<std::array<double, 9>> Rg(4);
      std::vector<std::vector<ppsfm::Vector3d>> lines(4, std::vector<ppsfm::Vector3d>(n));
      for (auto& p : poses) for (double& v : p) v = r.d();
      for (auto& g : Rg) for (double& v : g) v = r.d();
      for (auto& view : lines) for (auto& l : view) for (double& v : l) v = r.d();
      const ransac_lib::LORansacOptions o = ReadOptions(r);
      PlanarOffsetSolver solver(poses, lines, Rg, o.squared_inlier_threshold_);
      ransac_lib::LocallyOptimizedMSAC<PlanarOffsetSolver::Reconstruction, PlanarOffsetSolver::ReconstructionVector, PlanarOffsetSolver> lomsac;
      ransac_lib::RansacStatistics st;
      PlanarOffsetSolver::Reconstruction best;
      const int ninl = lomsac.EstimateModel(o, solver, &best, &st);
      PrintStats(st, ninl);
      std::printf("cams");
      for (const auto& c : best.cams) for (double v : c) std::printf(" %.17g", v);
      std::printf("\n");
    } else if (mode == "pose2d") {
      const int n = (int)r.i();
      std::vector<Vector2d> x(n), X(n);
      for (auto& p : x) for (double& v : p) v = r.d();
      for (auto& p : X) for (double& v : p) v = r.d();
      const ransac_lib::LORansacOptions o = ReadOptions(r);
      AbsolutePose2dSolver solver(x, X);
      ransac_lib::LocallyOptimizedMSAC<Pose2d, std::vector<Pose2d>, AbsolutePose2dSolver> lomsac;
      ransac_lib::RansacStatistics st;
      Pose2d best{};
      const int ninl = lomsac.EstimateModel(o, solver, &best, &st);
      PrintStats(st, ninl);
      std::printf("cams");
      for (double v : best) std::printf(" %.17g", v);
      std::printf("\n");
    } else if (mode == "fourview2d") {
      const int n = (int)r.i();
      std::vector<std::vector<Vector2d>> x(4, std::vector<Vector2d>(n));
      for (auto& view : x) for (auto& p : view) for (double& v : p) v = r.d();
      double frames[12];
      for (double& v : frames) v = r.d();
      const ransac_lib::LORansacOptions o = ReadOptions(r);
      FourView2dSolver solver(x, o.squared_inlier_threshold_, frames);
      ransac_lib::LocallyOptimizedMSAC<FourView2dSolver::Reconstruction, FourView2dSolver::ReconstructionVector, FourView2dSolver> lomsac;
      ransac_lib::RansacStatistics st;
      FourView2dSolver::Reconstruction best;
      const int ninl = lomsac.EstimateModel(o, solver, &best, &st);
      PrintStats(st, ninl);
      std::printf("cams");
      for (const auto& c : best.cams) for (double v : c) std::printf(" %.17g", v);
      std::printf("\n");
    } else {
      return 2;
    }
  } catch (const ppsfm::Error& e) {
    std::fprintf(stderr, "ppsfm::Error %d: %s\n", e.code, e.what());
    return 4;
  }
  return 0;
}

// Compile-and-link check of the C++ host mirror (ppsfm/ppsfm.hpp) against libppsfm_hip.so.
// Instantiates the colmap Estimator concept the way optim/ransac.h uses it.
#include <cstdio>

#include "../ppsfm/ppsfm.hpp"

template <typename Estimator>
int UseEstimatorConcept() {
  typename Estimator::X_t x;
  typename Estimator::Y_t y{{0, 0, 0}};
  typename Estimator::M_t m{};
  (void)x; (void)y; (void)m;
  return Estimator::kMinNumSamples;
}

int main() {
  static_assert(ppsfm::P6LEstimator::kMinNumSamples == 6, "kMinNumSamples");
  int n = UseEstimatorConcept<ppsfm::P6LEstimator>();
  ppsfm::RANSACOptions o;
  o.max_error = 0.012;
  o.Check();
  pp_ba_options bo;
  pp_ba_options_default(&bo);
  uint32_t s[12];
  if (pp_sampler_draw(0, 100, 6, 2, s) != PP_OK) return 1;
  int count = -1;
  const int rc = pp_device_count(&count);   // PP_ERR_HIP without a GPU: reported, never aborts
  std::printf("ok kMin=%d iters=%d first_sample=%u devices=%d rc=%d\n", n, bo.max_num_iterations, s[0], count, rc);
  // error path: exceptions carry pp_last_error(), never abort
  try {
    ppsfm::FeatureLines X(3);
    std::vector<ppsfm::Vector3d> Y(2);
    ppsfm::AbsolutePoseFromLinesRANSAC(o).Estimate(X, Y);
    return 2;
  } catch (const ppsfm::Error& e) {
    std::printf("caught: %s\n", e.what());
  }
  // the four-view initialisation entry point instantiates; without a GPU it reports the HIP error as an exception
  try {
    std::vector<ppsfm::FeatureLines> lines(4, ppsfm::FeatureLines(12));
    for (int v = 0; v < 4; ++v)
      for (int i = 0; i < 12; ++i) { lines[v][i].line = ppsfm::Vector3d{{1.0, 0.0, 0.1 * i}}; lines[v][i].is_aligned = i < 6; }
    std::vector<ppsfm::Vector3d> gravity(4, ppsfm::Vector3d{{0.0, 1.0, 0.0}});
    std::vector<ppsfm::init::Pose> poses;
    double ratio = 0;
    const bool ok = ppsfm::init::initialize_reconstruction(lines, gravity, ppsfm::init::InitOptions(), &poses, &ratio);
    std::printf("initialize_reconstruction returned %d\n", (int)ok);
  } catch (const ppsfm::Error& e) {
    std::printf("caught: %s\n", e.what());
  }
  return 0;
}

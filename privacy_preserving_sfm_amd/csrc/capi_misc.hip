// Small non-kernel entry points of the C ABI: error string, defaults, camera helpers, sampler.
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <limits>
#include <random>
#include <system_error>
#include <vector>

#include "camera_models.hpp"
#include "common.hpp"
#include "ransac_host.hpp"

namespace ppsfm {
static thread_local char g_last_error[1024] = "";
void SetLastError(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
  va_end(ap);
}
const char* LastError() { return g_last_error; }

int ApiExceptionToCode(const char* where) {
  try { throw; }
  catch (const std::bad_alloc&) { SetLastError("%s: out of host memory (std::bad_alloc)", where); return PP_ERR_NOMEM; }
  catch (const std::length_error& e) { SetLastError("%s: out of host memory (std::length_error: %s)", where, e.what()); return PP_ERR_NOMEM; }
  catch (const std::exception& e) { SetLastError("%s: C++ exception stopped at the C boundary: %s", where, e.what()); return PP_ERR_INTERNAL; }
  catch (...) { SetLastError("%s: unknown C++ exception stopped at the C boundary", where); return PP_ERR_INTERNAL; }
}
}  // namespace ppsfm

using namespace ppsfm;

extern "C" {

const char* pp_last_error(void) { return LastError(); }

// Test hook of the exception containment (tests/test_capi_host.py): raises the named C++ exception INSIDE a guarded entry point -
// 0 std::bad_alloc, 1 std::runtime_error, 2 a non-std exception, 3 std::bad_alloc in a worker thread of ParallelFor, 4 std::length_error
// (a real over-sized std::vector), 5 std::system_error.  Returns what the boundary made of it.
int pp_debug_raise(int kind) try {
  switch (kind) {
    case 0: throw std::bad_alloc();
    case 1: throw std::runtime_error("pp_debug_raise(1)");
    case 2: throw 42;
    case 3: ParallelFor(4, [](int t) { if (t == 2) throw std::bad_alloc(); }); break;
    case 4: { std::vector<double> v; v.resize(v.max_size() + 1); break; }
    case 5: throw std::system_error(std::make_error_code(std::errc::resource_unavailable_try_again), "pp_debug_raise(5)");
    default: break;
  }
  return PP_OK;
} PP_API_CATCH("pp_debug_raise")

int pp_device_count(int* count) try {
  PP_REQUIRE(count, "pp_device_count: null");
  *count = 0;
  PP_HIP_TRY(hipGetDeviceCount(count));
  return PP_OK;
} PP_API_CATCH("pp_device_count")

int pp_camera_num_params(int model_id) { return CameraNumParams(model_id); }

int pp_camera_image_to_world_threshold(int model_id, const double* params, double threshold_px, double* out) try {
  PP_REQUIRE(params && out && CameraNumParams(model_id) > 0, "pp_camera_image_to_world_threshold: bad argument");
  double f = 0;
  const int nf = CameraNumFocal(model_id);
  for (int i = 0; i < nf; ++i) f += params[i];
  f /= nf;
  *out = threshold_px / f;
  return PP_OK;
} PP_API_CATCH("pp_camera_image_to_world_threshold")

void pp_ba_options_default(pp_ba_options* o) {
  if (!o) return;
  o->max_num_iterations = 100;                // optim/bundle_adjustment.h:86
  o->max_num_consecutive_invalid_steps = 10;  // :88
  o->function_tolerance = 0.0;                // :81-83
  o->gradient_tolerance = 0.0;
  o->parameter_tolerance = 0.0;
  o->initial_trust_region_radius = 1e4;       // Ceres defaults the reference inherits
  o->max_trust_region_radius = 1e16;
  o->min_trust_region_radius = 1e-32;
  o->min_relative_decrease = 1e-3;
  o->min_lm_diagonal = 1e-6;
  o->max_lm_diagonal = 1e32;
  o->jacobi_scaling = 1;
  o->phase_timings = 0;
  o->max_linear_solver_iterations = 200;      // optim/bundle_adjustment.h:87
  o->reserved_ = 0;
  o->eta = 1e-1;                              // Ceres Solver::Options::eta
  o->iteration_callback = nullptr;            // controllers/bundle_adjustment.cc:87-88 registers one
  o->iteration_callback_ctx = nullptr;
}

void pp_ransac_options_default(pp_ransac_options* o) {
  if (!o) return;
  o->max_error = 0.0;                 // optim/ransac.h:47-66
  o->min_inlier_ratio = 0.1;
  o->confidence = 0.99;
  o->dyn_num_trials_multiplier = 3.0;
  o->min_num_trials = 0;
  o->max_num_trials = std::numeric_limits<uint64_t>::max();
  o->seed = 0;                        // util/random.h:46
  o->chunk_trials = 0;
}

int pp_sampler_draw(uint32_t seed, uint32_t n, int32_t k, int64_t count, uint32_t* out) try {
  PP_REQUIRE(out && k > 0 && (uint32_t)k <= n && count >= 0, "pp_sampler_draw: bad argument");
  RandomSampler sampler(k, seed);
  sampler.Initialize(n);
  for (int64_t i = 0; i < count; ++i) sampler.Sample(out + i * k);
  return PP_OK;
} PP_API_CATCH("pp_sampler_draw")

uint64_t pp_ransac_compute_num_trials(uint64_t num_inliers, uint64_t num_samples, double confidence, double mult) {
  return ComputeNumTrials(num_inliers, num_samples, confidence, mult, 6);
}

}  // extern "C"

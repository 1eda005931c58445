// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// CPU restatement of the sequential hypothesise-and-verify loop that drives the six-line solver.
//   RANSAC ctor / ComputeNumTrials / Estimate : reference src/optim/ransac.h:143-156, :158-176, :178-278
//   RandomSampler                              : src/optim/random_sampler.cc:43-62
//   Shuffle / RandomInteger / PRNG             : src/util/random.h:88-128, src/util/random.cc:36-50
//   EstimateAbsolutePoseFromLines glue         : src/estimators/pose.cc:52-94
//
// PRNG: std::mt19937 is fully specified by ISO C++ (10000th output of the default-seeded engine
// is 4123659995); it is restated here from the published MT19937 recurrence.  The integer
// distribution is NOT specified by the standard: the reference gets whatever its libstdc++
// ships.  This restates the libstdc++ >= 11 rule for a 32-bit engine (Lemire's nearly
// divisionless method: 64-bit product, rejection below (2^32 - range) mod range), which is the
// toolchain of this image (g++ 11.4).  tests/ pins it against std::uniform_int_distribution.
#pragma once
#include <cmath>
#include <cstdint>
#include <limits>
#include <vector>
#include "absolute_pose.h"

namespace oracle {

struct MT19937 {
  uint32_t mt[624]; int idx;
  explicit MT19937(uint32_t seed) { Seed(seed); }
  void Seed(uint32_t seed) {
    mt[0] = seed;
    for (int i = 1; i < 624; ++i) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
    idx = 624;
  }
  uint32_t Next() {
    if (idx >= 624) {
      for (int i = 0; i < 624; ++i) {
        const uint32_t y = (mt[i] & 0x80000000u) | (mt[(i + 1) % 624] & 0x7fffffffu);
        mt[i] = mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
      }
      idx = 0;
    }
    uint32_t y = mt[idx++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
  }
};

// uniform integer in [lo, hi] (closed), libstdc++-11 semantics for a 32-bit URBG
inline uint32_t UniformInt(MT19937& g, uint32_t lo, uint32_t hi) {
  const uint32_t urange = hi - lo;
  if (urange == 0xFFFFFFFFu) return g.Next();
  const uint32_t range = urange + 1;
  uint64_t product = (uint64_t)g.Next() * (uint64_t)range;
  uint32_t low = (uint32_t)product;
  if (low < range) {
    const uint32_t threshold = (0u - range) % range;
    while (low < threshold) {
      product = (uint64_t)g.Next() * (uint64_t)range;
      low = (uint32_t)product;
    }
  }
  return (uint32_t)(product >> 32) + lo;
}

// persistent-permutation partial Fisher-Yates sampler (random_sampler.cc:43-62, random.h:120-128)
struct RandomSampler {
  int k; std::vector<uint32_t> perm; MT19937* rng;
  RandomSampler(int k_, MT19937* g) : k(k_), rng(g) {}
  void Initialize(uint32_t total) { perm.resize(total); for (uint32_t i = 0; i < total; ++i) perm[i] = i; }
  void Sample(uint32_t* out) {
    const uint32_t last = (uint32_t)perm.size() - 1;
    for (uint32_t i = 0; i < (uint32_t)k; ++i) { const uint32_t j = UniformInt(*rng, i, last); std::swap(perm[i], perm[j]); }
    for (int i = 0; i < k; ++i) out[i] = perm[i];
  }
};

struct RansacOptions {  // defaults: ransac.h:47-66
  double max_error = 0.0;
  double min_inlier_ratio = 0.1;
  double confidence = 0.99;
  double dyn_num_trials_multiplier = 3.0;
  uint64_t min_num_trials = 0;
  uint64_t max_num_trials = std::numeric_limits<uint64_t>::max();
};

inline uint64_t ComputeNumTrials(uint64_t num_inliers, uint64_t num_samples, double confidence,
                                 double multiplier, int min_samples) {
  const double inlier_ratio = num_inliers / static_cast<double>(num_samples);
  const double nom = 1 - confidence;
  if (nom <= 0) return std::numeric_limits<uint64_t>::max();
  const double denom = 1 - std::pow(inlier_ratio, min_samples);
  if (denom <= 0) return 1;
  return static_cast<uint64_t>(std::ceil(std::log(nom) / std::log(denom) * multiplier));
}

struct RansacReport {
  bool success = false;
  uint64_t num_trials = 0;
  Support support;
  std::vector<char> inlier_mask;
  double model[12] = {0};
  // trace for parity tests (not in the reference's Report): trial that produced the winner, and
  // its index among that trial's models; -1 if none
  int64_t best_trial = -1; int best_model_idx = -1;
};

// RANSAC<P6LEstimator, InlierSupportMeasurer, RandomSampler>::Estimate
inline RansacReport P6LRansac(const RansacOptions& opt_in, int n, const double* lines, const double* pts,
                              const uint8_t* aligned, uint32_t seed) {
  const int kMin = 6;
  RansacOptions opt = opt_in;
  {  // ctor: cap max_num_trials from the a-priori inlier ratio (ransac.h:149-155)
    const uint64_t kNumSamples = 100000;
    const uint64_t dyn = ComputeNumTrials(static_cast<uint64_t>(opt.min_inlier_ratio * kNumSamples), kNumSamples,
                                          opt.confidence, opt.dyn_num_trials_multiplier, kMin);
    opt.max_num_trials = std::min(opt.max_num_trials, dyn);
  }
  RansacReport report;
  if (n < kMin) return report;

  Support best_support; double best_model[12] = {0};
  bool abort = false;
  const double max_residual = opt.max_error * opt.max_error;
  std::vector<double> residuals(n);
  MT19937 rng(seed);
  RandomSampler sampler(kMin, &rng);
  sampler.Initialize((uint32_t)n);
  uint64_t max_num_trials = opt.max_num_trials;
  uint64_t dyn_max_num_trials = max_num_trials;

  for (report.num_trials = 0; report.num_trials < max_num_trials; ++report.num_trials) {
    if (abort) { report.num_trials += 1; break; }
    uint32_t idx[6];
    sampler.Sample(idx);
    double l6[18], p6[18]; uint8_t a6[6];
    for (int i = 0; i < 6; ++i) {
      for (int c = 0; c < 3; ++c) { l6[3 * i + c] = lines[3 * idx[i] + c]; p6[3 * i + c] = pts[3 * idx[i] + c]; }
      a6[i] = aligned ? aligned[idx[i]] : 0;
    }
    double models[96];
    const int nm = P6LEstimate(l6, p6, a6, models);
    for (int m = 0; m < nm; ++m) {
      SquaredLineReprojectionError(n, lines, pts, models + 12 * m, residuals.data());
      const Support s = EvaluateSupport(n, residuals.data(), max_residual);
      if (SupportBetter(s, best_support)) {
        best_support = s;
        std::memcpy(best_model, models + 12 * m, sizeof(best_model));
        report.best_trial = (int64_t)report.num_trials; report.best_model_idx = m;
        dyn_max_num_trials = ComputeNumTrials(best_support.num_inliers, (uint64_t)n, opt.confidence,
                                              opt.dyn_num_trials_multiplier, kMin);
      }
      if (report.num_trials >= dyn_max_num_trials && report.num_trials >= opt.min_num_trials) { abort = true; break; }
    }
  }
  report.support = best_support;
  std::memcpy(report.model, best_model, sizeof(best_model));
  if (report.support.num_inliers < (uint64_t)kMin) return report;
  report.success = true;
  SquaredLineReprojectionError(n, lines, pts, report.model, residuals.data());
  report.inlier_mask.resize(n);
  for (int i = 0; i < n; ++i) report.inlier_mask[i] = residuals[i] <= max_residual;
  return report;
}

}  // namespace oracle

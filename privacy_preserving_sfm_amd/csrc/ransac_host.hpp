// Host-side pieces of the RANSAC driver that must stay sequential and bit-compatible with the
// reference: the sampler and the trial-count rule.
//
//   RandomSampler::{Initialize, Sample}  reference src/optim/random_sampler.cc:43-62
//   Shuffle / RandomInteger              src/util/random.h:88-128 (thread-local std::mt19937,
//                                        std::uniform_int_distribution<uint32_t>(i, last))
//   RANSAC::ComputeNumTrials             src/optim/ransac.h:158-176
// The sampler uses the host toolchain's own <random>, exactly like the reference does, so the
// index stream is the reference's stream on the same libstdc++ by construction.
#pragma once
#include <cmath>
#include <cstdint>
#include <limits>
#include <numeric>
#include <random>
#include <vector>

namespace ppsfm {

class RandomSampler {
 public:
  RandomSampler(int num_samples, uint32_t seed) : k_(num_samples), prng_(seed) {}
  void Initialize(uint32_t total) {
    idx_.resize(total);
    std::iota(idx_.begin(), idx_.end(), 0u);
  }
  // partial Fisher-Yates on the PERSISTENT permutation: sample t depends on all earlier samples
  void Sample(uint32_t* out) {
    const uint32_t last = static_cast<uint32_t>(idx_.size() - 1);
    for (uint32_t i = 0; i < static_cast<uint32_t>(k_); ++i) {
      std::uniform_int_distribution<uint32_t> dist(i, last);
      const uint32_t j = dist(prng_);
      std::swap(idx_[i], idx_[j]);
    }
    for (int i = 0; i < k_; ++i) out[i] = idx_[i];
  }

 private:
  int k_;
  std::mt19937 prng_;
  std::vector<uint32_t> idx_;
};

inline uint64_t ComputeNumTrials(uint64_t num_inliers, uint64_t num_samples, double confidence, double multiplier,
                                 int min_num_samples) {
  const double inlier_ratio = num_inliers / static_cast<double>(num_samples);
  const double nom = 1 - confidence;
  if (nom <= 0) return std::numeric_limits<uint64_t>::max();
  const double denom = 1 - std::pow(inlier_ratio, min_num_samples);
  if (denom <= 0) return 1;
  return static_cast<uint64_t>(std::ceil(std::log(nom) / std::log(denom) * multiplier));
}

}  // namespace ppsfm

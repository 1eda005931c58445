// TEST STUB — NOT Ceres Solver.  Ceres is not installed in this image (and there is no network), so ppsfm/ceres_adaptor.hpp
// could otherwise not even be type-checked here.  This file declares ONLY the three interface shapes the adaptor derives
// from, as the Ceres documentation specifies them (ceres::EvaluationCallback, ceres::CostFunction, ceres::SizedCostFunction),
// with no solver behind them.  tests/ceres_adaptor_gpu_test.cpp plays the role of Ceres' evaluator: it calls
// PrepareForEvaluation and then Evaluate on every residual block, the way ceres::Problem::Evaluate would.
// It pins nothing about Ceres' numerics and is never part of the product or the oracle.
#pragma once
#include <cstdint>
#include <vector>

namespace ceres {

class EvaluationCallback {
 public:
  virtual ~EvaluationCallback() {}
  virtual void PrepareForEvaluation(bool evaluate_jacobians, bool new_evaluation_point) = 0;
};

class CostFunction {
 public:
  CostFunction() : num_residuals_(0) {}
  virtual ~CostFunction() {}
  virtual bool Evaluate(double const* const* parameters, double* residuals, double** jacobians) const = 0;
  const std::vector<int32_t>& parameter_block_sizes() const { return parameter_block_sizes_; }
  int num_residuals() const { return num_residuals_; }

 protected:
  std::vector<int32_t>* mutable_parameter_block_sizes() { return &parameter_block_sizes_; }
  void set_num_residuals(int n) { num_residuals_ = n; }

 private:
  std::vector<int32_t> parameter_block_sizes_;
  int num_residuals_;
};

template <int kNumResiduals, int... Ns>
class SizedCostFunction : public CostFunction {
 public:
  SizedCostFunction() {
    set_num_residuals(kNumResiduals);
    *mutable_parameter_block_sizes() = std::vector<int32_t>{Ns...};
  }
  virtual ~SizedCostFunction() {}
};

}  // namespace ceres

// ceres_adaptor.hpp — keeps the reference's ceres::Problem surface and batches only the cost evaluation on the device
// (INTEGRATION.md 2b).  Needs Ceres >= 2.0 (ceres::EvaluationCallback in Problem::Options); compiled only where Ceres
// exists — it is NOT part of libppsfm_hip.so and nothing in this repository's product path includes it.
//
// What it replaces in the reference:
//   BundleAdjustmentLineCostFunction<CameraModel>::Create -> new ceres::AutoDiffCostFunction<F, 2, 4, 3, 3, kNumParams>
//        (src/base/cost_functions.h:55-60), registered by problem_->AddResidualBlock(cost_function, loss_function, qvec, tvec,
//        xyz, camera_params)                                     (src/optim/bundle_adjustment.cc:401-414)
//   BundleAdjustmentConstantPoseLineCostFunction<CameraModel>::Create -> AutoDiffCostFunction<F, 2, 3, kNumParams>
//        (cost_functions.h:130-137), AddResidualBlock(cost_function, loss_function, xyz, camera_params) (:383-398, :470-486)
// With these classes the residual block of observation o becomes a ceres::SizedCostFunction of the SAME block sizes whose
// Evaluate() copies slice o of one batched K1 evaluation; the batch runs once per evaluation point inside
// ceres::EvaluationCallback::PrepareForEvaluation.  Loss functions, local parameterisations / manifolds, the trust-region
// loop and the linear solver stay Ceres'.  (Replacing ceres::Solve as a whole is INTEGRATION.md 2a / pp_ba_solve.)
//
// Contracts used (Ceres documentation): with an evaluation callback Ceres updates the user's parameter blocks in place
// before PrepareForEvaluation(evaluate_jacobians, new_evaluation_point); CostFunction::Evaluate may be called concurrently
// from Ceres' worker threads (const, read-only here); jacobians may be null and any jacobians[i] may be null;
// jacobians[i] is row-major 2 x block_size_i.
//
// Wiring (what BundleAdjuster::SetUp does, src/optim/bundle_adjustment.cc:326-542):
//   ppsfm::ceres_adaptor::BatchedLineEvaluator eval(desc, device);           // desc: the flat problem, as for pp_ba_create
//   ceres::Problem::Options po; po.evaluation_callback = &eval; ceres::Problem problem(po);
//   for every pose c:    eval.SetPoseBlocks(c, image.Qvec().data(), image.Tvec().data());
//   for every point p:   eval.SetPointBlock(p, point3D.XYZ().data());
//   for every camera k:  eval.SetCameraBlock(k, camera.ParamsData());
//   for every observation o:
//     problem.AddResidualBlock(new ppsfm::ceres_adaptor::SlicedLineCostFunction<N>(&eval, o), loss, qvec, tvec, xyz, cam);
//     (constant pose:  new ppsfm::ceres_adaptor::SlicedConstantPoseLineCostFunction<N>(&eval, o), loss, xyz, cam)
#pragma once
#include <ceres/ceres.h>

#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../include/ppsfm_hip.h"

namespace ppsfm {
namespace ceres_adaptor {

class BatchedLineEvaluator : public ceres::EvaluationCallback {
 public:
  // want_cam: some camera block is variable (refine_focal_length / principal_point / extra_params), so J_cam is needed
  BatchedLineEvaluator(const pp_ba_problem_desc& desc, int device, bool want_cam)
      : C_(desc.num_poses), P_(desc.num_points), K_(desc.num_cameras), M_(desc.num_obs), want_cam_(want_cam),
        qvec_(C_, nullptr), tvec_(C_, nullptr), xyz_(P_, nullptr), cam_(K_, nullptr), poses_(7 * (size_t)C_), points_(3 * (size_t)P_),
        intr_((size_t)PP_CAM_STRIDE * K_, 0.0), cam_np_(K_) {
    if (pp_ba_create(&desc, device, &h_) != PP_OK) throw std::runtime_error(std::string("pp_ba_create: ") + pp_last_error());
    for (int k = 0; k < K_; ++k) cam_np_[k] = pp_camera_num_params(desc.camera_model[k]);
  }
  ~BatchedLineEvaluator() override { pp_ba_destroy(h_); }
  BatchedLineEvaluator(const BatchedLineEvaluator&) = delete;
  BatchedLineEvaluator& operator=(const BatchedLineEvaluator&) = delete;

  // the user-owned parameter blocks Ceres updates in place (Image::Qvec/Tvec, Point3D::XYZ, Camera::ParamsData)
  void SetPoseBlocks(int c, const double* qvec, const double* tvec) { qvec_[c] = qvec; tvec_[c] = tvec; }
  void SetPointBlock(int p, const double* xyz) { xyz_[p] = xyz; }
  void SetCameraBlock(int k, const double* params) { cam_[k] = params; }

  void PrepareForEvaluation(bool evaluate_jacobians, bool new_evaluation_point) override {
    if (!new_evaluation_point && (have_jacobians_ || !evaluate_jacobians)) return;      // same point, nothing new is asked for
    for (int c = 0; c < C_; ++c) {
      std::memcpy(&poses_[7 * (size_t)c], qvec_[c], 4 * sizeof(double));
      std::memcpy(&poses_[7 * (size_t)c + 4], tvec_[c], 3 * sizeof(double));
    }
    for (int p = 0; p < P_; ++p) std::memcpy(&points_[3 * (size_t)p], xyz_[p], 3 * sizeof(double));
    for (int k = 0; k < K_; ++k) std::memcpy(&intr_[(size_t)PP_CAM_STRIDE * k], cam_[k], sizeof(double) * cam_np_[k]);
    Check(pp_ba_set_parameters(h_, poses_.data(), points_.data(), intr_.data()), "pp_ba_set_parameters");
    // jac_mode 1 = the ambient layout Ceres expects from Evaluate: J_q (2x4) | J_t (2x3) per row, J_X 2x3, J_cam 2 x PP_CAM_STRIDE
    Check(pp_ba_eval_host_view(h_, /*jac_mode=*/1, want_cam_ ? 1 : 0, evaluate_jacobians ? 1 : 0, &r_, &jpose_, &jpoint_, &jcam_, nullptr),
          "pp_ba_eval_host_view");
    have_jacobians_ = evaluate_jacobians;
  }

  // slices of the current evaluation (pinned host memory owned by the handle)
  const double* residuals(int64_t o) const { return r_ + 2 * o; }
  const double* jpose(int64_t o) const { return jpose_ ? jpose_ + 14 * o : nullptr; }
  const double* jpoint(int64_t o) const { return jpoint_ ? jpoint_ + 6 * o : nullptr; }
  const double* jcam(int64_t o) const { return jcam_ ? jcam_ + 2 * PP_CAM_STRIDE * o : nullptr; }
  int64_t num_observations() const { return M_; }
  pp_ba_handle handle() const { return h_; }

 private:
  static void Check(int rc, const char* what) {
    if (rc != PP_OK) throw std::runtime_error(std::string(what) + ": " + pp_last_error());
  }
  pp_ba_handle h_ = nullptr;
  int C_, P_, K_;
  int64_t M_;
  bool want_cam_, have_jacobians_ = false;
  std::vector<const double*> qvec_, tvec_, xyz_, cam_;
  std::vector<double> poses_, points_, intr_;
  std::vector<int> cam_np_;
  const double *r_ = nullptr, *jpose_ = nullptr, *jpoint_ = nullptr, *jcam_ = nullptr;
};

namespace detail {
template <int kNumParams>
inline void CopyCameraJacobian(const double* jc, double* out) {      // 2 x PP_CAM_STRIDE (device) -> 2 x kNumParams (Ceres), row-major
  if (!jc) { std::memset(out, 0, sizeof(double) * 2 * kNumParams); return; }   // J_cam was not requested at construction: constant cameras
  std::memcpy(out, jc, sizeof(double) * kNumParams);
  std::memcpy(out + kNumParams, jc + PP_CAM_STRIDE, sizeof(double) * kNumParams);
}
}  // namespace detail

// residual block of one observation with a variable pose: parameter blocks (qvec[4], tvec[3], point3D[3], camera_params[N])
template <int kNumParams>
class SlicedLineCostFunction : public ceres::SizedCostFunction<2, 4, 3, 3, kNumParams> {
 public:
  SlicedLineCostFunction(const BatchedLineEvaluator* eval, int64_t observation) : eval_(eval), o_(observation) {}
  bool Evaluate(double const* const* /*parameters*/, double* residuals, double** jacobians) const override {
    const double* r = eval_->residuals(o_);
    residuals[0] = r[0]; residuals[1] = r[1];
    if (!(r[0] == r[0]) || !(r[1] == r[1])) return false;        // NaN: evaluation failure, Ceres rejects the step
    if (!jacobians) return true;
    const double* jp = eval_->jpose(o_);                          // 2 x 7 row-major: [dq (4) | dt (3)]
    if (!jp) return false;                                        // Jacobians asked for without PrepareForEvaluation(true, ...)
    if (jacobians[0]) { std::memcpy(jacobians[0], jp, 4 * sizeof(double)); std::memcpy(jacobians[0] + 4, jp + 7, 4 * sizeof(double)); }
    if (jacobians[1]) { std::memcpy(jacobians[1], jp + 4, 3 * sizeof(double)); std::memcpy(jacobians[1] + 3, jp + 11, 3 * sizeof(double)); }
    if (jacobians[2]) std::memcpy(jacobians[2], eval_->jpoint(o_), 6 * sizeof(double));
    if (jacobians[3]) detail::CopyCameraJacobian<kNumParams>(eval_->jcam(o_), jacobians[3]);
    return true;
  }

 private:
  const BatchedLineEvaluator* eval_;
  int64_t o_;
};

// residual block of one observation whose pose is constant: parameter blocks (point3D[3], camera_params[N])
template <int kNumParams>
class SlicedConstantPoseLineCostFunction : public ceres::SizedCostFunction<2, 3, kNumParams> {
 public:
  SlicedConstantPoseLineCostFunction(const BatchedLineEvaluator* eval, int64_t observation) : eval_(eval), o_(observation) {}
  bool Evaluate(double const* const* /*parameters*/, double* residuals, double** jacobians) const override {
    const double* r = eval_->residuals(o_);
    residuals[0] = r[0]; residuals[1] = r[1];
    if (!(r[0] == r[0]) || !(r[1] == r[1])) return false;
    if (!jacobians) return true;
    if (!eval_->jpoint(o_)) return false;
    if (jacobians[0]) std::memcpy(jacobians[0], eval_->jpoint(o_), 6 * sizeof(double));
    if (jacobians[1]) detail::CopyCameraJacobian<kNumParams>(eval_->jcam(o_), jacobians[1]);
    return true;
  }

 private:
  const BatchedLineEvaluator* eval_;
  int64_t o_;
};

}  // namespace ceres_adaptor
}  // namespace ppsfm

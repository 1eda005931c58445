// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// CPU restatement of the line-to-point reprojection residual of the reference's
// bundle adjustment and of its derivative.
//   residual                    : reference src/base/cost_functions.h:62-100 (variable pose)
//                                 and :139-178 (constant pose: identical arithmetic, q/t constants)
//   parameter-block dimensions  : :55-60  <2, 4,3,3,kNumParams>,  :130-137 <2, 3,kNumParams>
//   rotation                    : ceres::UnitQuaternionRotatePoint (third-party Ceres, NOT in
//                                 /root/reference, version unpinned) — the published polynomial
//                                 uv = 2 (q_xyz x X); out = X + q_w uv + q_xyz x uv; q is NOT
//                                 re-normalised inside.
// The derivative is obtained exactly as the reference obtains it: by pushing jets
// through the templated residual (Ceres AutoDiffCostFunction semantics).
// Output Jacobians are "ambient": J_q 2x4 (w,x,y,z), J_t 2x3, J_X 2x3, J_cam 2xN, row-major,
// which is what ceres::CostFunction::Evaluate hands back before local parameterisation.
#pragma once
#include "camera_models.h"

namespace oracle {

template <typename T>
inline void RotatePoint(const T q[4], const T X[3], T out[3]) {
  T uv0 = q[2] * X[2] - q[3] * X[1];
  T uv1 = q[3] * X[0] - q[1] * X[2];
  T uv2 = q[1] * X[1] - q[2] * X[0];
  uv0 = uv0 + uv0; uv1 = uv1 + uv1; uv2 = uv2 + uv2;
  out[0] = X[0] + q[0] * uv0;
  out[1] = X[1] + q[0] * uv1;
  out[2] = X[2] + q[0] * uv2;
  out[0] = out[0] + (q[2] * uv2 - q[3] * uv1);
  out[1] = out[1] + (q[3] * uv0 - q[1] * uv2);
  out[2] = out[2] + (q[1] * uv1 - q[2] * uv0);
}

// residual of one line observation (a,b,c), a^2+b^2 = 1, in normalised coordinates
template <typename T>
inline void LineResidual(int model, const double line[3], const T q[4], const T t[3],
                         const T X[3], const T* cam, T r[2]) {
  T p[3];
  RotatePoint(q, X, p);
  p[0] = p[0] + t[0]; p[1] = p[1] + t[1]; p[2] = p[2] + t[2];
  p[0] = p[0] / p[2];
  p[1] = p[1] / p[2];
  const T alpha = T(line[0]) * p[0] + T(line[1]) * p[1] + T(line[2]);
  const T fu = p[0] - alpha * T(line[0]);
  const T fv = p[1] - alpha * T(line[1]);
  T x, y, xf, yf;
  WorldToImage(model, cam, p[0], p[1], &x, &y);
  WorldToImage(model, cam, fu, fv, &xf, &yf);
  r[0] = x - xf;
  r[1] = y - yf;
}

template <int NCAM>
inline void LineCostJetsN(int model, const double line[3], const double q[4], const double t[3],
                          const double X[3], const double* cam, double r[2],
                          double* Jq, double* Jt, double* JX, double* Jcam) {
  constexpr int W = 10 + NCAM;
  typedef Jet<W> J;
  J jq[4], jt[3], jX[3], jc[NCAM > 0 ? NCAM : 1], jr[2];
  for (int i = 0; i < 4; ++i) jq[i] = J::Var(q[i], i);
  for (int i = 0; i < 3; ++i) jt[i] = J::Var(t[i], 4 + i);
  for (int i = 0; i < 3; ++i) jX[i] = J::Var(X[i], 7 + i);
  for (int i = 0; i < NCAM; ++i) jc[i] = J::Var(cam[i], 10 + i);
  LineResidual<J>(model, line, jq, jt, jX, jc, jr);
  for (int row = 0; row < 2; ++row) {
    r[row] = jr[row].a;
    if (Jq) for (int i = 0; i < 4; ++i) Jq[row * 4 + i] = jr[row].v[i];
    if (Jt) for (int i = 0; i < 3; ++i) Jt[row * 3 + i] = jr[row].v[4 + i];
    if (JX) for (int i = 0; i < 3; ++i) JX[row * 3 + i] = jr[row].v[7 + i];
    if (Jcam) for (int i = 0; i < NCAM; ++i) Jcam[row * NCAM + i] = jr[row].v[10 + i];
  }
}

// residual + ambient Jacobians (any Jacobian pointer may be null, as in ceres::CostFunction::Evaluate)
inline bool LineCostEvaluate(int model, const double line[3], const double q[4], const double t[3],
                             const double X[3], const double* cam, double r[2],
                             double* Jq, double* Jt, double* JX, double* Jcam) {
  switch (NumParams(model)) {
    case 3:  LineCostJetsN<3>(model, line, q, t, X, cam, r, Jq, Jt, JX, Jcam); return true;
    case 4:  LineCostJetsN<4>(model, line, q, t, X, cam, r, Jq, Jt, JX, Jcam); return true;
    case 5:  LineCostJetsN<5>(model, line, q, t, X, cam, r, Jq, Jt, JX, Jcam); return true;
    case 8:  LineCostJetsN<8>(model, line, q, t, X, cam, r, Jq, Jt, JX, Jcam); return true;
    case 12: LineCostJetsN<12>(model, line, q, t, X, cam, r, Jq, Jt, JX, Jcam); return true;
    default: return false;
  }
}

// plus-Jacobian of the quaternion local parameterisation, 4x3 row-major for q = (w,x,y,z).
// Third-party Ceres (QuaternionParameterization::ComputeJacobian), restated from its
// published definition Plus(q, d) = [cos|d|, sin|d| d/|d|] (x) q  =>  dPlus/dd at d = 0.
inline void QuaternionPlusJacobian(const double q[4], double J[12]) {
  J[0] = -q[1]; J[1] = -q[2]; J[2] = -q[3];
  J[3] = q[0];  J[4] = q[3];  J[5] = -q[2];
  J[6] = -q[3]; J[7] = q[0];  J[8] = q[1];
  J[9] = q[2];  J[10] = -q[1]; J[11] = q[0];
}

inline void QuaternionPlus(const double q[4], const double d[3], double out[4]) {
  const double n = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
  if (n > 0.0) {
    const double s = std::sin(n) / n;
    const double dq[4] = {std::cos(n), s * d[0], s * d[1], s * d[2]};
    // out = dq (x) q   (Hamilton product, w first)
    out[0] = dq[0] * q[0] - dq[1] * q[1] - dq[2] * q[2] - dq[3] * q[3];
    out[1] = dq[0] * q[1] + dq[1] * q[0] + dq[2] * q[3] - dq[3] * q[2];
    out[2] = dq[0] * q[2] - dq[1] * q[3] + dq[2] * q[0] + dq[3] * q[1];
    out[3] = dq[0] * q[3] + dq[1] * q[2] - dq[2] * q[1] + dq[3] * q[0];
  } else {
    for (int i = 0; i < 4; ++i) out[i] = q[i];
  }
}

}  // namespace oracle

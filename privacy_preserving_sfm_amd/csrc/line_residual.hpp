// Device-side line-to-point reprojection residual and its analytic Jacobian (kernel K1 body).
//
// Residual: reference src/base/cost_functions.h:62-100 (variable pose) / :139-178 (constant pose):
//   p = Rot(q) X + t ; (u,v) = p.xy / p.z ; alpha = a u + b v + c ; foot = (u,v) - alpha (a,b) ;
//   r = W(cam; u,v) - W(cam; foot)          with W = CameraModel::WorldToImage.
// The reference differentiates this with width-(10+N) Ceres jets.  Here the chain rule is written
// out so that only two width-2 duals (for DW at the two points) are needed:
//   d r/d(u,v) = DW(u,v) - DW(foot) (I - n n^T)
//   d(u,v)/dp  = 1/p.z [[1,0,-u],[0,1,-v]]
//   dp/dX = M(q) (the rotate-point polynomial's matrix), dp/dt = I,
//   dp/d(rotation tangent) = 2 [ . ] x (M X)   for Plus(q, d) = [cos|d|, sin|d| d/|d|] (x) q
//     (Ceres QuaternionParameterization; exact for unit q, which BundleAdjuster enforces by
//      Image::NormalizeQvec, bundle_adjustment.cc:355)
// jac_mode 1 additionally offers the ambient 2x4 d r/d q of the polynomial, as Ceres returns it.
#pragma once
#include "camera_models.hpp"

namespace ppsfm {

struct LineObsJac {
  double r[2];
  double Jt[6];    // 2x3  d r / d t      (also the "B" matrix)
  double JX[6];    // 2x3  d r / d X
  double Jrot[6];  // 2x3  d r / d rotation tangent
  double Jq[8];    // 2x4  d r / d q (ambient), filled only if AMBIENT
};

// rotation polynomial matrix M(q): p = M X, q not re-normalised (ceres::UnitQuaternionRotatePoint)
PP_HD void RotationPolynomialMatrix(const double q[4], double M[9]) {
  const double w = q[0], x = q[1], y = q[2], z = q[3];
  M[0] = 1.0 - 2.0 * (y * y + z * z); M[1] = 2.0 * (x * y - w * z);       M[2] = 2.0 * (x * z + w * y);
  M[3] = 2.0 * (x * y + w * z);       M[4] = 1.0 - 2.0 * (x * x + z * z); M[5] = 2.0 * (y * z - w * x);
  M[6] = 2.0 * (x * z - w * y);       M[7] = 2.0 * (y * z + w * x);       M[8] = 1.0 - 2.0 * (x * x + y * y);
}

// residual only
PP_HD void LineResidualOnly(int model, const double* cam, const double q[4], const double t[3], const double X[3],
                            double a, double b, double c, double r[2]) {
  double M[9];
  RotationPolynomialMatrix(q, M);
  const double px = M[0] * X[0] + M[1] * X[1] + M[2] * X[2] + t[0];
  const double py = M[3] * X[0] + M[4] * X[1] + M[5] * X[2] + t[1];
  const double pz = M[6] * X[0] + M[7] * X[1] + M[8] * X[2] + t[2];
  const double u = px / pz, v = py / pz;
  const double alpha = a * u + b * v + c;
  const double fu = u - alpha * a, fv = v - alpha * b;
  double x0, y0, x1, y1;
  WorldToImage<double, double>(model, cam, u, v, &x0, &y0);
  WorldToImage<double, double>(model, cam, fu, fv, &x1, &y1);
  r[0] = x0 - x1;
  r[1] = y0 - y1;
}

template <bool AMBIENT>
PP_HD void LineResidualJacobian(int model, const double* cam, const double q[4], const double t[3], const double X[3],
                                double a, double b, double c, LineObsJac* out) {
  double M[9];
  RotationPolynomialMatrix(q, M);
  const double rx = M[0] * X[0] + M[1] * X[1] + M[2] * X[2];
  const double ry = M[3] * X[0] + M[4] * X[1] + M[5] * X[2];
  const double rz = M[6] * X[0] + M[7] * X[1] + M[8] * X[2];
  const double px = rx + t[0], py = ry + t[1], pz = rz + t[2];
  const double iz = 1.0 / pz;
  const double u = px * iz, v = py * iz;
  const double alpha = a * u + b * v + c;
  const double fu = u - alpha * a, fv = v - alpha * b;

  typedef Dual<2> D2;
  D2 x0, y0, x1, y1;
  WorldToImage<D2, double>(model, cam, MakeVar<2>(u, 0), MakeVar<2>(v, 1), &x0, &y0);
  WorldToImage<D2, double>(model, cam, MakeVar<2>(fu, 0), MakeVar<2>(fv, 1), &x1, &y1);
  out->r[0] = x0.a - x1.a;
  out->r[1] = y0.a - y1.a;

  // A = DW0 - DW1 (I - n n^T)
  const double n00 = 1.0 - a * a, n01 = -a * b, n11 = 1.0 - b * b;
  const double A00 = x0.d[0] - (x1.d[0] * n00 + x1.d[1] * n01);
  const double A01 = x0.d[1] - (x1.d[0] * n01 + x1.d[1] * n11);
  const double A10 = y0.d[0] - (y1.d[0] * n00 + y1.d[1] * n01);
  const double A11 = y0.d[1] - (y1.d[0] * n01 + y1.d[1] * n11);
  // B = A * 1/pz [[1,0,-u],[0,1,-v]]
  double* B = out->Jt;
  B[0] = A00 * iz; B[1] = A01 * iz; B[2] = -(A00 * u + A01 * v) * iz;
  B[3] = A10 * iz; B[4] = A11 * iz; B[5] = -(A10 * u + A11 * v) * iz;
  // d r/dX = B M
#pragma unroll
  for (int row = 0; row < 2; ++row)
#pragma unroll
    for (int j = 0; j < 3; ++j)
      out->JX[3 * row + j] = B[3 * row] * M[j] + B[3 * row + 1] * M[3 + j] + B[3 * row + 2] * M[6 + j];
  // d r/d(rot tangent) = B * 2 [[0, rz, -ry], [-rz, 0, rx], [ry, -rx, 0]]
#pragma unroll
  for (int row = 0; row < 2; ++row) {
    const double b0 = B[3 * row], b1 = B[3 * row + 1], b2 = B[3 * row + 2];
    out->Jrot[3 * row + 0] = 2.0 * (b2 * ry - b1 * rz);
    out->Jrot[3 * row + 1] = 2.0 * (b0 * rz - b2 * rx);
    out->Jrot[3 * row + 2] = 2.0 * (b1 * rx - b0 * ry);
  }
  if (AMBIENT) {
    // d p / d q of p = X + 2 w (v x X) + 2 v x (v x X):
    //   d/dw = 2 (v x X) ;  d/dv[k] = 2 w (e_k x X) + 2 e_k x (v x X) + 2 v x (e_k x X)
    const double w = q[0], vx = q[1], vy = q[2], vz = q[3];
    const double cx = vy * X[2] - vz * X[1], cy = vz * X[0] - vx * X[2], cz = vx * X[1] - vy * X[0];  // v x X
    double dp[12];  // 3x4 row-major, columns (w,x,y,z)
    dp[0] = 2.0 * cx; dp[4] = 2.0 * cy; dp[8] = 2.0 * cz;
    // e_x: e x X = (0,-X2,X1); e x c = (0,-cz,cy); v x (e x X) = (vy*X1 + vz*X2, -vx*X1, -vx*X2)
    dp[1] = 2.0 * (vy * X[1] + vz * X[2]);
    dp[5] = 2.0 * (w * (-X[2]) - cz - vx * X[1]);
    dp[9] = 2.0 * (w * X[1] + cy - vx * X[2]);
    // e_y: e x X = (X2,0,-X0); e x c = (cz,0,-cx); v x (e x X) = (-vy*X0, vz*X2 + vx*X0, -vy*X2)
    dp[2] = 2.0 * (w * X[2] + cz - vy * X[0]);
    dp[6] = 2.0 * (vz * X[2] + vx * X[0]);
    dp[10] = 2.0 * (w * (-X[0]) - cx - vy * X[2]);
    // e_z: e x X = (-X1,X0,0); e x c = (-cy,cx,0); v x (e x X) = (-vz*X0, -vz*X1, vx*X0 + vy*X1)
    dp[3] = 2.0 * (w * (-X[1]) - cy - vz * X[0]);
    dp[7] = 2.0 * (w * X[0] + cx - vz * X[1]);
    dp[11] = 2.0 * (vx * X[0] + vy * X[1]);
#pragma unroll
    for (int row = 0; row < 2; ++row)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        out->Jq[4 * row + j] = B[3 * row] * dp[j] + B[3 * row + 1] * dp[4 + j] + B[3 * row + 2] * dp[8 + j];
  }
}

// d r / d intrinsics (2 x N): dW/dcam at (u,v) minus dW/dcam at the foot point.
template <int N>
PP_HD void LineResidualCameraJacobianN(int model, const double* cam, double u, double v, double fu, double fv, double* Jcam,
                                       int stride) {
  typedef Dual<N> DN;
  DN p[N];
#pragma unroll
  for (int i = 0; i < N; ++i) p[i] = MakeVar<N>(cam[i], i);
  DN x0, y0, x1, y1;
  WorldToImage<DN, DN>(model, p, MakeDual<N>(u), MakeDual<N>(v), &x0, &y0);
  WorldToImage<DN, DN>(model, p, MakeDual<N>(fu), MakeDual<N>(fv), &x1, &y1);
#pragma unroll
  for (int i = 0; i < N; ++i) { Jcam[i] = x0.d[i] - x1.d[i]; Jcam[stride + i] = y0.d[i] - y1.d[i]; }
}

// recompute (u,v,foot) and fill Jcam rows (row stride `stride` doubles)
PP_HD void LineResidualCameraJacobian(int model, const double* cam, const double q[4], const double t[3], const double X[3],
                                      double a, double b, double c, double* Jcam, int stride) {
  double M[9];
  RotationPolynomialMatrix(q, M);
  const double px = M[0] * X[0] + M[1] * X[1] + M[2] * X[2] + t[0];
  const double py = M[3] * X[0] + M[4] * X[1] + M[5] * X[2] + t[1];
  const double pz = M[6] * X[0] + M[7] * X[1] + M[8] * X[2] + t[2];
  const double u = px / pz, v = py / pz;
  const double alpha = a * u + b * v + c;
  const double fu = u - alpha * a, fv = v - alpha * b;
  switch (CameraNumParams(model)) {
    case 3: LineResidualCameraJacobianN<3>(model, cam, u, v, fu, fv, Jcam, stride); break;
    case 4: LineResidualCameraJacobianN<4>(model, cam, u, v, fu, fv, Jcam, stride); break;
    case 5: LineResidualCameraJacobianN<5>(model, cam, u, v, fu, fv, Jcam, stride); break;
    case 8: LineResidualCameraJacobianN<8>(model, cam, u, v, fu, fv, Jcam, stride); break;
    case 12: LineResidualCameraJacobianN<12>(model, cam, u, v, fu, fv, Jcam, stride); break;
    default: break;
  }
}

}  // namespace ppsfm

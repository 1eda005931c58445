// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// CPU restatement of the forward camera maps (normalised plane -> pixels) of the
// reference's 11 camera models, templated on the scalar so jets flow through.
// Follows reference src/base/camera_models.h:
//   ids / parameter counts        :189-349  (CAMERA_MODEL_DEFINITIONS)
//   SIMPLE_PINHOLE  WorldToImage  :614-626
//   PINHOLE                       :663-676
//   SIMPLE_RADIAL   + Distortion  :714-730, :746-757
//   RADIAL                        :783-799, :815-827
//   OPENCV                        :853-869, :887-902
//   OPENCV_FISHEYE                :928-944, :962-986
//   FULL_OPENCV                   :1023-1039, :1057-1079
//   FOV                           :1104-1119, :1136-1173
//   SIMPLE_RADIAL_FISHEYE         :1239-1255, :1271-1289
//   RADIAL_FISHEYE                :1315-1331, :1347-1367
//   THIN_PRISM_FISHEYE            :1405-1434, :1459-1481
//   ImageToWorldThreshold         :533-543
#pragma once
#include <limits>
#include "jet.h"

namespace oracle {

enum ModelId {
  kSimplePinhole = 0, kPinhole = 1, kSimpleRadial = 2, kRadial = 3, kOpenCV = 4,
  kOpenCVFisheye = 5, kFullOpenCV = 6, kFOV = 7, kSimpleRadialFisheye = 8,
  kRadialFisheye = 9, kThinPrismFisheye = 10, kNumModels = 11
};

inline int NumParams(int model) {
  static const int n[kNumModels] = {3, 4, 4, 5, 8, 8, 12, 5, 4, 5, 12};
  return (model >= 0 && model < kNumModels) ? n[model] : -1;
}
// number of focal-length parameters (1: f ; 2: fx, fy); they always lead the array
inline int NumFocal(int model) {
  static const int n[kNumModels] = {1, 2, 1, 1, 2, 2, 2, 2, 1, 1, 2};
  return n[model];
}
// mean focal length, used for pixel -> normalised threshold conversion (:533-543)
inline double ImageToWorldThreshold(int model, const double* p, double thr_px) {
  double f = 0.0;
  for (int i = 0; i < NumFocal(model); ++i) f += p[i];
  f /= NumFocal(model);
  return thr_px / f;
}

// ---- equidistant-fisheye radial helper shared by ids 5, 8, 9 ------------------
// du = u*thetad/r - u with thetad = theta*(1 + k1 th^2 + k2 th^4 + k3 th^6 + k4 th^8)
template <typename T>
inline void FisheyeRadial(const T& k1, const T& k2, const T& k3, const T& k4, int order,
                          const T& u, const T& v, T* du, T* dv) {
  const T r = sqrt(u * u + v * v);
  if (r > T(std::numeric_limits<double>::epsilon())) {
    const T theta = atan(r);
    const T theta2 = theta * theta;
    T series = T(1) + k1 * theta2;
    if (order >= 2) { const T theta4 = theta2 * theta2; series = series + k2 * theta4;
      if (order >= 4) { const T theta6 = theta4 * theta2; const T theta8 = theta4 * theta4;
        series = series + k3 * theta6 + k4 * theta8; } }
    const T thetad = theta * series;
    *du = u * thetad / r - u;
    *dv = v * thetad / r - v;
  } else {
    *du = T(0);
    *dv = T(0);
  }
}

// (u, v) on the normalised plane -> (x, y) in pixels.
template <typename T>
inline void WorldToImage(int model, const T* p, const T& u, const T& v, T* x, T* y) {
  switch (model) {
    case kSimplePinhole: {
      *x = p[0] * u + p[1];
      *y = p[0] * v + p[2];
      return;
    }
    case kPinhole: {
      *x = p[0] * u + p[2];
      *y = p[1] * v + p[3];
      return;
    }
    case kSimpleRadial: {
      const T u2 = u * u, v2 = v * v, r2 = u2 + v2;
      const T radial = p[3] * r2;
      const T xd = u + u * radial, yd = v + v * radial;
      *x = p[0] * xd + p[1];
      *y = p[0] * yd + p[2];
      return;
    }
    case kRadial: {
      const T u2 = u * u, v2 = v * v, r2 = u2 + v2;
      const T radial = p[3] * r2 + p[4] * r2 * r2;
      const T xd = u + u * radial, yd = v + v * radial;
      *x = p[0] * xd + p[1];
      *y = p[0] * yd + p[2];
      return;
    }
    case kOpenCV: {
      const T u2 = u * u, uv = u * v, v2 = v * v, r2 = u2 + v2;
      const T radial = p[4] * r2 + p[5] * r2 * r2;
      const T du = u * radial + T(2) * p[6] * uv + p[7] * (r2 + T(2) * u2);
      const T dv = v * radial + T(2) * p[7] * uv + p[6] * (r2 + T(2) * v2);
      *x = p[0] * (u + du) + p[2];
      *y = p[1] * (v + dv) + p[3];
      return;
    }
    case kOpenCVFisheye: {
      T du, dv;
      FisheyeRadial(p[4], p[5], p[6], p[7], 4, u, v, &du, &dv);
      *x = p[0] * (u + du) + p[2];
      *y = p[1] * (v + dv) + p[3];
      return;
    }
    case kFullOpenCV: {
      const T u2 = u * u, uv = u * v, v2 = v * v, r2 = u2 + v2;
      const T r4 = r2 * r2, r6 = r4 * r2;
      const T radial = (T(1) + p[4] * r2 + p[5] * r4 + p[8] * r6) /
                       (T(1) + p[9] * r2 + p[10] * r4 + p[11] * r6);
      const T du = u * radial + T(2) * p[6] * uv + p[7] * (r2 + T(2) * u2) - u;
      const T dv = v * radial + T(2) * p[7] * uv + p[6] * (r2 + T(2) * v2) - v;
      *x = p[0] * (u + du) + p[2];
      *y = p[1] * (v + dv) + p[3];
      return;
    }
    case kFOV: {
      const T omega = p[4];
      const T kEps = T(1e-4);
      const T radius2 = u * u + v * v;
      const T omega2 = omega * omega;
      T factor;
      if (omega2 < kEps) {
        factor = (omega2 * radius2) / T(3) - omega2 / T(12) + T(1);
      } else if (radius2 < kEps) {
        const T th = tan(omega / T(2));
        factor = (T(-2) * th * (T(4) * radius2 * th * th - T(3))) / (T(3) * omega);
      } else {
        const T radius = sqrt(radius2);
        const T numerator = atan(radius * T(2) * tan(omega / T(2)));
        factor = numerator / (radius * omega);
      }
      *x = p[0] * (u * factor) + p[2];
      *y = p[1] * (v * factor) + p[3];
      return;
    }
    case kSimpleRadialFisheye: {
      T du, dv;
      FisheyeRadial(p[3], T(0), T(0), T(0), 1, u, v, &du, &dv);
      *x = p[0] * (u + du) + p[1];
      *y = p[0] * (v + dv) + p[2];
      return;
    }
    case kRadialFisheye: {
      T du, dv;
      FisheyeRadial(p[3], p[4], T(0), T(0), 2, u, v, &du, &dv);
      *x = p[0] * (u + du) + p[1];
      *y = p[0] * (v + dv) + p[2];
      return;
    }
    case kThinPrismFisheye: {
      const T r = sqrt(u * u + v * v);
      T uu, vv;
      if (r > T(std::numeric_limits<double>::epsilon())) {
        const T theta = atan(r);
        uu = theta * u / r;
        vv = theta * v / r;
      } else {
        uu = u;
        vv = v;
      }
      const T u2 = uu * uu, uv = uu * vv, v2 = vv * vv, r2 = u2 + v2;
      const T r4 = r2 * r2, r6 = r4 * r2, r8 = r6 * r2;
      const T radial = p[4] * r2 + p[5] * r4 + p[8] * r6 + p[9] * r8;
      const T du = uu * radial + T(2) * p[6] * uv + p[7] * (r2 + T(2) * u2) + p[10] * r2;
      const T dv = vv * radial + T(2) * p[7] * uv + p[6] * (r2 + T(2) * v2) + p[11] * r2;
      *x = p[0] * (uu + du) + p[2];
      *y = p[1] * (vv + dv) + p[3];
      return;
    }
    default:
      *x = T(std::numeric_limits<double>::quiet_NaN());
      *y = *x;
  }
}

}  // namespace oracle

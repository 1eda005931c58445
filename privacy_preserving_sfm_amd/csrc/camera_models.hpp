// Device-side forward camera maps of the reference's 11 camera models
// (reference src/base/camera_models.h: WorldToImage/Distortion per model, ids :189-349).
//
// MI355X design: the maps are written once, templated on the scalar, and instantiated with
//   T = double            residual-only evaluation (cost at a trial point)
//   T = Dual<2>           value + 2x2 derivative w.r.t. the normalised coordinates (u,v): the chain
//                         rule of the line residual only ever needs DW at two points, so the pose /
//                         point Jacobians cost two width-2 duals instead of the reference's
//                         width-(10+N) Ceres jets
//   T = Dual<2+N>         additionally d/d(intrinsics), only when intrinsics are refined
// The model id is wave-uniform in practice (observations are grouped by image), so the switch
// does not diverge.
#pragma once
#include <hip/hip_runtime.h>

namespace ppsfm {

template <int N>
struct Dual {
  double a;
  double d[N];
};

#define PP_HD __host__ __device__ __forceinline__

template <int N> PP_HD Dual<N> MakeDual(double s) { Dual<N> r; r.a = s; for (int k = 0; k < N; ++k) r.d[k] = 0.0; return r; }
template <int N> PP_HD Dual<N> MakeVar(double s, int idx) { Dual<N> r = MakeDual<N>(s); r.d[idx] = 1.0; return r; }

template <int N> PP_HD Dual<N> operator+(const Dual<N>& x, const Dual<N>& y) { Dual<N> r; r.a = x.a + y.a; for (int k = 0; k < N; ++k) r.d[k] = x.d[k] + y.d[k]; return r; }
template <int N> PP_HD Dual<N> operator-(const Dual<N>& x, const Dual<N>& y) { Dual<N> r; r.a = x.a - y.a; for (int k = 0; k < N; ++k) r.d[k] = x.d[k] - y.d[k]; return r; }
template <int N> PP_HD Dual<N> operator*(const Dual<N>& x, const Dual<N>& y) { Dual<N> r; r.a = x.a * y.a; for (int k = 0; k < N; ++k) r.d[k] = x.a * y.d[k] + x.d[k] * y.a; return r; }
template <int N> PP_HD Dual<N> operator/(const Dual<N>& x, const Dual<N>& y) {
  Dual<N> r; const double inv = 1.0 / y.a; r.a = x.a * inv;
  for (int k = 0; k < N; ++k) r.d[k] = (x.d[k] - r.a * y.d[k]) * inv;
  return r;
}
template <int N> PP_HD Dual<N> operator+(const Dual<N>& x, double s) { Dual<N> r = x; r.a += s; return r; }
template <int N> PP_HD Dual<N> operator+(double s, const Dual<N>& x) { Dual<N> r = x; r.a += s; return r; }
template <int N> PP_HD Dual<N> operator-(const Dual<N>& x, double s) { Dual<N> r = x; r.a -= s; return r; }
template <int N> PP_HD Dual<N> operator-(double s, const Dual<N>& x) { Dual<N> r; r.a = s - x.a; for (int k = 0; k < N; ++k) r.d[k] = -x.d[k]; return r; }
template <int N> PP_HD Dual<N> operator*(const Dual<N>& x, double s) { Dual<N> r; r.a = x.a * s; for (int k = 0; k < N; ++k) r.d[k] = x.d[k] * s; return r; }
template <int N> PP_HD Dual<N> operator*(double s, const Dual<N>& x) { return x * s; }
template <int N> PP_HD Dual<N> operator/(const Dual<N>& x, double s) { return x * (1.0 / s); }
template <int N> PP_HD Dual<N> operator/(double s, const Dual<N>& y) {
  Dual<N> r; const double inv = 1.0 / y.a; r.a = s * inv;
  for (int k = 0; k < N; ++k) r.d[k] = -r.a * y.d[k] * inv;
  return r;
}

PP_HD double Val(double x) { return x; }
template <int N> PP_HD double Val(const Dual<N>& x) { return x.a; }

PP_HD double Sqrt(double x) { return sqrt(x); }
PP_HD double Atan(double x) { return atan(x); }
PP_HD double Tan(double x) { return tan(x); }
template <int N> PP_HD Dual<N> Sqrt(const Dual<N>& x) { Dual<N> r; r.a = sqrt(x.a); const double g = 0.5 / r.a; for (int k = 0; k < N; ++k) r.d[k] = g * x.d[k]; return r; }
template <int N> PP_HD Dual<N> Atan(const Dual<N>& x) { Dual<N> r; r.a = atan(x.a); const double g = 1.0 / (1.0 + x.a * x.a); for (int k = 0; k < N; ++k) r.d[k] = g * x.d[k]; return r; }
template <int N> PP_HD Dual<N> Tan(const Dual<N>& x) { Dual<N> r; r.a = tan(x.a); const double g = 1.0 + r.a * r.a; for (int k = 0; k < N; ++k) r.d[k] = g * x.d[k]; return r; }

enum CameraModelId {
  kSimplePinhole = 0, kPinhole = 1, kSimpleRadial = 2, kRadial = 3, kOpenCV = 4, kOpenCVFisheye = 5,
  kFullOpenCV = 6, kFOV = 7, kSimpleRadialFisheye = 8, kRadialFisheye = 9, kThinPrismFisheye = 10
};

PP_HD int CameraNumParams(int model) {
  switch (model) {
    case kSimplePinhole: return 3;
    case kPinhole: case kSimpleRadial: case kSimpleRadialFisheye: return 4;
    case kRadial: case kFOV: case kRadialFisheye: return 5;
    case kOpenCV: case kOpenCVFisheye: return 8;
    case kFullOpenCV: case kThinPrismFisheye: return 12;
    default: return -1;
  }
}
PP_HD int CameraNumFocal(int model) {
  return (model == kSimplePinhole || model == kSimpleRadial || model == kRadial || model == kSimpleRadialFisheye ||
          model == kRadialFisheye) ? 1 : 2;
}

constexpr double kDblEps = 2.220446049250313e-16;

// Equidistant fisheye radial term shared by OPENCV_FISHEYE / SIMPLE_RADIAL_FISHEYE / RADIAL_FISHEYE
// (camera_models.h:962-986, :1271-1289, :1347-1367): returns the distorted point (u+du, v+dv).
template <typename T, typename P, int ORDER>
PP_HD void FisheyeDistorted(const P* k, const T& u, const T& v, T* xd, T* yd) {
  const T r = Sqrt(u * u + v * v);
  if (Val(r) > kDblEps) {
    const T theta = Atan(r);
    const T t2 = theta * theta;
    T series = 1.0 + k[0] * t2;
    if (ORDER >= 2) {
      const T t4 = t2 * t2;
      series = series + k[1] * t4;
      if (ORDER >= 4) series = series + k[2] * (t4 * t2) + k[3] * (t4 * t4);
    }
    const T thetad = theta * series;
    // u + (u*thetad/r - u): keep the reference's two-step form so values match to rounding
    *xd = u + (u * thetad / r - u);
    *yd = v + (v * thetad / r - v);
  } else {
    *xd = u;
    *yd = v;
  }
}

// (u,v) -> pixel (x,y).  P is the intrinsics scalar type (double, or Dual when refining intrinsics).
template <typename T, typename P>
PP_HD void WorldToImage(int model, const P* p, const T& u, const T& v, T* x, T* y) {
  switch (model) {
    case kSimplePinhole:
      *x = p[0] * u + p[1];
      *y = p[0] * v + p[2];
      return;
    case kPinhole:
      *x = p[0] * u + p[2];
      *y = p[1] * v + p[3];
      return;
    case kSimpleRadial: {
      const T r2 = u * u + v * v;
      const T radial = p[3] * r2;
      *x = p[0] * (u + u * radial) + p[1];
      *y = p[0] * (v + v * radial) + p[2];
      return;
    }
    case kRadial: {
      const T r2 = u * u + v * v;
      const T radial = p[3] * r2 + p[4] * r2 * r2;
      *x = p[0] * (u + u * radial) + p[1];
      *y = p[0] * (v + v * radial) + p[2];
      return;
    }
    case kOpenCV: {
      const T u2 = u * u, uv = u * v, v2 = v * v, r2 = u2 + v2;
      const T radial = p[4] * r2 + p[5] * r2 * r2;
      const T du = u * radial + 2.0 * p[6] * uv + p[7] * (r2 + 2.0 * u2);
      const T dv = v * radial + 2.0 * p[7] * uv + p[6] * (r2 + 2.0 * v2);
      *x = p[0] * (u + du) + p[2];
      *y = p[1] * (v + dv) + p[3];
      return;
    }
    case kOpenCVFisheye: {
      T xd, yd;
      FisheyeDistorted<T, P, 4>(p + 4, u, v, &xd, &yd);
      *x = p[0] * xd + p[2];
      *y = p[1] * yd + p[3];
      return;
    }
    case kFullOpenCV: {
      const T u2 = u * u, uv = u * v, v2 = v * v, r2 = u2 + v2;
      const T r4 = r2 * r2, r6 = r4 * r2;
      const T radial = (1.0 + p[4] * r2 + p[5] * r4 + p[8] * r6) / (1.0 + p[9] * r2 + p[10] * r4 + p[11] * r6);
      const T du = u * radial + 2.0 * p[6] * uv + p[7] * (r2 + 2.0 * u2) - u;
      const T dv = v * radial + 2.0 * p[7] * uv + p[6] * (r2 + 2.0 * v2) - v;
      *x = p[0] * (u + du) + p[2];
      *y = p[1] * (v + dv) + p[3];
      return;
    }
    case kFOV: {
      const P omega = p[4];
      const T radius2 = u * u + v * v;
      const P omega2 = omega * omega;
      T factor;
      if (Val(omega2) < 1e-4) {
        factor = (omega2 * radius2) / 3.0 - omega2 / 12.0 + 1.0;
      } else if (Val(radius2) < 1e-4) {
        const P th = Tan(omega / 2.0);
        factor = (-2.0 * th * (4.0 * radius2 * th * th - 3.0)) / (3.0 * omega);
      } else {
        const T radius = Sqrt(radius2);
        const T numerator = Atan(radius * 2.0 * Tan(omega / 2.0));
        factor = numerator / (radius * omega);
      }
      *x = p[0] * (u * factor) + p[2];
      *y = p[1] * (v * factor) + p[3];
      return;
    }
    case kSimpleRadialFisheye: {
      T xd, yd;
      FisheyeDistorted<T, P, 1>(p + 3, u, v, &xd, &yd);
      *x = p[0] * xd + p[1];
      *y = p[0] * yd + p[2];
      return;
    }
    case kRadialFisheye: {
      T xd, yd;
      FisheyeDistorted<T, P, 2>(p + 3, u, v, &xd, &yd);
      *x = p[0] * xd + p[1];
      *y = p[0] * yd + p[2];
      return;
    }
    case kThinPrismFisheye: {
      const T r = Sqrt(u * u + v * v);
      T uu, vv;
      if (Val(r) > kDblEps) {
        const T theta = Atan(r);
        uu = theta * u / r;
        vv = theta * v / r;
      } else {
        uu = u;
        vv = v;
      }
      const T u2 = uu * uu, uv = uu * vv, v2 = vv * vv, r2 = u2 + v2;
      const T r4 = r2 * r2, r6 = r4 * r2, r8 = r6 * r2;
      const T radial = p[4] * r2 + p[5] * r4 + p[8] * r6 + p[9] * r8;
      const T du = uu * radial + 2.0 * p[6] * uv + p[7] * (r2 + 2.0 * u2) + p[10] * r2;
      const T dv = vv * radial + 2.0 * p[7] * uv + p[6] * (r2 + 2.0 * v2) + p[11] * r2;
      *x = p[0] * (uu + du) + p[2];
      *y = p[1] * (vv + dv) + p[3];
      return;
    }
    default:
      *x = u * __builtin_nan("");
      *y = *x;
  }
}

}  // namespace ppsfm

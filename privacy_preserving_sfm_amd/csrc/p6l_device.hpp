// K5 body — six-line absolute pose minimal solver, one lane per hypothesis.
//
//   P6LEstimator::Estimate   reference src/estimators/absolute_pose.cc:79-162
//   rotation_to_e3q3 / cayley_param                                   :46-75
//   re3q3                    reference lib/re3q3/re3q3/re3q3.h:16-200
//
// MI355X design notes
//   * everything is fixed-size and fully unrolled so it lives in VGPRs (no scratch): the 3x3
//     pivoted solves, the polynomial matrix M(x) of degrees [[2,2,3],[2,2,3],[3,3,4]] built with
//     small polynomial products (the reference writes its entries out as ~300 products), and its
//     degree-8 determinant.
//   * the reference finds the roots with Eigen::EigenSolver on the 8x8 companion matrix
//     (re3q3.h:152-165) — a data-dependent Hessenberg-QR iteration with dynamic indexing that maps
//     badly onto SIMT lanes.  Here all 8 complex roots are found by the Aberth-Ehrlich simultaneous
//     iteration (cubic convergence, identical control flow in every lane), the same |Im| <= 1e-8
//     acceptance is applied (re3q3.h:169-171), real roots get two Newton polish steps and are
//     returned in ASCENDING order (the reference's order is Eigen's, which is unspecified).
//   * the reference's degenerate branches draw rand() matrices (absolute_pose.cc:128-134,
//     re3q3.h:39-43); here they use FIXED matrices from a splitmix64 stream (seeds 2 and 1) unless
//     the caller injects them, so results are a pure function of the input.
#pragma once
#include "camera_models.hpp"  // PP_HD

namespace ppsfm {

struct DegenerateStream {
  unsigned long long s;
  PP_HD explicit DegenerateStream(unsigned long long seed) : s(seed) {}
  PP_HD double Next() {
    unsigned long long z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return (double)(z >> 11) * (2.0 / 9007199254740992.0) - 1.0;
  }
};

PP_HD double Det3x3(const double A[9]) {
  return A[0] * (A[4] * A[8] - A[5] * A[7]) - A[1] * (A[3] * A[8] - A[5] * A[6]) + A[2] * (A[3] * A[7] - A[4] * A[6]);
}

// A (3x3 row-major) X = B (3 x NR row-major), partial pivoting, fully unrolled with select-swaps
template <int NR>
PP_HD bool Solve3(double A[9], double B[3 * NR]) {
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    // pivot search among rows k..2
    int piv = k;
    double best = fabs(A[3 * k + k]);
#pragma unroll
    for (int i = k + 1; i < 3; ++i) {
      const double v = fabs(A[3 * i + k]);
      if (v > best) { best = v; piv = i; }
    }
    if (best == 0.0) return false;
#pragma unroll
    for (int i = k + 1; i < 3; ++i) {
      if (piv == i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) { const double tmp = A[3 * k + j]; A[3 * k + j] = A[3 * i + j]; A[3 * i + j] = tmp; }
#pragma unroll
        for (int j = 0; j < NR; ++j) { const double tmp = B[NR * k + j]; B[NR * k + j] = B[NR * i + j]; B[NR * i + j] = tmp; }
      }
    }
    const double inv = 1.0 / A[3 * k + k];
#pragma unroll
    for (int i = k + 1; i < 3; ++i) {
      const double f = A[3 * i + k] * inv;
#pragma unroll
      for (int j = k + 1; j < 3; ++j) A[3 * i + j] -= f * A[3 * k + j];
#pragma unroll
      for (int j = 0; j < NR; ++j) B[NR * i + j] -= f * B[NR * k + j];
    }
  }
#pragma unroll
  for (int j = 0; j < NR; ++j) {
    const double x2 = B[NR * 2 + j] / A[8];
    const double x1 = (B[NR * 1 + j] - A[5] * x2) / A[4];
    const double x0 = (B[j] - A[1] * x1 - A[2] * x2) / A[0];
    B[j] = x0; B[NR + j] = x1; B[2 * NR + j] = x2;
  }
  return true;
}

// out (deg DA+DB) += sign * a (deg DA) * b (deg DB); ascending coefficients
template <int DA, int DB>
PP_HD void PolyMac(const double* a, const double* b, double sign, double* out) {
#pragma unroll
  for (int i = 0; i <= DA; ++i)
#pragma unroll
    for (int j = 0; j <= DB; ++j) out[i + j] += sign * (a[i] * b[j]);
}

template <int D>
PP_HD double PolyVal(const double* p, double x) {
  double v = p[D];
#pragma unroll
  for (int e = D - 1; e >= 0; --e) v = v * x + p[e];
  return v;
}

// 1 / x inside the Aberth iteration: the iteration is self-correcting and its roots are polished by Newton steps in full
// precision afterwards, so the ~70 divisions per sweep use v_rcp_f64 + two Newton steps (5 instructions, ~1 ulp) instead of the
// IEEE division sequence (12); on the host the plain division.
PP_HD double AberthRcp(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  double r = __builtin_amdgcn_rcp(x);
  r = __builtin_fma(r, __builtin_fma(-x, r, 1.0), r);
  return __builtin_fma(r, __builtin_fma(-x, r, 1.0), r);
#else
  return 1.0 / x;
#endif
}

// All complex roots of the monic octic x^8 + a[7] x^7 + ... + a[0] by Aberth-Ehrlich.
#if defined(__HIPCC__)
// Sweep cap.  Measured on 16384 six-tuples (50 % outliers): with 40 sweeps every hypothesis returns the same number of models
// as with 80 and only 3 differ by more than 1e-9 relative — the same 3 that also differ between 60 and 80 sweeps, i.e. lanes
// that never converge (ill-conditioned octics) and whose result depends on the cap whatever it is.  A wavefront runs as long
// as its slowest lane, so those lanes set the solver's time: 80 -> 40 sweeps halves it.  PPSFM_ABERTH_SWEEPS (read at
// pp_pose_create) overrides the cap for experiments.
__device__ int g_aberth_sweeps = 40;
#endif
PP_HD void AberthOctic(const double a[8], double zr[8], double zi[8]) {
  // initial circle: 0.7 * max_k |a_k|^(1/(8-k)) (half the Fujiwara bound); single precision is plenty
  float rad = 0.0f;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const float m = fabsf((float)a[k]);
    const float e = (m > 0.0f) ? exp2f(log2f(m) * (1.0f / (8 - k))) : 0.0f;
    rad = fmaxf(rad, e);
  }
  const double radius = (rad > 1e-30f && rad < 1e30f) ? 0.7 * (double)rad : 1.0;
  // cos/sin(pi/4 * k + 0.35): the offset breaks the conjugate symmetry of the start configuration
  const double cs[8] = {0.9393727128473789, 0.42177145041023634, -0.3428978074554514, -0.9067021802217339,
                        -0.9393727128473789, -0.42177145041023684, 0.34289780745545084, 0.9067021802217337};
  const double sn[8] = {0.34289780745545134, 0.9067021802217339, 0.9393727128473789, 0.4217714504102364,
                        -0.34289780745545134, -0.9067021802217337, -0.9393727128473791, -0.4217714504102369};
#pragma unroll
  for (int k = 0; k < 8; ++k) { zr[k] = radius * cs[k]; zi[k] = radius * sn[k]; }
#if defined(__HIP_DEVICE_COMPILE__)
  const int max_sweeps = g_aberth_sweeps;
#else
  const int max_sweeps = 40;
#endif
#pragma unroll 1
  for (int it = 0; it < max_sweeps; ++it) {
    double worst = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const double xr = zr[k], xi = zi[k];
      // Horner for p and p'
      double pr = 1.0, pi = 0.0, dr = 0.0, di = 0.0;
#pragma unroll
      for (int e = 7; e >= 0; --e) {
        const double ndr = dr * xr - di * xi + pr, ndi = dr * xi + di * xr + pi;
        dr = ndr; di = ndi;
        const double npr = pr * xr - pi * xi + a[e], npi = pr * xi + pi * xr;
        pr = npr; pi = npi;
      }
      // w = p / p'
      const double dn = dr * dr + di * di;
      const double idn = (dn > 0.0) ? AberthRcp(dn) : 0.0;
      const double wr = (pr * dr + pi * di) * idn, wi = (pi * dr - pr * di) * idn;
      // s = sum_{j != k} 1 / (z_k - z_j)
      double sr = 0.0, si = 0.0;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (j != k) {
          const double er = xr - zr[j], ei = xi - zi[j];
          const double en = er * er + ei * ei;
          const double ien = (en > 0.0) ? AberthRcp(en) : 0.0;
          sr += er * ien; si -= ei * ien;
        }
      }
      // delta = w / (1 - w s)
      const double qr = 1.0 - (wr * sr - wi * si), qi = -(wr * si + wi * sr);
      const double qn = qr * qr + qi * qi;
      const double iqn = (qn > 0.0) ? AberthRcp(qn) : 0.0;
      const double cr = (qn > 0.0) ? (wr * qr + wi * qi) * iqn : wr;
      const double ci = (qn > 0.0) ? (wi * qr - wr * qi) * iqn : wi;
      zr[k] = xr - cr;
      zi[k] = xi - ci;
      const double step = fabs(cr) + fabs(ci), mag = fabs(xr) + fabs(xi);
      worst = fmax(worst, step - 1e-13 * mag);     // (2e-15 sat at the rounding level of the sweep itself: some lane of every
                                                   // wavefront never met it and all 80 sweeps ran — 151k VALU instructions per
                                                   // wavefront by PMC; the accepted roots are polished by Newton steps below)
    }
    if (!(worst > 0.0)) break;
  }
}

// coeffs 3x10 row-major (x^2, xy, xz, y^2, yz, z^2, x, y, z, 1); sol 3x8 row-major.
// affine: optional injected 3x4 change of variables for the all-degenerate case.
PP_HD int Re3q3Device(const double* cin, double* sol, bool allow_var_change, const double* affine) {
  double c[30];
#pragma unroll
  for (int i = 0; i < 30; ++i) c[i] = cin[i];

  // elimination variable: largest |det| of the quadratic block of the two kept unknowns (re3q3.h:17-37)
  double dets[3];
  {
    const double Ax[9] = {c[3], c[5], c[4], c[13], c[15], c[14], c[23], c[25], c[24]};
    const double Ay[9] = {c[0], c[5], c[2], c[10], c[15], c[12], c[20], c[25], c[22]};
    const double Az[9] = {c[3], c[0], c[1], c[13], c[10], c[11], c[23], c[20], c[21]};
    dets[0] = fabs(Det3x3(Ax)); dets[1] = fabs(Det3x3(Ay)); dets[2] = fabs(Det3x3(Az));
  }
  int elim = 0;
  double det = dets[0];
  if (det < dets[1]) { det = dets[1]; elim = 1; }
  if (det < dets[2]) { det = dets[2]; elim = 2; }

  double A4[12];
  bool changed = false;
  if (allow_var_change && det < 1e-10) {
    changed = true;
    if (affine) {
#pragma unroll
      for (int i = 0; i < 12; ++i) A4[i] = affine[i];
    } else {
      DegenerateStream rng(1);
      double q[4], n = 0;
      for (int i = 0; i < 4; ++i) { q[i] = rng.Next(); n += q[i] * q[i]; }
      n = sqrt(n);
      for (int i = 0; i < 4; ++i) q[i] /= n;
      const double w = q[0], x = q[1], y = q[2], z = q[3];
      const double R[9] = {1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                           2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                           2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)};
      double tv[3], tn = 0;
      for (int i = 0; i < 3; ++i) { tv[i] = rng.Next(); tn += tv[i] * tv[i]; }
      tn = sqrt(tn);
      for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) A4[4 * i + j] = R[3 * i + j]; A4[4 * i + 3] = tv[i] / tn; }
    }
    // old monomials in terms of the new unknowns: v_k = A4[k][0..2].(x',y',z') + A4[k][3]
    double c2[30];
#pragma unroll
    for (int i = 0; i < 30; ++i) c2[i] = 0.0;
    const int mr[6] = {0, 0, 0, 1, 1, 2}, ms[6] = {0, 1, 2, 1, 2, 2};
#pragma unroll
    for (int m = 0; m < 6; ++m) {
      const double* a = A4 + 4 * mr[m];
      const double* b = A4 + 4 * ms[m];
      const double e[10] = {a[0] * b[0], a[0] * b[1] + a[1] * b[0], a[0] * b[2] + a[2] * b[0], a[1] * b[1],
                            a[1] * b[2] + a[2] * b[1], a[2] * b[2], a[0] * b[3] + a[3] * b[0], a[1] * b[3] + a[3] * b[1],
                            a[2] * b[3] + a[3] * b[2], a[3] * b[3]};
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 10; ++j) c2[10 * i + j] += c[10 * i + m] * e[j];
    }
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        c2[10 * i + 6] += c[10 * i + 6 + k] * A4[4 * k];
        c2[10 * i + 7] += c[10 * i + 6 + k] * A4[4 * k + 1];
        c2[10 * i + 8] += c[10 * i + 6 + k] * A4[4 * k + 2];
        c2[10 * i + 9] += c[10 * i + 6 + k] * A4[4 * k + 3];
      }
#pragma unroll
    for (int i = 0; i < 3; ++i) c2[10 * i + 9] += c[10 * i + 9];
#pragma unroll
    for (int i = 0; i < 30; ++i) c[i] = c2[i];
    // re-pick the elimination variable on the transformed system (no second change of variables)
    const double Ax[9] = {c[3], c[5], c[4], c[13], c[15], c[14], c[23], c[25], c[24]};
    const double Ay[9] = {c[0], c[5], c[2], c[10], c[15], c[12], c[20], c[25], c[22]};
    const double Az[9] = {c[3], c[0], c[1], c[13], c[10], c[11], c[23], c[20], c[21]};
    dets[0] = fabs(Det3x3(Ax)); dets[1] = fabs(Det3x3(Ay)); dets[2] = fabs(Det3x3(Az));
    elim = 0; det = dets[0];
    if (det < dets[1]) { det = dets[1]; elim = 1; }
    if (det < dets[2]) { det = dets[2]; elim = 2; }
  }

  // rename so the eliminated unknown is X and the kept ones Y, Z; columns [Y^2 Z^2 YZ | X^2 XY XZ X Y Z 1]
  double A[9], B[21];
  {
    const int perm[3][10] = {{3, 5, 4, 0, 1, 2, 6, 7, 8, 9}, {0, 5, 2, 3, 1, 4, 7, 6, 8, 9}, {3, 0, 1, 5, 4, 2, 8, 7, 6, 9}};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
      for (int j = 0; j < 3; ++j) A[3 * i + j] = elim == 0 ? c[10 * i + perm[0][j]] : (elim == 1 ? c[10 * i + perm[1][j]] : c[10 * i + perm[2][j]]);
#pragma unroll
      for (int j = 0; j < 7; ++j) B[7 * i + j] = elim == 0 ? c[10 * i + perm[0][3 + j]] : (elim == 1 ? c[10 * i + perm[1][3 + j]] : c[10 * i + perm[2][3 + j]]);
    }
  }
  if (!Solve3<7>(A, B)) return 0;
  // Q_r = a_r(X) Y + b_r(X) Z + c_r(X), r = 0: Y^2, 1: Z^2, 2: YZ ; P = -A^-1 B over [X^2 XY XZ X Y Z 1]
  double pa[3][2], pb[3][2], pc[3][3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    pa[r][0] = -B[7 * r + 4]; pa[r][1] = -B[7 * r + 1];
    pb[r][0] = -B[7 * r + 5]; pb[r][1] = -B[7 * r + 2];
    pc[r][0] = -B[7 * r + 6]; pc[r][1] = -B[7 * r + 3]; pc[r][2] = -B[7 * r + 0];
  }
  // polynomial matrix: rows = identities Y*(YZ) = Z*(Y^2), Z*(YZ) = Y*(Z^2), (YZ)^2 = (Y^2)(Z^2);
  // columns = coefficients of Y, Z, 1
  double M00[3] = {0, 0, 0}, M01[3] = {0, 0, 0}, M02[4] = {0, 0, 0, 0};
  double M10[3] = {0, 0, 0}, M11[3] = {0, 0, 0}, M12[4] = {0, 0, 0, 0};
  double M20[4] = {0, 0, 0, 0}, M21[4] = {0, 0, 0, 0}, M22[5] = {0, 0, 0, 0, 0};
  {
    const int Y = 0, Z = 1, W = 2;
    double lin[2];
    // row 0: aW*Q_Y + (bW - aY)*Q_W - bY*Q_Z + cW*Y - cY*Z
    PolyMac<1, 1>(pa[W], pa[Y], 1.0, M00); PolyMac<1, 1>(pa[W], pb[Y], 1.0, M01); PolyMac<1, 2>(pa[W], pc[Y], 1.0, M02);
    lin[0] = pb[W][0] - pa[Y][0]; lin[1] = pb[W][1] - pa[Y][1];
    PolyMac<1, 1>(lin, pa[W], 1.0, M00); PolyMac<1, 1>(lin, pb[W], 1.0, M01); PolyMac<1, 2>(lin, pc[W], 1.0, M02);
    PolyMac<1, 1>(pb[Y], pa[Z], -1.0, M00); PolyMac<1, 1>(pb[Y], pb[Z], -1.0, M01); PolyMac<1, 2>(pb[Y], pc[Z], -1.0, M02);
#pragma unroll
    for (int k = 0; k < 3; ++k) { M00[k] += pc[W][k]; M01[k] -= pc[Y][k]; }
    // row 1: bW*Q_Z + (aW - bZ)*Q_W - aZ*Q_Y + cW*Z - cZ*Y
    PolyMac<1, 1>(pb[W], pa[Z], 1.0, M10); PolyMac<1, 1>(pb[W], pb[Z], 1.0, M11); PolyMac<1, 2>(pb[W], pc[Z], 1.0, M12);
    lin[0] = pa[W][0] - pb[Z][0]; lin[1] = pa[W][1] - pb[Z][1];
    PolyMac<1, 1>(lin, pa[W], 1.0, M10); PolyMac<1, 1>(lin, pb[W], 1.0, M11); PolyMac<1, 2>(lin, pc[W], 1.0, M12);
    PolyMac<1, 1>(pa[Z], pa[Y], -1.0, M10); PolyMac<1, 1>(pa[Z], pb[Y], -1.0, M11); PolyMac<1, 2>(pa[Z], pc[Y], -1.0, M12);
#pragma unroll
    for (int k = 0; k < 3; ++k) { M11[k] += pc[W][k]; M10[k] -= pc[Z][k]; }
    // row 2: Q_W^2 - Q_Y Q_Z, quadratic monomials substituted once more
    double ky[3] = {0, 0, 0}, kz[3] = {0, 0, 0}, kyz[3] = {0, 0, 0};
    PolyMac<1, 1>(pa[W], pa[W], 1.0, ky); PolyMac<1, 1>(pa[Y], pa[Z], -1.0, ky);
    PolyMac<1, 1>(pb[W], pb[W], 1.0, kz); PolyMac<1, 1>(pb[Y], pb[Z], -1.0, kz);
    PolyMac<1, 1>(pa[W], pb[W], 2.0, kyz); PolyMac<1, 1>(pa[Y], pb[Z], -1.0, kyz); PolyMac<1, 1>(pb[Y], pa[Z], -1.0, kyz);
    PolyMac<1, 2>(pa[W], pc[W], 2.0, M20); PolyMac<1, 2>(pa[Y], pc[Z], -1.0, M20); PolyMac<1, 2>(pa[Z], pc[Y], -1.0, M20);
    PolyMac<1, 2>(pb[W], pc[W], 2.0, M21); PolyMac<1, 2>(pb[Y], pc[Z], -1.0, M21); PolyMac<1, 2>(pb[Z], pc[Y], -1.0, M21);
    PolyMac<2, 2>(pc[W], pc[W], 1.0, M22); PolyMac<2, 2>(pc[Y], pc[Z], -1.0, M22);
    PolyMac<2, 1>(ky, pa[Y], 1.0, M20); PolyMac<2, 1>(ky, pb[Y], 1.0, M21); PolyMac<2, 2>(ky, pc[Y], 1.0, M22);
    PolyMac<2, 1>(kz, pa[Z], 1.0, M20); PolyMac<2, 1>(kz, pb[Z], 1.0, M21); PolyMac<2, 2>(kz, pc[Z], 1.0, M22);
    PolyMac<2, 1>(kyz, pa[W], 1.0, M20); PolyMac<2, 1>(kyz, pb[W], 1.0, M21); PolyMac<2, 2>(kyz, pc[W], 1.0, M22);
  }
  // det M(X): expansion along the third column
  double d[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  {
    double r12[6] = {0, 0, 0, 0, 0, 0}, r02[6] = {0, 0, 0, 0, 0, 0}, r01[5] = {0, 0, 0, 0, 0};
    PolyMac<2, 3>(M10, M21, 1.0, r12); PolyMac<2, 3>(M11, M20, -1.0, r12);
    PolyMac<2, 3>(M00, M21, 1.0, r02); PolyMac<2, 3>(M01, M20, -1.0, r02);
    PolyMac<2, 2>(M00, M11, 1.0, r01); PolyMac<2, 2>(M01, M10, -1.0, r01);
    PolyMac<3, 5>(M02, r12, 1.0, d);
    PolyMac<3, 5>(M12, r02, -1.0, d);
    PolyMac<4, 4>(M22, r01, 1.0, d);
  }
  const double lead = d[8];
  if (!(fabs(lead) > 0.0) || !isfinite(lead)) return 0;
  double mono[8];
  bool finite = true;
#pragma unroll
  for (int k = 0; k < 8; ++k) { mono[k] = d[k] / lead; finite = finite && isfinite(mono[k]); }
  if (!finite) return 0;
  double zr[8], zi[8];
  AberthOctic(mono, zr, zi);

  // accept |Im| <= 1e-8 (re3q3.h:169-171), polish on the real polynomial, sort ascending
  double xs[8];
  int n = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    if (fabs(zi[k]) > 1e-8 || !isfinite(zr[k])) continue;
    double x = zr[k];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      double p = 1.0, dp = 0.0;
#pragma unroll
      for (int e = 7; e >= 0; --e) { dp = dp * x + p; p = p * x + mono[e]; }
      if (dp != 0.0) { const double nx = x - p / dp; if (isfinite(nx) && fabs(nx - x) <= 1e-6 * (1.0 + fabs(x))) x = nx; }
    }
    // insertion into the sorted prefix (n <= 8, unrolled compare-swaps)
    double v = x;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      if (s < n && v < xs[s]) { const double tmp = xs[s]; xs[s] = v; v = tmp; }
    }
#pragma unroll
    for (int s = 0; s < 8; ++s) if (s == n) xs[s] = v;
    ++n;
  }
  // back-substitution from rows 0-1 of M(X) (re3q3.h:177-188)
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    if (k >= n) continue;
    const double x = xs[k];
    const double m00 = PolyVal<2>(M00, x), m01 = PolyVal<2>(M01, x), m02 = PolyVal<3>(M02, x);
    const double m10 = PolyVal<2>(M10, x), m11 = PolyVal<2>(M11, x), m12 = PolyVal<3>(M12, x);
    const double y = (m12 * m01 - m02 * m11) / (m00 * m11 - m10 * m01);
    const double z = (m12 * m00 - m02 * m10) / (m01 * m10 - m11 * m00);
    double v0, v1, v2;
    if (elim == 0) { v0 = x; v1 = y; v2 = z; }
    else if (elim == 1) { v0 = y; v1 = x; v2 = z; }
    else { v0 = z; v1 = y; v2 = x; }
    if (changed) {
      const double w0 = A4[0] * v0 + A4[1] * v1 + A4[2] * v2 + A4[3];
      const double w1 = A4[4] * v0 + A4[5] * v1 + A4[6] * v2 + A4[7];
      const double w2 = A4[8] * v0 + A4[9] * v1 + A4[10] * v2 + A4[11];
      v0 = w0; v1 = w1; v2 = w2;
    }
    sol[k] = v0; sol[8 + k] = v1; sol[16 + k] = v2;
  }
  return n;
}

// six (line, point) pairs -> up to 8 poses [R|t] (3x4 row-major, 12 doubles each)
PP_HD int P6LDevice(const double* L /*6x3*/, const double* X /*6x3*/, bool all_aligned, double* models,
                    const double* mix, const double* affine) {
  if (all_aligned) return 0;                                       // absolute_pose.cc:87-97
  // tt / Rc rows: kron(X_i', l_i')  (:101-123)
  double tt[27], Rc[27];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        tt[9 * i + 3 * a + b] = X[3 * i + a] * L[3 * i + b];
        Rc[9 * i + 3 * a + b] = X[3 * (i + 3) + a] * L[3 * (i + 3) + b];
      }
  double Bt[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) Bt[i] = L[i];   // rows of B^T are lines 0..2
  if (fabs(Det3x3(Bt)) < 1e-10) {              // :126-134
    double A[9];
    if (mix) { for (int i = 0; i < 9; ++i) A[i] = mix[i]; }
    else { DegenerateStream rng(2); for (int i = 0; i < 9; ++i) A[i] = rng.Next(); }
    double tt2[27];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 9; ++j) tt2[9 * i + j] = tt[9 * i + j] + (A[3 * i] * Rc[j] + A[3 * i + 1] * Rc[9 + j] + A[3 * i + 2] * Rc[18 + j]);
#pragma unroll
    for (int i = 0; i < 27; ++i) tt[i] = tt2[i];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) Bt[3 * i + j] += A[3 * i] * L[9 + j] + A[3 * i + 1] * L[12 + j] + A[3 * i + 2] * L[15 + j];
  }
  if (!Solve3<9>(Bt, tt)) return 0;            // tt = (B^T)^-1 tt   (:137)
#pragma unroll
  for (int i = 0; i < 3; ++i)                  // Rc -= L1^T tt       (:138)
#pragma unroll
    for (int j = 0; j < 9; ++j) Rc[9 * i + j] -= L[3 * (i + 3)] * tt[j] + L[3 * (i + 3) + 1] * tt[9 + j] + L[3 * (i + 3) + 2] * tt[18 + j];
  // linear constraints on vec(R) (column-major) -> quadrics in the Cayley parameters (:46-62)
  double co[30];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const double* r = Rc + 9 * k;
    double* o = co + 10 * k;
    o[0] = r[0] - r[4] - r[8];
    o[1] = 2 * r[1] + 2 * r[3];
    o[2] = 2 * r[2] + 2 * r[6];
    o[3] = r[4] - r[0] - r[8];
    o[4] = 2 * r[5] + 2 * r[7];
    o[5] = r[8] - r[4] - r[0];
    o[6] = 2 * r[5] - 2 * r[7];
    o[7] = 2 * r[6] - 2 * r[2];
    o[8] = 2 * r[1] - 2 * r[3];
    o[9] = r[0] + r[4] + r[8];
  }
  double sol[24];
  const int n = Re3q3Device(co, sol, true, affine);
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    if (s >= n) continue;
    const double x = sol[s], y = sol[8 + s], z = sol[16 + s];
    double R[9];                                // cayley_param (:64-75)
    R[0] = x * x - y * y - z * z + 1; R[1] = 2 * x * y - 2 * z;         R[2] = 2 * y + 2 * x * z;
    R[3] = 2 * z + 2 * x * y;         R[4] = y * y - x * x - z * z + 1; R[5] = 2 * y * z - 2 * x;
    R[6] = 2 * x * z - 2 * y;         R[7] = 2 * x + 2 * y * z;         R[8] = z * z - y * y - x * x + 1;
    const double sc = 1 + x * x + y * y + z * z;
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] /= sc;
    double* Mo = models + 12 * s;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      double t = 0;                             // t = -tt vec(R), vec column-major (:154)
#pragma unroll
      for (int col = 0; col < 3; ++col)
#pragma unroll
        for (int row = 0; row < 3; ++row) t += tt[9 * i + 3 * col + row] * R[3 * row + col];
      Mo[4 * i] = R[3 * i]; Mo[4 * i + 1] = R[3 * i + 1]; Mo[4 * i + 2] = R[3 * i + 2]; Mo[4 * i + 3] = -t;
    }
  }
  return n;
}

}  // namespace ppsfm

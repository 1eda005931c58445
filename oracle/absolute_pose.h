// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// CPU restatement of the six-line absolute-pose hypothesis path:
//   * squared line reprojection error     reference src/estimators/utils.cc:40-89
//     (exact association of :70-84 is kept: the inlier test must be bit-identical)
//   * inlier support (count, sum) + order  src/optim/support_measurement.cc:36-60
//   * three quadrics in three unknowns     lib/re3q3/re3q3/re3q3.h:16-200
//   * P6L minimal solver                   src/estimators/absolute_pose.cc:46-162
//
// re3q3 is restated as ALGEBRA, not as the reference's expanded expressions: after solving
// for the three pure-quadratic monomials of the two kept unknowns (re3q3.h:66-80) the three
// compatibility identities  y(yz)=z(y^2),  z(yz)=y(z^2),  (yz)^2=(y^2)(z^2)  give a 3x3
// polynomial matrix M(x) of degrees [[2,2,3],[2,2,3],[3,3,4]] (re3q3.h:84-137 are its entries
// written out); det M(x) is the degree-8 resultant (:139-150).  Here the entries and the
// determinant are formed with small fixed-size polynomial arithmetic.  Same mathematics,
// different rounding; root order: ASCENDING in the eliminated variable (the reference's order
// is whatever Eigen::EigenSolver returns, which is not reproducible without Eigen).
#pragma once
#include <algorithm>
#include <cfloat>
#include <cstdint>
#include <cstring>
#include "linalg.h"

namespace oracle {

// ---- scoring --------------------------------------------------------------------------
// P is 3x4 row-major.  residuals[i] = squared normalised point-to-line distance, DBL_MAX if
// the point is not in front of the camera.
inline void SquaredLineReprojectionError(int n, const double* lines /*n x 3*/, const double* pts /*n x 3*/,
                                         const double P[12], double* residuals) {
  const double P00 = P[0], P01 = P[1], P02 = P[2], P03 = P[3];
  const double P10 = P[4], P11 = P[5], P12 = P[6], P13 = P[7];
  const double P20 = P[8], P21 = P[9], P22 = P[10], P23 = P[11];
  for (int i = 0; i < n; ++i) {
    const double X0 = pts[3 * i], X1 = pts[3 * i + 1], X2 = pts[3 * i + 2];
    const double pz = P20 * X0 + P21 * X1 + P22 * X2 + P23;
    if (pz > DBL_EPSILON) {
      const double px = P00 * X0 + P01 * X1 + P02 * X2 + P03;
      const double py = P10 * X0 + P11 * X1 + P12 * X2 + P13;
      const double l0 = lines[3 * i], l1 = lines[3 * i + 1], l2 = lines[3 * i + 2];
      const double inv = 1.0 / pz;
      const double res = px * l0 * inv + py * l1 * inv + l2;
      residuals[i] = res * res;
    } else {
      residuals[i] = DBL_MAX;
    }
  }
}

struct Support { uint64_t num_inliers = 0; double residual_sum = DBL_MAX; };  // defaults: support_measurement.h:44-50

inline Support EvaluateSupport(int n, const double* residuals, double max_residual) {
  Support s;
  s.num_inliers = 0; s.residual_sum = 0;
  for (int i = 0; i < n; ++i)
    if (residuals[i] <= max_residual) { s.num_inliers += 1; s.residual_sum += residuals[i]; }
  return s;
}
inline bool SupportBetter(const Support& a, const Support& b) {
  if (a.num_inliers > b.num_inliers) return true;
  return a.num_inliers == b.num_inliers && a.residual_sum < b.residual_sum;
}

// ---- tiny polynomial helpers (coefficients ascending: p[k] multiplies x^k) -----------------
template <int DA, int DB>
inline void PolyMulAcc(const double* a, const double* b, double sign, double* out /*deg DA+DB*/) {
  for (int i = 0; i <= DA; ++i)
    for (int j = 0; j <= DB; ++j) out[i + j] += sign * a[i] * b[j];
}

// deterministic stand-in for the C rand() draws of the reference's degenerate branches
// (re3q3.h:41-42, absolute_pose.cc:130-131): a fixed-seed splitmix64 stream in [-1,1).
struct DegenerateRng {
  uint64_t s;
  explicit DegenerateRng(uint64_t seed) : s(seed) {}
  double next() {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return (double)(z >> 11) * (2.0 / 9007199254740992.0) - 1.0;
  }
};

// coeffs: 3 x 10 row-major, monomials x^2, xy, xz, y^2, yz, z^2, x, y, z, 1 (re3q3.h:12-15).
// solutions: 3 x 8 row-major (column k = k-th solution).  `affine` (3x4 row-major, rotation |
// translation) is the change of variables used when all three elimination determinants are
// < 1e-10 (re3q3.h:39-64); null => drawn from DegenerateRng(seed 1).
inline int Re3q3(const double coeffs_in[30], double solutions[24], bool try_var_change = true,
                 const double* affine = nullptr) {
  double c[3][10];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 10; ++j) c[i][j] = coeffs_in[i * 10 + j];

  // choice of elimination variable by the largest |det| of the 3x3 block of the OTHER two
  // variables' quadratic monomials (re3q3.h:17-37)
  const int qcols[3][3] = {{3, 5, 4}, {0, 5, 2}, {3, 0, 1}};
  double dets[3];
  for (int e = 0; e < 3; ++e) {
    double A[9];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) A[i * 3 + j] = c[i][qcols[e][j]];
    dets[e] = std::fabs(Det3(A));
  }
  int elim = 0; double det = dets[0];
  if (det < dets[1]) { det = dets[1]; elim = 1; }
  if (det < dets[2]) { det = dets[2]; elim = 2; }

  if (try_var_change && det < 1e-10) {
    double A[12];
    if (affine) { std::memcpy(A, affine, sizeof(A)); }
    else {
      DegenerateRng rng(1);
      double q[4], n = 0; for (int i = 0; i < 4; ++i) { q[i] = rng.next(); n += q[i] * q[i]; }
      n = std::sqrt(n); for (int i = 0; i < 4; ++i) q[i] /= n;
      const double w = q[0], x = q[1], y = q[2], z = q[3];
      const double R[9] = {1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                           2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                           2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)};
      double tv[3], tn = 0; for (int i = 0; i < 3; ++i) { tv[i] = rng.next(); tn += tv[i] * tv[i]; }
      tn = std::sqrt(tn);
      for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) A[i * 4 + j] = R[i * 3 + j]; A[i * 4 + 3] = tv[i] / tn; }
    }
    // substitute (x,y,z) = A3 (x',y',z') + a : every monomial becomes a quadratic form in the new vars
    // v_k = A[k][0] x' + A[k][1] y' + A[k][2] z' + A[k][3]
    auto prod = [&](int r, int s, double out[10]) {  // coefficients of v_r * v_s in the 10 monomials
      const double* a = A + 4 * r; const double* b = A + 4 * s;
      out[0] = a[0] * b[0]; out[1] = a[0] * b[1] + a[1] * b[0]; out[2] = a[0] * b[2] + a[2] * b[0];
      out[3] = a[1] * b[1]; out[4] = a[1] * b[2] + a[2] * b[1]; out[5] = a[2] * b[2];
      out[6] = a[0] * b[3] + a[3] * b[0]; out[7] = a[1] * b[3] + a[3] * b[1]; out[8] = a[2] * b[3] + a[3] * b[2];
      out[9] = a[3] * b[3];
    };
    double B[10][10] = {};
    prod(0, 0, B[0]); prod(0, 1, B[1]); prod(0, 2, B[2]); prod(1, 1, B[3]); prod(1, 2, B[4]); prod(2, 2, B[5]);
    for (int k = 0; k < 3; ++k) { B[6 + k][6] = A[4 * k]; B[6 + k][7] = A[4 * k + 1]; B[6 + k][8] = A[4 * k + 2]; B[6 + k][9] = A[4 * k + 3]; }
    B[9][9] = 1.0;
    double c2[30];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 10; ++j) { double s = 0; for (int k = 0; k < 10; ++k) s += c[i][k] * B[k][j]; c2[i * 10 + j] = s; }
    double sol2[24];
    const int n = Re3q3(c2, sol2, false, nullptr);
    for (int k = 0; k < n; ++k)
      for (int i = 0; i < 3; ++i)
        solutions[i * 8 + k] = A[4 * i] * sol2[k] + A[4 * i + 1] * sol2[8 + k] + A[4 * i + 2] * sol2[16 + k] + A[4 * i + 3];
    return n;
  }

  // Rename variables so that the eliminated one is "x" and the kept ones are "y","z":
  //   elim 0 (x): (x,y,z) ; elim 1 (y): (y,x,z) ; elim 2 (z): (z,y,x)      (re3q3.h:68-80, 193-197)
  // cols: [y^2, z^2, yz | x^2, xy, xz, x, y, z, 1] in the renamed variables
  const int perm[3][10] = {{3, 5, 4, 0, 1, 2, 6, 7, 8, 9},
                           {0, 5, 2, 3, 1, 4, 7, 6, 8, 9},
                           {3, 0, 1, 5, 4, 2, 8, 7, 6, 9}};
  double A[9], Bm[21];
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) A[i * 3 + j] = c[i][perm[elim][j]];
    for (int j = 0; j < 7; ++j) Bm[i * 7 + j] = c[i][perm[elim][3 + j]];
  }
  if (!LuSolve(3, 7, A, Bm)) return 0;
  // P[r] (r = 0: y^2, 1: z^2, 2: yz) = -(A^-1 B) over monomials [x^2, xy, xz, x, y, z, 1]
  double P[3][7];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 7; ++j) P[i][j] = -Bm[i * 7 + j];

  // q_r = a_r(x) y + b_r(x) z + c_r(x),  deg a,b = 1, deg c = 2
  double a[3][2], b[3][2], cc[3][3];
  for (int r = 0; r < 3; ++r) {
    a[r][0] = P[r][4]; a[r][1] = P[r][1];
    b[r][0] = P[r][5]; b[r][1] = P[r][2];
    cc[r][0] = P[r][6]; cc[r][1] = P[r][3]; cc[r][2] = P[r][0];
  }
  // M(x): rows = identities, cols = coefficient of y, z, 1.  Y = y^2 (r=0), Z = z^2 (r=1), W = yz (r=2).
  // row0: y*W - z*Y = 0  ->  aW*Y + (bW - aY)*W + cW*y - bY*Z - cY*z
  // row1: z*W - y*Z = 0  ->  bW*Z + (aW - bZ)*W + cW*z - aZ*Y - cZ*y
  // row2: W*W - Y*Z = 0  (expanded below)
  double M[3][3][5] = {};
  {
    const int Y = 0, Z = 1, W = 2;
    // helper: out += sign * lin(deg1) * q_r  spread over (y,z,1) columns
    auto addLinTimesQ = [&](int row, const double lin[2], int r, double sign) {
      PolyMulAcc<1, 1>(lin, a[r], sign, M[row][0]);
      PolyMulAcc<1, 1>(lin, b[r], sign, M[row][1]);
      PolyMulAcc<1, 2>(lin, cc[r], sign, M[row][2]);
    };
    double lin[2];
    // row 0
    addLinTimesQ(0, a[W], Y, 1.0);
    lin[0] = b[W][0] - a[Y][0]; lin[1] = b[W][1] - a[Y][1]; addLinTimesQ(0, lin, W, 1.0);
    addLinTimesQ(0, b[Y], Z, -1.0);
    for (int k = 0; k < 3; ++k) { M[0][0][k] += cc[W][k]; M[0][1][k] -= cc[Y][k]; }
    // row 1
    addLinTimesQ(1, b[W], Z, 1.0);
    lin[0] = a[W][0] - b[Z][0]; lin[1] = a[W][1] - b[Z][1]; addLinTimesQ(1, lin, W, 1.0);
    addLinTimesQ(1, a[Z], Y, -1.0);
    for (int k = 0; k < 3; ++k) { M[1][1][k] += cc[W][k]; M[1][0][k] -= cc[Z][k]; }
    // row 2: (aW y + bW z + cW)^2 - (aY y + bY z + cY)(aZ y + bZ z + cZ)
    //  y^2: aW^2 - aY aZ ; z^2: bW^2 - bY bZ ; yz: 2 aW bW - aY bZ - bY aZ        (deg 2 each)
    //  y: 2 aW cW - aY cZ - cY aZ ; z: 2 bW cW - bY cZ - cY bZ                    (deg 3)
    //  1: cW^2 - cY cZ                                                            (deg 4)
    double ky[3] = {}, kz[3] = {}, kyz[3] = {};
    PolyMulAcc<1, 1>(a[W], a[W], 1.0, ky); PolyMulAcc<1, 1>(a[Y], a[Z], -1.0, ky);
    PolyMulAcc<1, 1>(b[W], b[W], 1.0, kz); PolyMulAcc<1, 1>(b[Y], b[Z], -1.0, kz);
    PolyMulAcc<1, 1>(a[W], b[W], 2.0, kyz); PolyMulAcc<1, 1>(a[Y], b[Z], -1.0, kyz); PolyMulAcc<1, 1>(b[Y], a[Z], -1.0, kyz);
    PolyMulAcc<1, 2>(a[W], cc[W], 2.0, M[2][0]); PolyMulAcc<1, 2>(a[Y], cc[Z], -1.0, M[2][0]); PolyMulAcc<1, 2>(a[Z], cc[Y], -1.0, M[2][0]);
    PolyMulAcc<1, 2>(b[W], cc[W], 2.0, M[2][1]); PolyMulAcc<1, 2>(b[Y], cc[Z], -1.0, M[2][1]); PolyMulAcc<1, 2>(b[Z], cc[Y], -1.0, M[2][1]);
    PolyMulAcc<2, 2>(cc[W], cc[W], 1.0, M[2][2]); PolyMulAcc<2, 2>(cc[Y], cc[Z], -1.0, M[2][2]);
    // substitute y^2, z^2, yz once more
    for (int r = 0; r < 3; ++r) {
      const double* k = (r == 0) ? ky : (r == 1) ? kz : kyz;
      PolyMulAcc<2, 1>(k, a[r], 1.0, M[2][0]);
      PolyMulAcc<2, 1>(k, b[r], 1.0, M[2][1]);
      PolyMulAcc<2, 2>(k, cc[r], 1.0, M[2][2]);
    }
  }
  // det M(x), degree 8:  expand along the third column (degrees 3,3,4)
  double d[9] = {};
  {
    // 2x2 minors of columns (0,1) over rows (1,2), (0,2), (0,1)
    double r12[6] = {}, r02[6] = {}, r01[5] = {};
    PolyMulAcc<2, 3>(M[1][0], M[2][1], 1.0, r12); PolyMulAcc<2, 3>(M[1][1], M[2][0], -1.0, r12);
    PolyMulAcc<2, 3>(M[0][0], M[2][1], 1.0, r02); PolyMulAcc<2, 3>(M[0][1], M[2][0], -1.0, r02);
    PolyMulAcc<2, 2>(M[0][0], M[1][1], 1.0, r01); PolyMulAcc<2, 2>(M[0][1], M[1][0], -1.0, r01);
    PolyMulAcc<3, 5>(M[0][2], r12, 1.0, d);
    PolyMulAcc<3, 5>(M[1][2], r02, -1.0, d);
    PolyMulAcc<4, 4>(M[2][2], r01, 1.0, d);
  }
  // companion matrix of the monic polynomial (normalised by the LEADING... see note) — the
  // reference divides by c(0), its x^8... coefficient is c(0) there because it stores
  // descending powers (re3q3.h:152-160).  Here d[8] is the leading coefficient.
  double comp[64] = {};
  for (int j = 0; j < 8; ++j) comp[j] = -d[7 - j] / d[8];
  for (int i = 1; i < 8; ++i) comp[i * 8 + (i - 1)] = 1.0;
  std::complex<double> roots[8];
  if (!HessenbergEigenvalues(8, comp, roots)) return 0;

  double xs[8]; int n = 0;
  for (int i = 0; i < 8; ++i) {
    if (std::fabs(roots[i].imag()) > 1e-8) continue;
    if (!std::isfinite(roots[i].real())) continue;
    xs[n++] = roots[i].real();
  }
  std::sort(xs, xs + n);
  for (int k = 0; k < n; ++k) {
    const double x = xs[k];
    double Mv[2][3];
    for (int r = 0; r < 2; ++r)
      for (int col = 0; col < 3; ++col) {
        double v = 0.0; for (int e = 4; e >= 0; --e) v = v * x + M[r][col][e];
        Mv[r][col] = v;
      }
    // rows 0-1:  M00 y + M01 z + M02 = 0 ; M10 y + M11 z + M12 = 0   (2x2 Cramer, re3q3.h:182-188)
    const double y = (Mv[1][2] * Mv[0][1] - Mv[0][2] * Mv[1][1]) / (Mv[0][0] * Mv[1][1] - Mv[1][0] * Mv[0][1]);
    const double z = (Mv[1][2] * Mv[0][0] - Mv[0][2] * Mv[1][0]) / (Mv[0][1] * Mv[1][0] - Mv[1][1] * Mv[0][0]);
    double v[3];
    if (elim == 0) { v[0] = x; v[1] = y; v[2] = z; }
    else if (elim == 1) { v[0] = y; v[1] = x; v[2] = z; }
    else { v[0] = z; v[1] = y; v[2] = x; }
    solutions[k] = v[0]; solutions[8 + k] = v[1]; solutions[16 + k] = v[2];
  }
  return n;
}

// Cayley rotation from (x,y,z)                                   (absolute_pose.cc:64-75)
inline void CayleyRotation(const double c[3], double R[9]) {
  const double x = c[0], y = c[1], z = c[2];
  R[0] = x * x - y * y - z * z + 1; R[1] = 2 * x * y - 2 * z;         R[2] = 2 * y + 2 * x * z;
  R[3] = 2 * z + 2 * x * y;         R[4] = y * y - x * x - z * z + 1; R[5] = 2 * y * z - 2 * x;
  R[6] = 2 * x * z - 2 * y;         R[7] = 2 * x + 2 * y * z;         R[8] = z * z - y * y - x * x + 1;
  const double s = 1 + x * x + y * y + z * z;
  for (int i = 0; i < 9; ++i) R[i] /= s;
}

// lines6 6x3, points6 6x3 row-major, aligned6: gravity-aligned flags.  models: up to 8 x 12 (3x4 row-major).
// mix (3x3 row-major) is the combination matrix of the degenerate branch (absolute_pose.cc:128-134);
// null => DegenerateRng(seed 2).
inline int P6LEstimate(const double* lines6, const double* points6, const uint8_t* aligned6,
                       double* models, const double* mix = nullptr, const double* affine = nullptr) {
  bool all_aligned = true;
  for (int i = 0; i < 6; ++i) all_aligned = all_aligned && (aligned6 && aligned6[i]);
  if (all_aligned) return 0;

  // row i of tt / Rc = kron(X_i', l_i') :  entry 3*a+b = X_i[a] * l_i[b]      (:101-123)
  double tt[27], Rc[27];
  for (int i = 0; i < 3; ++i)
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 3; ++b) {
        tt[i * 9 + 3 * a + b] = points6[3 * i + a] * lines6[3 * i + b];
        Rc[i * 9 + 3 * a + b] = points6[3 * (i + 3) + a] * lines6[3 * (i + 3) + b];
      }
  // L0 = [l0 l1 l2] (3x3, columns = lines), B = L0; need B^T t-coefficients: rows of B^T are lines
  double Bt[9];  // B^T row-major: row i = line i
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Bt[i * 3 + j] = lines6[3 * i + j];
  if (std::fabs(Det3(Bt)) < 1e-10) {
    double A[9];
    if (mix) std::memcpy(A, mix, sizeof(A));
    else { DegenerateRng rng(2); for (int i = 0; i < 9; ++i) A[i] = rng.next(); }
    // tt += A * Rc ; B += L1 * A^T  =>  B^T += A * L1^T  (L1^T rows = lines 3..5)
    double tt2[27];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 9; ++j) { double s = tt[i * 9 + j]; for (int k = 0; k < 3; ++k) s += A[i * 3 + k] * Rc[k * 9 + j]; tt2[i * 9 + j] = s; }
    std::memcpy(tt, tt2, sizeof(tt));
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { double s = Bt[i * 3 + j]; for (int k = 0; k < 3; ++k) s += A[i * 3 + k] * lines6[3 * (k + 3) + j]; Bt[i * 3 + j] = s; }
  }
  if (!LuSolve(3, 9, Bt, tt)) return 0;                       // tt = (B^T)^-1 tt        (:137)
  for (int i = 0; i < 3; ++i)                                  // Rc -= L1^T tt          (:138)
    for (int j = 0; j < 9; ++j) {
      double s = 0; for (int k = 0; k < 3; ++k) s += lines6[3 * (i + 3) + k] * tt[k * 9 + j];
      Rc[i * 9 + j] -= s;
    }
  // linear constraints on vec(R) (column-major r = [R00 R10 R20 R01 ...]) -> quadrics in the Cayley
  // parameters (:46-62)
  double co[30];
  for (int k = 0; k < 3; ++k) {
    const double* r = Rc + 9 * k; double* o = co + 10 * k;
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
  const int n = Re3q3(co, sol, true, affine);
  for (int s = 0; s < n; ++s) {
    const double cay[3] = {sol[s], sol[8 + s], sol[16 + s]};
    double R[9];
    CayleyRotation(cay, R);
    double* M = models + 12 * s;
    for (int i = 0; i < 3; ++i) {
      // t_i = - sum_j tt[i][j] * vec(R)[j], vec column-major: j = 3*col + row     (:154)
      double t = 0;
      for (int col = 0; col < 3; ++col) for (int row = 0; row < 3; ++row) t += tt[i * 9 + 3 * col + row] * R[row * 3 + col];
      M[i * 4 + 0] = R[i * 3 + 0]; M[i * 4 + 1] = R[i * 3 + 1]; M[i * 4 + 2] = R[i * 3 + 2];
      M[i * 4 + 3] = -t;
    }
  }
  return n;
}

}  // namespace oracle

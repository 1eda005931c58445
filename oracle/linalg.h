// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// Small dense linear algebra the restatement needs in place of Eigen (third-party, NOT in
// /root/reference, version unpinned — CMakeLists.txt:94):
//   * 3x3 determinant                      (Eigen determinant():  absolute_pose.cc:126, re3q3.h:23-25)
//   * LU with partial pivoting, n <= 3     (partialPivLu()/lu():  absolute_pose.cc:137, re3q3.h:71-79)
//   * eigenvalues of a real upper-Hessenberg matrix by the shifted Francis QR iteration
//     (Eigen::EigenSolver -> RealSchur, no balancing: re3q3.h:164-165).  Published algorithm
//     (EISPACK hqr); bits differ from Eigen's, roots agree to ~1e-10 relative.
//   * dense Cholesky (for the reduced camera system of the BA restatement)
#pragma once
#include <algorithm>
#include <cmath>
#include <complex>
#include <vector>

namespace oracle {

inline double Det3(const double A[9]) {  // row-major
  return A[0] * (A[4] * A[8] - A[5] * A[7]) - A[1] * (A[3] * A[8] - A[5] * A[6]) +
         A[2] * (A[3] * A[7] - A[4] * A[6]);
}

// Solve A X = B, A n x n row-major (overwritten), B n x m row-major (overwritten with X).
// Partial (row) pivoting.  Returns false if a pivot is exactly zero.
inline bool LuSolve(int n, int m, double* A, double* B) {
  for (int k = 0; k < n; ++k) {
    int piv = k; double best = std::fabs(A[k * n + k]);
    for (int i = k + 1; i < n; ++i) { const double v = std::fabs(A[i * n + k]); if (v > best) { best = v; piv = i; } }
    if (best == 0.0) return false;
    if (piv != k) {
      for (int j = 0; j < n; ++j) std::swap(A[k * n + j], A[piv * n + j]);
      for (int j = 0; j < m; ++j) std::swap(B[k * m + j], B[piv * m + j]);
    }
    for (int i = k + 1; i < n; ++i) {
      const double f = A[i * n + k] / A[k * n + k];
      A[i * n + k] = f;
      for (int j = k + 1; j < n; ++j) A[i * n + j] -= f * A[k * n + j];
      for (int j = 0; j < m; ++j) B[i * m + j] -= f * B[k * m + j];
    }
  }
  for (int j = 0; j < m; ++j) {
    for (int i = n - 1; i >= 0; --i) {
      double s = B[i * m + j];
      for (int c = i + 1; c < n; ++c) s -= A[i * n + c] * B[c * m + j];
      B[i * m + j] = s / A[i * n + i];
    }
  }
  return true;
}

// Eigenvalues of an n x n real upper-Hessenberg matrix H (row-major, destroyed).
// Returns false if the QR iteration does not converge.
inline bool HessenbergEigenvalues(int n, double* a, std::complex<double>* w) {
  auto A = [&](int i, int j) -> double& { return a[i * n + j]; };
  double anorm = 0.0;
  for (int i = 0; i < n; ++i)
    for (int j = (i > 0 ? i - 1 : 0); j < n; ++j) anorm += std::fabs(A(i, j));
  int nn = n - 1;
  double t = 0.0;
  double p = 0, q = 0, r = 0, s = 0, x = 0, y = 0, z = 0, ww = 0;
  while (nn >= 0) {
    int its = 0, l;
    do {
      for (l = nn; l >= 1; --l) {
        s = std::fabs(A(l - 1, l - 1)) + std::fabs(A(l, l));
        if (s == 0.0) s = anorm;
        if (std::fabs(A(l, l - 1)) + s == s) { A(l, l - 1) = 0.0; break; }
      }
      x = A(nn, nn);
      if (l == nn) {  // one real root
        w[nn--] = std::complex<double>(x + t, 0.0);
      } else {
        y = A(nn - 1, nn - 1);
        ww = A(nn, nn - 1) * A(nn - 1, nn);
        if (l == nn - 1) {  // a 2x2 block: two roots
          p = 0.5 * (y - x);
          q = p * p + ww;
          z = std::sqrt(std::fabs(q));
          x += t;
          if (q >= 0.0) {
            z = p + (p >= 0.0 ? std::fabs(z) : -std::fabs(z));
            w[nn - 1] = w[nn] = std::complex<double>(x + z, 0.0);
            if (z != 0.0) w[nn] = std::complex<double>(x - ww / z, 0.0);
          } else {
            w[nn - 1] = std::complex<double>(x + p, z);
            w[nn] = std::complex<double>(x + p, -z);
          }
          nn -= 2;
        } else {
          if (its == 60) return false;
          if (its == 10 || its == 20) {  // exceptional shift
            t += x;
            for (int i = 0; i <= nn; ++i) A(i, i) -= x;
            s = std::fabs(A(nn, nn - 1)) + std::fabs(A(nn - 1, nn - 2));
            y = x = 0.75 * s;
            ww = -0.4375 * s * s;
          }
          ++its;
          int m;
          for (m = nn - 2; m >= l; --m) {
            z = A(m, m);
            r = x - z;
            s = y - z;
            p = (r * s - ww) / A(m + 1, m) + A(m, m + 1);
            q = A(m + 1, m + 1) - z - r - s;
            r = A(m + 2, m + 1);
            s = std::fabs(p) + std::fabs(q) + std::fabs(r);
            p /= s; q /= s; r /= s;
            if (m == l) break;
            const double u = std::fabs(A(m, m - 1)) * (std::fabs(q) + std::fabs(r));
            const double v = std::fabs(p) * (std::fabs(A(m - 1, m - 1)) + std::fabs(z) + std::fabs(A(m + 1, m + 1)));
            if (u + v == v) break;
          }
          for (int i = m + 2; i <= nn; ++i) {
            A(i, i - 2) = 0.0;
            if (i != m + 2) A(i, i - 3) = 0.0;
          }
          for (int k = m; k <= nn - 1; ++k) {
            if (k != m) {
              p = A(k, k - 1);
              q = A(k + 1, k - 1);
              r = 0.0;
              if (k != nn - 1) r = A(k + 2, k - 1);
              if ((x = std::fabs(p) + std::fabs(q) + std::fabs(r)) != 0.0) { p /= x; q /= x; r /= x; }
            }
            const double nrm = std::sqrt(p * p + q * q + r * r);
            s = (p >= 0.0 ? nrm : -nrm);
            if (s != 0.0) {
              if (k == m) {
                if (l != m) A(k, k - 1) = -A(k, k - 1);
              } else {
                A(k, k - 1) = -s * x;
              }
              p += s;
              x = p / s; y = q / s; z = r / s;
              q /= p; r /= p;
              for (int j = k; j <= nn; ++j) {
                p = A(k, j) + q * A(k + 1, j);
                if (k != nn - 1) { p += r * A(k + 2, j); A(k + 2, j) -= p * z; }
                A(k + 1, j) -= p * y;
                A(k, j) -= p * x;
              }
              const int mmin = nn < k + 3 ? nn : k + 3;
              for (int i = l; i <= mmin; ++i) {
                p = x * A(i, k) + y * A(i, k + 1);
                if (k != nn - 1) { p += z * A(i, k + 2); A(i, k + 2) -= p * r; }
                A(i, k + 1) -= p * q;
                A(i, k) -= p;
              }
            }
          }
        }
      }
    } while (l < nn - 1);
  }
  return true;
}

// In-place dense Cholesky A = L L^T (lower triangle of row-major n x n), returns false if not SPD.
// Column by column; the rows below the pivot are independent dot products (OpenMP when available: the
// cpu_baseline leg of bench.py times this on all host cores).
inline bool CholeskyFactor(int n, double* A) {
  for (int j = 0; j < n; ++j) {
    double d = A[(size_t)j * n + j];
    const double* rj = A + (size_t)j * n;
    for (int k = 0; k < j; ++k) d -= rj[k] * rj[k];
    if (!(d > 0.0)) return false;
    d = std::sqrt(d);
    A[(size_t)j * n + j] = d;
#pragma omp parallel for schedule(static) if (n - j > 256)
    for (int i = j + 1; i < n; ++i) {
      const double* ri = A + (size_t)i * n;
      double s = ri[j];
      for (int k = 0; k < j; ++k) s -= ri[k] * rj[k];
      A[(size_t)i * n + j] = s / d;
    }
  }
  return true;
}
// The same factorisation, cache-blocked (right-looking, 64-column panels, OpenMP over the rows of the panel solve and over the
// tiles of the trailing update; every inner loop is a dot product of two contiguous row segments).  NOT used by the parity
// tests (they keep the column-by-column form above, whose summation order is the documented one); bench.py's cpu_baseline
// leg times the LM solve with this one (BAOptions::blocked_cholesky) so that the CPU figure is not an artefact of a
// cache-hostile Cholesky: at n = 3000 the unblocked form streams ~36 GB through the caches per factorisation.
inline bool CholeskyFactorBlocked(int n, double* A) {
  const int B = 64;
  for (int k = 0; k < n; k += B) {
    const int kb = std::min(B, n - k);
    for (int j = k; j < k + kb; ++j) {      // the diagonal block, unblocked
      double* rj = A + (size_t)j * n;
      double d = rj[j];
      for (int l = k; l < j; ++l) d -= rj[l] * rj[l];
      if (!(d > 0.0)) return false;
      d = std::sqrt(d);
      rj[j] = d;
      for (int i = j + 1; i < k + kb; ++i) {
        double* ri = A + (size_t)i * n;
        double s = ri[j];
        for (int l = k; l < j; ++l) s -= ri[l] * rj[l];
        ri[j] = s / d;
      }
    }
#pragma omp parallel for schedule(static)
    for (int i = k + kb; i < n; ++i) {      // panel: rows below, forward substitution with L_kk
      double* ri = A + (size_t)i * n;
      for (int j = k; j < k + kb; ++j) {
        const double* rj = A + (size_t)j * n;
        double s = ri[j];
        for (int l = k; l < j; ++l) s -= ri[l] * rj[l];
        ri[j] = s / rj[j];
      }
    }
    const int first = k + kb, nt = (n - first + B - 1) / B;
#pragma omp parallel for schedule(dynamic, 1) collapse(2)
    for (int ti = 0; ti < nt; ++ti)
      for (int tj = 0; tj < nt; ++tj) {      // trailing update, lower tiles only
        if (tj > ti) continue;
        const int i0 = first + ti * B, i1 = std::min(i0 + B, n), j0 = first + tj * B, j1 = std::min(j0 + B, n);
        for (int i = i0; i < i1; ++i) {
          double* ri = A + (size_t)i * n;
          const double* pi = ri + k;
          const int jend = std::min(j1, i + 1);
          for (int j = j0; j < jend; ++j) {
            const double* pj = A + (size_t)j * n + k;
            double s = 0.0;
            for (int l = 0; l < kb; ++l) s += pi[l] * pj[l];
            ri[j] -= s;
          }
        }
      }
  }
  return true;
}
inline void CholeskySolve(int n, const double* L, double* b) {
  for (int i = 0; i < n; ++i) { double s = b[i]; for (int k = 0; k < i; ++k) s -= L[i * n + k] * b[k]; b[i] = s / L[i * n + i]; }
  for (int i = n - 1; i >= 0; --i) { double s = b[i]; for (int k = i + 1; k < n; ++k) s -= L[k * n + i] * b[k]; b[i] = s / L[i * n + i]; }
}

}  // namespace oracle

// Smallest-eigenvalue eigenvector of a small symmetric matrix in registers (shared by the init solvers and the track
// triangulation): what Eigen's JacobiSVD(A).matrixV().col(last) returns for the matrix whose Gram matrix S = A^T A is.
#pragma once
#include "common.hpp"

namespace ppsfm {

// Cyclic Jacobi on a symmetric N x N matrix held in registers (all indices compile-time); returns the unit
// eigenvector of the smallest eigenvalue == last right singular vector of the matrix whose Gram matrix S is.
template <int N>
__device__ __forceinline__ void SmallestEigenvector(double (&S)[N * N], double (&v)[N]) {
  double V[N * N];
#pragma unroll
  for (int i = 0; i < N; ++i)
#pragma unroll
    for (int j = 0; j < N; ++j) V[i * N + j] = (i == j) ? 1.0 : 0.0;
#pragma unroll 1
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0.0;
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
      for (int j = i + 1; j < N; ++j) off += S[i * N + j] * S[i * N + j];
    if (off < 1e-300) break;
#pragma unroll
    for (int p = 0; p < N; ++p)
#pragma unroll
      for (int q = p + 1; q < N; ++q) {
        const double apq = S[p * N + q];
        if (apq != 0.0) {
          const double theta = (S[q * N + q] - S[p * N + p]) / (2.0 * apq);
          const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
          const double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
#pragma unroll
          for (int k = 0; k < N; ++k) { const double a = S[k * N + p], b = S[k * N + q]; S[k * N + p] = c * a - sn * b; S[k * N + q] = sn * a + c * b; }
#pragma unroll
          for (int k = 0; k < N; ++k) { const double a = S[p * N + k], b = S[q * N + k]; S[p * N + k] = c * a - sn * b; S[q * N + k] = sn * a + c * b; }
#pragma unroll
          for (int k = 0; k < N; ++k) { const double a = V[k * N + p], b = V[k * N + q]; V[k * N + p] = c * a - sn * b; V[k * N + q] = sn * a + c * b; }
        }
      }
  }
  double best = S[0];
#pragma unroll
  for (int k = 0; k < N; ++k) v[k] = V[k * N];
#pragma unroll
  for (int j = 1; j < N; ++j) {
    const bool lt = S[j * N + j] < best;
    best = lt ? S[j * N + j] : best;
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] = lt ? V[k * N + j] : v[k];
  }
}

}  // namespace ppsfm

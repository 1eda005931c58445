// K6 — four-view line initialisation: batched minimal solver, triangulate-all + score, LO-MSAC driver.
//
//   PlanarOffsetEstimator::{MinimalSolver, NonMinimalSolver, EvaluateModelOnPoint}, four_view_triangulate
//                                   reference src/init/initializer.cc:219-333
//   FourView2dEstimator::EvaluateModelOnPoint, three_view_triangulate2d   src/init/sfm2d.cc:194-213, 302-319
//   ransac_lib::LocallyOptimizedMSAC::EstimateModel and helpers            lib/RansacLib/RansacLib/ransac.h:127-428
//   UniformSampling / RandomShuffleAndResize / NumRequiredIterations     lib/RansacLib/RansacLib/{sampling,utils}.h
//
// MI355X mapping.  Every LO-MSAC hypothesis re-triangulates ALL N tracks before it can be scored
// (initializer.cc:273, sfm2d.cc:433).  In the planar-offset stage a model is just three numbers (t_y of
// cameras 1..3): the 4x3 triangulation matrix of a track does not depend on the hypothesis, so its
// pseudo-inverse is precomputed once per track (36 doubles/track) and the per-(hypothesis, track) work is a
// 3x4 mat-vec + 4 projections.  Kernel shape: ONE LANE PER HYPOTHESIS walking the tracks in index order —
// track records are wave-uniform (scalar loads), and the MSAC score sum_i min(err_i, thr) is accumulated in
// exactly the reference's order (ransac.h:291-299), so `score < best_score` comparisons see the same sums a
// sequential CPU loop would.  The LO-MSAC control flow stays on the host, speculating a chunk of iterations at
// a time (the sampler stream does not depend on results) and replaying the accept / LO / termination logic in
// iteration order.  Host RNG = this toolchain's <random>, exactly as the reference uses it.
#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstring>
#include <limits>
#include <numeric>
#include <random>
#include <vector>

#include "common.hpp"
#include "p6l_device.hpp"   // Solve3, Det3x3
#include "init_lsq.hpp"
#include "small_eigen.hpp"

namespace ppsfm {
// std::max(e1, std::max(e2, std::max(e3, e4))) exactly as the reference nests it (initializer.cc:332, sfm2d.cc:316; std::max(a, b) = a < b ? b : a).  Finite
// errors: the maximum, whatever the order.  With NaNs (the model of a degenerate sample, a NaN bearing) the nesting IS the result: a NaN model's error is NaN,
// which no threshold test counts as an inlier - a running fmax from 0 dropped the NaNs and scored such a model perfect (tools/fuzz_hostile_inputs.py).
__host__ __device__ inline double RefMax4(double e0, double e1, double e2, double e3) {
  double m = (e2 < e3) ? e3 : e2;
  m = (e1 < m) ? m : e1;
  return (e0 < m) ? m : e0;
}

constexpr int kRec = 36;   // doubles per track record: Minv 3x4 | a 4x3 | b0 4 | g 4 | invnorm 4

struct PlanarView {   // per-view constants
  double r3[4][3];    // third row of R_j = Rg_j^T P_j[:, :3]
  double c0[4], c1[4];   // z_j(2) = r3_j . X + c0_j + c1_j * ty_j
};

}  // namespace ppsfm

struct pp_planar_impl {
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  int32_t n = 0;
  double* rec = nullptr;       // n x kRec
  double* lines = nullptr;     // 4 x n x 3
  ppsfm::PlanarView view;
  double poses[48], Rg[36];
  // work buffers
  int64_t cap = 0;
  int32_t* samples = nullptr;
  double *offsets = nullptr, *scores = nullptr, *err = nullptr, *X = nullptr;
  int32_t* inl = nullptr;
  double *d_poses = nullptr, *d_Rg = nullptr;
};

struct pp_fourview2d_impl {
  int device = 0;
  hipStream_t stream = nullptr;
  int32_t n = 0;
  double* x = nullptr;   // 4 x n x 2 unit bearings
  int64_t cap = 0;
  double *cams = nullptr, *scores = nullptr, *err = nullptr, *X = nullptr;
  int32_t* inl = nullptr;
  int64_t hyp_cap = 0;     // minimal-solver batch buffers
  int32_t hyp_m = 0;
  int32_t *samples = nullptr, *counts = nullptr, *best_index = nullptr;
  double *models = nullptr, *mscores = nullptr, *best_cams = nullptr, *best_score = nullptr;
  int32_t* minl = nullptr;
  // LeastSquares / LO-MSAC: scratch of the two LM kernels, a ring of sample buffers, and the POOL of models that live on the device during a run (a refined
  // model carries its points, as the reference's Reconstruction does): slot s = 24 camera doubles (pool_cams + 24 s), its MSAC score once somebody asked
  // for it (pool_scores + s), n x 2 points (chunks of kPoolChunk slots).  The host keeps a model as 24 doubles + its slot number; cameras and scores of
  // refined models come back in ONE copy per local optimisation.
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  double *lsq_scale = nullptr, *lsq_Xc = nullptr, *xch = nullptr;
  int32_t* d_sample = nullptr;      // kSampleRing buffers of sample_cap ints
  int64_t sample_cap = 0;
  int sample_next = 0;
  int32_t* d_iterations = nullptr;  // iteration count of the last k_fv2d_points_wave (diagnostics)
  double *pool_cams = nullptr, *pool_scores = nullptr;
  int pool_cap = 0, pool_used = 0;
  std::vector<double*> pool_X;      // chunk c holds the points of slots [c * kPoolChunk, (c + 1) * kPoolChunk)
  std::vector<uint8_t> slot_refined, slot_has_X, slot_scored;
  int dev_err_slot = -1, host_err_slot = -1;      // whose errors h->err / the backend's host copy hold
  void* pinned = nullptr;           // staging: samples ring | errors (n doubles) | 32 doubles | ticket
  void* pinned_dev = nullptr;       // the same block as the device sees it
  unsigned long long ticket_seq = 0;
  int pending_score_slot = -1;      // a score whose kernel is launched behind the next shipment of errors (off the host's critical path)
};

struct pp_pose2d_impl {
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  int32_t n = 0;
  double *x = nullptr, *X = nullptr;   // n x 2 unit bearings, n x 2 points
  int64_t cap = 0;
  int32_t cap_m = 0;
  int32_t* samples = nullptr;
  double *poses = nullptr, *scores = nullptr, *err = nullptr;
  int32_t* inl = nullptr;
};

namespace ppsfm {

// MSAC score (ransac.h:291-299) + strict-< inlier count of ONE model by ONE wavefront: lane l takes the tracks l, l+64, ...
// in index order, then a xor butterfly.  A fixed order, reproduced on the host by TreeMsacScore; NOT the reference's
// sequential order (the two sums agree to ~1e-15 relative, which only matters between models whose scores tie at that
// level).  One lane per model walking all tracks would keep the sequential order but leaves the chip idle: a chunk of
// 1024 four-view samples is 16384 candidates x n tracks.
template <class ErrFn>
__device__ __forceinline__ void WaveMsac(int n, double thr, ErrFn err, double* __restrict__ score_out, int32_t* __restrict__ inl_out) {
  const int lane = threadIdx.x & 63;
  double score = 0.0;
  int cnt = 0;
  for (int i = lane; i < n; i += 64) {
    const double e = err(i);
    score += (thr < e) ? thr : e;      // std::min(e, thr) as ransac.h:302-305 writes it: a NaN error makes the score NaN, and no `score < best` accepts that model (fmin dropped it)
    cnt += (e < thr) ? 1 : 0;
  }
  score = WaveSum(score);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off, 64);
  if (lane == 0) { *score_out = score; *inl_out = cnt; }
}

// ---- planar offset: error of one (model, track) -----------------------------------------------------
__device__ __forceinline__ double PlanarTrackError(const double* __restrict__ r, const PlanarView& v, double ty1, double ty2, double ty3,
                                                   double X[3]) {
  const double ty[4] = {0.0, ty1, ty2, ty3};
  double b[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) b[j] = r[24 + j] - r[28 + j] * ty[j];
#pragma unroll
  for (int c = 0; c < 3; ++c) X[c] = r[4 * c] * b[0] + r[4 * c + 1] * b[1] + r[4 * c + 2] * b[2] + r[4 * c + 3] * b[3];
  double e[4];
  bool behind = false;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const double num = r[12 + 3 * j] * X[0] + r[12 + 3 * j + 1] * X[1] + r[12 + 3 * j + 2] * X[2] - b[j];
    const double den = v.r3[j][0] * X[0] + v.r3[j][1] * X[1] + v.r3[j][2] * X[2] + v.c0[j] + v.c1[j] * ty[j];
    behind = behind || (den < 0.0);
    e[j] = fabs(num / den) * r[32 + j];
  }
  return behind ? 100000.0 : RefMax4(e[0], e[1], e[2], e[3]);     // initializer.cc:318-320, :332
}

// one wavefront per model (WaveMsac)
__global__ __launch_bounds__(256) void k_planar_score(int n, const double* __restrict__ rec, PlanarView v, int num, const double* __restrict__ offsets,
                                                      double thr, double* __restrict__ scores, int32_t* __restrict__ inl) {
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= num) return;
  const double t1 = offsets[3 * m], t2 = offsets[3 * m + 1], t3 = offsets[3 * m + 2];
  WaveMsac(n, thr, [&](int i) { double X[3]; return PlanarTrackError(rec + (size_t)kRec * i, v, t1, t2, t3, X); }, scores + m, inl + m);
}

// one lane per track: errors + points of ONE model
__global__ __launch_bounds__(256) void k_planar_evaluate(int n, const double* __restrict__ rec, PlanarView v, double t1, double t2, double t3,
                                                         double* __restrict__ err, double* __restrict__ Xout) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double X[3];
  err[i] = PlanarTrackError(rec + (size_t)kRec * i, v, t1, t2, t3, X);
  if (Xout) { Xout[3 * i] = X[0]; Xout[3 * i + 1] = X[1]; Xout[3 * i + 2] = X[2]; }
}

// PlanarOffsetEstimator::MinimalSolver up to the offsets (initializer.cc:236-262), one lane per sample.
// sample_size == 3: direct 3x3 solve; larger samples (the LO non-minimal solver): least squares by normal equations.
__global__ __launch_bounds__(64) void k_planar_solve(int n, const double* __restrict__ lines, const double* __restrict__ poses,
                                                     const double* __restrict__ Rg, int64_t num, int sample_size, const int32_t* __restrict__ samples,
                                                     double* __restrict__ offsets) {
  const int64_t h = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (h >= num) return;
  double AtA[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, Atb[3] = {0, 0, 0}, Ad[9], bd[3];
  bool ok = true;
  for (int s = 0; s < sample_size; ++s) {
    const int idx = samples[h * sample_size + s];
    double A0[9], B0[12];
#pragma unroll
    for (int e = 0; e < 12; ++e) B0[e] = 0.0;
#pragma unroll
    for (int j = 1; j < 4; ++j) {
      const double* l = lines + ((size_t)j * n + idx) * 3;
      const double* R = Rg + 9 * j;
      const double* P = poses + 12 * j;
      double lg[3];
#pragma unroll
      for (int r = 0; r < 3; ++r) lg[r] = R[3 * r] * l[0] + R[3 * r + 1] * l[1] + R[3 * r + 2] * l[2];
#pragma unroll
      for (int c = 0; c < 3; ++c) A0[3 * (j - 1) + c] = lg[0] * P[c] + lg[1] * P[4 + c] + lg[2] * P[8 + c];
      B0[4 * (j - 1) + (j - 1)] = lg[1];
      B0[4 * (j - 1) + 3] = lg[0] * P[3] + lg[2] * P[11];
    }
    if (!Solve3<4>(A0, B0)) { ok = false; break; }          // B0 <- A0^-1 B0 (partial pivoting)
    const double* R0 = Rg;                                    // Rg_0^T B0
    double RB[12];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) RB[4 * r + c] = R0[r] * B0[c] + R0[3 + r] * B0[4 + c] + R0[6 + r] * B0[8 + c];
    const double* l0 = lines + (size_t)idx * 3;
    double row[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) row[c] = l0[0] * RB[c] + l0[1] * RB[4 + c] + l0[2] * RB[8 + c];
    const double bi = -(l0[0] * RB[3] + l0[1] * RB[7] + l0[2] * RB[11]);
    if (s < 3) {
#pragma unroll
      for (int c = 0; c < 3; ++c) Ad[3 * s + c] = row[c];
      bd[s] = bi;
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      Atb[r] += row[r] * bi;
#pragma unroll
      for (int c = 0; c < 3; ++c) AtA[3 * r + c] += row[r] * row[c];
    }
  }
  double out[3] = {NAN, NAN, NAN};
  if (ok) {
    if (sample_size == 3) { if (Solve3<1>(Ad, bd)) { out[0] = bd[0]; out[1] = bd[1]; out[2] = bd[2]; } }
    else { if (Solve3<1>(AtA, Atb)) { out[0] = Atb[0]; out[1] = Atb[1]; out[2] = Atb[2]; } }
  }
  offsets[3 * h] = out[0]; offsets[3 * h + 1] = out[1]; offsets[3 * h + 2] = out[2];
}

// ---- four-view 2D: triangulate from views 0..2 + 1D bearing error, one (model, track) ----------------------
__device__ __forceinline__ double FourView2dTrackError(const double* __restrict__ cams /*4x6*/, const double* __restrict__ x, int n, int i, double X[2]) {
  // normal equations of the 3x2 system of sfm2d.cc:194-213
  double a00 = 0, a01 = 0, a11 = 0, r0 = 0, r1 = 0;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const double xa = x[((size_t)j * n + i) * 2], xb = x[((size_t)j * n + i) * 2 + 1];
    const double* P = cams + 6 * j;
    const double A0 = xa * P[3] - xb * P[0], A1 = xa * P[4] - xb * P[1], b = xb * P[2] - xa * P[5];
    a00 += A0 * A0; a01 += A0 * A1; a11 += A1 * A1; r0 += A0 * b; r1 += A1 * b;
  }
  const double det = a00 * a11 - a01 * a01;
  X[0] = (a11 * r0 - a01 * r1) / det;
  X[1] = (a00 * r1 - a01 * r0) / det;
  double e[4];
  bool behind = false;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const double* P = cams + 6 * j;
    const double z0 = P[0] * X[0] + P[1] * X[1] + P[2], z1 = P[3] * X[0] + P[4] * X[1] + P[5];
    behind = behind || (z1 < 0.0);
    const double xa = x[((size_t)j * n + i) * 2], xb = x[((size_t)j * n + i) * 2 + 1];
    e[j] = fabs(xa / xb - z0 / z1);
  }
  return behind ? 1000000.0 : RefMax4(e[0], e[1], e[2], e[3]);    // sfm2d.cc:308-309, :316
}

__global__ __launch_bounds__(256) void k_fourview2d_score(int n, const double* __restrict__ x, int num, const double* __restrict__ cams, double thr,
                                                          double* __restrict__ scores, int32_t* __restrict__ inl) {
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= num) return;
  double c[24];
#pragma unroll
  for (int e = 0; e < 24; ++e) c[e] = cams[(size_t)m * 24 + e];
  WaveMsac(n, thr, [&](int i) { double X[2]; return FourView2dTrackError(c, x, n, i, X); }, scores + m, inl + m);
}

__global__ __launch_bounds__(256) void k_fourview2d_evaluate(int n, const double* __restrict__ x, const double* __restrict__ cams, double* __restrict__ err,
                                                             double* __restrict__ Xout) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double c[24];
#pragma unroll
  for (int e = 0; e < 24; ++e) c[e] = cams[e];
  double X[2];
  err[i] = FourView2dTrackError(c, x, n, i, X);
  if (Xout) { Xout[2 * i] = X[0]; Xout[2 * i + 1] = X[1]; }
}


// EvaluateModelOnPoint with the model's OWN points (a model refined by LeastSquares carries them, sfm2d.cc:302-319)
__global__ __launch_bounds__(256) void k_fourview2d_errors_stored(int n, const double* __restrict__ x, const double* __restrict__ cams, const double* __restrict__ X,
                                                                  double* __restrict__ err) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const double X0 = X[2 * (size_t)i], X1 = X[2 * (size_t)i + 1];
  double e[4];
  bool behind = false;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const double* P = cams + 6 * j;
    const double z0 = P[0] * X0 + P[1] * X1 + P[2], z1 = P[3] * X0 + P[4] * X1 + P[5];
    behind = behind || (z1 < 0.0);
    e[j] = fabs(x[((size_t)j * n + i) * 2] / x[((size_t)j * n + i) * 2 + 1] - z0 / z1);
  }
  err[i] = behind ? 1000000.0 : RefMax4(e[0], e[1], e[2], e[3]);
}

// ---- four-view 2D minimal solver, one lane per sample (sfm2d.cc:178-298, 363-444) ---------------------------
struct TrifocalFrames { double A1[4], A2[4], A3[4]; };   // the three 2x2 coordinate changes of sfm2d.cc:231-235

__device__ __forceinline__ void Triangulate3(const double* __restrict__ c /*3x6*/, const double* __restrict__ x, int n, int i, double X[2]) {
  double a00 = 0, a01 = 0, a11 = 0, r0 = 0, r1 = 0;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const double xa = x[((size_t)j * n + i) * 2], xb = x[((size_t)j * n + i) * 2 + 1];
    const double* P = c + 6 * j;
    const double A0 = xa * P[3] - xb * P[0], A1 = xa * P[4] - xb * P[1], b = xb * P[2] - xa * P[5];
    a00 += A0 * A0; a01 += A0 * A1; a11 += A1 * A1; r0 += A0 * b; r1 += A1 * b;
  }
  const double det = a00 * a11 - a01 * a01;
  X[0] = (a11 * r0 - a01 * r1) / det;
  X[1] = (a00 * r1 - a01 * r0) / det;
}

// models: num x 16 x 24 (camera-major 2x3 row-major); counts: 0 (negative discriminant) or 16
// kLanes = 1: one lane per sample walks its 2 x 8 candidates (a RANSAC chunk: a thousand samples side by side).  kLanes = 16: one lane per CANDIDATE - the
// sixteen lanes of a sample repeat the tensor and its factorisation (free in SIMD) and each completes ONE of the 2 roots x 8 sign choices: the same
// arithmetic per candidate, the same bits, a sixteenth of the flips loop on the critical path - for the one-sample calls of LO-MSAC's NonMinimalSolver
// (ransac.h:337-406 calls it num_lo_steps times per local optimisation, each on the critical path of the host's replay).
template <int kLanes>
__global__ __launch_bounds__(64) void k_fourview2d_minimal(int n, const double* __restrict__ x, int64_t num, int m, const int32_t* __restrict__ samples,
                                                           TrifocalFrames fr, double* __restrict__ models, int32_t* __restrict__ counts) {
  const int64_t gid = (int64_t)blockIdx.x * 64 + threadIdx.x;
  const int64_t h = gid / kLanes;
  const int cand = (int)(gid % kLanes);
  if (h >= num) return;
  const int f_begin = kLanes == 16 ? (cand >> 3) : 0, f_end = kLanes == 16 ? f_begin + 1 : 2;
  const int flips_begin = kLanes == 16 ? (cand & 7) : 0, flips_end = kLanes == 16 ? flips_begin + 1 : 8;
  const int32_t* smp = samples + h * m;
  // 1. trifocal tensor: null vector of the m x 6 incidence system (sfm2d.cc:364-379)
  double S[36];
#pragma unroll
  for (int e = 0; e < 36; ++e) S[e] = 0.0;
#pragma unroll 1
  for (int i = 0; i < m; ++i) {
    const int s = smp[i];
    const double a0 = x[(size_t)s * 2], a1 = x[(size_t)s * 2 + 1];
    const double b0 = x[((size_t)n + s) * 2], b1 = x[((size_t)n + s) * 2 + 1];
    const double c0 = x[((size_t)2 * n + s) * 2], c1 = x[((size_t)2 * n + s) * 2 + 1];
    double mono[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) mono[e] = ((e & 1) ? a1 : a0) * ((e & 2) ? b1 : b0) * ((e & 4) ? c1 : c0);
    double row[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) row[k] = mono[k + 2];
    row[1] += mono[0]; row[3] += mono[0]; row[4] += mono[0];
    row[5] += mono[1]; row[0] -= mono[1]; row[2] -= mono[1];
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
      for (int c = 0; c < 6; ++c) S[r * 6 + c] += row[r] * row[c];
  }
  double t[6];
  SmallestEigenvector<6>(S, t);
  double T[8];
  T[0] = t[1] + t[3] + t[4]; T[1] = -t[2] - t[0] + t[5];
#pragma unroll
  for (int k = 0; k < 6; ++k) T[k + 2] = t[k];
  // 2. change of coordinates, quadratic for the second camera's first column (sfm2d.cc:215-249)
  double AT[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int ap = e & 1, bp = (e >> 1) & 1, cp = e >> 2;
    double acc = 0.0;
#pragma unroll
    for (int f = 0; f < 8; ++f) {
      const int a = f & 1, b = (f >> 1) & 1, c = f >> 2;
      acc += fr.A1[2 * a + ap] * fr.A2[2 * b + bp] * fr.A3[2 * c + cp] * T[f];
    }
    AT[e] = acc;
  }
  const double alpha = AT[2] * AT[7] - AT[3] * AT[6];
  const double beta = AT[1] * AT[6] + AT[3] * AT[4] - AT[0] * AT[7] - AT[2] * AT[5];
  const double gamma = AT[0] * AT[5] - AT[1] * AT[4];
  const double disc = beta * beta - 4.0 * alpha * gamma;
  double* out = models + h * 16 * 24;
  if (disc < 0.0) { if (cand == 0) counts[h] = 0; return; }
  const double sq = sqrt(disc);
  const double aa0 = (beta > 0.0) ? (2.0 * gamma) / (-beta - sq) : (2.0 * gamma) / (-beta + sq);
  const double aa1 = gamma / (alpha * aa0);
  const double idet = 1.0 / (fr.A1[0] * fr.A1[3] - fr.A1[1] * fr.A1[2]);
  const double A1i[4] = {fr.A1[3] * idet, -fr.A1[1] * idet, -fr.A1[2] * idet, fr.A1[0] * idet};
#pragma unroll 1
  for (int f = f_begin; f < f_end; ++f) {
    double a1 = f == 0 ? aa0 : aa1;
    const double sn = sqrt(1.0 + a1 * a1);
    a1 /= sn;
    const double a2 = 1.0 / sn;
    const double rho = -(AT[1] * a2 - AT[3] * a1) / (AT[2] * a1 - AT[0] * a2);
    const double b1 = rho * a1, b2 = rho * a2, c1 = -a2, c2 = a1;
    // 3. third camera: null vector of the 7 x 6 system (sfm2d.cc:263-273), Gram matrix accumulated row by row
    double G[36];
#pragma unroll
    for (int e = 0; e < 36; ++e) G[e] = 0.0;
    auto add_row = [&](const double (&g)[6]) {
#pragma unroll
      for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int c = 0; c < 6; ++c) G[r * 6 + c] += g[r] * g[c];
    };
    { const double g[6] = {0, AT[7] * c2, -AT[0] * c1, 0, AT[0] * b1, -AT[7] * a2}; add_row(g); }
    { const double g[6] = {0, 0, -AT[1] * c1, AT[7] * c2, AT[1] * b1, -AT[7] * b2}; add_row(g); }
    { const double g[6] = {0, -AT[7] * c1, -AT[2] * c1, 0, AT[2] * b1, AT[7] * a1}; add_row(g); }
    { const double g[6] = {0, 0, -AT[3] * c1, -AT[7] * c1, AT[3] * b1, AT[7] * b1}; add_row(g); }
    { const double g[6] = {-AT[7] * c2, 0, -AT[4] * c1, 0, AT[7] * a2 + AT[4] * b1, 0}; add_row(g); }
    { const double g[6] = {0, 0, -AT[5] * c1 - AT[7] * c2, 0, AT[7] * b2 + AT[5] * b1, 0}; add_row(g); }
    { const double g[6] = {AT[7] * c1, 0, -AT[6] * c1, 0, -AT[7] * a1 + AT[6] * b1, 0}; add_row(g); }
    double d[6];
    SmallestEigenvector<6>(G, d);
    // 4. back to the image coordinates, metric upgrade, normalisation (sfm2d.cc:178-191, 275-296, 381-403)
    double P[2][6];
    {
      const double Q2[6] = {a1, b1, c1, a2, b2, c2};
      const double Q3[6] = {d[0], d[2], d[4], d[1], d[3], d[5]};
#pragma unroll
      for (int v = 0; v < 2; ++v) {
        const double* A = v == 0 ? fr.A2 : fr.A3;
        const double* Q = v == 0 ? Q2 : Q3;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          const double m0 = A[2 * r] * Q[0] + A[2 * r + 1] * Q[3], m1 = A[2 * r] * Q[1] + A[2 * r + 1] * Q[4], m2 = A[2 * r] * Q[2] + A[2 * r + 1] * Q[5];
          P[v][3 * r] = m0 * A1i[0] + m1 * A1i[2];
          P[v][3 * r + 1] = m0 * A1i[1] + m1 * A1i[3];
          P[v][3 * r + 2] = m2;
        }
      }
    }
    {
      // least squares of the 4 x 2 system by its normal equations
      const double r0[4] = {P[0][2], P[0][5], P[1][2], P[1][5]};
      const double r1[4] = {-P[0][5], P[0][2], -P[1][5], P[1][2]};
      const double bb[4] = {P[0][4] - P[0][0], -P[0][1] - P[0][3], P[1][4] - P[1][0], -P[1][1] - P[1][3]};
      double n00 = 0, n01 = 0, n11 = 0, q0 = 0, q1 = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) { n00 += r0[k] * r0[k]; n01 += r0[k] * r1[k]; n11 += r1[k] * r1[k]; q0 += r0[k] * bb[k]; q1 += r1[k] * bb[k]; }
      const double det = n00 * n11 - n01 * n01;
      const double h0 = (n11 * q0 - n01 * q1) / det, h1 = (n00 * q1 - n01 * q0) / det;
#pragma unroll
      for (int v = 0; v < 2; ++v)
#pragma unroll
        for (int r = 0; r < 2; ++r) { P[v][3 * r] += P[v][3 * r + 2] * h0; P[v][3 * r + 1] += P[v][3 * r + 2] * h1; }
#pragma unroll
      for (int v = 0; v < 2; ++v) {
        const double nr = sqrt(P[v][0] * P[v][0] + P[v][3] * P[v][3]);
#pragma unroll
        for (int e = 0; e < 6; ++e) P[v][e] /= nr;
      }
      const double sc = sqrt(P[0][2] * P[0][2] + P[0][5] * P[0][5]);
      P[0][2] /= sc; P[0][5] /= sc; P[1][2] /= sc; P[1][5] /= sc;
      // the (numerically idle) second normalisation of sfm2d.cc:412-413
      const double nt = sqrt(P[0][2] * P[0][2] + P[0][5] * P[0][5]);
      P[1][2] /= nt; P[1][5] /= nt;
      const double nt2 = sqrt(P[0][2] * P[0][2] + P[0][5] * P[0][5]);
      P[0][2] /= nt2; P[0][5] /= nt2;
    }
    // 5. the eight sign choices + fourth camera from the sample (sfm2d.cc:405-437, 321-361)
#pragma unroll 1
    for (int flips = flips_begin; flips < flips_end; ++flips) {
      const bool f1 = flips & 4, f2 = flips & 2, f3 = flips & 1;    // flip1 outermost, as the reference nests them
      double c[24];
      c[0] = 1; c[1] = 0; c[2] = 0; c[3] = 0; c[4] = 1; c[5] = 0;
#pragma unroll
      for (int e = 0; e < 6; ++e) { c[6 + e] = P[0][e]; c[12 + e] = P[1][e]; }
      if (f1) { c[8] = -c[8]; c[11] = -c[11]; c[14] = -c[14]; c[17] = -c[17]; }
      if (f2) {
#pragma unroll
        for (int e = 0; e < 6; ++e) c[6 + e] = -c[6 + e];
      }
      if (f3) {
#pragma unroll
        for (int e = 0; e < 6; ++e) c[12 + e] = -c[12 + e];
      }
      // AbsPoseSolver on (x4, triangulated sample points)
      double btb0 = 0, btb1 = 0, btb3 = 0, bta0 = 0, bta1 = 0, bta2 = 0, bta3 = 0;
#pragma unroll 1
      for (int i = 0; i < m; ++i) {
        const int s = smp[i];
        double X[2];
        Triangulate3(c, x, n, s, X);
        const double u = x[((size_t)3 * n + s) * 2], w = x[((size_t)3 * n + s) * 2 + 1];
        const double A0 = X[0] * w - X[1] * u, A1 = -X[0] * u - X[1] * w, B0 = w, B1 = -u;
        btb0 += B0 * B0; btb1 += B0 * B1; btb3 += B1 * B1;
        bta0 += B0 * A0; bta1 += B0 * A1; bta2 += B1 * A0; bta3 += B1 * A1;
      }
      const double det = btb0 * btb3 - btb1 * btb1;
      const double i0 = btb3 / det, i1 = -btb1 / det, i3 = btb0 / det;
      const double C0 = -(i0 * bta0 + i1 * bta2), C1 = -(i0 * bta1 + i1 * bta3), C2 = -(i1 * bta0 + i3 * bta2), C3 = -(i1 * bta1 + i3 * bta3);
      double M[4] = {0, 0, 0, 0};
      double X0[2] = {0, 0};
#pragma unroll 1
      for (int i = 0; i < m; ++i) {
        const int s = smp[i];
        double X[2];
        Triangulate3(c, x, n, s, X);
        if (i == 0) { X0[0] = X[0]; X0[1] = X[1]; }
        const double u = x[((size_t)3 * n + s) * 2], w = x[((size_t)3 * n + s) * 2 + 1];
        const double A0 = X[0] * w - X[1] * u, A1 = -X[0] * u - X[1] * w, B0 = w, B1 = -u;
        const double r0 = A0 + B0 * C0 + B1 * C2, r1 = A1 + B0 * C1 + B1 * C3;
        M[0] += r0 * r0; M[1] += r0 * r1; M[2] += r0 * r1; M[3] += r1 * r1;
      }
      double ab[2];
      SmallestEigenvector<2>(M, ab);
      const double nr = sqrt(ab[0] * ab[0] + ab[1] * ab[1]);
      ab[0] /= nr; ab[1] /= nr;
      c[18] = ab[0]; c[19] = -ab[1]; c[20] = C0 * ab[0] + C1 * ab[1];
      c[21] = ab[1]; c[22] = ab[0]; c[23] = C2 * ab[0] + C3 * ab[1];
      if (c[21] * X0[0] + c[22] * X0[1] + c[23] < 0.0) {
#pragma unroll
        for (int e = 18; e < 24; ++e) c[e] = -c[e];
      }
      double* o = out + (size_t)(f * 8 + flips) * 24;
#pragma unroll
      for (int e = 0; e < 24; ++e) o[e] = c[e];
    }
  }
  if (cand == 0) counts[h] = 16;
}

// MSAC score of ONE model from its error array (k_fourview2d_evaluate / _errors_stored), in WaveMsac's order (= the host's TreeMsacScore): the score of a
// model the LO-MSAC replay keeps on the device never travels as n doubles - one double per model comes back when the local optimisation is over.
__global__ __launch_bounds__(64) void k_fv2d_score_errors(int n, const double* __restrict__ err, double thr, double* __restrict__ score_out, int32_t* __restrict__ inl_out) {
  int32_t dummy;
  WaveMsac(n, thr, [&](int i) { return err[i]; }, score_out, inl_out ? inl_out : &dummy);
}

// The errors of a model (+ a few doubles that ride along) straight into pinned host memory, then a ticket the host polls: what the replay waits for per
// LeastSquaresFit is this one workgroup, not a copy-engine transfer plus a stream synchronisation (as the LM loop's scalars in ba_solver.hip, WaitTicket).
__global__ __launch_bounds__(1024) void k_fv2d_ship(int n, const double* __restrict__ err, const double* __restrict__ extra, int extra_n, double* host_dst,
                                                    unsigned long long* host_ticket, unsigned long long ticket) {
  for (int i = threadIdx.x; i < n; i += 1024) __builtin_nontemporal_store(err[i], host_dst + i);
  if ((int)threadIdx.x < extra_n) __builtin_nontemporal_store(extra[threadIdx.x], host_dst + n + threadIdx.x);
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(host_ticket, ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// NonMinimalSolver's winner (k_fourview2d_select of ONE sample) becomes a pooled model: cameras and score copied to its slot
__global__ __launch_bounds__(64) void k_fv2d_adopt(const double* __restrict__ best_cams, const double* __restrict__ best_score, double* __restrict__ slot_cams, double* __restrict__ slot_score) {
  if (threadIdx.x < 24) slot_cams[threadIdx.x] = best_cams[threadIdx.x];
  if (threadIdx.x == 24) *slot_score = *best_score;
}

// NonMinimalSolver's selection (sfm2d.cc:446-467): first strictly-smallest MSAC score among the sample's models
__global__ __launch_bounds__(64) void k_fourview2d_select(int64_t num, const int32_t* __restrict__ counts, const double* __restrict__ scores /*num x 16*/,
                                                          const double* __restrict__ models, double* __restrict__ best_cams, double* __restrict__ best_score,
                                                          int32_t* __restrict__ best_index) {
  const int64_t h = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (h >= num) return;
  double best = 1.7976931348623157e308;
  int bi = -1;
  for (int k = 0; k < counts[h]; ++k) { const double s = scores[h * 16 + k]; if (s < best) { best = s; bi = k; } }
  best_score[h] = best;
  best_index[h] = bi;
  for (int e = 0; e < 24; ++e) best_cams[h * 24 + e] = bi >= 0 ? models[(h * 16 + bi) * 24 + e] : NAN;
}

// ---- AbsolutePose2dEstimator (sfm2d.cc:491-530; used by the reference's tests) -----------------------------------
// NonMinimalSolver: null vector of the m x 4 system [X1 x2 - X2 x1, -X1 x1 - X2 x2, x2, -x1] -> similarity-free pose
// [a -b t0; b a t1] with a^2 + b^2 = 1, sign such that the first sample point is in front.  One lane per sample.
__global__ __launch_bounds__(64) void k_pose2d_solve(int n, const double* __restrict__ x, const double* __restrict__ X, int64_t num, int m,
                                                     const int32_t* __restrict__ samples, double* __restrict__ poses) {
  const int64_t h = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (h >= num) return;
  const int32_t* smp = samples + h * m;
  double S[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) S[e] = 0.0;
#pragma unroll 1
  for (int i = 0; i < m; ++i) {
    const int s = smp[i];
    const double x1 = x[2 * (size_t)s], x2 = x[2 * (size_t)s + 1], X1 = X[2 * (size_t)s], X2 = X[2 * (size_t)s + 1];
    const double row[4] = {X1 * x2 - X2 * x1, -X1 * x1 - X2 * x2, x2, -x1};
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) S[4 * r + c] += row[r] * row[c];
  }
  double t[4];
  SmallestEigenvector<4>(S, t);
  const double nr = sqrt(t[0] * t[0] + t[1] * t[1]);
#pragma unroll
  for (int e = 0; e < 4; ++e) t[e] /= nr;
  double M[6] = {t[0], -t[1], t[2], t[1], t[0], t[3]};
  const int s0 = smp[0];
  if (M[3] * X[2 * (size_t)s0] + M[4] * X[2 * (size_t)s0 + 1] + M[5] < 0.0) {
#pragma unroll
    for (int e = 0; e < 6; ++e) M[e] = -M[e];
  }
#pragma unroll
  for (int e = 0; e < 6; ++e) poses[h * 6 + e] = M[e];
}

// EvaluateModelOnPoint: 1 - x . normalize(P X~)
__device__ __forceinline__ double Pose2dError(const double* __restrict__ P, const double* __restrict__ x, const double* __restrict__ X, int i) {
  const double X1 = X[2 * (size_t)i], X2 = X[2 * (size_t)i + 1];
  const double z0 = P[0] * X1 + P[1] * X2 + P[2], z1 = P[3] * X1 + P[4] * X2 + P[5];
  const double nr = sqrt(z0 * z0 + z1 * z1);
  return 1.0 - (x[2 * (size_t)i] * (z0 / nr) + x[2 * (size_t)i + 1] * (z1 / nr));
}
__global__ __launch_bounds__(256) void k_pose2d_score(int n, const double* __restrict__ x, const double* __restrict__ X, int num, const double* __restrict__ poses,
                                                      double thr, double* __restrict__ scores, int32_t* __restrict__ inl) {
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= num) return;
  double P[6];
#pragma unroll
  for (int e = 0; e < 6; ++e) P[e] = poses[(size_t)m * 6 + e];
  WaveMsac(n, thr, [&](int i) { return Pose2dError(P, x, X, i); }, scores + m, inl + m);
}
__global__ __launch_bounds__(256) void k_pose2d_evaluate(int n, const double* __restrict__ x, const double* __restrict__ X, const double* __restrict__ pose,
                                                         double* __restrict__ err) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double P[6];
#pragma unroll
  for (int e = 0; e < 6; ++e) P[e] = pose[e];
  err[i] = Pose2dError(P, x, X, i);
}

// ---- host helpers -------------------------------------------------------------------------------------
// the MSAC sum in WaveMsac's order, so that a model scored on the host (ScoreModel, from downloaded errors) and in a batch
// on the device gets the same bits
static double TreeMsacScore(const double* err, int n, double thr) {
  double acc[64];
  for (int l = 0; l < 64; ++l) acc[l] = 0.0;
  for (int i = 0; i < n; ++i) acc[i & 63] += std::min(err[i], thr);   // ransac.h:296 (a NaN error poisons the score)
  for (int off = 32; off > 0; off >>= 1) {
    double nxt[64];
    for (int l = 0; l < 64; ++l) nxt[l] = acc[l] + acc[l ^ off];
    for (int l = 0; l < 64; ++l) acc[l] = nxt[l];
  }
  return acc[0];
}

static int PlanarEnsure(pp_planar_impl* h, int64_t cap) {
  if (cap <= h->cap) return PP_OK;
  void* old[] = {h->samples, h->offsets, h->scores, h->inl};
  for (void* p : old) if (p) (void)hipFree(p);
  h->samples = nullptr; h->offsets = nullptr; h->scores = nullptr; h->inl = nullptr; h->cap = 0;
  int rc;
  if ((rc = DeviceAlloc(&h->samples, (size_t)cap * 32))) return rc;
  if ((rc = DeviceAlloc(&h->offsets, (size_t)cap * 3))) return rc;
  if ((rc = DeviceAlloc(&h->scores, (size_t)cap))) return rc;
  if ((rc = DeviceAlloc(&h->inl, (size_t)cap))) return rc;
  h->cap = cap;
  return PP_OK;
}

static void CamsFromOffsets(const pp_planar_impl* h, const double* tt, double* cams) {
  for (int j = 0; j < 4; ++j) {
    double p[12];
    std::memcpy(p, h->poses + 12 * j, sizeof(p));
    if (j > 0) p[7] = tt[j - 1];
    const double* R = h->Rg + 9 * j;
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 4; ++c) cams[12 * j + 4 * r + c] = R[r] * p[c] + R[3 + r] * p[4 + c] + R[6 + r] * p[8 + c];   // Rg^T * pose
  }
}

// LO-MSAC over the planar-offset solver: models are offset triples; all scoring happens on the device
struct PlanarBackend {
  static constexpr bool kDeferredScores = false;
  static constexpr int kDim = 3, kMinSample = 3, kNonMinSample = 20;   // initializer.h: min_sample_size / non_minimal_sample_size
  pp_planar_impl* h;
  double thr;
  std::vector<double> err;
  int rc = PP_OK;
  int n() const { return h->n; }
  void LeastSquares(const std::vector<int>&, double*) {}   // PlanarOffsetEstimator::LeastSquares returns immediately (initializer.cc:450-451)
  int BatchSolveScore(uint32_t want, const int32_t* samples, std::vector<double>* models, std::vector<double>* scores, double* dev_s) {
    int r = PlanarEnsure(h, want); if (r) return r;
    r = Upload(h->samples, samples, (size_t)want * 3, h->stream); if (r) return r;
    PP_HIP_TRY(hipEventRecord(h->ev0, h->stream));
    hipLaunchKernelGGL(k_planar_solve, dim3(CeilDiv(want, 64)), dim3(64), 0, h->stream, h->n, h->lines, h->d_poses, h->d_Rg, (int64_t)want, 3, h->samples, h->offsets);
    hipLaunchKernelGGL(k_planar_score, dim3(CeilDiv(want, 4)), dim3(256), 0, h->stream, h->n, h->rec, h->view, (int)want, h->offsets, thr, h->scores, h->inl);
    PP_HIP_TRY(hipGetLastError());
    PP_HIP_TRY(hipEventRecord(h->ev1, h->stream));
    models->resize((size_t)want * 3); scores->resize(want);
    r = Download(models->data(), h->offsets, models->size(), h->stream); if (r) return r;
    r = Download(scores->data(), h->scores, scores->size(), h->stream); if (r) return r;
    PP_HIP_TRY(hipStreamSynchronize(h->stream));
    float ms = 0; PP_HIP_TRY(hipEventElapsedTime(&ms, h->ev0, h->ev1)); *dev_s += ms * 1e-3;
    return PP_OK;
  }
  double cached[3]; bool have_cached = false;      // (as Pose2dBackend: ScoreModel and the GetInliers behind it ask about the same model)
  int Evaluate(const double* model) {   // fills err (n)
    if (have_cached && std::memcmp(cached, model, sizeof(cached)) == 0 && (int)err.size() == h->n) return PP_OK;
    have_cached = false;
    err.resize(h->n);
    hipLaunchKernelGGL(k_planar_evaluate, dim3(CeilDiv(h->n, 256)), dim3(256), 0, h->stream, h->n, h->rec, h->view, model[0], model[1], model[2],
                       h->err, (double*)nullptr);
    if (hipGetLastError() != hipSuccess) return PP_ERR_HIP;
    if (hipMemcpyAsync(err.data(), h->err, sizeof(double) * h->n, hipMemcpyDeviceToHost, h->stream) != hipSuccess) return PP_ERR_HIP;
    if (hipStreamSynchronize(h->stream) != hipSuccess) return PP_ERR_HIP;
    std::memcpy(cached, model, sizeof(cached)); have_cached = true;
    return PP_OK;
  }
  double ScoreModel(const double* model) {
    if ((rc = Evaluate(model))) return std::numeric_limits<double>::max();
    return TreeMsacScore(err.data(), h->n, thr);
  }
  int GetInliers(const double* model, double t, std::vector<int>* inl) {
    if ((rc = Evaluate(model))) return 0;
    inl->clear();
    for (int i = 0; i < h->n; ++i) if (err[i] < t) inl->push_back(i);
    return (int)inl->size();
  }
  bool Solve(const std::vector<int>& sample, double* model) {
    const int m = (int)sample.size();
    if (m > 32) return false;
    if ((rc = PlanarEnsure(h, 64))) return false;
    if (hipMemcpyAsync(h->samples, sample.data(), sizeof(int32_t) * m, hipMemcpyHostToDevice, h->stream) != hipSuccess) { rc = PP_ERR_HIP; return false; }
    hipLaunchKernelGGL(k_planar_solve, dim3(1), dim3(64), 0, h->stream, h->n, h->lines, h->d_poses, h->d_Rg, (int64_t)1, m, h->samples, h->offsets);
    if (hipMemcpyAsync(model, h->offsets, sizeof(double) * 3, hipMemcpyDeviceToHost, h->stream) != hipSuccess) { rc = PP_ERR_HIP; return false; }
    if (hipStreamSynchronize(h->stream) != hipSuccess) { rc = PP_ERR_HIP; return false; }
    return std::isfinite(model[0]) && std::isfinite(model[1]) && std::isfinite(model[2]);
  }
};

static uint32_t NumRequiredIterations(double inlier_ratio, double prob_missing, int sample_size, uint32_t min_it, uint32_t max_it) {
  if (inlier_ratio <= 0.0) return max_it;        // utils.h:110-132
  if (inlier_ratio >= 1.0) return min_it;
  const double p = 1.0 - std::pow(inlier_ratio, static_cast<double>(sample_size));
  const double it = std::ceil(std::log(prob_missing) / std::log(p) + 0.5);
  return std::max(min_it, std::min(static_cast<uint32_t>(it), max_it));
}
static void RandomShuffle(std::mt19937* rng, std::vector<int>* v) {   // utils.h:48-58
  const int n = static_cast<int>(v->size());
  for (int i = 0; i < n - 1; ++i) { std::uniform_int_distribution<int> d(i, n - 1); std::swap((*v)[i], (*v)[d(*rng)]); }
}

class UniformSampling {   // sampling.h:46-135
 public:
  UniformSampling(unsigned seed, int num_data, int sample_size) : n_(num_data), k_(sample_size) {
    rng_.seed(seed);
    draw_ = static_cast<double>(num_data) / static_cast<double>(num_data - sample_size) < M_E;
    dist_.param(std::uniform_int_distribution<int>::param_type(0, n_ - 1));
  }
  void Sample(int* out) {
    if (draw_) {
      for (int i = 0; i < k_; ++i) {
        bool found = true;
        while (found) { found = false; out[i] = dist_(rng_); for (int j = 0; j < i; ++j) if (out[j] == out[i]) { found = true; break; } }
      }
    } else {
      std::vector<int> v(n_);
      std::iota(v.begin(), v.end(), 0);
      if (k_ != n_) RandomShuffle(&rng_, &v);
      for (int i = 0; i < k_; ++i) out[i] = v[i];
    }
  }
 private:
  std::mt19937 rng_; std::uniform_int_distribution<int> dist_; int n_, k_; bool draw_;
};

typedef std::array<double, 3> Offsets;

// LocalOptimization (ransac.h:337-406) over a device backend.  Backend: kDim doubles per model, kMinSample,
// kNonMinSample, n(), ScoreModel, GetInliers, Solve (NonMinimalSolver), LeastSquares, BatchSolveScore, rc.
template <class Backend>
static void LocalOptimization(const pp_lomsac_options& o, Backend& be, std::array<double, Backend::kDim>* best_min, double* score_best) {
  typedef std::array<double, Backend::kDim> Model;
  const int kN = be.n(), kMinNonMin = Backend::kNonMinSample, kMin = Backend::kMinSample;
  if (kMinNonMin > kN) return;
  const double thr = o.squared_inlier_threshold, mult = o.threshold_multiplier;
  std::mt19937 rng; rng.seed(o.random_seed);
  // ScoreModel + UpdateBestModel (ransac.h:399-404).  Nothing inside a local optimisation READS the best score or model - the loop's control flow depends on
  // inlier lists and on whether the non-minimal solver found a model -, so a backend whose models live on the device (kDeferredScores) only enqueues the
  // score here and the candidates are compared, in the order they were produced and with the same strict <, when the local optimisation is over.
  struct Cand { double score; Model m; int ticket; };
  std::vector<Cand> cand;
  auto consider = [&](Model& m) {
    if constexpr (Backend::kDeferredScores) { const int t = be.ScoreModelDeferred(m.data()); cand.push_back(Cand{0.0, m, t}); }
    else { const double sc = be.ScoreModel(m.data()); if (sc < *score_best) { *score_best = sc; *best_min = m; } }
  };
  auto lsq_fit = [&](double thresh, Model* m) {   // LeastSquaresFit: the rng draws happen even where LeastSquares is a no-op
    const int kSize = o.min_sample_multiplicator * kMin;
    std::vector<int> inl;
    const int ni = be.GetInliers(m->data(), thresh, &inl);
    if (ni < kMin) return;
    RandomShuffle(&rng, &inl);
    inl.resize(std::min(kSize, ni));
    be.LeastSquares(inl, m->data());
  };
  Model m_init = *best_min;
  lsq_fit(thr * mult, &m_init);
  consider(m_init);
  std::vector<int> base;
  be.GetInliers(m_init.data(), thr, &base);
  const int kNonMin = std::max(kMinNonMin, std::min(kMin * o.non_min_sample_multiplier, static_cast<int>(base.size()) / 2));
  for (int r = 0; r < o.num_lo_steps; ++r) {
    std::vector<int> sample = base;
    RandomShuffle(&rng, &sample);
    sample.resize(kNonMin);     // vector::resize value-initialises missing entries, as RandomShuffleAndResize does
    Model m_non_min;
    if (!be.Solve(sample, m_non_min.data())) continue;
    consider(m_non_min);
    lsq_fit(thr, &m_non_min);
    double thresh = mult * thr;
    const double upd = (mult - 1.0) * thr / static_cast<int>(o.num_lsq_iterations - 1);
    for (int i = 0; i < o.num_lsq_iterations; ++i) {
      lsq_fit(thresh, &m_non_min);
      consider(m_non_min);
      thresh -= upd;
    }
  }
  if constexpr (Backend::kDeferredScores) {
    be.ResolveScores(&cand);
    for (const Cand& c : cand) if (c.score < *score_best) { *score_best = c.score; *best_min = c.m; }
  }
}

// LocallyOptimizedMSAC::EstimateModel (ransac.h:127-271): the minimal solves + scores of a chunk of iterations run on the
// device in one batch, the bookkeeping is replayed on the host in iteration order
template <class Backend>
static int LoMsacRun(const pp_lomsac_options* o, Backend& be, pp_lomsac_report* rep, std::array<double, Backend::kDim>* best_out, std::vector<int>* inliers_out) {
  typedef std::array<double, Backend::kDim> Model;
  const auto t0 = std::chrono::steady_clock::now();
  std::memset(rep, 0, sizeof(*rep));
  rep->best_model_score = std::numeric_limits<double>::max();
  const int kMin = Backend::kMinSample, kN = be.n();
  Model best_model; best_model.fill(0.0);
  Model best_min = best_model;
  std::vector<int>& inliers = *inliers_out;
  inliers.clear();
  if (kMin > kN) { *best_out = best_model; return PP_OK; }
  const double thr = o->squared_inlier_threshold;
  const double kMax = std::numeric_limits<double>::max();
  UniformSampling sampler(o->random_seed, kN, kMin);
  uint32_t max_it = std::max(o->max_num_iterations, o->min_num_iterations);
  double best_min_score = kMax;
  auto refresh = [&]() {
    rep->best_num_inliers = be.GetInliers(best_model.data(), thr, &inliers);
    rep->inlier_ratio = static_cast<double>(rep->best_num_inliers) / static_cast<double>(kN);
    max_it = NumRequiredIterations(rep->inlier_ratio, 1.0 - o->success_probability, kMin, o->min_num_iterations, o->max_num_iterations);
  };
  auto update_best = [&](double sc, const Model& m) { if (sc < rep->best_model_score) { rep->best_model_score = sc; best_model = m; } };
  const uint32_t chunk = o->chunk_iterations ? o->chunk_iterations : 1024;
  std::vector<int32_t> hs; std::vector<double> models, sc;
  uint32_t it = 0;
  double dev_s = 0;
  while (it < max_it) {
    const uint32_t want = std::min<uint32_t>(chunk, max_it - it);
    hs.resize((size_t)want * kMin);
    for (uint32_t i = 0; i < want; ++i) sampler.Sample(&hs[(size_t)kMin * i]);
    const int rc = be.BatchSolveScore(want, hs.data(), &models, &sc, &dev_s);
    if (rc) return rc;
    rep->hypotheses_evaluated += want;
    // replay of ransac.h:155-237 in iteration order; the sampler has already been advanced for the whole
    // chunk, which is harmless because nothing after an early exit draws from it
    for (uint32_t i = 0; i < want && it < max_it; ++i, ++it) {
      if (it == o->lo_starting_iterations && best_min_score < kMax) {
        ++rep->number_lo_iterations;
        LocalOptimization(*o, be, &best_model, &rep->best_model_score);
        refresh();
      }
      Model m;
      bool finite = true;
      for (int k = 0; k < Backend::kDim; ++k) { m[k] = models[(size_t)Backend::kDim * i + k]; finite = finite && std::isfinite(m[k]); }
      if (!finite) continue;   // MinimalSolver returned 0 models
      const double best_local = sc[i];
      if (best_local < best_min_score || it == o->lo_starting_iterations) {
        const bool kBestMin = best_local < best_min_score;
        if (kBestMin) { best_min_score = best_local; best_min = m; update_best(best_min_score, best_min); }
        const bool kRunLO = it >= o->lo_starting_iterations && best_min_score < kMax;
        if (!kBestMin && !kRunLO) continue;
        if (kRunLO) {
          ++rep->number_lo_iterations;
          double score = best_min_score;
          LocalOptimization(*o, be, &best_min, &score);
          update_best(score, best_min);
        }
        refresh();
      }
    }
    if (be.rc) { SetLastError("LO-MSAC: device evaluation failed"); return be.rc; }
  }
  rep->num_iterations = it;
  if (it <= o->lo_starting_iterations && rep->best_model_score < kMax) {
    ++rep->number_lo_iterations;
    LocalOptimization(*o, be, &best_model, &rep->best_model_score);
    rep->best_num_inliers = be.GetInliers(best_model.data(), thr, &inliers);
    rep->inlier_ratio = static_cast<double>(rep->best_num_inliers) / static_cast<double>(kN);
  }
  if (o->final_least_squares) {   // ransac.h:253-268
    Model refined = best_model;
    be.LeastSquares(inliers, refined.data());
    const double score = be.ScoreModel(refined.data());
    if (score < rep->best_model_score) {
      rep->best_model_score = score; best_model = refined;
      rep->best_num_inliers = be.GetInliers(best_model.data(), thr, &inliers);
      rep->inlier_ratio = static_cast<double>(rep->best_num_inliers) / static_cast<double>(kN);
    }
  }
  if (be.rc) { SetLastError("LO-MSAC: device evaluation failed"); return be.rc; }
  *best_out = best_model;
  rep->num_inlier_indices = (int32_t)inliers.size();
  rep->device_time_s = dev_s;
  rep->total_time_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  return PP_OK;
}


static int Pose2dEnsure(pp_pose2d_impl* h, int64_t cap, int32_t m) {
  if (cap <= h->cap && m <= h->cap_m) return PP_OK;
  void* old[] = {h->samples, h->poses, h->scores, h->inl};
  for (void* p : old) if (p) (void)hipFree(p);
  h->samples = nullptr; h->poses = nullptr; h->scores = nullptr; h->inl = nullptr;
  const int64_t c = std::max(cap, h->cap);
  const int32_t mm = std::max(m, h->cap_m);
  h->cap = 0; h->cap_m = 0;
  int rc;
  if ((rc = DeviceAlloc(&h->samples, (size_t)c * mm)) || (rc = DeviceAlloc(&h->poses, (size_t)c * 6)) || (rc = DeviceAlloc(&h->scores, (size_t)c)) ||
      (rc = DeviceAlloc(&h->inl, (size_t)c))) return rc;
  h->cap = c; h->cap_m = mm;
  return PP_OK;
}

// LO-MSAC over AbsolutePose2dEstimator: models are 2x3 poses; LeastSquares == NonMinimalSolver (sfm2d.h:141-143)
struct Pose2dBackend {
  static constexpr bool kDeferredScores = false;
  static constexpr int kDim = 6, kMinSample = 3, kNonMinSample = 6;   // sfm2d.h:113-119
  pp_pose2d_impl* h;
  double thr;
  std::vector<double> err;
  int rc = PP_OK;
  int n() const { return h->n; }
  // the errors of the model evaluated last stay on the host: ScoreModel and the GetInliers that follows it in a local optimisation (ransac.h:337-406) ask
  // about the SAME model - one launch, one copy, one synchronisation instead of two
  double cached[6]; bool have_cached = false;
  int Evaluate(const double* model) {
    if (have_cached && std::memcmp(cached, model, sizeof(cached)) == 0 && (int)err.size() == h->n) return PP_OK;
    have_cached = false;
    err.resize(h->n);
    if ((rc = Pose2dEnsure(h, 1, 3))) return rc;
    if (hipMemcpyAsync(h->poses, model, sizeof(double) * 6, hipMemcpyHostToDevice, h->stream) != hipSuccess) return PP_ERR_HIP;
    hipLaunchKernelGGL(k_pose2d_evaluate, dim3(CeilDiv(h->n, 256)), dim3(256), 0, h->stream, h->n, h->x, h->X, h->poses, h->err);
    if (hipGetLastError() != hipSuccess) return PP_ERR_HIP;
    if (hipMemcpyAsync(err.data(), h->err, sizeof(double) * h->n, hipMemcpyDeviceToHost, h->stream) != hipSuccess) return PP_ERR_HIP;
    if (hipStreamSynchronize(h->stream) != hipSuccess) return PP_ERR_HIP;
    std::memcpy(cached, model, sizeof(cached)); have_cached = true;
    return PP_OK;
  }
  double ScoreModel(const double* model) {
    if ((rc = Evaluate(model))) return std::numeric_limits<double>::max();
    return TreeMsacScore(err.data(), h->n, thr);
  }
  int GetInliers(const double* model, double t, std::vector<int>* inl) {
    if ((rc = Evaluate(model))) return 0;
    inl->clear();
    for (int i = 0; i < h->n; ++i) if (err[i] < t) inl->push_back(i);
    return (int)inl->size();
  }
  bool Solve(const std::vector<int>& sample, double* model) {
    const int m = (int)sample.size();
    if (m < 1) return false;
    if ((rc = Pose2dEnsure(h, 1, m))) return false;
    if (hipMemcpyAsync(h->samples, sample.data(), sizeof(int32_t) * m, hipMemcpyHostToDevice, h->stream) != hipSuccess) { rc = PP_ERR_HIP; return false; }
    hipLaunchKernelGGL(k_pose2d_solve, dim3(1), dim3(64), 0, h->stream, h->n, h->x, h->X, (int64_t)1, m, h->samples, h->poses);
    if (hipMemcpyAsync(model, h->poses, sizeof(double) * 6, hipMemcpyDeviceToHost, h->stream) != hipSuccess) { rc = PP_ERR_HIP; return false; }
    if (hipStreamSynchronize(h->stream) != hipSuccess) { rc = PP_ERR_HIP; return false; }
    return true;     // NonMinimalSolver always returns one model (sfm2d.cc:491-514)
  }
  void LeastSquares(const std::vector<int>& sample, double* model) { (void)Solve(sample, model); }
  int BatchSolveScore(uint32_t want, const int32_t* samples, std::vector<double>* models, std::vector<double>* scores, double* dev_s) {
    int r = Pose2dEnsure(h, want, 3); if (r) return r;
    r = Upload(h->samples, samples, (size_t)want * 3, h->stream); if (r) return r;
    PP_HIP_TRY(hipEventRecord(h->ev0, h->stream));
    hipLaunchKernelGGL(k_pose2d_solve, dim3(CeilDiv(want, 64)), dim3(64), 0, h->stream, h->n, h->x, h->X, (int64_t)want, 3, h->samples, h->poses);
    hipLaunchKernelGGL(k_pose2d_score, dim3(CeilDiv(want, 4)), dim3(256), 0, h->stream, h->n, h->x, h->X, (int)want, h->poses, thr, h->scores, h->inl);
    PP_HIP_TRY(hipGetLastError());
    PP_HIP_TRY(hipEventRecord(h->ev1, h->stream));
    models->resize((size_t)want * 6); scores->resize(want);
    r = Download(models->data(), h->poses, models->size(), h->stream); if (r) return r;
    r = Download(scores->data(), h->scores, scores->size(), h->stream); if (r) return r;
    PP_HIP_TRY(hipStreamSynchronize(h->stream));
    float ms = 0; PP_HIP_TRY(hipEventElapsedTime(&ms, h->ev0, h->ev1)); *dev_s += ms * 1e-3;
    return PP_OK;
  }
};


// LO-MSAC over FourView2dEstimator.  A model = 4 cameras (24 doubles) + the index of its point array in the handle's
// pool (-1: the points are the three-view triangulation of its cameras, what MinimalSolver produces).
static int FourViewEnsureHyp(pp_fourview2d_impl* h, int64_t num, int32_t m) {
  if (num <= h->hyp_cap && m <= h->hyp_m) return PP_OK;
  void* old[] = {h->samples, h->counts, h->best_index, h->models, h->mscores, h->best_cams, h->best_score, h->minl};
  for (void* p : old) if (p) (void)hipFree(p);
  h->samples = h->counts = h->best_index = h->minl = nullptr; h->models = h->mscores = h->best_cams = h->best_score = nullptr;
  h->hyp_cap = 0; h->hyp_m = 0;
  const int64_t cap = std::max<int64_t>(num, h->hyp_cap);
  const int32_t mm = std::max(m, h->hyp_m);
  int rc;
  if ((rc = DeviceAlloc(&h->samples, (size_t)cap * mm)) || (rc = DeviceAlloc(&h->counts, (size_t)cap)) || (rc = DeviceAlloc(&h->best_index, (size_t)cap)) ||
      (rc = DeviceAlloc(&h->models, (size_t)cap * 16 * 24)) || (rc = DeviceAlloc(&h->mscores, (size_t)cap * 16)) || (rc = DeviceAlloc(&h->minl, (size_t)cap * 16)) ||
      (rc = DeviceAlloc(&h->best_cams, (size_t)cap * 24)) || (rc = DeviceAlloc(&h->best_score, (size_t)cap))) return rc;
  h->hyp_cap = cap; h->hyp_m = mm;
  return PP_OK;
}

static int FourViewLaunchMinimal(pp_fourview2d_impl* h, int64_t num, int32_t m, const int32_t* samples, const double* frames) {
  for (int64_t i = 0; i < num * m; ++i) if (samples[i] < 0 || samples[i] >= h->n) { SetLastError("pp_fourview2d: sample index %d out of range", samples[i]); return PP_ERR_INVALID; }
  int rc = FourViewEnsureHyp(h, num, m); if (rc) return rc;
  rc = Upload(h->samples, samples, (size_t)num * m, h->stream); if (rc) return rc;
  TrifocalFrames fr;
  double def[12];
  if (!frames) { pp_fourview2d_default_frames(def); frames = def; }
  for (int e = 0; e < 4; ++e) { fr.A1[e] = frames[e]; fr.A2[e] = frames[4 + e]; fr.A3[e] = frames[8 + e]; }
  if (num <= 16) hipLaunchKernelGGL(k_fourview2d_minimal<16>, dim3(CeilDiv(num * 16, 64)), dim3(64), 0, h->stream, h->n, h->x, num, m, h->samples, fr, h->models, h->counts);      // (a lane per candidate)
  else hipLaunchKernelGGL(k_fourview2d_minimal<1>, dim3(CeilDiv(num, 64)), dim3(64), 0, h->stream, h->n, h->x, num, m, h->samples, fr, h->models, h->counts);
  PP_HIP_TRY(hipGetLastError());
  return PP_OK;
}

constexpr int kPoolChunk = 32, kSampleRing = 8;

static void FourViewPoolReset(pp_fourview2d_impl* h) {      // (the memory stays with the handle)
  h->pool_used = 0; h->dev_err_slot = -1; h->host_err_slot = -1; h->pending_score_slot = -1;
}
static int FourViewPoolEnsure(pp_fourview2d_impl* h, int slots) {
  if (slots > h->pool_cap) {
    const int cap = std::max(256, std::max(slots, 2 * h->pool_cap));
    double *cams = nullptr, *scores = nullptr;
    int rc;
    if ((rc = DeviceAlloc(&cams, (size_t)cap * 24)) || (rc = DeviceAlloc(&scores, (size_t)cap))) { if (cams) (void)hipFree(cams); return rc; }
    if (h->pool_used > 0) {
      PP_HIP_TRY(hipMemcpyAsync(cams, h->pool_cams, sizeof(double) * 24 * h->pool_used, hipMemcpyDeviceToDevice, h->stream));
      PP_HIP_TRY(hipMemcpyAsync(scores, h->pool_scores, sizeof(double) * h->pool_used, hipMemcpyDeviceToDevice, h->stream));
      PP_HIP_TRY(hipStreamSynchronize(h->stream));
    }
    if (h->pool_cams) (void)hipFree(h->pool_cams);
    if (h->pool_scores) (void)hipFree(h->pool_scores);
    h->pool_cams = cams; h->pool_scores = scores; h->pool_cap = cap;
  }
  while ((int)h->pool_X.size() * kPoolChunk < slots) {
    double* chunk = nullptr;
    const int rc = DeviceAlloc(&chunk, (size_t)kPoolChunk * 2 * h->n); if (rc) return rc;
    h->pool_X.push_back(chunk);
  }
  if ((int)h->slot_refined.size() < slots) { h->slot_refined.resize(slots, 0); h->slot_has_X.resize(slots, 0); h->slot_scored.resize(slots, 0); }
  return PP_OK;
}

// The Solver concept of FourView2dEstimator (sfm2d.h:71-77, sfm2d.cc:300-489) over device-resident models.  What the host's replay of LO-MSAC needs from a
// model is (a) its error array where an inlier list is drawn from (one download + one synchronisation per LeastSquaresFit: the shuffle of the inliers is the
// reference's std::mt19937 stream, on the host) and (b), at the END of a local optimisation, the scores and cameras of its candidates (one download).
// Everything else - cameras, points, scores, the refinements themselves - stays on the device and is enqueued without waiting.
struct FourView2dBackend {
  static constexpr int kDim = 25, kMinSample = 5, kNonMinSample = 10;    // sfm2d.h:71-77
  static constexpr bool kDeferredScores = true;
  pp_fourview2d_impl* h;
  double thr;
  const double* frames;
  std::vector<double> err;
  int rc = PP_OK;
  int n() const { return h->n; }
  double* SlotCams(int s) const { return h->pool_cams + (size_t)24 * s; }
  double* SlotX(int s) const { return h->pool_X[s / kPoolChunk] + (size_t)(s % kPoolChunk) * 2 * h->n; }
  int EnsureLsq() {
    if (h->lsq_scale) return PP_OK;
    int r;
    if ((r = DeviceAlloc(&h->lsq_scale, (size_t)6 * h->n))   // scale (2n) + observation ratios (4n): k_fv2d_points, the many-points fallback
         || (r = DeviceAlloc(&h->lsq_Xc, (size_t)2 * h->n)) || (r = DeviceAlloc(&h->xch, kPointsSlotDoubles))   // the point kernels' exchange slots
         || (r = DeviceAlloc(&h->d_iterations, 4))) return r;
    h->sample_cap = std::max<int64_t>(1024, h->n);
    if ((r = DeviceAlloc(&h->d_sample, (size_t)kSampleRing * h->sample_cap))) return r;
    if (hipHostMalloc(&h->pinned, sizeof(int32_t) * kSampleRing * h->sample_cap + sizeof(double) * ((size_t)h->n + 32 + 2), hipHostMallocDefault) != hipSuccess) { h->pinned = nullptr; return PP_ERR_HIP; }
    if (hipHostGetDevicePointer(&h->pinned_dev, h->pinned, 0) != hipSuccess) { h->pinned_dev = nullptr; (void)hipGetLastError(); }
    *PinnedTicket() = 0;
    return FourViewPoolEnsure(h, 64);
  }
  double* PinnedErr() const { return reinterpret_cast<double*>(static_cast<char*>(h->pinned) + sizeof(int32_t) * kSampleRing * h->sample_cap); }
  volatile unsigned long long* PinnedTicket() const { return reinterpret_cast<volatile unsigned long long*>(PinnedErr() + h->n + 32); }
  // the host's wait for a shipment (k_fv2d_ship): a spin on the pinned ticket; after ~2 s the stream is synchronised once and the ticket looked at again
  int WaitTicket(unsigned long long ticket) {
    const volatile unsigned long long* t = PinnedTicket();
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 0; *t != ticket; ++spins) {
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
      __builtin_ia32_pause();
#endif
      if ((spins & 0xFFF) != 0xFFF) continue;
      if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) {
        if (hipStreamSynchronize(h->stream) != hipSuccess || *t != ticket) { SetLastError("pp_fourview2d: a model's errors never arrived"); return PP_ERR_HIP; }
        break;
      }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    return PP_OK;
  }
  int FlushScore() {      // the deferred score kernel of ScoreAsync (h->err still holds that model's errors: every EvaluateAsync of another model flushes first)
    if (h->pending_score_slot < 0) return PP_OK;
    hipLaunchKernelGGL(k_fv2d_score_errors, dim3(1), dim3(64), 0, h->stream, h->n, h->err, thr, h->pool_scores + h->pending_score_slot, (int32_t*)nullptr);
    h->pending_score_slot = -1;
    return hipGetLastError() == hipSuccess ? PP_OK : PP_ERR_HIP;
  }
  int NewSlot() {
    if ((rc = FourViewPoolEnsure(h, h->pool_used + 1))) return -1;
    const int s = h->pool_used++;
    h->slot_refined[s] = 0; h->slot_has_X[s] = 0; h->slot_scored[s] = 0;
    return s;
  }
  // a model the host holds as 24 doubles (a minimal solver's candidate) gets a slot of its own; a pooled model keeps its slot
  int Materialize(double* model) {
    int s = (int)model[24];
    if (s >= 0) return s;
    if ((s = NewSlot()) < 0) return -1;
    if (hipMemcpyAsync(SlotCams(s), model, sizeof(double) * 24, hipMemcpyHostToDevice, h->stream) != hipSuccess) { rc = PP_ERR_HIP; return -1; }      // (pageable source: staged before the call returns)
    model[24] = (double)s;
    return s;
  }
  // h->err <- the errors of slot s (enqueued); an unrefined model is triangulated from views 0..2 every time, exactly as EvaluateModelOnPoint does
  // (sfm2d.cc:302-319), a refined one is measured at its own points
  int EvaluateAsync(int s) {
    if (h->dev_err_slot == s) return PP_OK;
    { const int r = FlushScore(); if (r) return r; }
    if (h->slot_refined[s]) hipLaunchKernelGGL(k_fourview2d_errors_stored, dim3(CeilDiv(h->n, 256)), dim3(256), 0, h->stream, h->n, h->x, SlotCams(s), SlotX(s), h->err);
    else { hipLaunchKernelGGL(k_fourview2d_evaluate, dim3(CeilDiv(h->n, 256)), dim3(256), 0, h->stream, h->n, h->x, SlotCams(s), h->err, SlotX(s)); h->slot_has_X[s] = 1; }
    if (hipGetLastError() != hipSuccess) return PP_ERR_HIP;
    h->dev_err_slot = s;
    return PP_OK;
  }
  int ScoreAsync(int s) {      // pool_scores[s] <- MSAC score (enqueued, once per model)
    if (h->slot_scored[s]) return PP_OK;
    int r = EvaluateAsync(s); if (r) return r;
    // (launched by FlushScore: behind the shipment of these same errors to the host when the replay asks for them next - the host's wait does not include
    // the score kernel -, before h->err is overwritten, or when the scores are read)
    h->pending_score_slot = s;
    h->slot_scored[s] = 1;
    return PP_OK;
  }
  int FetchErrors(int s, int extra_doubles = 0, const double* extra_src = nullptr) {      // host copy of the errors of slot s (+ a few doubles riding on the same synchronisation)
    if (h->host_err_slot == s && extra_doubles == 0 && (int)err.size() == h->n) return PP_OK;      // (err belongs to THIS backend object, the slot marks to the handle)
    int r = EvaluateAsync(s); if (r) return r;
    double* pin = PinnedErr();
    if (h->pinned_dev && extra_doubles <= 32) {
      const unsigned long long ticket = ++h->ticket_seq;
      char* dev = static_cast<char*>(h->pinned_dev) + (reinterpret_cast<char*>(pin) - static_cast<char*>(h->pinned));
      hipLaunchKernelGGL(k_fv2d_ship, dim3(1), dim3(1024), 0, h->stream, h->n, (const double*)h->err, extra_src, extra_doubles, reinterpret_cast<double*>(dev),
                         reinterpret_cast<unsigned long long*>(dev + sizeof(double) * ((size_t)h->n + 32)), ticket);
      if (hipGetLastError() != hipSuccess) return PP_ERR_HIP;
      if ((r = FlushScore())) return r;
      if ((r = WaitTicket(ticket))) return r;
    } else {
      if (hipMemcpyAsync(pin, h->err, sizeof(double) * h->n, hipMemcpyDeviceToHost, h->stream) != hipSuccess) return PP_ERR_HIP;
      if (extra_doubles > 0 && hipMemcpyAsync(pin + h->n, extra_src, sizeof(double) * extra_doubles, hipMemcpyDeviceToHost, h->stream) != hipSuccess) return PP_ERR_HIP;
      if ((r = FlushScore())) return r;
      if (hipStreamSynchronize(h->stream) != hipSuccess) return PP_ERR_HIP;
    }
    err.assign(pin, pin + h->n);
    h->host_err_slot = s;
    return PP_OK;
  }
  double ScoreModel(double* model) {      // immediate: the callers outside a local optimisation (final least squares)
    const int s = Materialize(model);
    if (s < 0 || (rc = ScoreAsync(s)) || (rc = FlushScore())) return std::numeric_limits<double>::max();
    double v = 0;
    if (hipMemcpyAsync(&v, h->pool_scores + s, sizeof(double), hipMemcpyDeviceToHost, h->stream) != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess) { rc = PP_ERR_HIP; return std::numeric_limits<double>::max(); }
    return v;
  }
  int ScoreModelDeferred(double* model) {      // -> the ticket ResolveScores takes (the model's slot)
    const int s = Materialize(model);
    if (s < 0 || (rc = ScoreAsync(s))) return -1;
    return s;
  }
  // scores and cameras of the candidates of one local optimisation: slots [first, pool_used) in one copy each
  template <class Cand>
  void ResolveScores(std::vector<Cand>* cand) {
    int lo = h->pool_used, hi = -1;
    for (const Cand& c : *cand) if (c.ticket >= 0) { lo = std::min(lo, c.ticket); hi = std::max(hi, c.ticket); }
    std::vector<double> sc, cm;
    if (!rc) rc = FlushScore();
    if (hi >= lo && !rc) {
      sc.resize(hi - lo + 1); cm.resize((size_t)24 * (hi - lo + 1));
      if (hipMemcpyAsync(sc.data(), h->pool_scores + lo, sizeof(double) * sc.size(), hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
          hipMemcpyAsync(cm.data(), h->pool_cams + (size_t)24 * lo, sizeof(double) * cm.size(), hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
          hipStreamSynchronize(h->stream) != hipSuccess) rc = PP_ERR_HIP;
    }
    for (Cand& c : *cand) {
      if (c.ticket < 0 || rc) { c.score = std::numeric_limits<double>::max(); continue; }
      c.score = sc[c.ticket - lo];
      for (int k = 0; k < 24; ++k) c.m[k] = cm[(size_t)24 * (c.ticket - lo) + k];
    }
  }
  int GetInliers(double* model, double t, std::vector<int>* inl) {
    inl->clear();
    const int s = Materialize(model);
    if (s < 0 || (rc = FetchErrors(s))) return 0;
    for (int i = 0; i < h->n; ++i) if (err[i] < t) inl->push_back(i);
    return (int)inl->size();
  }
  // MinimalSolver on `num` samples + score of every candidate + first strictly-smallest (what the RANSAC loop and
  // NonMinimalSolver both do with the <= 16 candidates of a sample): enqueued, results in h->best_cams / best_score
  int SolveBestAsync(int64_t num, int32_t m, const int32_t* samples) {
    int r = FourViewLaunchMinimal(h, num, m, samples, frames); if (r) return r;
    hipLaunchKernelGGL(k_fourview2d_score, dim3(CeilDiv(num * 16, 4)), dim3(256), 0, h->stream, h->n, h->x, (int)(num * 16), h->models, thr, h->mscores, h->minl);
    hipLaunchKernelGGL(k_fourview2d_select, dim3(CeilDiv(num, 64)), dim3(64), 0, h->stream, num, h->counts, h->mscores, h->models, h->best_cams, h->best_score,
                       h->best_index);
    PP_HIP_TRY(hipGetLastError());
    return PP_OK;
  }
  int SolveBest(int64_t num, int32_t m, const int32_t* samples, std::vector<double>* models, std::vector<double>* scores) {
    int r = SolveBestAsync(num, m, samples); if (r) return r;
    std::vector<double> cams((size_t)num * 24);
    scores->resize(num);
    r = Download(cams.data(), h->best_cams, cams.size(), h->stream); if (r) return r;
    r = Download(scores->data(), h->best_score, (size_t)num, h->stream); if (r) return r;
    PP_HIP_TRY(hipStreamSynchronize(h->stream));
    models->resize((size_t)num * kDim);
    for (int64_t i = 0; i < num; ++i) {
      for (int k = 0; k < 24; ++k) (*models)[(size_t)i * kDim + k] = cams[(size_t)i * 24 + k];
      (*models)[(size_t)i * kDim + 24] = -1.0;
    }
    return PP_OK;
  }
  int BatchSolveScore(uint32_t want, const int32_t* samples, std::vector<double>* models, std::vector<double>* scores, double* dev_s) {
    PP_HIP_TRY(hipEventRecord(h->ev0, h->stream));
    const int r = SolveBest(want, kMinSample, samples, models, scores);
    if (r) return r;
    PP_HIP_TRY(hipEventRecord(h->ev1, h->stream));
    PP_HIP_TRY(hipEventSynchronize(h->ev1));
    float ms = 0; PP_HIP_TRY(hipEventElapsedTime(&ms, h->ev0, h->ev1)); *dev_s += ms * 1e-3;
    return PP_OK;
  }
  // NonMinimalSolver (sfm2d.cc:446-467): the winner of the sample's candidates becomes a pooled model with its score; ONE synchronisation brings back
  // whether there is a model at all and - on the same wait - the errors the LeastSquaresFit that follows draws its inliers from
  bool Solve(const std::vector<int>& sample, double* model) {
    std::vector<int32_t> s32(sample.begin(), sample.end());
    for (int v : s32) if (v < 0 || v >= h->n) return false;
    if ((rc = SolveBestAsync(1, (int32_t)s32.size(), s32.data()))) return false;
    const int s = NewSlot();
    if (s < 0) return false;
    hipLaunchKernelGGL(k_fv2d_adopt, dim3(1), dim3(64), 0, h->stream, h->best_cams, h->best_score, SlotCams(s), h->pool_scores + s);
    h->slot_scored[s] = 1;      // (k_fourview2d_score's sum is WaveMsac's, like k_fv2d_score_errors': the same bits as a later ScoreModel)
    if ((rc = FetchErrors(s, 24, SlotCams(s)))) return false;
    const double* cams = PinnedErr() + h->n;
    for (int k = 0; k < 24; ++k) model[k] = cams[k];
    model[24] = (double)s;
    return std::isfinite(model[0]);
  }
  // FourView2dEstimator::LeastSquares (sfm2d.cc:469-489): a NEW pooled model refined from `model` - enqueued, nothing comes back (model[0..23] are not
  // current afterwards: ResolveScores / ModelCams read them when somebody needs them)
  void LeastSquares(const std::vector<int>& sample, double* model) {
    const int m = (int)sample.size();
    const int src = Materialize(model);
    if (src < 0) return;
    if (!h->slot_refined[src] && !h->slot_has_X[src]) { h->dev_err_slot = -1; if ((rc = EvaluateAsync(src))) return; }      // its points: the three-view triangulation
    const int dst = NewSlot();
    if (dst < 0) return;
    if (m > h->sample_cap) { rc = PP_ERR_INVALID; return; }
    int32_t* stage = static_cast<int32_t*>(h->pinned) + (size_t)h->sample_next * h->sample_cap;
    int32_t* d_s = h->d_sample + (size_t)h->sample_next * h->sample_cap;
    h->sample_next = (h->sample_next + 1) % kSampleRing;
    for (int i = 0; i < m; ++i) stage[i] = sample[i];
    if (m > 0 && hipMemcpyAsync(d_s, stage, sizeof(int32_t) * m, hipMemcpyHostToDevice, h->stream) != hipSuccess) { rc = PP_ERR_HIP; return; }
    const bool wave = h->n <= kPointsWaveMax;
    if (m <= 64) hipLaunchKernelGGL(k_fv2d_bundle<1>, dim3(1), dim3(64), 0, h->stream, h->n, h->x, m, d_s, (const double*)SlotCams(src), SlotCams(dst), (const double*)SlotX(src), SlotX(dst),
                                    h->lsq_scale, h->lsq_Xc, h->xch, kPointsSlotDoubles);
    else hipLaunchKernelGGL(k_fv2d_bundle<4>, dim3(1), dim3(256), 0, h->stream, h->n, h->x, m, d_s, (const double*)SlotCams(src), SlotCams(dst), (const double*)SlotX(src), SlotX(dst),
                            h->lsq_scale, h->lsq_Xc, h->xch, kPointsSlotDoubles);
    if (wave) {
      hipLaunchKernelGGL(k_fv2d_points_reg, dim3(CeilDiv(h->n, 256)), dim3(256), 0, h->stream, h->n, h->x, (const double*)SlotCams(dst), SlotX(dst), h->xch, h->d_iterations);
    } else {
      const int point_groups = std::min(kPointsMaxGroups, std::max(1, (h->n + kPointsThreads - 1) / kPointsThreads));
      hipLaunchKernelGGL(k_fv2d_points, dim3(point_groups), dim3(kPointsThreads), 0, h->stream, h->n, h->x, (const double*)SlotCams(dst), SlotX(dst), h->lsq_scale, h->lsq_Xc,
                         h->lsq_scale + 2 * (size_t)h->n, h->xch);
    }
    if (hipGetLastError() != hipSuccess) { rc = PP_ERR_HIP; return; }
    h->slot_refined[dst] = 1; h->slot_has_X[dst] = 1;
    model[24] = (double)dst;
  }
  // the sample ring holds kSampleRing outstanding refinements; the replay synchronises at least once per LeastSquaresFit (its inlier list), so at most one is in flight
  int ModelCams(double* model) {      // model[0..23] <- the cameras of its slot
    const int s = (int)model[24];
    if (s < 0) return PP_OK;
    if (hipMemcpyAsync(model, SlotCams(s), sizeof(double) * 24, hipMemcpyDeviceToHost, h->stream) != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess) return (rc = PP_ERR_HIP);
    return PP_OK;
  }
  const double* ModelPoints(double* model) {      // device pointer to the model's points (the triangulation from views 0..2 when it was never refined)
    const int s = Materialize(model);
    if (s < 0) return nullptr;
    if (!h->slot_has_X[s]) { h->dev_err_slot = -1; if ((rc = EvaluateAsync(s))) return nullptr; }
    return SlotX(s);
  }
  void FreeSlots() { FourViewPoolReset(h); }
};

}  // namespace ppsfm

using namespace ppsfm;

extern "C" {

void pp_lomsac_options_default(pp_lomsac_options* o) {
  if (!o) return;
  o->min_num_iterations = 100; o->max_num_iterations = 10000; o->success_probability = 0.9999; o->squared_inlier_threshold = 1.0;
  o->random_seed = 0; o->num_lo_steps = 10; o->threshold_multiplier = std::sqrt(2.0); o->num_lsq_iterations = 4;
  o->min_sample_multiplicator = 7; o->non_min_sample_multiplier = 3; o->lo_starting_iterations = 50; o->final_least_squares = 0;
  o->chunk_iterations = 0;
}

int pp_planar_destroy(pp_planar_handle h) try {
  if (!h) return PP_OK;
  (void)hipSetDevice(h->device);
  void* bufs[] = {h->rec, h->lines, h->samples, h->offsets, h->scores, h->err, h->X, h->inl, h->d_poses, h->d_Rg};
  for (void* b : bufs) if (b) (void)hipFree(b);
  if (h->ev0) (void)hipEventDestroy(h->ev0);
  if (h->ev1) (void)hipEventDestroy(h->ev1);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
  return PP_OK;
} PP_API_CATCH("pp_planar_destroy")

int pp_planar_create(int32_t n, const double* poses, const double* lines, const double* Rg, int device, pp_planar_handle* out) try {
  PP_REQUIRE(out && n > 0 && poses && lines && Rg, "pp_planar_create: bad argument");
  *out = nullptr;
  int ndev = 0;
  PP_HIP_TRY(hipGetDeviceCount(&ndev));
  PP_REQUIRE(device >= 0 && device < ndev, "pp_planar_create: device %d of %d", device, ndev);
  PP_HIP_TRY(hipSetDevice(device));
  pp_planar_impl* h = new pp_planar_impl();
  OnUnwind unwind{[&] { pp_planar_destroy(h); }};
  h->device = device; h->n = n;
  std::memcpy(h->poses, poses, sizeof(h->poses));
  std::memcpy(h->Rg, Rg, sizeof(h->Rg));
  // per-view constants
  double R[4][9], t0[4][3];
  for (int j = 0; j < 4; ++j) {
    const double* G = Rg + 9 * j; const double* P = poses + 12 * j;
    for (int r = 0; r < 3; ++r) {
      for (int c = 0; c < 3; ++c) R[j][3 * r + c] = G[r] * P[c] + G[3 + r] * P[4 + c] + G[6 + r] * P[8 + c];
      t0[j][r] = G[r] * P[3] + G[3 + r] * P[7] + G[6 + r] * P[11];
    }
    for (int c = 0; c < 3; ++c) h->view.r3[j][c] = R[j][6 + c];
    h->view.c0[j] = t0[j][2];
    h->view.c1[j] = (j > 0) ? G[3 + 2] : 0.0;    // d t_j(2) / d ty_j = Rg_j[1][2]
  }
  // per-track records
  std::vector<double> rec((size_t)n * kRec);
  for (int i = 0; i < n; ++i) {
    double* r = &rec[(size_t)i * kRec];
    double A[12];
    for (int j = 0; j < 4; ++j) {
      const double* l = lines + ((size_t)j * n + i) * 3;
      const double* G = Rg + 9 * j; const double* P = poses + 12 * j;
      for (int c = 0; c < 3; ++c) A[3 * j + c] = l[0] * R[j][c] + l[1] * R[j][3 + c] + l[2] * R[j][6 + c];   // a_j = R_j^T l
      double lg[3];
      for (int rr = 0; rr < 3; ++rr) lg[rr] = G[3 * rr] * l[0] + G[3 * rr + 1] * l[1] + G[3 * rr + 2] * l[2];
      r[24 + j] = -(lg[0] * P[3] + lg[1] * P[7] + lg[2] * P[11]);     // b0_j = -lg . P[:,3]
      r[28 + j] = (j > 0) ? lg[1] : 0.0;                               // g_j
      r[32 + j] = 1.0 / std::sqrt(l[0] * l[0] + l[1] * l[1]);
      for (int c = 0; c < 3; ++c) r[12 + 3 * j + c] = A[3 * j + c];
    }
    // Minv = (A^T A)^-1 A^T  (3 x 4)
    double AtA[9] = {0};
    for (int j = 0; j < 4; ++j) for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) AtA[3 * a + b] += A[3 * j + a] * A[3 * j + b];
    const double det = Det3x3(AtA);
    double inv[9];
    inv[0] = (AtA[4] * AtA[8] - AtA[5] * AtA[7]) / det; inv[1] = (AtA[2] * AtA[7] - AtA[1] * AtA[8]) / det; inv[2] = (AtA[1] * AtA[5] - AtA[2] * AtA[4]) / det;
    inv[3] = inv[1]; inv[4] = (AtA[0] * AtA[8] - AtA[2] * AtA[6]) / det; inv[5] = (AtA[2] * AtA[3] - AtA[0] * AtA[5]) / det;
    inv[6] = inv[2]; inv[7] = inv[5]; inv[8] = (AtA[0] * AtA[4] - AtA[1] * AtA[3]) / det;
    for (int c = 0; c < 3; ++c) for (int j = 0; j < 4; ++j) r[4 * c + j] = inv[3 * c] * A[3 * j] + inv[3 * c + 1] * A[3 * j + 1] + inv[3 * c + 2] * A[3 * j + 2];
  }
  int rc = PP_OK;
#define TRY(x) do { rc = (x); if (rc) { pp_planar_destroy(h); return rc; } } while (0)
#define TRYH(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { SetLastError("%s: %s", #x, hipGetErrorString(e_)); pp_planar_destroy(h); return PP_ERR_HIP; } } while (0)
  TRYH(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
  TRYH(hipEventCreate(&h->ev0)); TRYH(hipEventCreate(&h->ev1));
  TRY(DeviceAlloc(&h->rec, rec.size())); TRY(DeviceAlloc(&h->lines, (size_t)12 * n)); TRY(DeviceAlloc(&h->err, (size_t)n)); TRY(DeviceAlloc(&h->X, (size_t)3 * n));
  TRY(DeviceAlloc(&h->d_poses, 48)); TRY(DeviceAlloc(&h->d_Rg, 36));
  TRY(Upload(h->rec, rec.data(), rec.size(), h->stream)); TRY(Upload(h->lines, lines, (size_t)12 * n, h->stream));
  TRY(Upload(h->d_poses, poses, 48, h->stream)); TRY(Upload(h->d_Rg, Rg, 36, h->stream));
  TRYH(hipStreamSynchronize(h->stream));
#undef TRY
#undef TRYH
  *out = h;
  return PP_OK;
} PP_API_CATCH("pp_planar_create")

int pp_planar_solve_batch(pp_planar_handle h, int64_t num, int32_t sample_size, const int32_t* samples, double* offsets) try {
  PP_REQUIRE(h && num >= 0 && sample_size >= 3 && sample_size <= 32 && (num == 0 || (samples && offsets)), "pp_planar_solve_batch: bad argument");
  if (num == 0) return PP_OK;
  for (int64_t i = 0; i < num * sample_size; ++i) PP_REQUIRE(samples[i] >= 0 && samples[i] < h->n, "pp_planar_solve_batch: sample index out of range");
  PP_HIP_TRY(hipSetDevice(h->device));
  int rc = PlanarEnsure(h, num); if (rc) return rc;
  rc = Upload(h->samples, samples, (size_t)num * sample_size, h->stream); if (rc) return rc;
  hipLaunchKernelGGL(k_planar_solve, dim3(CeilDiv(num, 64)), dim3(64), 0, h->stream, h->n, h->lines, h->d_poses, h->d_Rg, num, sample_size, h->samples, h->offsets);
  PP_HIP_TRY(hipGetLastError());
  rc = Download(offsets, h->offsets, (size_t)num * 3, h->stream); if (rc) return rc;
  PP_HIP_TRY(hipStreamSynchronize(h->stream));
  return PP_OK;
} PP_API_CATCH("pp_planar_solve_batch")

int pp_planar_score(pp_planar_handle h, int32_t num, const double* offsets, double thr, double* msac, int32_t* inl) try {
  PP_REQUIRE(h && num >= 0 && (num == 0 || (offsets && msac && inl)), "pp_planar_score: bad argument");
  if (num == 0) return PP_OK;
  PP_HIP_TRY(hipSetDevice(h->device));
  int rc = PlanarEnsure(h, num); if (rc) return rc;
  rc = Upload(h->offsets, offsets, (size_t)num * 3, h->stream); if (rc) return rc;
  hipLaunchKernelGGL(k_planar_score, dim3(CeilDiv(num, 4)), dim3(256), 0, h->stream, h->n, h->rec, h->view, num, h->offsets, thr, h->scores, h->inl);
  PP_HIP_TRY(hipGetLastError());
  rc = Download(msac, h->scores, (size_t)num, h->stream); if (rc) return rc;
  rc = Download(inl, h->inl, (size_t)num, h->stream); if (rc) return rc;
  PP_HIP_TRY(hipStreamSynchronize(h->stream));
  return PP_OK;
} PP_API_CATCH("pp_planar_score")

int pp_planar_evaluate(pp_planar_handle h, const double* offsets, double* errors, double* X, double* cams_out) try {
  PP_REQUIRE(h && offsets && errors, "pp_planar_evaluate: bad argument");
  PP_HIP_TRY(hipSetDevice(h->device));
  hipLaunchKernelGGL(k_planar_evaluate, dim3(CeilDiv(h->n, 256)), dim3(256), 0, h->stream, h->n, h->rec, h->view, offsets[0], offsets[1], offsets[2], h->err, h->X);
  PP_HIP_TRY(hipGetLastError());
  int rc = Download(errors, h->err, (size_t)h->n, h->stream); if (rc) return rc;
  if (X) { rc = Download(X, h->X, (size_t)3 * h->n, h->stream); if (rc) return rc; }
  PP_HIP_TRY(hipStreamSynchronize(h->stream));
  if (cams_out) CamsFromOffsets(h, offsets, cams_out);
  return PP_OK;
} PP_API_CATCH("pp_planar_evaluate")

int pp_planar_lomsac(pp_planar_handle h, const pp_lomsac_options* o, pp_lomsac_report* rep, double* offsets_out, double* cams_out, int32_t* inlier_indices) try {
  PP_REQUIRE(h && o && rep, "pp_planar_lomsac: null argument");
  PP_REQUIRE(o->num_lsq_iterations >= 2 && o->num_lo_steps >= 0, "pp_planar_lomsac: bad options");
  PP_HIP_TRY(hipSetDevice(h->device));
  PlanarBackend be{h, o->squared_inlier_threshold, {}, PP_OK};
  Offsets best;
  std::vector<int> inliers;
  const int rc = LoMsacRun(o, be, rep, &best, &inliers);
  if (rc) return rc;
  if (inlier_indices) for (size_t i = 0; i < inliers.size(); ++i) inlier_indices[i] = inliers[i];
  if (offsets_out) for (int k = 0; k < 3; ++k) offsets_out[k] = best[k];
  if (cams_out) CamsFromOffsets(h, best.data(), cams_out);
  return PP_OK;
} PP_API_CATCH("pp_planar_lomsac")


int pp_pose2d_destroy(pp_pose2d_handle h) try {
  if (!h) return PP_OK;
  (void)hipSetDevice(h->device);
  void* bufs[] = {h->x, h->X, h->samples, h->poses, h->scores, h->err, h->inl};
  for (void* b : bufs) if (b) (void)hipFree(b);
  if (h->ev0) (void)hipEventDestroy(h->ev0);
  if (h->ev1) (void)hipEventDestroy(h->ev1);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
  return PP_OK;
} PP_API_CATCH("pp_pose2d_destroy")

int pp_pose2d_create(int32_t n, const double* x, const double* X, int device, pp_pose2d_handle* out) try {
  PP_REQUIRE(out && n > 0 && x && X, "pp_pose2d_create: bad argument");
  *out = nullptr;
  int ndev = 0;
  PP_HIP_TRY(hipGetDeviceCount(&ndev));
  PP_REQUIRE(device >= 0 && device < ndev, "pp_pose2d_create: device %d of %d", device, ndev);
  PP_HIP_TRY(hipSetDevice(device));
  pp_pose2d_impl* h = new pp_pose2d_impl();
  OnUnwind unwind{[&] { pp_pose2d_destroy(h); }};
  h->device = device; h->n = n;
  std::vector<double> xn(x, x + (size_t)2 * n);
  for (int i = 0; i < n; ++i) { const double nr = std::sqrt(xn[2 * i] * xn[2 * i] + xn[2 * i + 1] * xn[2 * i + 1]); xn[2 * i] /= nr; xn[2 * i + 1] /= nr; }   // sfm2d.h:104-109
  int rc = PP_OK;
  if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess || hipEventCreate(&h->ev0) != hipSuccess || hipEventCreate(&h->ev1) != hipSuccess) {
    SetLastError("pp_pose2d_create: stream/event creation failed"); pp_pose2d_destroy(h); return PP_ERR_HIP;
  }
  if ((rc = DeviceAlloc(&h->x, xn.size())) || (rc = DeviceAlloc(&h->X, (size_t)2 * n)) || (rc = DeviceAlloc(&h->err, (size_t)n)) ||
      (rc = Upload(h->x, xn.data(), xn.size(), h->stream)) || (rc = Upload(h->X, X, (size_t)2 * n, h->stream))) { pp_pose2d_destroy(h); return rc; }
  if (hipStreamSynchronize(h->stream) != hipSuccess) { pp_pose2d_destroy(h); return PP_ERR_HIP; }
  *out = h;
  return PP_OK;
} PP_API_CATCH("pp_pose2d_create")

int pp_pose2d_solve_batch(pp_pose2d_handle h, int64_t num, int32_t sample_size, const int32_t* samples, double* poses) try {
  PP_REQUIRE(h && num >= 0 && sample_size >= 1 && (num == 0 || (samples && poses)), "pp_pose2d_solve_batch: bad argument");
  if (num == 0) return PP_OK;
  for (int64_t i = 0; i < num * sample_size; ++i) PP_REQUIRE(samples[i] >= 0 && samples[i] < h->n, "pp_pose2d_solve_batch: sample index out of range");
  PP_HIP_TRY(hipSetDevice(h->device));
  int rc = Pose2dEnsure(h, num, sample_size); if (rc) return rc;
  rc = Upload(h->samples, samples, (size_t)num * sample_size, h->stream); if (rc) return rc;
  hipLaunchKernelGGL(k_pose2d_solve, dim3(CeilDiv(num, 64)), dim3(64), 0, h->stream, h->n, h->x, h->X, num, sample_size, h->samples, h->poses);
  PP_HIP_TRY(hipGetLastError());
  rc = Download(poses, h->poses, (size_t)num * 6, h->stream); if (rc) return rc;
  PP_HIP_TRY(hipStreamSynchronize(h->stream));
  return PP_OK;
} PP_API_CATCH("pp_pose2d_solve_batch")

int pp_pose2d_score(pp_pose2d_handle h, int32_t num, const double* poses, double thr, double* msac, int32_t* inl) try {
  PP_REQUIRE(h && num >= 0 && (num == 0 || (poses && msac && inl)), "pp_pose2d_score: bad argument");
  if (num == 0) return PP_OK;
  PP_HIP_TRY(hipSetDevice(h->device));
  int rc = Pose2dEnsure(h, num, 3); if (rc) return rc;
  rc = Upload(h->poses, poses, (size_t)num * 6, h->stream); if (rc) return rc;
  hipLaunchKernelGGL(k_pose2d_score, dim3(CeilDiv(num, 4)), dim3(256), 0, h->stream, h->n, h->x, h->X, num, h->poses, thr, h->scores, h->inl);
  PP_HIP_TRY(hipGetLastError());
  rc = Download(msac, h->scores, (size_t)num, h->stream); if (rc) return rc;
  rc = Download(inl, h->inl, (size_t)num, h->stream); if (rc) return rc;
  PP_HIP_TRY(hipStreamSynchronize(h->stream));
  return PP_OK;
} PP_API_CATCH("pp_pose2d_score")

int pp_pose2d_evaluate(pp_pose2d_handle h, const double* pose, double* errors) try {
  PP_REQUIRE(h && pose && errors, "pp_pose2d_evaluate: bad argument");
  PP_HIP_TRY(hipSetDevice(h->device));
  int rc = Pose2dEnsure(h, 1, 3); if (rc) return rc;
  if (!h->err && (rc = DeviceAlloc(&h->err, (size_t)h->n))) return rc;
  rc = Upload(h->poses, pose, 6, h->stream); if (rc) return rc;
  hipLaunchKernelGGL(k_pose2d_evaluate, dim3(CeilDiv(h->n, 256)), dim3(256), 0, h->stream, h->n, h->x, h->X, h->poses, h->err);
  PP_HIP_TRY(hipGetLastError());
  rc = Download(errors, h->err, (size_t)h->n, h->stream); if (rc) return rc;
  PP_HIP_TRY(hipStreamSynchronize(h->stream));
  return PP_OK;
} PP_API_CATCH("pp_pose2d_evaluate")

int pp_pose2d_lomsac(pp_pose2d_handle h, const pp_lomsac_options* o, pp_lomsac_report* rep, double* pose_out, int32_t* inlier_indices) try {
  PP_REQUIRE(h && o && rep, "pp_pose2d_lomsac: null argument");
  PP_REQUIRE(o->num_lsq_iterations >= 2 && o->num_lo_steps >= 0, "pp_pose2d_lomsac: bad options");
  PP_HIP_TRY(hipSetDevice(h->device));
  Pose2dBackend be{h, o->squared_inlier_threshold, {}, PP_OK};
  std::array<double, 6> best;
  std::vector<int> inliers;
  const int rc = LoMsacRun(o, be, rep, &best, &inliers);
  if (rc) return rc;
  if (inlier_indices) for (size_t i = 0; i < inliers.size(); ++i) inlier_indices[i] = inliers[i];
  if (pose_out) for (int k = 0; k < 6; ++k) pose_out[k] = best[k];
  return PP_OK;
} PP_API_CATCH("pp_pose2d_lomsac")

int pp_fourview2d_destroy(pp_fourview2d_handle h) try {
  if (!h) return PP_OK;
  (void)hipSetDevice(h->device);
  void* bufs[] = {h->x, h->cams, h->scores, h->err, h->X, h->inl, h->samples, h->counts, h->best_index, h->models, h->mscores, h->best_cams, h->best_score, h->minl,
                  h->lsq_scale, h->lsq_Xc, h->xch, h->d_sample, h->d_iterations, h->pool_cams, h->pool_scores};
  for (void* b : bufs) if (b) (void)hipFree(b);
  for (double* p : h->pool_X) if (p) (void)hipFree(p);
  if (h->pinned) (void)hipHostFree(h->pinned);
  if (h->ev0) (void)hipEventDestroy(h->ev0);
  if (h->ev1) (void)hipEventDestroy(h->ev1);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
  return PP_OK;
} PP_API_CATCH("pp_fourview2d_destroy")

int pp_fourview2d_create(int32_t n, const double* x, int device, pp_fourview2d_handle* out) try {
  PP_REQUIRE(out && n > 0 && x, "pp_fourview2d_create: bad argument");
  *out = nullptr;
  int ndev = 0;
  PP_HIP_TRY(hipGetDeviceCount(&ndev));
  PP_REQUIRE(device >= 0 && device < ndev, "pp_fourview2d_create: device %d of %d", device, ndev);
  PP_HIP_TRY(hipSetDevice(device));
  pp_fourview2d_impl* h = new pp_fourview2d_impl();
  OnUnwind unwind{[&] { pp_fourview2d_destroy(h); }};
  h->device = device; h->n = n;
  std::vector<double> xn(x, x + (size_t)8 * n);
  for (size_t i = 0; i < (size_t)4 * n; ++i) { const double nr = std::sqrt(xn[2 * i] * xn[2 * i] + xn[2 * i + 1] * xn[2 * i + 1]); xn[2 * i] /= nr; xn[2 * i + 1] /= nr; }   // sfm2d.h:62-67
  int rc = PP_OK;
  hipError_t e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipEventCreate(&h->ev0);
  if (e == hipSuccess) e = hipEventCreate(&h->ev1);
  if (e != hipSuccess) { SetLastError("hipStreamCreate: %s", hipGetErrorString(e)); pp_fourview2d_destroy(h); return PP_ERR_HIP; }
  if ((rc = DeviceAlloc(&h->x, xn.size())) || (rc = DeviceAlloc(&h->err, (size_t)n)) || (rc = DeviceAlloc(&h->X, (size_t)2 * n)) ||
      (rc = Upload(h->x, xn.data(), xn.size(), h->stream))) { pp_fourview2d_destroy(h); return rc; }
  if (hipStreamSynchronize(h->stream) != hipSuccess) { pp_fourview2d_destroy(h); return PP_ERR_HIP; }
  *out = h;
  return PP_OK;
} PP_API_CATCH("pp_fourview2d_create")

int pp_fourview2d_score(pp_fourview2d_handle h, int32_t num, const double* cams, double thr, double* msac, int32_t* inl) try {
  PP_REQUIRE(h && num >= 0 && (num == 0 || (cams && msac && inl)), "pp_fourview2d_score: bad argument");
  if (num == 0) return PP_OK;
  PP_HIP_TRY(hipSetDevice(h->device));
  if (num > h->cap) {
    void* old[] = {h->cams, h->scores, h->inl};
    for (void* p : old) if (p) (void)hipFree(p);
    h->cams = nullptr; h->scores = nullptr; h->inl = nullptr; h->cap = 0;
    int rc;
    if ((rc = DeviceAlloc(&h->cams, (size_t)num * 24)) || (rc = DeviceAlloc(&h->scores, (size_t)num)) || (rc = DeviceAlloc(&h->inl, (size_t)num))) return rc;
    h->cap = num;
  }
  int rc = Upload(h->cams, cams, (size_t)num * 24, h->stream); if (rc) return rc;
  hipLaunchKernelGGL(k_fourview2d_score, dim3(CeilDiv(num, 4)), dim3(256), 0, h->stream, h->n, h->x, num, h->cams, thr, h->scores, h->inl);
  PP_HIP_TRY(hipGetLastError());
  rc = Download(msac, h->scores, (size_t)num, h->stream); if (rc) return rc;
  rc = Download(inl, h->inl, (size_t)num, h->stream); if (rc) return rc;
  PP_HIP_TRY(hipStreamSynchronize(h->stream));
  return PP_OK;
} PP_API_CATCH("pp_fourview2d_score")

int pp_fourview2d_evaluate(pp_fourview2d_handle h, const double* cams, double* errors, double* X) try {
  PP_REQUIRE(h && cams && errors, "pp_fourview2d_evaluate: bad argument");
  PP_HIP_TRY(hipSetDevice(h->device));
  if (h->cap < 1) {
    int rc;
    if ((rc = DeviceAlloc(&h->cams, 24)) || (rc = DeviceAlloc(&h->scores, 1)) || (rc = DeviceAlloc(&h->inl, 1))) return rc;
    h->cap = 1;
  }
  int rc = Upload(h->cams, cams, 24, h->stream); if (rc) return rc;
  hipLaunchKernelGGL(k_fourview2d_evaluate, dim3(CeilDiv(h->n, 256)), dim3(256), 0, h->stream, h->n, h->x, h->cams, h->err, h->X);
  PP_HIP_TRY(hipGetLastError());
  rc = Download(errors, h->err, (size_t)h->n, h->stream); if (rc) return rc;
  if (X) { rc = Download(X, h->X, (size_t)2 * h->n, h->stream); if (rc) return rc; }
  PP_HIP_TRY(hipStreamSynchronize(h->stream));
  return PP_OK;
} PP_API_CATCH("pp_fourview2d_evaluate")


int pp_fourview2d_evaluate_points(pp_fourview2d_handle h, const double* cams, const double* X, double* errors) try {
  PP_REQUIRE(h && cams && X && errors, "pp_fourview2d_evaluate_points: bad argument");
  PP_HIP_TRY(hipSetDevice(h->device));
  if (h->cap < 1) {
    int rc;
    if ((rc = DeviceAlloc(&h->cams, 24)) || (rc = DeviceAlloc(&h->scores, 1)) || (rc = DeviceAlloc(&h->inl, 1))) return rc;
    h->cap = 1;
  }
  int rc = Upload(h->cams, cams, 24, h->stream); if (rc) return rc;
  rc = Upload(h->X, X, (size_t)2 * h->n, h->stream); if (rc) return rc;
  hipLaunchKernelGGL(k_fourview2d_errors_stored, dim3(CeilDiv(h->n, 256)), dim3(256), 0, h->stream, h->n, h->x, h->cams, h->X, h->err);
  PP_HIP_TRY(hipGetLastError());
  rc = Download(errors, h->err, (size_t)h->n, h->stream); if (rc) return rc;
  PP_HIP_TRY(hipStreamSynchronize(h->stream));
  return PP_OK;
} PP_API_CATCH("pp_fourview2d_evaluate_points")

int pp_fourview2d_default_frames(double* frames) try {
  PP_REQUIRE(frames, "pp_fourview2d_default_frames: null");
  // fixed, well-conditioned stand-in for the reference's per-call Matrix2d::setRandom() (sfm2d.cc:231-235)
  uint64_t state = 0x243F6A8885A308D3ull;
  for (int k = 0; k < 3; ++k) {
    double* A = frames + 4 * k;
    for (;;) {
      for (int e = 0; e < 4; ++e) {
        state += 0x9E3779B97F4A7C15ull;
        uint64_t z = state; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
        A[e] = 2.0 * ((double)(z >> 11) * (1.0 / 9007199254740992.0)) - 1.0;
      }
      if (std::fabs(A[0] * A[3] - A[1] * A[2]) > 0.25) break;
    }
  }
  return PP_OK;
} PP_API_CATCH("pp_fourview2d_default_frames")

int pp_fourview2d_minimal_batch(pp_fourview2d_handle h, int64_t num, int32_t sample_size, const int32_t* samples, const double* frames, double* cams,
                                int32_t* counts) try {
  PP_REQUIRE(h && num >= 0 && sample_size >= 5 && (num == 0 || (samples && cams && counts)), "pp_fourview2d_minimal_batch: bad argument (sample_size >= 5)");
  if (num == 0) return PP_OK;
  PP_HIP_TRY(hipSetDevice(h->device));
  int rc = FourViewLaunchMinimal(h, num, sample_size, samples, frames); if (rc) return rc;
  rc = Download(counts, h->counts, (size_t)num, h->stream); if (rc) return rc;
  rc = Download(cams, h->models, (size_t)num * 16 * 24, h->stream); if (rc) return rc;
  PP_HIP_TRY(hipStreamSynchronize(h->stream));
  for (int64_t i = 0; i < num; ++i) if (counts[i] == 0) for (int e = 0; e < 16 * 24; ++e) cams[i * 16 * 24 + e] = NAN;
  return PP_OK;
} PP_API_CATCH("pp_fourview2d_minimal_batch")

int pp_fourview2d_nonminimal_batch(pp_fourview2d_handle h, int64_t num, int32_t sample_size, const int32_t* samples, const double* frames, double threshold,
                                   double* cams, double* msac_score, int32_t* model_index) try {
  PP_REQUIRE(h && num >= 0 && sample_size >= 5 && (num == 0 || (samples && cams && msac_score)), "pp_fourview2d_nonminimal_batch: bad argument");
  if (num == 0) return PP_OK;
  PP_HIP_TRY(hipSetDevice(h->device));
  int rc = FourViewLaunchMinimal(h, num, sample_size, samples, frames); if (rc) return rc;
  hipLaunchKernelGGL(k_fourview2d_score, dim3(CeilDiv(num * 16, 4)), dim3(256), 0, h->stream, h->n, h->x, (int)(num * 16), h->models, threshold, h->mscores, h->minl);
  hipLaunchKernelGGL(k_fourview2d_select, dim3(CeilDiv(num, 64)), dim3(64), 0, h->stream, num, h->counts, h->mscores, h->models, h->best_cams, h->best_score,
                     h->best_index);
  PP_HIP_TRY(hipGetLastError());
  rc = Download(cams, h->best_cams, (size_t)num * 24, h->stream); if (rc) return rc;
  rc = Download(msac_score, h->best_score, (size_t)num, h->stream); if (rc) return rc;
  if (model_index) { rc = Download(model_index, h->best_index, (size_t)num, h->stream); if (rc) return rc; }
  PP_HIP_TRY(hipStreamSynchronize(h->stream));
  return PP_OK;
} PP_API_CATCH("pp_fourview2d_nonminimal_batch")


int pp_fourview2d_least_squares(pp_fourview2d_handle h, int32_t m, const int32_t* sample, double* cams_inout, double* X_inout) try {
  PP_REQUIRE(h && m >= 0 && (m == 0 || sample) && cams_inout && X_inout, "pp_fourview2d_least_squares: bad argument");
  for (int i = 0; i < m; ++i) PP_REQUIRE(sample[i] >= 0 && sample[i] < h->n, "pp_fourview2d_least_squares: sample index out of range");
  PP_HIP_TRY(hipSetDevice(h->device));
  FourView2dBackend be{h, 1.0, nullptr, {}, PP_OK};
  int rc = be.EnsureLsq(); if (rc) return rc;
  // the model's cameras and points are given: a pooled model of their own, refined into a second one, which is read back
  FourViewPoolReset(h);
  const int src = be.NewSlot();
  if (src < 0) return be.rc;
  rc = Upload(be.SlotCams(src), cams_inout, 24, h->stream); if (rc) return rc;
  rc = Upload(be.SlotX(src), X_inout, (size_t)2 * h->n, h->stream); if (rc) return rc;
  h->slot_refined[src] = 1; h->slot_has_X[src] = 1;
  double model[25];
  for (int k = 0; k < 24; ++k) model[k] = cams_inout[k];
  model[24] = (double)src;
  std::vector<int> s(sample, sample + m);
  be.LeastSquares(s, model);
  if (!be.rc) be.rc = Download(X_inout, be.SlotX((int)model[24]), (size_t)2 * h->n, h->stream);
  if (!be.rc) be.ModelCams(model);      // (synchronises)
  FourViewPoolReset(h);
  if (be.rc) { SetLastError("pp_fourview2d_least_squares: device failure"); return be.rc; }
  for (int k = 0; k < 24; ++k) cams_inout[k] = model[k];
  return PP_OK;
} PP_API_CATCH("pp_fourview2d_least_squares")

int pp_fourview2d_lomsac(pp_fourview2d_handle h, const pp_lomsac_options* o, const double* frames, pp_lomsac_report* rep, double* cams_out, double* X_out,
                         int32_t* inlier_indices) try {
  PP_REQUIRE(h && o && rep, "pp_fourview2d_lomsac: null argument");
  PP_REQUIRE(o->num_lsq_iterations >= 2 && o->num_lo_steps >= 0, "pp_fourview2d_lomsac: bad options");
  PP_HIP_TRY(hipSetDevice(h->device));
  double def[12];
  if (!frames) { pp_fourview2d_default_frames(def); frames = def; }
  FourView2dBackend be{h, o->squared_inlier_threshold, frames, {}, PP_OK};
  int rc = be.EnsureLsq(); if (rc) return rc;
  std::array<double, 25> best;
  std::vector<int> inliers;
  FourViewPoolReset(h);
  rc = LoMsacRun(o, be, rep, &best, &inliers);
  if (!rc) rc = be.ModelCams(best.data());      // (a model refined by the final least squares: its cameras are still on the device)
  if (!rc && X_out) {     // the best model's points (its own if it was refined, the three-view triangulation otherwise)
    const double* Xd = be.ModelPoints(best.data());
    if (!Xd) rc = be.rc ? be.rc : PP_ERR_HIP;
    if (!rc) rc = Download(X_out, Xd, (size_t)2 * h->n, h->stream);
    if (!rc && hipStreamSynchronize(h->stream) != hipSuccess) rc = PP_ERR_HIP;
  }
  be.FreeSlots();
  if (rc) return rc;
  if (inlier_indices) for (size_t i = 0; i < inliers.size(); ++i) inlier_indices[i] = inliers[i];
  if (cams_out) for (int k = 0; k < 24; ++k) cams_out[k] = best[k];
  return PP_OK;
} PP_API_CATCH("pp_fourview2d_lomsac")

}  // extern "C"

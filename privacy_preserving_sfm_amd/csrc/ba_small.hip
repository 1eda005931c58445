// The whole Levenberg-Marquardt solve of a SMALL bundle adjustment in ONE launch of ONE workgroup.
//
// Replaces, for problems of up to 21 images, the ceres::Solve the reference runs with DENSE_SCHUR (src/optim/bundle_adjustment.cc:277-279:
// at most 50 images) - the shape of the mapper's LOCAL bundle adjustment, which builds a new BundleAdjuster for every registered image
// (src/sfm/incremental_mapper.cc:813-858: 6 images, a few thousand observations, at most 25 iterations).  The per-kernel path of
// ba_solver.hip is bound by its launches there: ~14 launches and a host round trip per LM iteration = 105-190 us for 2000 observations,
// and its decomposition (one workgroup per image, six lanes per image pair) leaves a six-image problem with six busy workgroups walking
// lists of hundreds of entries.  Here the reduced camera system (6 C + 1 <= 128 columns) lives in the LDS of one CU:
//   evaluation (K1's arithmetic, one lane per observation) -> per-image / per-point sums (one wavefront per image, four lanes per point)
//   -> (V + D^2)^-1 per point and the 192-byte records -> Schur complement: diagonal blocks + rhs by a wavefront per image, off-diagonal
//   blocks by the pair lists cut into 16-entry chunks (six lanes per chunk, partial blocks summed per pair in chunk order: deterministic,
//   no atomics) -> blocked Cholesky on 64x64 LDS tiles (chol_block.hpp: register panels, MFMA trailing updates and tile inverses), the
//   forward substitution folded in (rhs as a row), back substitution with the explicit block inverses -> point steps, model cost change,
//   trial point and its cost -> the trust-region decision - and the next iteration, without leaving the kernel.
// The host launches once per pp_ba_solve and reads the summary and the per-iteration trace.  The LM logic is the one of pp_ba_solve
// (ba_solver.hip; Ceres' published trust-region algorithm), minus its speculation: a rejected step re-evaluates nothing
// here (the Jacobians at the current point are still in place).  Sums run in a fixed order (wave butterflies, then wavefront order):
// results are deterministic and agree with the per-kernel path to rounding (different association), with the oracle to its tolerance.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "chol_block.hpp"
#include "resource_pool.hpp"
#include "line_residual.hpp"

namespace ppsfm {

constexpr int kSmallThreads = kPanelThreads;      // 16 wavefronts
// (eligibility - at most 21 images: 6 C + 1 <= 128 - and the 16-entry chunks of the pair lists are fixed at pp_ba_create, ba_eval.hip)
constexpr double kSmallBig = 1e100;               // corner of the augmented system (ba_solver.hip kBig)

struct SmallOptions {
  int max_num_iterations, max_invalid, jacobi;
  double function_tolerance, gradient_tolerance, parameter_tolerance, radius0, max_radius, min_radius, min_relative_decrease, dmin, dmax;
};
struct SmallResult {      // written by thread 0 at the end
  double initial_cost, final_cost;
  int32_t successful, unsuccessful, termination, trace_rows, current, flag_bits, pad0, pad1;
  double phase_ticks[8];      // thread 0's time per phase over the whole solve (100 MHz ticks): evaluate, reduce, norms, assemble, factor + solve, step, rest
};
struct SmallArgs {
  int C, P, N, rhs_row;
  int64_t M;
  const double *la, *lb, *lc, *intr;
  const int32_t *obs_pose, *obs_point, *obs_cam, *pt_start, *pt_obs, *pose_start, *pose_obs;
  const int32_t *pair_range, *pair_ij, *pair_entries, *chunk, *pair_chunk;      // chunk: (pair, first entry, last + 1); pair_chunk: first chunk of pair i (num_pairs + 1)
  int num_pairs, num_chunks;
  const uint8_t *pose_const, *tvec_mask, *point_const;
  double *poses[2], *points[2];      // current / candidate (they swap roles on an accepted step; SmallResult::current says where the solution is)
  double *r, *Jpose, *Jpoint, *U, *gc, *V, *gp, *Vinv, *vb, *scale_c, *scale_p, *diag_c, *diag_p, *rec, *step_c, *step_p, *partial_blocks;
  int loss_type;
  double loss_scale;
  SmallOptions o;
  double* trace;      // rows of 7
  int trace_capacity;
  SmallResult* result;
};

// ---- block-wide reductions in a fixed order: wave butterfly, then the sixteen wave totals in wavefront order ---------------------------
__device__ __forceinline__ double BlockSum(double v, double* sh /* 17 */) {
  v = WaveSum(v);
  __syncthreads();                                 // (sh may still be read from the previous call)
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  double t = 0.0;
#pragma unroll
  for (int i = 0; i < kSmallThreads / 64; ++i) t += sh[i];
  return t;
}
__device__ __forceinline__ double BlockMax(double v, double* sh) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  double t = 0.0;
#pragma unroll
  for (int i = 0; i < kSmallThreads / 64; ++i) t = fmax(t, sh[i]);
  return t;
}

__device__ __forceinline__ void SmallQuatPlus(const double* q, double d0, double d1, double d2, double* out) {      // ba_solver.hip QuatPlus
  const double n = sqrt(d0 * d0 + d1 * d1 + d2 * d2);
  if (n > 0.0) {
    const double s = sin(n) / n, w1 = cos(n);
    const double x1 = s * d0, y1 = s * d1, z1 = s * d2;
    const double w2 = q[0], x2 = q[1], y2 = q[2], z2 = q[3];
    out[0] = w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2;
    out[1] = w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2;
    out[2] = w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2;
    out[3] = w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2;
  } else {
    out[0] = q[0]; out[1] = q[1]; out[2] = q[2]; out[3] = q[3];
  }
}
__device__ __forceinline__ double SmallScaledStep(double scale, double step) {
#pragma clang fp contract(off)
  return scale * step;
}
__device__ __forceinline__ void SmallPointInverse(const double* __restrict__ v, double s0, double s1, double s2, double d0, double d1, double d2, double inv_radius,
                                                  double (&vi)[6], double* det_out) {      // ba_solver.hip PointBlockInverse
  const double a = s0 * s0 * v[0] + d0 * inv_radius, b = s0 * s1 * v[1], c = s0 * s2 * v[2];
  const double d = s1 * s1 * v[3] + d1 * inv_radius, e = s1 * s2 * v[4];
  const double f = s2 * s2 * v[5] + d2 * inv_radius;
  const double c00 = d * f - e * e, c01 = c * e - b * f, c02 = b * e - c * d;
  const double det = a * c00 + b * c01 + c * c02;
  const double id = 1.0 / det;
  vi[0] = c00 * id; vi[1] = c01 * id; vi[2] = c02 * id;
  vi[3] = (a * f - c * c) * id; vi[4] = (b * c - a * e) * id; vi[5] = (a * d - b * b) * id;
  *det_out = det;
}

// element (i, j), i >= j, of the reduced system in its 64x64 LDS tiles: [A11 | A21 | A22], row stride kLS
__device__ __forceinline__ double* SRef(double* tiles, int i, int j) {
  if (i < kNB) return tiles + i * kLS + j;
  if (j < kNB) return tiles + kNB * kLS + (i - kNB) * kLS + j;
  return tiles + 2 * kNB * kLS + (i - kNB) * kLS + (j - kNB);
}

// Every phase is a function of its own (NOT inlined): the kernel has 1024 lanes (the blocked Cholesky's sixteen wavefronts), i.e. 128 VGPRs per
// lane, and with the camera models, the wave-per-image reductions and the panels inlined into one body the allocator spilled ~900 registers (290 us per
// LM iteration).  The arguments live in LDS (`sa`): a call passes a pointer.
struct SmallShared {
  double tiles[4 * kNB * kLS];      // A11, A21, A22, M
  double inv_diag[kNB];
  double sh[kSmallThreads / 64 + 1];
  double red[kSmallThreads / 64][28];
  double xs[2 * kNB], ys[2 * kNB];
  int32_t flag;
  SmallArgs a;
};
#define PP_SMALL_PHASE __device__ __noinline__

// K1 at poses / points: residuals, loss-corrected Jacobians (k_line_eval<1, false, true>), the cost
PP_SMALL_PHASE double SmallEvaluate(SmallShared* S, const double* poses, const double* points) {
  const SmallArgs& a = S->a;
  const int tid = threadIdx.x;
  double half_rho = 0.0;
  for (int64_t o = tid; o < a.M; o += kSmallThreads) {
    const int c = a.obs_pose[o], p = a.obs_point[o], ck = a.obs_cam[o];
    const double* pose = poses + (size_t)7 * c;
    const double q[4] = {pose[0], pose[1], pose[2], pose[3]};
    const double t[3] = {pose[4], pose[5], pose[6]};
    const double X[3] = {points[3 * (size_t)p], points[3 * (size_t)p + 1], points[3 * (size_t)p + 2]};
    LineObsJac J;
    LineResidualJacobian<false>(ck & 15, a.intr + (size_t)kCamStride * (ck >> 4), q, t, X, a.la[o], a.lb[o], a.lc[o], &J);
    double rho0, rho1;
    LossRho(a.loss_type, a.loss_scale, J.r[0] * J.r[0] + J.r[1] * J.r[1], &rho0, &rho1);
    half_rho += 0.5 * rho0;
    const double sr = sqrt(rho1);      // Ceres Corrector with alpha = 0
    a.r[2 * o] = sr * J.r[0]; a.r[2 * o + 1] = sr * J.r[1];
    double* jp = a.Jpose + (size_t)12 * o;
#pragma unroll
    for (int i = 0; i < 3; ++i) { jp[i] = sr * J.Jrot[i]; jp[3 + i] = sr * J.Jt[i]; jp[6 + i] = sr * J.Jrot[3 + i]; jp[9 + i] = sr * J.Jt[3 + i]; }
    double* jx = a.Jpoint + (size_t)6 * o;
#pragma unroll
    for (int i = 0; i < 6; ++i) jx[i] = sr * J.JX[i];
  }
  return BlockSum(half_rho, S->sh);      // (its barriers also publish the Jacobians to the workgroup)
}

// K2: U_c, g_c by one wavefront per image; V_p, g_p by four lanes per point
PP_SMALL_PHASE void SmallReduce(SmallShared* S) {
  const SmallArgs& a = S->a;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  for (int c = w; c < a.C; c += kSmallThreads / 64) {
    double u[21], g[6];
#pragma unroll
    for (int i = 0; i < 21; ++i) u[i] = 0.0;
#pragma unroll
    for (int i = 0; i < 6; ++i) g[i] = 0.0;
    for (int e = a.pose_start[c] + lane; e < a.pose_start[c + 1]; e += 64) {
      const int o = a.pose_obs[e];
      const double* jp = a.Jpose + (size_t)12 * o;
      const double r0 = a.r[2 * (size_t)o], r1 = a.r[2 * (size_t)o + 1];
      int idx = 0;
#pragma unroll
      for (int x = 0; x < 6; ++x) {
        g[x] += jp[x] * r0 + jp[6 + x] * r1;
#pragma unroll
        for (int y = x; y < 6; ++y) u[idx++] += jp[x] * jp[y] + jp[6 + x] * jp[6 + y];
      }
    }
#pragma unroll
    for (int i = 0; i < 21; ++i) u[i] = WaveSum(u[i]);
#pragma unroll
    for (int i = 0; i < 6; ++i) g[i] = WaveSum(g[i]);
    if (lane == 0) {
      int idx = 0;
#pragma unroll
      for (int x = 0; x < 6; ++x) {
        a.gc[6 * (size_t)c + x] = g[x];
#pragma unroll
        for (int y = x; y < 6; ++y) { a.U[36 * (size_t)c + 6 * x + y] = u[idx]; a.U[36 * (size_t)c + 6 * y + x] = u[idx]; ++idx; }
      }
    }
  }
  for (int gid = tid; gid < 4 * ((a.P + 15) / 16) * 16; gid += kSmallThreads) {      // (whole 64-lane groups: the butterflies need their partners)
    const int p = gid >> 2, q = gid & 3;
    double v[6] = {0, 0, 0, 0, 0, 0}, g[3] = {0, 0, 0};
    if (p < a.P) {
      for (int e = a.pt_start[p] + q; e < a.pt_start[p + 1]; e += 4) {
        const int o = a.pt_obs[e];
        const double* jx = a.Jpoint + (size_t)6 * o;
        const double r0 = a.r[2 * (size_t)o], r1 = a.r[2 * (size_t)o + 1];
        v[0] += jx[0] * jx[0] + jx[3] * jx[3]; v[1] += jx[0] * jx[1] + jx[3] * jx[4]; v[2] += jx[0] * jx[2] + jx[3] * jx[5];
        v[3] += jx[1] * jx[1] + jx[4] * jx[4]; v[4] += jx[1] * jx[2] + jx[4] * jx[5]; v[5] += jx[2] * jx[2] + jx[5] * jx[5];
        g[0] += jx[0] * r0 + jx[3] * r1; g[1] += jx[1] * r0 + jx[4] * r1; g[2] += jx[2] * r0 + jx[5] * r1;
      }
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) { v[i] += __shfl_xor(v[i], 1, 64); v[i] += __shfl_xor(v[i], 2, 64); }
#pragma unroll
    for (int i = 0; i < 3; ++i) { g[i] += __shfl_xor(g[i], 1, 64); g[i] += __shfl_xor(g[i], 2, 64); }
    if (p < a.P && q == 0) {
#pragma unroll
      for (int i = 0; i < 6; ++i) a.V[6 * (size_t)p + i] = v[i];
#pragma unroll
      for (int i = 0; i < 3; ++i) a.gp[3 * (size_t)p + i] = g[i];
    }
  }
  __syncthreads();
}

// gradient max norm (Ceres 2.x: ||x - Plus(x, -g)||_inf) at poses / points; with_step: |delta|^2 and |x|^2 as well (k_norms_partial)
PP_SMALL_PHASE double SmallNorms(SmallShared* S, const double* poses, const double* points, bool with_step, double* step_norm2, double* x_norm2) {
  const SmallArgs& a = S->a;
  const int tid = threadIdx.x;
  double gmax = 0.0, st = 0.0, xn = 0.0;
  for (int c = tid; c < a.C; c += kSmallThreads) {
    const double* q = poses + 7 * (size_t)c;
    if (a.scale_c[6 * c] != 0.0) {
      double qn[4];
      SmallQuatPlus(q, -a.gc[6 * (size_t)c], -a.gc[6 * (size_t)c + 1], -a.gc[6 * (size_t)c + 2], qn);
#pragma unroll
      for (int j = 0; j < 4; ++j) gmax = fmax(gmax, fabs(q[j] - qn[j]));
#pragma unroll
      for (int j = 0; j < 7; ++j) xn += q[j] * q[j];
    }
#pragma unroll
    for (int j = 3; j < 6; ++j) if (a.scale_c[6 * c + j] != 0.0) gmax = fmax(gmax, fabs(a.gc[6 * (size_t)c + j]));
    if (with_step) {
#pragma unroll
      for (int j = 0; j < 6; ++j) { const double d = a.scale_c[6 * c + j] * a.step_c[6 * c + j]; st += d * d; }
    }
  }
  for (int i = tid; i < 3 * a.P; i += kSmallThreads) {
    const double sp = a.scale_p[i], gv = a.gp[i], xv = points[i], sv = with_step ? a.step_p[i] : 0.0;
    if (sp != 0.0) { gmax = fmax(gmax, fabs(gv)); xn += xv * xv; }
    const double d = sp * sv;
    st += d * d;
  }
  gmax = BlockMax(gmax, S->sh);
  if (with_step) { *step_norm2 = BlockSum(st, S->sh); *x_norm2 = BlockSum(xn, S->sh); }
  return gmax;
}

// Jacobi scaling 1 / (1 + ||column||), fixed at the first point; 0 for constant columns (k_jacobi_scale)
PP_SMALL_PHASE void SmallJacobiScale(SmallShared* S) {
  const SmallArgs& a = S->a;
  const int tid = threadIdx.x;
  for (int i = tid; i < 6 * a.C; i += kSmallThreads) {
    const int c = i / 6, j = i % 6;
    const bool fixed = a.pose_const[c] || (j >= 3 && ((a.tvec_mask[c] >> (j - 3)) & 1));
    a.scale_c[i] = fixed ? 0.0 : (a.o.jacobi ? 1.0 / (1.0 + sqrt(a.U[36 * (size_t)c + 7 * j])) : 1.0);
  }
  for (int i = tid; i < 3 * a.P; i += kSmallThreads) {
    const int p = i / 3, j = i % 3;
    const int di = j == 0 ? 0 : (j == 1 ? 3 : 5);
    a.scale_p[i] = a.point_const[p] ? 0.0 : (a.o.jacobi ? 1.0 / (1.0 + sqrt(a.V[6 * (size_t)p + di])) : 1.0);
  }
  __syncthreads();
}

// LM diagonal (on an accepted step), per point (V + D^2 / radius)^-1 and V^-1 b, per observation the records, then the reduced camera system
// in its LDS tiles
PP_SMALL_PHASE void SmallAssemble(SmallShared* S, double inv_radius, bool refresh_diagonal) {
  const SmallArgs& a = S->a;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int C = a.C, P = a.P, T = a.N / kNB, n = a.rhs_row;
  double* tiles = S->tiles;
  if (refresh_diagonal) {
    for (int i = tid; i < 6 * C; i += kSmallThreads) {
      const double s = a.scale_c[i];
      a.diag_c[i] = fmin(fmax(s * s * a.U[36 * (size_t)(i / 6) + 7 * (i % 6)], a.o.dmin), a.o.dmax);
    }
    for (int i = tid; i < 3 * P; i += kSmallThreads) {
      const int p = i / 3, j = i % 3;
      const int di = j == 0 ? 0 : (j == 1 ? 3 : 5);
      const double s = a.scale_p[i];
      a.diag_p[i] = fmin(fmax(s * s * a.V[6 * (size_t)p + di], a.o.dmin), a.o.dmax);
    }
    __syncthreads();
  }
  for (int p = tid; p < P; p += kSmallThreads) {
    double* vi = a.Vinv + 6 * (size_t)p;
    double* vbp = a.vb + 3 * (size_t)p;
    if (a.point_const[p]) {
#pragma unroll
      for (int i = 0; i < 6; ++i) vi[i] = 0.0;
      vbp[0] = vbp[1] = vbp[2] = 0.0;
      continue;
    }
    const double s0 = a.scale_p[3 * p], s1 = a.scale_p[3 * p + 1], s2 = a.scale_p[3 * p + 2];
    double wv[6], det;
    SmallPointInverse(a.V + 6 * (size_t)p, s0, s1, s2, a.diag_p[3 * p], a.diag_p[3 * p + 1], a.diag_p[3 * p + 2], inv_radius, wv, &det);
    if (!(det > 0.0) || !isfinite(det)) atomicOr(&S->flag, 2);
#pragma unroll
    for (int i = 0; i < 6; ++i) vi[i] = wv[i];
    const double b0 = -s0 * a.gp[3 * p], b1 = -s1 * a.gp[3 * p + 1], b2 = -s2 * a.gp[3 * p + 2];
    vbp[0] = wv[0] * b0 + wv[1] * b1 + wv[2] * b2;
    vbp[1] = wv[1] * b0 + wv[3] * b1 + wv[4] * b2;
    vbp[2] = wv[2] * b0 + wv[4] * b1 + wv[5] * b2;
  }
  __syncthreads();
  for (int64_t o = tid; o < a.M; o += kSmallThreads) {      // record [T_o = J_pt s_p (V + D^2)^-1 s_p | J_pose diag(s_c) | J_pt]  (ObsPrepareBody)
    const int c = a.obs_pose[o], p = a.obs_point[o];
    const double* jp = a.Jpose + (size_t)12 * o;
    const double* jx = a.Jpoint + (size_t)6 * o;
    double* rec = a.rec + (size_t)kRecStride * o;
#pragma unroll
    for (int i = 0; i < 12; ++i) rec[6 + i] = jp[i] * a.scale_c[6 * c + (i % 6)];
    const double s0 = a.scale_p[3 * p], s1 = a.scale_p[3 * p + 1], s2 = a.scale_p[3 * p + 2];
    const double* vi = a.Vinv + 6 * (size_t)p;
    const double v00 = vi[0] * s0 * s0, v01 = vi[1] * s0 * s1, v02 = vi[2] * s0 * s2, v11 = vi[3] * s1 * s1, v12 = vi[4] * s1 * s2, v22 = vi[5] * s2 * s2;
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      rec[3 * rr + 0] = jx[3 * rr] * v00 + jx[3 * rr + 1] * v01 + jx[3 * rr + 2] * v02;
      rec[3 * rr + 1] = jx[3 * rr] * v01 + jx[3 * rr + 1] * v11 + jx[3 * rr + 2] * v12;
      rec[3 * rr + 2] = jx[3 * rr] * v02 + jx[3 * rr + 1] * v12 + jx[3 * rr + 2] * v22;
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) rec[18 + i] = jx[i];
  }
  // the tiles: cleared (the factorisation overwrote them), then the blocks
  for (int i = tid; i < (T == 1 ? 1 : 3) * kNB * kLS / 2; i += kSmallThreads) reinterpret_cast<double2*>(tiles)[i] = make_double2(0.0, 0.0);
  __syncthreads();
  // diagonal blocks U_s + D^2 - sum_o J_o^T G_oo J_o and the reduced rhs: one wavefront per image (SchurSelfRhsBody)
  for (int c = w; c < C; c += kSmallThreads / 64) {
    double u[21], acc[6];
#pragma unroll
    for (int i = 0; i < 21; ++i) u[i] = 0.0;
#pragma unroll
    for (int i = 0; i < 6; ++i) acc[i] = 0.0;
    for (int e = a.pose_start[c] + lane; e < a.pose_start[c + 1]; e += 64) {
      const int o = a.pose_obs[e];
      const int p = a.obs_point[o];
      const double* q = a.rec + (size_t)kRecStride * o;      // T (0..5), J_pose scaled (6..17), J_pt (18..23)
      const double* jp = q + 6;
      const double g00 = q[0] * q[18] + q[1] * q[19] + q[2] * q[20], g01 = q[0] * q[21] + q[1] * q[22] + q[2] * q[23];
      const double g10 = q[3] * q[18] + q[4] * q[19] + q[5] * q[20], g11 = q[3] * q[21] + q[4] * q[22] + q[5] * q[23];
      int idx = 0;
#pragma unroll
      for (int x = 0; x < 6; ++x) {
        const double h0 = jp[x] * g00 + jp[6 + x] * g10, h1 = jp[x] * g01 + jp[6 + x] * g11;
#pragma unroll
        for (int y = x; y < 6; ++y) u[idx++] += h0 * jp[y] + h1 * jp[6 + y];
      }
      const double w0 = a.scale_p[3 * p] * a.vb[3 * (size_t)p], w1 = a.scale_p[3 * p + 1] * a.vb[3 * (size_t)p + 1], w2 = a.scale_p[3 * p + 2] * a.vb[3 * (size_t)p + 2];
      const double t0 = q[18] * w0 + q[19] * w1 + q[20] * w2, t1 = q[21] * w0 + q[22] * w1 + q[23] * w2;
#pragma unroll
      for (int j = 0; j < 6; ++j) acc[j] += jp[j] * t0 + jp[6 + j] * t1;
    }
#pragma unroll
    for (int i = 0; i < 21; ++i) u[i] = WaveSum(u[i]);
#pragma unroll
    for (int i = 0; i < 6; ++i) acc[i] = WaveSum(acc[i]);
    if (lane == 0) {
#pragma unroll
      for (int i = 0; i < 21; ++i) S->red[w][i] = u[i];
#pragma unroll
      for (int i = 0; i < 6; ++i) S->red[w][21 + i] = acc[i];
    }
    __builtin_amdgcn_wave_barrier();
    if (lane < 27) {      // (the same wavefront: its own LDS writes are visible to it in program order)
      const double sum = S->red[w][lane];
      if (lane >= 21) {
        const int j = lane - 21;
        *SRef(tiles, n, 6 * c + j) = -a.scale_c[6 * c + j] * a.gc[6 * (size_t)c + j] - sum;
      } else {
        int x = 0, rem = lane;
        while (rem >= 6 - x) { rem -= 6 - x; ++x; }
        const int y = x + rem;
        const double sa = a.scale_c[6 * c + x], sb = a.scale_c[6 * c + y];
        double v = sa * sb * a.U[36 * (size_t)c + 6 * x + y];
        if (x == y) v = (sa == 0.0) ? 1.0 : v + a.diag_c[6 * c + x] * inv_radius;
        *SRef(tiles, 6 * c + y, 6 * c + x) = v - sum;      // (row >= column)
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (tid < a.N - n) *SRef(tiles, n + tid, n + tid) = tid == 0 ? kSmallBig : 1.0;      // the corner of the augmented system, identity padding below
  // off-diagonal blocks: the pair lists in chunks of 16 entries, six lanes per chunk (one row of the 6x6 block each: SchurPairsBody's arithmetic),
  // partial blocks to memory; then every block element is the sum of its chunks in chunk order
  for (int it = tid; it < 6 * a.num_chunks; it += kSmallThreads) {
    const int ch = it / 6, ar = it % 6;
    const int e0 = a.chunk[3 * ch + 1], e1 = a.chunk[3 * ch + 2];
    double acc[6] = {0, 0, 0, 0, 0, 0};
    for (int e = e0; e < e1; ++e) {
      const int oi = a.pair_entries[2 * (size_t)e], oj = a.pair_entries[2 * (size_t)e + 1];
      const double* ri = a.rec + (size_t)kRecStride * oi;
      const double* rj = a.rec + (size_t)kRecStride * oj;
      const double g00 = ri[0] * rj[18] + ri[1] * rj[19] + ri[2] * rj[20], g01 = ri[0] * rj[21] + ri[1] * rj[22] + ri[2] * rj[23];
      const double g10 = ri[3] * rj[18] + ri[4] * rj[19] + ri[5] * rj[20], g11 = ri[3] * rj[21] + ri[4] * rj[22] + ri[5] * rj[23];
      const double pi0 = ri[6 + ar], pi1 = ri[12 + ar];
      const double h0 = pi0 * g00 + pi1 * g10, h1 = pi0 * g01 + pi1 * g11;
#pragma unroll
      for (int y = 0; y < 6; ++y) acc[y] += h0 * rj[6 + y] + h1 * rj[12 + y];
    }
#pragma unroll
    for (int y = 0; y < 6; ++y) a.partial_blocks[36 * (size_t)ch + 6 * ar + y] = acc[y];
  }
  __syncthreads();
  for (int it = tid; it < 36 * a.num_pairs; it += kSmallThreads) {
    const int pr = it / 36, el = it % 36;
    const int c0 = a.pair_chunk[pr], c1 = a.pair_chunk[pr + 1];
    if (c0 == c1) continue;
    double sum = 0.0;
    for (int ch = c0; ch < c1; ++ch) sum += a.partial_blocks[36 * (size_t)ch + el];
    const int bi = a.pair_ij[2 * pr], bj = a.pair_ij[2 * pr + 1];
    const int i = 6 * bi + el / 6, j = 6 * bj + el % 6;
    if (i >= j) *SRef(tiles, i, j) -= sum;      // (a pair of two observations of ONE image lands in its diagonal block: the lower half is what the factorisation reads)
  }
  __syncthreads();
}

// Cholesky of the augmented system on the tiles + the back substitution (chol_block.hpp SmallFactorSolveTiles) -> S->xs and step_c
PP_SMALL_PHASE void SmallFactorSolve(SmallShared* S) {
  SmallFactorSolveTiles(S->tiles, S->inv_diag, S->xs, S->ys, &S->flag, S->a.N / kNB, S->a.rhs_row, S->a.step_c);
}

// point steps (k_backsub_points), the trial point, then per observation the model cost change and the cost at its trial pose / point
PP_SMALL_PHASE void SmallStep(SmallShared* S, const double* poses, const double* points, double* poses_c, double* points_c, double* model_change, double* cand_cost) {
  const SmallArgs& a = S->a;
  const int tid = threadIdx.x;
  const double* xs = S->xs;
  for (int gid = tid; gid < 4 * ((a.P + 15) / 16) * 16; gid += kSmallThreads) {
    const int p = gid >> 2, q = gid & 3;
    double acc[3] = {0, 0, 0};
    if (p < a.P) {
      for (int e = a.pt_start[p] + q; e < a.pt_start[p + 1]; e += 4) {
        const int o = a.pt_obs[e];
        const int c = a.obs_pose[o];
        const double* jp = a.Jpose + (size_t)12 * o;
        const double* jx = a.Jpoint + (size_t)6 * o;
        double m0 = 0.0, m1 = 0.0;
#pragma unroll
        for (int j = 0; j < 6; ++j) { const double d = a.scale_c[6 * c + j] * xs[6 * c + j]; m0 += jp[j] * d; m1 += jp[6 + j] * d; }
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[j] += jx[j] * m0 + jx[3 + j] * m1;
      }
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) { acc[j] += __shfl_xor(acc[j], 1, 64); acc[j] += __shfl_xor(acc[j], 2, 64); }
    if (p < a.P && q == 0) {
      const double s0 = a.scale_p[3 * p], s1 = a.scale_p[3 * p + 1], s2 = a.scale_p[3 * p + 2];
      const double w0 = s0 * acc[0], w1 = s1 * acc[1], w2 = s2 * acc[2];
      const double* vi = a.Vinv + 6 * (size_t)p;
      const double e0 = a.vb[3 * (size_t)p + 0] - (vi[0] * w0 + vi[1] * w1 + vi[2] * w2);
      const double e1 = a.vb[3 * (size_t)p + 1] - (vi[1] * w0 + vi[3] * w1 + vi[4] * w2);
      const double e2 = a.vb[3 * (size_t)p + 2] - (vi[2] * w0 + vi[4] * w1 + vi[5] * w2);
      a.step_p[3 * (size_t)p] = e0; a.step_p[3 * (size_t)p + 1] = e1; a.step_p[3 * (size_t)p + 2] = e2;
      points_c[3 * (size_t)p] = points[3 * (size_t)p] + SmallScaledStep(s0, e0);
      points_c[3 * (size_t)p + 1] = points[3 * (size_t)p + 1] + SmallScaledStep(s1, e1);
      points_c[3 * (size_t)p + 2] = points[3 * (size_t)p + 2] + SmallScaledStep(s2, e2);
    }
  }
  for (int c = tid; c < a.C; c += kSmallThreads) {      // the trial poses (ApplyStepBody)
    const double* q = poses + 7 * (size_t)c;
    double d[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) d[j] = SmallScaledStep(a.scale_c[6 * c + j], xs[6 * c + j]);
    double qn[4];
    SmallQuatPlus(q, d[0], d[1], d[2], qn);
    double* o = poses_c + 7 * (size_t)c;
    o[0] = qn[0]; o[1] = qn[1]; o[2] = qn[2]; o[3] = qn[3];
    o[4] = q[4] + d[3]; o[5] = q[5] + d[4]; o[6] = q[6] + d[5];
  }
  __syncthreads();
  double val = 0.0, half_rho = 0.0;
  for (int64_t o = tid; o < a.M; o += kSmallThreads) {      // ModelCostBody
    const int c = a.obs_pose[o], p = a.obs_point[o];
    const double* jp = a.Jpose + (size_t)12 * o;
    const double* jx = a.Jpoint + (size_t)6 * o;
    double dc[6], dp[3];
#pragma unroll
    for (int j = 0; j < 6; ++j) dc[j] = SmallScaledStep(a.scale_c[6 * c + j], xs[6 * c + j]);
#pragma unroll
    for (int j = 0; j < 3; ++j) dp[j] = SmallScaledStep(a.scale_p[3 * p + j], a.step_p[3 * (size_t)p + j]);
    double m0 = 0.0, m1 = 0.0;
#pragma unroll
    for (int j = 0; j < 6; ++j) { m0 += jp[j] * dc[j]; m1 += jp[6 + j] * dc[j]; }
#pragma unroll
    for (int j = 0; j < 3; ++j) { m0 += jx[j] * dp[j]; m1 += jx[3 + j] * dp[j]; }
    val -= m0 * (a.r[2 * o] + m0 / 2.0) + m1 * (a.r[2 * o + 1] + m1 / 2.0);
    const int ck = a.obs_cam[o];
    double res[2];
    LineResidualOnly(ck & 15, a.intr + (size_t)kCamStride * (ck >> 4), poses_c + 7 * (size_t)c, poses_c + 7 * (size_t)c + 4, points_c + 3 * (size_t)p, a.la[o], a.lb[o],
                     a.lc[o], res);
    double rho0, rho1;
    LossRho(a.loss_type, a.loss_scale, res[0] * res[0] + res[1] * res[1], &rho0, &rho1);
    half_rho += 0.5 * rho0;
  }
  *model_change = BlockSum(val, S->sh);
  *cand_cost = BlockSum(half_rho, S->sh);
}

__global__ __launch_bounds__(kSmallThreads) void k_small_ba(SmallArgs args) {
  __shared__ __attribute__((aligned(16))) SmallShared S;
  const int tid = threadIdx.x;
  if (tid == 0) { S.a = args; S.flag = 0; }
  __syncthreads();
  const SmallArgs& a = S.a;
  int cur = 0;
  long long ticks[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_last = wall_clock64();
  auto lap = [&](int slot) { const long long t = wall_clock64(); ticks[slot] += t - t_last; t_last = t; };
  double cost = SmallEvaluate(&S, a.poses[0], a.points[0]);
  lap(0);
  SmallReduce(&S);
  lap(1);
  SmallJacobiScale(&S);
  double gmax = SmallNorms(&S, a.poses[0], a.points[0], false, nullptr, nullptr);
  lap(2);
  const double initial_cost = cost;
  double radius = a.o.radius0, decrease_factor = 2.0;
  bool reuse_diagonal = false, last_successful = true;
  int invalid = 0, successful = 0, unsuccessful = 0, rows = 0, termination = PP_TERM_NO_CONVERGENCE, flag_bits = 0;
  auto push = [&](double c0, double dc, double g, double sn, double rel, double rad, int ok) {
    if (tid == 0 && rows < a.trace_capacity) {
      double* row = a.trace + 7 * (size_t)rows;
      row[0] = c0; row[1] = dc; row[2] = g; row[3] = sn; row[4] = rel; row[5] = rad; row[6] = (double)ok;
    }
    ++rows;
  };
  push(cost, 0, gmax, 0, 0, radius, 1);
  if (!isfinite(cost)) termination = PP_TERM_FAILURE;
  for (int iter = 1; termination != PP_TERM_FAILURE; ++iter) {
    if (last_successful && gmax <= a.o.gradient_tolerance) { termination = PP_TERM_CONVERGENCE; break; }
    if (iter > a.o.max_num_iterations) { termination = PP_TERM_NO_CONVERGENCE; break; }
    if (radius < a.o.min_radius) { termination = PP_TERM_CONVERGENCE; break; }
    lap(6);
    SmallAssemble(&S, 1.0 / radius, !reuse_diagonal);
    lap(3);
    SmallFactorSolve(&S);
    lap(4);
    double model_change, ccost, step_norm2 = 0.0, x_norm2 = 0.0;
    SmallStep(&S, a.poses[cur], a.points[cur], a.poses[cur ^ 1], a.points[cur ^ 1], &model_change, &ccost);
    lap(5);
    (void)SmallNorms(&S, a.poses[cur], a.points[cur], true, &step_norm2, &x_norm2);
    lap(2);
    reuse_diagonal = true;
    const int flag = S.flag;      // (every thread reads it before thread 0 clears it: the barriers inside SmallNorms precede, one follows)
    __syncthreads();
    if (tid == 0 && flag) S.flag = 0;
    flag_bits |= flag;
    const double step_norm = sqrt(step_norm2), x_norm = sqrt(x_norm2);
    const bool valid = flag == 0 && isfinite(model_change) && model_change > 0.0 && isfinite(step_norm);
    if (!valid) {
      ++invalid;
      if (invalid >= a.o.max_invalid) { termination = PP_TERM_FAILURE; break; }
      radius /= decrease_factor; decrease_factor *= 2.0;
      push(cost, 0, gmax, 0, 0, radius, 0);
      ++unsuccessful; last_successful = false;
      continue;
    }
    invalid = 0;
    if (step_norm <= a.o.parameter_tolerance * (x_norm + a.o.parameter_tolerance)) { termination = PP_TERM_CONVERGENCE; break; }
    const double cost_change = cost - ccost;
    if (fabs(cost_change) <= a.o.function_tolerance * cost) { termination = PP_TERM_CONVERGENCE; break; }
    const double rel = cost_change / model_change;
    if (rel > a.o.min_relative_decrease) {
      cur ^= 1;
      lap(6);
      cost = SmallEvaluate(&S, a.poses[cur], a.points[cur]);
      lap(0);
      SmallReduce(&S);
      lap(1);
      gmax = SmallNorms(&S, a.poses[cur], a.points[cur], false, nullptr, nullptr);
      lap(2);
      const double v = 2.0 * rel - 1.0;
      radius = radius / fmax(1.0 / 3.0, 1.0 - v * v * v);
      radius = fmin(a.o.max_radius, radius);
      decrease_factor = 2.0; reuse_diagonal = false;
      ++successful; last_successful = true;
      push(cost, cost_change, gmax, step_norm, rel, radius, 1);
    } else {
      radius /= decrease_factor; decrease_factor *= 2.0; reuse_diagonal = true;
      ++unsuccessful; last_successful = false;
      push(cost, cost_change, gmax, step_norm, rel, radius, 0);
    }
  }
  if (tid == 0) {
    SmallResult* res = a.result;
    res->initial_cost = initial_cost; res->final_cost = cost;
    res->successful = successful; res->unsuccessful = unsuccessful; res->termination = termination;
    res->trace_rows = rows < a.trace_capacity ? rows : a.trace_capacity; res->current = cur; res->flag_bits = flag_bits;
    for (int i = 0; i < 8; ++i) res->phase_ticks[i] = (double)ticks[i];
  }
}

// true if pp_ba_solve can hand this solve to the one-launch kernel
bool SmallSolveEligible(const pp_ba_impl* h, const pp_ba_options* o) {
  // Opt-in (PPSFM_BA_SMALL=1).  Measured on MI355X (20 images / 2000 observations and 6 images / 2004 observations): 280 and 206 us per LM iteration
  // against 105 and 190 us of the per-kernel path - evaluate 14, sums 9, norms 7, ASSEMBLE 113-200, factor + solve 14 (N = 64) / 35 (N = 128), step 40.
  // The Schur gather is what one CU cannot do: 7000 pair entries x 26 doubles x six lanes through ONE texture-address path (the per-kernel path
  // spreads it over the chip), and the 384 KB of records do not fit the LDS beside the tiles.  What does fit one CU - the factorisation and the
  // solve - runs there in one launch for every small system (k_small_cholesky).
  const char* env = std::getenv("PPSFM_BA_SMALL");      // (read per solve: the tests switch it)
  const bool enabled = env && std::atoi(env) != 0;
  return enabled && h->small_ready && !h->iterative && h->NI == 0 && !BaInGroup(h) && !o->iteration_callback && !o->phase_timings;
}

int SmallSolve(pp_ba_impl* h, const pp_ba_options* o, pp_ba_summary* sum) {
  const auto t_start = std::chrono::steady_clock::now();
  hipStream_t s = h->stream;
  const int cap = o->max_num_iterations + 2;
  if (h->small_trace_cap < cap) {
    if (h->small_trace) PoolPinnedFree(h->small_trace);
    h->small_trace = nullptr;
    { const int rcp = PoolPinnedAlloc(reinterpret_cast<void**>(&h->small_trace), sizeof(double) * (7 * (size_t)cap) + sizeof(SmallResult)); if (rcp) return rcp; }
    h->small_trace_cap = cap;
  }
  SmallResult* res = reinterpret_cast<SmallResult*>(h->small_trace + 7 * (size_t)h->small_trace_cap);
  double* dev_trace = nullptr;
  PP_HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void**>(&dev_trace), h->small_trace, 0));
  SmallArgs a;
  a.C = h->C; a.P = h->P; a.N = h->N; a.rhs_row = h->n_red; a.M = h->M;
  a.la = h->la; a.lb = h->lb; a.lc = h->lc; a.intr = h->intr;
  a.obs_pose = h->obs_pose; a.obs_point = h->obs_point; a.obs_cam = h->obs_cam; a.pt_start = h->pt_start; a.pt_obs = h->pt_obs;
  a.pose_start = h->pose_start; a.pose_obs = h->pose_obs;
  a.pair_range = h->pair_start; a.pair_ij = h->pair_ij; a.pair_entries = h->pair_entries; a.chunk = h->small_chunk; a.pair_chunk = h->small_pair_chunk;
  a.num_pairs = (int)h->num_pairs; a.num_chunks = h->small_num_chunks;
  a.pose_const = h->pose_const; a.tvec_mask = h->tvec_mask; a.point_const = h->point_const;
  a.poses[0] = h->poses; a.poses[1] = h->poses_c; a.points[0] = h->points; a.points[1] = h->points_c;
  a.r = h->r; a.Jpose = h->Jpose; a.Jpoint = h->Jpoint; a.U = h->U; a.gc = h->gc; a.V = h->V; a.gp = h->gp; a.Vinv = h->Vinv; a.vb = h->vb;
  a.scale_c = h->scale_c; a.scale_p = h->scale_p; a.diag_c = h->diag_c; a.diag_p = h->diag_p; a.rec = h->JpS; a.step_c = h->step_c; a.step_p = h->step_p;
  a.partial_blocks = h->small_partials;
  a.loss_type = h->loss_type; a.loss_scale = h->loss_scale;
  a.o.max_num_iterations = o->max_num_iterations; a.o.max_invalid = o->max_num_consecutive_invalid_steps; a.o.jacobi = o->jacobi_scaling;
  a.o.function_tolerance = o->function_tolerance; a.o.gradient_tolerance = o->gradient_tolerance; a.o.parameter_tolerance = o->parameter_tolerance;
  a.o.radius0 = o->initial_trust_region_radius; a.o.max_radius = o->max_trust_region_radius; a.o.min_radius = o->min_trust_region_radius;
  a.o.min_relative_decrease = o->min_relative_decrease; a.o.dmin = o->min_lm_diagonal; a.o.dmax = o->max_lm_diagonal;
  a.trace = dev_trace; a.trace_capacity = h->small_trace_cap;
  a.result = reinterpret_cast<SmallResult*>(dev_trace + 7 * (size_t)h->small_trace_cap);
  PP_HIP_TRY(hipEventRecord(h->ev0, s));
  hipLaunchKernelGGL(k_small_ba, dim3(1), dim3(kSmallThreads), 0, s, a);
  PP_HIP_TRY(hipGetLastError());
  PP_HIP_TRY(hipEventRecord(h->ev1, s));
  PP_HIP_TRY(hipEventSynchronize(h->ev1));
  float ms = 0;
  PP_HIP_TRY(hipEventElapsedTime(&ms, h->ev0, h->ev1));
  if (res->current == 1) { std::swap(h->poses, h->poses_c); std::swap(h->points, h->points_c); }      // the solution is the current point
  {
    static const bool show = []() { const char* e = std::getenv("PPSFM_BA_SMALL_TIMING"); return e && std::atoi(e) != 0; }();
    if (show) {
      const int it = std::max(res->successful + res->unsuccessful, 1);
      fprintf(stderr, "k_small_ba: %d iterations, %.1f us on the device; per iteration [us]: evaluate %.1f reduce %.1f norms %.1f assemble %.1f factor+solve %.1f step %.1f rest %.1f\n", it,
              ms * 1e3, res->phase_ticks[0] * 0.01 / it, res->phase_ticks[1] * 0.01 / it, res->phase_ticks[2] * 0.01 / it, res->phase_ticks[3] * 0.01 / it,
              res->phase_ticks[4] * 0.01 / it, res->phase_ticks[5] * 0.01 / it, res->phase_ticks[6] * 0.01 / it);
    }
  }
  h->trace.assign(h->small_trace, h->small_trace + 7 * (size_t)res->trace_rows);
  sum->initial_cost = res->initial_cost; sum->final_cost = res->final_cost;
  sum->num_successful_steps = res->successful; sum->num_unsuccessful_steps = res->unsuccessful;
  sum->termination = res->termination; sum->num_iterations = res->successful + res->unsuccessful;
  sum->num_residuals = (int32_t)(2 * h->M); sum->num_effective_parameters = h->num_effective_pose_point;
  sum->device_time_s = ms * 1e-3;
  sum->linear_solver = PP_LINSOLVE_CHOLESKY_SMALL; sum->cholesky_fallbacks = h->chol_aux.fallbacks; sum->linear_solver_iterations = 0;
  sum->total_time_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
  if (res->termination == PP_TERM_FAILURE) {
    if (!std::isfinite(res->initial_cost)) SetLastError("pp_ba_solve: initial cost is not finite");
    else SetLastError("pp_ba_solve: %d consecutive invalid steps (linear system not positive definite or step without model decrease)", o->max_num_consecutive_invalid_steps);
    return PP_ERR_NUMERIC;
  }
  return PP_OK;
}

}  // namespace ppsfm

// K2 / K3 + the Levenberg-Marquardt driver: the device-side replacement of ceres::Solve as the
// reference configures it (src/optim/bundle_adjustment.cc:273-306, options from
// src/optim/bundle_adjustment.h:80-93 and src/controllers/incremental_mapper.cc:196-243).
//
// Per LM iteration (all on the handle's stream; the host only reads back a few scalars):
//   K1   k_line_eval          residuals + Jacobians, loss-corrected            (ba_eval.hip)
//   K2   k_reduce             U_c = sum J_c^T J_c (6x6), g_c = J_c^T r   one WORKGROUP per image,
//                             27 running sums per lane, butterfly + fixed-order LDS reduction (no atomics)
//                             V_p (3x3), g_p                              one lane per point (same launch)
//   K3a  k_prepare            one launch, two roles: per point (V_p + D_p^2)^-1 and V^-1 b_p for the current trust-region radius; per
//                             observation the 192-byte record (J_pt V^-1 | scaled J_pose | J_pt) of the gather, its point's inverse
//                             block recomputed on the spot (the same 72 bytes gathered: no dependence between the roles)
//        k_schur_self_rhs     per image (one workgroup): diagonal block of S = U + D_c^2 - sum_p W V^-1 W^T and the
//                             reduced right-hand side b_c - sum W V^-1 b_p
//        k_schur_pairs        off-diagonal blocks by GATHER: six lanes per 6x6 block pair (i,j), ten pairs per
//                             wavefront, each walking the precomputed list of observation pairs that share a point
//                             (pairs ordered by list length) — deterministic, no fp64 atomics
//   K3b  CholeskySolveAugmented   dense fp64 MFMA Cholesky of S (cholesky.hip)
//   K3c  k_step_points        point steps, -(J d)^T (r + J d / 2), the trial point x (+) d (quaternion Plus) and the COST at the trial point in one
//                             pass over the observations (four lanes per point; every observation applies the step to its own pose).
//                             With variable intrinsics or a host-callback group: k_backsub_points, k_model_cost_apply and K1 in
//                             cost-only mode at the stored trial point instead
// Columns of constant blocks (constant pose, SubsetParameterization of tvec, constant points) keep
// their slot but get Jacobi scale 0, so their step is exactly 0 and their diagonal is 1.
//
// The trust-region logic restates Ceres' published Levenberg-Marquardt strategy (radius update,
// Jacobi scaling fixed at the first point, clamped LM diagonal, invalid/unsuccessful step handling),
// see DESIGN.md §LM; Ceres itself is third-party and not part of /root/reference.
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <thread>

#include "ba_impl.hpp"
#include "resource_pool.hpp"
#include "line_residual.hpp"
#include "rccl_comm.hpp"

namespace ppsfm {

constexpr double kBig = 1e100;  // diagonal of the augmented rhs row: large enough that BIG - y^T y > 0

struct Jrow {  // one observation's scaled camera-side Jacobian (2 x 6) and point-side Jacobian (2 x 3)
  double jp[12];
  double jx[6];
};

__device__ __forceinline__ void LoadJp(const double* __restrict__ Jpose, int o, double jp[12]) {
  const double2* p = reinterpret_cast<const double2*>(Jpose + (size_t)12 * o);
#pragma unroll
  for (int i = 0; i < 6; ++i) { const double2 v = p[i]; jp[2 * i] = v.x; jp[2 * i + 1] = v.y; }
}
__device__ __forceinline__ void LoadJx(const double* __restrict__ Jpoint, int o, double jx[6]) {
  const double2* p = reinterpret_cast<const double2*>(Jpoint + (size_t)6 * o);
#pragma unroll
  for (int i = 0; i < 3; ++i) { const double2 v = p[i]; jx[2 * i] = v.x; jx[2 * i + 1] = v.y; }
}

// ---- K2 ---------------------------------------------------------------------------------------
__device__ __forceinline__ void PoseReduceBody(int c, const int32_t* __restrict__ pose_start, const int32_t* __restrict__ pose_obs,
                                               const double* __restrict__ Jpose, const double* __restrict__ r,
                                               double* __restrict__ U, double* __restrict__ gc) {
  // one WORKGROUP per image: its observations are strided over 256 lanes, wave sums by DPP row exchanges + v_readlane (WaveSumDpp:
  // the 27 ds_bpermute butterflies of four wavefronts queued up on the LDS crossbar), the four wave totals are added in wave order
  // through LDS (fixed order: deterministic).  k_reduce 22.0 -> 16.0 us at 500 images x 400 observations (rocprofv3) with the two
  // changes; the same two changes in SchurSelfRhsBody / k_pcg_images bought nothing (they share their launch with, or are as long
  // as, a gather that is the bound) and cost k_schur_blocks 2 us of occupancy: not applied there.
  __shared__ double red[4][27];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  double u[21], g[6];
#pragma unroll
  for (int i = 0; i < 21; ++i) u[i] = 0.0;
#pragma unroll
  for (int i = 0; i < 6; ++i) g[i] = 0.0;
  // two observations per lane and round, both index loads and then both rows in flight together (an image has ~400 observations: one
  // round; one observation per round was two dependent round trips per round, twice)
  auto add = [&](const double (&jp)[12], double r0, double r1) {
    int idx = 0;
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      g[a] += jp[a] * r0 + jp[6 + a] * r1;
#pragma unroll
      for (int b = a; b < 6; ++b) u[idx++] += jp[a] * jp[b] + jp[6 + a] * jp[6 + b];
    }
  };
  const int begin = pose_start[c], end = pose_start[c + 1];
  if (end - begin <= 256) {      // (workgroup-uniform) one round at most: nothing to overlap
    const int e = begin + (int)threadIdx.x;
    if (e < end) {
      const int o = pose_obs[e];
      double jp[12];
      LoadJp(Jpose, o, jp);
      add(jp, r[2 * (size_t)o], r[2 * (size_t)o + 1]);
    }
  } else {
    for (int e = begin + (int)threadIdx.x; e < end; e += 512) {
      const bool two = e + 256 < end;
      const int o = pose_obs[e], o2 = pose_obs[two ? e + 256 : e];
      double jp[12], jq[12];
      LoadJp(Jpose, o, jp);
      LoadJp(Jpose, o2, jq);
      const double r0 = r[2 * (size_t)o], r1 = r[2 * (size_t)o + 1];
      const double w = two ? 1.0 : 0.0;
      const double q0 = w * r[2 * (size_t)o2], q1 = w * r[2 * (size_t)o2 + 1];
#pragma unroll
      for (int i = 0; i < 12; ++i) jq[i] *= w;
      add(jp, r0, r1);
      add(jq, q0, q1);
    }
  }
#pragma unroll
  for (int i = 0; i < 21; ++i) u[i] = WaveSumDpp(u[i]);
#pragma unroll
  for (int i = 0; i < 6; ++i) g[i] = WaveSumDpp(g[i]);
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < 21; ++i) red[wv][i] = u[i];
#pragma unroll
    for (int i = 0; i < 6; ++i) red[wv][21 + i] = g[i];
  }
  __syncthreads();
  if (threadIdx.x < 27) {
    const int i = threadIdx.x;
    const double v = ((red[0][i] + red[1][i]) + red[2][i]) + red[3][i];
    if (i >= 21) {
      gc[6 * (size_t)c + (i - 21)] = v;
    } else {
      int a = 0, rem = i;
      while (rem >= 6 - a) { rem -= 6 - a; ++a; }
      const int b = a + rem;
      U[36 * (size_t)c + 6 * a + b] = v; U[36 * (size_t)c + 6 * b + a] = v;
    }
  }
}

// four lanes per point (lane q: observations q, q+4, ... of the point, then a fixed two-step butterfly); gid = 4 p + q
__device__ __forceinline__ void PointReduceBody(int gid, int P, const int32_t* __restrict__ pt_start, const int32_t* __restrict__ pt_obs,
                                                const double* __restrict__ Jpoint, const double* __restrict__ r,
                                                double* __restrict__ V, double* __restrict__ gp) {
  const int p = gid >> 2, q = gid & 3;
  double v[6] = {0, 0, 0, 0, 0, 0}, g[3] = {0, 0, 0};
  if (p < P) {
    for (int e = pt_start[p] + q; e < pt_start[p + 1]; e += 4) {
      const int o = pt_obs[e];
      double jx[6];
      LoadJx(Jpoint, o, jx);
      const double r0 = r[2 * (size_t)o], r1 = r[2 * (size_t)o + 1];
      v[0] += jx[0] * jx[0] + jx[3] * jx[3]; v[1] += jx[0] * jx[1] + jx[3] * jx[4]; v[2] += jx[0] * jx[2] + jx[3] * jx[5];
      v[3] += jx[1] * jx[1] + jx[4] * jx[4]; v[4] += jx[1] * jx[2] + jx[4] * jx[5]; v[5] += jx[2] * jx[2] + jx[5] * jx[5];
      g[0] += jx[0] * r0 + jx[3] * r1; g[1] += jx[1] * r0 + jx[4] * r1; g[2] += jx[2] * r0 + jx[5] * r1;
    }
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) { v[i] += __shfl_xor(v[i], 1, 64); v[i] += __shfl_xor(v[i], 2, 64); }
#pragma unroll
  for (int i = 0; i < 3; ++i) { g[i] += __shfl_xor(g[i], 1, 64); g[i] += __shfl_xor(g[i], 2, 64); }
  if (p >= P || q != 0) return;
#pragma unroll
  for (int i = 0; i < 6; ++i) V[6 * (size_t)p + i] = v[i];
#pragma unroll
  for (int i = 0; i < 3; ++i) gp[3 * (size_t)p + i] = g[i];
}

// K2 in one launch: the first C workgroups reduce one image each (U_c, g_c), the others 64 points each (V_p, g_p); the two
// parts are independent and both latency-bound, so they overlap instead of running back to back (16 + 12 us -> ~17 us)
__global__ __launch_bounds__(256) void k_reduce(int C, int P, const int32_t* __restrict__ pose_start, const int32_t* __restrict__ pose_obs,
                                                const int32_t* __restrict__ pt_start, const int32_t* __restrict__ pt_obs, const double* __restrict__ Jpose,
                                                const double* __restrict__ Jpoint, const double* __restrict__ r, double* __restrict__ U,
                                                double* __restrict__ gc, double* __restrict__ V, double* __restrict__ gp) {
  if ((int)blockIdx.x < C) PoseReduceBody(blockIdx.x, pose_start, pose_obs, Jpose, r, U, gc);
  else PointReduceBody(((int)blockIdx.x - C) * 256 + threadIdx.x, P, pt_start, pt_obs, Jpoint, r, V, gp);     // 64 points per workgroup
}

// Jacobi scaling 1/(1+||col||) (Ceres jacobi_scaling, fixed at the first evaluation); 0 for constant columns
__global__ __launch_bounds__(256) void k_jacobi_scale(int C, int P, const double* __restrict__ U, const double* __restrict__ V,
                                                      const uint8_t* __restrict__ pose_const, const uint8_t* __restrict__ tvec_mask,
                                                      const uint8_t* __restrict__ point_const, int jacobi, double* __restrict__ scale_c,
                                                      double* __restrict__ scale_p) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < 6 * C) {
    const int c = i / 6, j = i % 6;
    const bool fixed = pose_const[c] || (j >= 3 && ((tvec_mask[c] >> (j - 3)) & 1));
    const double n2 = U[36 * (size_t)c + 7 * j];
    scale_c[i] = fixed ? 0.0 : (jacobi ? 1.0 / (1.0 + sqrt(n2)) : 1.0);
  }
  if (i < 3 * P) {
    const int p = i / 3, j = i % 3;
    const int di = j == 0 ? 0 : (j == 1 ? 3 : 5);
    const double n2 = V[6 * (size_t)p + di];
    scale_p[i] = point_const[p] ? 0.0 : (jacobi ? 1.0 / (1.0 + sqrt(n2)) : 1.0);
  }
}

// LM diagonal: clamp(diag(J_s^T J_s), min, max)   (LevenbergMarquardtStrategy::ComputeStep)
__global__ __launch_bounds__(256) void k_lm_diagonal(int C, int P, const double* __restrict__ U, const double* __restrict__ V,
                                                     const double* __restrict__ scale_c, const double* __restrict__ scale_p, double dmin,
                                                     double dmax, double* __restrict__ diag_c, double* __restrict__ diag_p) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < 6 * C) {
    const int c = i / 6, j = i % 6;
    const double s = scale_c[i];
    diag_c[i] = fmin(fmax(s * s * U[36 * (size_t)c + 7 * j], dmin), dmax);
  }
  if (i < 3 * P) {
    const int p = i / 3, j = i % 3;
    const int di = j == 0 ? 0 : (j == 1 ? 3 : 5);
    const double s = scale_p[i];
    diag_p[i] = fmin(fmax(s * s * V[6 * (size_t)p + di], dmin), dmax);
  }
}

// ---- K3a --------------------------------------------------------------------------------------
// kWithDiagonal: also refreshes the LM diagonal (k_lm_diagonal's work: the pose part by the first 6C threads, a point's
// three entries by its own thread) — one launch less on every accepted step; grid covers max(P, 6C) threads then
// (s (V + D^2 / radius) s)^-1 of one point, D^2 = its LM diagonal: vi = the six entries of the symmetric inverse; returns the determinant
__device__ __forceinline__ double PointBlockInverse(const double* __restrict__ v, double s0, double s1, double s2, double d0, double d1, double d2,
                                                    double inv_radius, double (&vi)[6]) {
  const double a = s0 * s0 * v[0] + d0 * inv_radius, b = s0 * s1 * v[1], c = s0 * s2 * v[2];
  const double d = s1 * s1 * v[3] + d1 * inv_radius, e = s1 * s2 * v[4];
  const double f = s2 * s2 * v[5] + d2 * inv_radius;
  const double c00 = d * f - e * e, c01 = c * e - b * f, c02 = b * e - c * d;
  const double det = a * c00 + b * c01 + c * c02;
  const double id = 1.0 / det;
  vi[0] = c00 * id; vi[1] = c01 * id; vi[2] = c02 * id;
  vi[3] = (a * f - c * c) * id; vi[4] = (b * c - a * e) * id; vi[5] = (a * d - b * b) * id;
  return det;
}
__device__ __forceinline__ double LmDiagonalEntry(double scale, double jtj, double dmin, double dmax) { return fmin(fmax(scale * scale * jtj, dmin), dmax); }

template <bool kWithDiagonal>
__device__ __forceinline__ void PointPrepareBody(int block, int P, const double* __restrict__ V, const double* __restrict__ gp,
                                                 const double* __restrict__ scale_p, double* __restrict__ diag_p,
                                                 const uint8_t* __restrict__ point_const, double inv_radius,
                                                 double* __restrict__ Vinv, double* __restrict__ vb, int32_t* __restrict__ flag,
                                                 int C, const double* __restrict__ U, const double* __restrict__ scale_c, double dmin, double dmax,
                                                 double* __restrict__ diag_c) {
  const int p = block * 256 + threadIdx.x;
  if (kWithDiagonal) {
    if (p < 6 * C) {
      const int c = p / 6, j = p % 6;
      diag_c[p] = LmDiagonalEntry(scale_c[p], U[36 * (size_t)c + 7 * j], dmin, dmax);
    }
    if (p < P) {
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int di = j == 0 ? 0 : (j == 1 ? 3 : 5);
        diag_p[3 * p + j] = LmDiagonalEntry(scale_p[3 * p + j], V[6 * (size_t)p + di], dmin, dmax);
      }
    }
  }
  if (p >= P) return;
  double* vi = Vinv + 6 * (size_t)p;
  double* vbp = vb + 3 * (size_t)p;
  if (point_const[p]) {
#pragma unroll
    for (int i = 0; i < 6; ++i) vi[i] = 0.0;
    vbp[0] = vbp[1] = vbp[2] = 0.0;
    return;
  }
  const double s0 = scale_p[3 * p], s1 = scale_p[3 * p + 1], s2 = scale_p[3 * p + 2];
  double w[6];
  const double det = PointBlockInverse(V + 6 * (size_t)p, s0, s1, s2, diag_p[3 * p], diag_p[3 * p + 1], diag_p[3 * p + 2], inv_radius, w);
  if (!(det > 0.0) || !isfinite(det)) { atomicOr(flag, 2); }
#pragma unroll
  for (int i = 0; i < 6; ++i) vi[i] = w[i];
  const double b0 = -s0 * gp[3 * p], b1 = -s1 * gp[3 * p + 1], b2 = -s2 * gp[3 * p + 2];
  vbp[0] = w[0] * b0 + w[1] * b1 + w[2] * b2;
  vbp[1] = w[1] * b0 + w[3] * b1 + w[4] * b2;
  vbp[2] = w[2] * b0 + w[4] * b1 + w[5] * b2;
}

struct SchurArgs {
  int C, N, rhs_row;
  const int32_t *pose_start, *pose_obs, *obs_point;
  const double *Jpose, *Jpoint, *U, *gc, *Vinv, *vb, *scale_c, *scale_p, *diag_c;
  double inv_radius;
  double* S;
  int add_diagonal;  // group rank 0 adds U + D^2 (point-sharded multi-GPU: the sum over ranks must contain it once)
  // iterative handles (ba_pcg.hip) have no N x N system: the diagonal blocks go to Sd [C][36] and the reduced rhs to rhs_out [6C]
  double* Sd = nullptr;
  double* rhs_out = nullptr;
  // position in S of every column of the parameter vectors (pose c: columns 6c .. 6c+5 of the vectors; pp_ba_impl::spos): 6c unless every image carries its
  // own variable intrinsics beside its pose columns
  const int32_t* spos = nullptr;
};

// per observation and per attempt: the scaled Jacobian rows the Schur gather needs, ONE 192-byte record (ba_impl.hpp, kRecStride):
//   [ T_o = J_pt,o s_p (V+D^2)^-1 s_p (2 x 3) | J_pose,o diag(s_c) (2 x 6) | J_pt,o (2 x 3) ]      so that  G_oo' = T_o J_pt,o'^T
// The records of a workgroup's 256 observations are staged through LDS and written as one contiguous 48 KB slab with fully
// coalesced 16-byte stores (as K1 does; one lane writing 16-byte pieces at a record stride touched 48 cache lines per store).
// The point's inverse block is computed HERE, per observation, from V / the scales / the LM diagonal (PointBlockInverse: the arithmetic of
// the point role) instead of being read back from Vinv: the same 72 bytes gathered per observation, and no dependence on the point role -
// the two run in ONE launch (k_prepare).  kWithDiagonal: the LM diagonal is being refreshed in this launch: taken from V as well.
template <bool kWithDiagonal>
__device__ __forceinline__ void ObsPrepareBody(int block, int64_t M, const int32_t* __restrict__ obs_pose, const int32_t* __restrict__ obs_point,
                                               const double* __restrict__ Jpose, const double* __restrict__ Jpoint,
                                               const double* __restrict__ V, const double* __restrict__ diag_p, const uint8_t* __restrict__ point_const,
                                               double inv_radius, double dmin, double dmax, const double* __restrict__ scale_c,
                                               const double* __restrict__ scale_p, double* __restrict__ rec) {
  __shared__ __attribute__((aligned(16))) double sR[256 * kRecStride];
  const int tid = threadIdx.x;
  const int64_t o0 = (int64_t)block * 256, o = o0 + tid;
  if (o < M) {
    const int c = obs_pose[o], p = obs_point[o];
    double jp[12], jx[6];
    LoadJp(Jpose, (int)o, jp);
    LoadJx(Jpoint, (int)o, jx);
    double2* jo = reinterpret_cast<double2*>(sR + kRecStride * tid + 6);
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int j0 = (2 * i) % 6, j1 = (2 * i + 1) % 6;
      jo[i] = make_double2(jp[2 * i] * scale_c[6 * c + j0], jp[2 * i + 1] * scale_c[6 * c + j1]);
    }
    const double s0 = scale_p[3 * p], s1 = scale_p[3 * p + 1], s2 = scale_p[3 * p + 2];
    double vi[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    if (!point_const[p]) {
      const double* v = V + 6 * (size_t)p;
      double d0, d1, d2;
      if (kWithDiagonal) { d0 = LmDiagonalEntry(s0, v[0], dmin, dmax); d1 = LmDiagonalEntry(s1, v[3], dmin, dmax); d2 = LmDiagonalEntry(s2, v[5], dmin, dmax); }
      else { d0 = diag_p[3 * p]; d1 = diag_p[3 * p + 1]; d2 = diag_p[3 * p + 2]; }
      (void)PointBlockInverse(v, s0, s1, s2, d0, d1, d2, inv_radius, vi);
    }
    const double v00 = vi[0] * s0 * s0, v01 = vi[1] * s0 * s1, v02 = vi[2] * s0 * s2, v11 = vi[3] * s1 * s1, v12 = vi[4] * s1 * s2,
                 v22 = vi[5] * s2 * s2;
    double t[6];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      t[3 * r + 0] = jx[3 * r] * v00 + jx[3 * r + 1] * v01 + jx[3 * r + 2] * v02;
      t[3 * r + 1] = jx[3 * r] * v01 + jx[3 * r + 1] * v11 + jx[3 * r + 2] * v12;
      t[3 * r + 2] = jx[3 * r] * v02 + jx[3 * r + 1] * v12 + jx[3 * r + 2] * v22;
    }
    double2* to = reinterpret_cast<double2*>(sR + kRecStride * tid);
    double2* xo = reinterpret_cast<double2*>(sR + kRecStride * tid + 18);
    to[0] = make_double2(t[0], t[1]); to[1] = make_double2(t[2], t[3]); to[2] = make_double2(t[4], t[5]);
    xo[0] = make_double2(jx[0], jx[1]); xo[1] = make_double2(jx[2], jx[3]); xo[2] = make_double2(jx[4], jx[5]);
  }
  __syncthreads();
  const int64_t left = M - o0;
  const int n2 = (left < 256 ? (int)left : 256) * (kRecStride / 2);     // double2 chunks of the slab
  double2* dr = reinterpret_cast<double2*>(rec + (size_t)kRecStride * o0);
  const double2* sr = reinterpret_cast<const double2*>(sR);
#pragma unroll
  for (int it = 0; it < kRecStride / 2; ++it) {
    const int idx = it * 256 + tid;
    if (idx < n2) dr[idx] = sr[idx];
  }
}

// point role (the first point_blocks workgroups) and observation role in ONE launch: neither reads what the other writes
template <bool kWithDiagonal>
__global__ __launch_bounds__(256) void k_prepare(int point_blocks, int P, const double* __restrict__ V, const double* __restrict__ gp, const double* __restrict__ scale_p,
                                                 double* __restrict__ diag_p, const uint8_t* __restrict__ point_const, double inv_radius, double* __restrict__ Vinv,
                                                 double* __restrict__ vb, int32_t* __restrict__ flag, int C, const double* __restrict__ U,
                                                 const double* __restrict__ scale_c, double dmin, double dmax, double* __restrict__ diag_c, int64_t M,
                                                 const int32_t* __restrict__ obs_pose, const int32_t* __restrict__ obs_point, const double* __restrict__ Jpose,
                                                 const double* __restrict__ Jpoint, double* __restrict__ rec, int obs_blocks, double* __restrict__ zS, int zN,
                                                 const int32_t* __restrict__ ztiles) {
  if ((int)blockIdx.x < point_blocks)
    PointPrepareBody<kWithDiagonal>(blockIdx.x, P, V, gp, scale_p, diag_p, point_const, inv_radius, Vinv, vb, flag, C, U, scale_c, dmin, dmax, diag_c);
  else if ((int)blockIdx.x < point_blocks + obs_blocks)
    ObsPrepareBody<kWithDiagonal>((int)blockIdx.x - point_blocks, M, obs_pose, obs_point, Jpose, Jpoint, V, diag_p, point_const, inv_radius, dmin, dmax, scale_c, scale_p, rec);
  else {
    // third role (block-sparse systems): clears the listed 64x64 tiles of S - the factorisation fills exactly these, all others stay zero for good (a launch
    // of its own, 6 us, before this one until the sequence scenes made the rest of the iteration count)
    const int zt = (int)blockIdx.x - point_blocks - obs_blocks;
    const int ti = ztiles[2 * zt], tj = ztiles[2 * zt + 1];
    double2* base = reinterpret_cast<double2*>(zS + (size_t)ti * 64 * zN + (size_t)tj * 64);
    for (int idx = threadIdx.x; idx < 64 * 32; idx += 256) base[(size_t)(idx >> 5) * (zN / 2) + (idx & 31)] = make_double2(0.0, 0.0);
  }
}

// the augmented corner and the identity padding
// Per image, ONE workgroup (4 wavefronts, observations strided over 256 lanes, wave butterfly + fixed-order LDS sum):
//   diagonal 6x6 block   U_s + D^2 - sum_{o of this image} J_o^T G_oo J_o            (identity on constant columns)
//   reduced rhs          b_c - sum_{o in c} J_c,o^T (J_p,o (V^-1 b_p))               -> row rhs_row of S
// (both walk the same observation list and the same 96-byte records; they used to be two wavefront-per-image kernels
// of ~25 us each)
__device__ __forceinline__ void SchurSelfRhsBody(const SchurArgs& a, const double* __restrict__ rec, int c) {
  __shared__ double red[4][27];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (c == 0 && threadIdx.x < 64 && !a.Sd) {     // the corner of the augmented system: BIG at (rhs_row, rhs_row), identity padding below
    const int j = a.rhs_row + threadIdx.x;
    if (j < a.N) {
      if (threadIdx.x > 0) a.S[(size_t)j * a.N + j] = 1.0;
      else if (a.add_diagonal) a.S[(size_t)j * a.N + j] = kBig;
    }
  }
  double u[21], acc[6];
#pragma unroll
  for (int i = 0; i < 21; ++i) u[i] = 0.0;
#pragma unroll
  for (int i = 0; i < 6; ++i) acc[i] = 0.0;
  for (int e = a.pose_start[c] + (int)threadIdx.x; e < a.pose_start[c + 1]; e += 256) {
    const int o = a.pose_obs[e];
    const int p = a.obs_point[o];
    double jp[12], q[12];
    {      // the whole record: [T_o | J_pose,o diag(s_c) | J_pt,o]
      const double2* p = reinterpret_cast<const double2*>(RecT(rec, (size_t)o));
#pragma unroll
      for (int i = 0; i < 3; ++i) { const double2 v = p[i]; q[2 * i] = v.x; q[2 * i + 1] = v.y; }
#pragma unroll
      for (int i = 0; i < 6; ++i) { const double2 v = p[3 + i]; jp[2 * i] = v.x; jp[2 * i + 1] = v.y; }
#pragma unroll
      for (int i = 0; i < 3; ++i) { const double2 v = p[9 + i]; q[6 + 2 * i] = v.x; q[6 + 2 * i + 1] = v.y; }
    }
    const double g00 = q[0] * q[6] + q[1] * q[7] + q[2] * q[8], g01 = q[0] * q[9] + q[1] * q[10] + q[2] * q[11];
    const double g10 = q[3] * q[6] + q[4] * q[7] + q[5] * q[8], g11 = q[3] * q[9] + q[4] * q[10] + q[5] * q[11];
    int idx = 0;
#pragma unroll
    for (int x = 0; x < 6; ++x) {
      const double h0 = jp[x] * g00 + jp[6 + x] * g10, h1 = jp[x] * g01 + jp[6 + x] * g11;
#pragma unroll
      for (int y = x; y < 6; ++y) u[idx++] += h0 * jp[y] + h1 * jp[6 + y];
    }
    const double w0 = a.scale_p[3 * p] * a.vb[3 * (size_t)p], w1 = a.scale_p[3 * p + 1] * a.vb[3 * (size_t)p + 1],
                 w2 = a.scale_p[3 * p + 2] * a.vb[3 * (size_t)p + 2];
    const double t0 = q[6] * w0 + q[7] * w1 + q[8] * w2, t1 = q[9] * w0 + q[10] * w1 + q[11] * w2;
#pragma unroll
    for (int j = 0; j < 6; ++j) acc[j] += jp[j] * t0 + jp[6 + j] * t1;    // already scaled by s_c (JpS)
  }
#pragma unroll
  for (int i = 0; i < 21; ++i) u[i] = WaveSumDpp(u[i]);
#pragma unroll
  for (int i = 0; i < 6; ++i) acc[i] = WaveSumDpp(acc[i]);
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < 21; ++i) red[wv][i] = u[i];
#pragma unroll
    for (int i = 0; i < 6; ++i) red[wv][21 + i] = acc[i];
  }
  __syncthreads();
  if (threadIdx.x < 27) {
    const int i = threadIdx.x;
    const double sum = ((red[0][i] + red[1][i]) + red[2][i]) + red[3][i];
    if (i >= 21) {
      const int j = i - 21;
      const double s = a.scale_c[6 * c + j];
      const double own = a.add_diagonal ? -s * a.gc[6 * (size_t)c + j] : 0.0;
      if (a.Sd) a.rhs_out[6 * (size_t)c + j] = own - sum;
      else a.S[(size_t)a.rhs_row * a.N + a.spos[6 * c] + j] = own - sum;
    } else {
      int x = 0, rem = i;
      while (rem >= 6 - x) { rem -= 6 - x; ++x; }
      const int y = x + rem;
      const double sa = a.scale_c[6 * c + x], sb = a.scale_c[6 * c + y];
      double v = 0.0;
      if (a.add_diagonal) {
        v = sa * sb * a.U[36 * (size_t)c + 6 * x + y];
        if (x == y) v = (sa == 0.0) ? 1.0 : v + a.diag_c[6 * c + x] * a.inv_radius;
      }
      v -= sum;
      if (a.Sd) { a.Sd[36 * (size_t)c + 6 * x + y] = v; a.Sd[36 * (size_t)c + 6 * y + x] = v; }
      else {
        const int sp = a.spos[6 * c];
        a.S[(size_t)(sp + x) * a.N + sp + y] = v;
        a.S[(size_t)(sp + y) * a.N + sp + x] = v;
      }
    }
  }
}

// Off-diagonal blocks: S_ij -= sum over observation pairs sharing a point of J_i^T (T_oi J_pt,oj^T) J_j.
// SIX LANES PER BLOCK PAIR (lane = one row of the 6x6 block), ten pairs per wavefront: the lists are short (5.6
// entries per pair on the 500-camera scene, 124k pairs), so a wavefront per pair was all fixed cost and latency
// (122 us); here a wavefront walks ten lists at once and the six lanes of a pair read the same 96-byte records
// (one transaction).  Entries are summed in list order: deterministic, no atomics.
template <bool kStore>      // kStore: the block is written, not accumulated into (pp_ba_impl::pairs_complete)
__device__ __forceinline__ void SchurPairsBody(const SchurArgs& a, const double* __restrict__ rec,
                                               int64_t num_pairs, const int32_t* __restrict__ pair_start,
                                               const int32_t* __restrict__ pair_ij, const int32_t* __restrict__ pair_entries, int64_t block) {
  const int lane = threadIdx.x & 63;
  const int slot = lane / 6, ar = lane % 6;
  const int64_t wave = block * 4 + (threadIdx.x >> 6);
  const int64_t pr = wave * 10 + slot;
  if (slot >= 10 || pr >= num_pairs) return;
  const int bi = pair_ij[2 * pr], bj = pair_ij[2 * pr + 1];
  const int e0 = pair_start[2 * pr], e1 = pair_start[2 * pr + 1];    // (first, last + 1), pairs ordered by list length
  double acc[6] = {0, 0, 0, 0, 0, 0};
  int2 next = e0 < e1 ? *reinterpret_cast<const int2*>(pair_entries + 2 * (size_t)e0) : make_int2(0, 0);
  for (int e = e0; e < e1; ++e) {
    const int2 oo = next;
    if (e + 1 < e1) next = *reinterpret_cast<const int2*>(pair_entries + 2 * (size_t)(e + 1));   // the next entry's indices travel with this entry's records
    const double2* qi = reinterpret_cast<const double2*>(RecT(rec, (size_t)oo.x));       // T_oi (2x3)
    const double2* qj = reinterpret_cast<const double2*>(RecX(rec, (size_t)oo.y));       // J_pt,oj (2x3)
    const double2* pj = reinterpret_cast<const double2*>(RecJ(rec, (size_t)oo.y));
    const double2 t0 = qi[0], t1 = qi[1], t2 = qi[2];
    const double2 x0 = qj[0], x1 = qj[1], x2 = qj[2];
    const double pi0 = RecJ(rec, (size_t)oo.x)[ar], pi1 = RecJ(rec, (size_t)oo.x)[6 + ar];
    // G = T X^T : T rows (t0.x t0.y t1.x | t1.y t2.x t2.y), X rows (x0.x x0.y x1.x | x1.y x2.x x2.y)
    const double g00 = t0.x * x0.x + t0.y * x0.y + t1.x * x1.x, g01 = t0.x * x1.y + t0.y * x2.x + t1.x * x2.y;
    const double g10 = t1.y * x0.x + t2.x * x0.y + t2.y * x1.x, g11 = t1.y * x1.y + t2.x * x2.x + t2.y * x2.y;
    const double h0 = pi0 * g00 + pi1 * g10, h1 = pi0 * g01 + pi1 * g11;
    const double2 j0 = pj[0], j1 = pj[1], j2 = pj[2], j3 = pj[3], j4 = pj[4], j5 = pj[5];   // rows: (j0 j1 j2), (j3 j4 j5)
    acc[0] += h0 * j0.x + h1 * j3.x; acc[1] += h0 * j0.y + h1 * j3.y;
    acc[2] += h0 * j1.x + h1 * j4.x; acc[3] += h0 * j1.y + h1 * j4.y;
    acc[4] += h0 * j2.x + h1 * j5.x; acc[5] += h0 * j2.y + h1 * j5.y;
  }
  double2* dst = reinterpret_cast<double2*>(a.S + (size_t)(a.spos[6 * bi] + ar) * a.N + a.spos[6 * bj]);
  double2 d0 = make_double2(0.0, 0.0), d1 = d0, d2 = d0;
  if (!kStore) { d0 = dst[0]; d1 = dst[1]; d2 = dst[2]; }
  d0.x -= acc[0]; d0.y -= acc[1]; d1.x -= acc[2]; d1.y -= acc[3]; d2.x -= acc[4]; d2.y -= acc[5];
  dst[0] = d0; dst[1] = d1; dst[2] = d2;
}

__global__ __launch_bounds__(256) void k_schur_self_rhs(SchurArgs a, const double* __restrict__ rec) {
  SchurSelfRhsBody(a, rec, blockIdx.x);
}
template <bool kStore>
__global__ __launch_bounds__(256) void k_schur_pairs(SchurArgs a, const double* __restrict__ rec, int64_t num_pairs,
                                                     const int32_t* __restrict__ pair_start, const int32_t* __restrict__ pair_ij,
                                                     const int32_t* __restrict__ pair_entries) {
  SchurPairsBody<kStore>(a, rec, num_pairs, pair_start, pair_ij, pair_entries, blockIdx.x);
}
// store mode: the diagonal blocks + rhs (first C workgroups) and the off-diagonal blocks write disjoint parts of S and read the
// same records — one launch, the ~500 per-image workgroups run under the pair gather instead of before it
__global__ __launch_bounds__(256) void k_schur_blocks(SchurArgs a, const double* __restrict__ rec, int64_t num_pairs,
                                                      const int32_t* __restrict__ pair_start, const int32_t* __restrict__ pair_ij,
                                                      const int32_t* __restrict__ pair_entries) {
  if ((int)blockIdx.x < a.C) SchurSelfRhsBody(a, rec, blockIdx.x);
  else SchurPairsBody<true>(a, rec, num_pairs, pair_start, pair_ij, pair_entries, (int64_t)blockIdx.x - a.C);
}

// ---- every image with variable intrinsics of its own beside its pose columns (pp_ba_impl::intr_wide_nv = NV; image_ordering.hip PrivateIntrinsicsColumns) ----
// The image's 6 + NV columns are ONE block of the reduced system, J_w,o = [J_pose,o diag(s_c) | J_intr,o diag(s_k)] (2 x W, W = 6 + NV: the record's pose rows and
// the compact scaled rows of JkS), and the blocks are those of the pose gather with wider rows:
//   diagonal   S_cc = U_c + D^2 (pose part, as SchurSelfRhsBody) + sum_{o of c} J_w^T J_w (every entry with an intrinsics column) - sum_o J_w^T G_oo J_w
//   off it     S_ij -= sum over the pair list of (i, j) of J_w,oi^T (T_oi X_oj^T) J_w,oj          (the lists of pp_ba_create, every image listed whatever its pose)
//   rhs        -s g - sum_o J_w^T (J_pt,o (V^-1 b_p))
// instead of the block-pair lists of ba_intr.hip (k_intr_L / k_schur_gen / k_intr_kk / k_intr_sums<1>: built for cameras SHARED between images; a camera per
// image gave 250 000 twelve-lane pairs of one chunk - 0.46 ms per assembly at 500 images / 200k observations against 0.11 for the pose blocks).
template <int NV>
__global__ __launch_bounds__(256) void k_schur_wide_self(SchurArgs a, const double* __restrict__ rec, const double* __restrict__ JkS,
                                                         const int32_t* __restrict__ pose_camera, const int32_t* __restrict__ intr_off) {
  constexpr int W = 6 + NV, NU = W * (W + 1) / 2;
  __shared__ double red[4][NU + W];
  const int c = blockIdx.x;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (c == 0 && threadIdx.x < 64) {     // the corner of the augmented system: BIG at (rhs_row, rhs_row), identity padding below
    const int j = a.rhs_row + threadIdx.x;
    if (j < a.N) {
      if (threadIdx.x > 0) a.S[(size_t)j * a.N + j] = 1.0;
      else if (a.add_diagonal) a.S[(size_t)j * a.N + j] = kBig;
    }
  }
  double u[NU], acc[W];
#pragma unroll
  for (int i = 0; i < NU; ++i) u[i] = 0.0;
#pragma unroll
  for (int i = 0; i < W; ++i) acc[i] = 0.0;
  for (int e = a.pose_start[c] + (int)threadIdx.x; e < a.pose_start[c + 1]; e += 256) {
    const int o = a.pose_obs[e];
    const int p = a.obs_point[o];
    double j0[W], j1[W], q[12];
    {
      const double2* r2 = reinterpret_cast<const double2*>(RecT(rec, (size_t)o));
#pragma unroll
      for (int i = 0; i < 3; ++i) { const double2 v = r2[i]; q[2 * i] = v.x; q[2 * i + 1] = v.y; }
#pragma unroll
      for (int i = 0; i < 3; ++i) { const double2 v = r2[3 + i]; j0[2 * i] = v.x; j0[2 * i + 1] = v.y; }
#pragma unroll
      for (int i = 0; i < 3; ++i) { const double2 v = r2[6 + i]; j1[2 * i] = v.x; j1[2 * i + 1] = v.y; }
#pragma unroll
      for (int i = 0; i < 3; ++i) { const double2 v = r2[9 + i]; q[6 + 2 * i] = v.x; q[6 + 2 * i + 1] = v.y; }
      const double2* k2 = reinterpret_cast<const double2*>(JkS + (size_t)2 * kCamStride * o);
#pragma unroll
      for (int i = 0; i < NV / 2; ++i) { const double2 v0 = k2[i], v1 = k2[kCamStride / 2 + i]; j0[6 + 2 * i] = v0.x; j0[7 + 2 * i] = v0.y; j1[6 + 2 * i] = v1.x; j1[7 + 2 * i] = v1.y; }
    }
    const double g00 = q[0] * q[6] + q[1] * q[7] + q[2] * q[8], g01 = q[0] * q[9] + q[1] * q[10] + q[2] * q[11];
    const double g10 = q[3] * q[6] + q[4] * q[7] + q[5] * q[8], g11 = q[3] * q[9] + q[4] * q[10] + q[5] * q[11];
    int idx = 0;
#pragma unroll
    for (int x = 0; x < W; ++x) {
      const double h0 = j0[x] * g00 + j1[x] * g10, h1 = j0[x] * g01 + j1[x] * g11;
      const double d0 = h0 - j0[x], d1 = h1 - j1[x];      // (G - I): the direct term J^T J of every entry with an intrinsics column (the pose part's is U)
#pragma unroll
      for (int y = x; y < W; ++y) u[idx++] += (y < 6) ? h0 * j0[y] + h1 * j1[y] : d0 * j0[y] + d1 * j1[y];
    }
    const double w0 = a.scale_p[3 * p] * a.vb[3 * (size_t)p], w1 = a.scale_p[3 * p + 1] * a.vb[3 * (size_t)p + 1],
                 w2 = a.scale_p[3 * p + 2] * a.vb[3 * (size_t)p + 2];
    const double t0 = q[6] * w0 + q[7] * w1 + q[8] * w2, t1 = q[9] * w0 + q[10] * w1 + q[11] * w2;
#pragma unroll
    for (int j = 0; j < W; ++j) acc[j] += j0[j] * t0 + j1[j] * t1;
  }
#pragma unroll
  for (int i = 0; i < NU; ++i) u[i] = WaveSumDpp(u[i]);
#pragma unroll
  for (int i = 0; i < W; ++i) acc[i] = WaveSumDpp(acc[i]);
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < NU; ++i) red[wv][i] = u[i];
#pragma unroll
    for (int i = 0; i < W; ++i) red[wv][NU + i] = acc[i];
  }
  __syncthreads();
  if (threadIdx.x < NU + W) {
    const int i = threadIdx.x;
    const double sum = ((red[0][i] + red[1][i]) + red[2][i]) + red[3][i];
    const int sp = a.spos[6 * c];
    const int ioff = 6 * a.C + intr_off[pose_camera[c]];      // the image's intrinsics columns in the parameter vectors
    if (i >= NU) {
      const int j = i - NU;
      const int v = j < 6 ? 6 * c + j : ioff + j - 6;
      const double own = a.add_diagonal ? -a.scale_c[v] * a.gc[v] : 0.0;
      a.S[(size_t)a.rhs_row * a.N + sp + j] = own - sum;
    } else {
      int x = 0, rem = i;
      while (rem >= W - x) { rem -= W - x; ++x; }
      const int y = x + rem;
      double v = 0.0;
      if (a.add_diagonal) {
        if (y < 6) {
          const double sa = a.scale_c[6 * c + x], sb = a.scale_c[6 * c + y];
          v = sa * sb * a.U[36 * (size_t)c + 6 * x + y];
          if (x == y) v = (sa == 0.0) ? 1.0 : v + a.diag_c[6 * c + x] * a.inv_radius;
        } else if (x == y) v = a.diag_c[ioff + x - 6] * a.inv_radius;
      }
      v -= sum;
      a.S[(size_t)(sp + x) * a.N + sp + y] = v;
      a.S[(size_t)(sp + y) * a.N + sp + x] = v;
    }
  }
}
// W = 6 + NV lanes per block pair (lane = one row of the W x W block), 64 / W pairs per wavefront; entries summed in list order (deterministic, no atomics),
// the block accumulated into S - or, kStore, written (every pair of images has a list, pp_ba_impl::pairs_complete: S needs no clearing)
template <int NV, bool kStore>
__global__ __launch_bounds__(256) void k_schur_wide_pairs(SchurArgs a, const double* __restrict__ rec, const double* __restrict__ JkS, int64_t num_pairs,
                                                          const int32_t* __restrict__ pair_start, const int32_t* __restrict__ pair_ij,
                                                          const int32_t* __restrict__ pair_entries) {
  constexpr int W = 6 + NV, kSlots = 64 / W;
  const int lane = threadIdx.x & 63;
  const int slot = lane / W, ar = lane % W;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t pr = wave * kSlots + slot;
  if (slot >= kSlots || pr >= num_pairs) return;
  const int bi = pair_ij[2 * pr], bj = pair_ij[2 * pr + 1];
  const int e0 = pair_start[2 * pr], e1 = pair_start[2 * pr + 1];    // (first, last + 1), pairs ordered by list length
  double acc[W];
#pragma unroll
  for (int i = 0; i < W; ++i) acc[i] = 0.0;
  int2 next = e0 < e1 ? *reinterpret_cast<const int2*>(pair_entries + 2 * (size_t)e0) : make_int2(0, 0);
  for (int e = e0; e < e1; ++e) {
    const int2 oo = next;
    if (e + 1 < e1) next = *reinterpret_cast<const int2*>(pair_entries + 2 * (size_t)(e + 1));
    const double2* qi = reinterpret_cast<const double2*>(RecT(rec, (size_t)oo.x));       // T_oi (2x3)
    const double2* qj = reinterpret_cast<const double2*>(RecX(rec, (size_t)oo.y));       // J_pt,oj (2x3)
    const double2* pj = reinterpret_cast<const double2*>(RecJ(rec, (size_t)oo.y));
    const double2* kj = reinterpret_cast<const double2*>(JkS + (size_t)2 * kCamStride * oo.y);
    const double2 t0 = qi[0], t1 = qi[1], t2 = qi[2];
    const double2 x0 = qj[0], x1 = qj[1], x2 = qj[2];
    const double* ri = ar < 6 ? RecJ(rec, (size_t)oo.x) + ar : JkS + (size_t)2 * kCamStride * oo.x + (ar - 6);      // this lane's row of J_w,oi^T: (ri[0], ri[stride])
    const double pi0 = ri[0], pi1 = ri[ar < 6 ? 6 : kCamStride];
    const double g00 = t0.x * x0.x + t0.y * x0.y + t1.x * x1.x, g01 = t0.x * x1.y + t0.y * x2.x + t1.x * x2.y;
    const double g10 = t1.y * x0.x + t2.x * x0.y + t2.y * x1.x, g11 = t1.y * x1.y + t2.x * x2.x + t2.y * x2.y;
    const double h0 = pi0 * g00 + pi1 * g10, h1 = pi0 * g01 + pi1 * g11;
    const double2 j0 = pj[0], j1 = pj[1], j2 = pj[2], j3 = pj[3], j4 = pj[4], j5 = pj[5];   // rows: (j0 j1 j2), (j3 j4 j5)
    acc[0] += h0 * j0.x + h1 * j3.x; acc[1] += h0 * j0.y + h1 * j3.y;
    acc[2] += h0 * j1.x + h1 * j4.x; acc[3] += h0 * j1.y + h1 * j4.y;
    acc[4] += h0 * j2.x + h1 * j5.x; acc[5] += h0 * j2.y + h1 * j5.y;
#pragma unroll
    for (int b = 0; b < NV / 2; ++b) {
      const double2 k0 = kj[b], k1 = kj[kCamStride / 2 + b];
      acc[6 + 2 * b] += h0 * k0.x + h1 * k1.x; acc[7 + 2 * b] += h0 * k0.y + h1 * k1.y;
    }
  }
  double2* dst = reinterpret_cast<double2*>(a.S + (size_t)(a.spos[6 * bi] + ar) * a.N + a.spos[6 * bj]);
#pragma unroll
  for (int b = 0; b < W / 2; ++b) { double2 d = kStore ? make_double2(0.0, 0.0) : dst[b]; d.x -= acc[2 * b]; d.y -= acc[2 * b + 1]; dst[b] = d; }
}

// Few images, many shared points (the mapper's local bundle adjustment: 6 images, src/sfm/incremental_mapper.cc:813-858): fifteen block pairs with
// lists of hundreds of entries, each walked by six lanes - 134 us of a 190 us LM iteration at 6 images / 2004 observations.  When a list is longer
// than 64 entries (pp_ba_create) the lists are cut into chunks of 32: the same six lanes per CHUNK (SchurPairsBody's arithmetic on a sub-range),
// partial blocks to memory, then every block = the sum of its chunks in chunk order (k_schur_chunk_reduce) - deterministic, no atomics.
__device__ __forceinline__ void SchurChunksBody(const double* __restrict__ rec, int num_chunks, const int32_t* __restrict__ chunk, const int32_t* __restrict__ pair_entries,
                                                double* __restrict__ partials, int64_t block) {
  const int lane = threadIdx.x & 63;
  const int slot = lane / 6, ar = lane % 6;
  const int64_t ch = (block * 4 + (threadIdx.x >> 6)) * 10 + slot;
  if (slot >= 10 || ch >= num_chunks) return;
  const int e0 = chunk[3 * ch + 1], e1 = chunk[3 * ch + 2];
  double acc[6] = {0, 0, 0, 0, 0, 0};
  int2 next = e0 < e1 ? *reinterpret_cast<const int2*>(pair_entries + 2 * (size_t)e0) : make_int2(0, 0);
  for (int e = e0; e < e1; ++e) {
    const int2 oo = next;
    if (e + 1 < e1) next = *reinterpret_cast<const int2*>(pair_entries + 2 * (size_t)(e + 1));
    const double2* qi = reinterpret_cast<const double2*>(RecT(rec, (size_t)oo.x));
    const double2* qj = reinterpret_cast<const double2*>(RecX(rec, (size_t)oo.y));
    const double2* pj = reinterpret_cast<const double2*>(RecJ(rec, (size_t)oo.y));
    const double2 t0 = qi[0], t1 = qi[1], t2 = qi[2];
    const double2 x0 = qj[0], x1 = qj[1], x2 = qj[2];
    const double pi0 = RecJ(rec, (size_t)oo.x)[ar], pi1 = RecJ(rec, (size_t)oo.x)[6 + ar];
    const double g00 = t0.x * x0.x + t0.y * x0.y + t1.x * x1.x, g01 = t0.x * x1.y + t0.y * x2.x + t1.x * x2.y;
    const double g10 = t1.y * x0.x + t2.x * x0.y + t2.y * x1.x, g11 = t1.y * x1.y + t2.x * x2.x + t2.y * x2.y;
    const double h0 = pi0 * g00 + pi1 * g10, h1 = pi0 * g01 + pi1 * g11;
    const double2 j0 = pj[0], j1 = pj[1], j2 = pj[2], j3 = pj[3], j4 = pj[4], j5 = pj[5];
    acc[0] += h0 * j0.x + h1 * j3.x; acc[1] += h0 * j0.y + h1 * j3.y;
    acc[2] += h0 * j1.x + h1 * j4.x; acc[3] += h0 * j1.y + h1 * j4.y;
    acc[4] += h0 * j2.x + h1 * j5.x; acc[5] += h0 * j2.y + h1 * j5.y;
  }
  double2* dst = reinterpret_cast<double2*>(partials + 36 * (size_t)chunk[3 * ch] + 6 * ar);      // (the chunk's id: the processing order is pp_ba_create's, see "L2 locality of the chunk kernel")
  dst[0] = make_double2(acc[0], acc[1]); dst[1] = make_double2(acc[2], acc[3]); dst[2] = make_double2(acc[4], acc[5]);
}
// the per-image part (first C workgroups) and the chunks in one launch
__global__ __launch_bounds__(256) void k_schur_self_chunks(SchurArgs a, const double* __restrict__ rec, int num_chunks, const int32_t* __restrict__ chunk,
                                                           const int32_t* __restrict__ pair_entries, double* __restrict__ partials) {
  if ((int)blockIdx.x < a.C) SchurSelfRhsBody(a, rec, blockIdx.x);
  else SchurChunksBody(rec, num_chunks, chunk, pair_entries, partials, (int64_t)blockIdx.x - a.C);
}
// the chunks alone (A/B: PPSFM_BA_CHUNK_SPLIT=1 launches the per-image part as k_schur_self_rhs and the chunks here, at their own register count)
__global__ __launch_bounds__(256) void k_schur_chunks(const double* __restrict__ rec, int num_chunks, const int32_t* __restrict__ chunk,
                                                      const int32_t* __restrict__ pair_entries, double* __restrict__ partials) {
  SchurChunksBody(rec, num_chunks, chunk, pair_entries, partials, (int64_t)blockIdx.x);
}
template <bool kStore>      // kStore: the block is written, not accumulated into (pp_ba_impl::pairs_complete)
__global__ __launch_bounds__(256) void k_schur_chunk_reduce(SchurArgs a, int64_t num_pairs, const int32_t* __restrict__ pair_ij, const int32_t* __restrict__ pair_chunk,
                                                            const double* __restrict__ partials) {
  const int64_t it = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (it >= 36 * num_pairs) return;
  const int64_t pr = it / 36;
  const int el = (int)(it % 36);
  const int c0 = pair_chunk[pr], c1 = pair_chunk[pr + 1];
  if (c0 == c1 && !kStore) return;
  double sum = 0.0;
  int ch = c0;
  for (; ch + 8 <= c1; ch += 8) {      // eight partials in flight, added in chunk order (a load per addition was 0.3 us per chunk: 14 us for the 42 chunks of a 334-entry list)
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = partials[36 * (size_t)(ch + u) + el];
#pragma unroll
    for (int u = 0; u < 8; ++u) sum += v[u];
  }
  for (; ch < c1; ++ch) sum += partials[36 * (size_t)ch + el];
  double* dst = a.S + (size_t)(a.spos[6 * pair_ij[2 * pr]] + el / 6) * a.N + a.spos[6 * pair_ij[2 * pr + 1]] + el % 6;
  *dst = (kStore ? 0.0 : *dst) - sum;
}

// ---- K3c --------------------------------------------------------------------------------------
struct StepArgs {
  int C, P;
  int64_t M;
  const int32_t *pt_start, *pt_obs, *obs_pose, *obs_point;
  const double *Jpose, *Jpoint, *r, *Vinv, *vb, *scale_c, *scale_p, *step_c;
  double* step_p;
  double* partials;
  // variable intrinsics (JkS == nullptr: none): compact scaled Jacobian records, block offsets/widths
  const double* JkS;
  const int32_t *obs_cam, *intr_off, *intr_nv;
  // cand_partials != nullptr (no variable intrinsics): k_model_cost_apply also evaluates the cost AT THE TRIAL POINT - each observation
  // applies the step to its own pose and point (the arithmetic of ApplyStepBody) and evaluates its residual there (k_line_eval<0>'s
  // arithmetic and block sums: the same bits), one launch instead of two
  double* cand_partials = nullptr;
  const double *la = nullptr, *lb = nullptr, *lc = nullptr, *poses = nullptr, *points = nullptr, *intr = nullptr;
  int loss_type = 0;
  double loss_scale = 1.0;
};

// (J_k diag(s)) . step of the intrinsics block of observation o  (two rows)
__device__ __forceinline__ void IntrStepProduct(const StepArgs& a, int64_t o, double* m0, double* m1) {
  if (!a.JkS) return;
  const int k = a.obs_cam[o] >> 4;
  const int off = a.intr_off[k];
  if (off < 0) return;
  const int nv = a.intr_nv[k];
  const double* j = a.JkS + (size_t)2 * kCamStride * o;
  const double* d = a.step_c + 6 * a.C + off;
  for (int c = 0; c < nv; ++c) { *m0 += j[c] * d[c]; *m1 += j[kCamStride + c] * d[c]; }
}

// FOUR lanes per point (lane q takes the point's observations q, q+4, ... in list order, then a fixed two-step butterfly):
// one lane per point walked its ~8 observations as eight dependent gather rounds on 391 wavefronts (19 us)
__global__ __launch_bounds__(256) void k_backsub_points(StepArgs a) {
  const int gid = blockIdx.x * 256 + threadIdx.x;
  const int p = gid >> 2, q = gid & 3;
  double acc[3] = {0, 0, 0};
  if (p < a.P) {
    for (int e = a.pt_start[p] + q; e < a.pt_start[p + 1]; e += 4) {
      const int o = a.pt_obs[e];
      const int c = a.obs_pose[o];
      double jp[12], jx[6];
      LoadJp(a.Jpose, o, jp);
      LoadJx(a.Jpoint, o, jx);
      double m0 = 0.0, m1 = 0.0;
#pragma unroll
      for (int j = 0; j < 6; ++j) { const double d = a.scale_c[6 * c + j] * a.step_c[6 * c + j]; m0 += jp[j] * d; m1 += jp[6 + j] * d; }
      IntrStepProduct(a, o, &m0, &m1);
#pragma unroll
      for (int j = 0; j < 3; ++j) acc[j] += jx[j] * m0 + jx[3 + j] * m1;
    }
  }
#pragma unroll
  for (int j = 0; j < 3; ++j) { acc[j] += __shfl_xor(acc[j], 1, 64); acc[j] += __shfl_xor(acc[j], 2, 64); }
  if (p >= a.P || q != 0) return;
  const double s0 = a.scale_p[3 * p], s1 = a.scale_p[3 * p + 1], s2 = a.scale_p[3 * p + 2];
  const double w0 = s0 * acc[0], w1 = s1 * acc[1], w2 = s2 * acc[2];
  const double* vi = a.Vinv + 6 * (size_t)p;
  a.step_p[3 * (size_t)p + 0] = a.vb[3 * (size_t)p + 0] - (vi[0] * w0 + vi[1] * w1 + vi[2] * w2);
  a.step_p[3 * (size_t)p + 1] = a.vb[3 * (size_t)p + 1] - (vi[1] * w0 + vi[3] * w1 + vi[4] * w2);
  a.step_p[3 * (size_t)p + 2] = a.vb[3 * (size_t)p + 2] - (vi[2] * w0 + vi[4] * w1 + vi[5] * w2);
}

// model_cost_change = -sum_o (J d)_o . (r_o + (J d)_o / 2)     (TrustRegionMinimizer)
// "last block done": every block calls this after its global results are written; exactly one block (the one whose
// increment completes the count) gets true, with the other blocks' results visible to it.  The counter resets itself.
__device__ __forceinline__ bool LastBlockDone(int32_t* counter, int num_blocks) {
  __shared__ int s_last;
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();                                   // release this block's results
    const int t = atomicAdd(counter, 1);
    s_last = (t == num_blocks - 1) ? 1 : 0;
    if (s_last) { *counter = 0; __threadfence(); }     // acquire the others'
  }
  __syncthreads();
  return s_last != 0;
}
// sum of n values by the 256 threads of a block in a fixed order -> *out
// (the values were written by EARLIER kernels: plain loads)
__device__ __forceinline__ void BlockSumTo(const double* __restrict__ v, int n, double* __restrict__ out) {
  __shared__ double sh[256];
  double acc = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) acc += v[i];
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) *out = sh[0];
}

__device__ __forceinline__ void QuatPlus(const double* q, double d0, double d1, double d2, double* out);
// delta = scale * step, never contracted into the addition that applies it: ApplyStepBody (the stored trial point) and ModelCostBody (every
// observation's own copy of it) must produce the same bits
__device__ __forceinline__ double ScaledStep(double scale, double step) {
#pragma clang fp contract(off)
  return scale * step;
}
__device__ __forceinline__ void ModelCostBody(const StepArgs& a, int block) {
  const int64_t o = (int64_t)block * 256 + threadIdx.x;
  double val = 0.0, half_rho = 0.0;
  if (o < a.M) {
    const int c = a.obs_pose[o], p = a.obs_point[o];
    double jp[12], jx[6];
    LoadJp(a.Jpose, (int)o, jp);
    LoadJx(a.Jpoint, (int)o, jx);
    double dc[6], dp[3];
#pragma unroll
    for (int j = 0; j < 6; ++j) dc[j] = ScaledStep(a.scale_c[6 * c + j], a.step_c[6 * c + j]);
#pragma unroll
    for (int j = 0; j < 3; ++j) dp[j] = ScaledStep(a.scale_p[3 * p + j], a.step_p[3 * (size_t)p + j]);
    double m0 = 0.0, m1 = 0.0;
#pragma unroll
    for (int j = 0; j < 6; ++j) { m0 += jp[j] * dc[j]; m1 += jp[6 + j] * dc[j]; }
    IntrStepProduct(a, o, &m0, &m1);
#pragma unroll
    for (int j = 0; j < 3; ++j) { m0 += jx[j] * dp[j]; m1 += jx[3 + j] * dp[j]; }
    val = -(m0 * (a.r[2 * o] + m0 / 2.0) + m1 * (a.r[2 * o + 1] + m1 / 2.0));
    if (a.cand_partials) {      // the residual at this observation's trial pose / point
      const double* pose = a.poses + 7 * (size_t)c;
      double qn[4];
      QuatPlus(pose, dc[0], dc[1], dc[2], qn);
      const double tn[3] = {pose[4] + dc[3], pose[5] + dc[4], pose[6] + dc[5]};
      const double Xn[3] = {a.points[3 * (size_t)p] + dp[0], a.points[3 * (size_t)p + 1] + dp[1], a.points[3 * (size_t)p + 2] + dp[2]};
      const int ck = a.obs_cam[o];
      double res[2];
      LineResidualOnly(ck & 15, a.intr + (size_t)kCamStride * (ck >> 4), qn, tn, Xn, a.la[o], a.lb[o], a.lc[o], res);
      double rho0, rho1;
      LossRho(a.loss_type, a.loss_scale, res[0] * res[0] + res[1] * res[1], &rho0, &rho1);
      half_rho = 0.5 * rho0;
    }
  }
  __shared__ double wsum[4], csum[4];
  val = WaveSum(val);
  if (a.cand_partials) half_rho = WaveSum(half_rho);
  if ((threadIdx.x & 63) == 0) { wsum[threadIdx.x >> 6] = val; csum[threadIdx.x >> 6] = half_rho; }
  __syncthreads();
  if (threadIdx.x == 0) {
    a.partials[block] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
    if (a.cand_partials) a.cand_partials[block] = (csum[0] + csum[1]) + (csum[2] + csum[3]);
  }
}
// second stage (a last-block-done fold was measured SLOWER here: ~800 blocks each paying an agent-scope release fence)
__global__ __launch_bounds__(256) void k_sum(const double* __restrict__ partials, int n, double* __restrict__ out) { BlockSumTo(partials, n, out); }

__device__ __forceinline__ void QuatPlus(const double* q, double d0, double d1, double d2, double* out) {
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

// trial point x (+) delta, delta = scale * step
__device__ __forceinline__ void ApplyStepBody(int i, int C, int P, const double* __restrict__ poses, const double* __restrict__ points,
                                              const double* __restrict__ scale_c, const double* __restrict__ scale_p,
                                              const double* __restrict__ step_c, const double* __restrict__ step_p,
                                              double* __restrict__ poses_c, double* __restrict__ points_c) {
  if (i < C) {
    const double* q = poses + 7 * (size_t)i;
    double d[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) d[j] = ScaledStep(scale_c[6 * i + j], step_c[6 * i + j]);
    double qn[4];
    QuatPlus(q, d[0], d[1], d[2], qn);
    double* o = poses_c + 7 * (size_t)i;
    o[0] = qn[0]; o[1] = qn[1]; o[2] = qn[2]; o[3] = qn[3];
    o[4] = q[4] + d[3]; o[5] = q[5] + d[4]; o[6] = q[6] + d[5];
  }
  if (i < P) {
#pragma unroll
    for (int j = 0; j < 3; ++j) points_c[3 * (size_t)i + j] = points[3 * (size_t)i + j] + ScaledStep(scale_p[3 * i + j], step_p[3 * (size_t)i + j]);
  }
}

// the model cost change of the step (first obs_blocks workgroups, one per 256 observations) and the trial point x (+) d (the
// others): independent of each other, one launch
__global__ __launch_bounds__(256) void k_model_cost_apply(StepArgs a, int obs_blocks, const double* __restrict__ poses, const double* __restrict__ points,
                                                          double* __restrict__ poses_c, double* __restrict__ points_c) {
  if ((int)blockIdx.x < obs_blocks) ModelCostBody(a, blockIdx.x);
  else ApplyStepBody(((int)blockIdx.x - obs_blocks) * 256 + threadIdx.x, a.C, a.P, poses, points, a.scale_c, a.scale_p, a.step_c, a.step_p, poses_c, points_c);
}

// The point step, the model cost change, the trial point and the cost there in ONE pass over the observations (no variable
// intrinsics): four lanes per point as in k_backsub_points; a lane keeps its first two observations' rows in registers between the
// sum that gives the point's step and the terms that need it (further observations of a point are loaded again), so the Jacobian is
// read once where k_backsub_points + k_model_cost_apply read it twice.  Sums: per workgroup in lane / wavefront order, then the norms
// kernel's fixed-order sums - deterministic, a different association than the per-observation kernels' (the costs agree to rounding).
// The last pose_blocks workgroups write the trial poses.
// kIntr (variable intrinsics): every observation's camera-side product also carries (J_k diag(s)) . step of its intrinsics block, and the trial residual is taken with
// the trial intrinsics a.intr (k_apply_intr runs before this kernel then).
template <bool kIntr>
__global__ __launch_bounds__(256) void k_step_points(StepArgs a, int point_blocks, const double* __restrict__ poses, const double* __restrict__ points,
                                                     double* __restrict__ poses_c, double* __restrict__ points_c) {
  if ((int)blockIdx.x >= point_blocks) {
    const int i = ((int)blockIdx.x - point_blocks) * 256 + threadIdx.x;
    if (i < a.C) {
      const double* q = poses + 7 * (size_t)i;
      double d[6];
#pragma unroll
      for (int j = 0; j < 6; ++j) d[j] = ScaledStep(a.scale_c[6 * i + j], a.step_c[6 * i + j]);
      double qn[4];
      QuatPlus(q, d[0], d[1], d[2], qn);
      double* o = poses_c + 7 * (size_t)i;
      o[0] = qn[0]; o[1] = qn[1]; o[2] = qn[2]; o[3] = qn[3];
      o[4] = q[4] + d[3]; o[5] = q[5] + d[4]; o[6] = q[6] + d[5];
    }
    return;
  }
  const int gid = blockIdx.x * 256 + threadIdx.x;
  const int p = gid >> 2, q = gid & 3;
  const bool live = p < a.P;
  struct Row { double jp[12], jx[6], dc[6], u0, u1; int o, c; };
  Row R[2];
  auto load = [&](int o, Row& w) {
    w.o = o; w.c = a.obs_pose[o];
    LoadJp(a.Jpose, o, w.jp);
    LoadJx(a.Jpoint, o, w.jx);
    w.u0 = 0.0; w.u1 = 0.0;
#pragma unroll
    for (int j = 0; j < 6; ++j) { w.dc[j] = ScaledStep(a.scale_c[6 * w.c + j], a.step_c[6 * w.c + j]); w.u0 += w.jp[j] * w.dc[j]; w.u1 += w.jp[6 + j] * w.dc[j]; }
    if (kIntr) IntrStepProduct(a, o, &w.u0, &w.u1);
  };
  double acc[3] = {0.0, 0.0, 0.0};
  const int begin = live ? a.pt_start[p] + q : 0, end = live ? a.pt_start[p + 1] : 0;
  bool have[2] = {false, false};
  if (begin < end) {      // the first two observations of this lane: both index loads and then both rows in flight together
    have[0] = true; have[1] = begin + 4 < end;
    const int o0 = a.pt_obs[begin], o1 = a.pt_obs[have[1] ? begin + 4 : begin];
    load(o0, R[0]);
    load(o1, R[1]);
#pragma unroll
    for (int j = 0; j < 3; ++j) acc[j] += R[0].jx[j] * R[0].u0 + R[0].jx[3 + j] * R[0].u1;
    if (have[1]) {
#pragma unroll
      for (int j = 0; j < 3; ++j) acc[j] += R[1].jx[j] * R[1].u0 + R[1].jx[3 + j] * R[1].u1;
    }
  }
  for (int e = begin + 8; e < end; e += 4) {
    Row w;
    load(a.pt_obs[e], w);
#pragma unroll
    for (int j = 0; j < 3; ++j) acc[j] += w.jx[j] * w.u0 + w.jx[3 + j] * w.u1;
  }
#pragma unroll
  for (int j = 0; j < 3; ++j) { acc[j] += __shfl_xor(acc[j], 1, 64); acc[j] += __shfl_xor(acc[j], 2, 64); }
  double dp[3] = {0.0, 0.0, 0.0}, Xn[3] = {0.0, 0.0, 0.0};
  if (live) {
    const double s0 = a.scale_p[3 * p], s1 = a.scale_p[3 * p + 1], s2 = a.scale_p[3 * p + 2];
    const double w0 = s0 * acc[0], w1 = s1 * acc[1], w2 = s2 * acc[2];
    const double* vi = a.Vinv + 6 * (size_t)p;
    const double e0 = a.vb[3 * (size_t)p + 0] - (vi[0] * w0 + vi[1] * w1 + vi[2] * w2);
    const double e1 = a.vb[3 * (size_t)p + 1] - (vi[1] * w0 + vi[3] * w1 + vi[4] * w2);
    const double e2 = a.vb[3 * (size_t)p + 2] - (vi[2] * w0 + vi[4] * w1 + vi[5] * w2);
    dp[0] = ScaledStep(s0, e0); dp[1] = ScaledStep(s1, e1); dp[2] = ScaledStep(s2, e2);
#pragma unroll
    for (int j = 0; j < 3; ++j) Xn[j] = points[3 * (size_t)p + j] + dp[j];
    if (q == 0) {
      a.step_p[3 * (size_t)p + 0] = e0; a.step_p[3 * (size_t)p + 1] = e1; a.step_p[3 * (size_t)p + 2] = e2;
#pragma unroll
      for (int j = 0; j < 3; ++j) points_c[3 * (size_t)p + j] = Xn[j];
    }
  }
  double val = 0.0, half_rho = 0.0;
  auto terms = [&](const Row& w) {
    double m0 = w.u0, m1 = w.u1;
#pragma unroll
    for (int j = 0; j < 3; ++j) { m0 += w.jx[j] * dp[j]; m1 += w.jx[3 + j] * dp[j]; }
    val -= m0 * (a.r[2 * (size_t)w.o] + m0 / 2.0) + m1 * (a.r[2 * (size_t)w.o + 1] + m1 / 2.0);
    const double* pose = poses + 7 * (size_t)w.c;
    double qn[4];
    QuatPlus(pose, w.dc[0], w.dc[1], w.dc[2], qn);
    const double tn[3] = {pose[4] + w.dc[3], pose[5] + w.dc[4], pose[6] + w.dc[5]};
    const int ck = a.obs_cam[w.o];
    double res[2];
    LineResidualOnly(ck & 15, a.intr + (size_t)kCamStride * (ck >> 4), qn, tn, Xn, a.la[w.o], a.lb[w.o], a.lc[w.o], res);
    double rho0, rho1;
    LossRho(a.loss_type, a.loss_scale, res[0] * res[0] + res[1] * res[1], &rho0, &rho1);
    half_rho += 0.5 * rho0;
  };
  if (have[0]) terms(R[0]);
  if (have[1]) terms(R[1]);
  for (int e = begin + 8; e < end; e += 4) {
    Row w;
    load(a.pt_obs[e], w);
    terms(w);
  }
  __shared__ double wsum[4], csum[4];
  val = WaveSum(val);
  half_rho = WaveSum(half_rho);
  if ((threadIdx.x & 63) == 0) { wsum[threadIdx.x >> 6] = val; csum[threadIdx.x >> 6] = half_rho; }
  __syncthreads();
  if (threadIdx.x == 0) {
    a.partials[blockIdx.x] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
    a.cand_partials[blockIdx.x] = (csum[0] + csum[1]) + (csum[2] + csum[3]);
  }
}

// trial intrinsics: variable parameters move by scale * step, the others are copied
__global__ __launch_bounds__(256) void k_apply_intr(int K, int C, const int32_t* __restrict__ intr_off, const int32_t* __restrict__ intr_col,
                                                    const double* __restrict__ intr, const double* __restrict__ scale_c, const double* __restrict__ step_c,
                                                    double* __restrict__ intr_c) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= K * kCamStride) return;
  const int k = i / kCamStride;
  const int col = intr_col[i], off = intr_off[k];
  double v = intr[i];
  if (off >= 0 && col >= 0) v += scale_c[6 * C + off + col] * step_c[6 * C + off + col];
  intr_c[i] = v;
}

// intrinsics part of the three norms, folded into the scalars by one lane (NI is small): gradient max-norm over the
// variable columns (Euclidean parameters: |g|), |delta|^2, and |x|^2 over every parameter of a variable block
constexpr int kNormsIntrThreads = 1024;
__global__ __launch_bounds__(kNormsIntrThreads) void k_norms_intr(int K, int C, const int32_t* __restrict__ intr_off, const int32_t* __restrict__ intr_nv,
                                                   const int32_t* __restrict__ camera_model_np, const double* __restrict__ intr, const double* __restrict__ gc,
                                                   const double* __restrict__ scale_c, const double* __restrict__ step_c, double* __restrict__ scal, int count_norms,
                                                   double* __restrict__ host_out, unsigned long long ticket) {
  // one workgroup of sixteen wavefronts, thread t the cameras t, t + 1024, ..: fixed order (a single thread walking 1100 cameras with their dependent loads
  // took 350 us, one wavefront 14.5 us at 500 cameras: a camera per image puts this kernel twice into every LM iteration)
  __shared__ double wmax[kNormsIntrThreads / 64], wst[kNormsIntrThreads / 64], wxn[kNormsIntrThreads / 64];
  double gmax = 0.0, st = 0.0, xn = 0.0;
  for (int k = threadIdx.x; k < K; k += kNormsIntrThreads) {
    const int off = intr_off[k];
    if (off < 0) continue;
    for (int j = 0; j < intr_nv[k]; ++j) {
      gmax = fmax(gmax, fabs(gc[6 * C + off + j]));
      if (step_c && count_norms) { const double d = scale_c[6 * C + off + j] * step_c[6 * C + off + j]; st += d * d; }
    }
    if (count_norms) for (int j = 0; j < camera_model_np[k]; ++j) xn += intr[k * kCamStride + j] * intr[k * kCamStride + j];
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) gmax = fmax(gmax, __shfl_xor(gmax, off, 64));
  st = WaveSum(st); xn = WaveSum(xn);
  if ((threadIdx.x & 63) == 0) { wmax[threadIdx.x >> 6] = gmax; wst[threadIdx.x >> 6] = st; wxn[threadIdx.x >> 6] = xn; }
  __syncthreads();
  if (threadIdx.x == 0) {
    gmax = 0.0; st = 0.0; xn = 0.0;
    for (int w = 0; w < kNormsIntrThreads / 64; ++w) { gmax = fmax(gmax, wmax[w]); st += wst[w]; xn += wxn[w]; }      // (wavefront order: fixed)
    scal[kGradMax] = fmax(scal[kGradMax], gmax); scal[kStepNorm2] += st; scal[kXNorm2] += xn; __threadfence();
  }
  __syncthreads();
  if (threadIdx.x >= 64) return;
  // with variable intrinsics THIS kernel is the last one to touch the scalars: it hands them to the host's pinned slot (and the ticket the host
  // polls) the way k_norms_partial does without them
  if (host_out) {
    const int b = threadIdx.x;
    if (b < kNumScalars && b != kTicketSlot) {
      const unsigned long long v = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(scal) + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(reinterpret_cast<unsigned long long*>(host_out) + b, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (ticket != 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");        // system scope
      if (b == 0) __hip_atomic_store(reinterpret_cast<unsigned long long*>(host_out) + kTicketSlot, ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// gradient max norm (Ceres 2.x: ||x - Plus(x, -g)||_inf), |delta|^2, |x|^2: per-block partials + a final block
__global__ __launch_bounds__(256) void k_norms_partial(int C, int P, const double* __restrict__ poses, const double* __restrict__ points,
                                                       const double* __restrict__ gc, const double* __restrict__ gp, const double* __restrict__ scale_c,
                                                       const double* __restrict__ scale_p, const double* __restrict__ step_c,
                                                       const double* __restrict__ step_p, double* __restrict__ part, int32_t* __restrict__ done_counter,
                                                       double* __restrict__ scal, int norm_blocks, const double* __restrict__ sum0, int n0,
                                                       double* __restrict__ out0, const double* __restrict__ sum1, int n1, double* __restrict__ out1,
                                                       double* __restrict__ host_out, unsigned long long ticket, int count_pose_norms, int pose_blocks) {
  __shared__ double smax[256], sstep[256], sx[256];
  // second stages folded in (one launch each saved): the cost partials of the evaluation before this kernel and the model-cost
  // partials of the trial step are summed by two workgroups of their own, beside the norms (inside the last norm block they
  // were ~2 us each on the critical path of the launch)
  if ((int)blockIdx.x >= norm_blocks) {
    if ((int)blockIdx.x == norm_blocks) BlockSumTo(sum0, n0, out0);
    else BlockSumTo(sum1, n1, out1);
  } else {
    // the LAST pose_blocks of the norm blocks take the poses (a quaternion Plus each: an fp64 sin / cos, 1.8 us on the two workgroups that
    // also had their share of the points), the others the points (every load unconditional: one round trip per pass instead of two)
    double gmax = 0.0, st = 0.0, xn = 0.0;
    const int point_blocks = norm_blocks - pose_blocks;
    const bool pose_role = (int)blockIdx.x >= point_blocks;
    const int stride = (pose_role ? pose_blocks : point_blocks) * 256, t0 = ((int)blockIdx.x - (pose_role ? point_blocks : 0)) * 256 + threadIdx.x;
    for (int c = t0; c < C && pose_role; c += stride) {
      const double* q = poses + 7 * (size_t)c;
      if (scale_c[6 * c] != 0.0) {
        double qn[4];
        QuatPlus(q, -gc[6 * (size_t)c], -gc[6 * (size_t)c + 1], -gc[6 * (size_t)c + 2], qn);
#pragma unroll
        for (int j = 0; j < 4; ++j) gmax = fmax(gmax, fabs(q[j] - qn[j]));
        if (count_pose_norms) {      // (a point-sharded group: the poses are replicated, their part of |x|^2 and |step|^2 is counted on rank 0 only)
#pragma unroll
          for (int j = 0; j < 7; ++j) xn += q[j] * q[j];
        }
      }
#pragma unroll
      for (int j = 3; j < 6; ++j) if (scale_c[6 * c + j] != 0.0) gmax = fmax(gmax, fabs(gc[6 * (size_t)c + j]));
      if (step_c && count_pose_norms) {
#pragma unroll
        for (int j = 0; j < 6; ++j) { const double d = scale_c[6 * c + j] * step_c[6 * c + j]; st += d * d; }
      }
    }
    for (int i = t0; i < 3 * P && !pose_role; i += stride) {
      const double sp = scale_p[i], gv = gp[i], xv = points[i], sv = step_p ? step_p[i] : 0.0;
      if (sp != 0.0) { gmax = fmax(gmax, fabs(gv)); xn += xv * xv; }
      const double d = sp * sv;
      st += d * d;
    }
    smax[threadIdx.x] = gmax; sstep[threadIdx.x] = st; sx[threadIdx.x] = xn;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if ((int)threadIdx.x < s) {
        smax[threadIdx.x] = fmax(smax[threadIdx.x], smax[threadIdx.x + s]);
        sstep[threadIdx.x] += sstep[threadIdx.x + s];
        sx[threadIdx.x] += sx[threadIdx.x + s];
      }
      __syncthreads();
    }
    if (threadIdx.x == 0) { part[3 * blockIdx.x] = smax[0]; part[3 * blockIdx.x + 1] = sstep[0]; part[3 * blockIdx.x + 2] = sx[0]; }
  }
  // the block that finishes last (of ALL blocks) combines the per-block norm partials (fixed order: block b -> slot b, then the
  // same tree) and, when asked to, hands the scalars to the host: it writes them into the pinned host slot itself.  The
  // hipMemcpyAsync this replaces cost ~4 us of copy engine plus ~5 us before the next kernel could start.
  if (LastBlockDone(done_counter, (int)gridDim.x)) {
    const int b = threadIdx.x;
    const bool in = b < norm_blocks;
    smax[b] = in ? __hip_atomic_load(part + 3 * b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
    sstep[b] = in ? __hip_atomic_load(part + 3 * b + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
    sx[b] = in ? __hip_atomic_load(part + 3 * b + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if (b < s) { smax[b] = fmax(smax[b], smax[b + s]); sstep[b] += sstep[b + s]; sx[b] += sx[b + s]; }
      __syncthreads();
    }
    if (b == 0) {
      __hip_atomic_store(scal + kGradMax, smax[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(scal + kStepNorm2, sstep[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(scal + kXNorm2, sx[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (host_out) {
      __syncthreads();
      if (b < kNumScalars) {      // the three norms from LDS, the other slots (sums of other blocks / earlier kernels, the flag) from memory
        unsigned long long v;
        if (b == kGradMax) v = __double_as_longlong(smax[0]);
        else if (b == kStepNorm2) v = __double_as_longlong(sstep[0]);
        else if (b == kXNorm2) v = __double_as_longlong(sx[0]);
        else v = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(scal) + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (b != kTicketSlot) __hip_atomic_store(reinterpret_cast<unsigned long long*>(host_out) + b, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      // the ticket goes last (the host polls it instead of waiting on an event: an event record between this kernel and the
      // next one was a ~5 us hole in the stream)
      if (ticket != 0 && b < 64) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");        // system scope
        if (b == 0) __hip_atomic_store(reinterpret_cast<unsigned long long*>(host_out) + kTicketSlot, ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
}

// ---- host driver ----------------------------------------------------------------------------------
static bool InGroup(const pp_ba_impl* h) { return h->comm != nullptr || h->allreduce != nullptr; }
// The tile pattern of a block-sparse reduced system is built from THIS rank's pair lists.  In a point-sharded group the all-reduced
// system has the union of every rank's co-visibility: a tile whose shared points all live on another rank is non-zero after the
// exchange, and a rank that skipped it would factor a different matrix than its peers (replicated poses diverging between ranks).
// So the block-sparse path is only taken outside a group; attaching / detaching a communicator or callback switches it.
// (inside a point-sharded group only when the tile map is the group's: built from the union co-visibility every rank was given, pp_ba_problem_desc::covisibility)
static bool SparseActive(const pp_ba_impl* h) { return h->sparse_tiles && (!InGroup(h) || h->structure_from_covisibility); }
// (re)binds the factorisation's launch structure to the handle's current state: the tile map (or none), the solved-tile array of the
// one-launch mode (allocated only when that mode can run: N x N doubles, 7 GB at 5000 images), the per-size device lists
static int ApplyLinearSolverStructure(pp_ba_impl* h) {
  if (!h->S || h->iterative) return PP_OK;      // EnsureSolverBuffers calls this once the buffers exist; an iterative handle has no factorisation
  // the setters that end up here (pp_ba_set_communicator / pp_ba_set_allreduce) may be called, after a solve, from a host thread whose
  // current device is another one: the allocations below belong on the handle's device
  PP_HIP_TRY(hipSetDevice(h->device));
  std::lock_guard<std::recursive_mutex> setup_lock(DeviceSetupMutex());
  ppsfm::CholeskyAux* aux = &h->chol_aux;
  const uint8_t* want = SparseActive(h) ? h->tile_nz.data() : nullptr;
  if (aux->tile_nz != want) {
    PP_HIP_TRY(hipStreamSynchronize(h->stream));
    if (aux->graph_exec) { (void)hipGraphExecDestroy(aux->graph_exec); aux->graph_exec = nullptr; }
    if (aux->sparse_lists) { (void)hipFree(aux->sparse_lists); aux->sparse_lists = nullptr; }
    if (aux->sparse_nz) { (void)hipFree(aux->sparse_nz); aux->sparse_nz = nullptr; }
    aux->sparse_T = 0;
    aux->tile_nz = want; aux->tile_T = want ? h->N / 64 : 0;
    // whatever an earlier factorisation left outside the tiles the new structure rewrites
    PP_HIP_TRY(hipMemsetAsync(h->S, 0, sizeof(double) * (size_t)h->N * h->N, h->stream));
  }
  if (!h->Lfac && CholeskyWantsFactorArray(aux, h->N)) { const int rc = HandleAlloc(&h->Lfac, (size_t)h->N * h->N); if (rc) return rc; }
  return CholeskyPrepare(aux, h->N, h->Lfac != nullptr, h->stream);
}

static int EnsureSolverBuffers(pp_ba_impl* h) {
  if (h->S || h->pcg_state) return PP_OK;
  std::lock_guard<std::recursive_mutex> setup_lock(DeviceSetupMutex());      // (allocations: not beside another host thread's graph capture)
  const int C = h->C, P = h->P;
  h->N = ((h->n_red + 1 + 63) / 64) * 64;
  int rc;
#define A(ptr, n) if ((rc = HandleAlloc(&h->ptr, (size_t)(n)))) return rc
  A(U, 36 * (size_t)C); A(gc, (size_t)h->n_red); A(V, 6 * (size_t)P); A(gp, 3 * (size_t)P); A(Vinv, 6 * (size_t)P); A(vb, 3 * (size_t)P);
  A(scale_c, (size_t)h->n_red); A(scale_p, 3 * (size_t)P); A(diag_c, (size_t)h->n_red); A(diag_p, 3 * (size_t)P);
  if (!h->iterative) { A(S, (size_t)h->N * h->N); A(Linv, CholeskyWorkspaceDoubles(h->N)); }
  A(JpS, kRecStride * (size_t)h->M); A(norm_part, 3 * 256); A(step_c, (size_t)h->N); A(step_p, 3 * (size_t)P);
  if (!h->spos_identity) { A(step_s, (size_t)h->N); }
#undef A
  for (int i = 0; i < 8; ++i) if ((rc = PoolEventAcquire(&h->tev[i], true))) return rc;
  for (int i = 0; i < 2; ++i) if ((rc = PoolEventAcquire(&h->tev_eval[i], true))) return rc;
  if ((rc = PoolEventAcquire(&h->ev_readback, false))) return rc;
  if (h->iterative) return PcgEnsureBuffers(h);
  if ((rc = CholeskyAuxCreate(&h->chol_aux))) return rc;
  PP_HIP_TRY(hipMemsetAsync(h->S, 0, sizeof(double) * (size_t)h->N * h->N, h->stream));
  if (h->sparse_tiles) {      // the factorisation and the assembly skip the tiles that stay zero
    const int T = h->N / 64;
    std::vector<int32_t> list;
    for (int i = 0; i < T; ++i) for (int j = 0; j <= i; ++j) if (h->tile_nz[(size_t)i * T + j]) { list.push_back(i); list.push_back(j); }
    if ((rc = HandleAlloc(&h->nz_tile_list, list.size()))) return rc;
    PP_HIP_TRY(hipMemcpyAsync(h->nz_tile_list, list.data(), sizeof(int32_t) * list.size(), hipMemcpyHostToDevice, h->stream));
    PP_HIP_TRY(hipStreamSynchronize(h->stream));
  }
  return ApplyLinearSolverStructure(h);
}

__global__ __launch_bounds__(256) void k_gather_step(int n, const int32_t* __restrict__ spos, const double* __restrict__ x, double* __restrict__ step) {
  const int v = blockIdx.x * 256 + threadIdx.x;
  if (v < n) step[v] = x[spos[v]];
}

static SchurArgs MakeSchurArgs(pp_ba_impl* h, double radius) {
  SchurArgs a;
  a.C = h->C; a.N = h->N; a.rhs_row = h->n_red;
  a.pose_start = h->pose_start; a.pose_obs = h->pose_obs; a.obs_point = h->obs_point;
  a.Jpose = h->Jpose; a.Jpoint = h->Jpoint; a.U = h->U; a.gc = h->gc; a.Vinv = h->Vinv; a.vb = h->vb;
  a.scale_c = h->scale_c; a.scale_p = h->scale_p; a.diag_c = h->diag_c;
  a.inv_radius = 1.0 / radius; a.S = h->S; a.add_diagonal = h->group_rank == 0 ? 1 : 0;
  a.spos = h->spos;
  return a;
}
static StepArgs MakeStepArgs(pp_ba_impl* h) {
  StepArgs a;
  a.C = h->C; a.P = h->P; a.M = h->M;
  a.pt_start = h->pt_start; a.pt_obs = h->pt_obs; a.obs_pose = h->obs_pose; a.obs_point = h->obs_point;
  a.Jpose = h->Jpose; a.Jpoint = h->Jpoint; a.r = h->r; a.Vinv = h->Vinv; a.vb = h->vb;
  a.scale_c = h->scale_c; a.scale_p = h->scale_p; a.step_c = h->step_c; a.step_p = h->step_p; a.partials = h->partials + h->partials_stride;   // second region: K1's partials stay valid
  a.JkS = h->NI > 0 ? h->JkS_intr : nullptr; a.obs_cam = h->obs_cam; a.intr_off = h->intr_off; a.intr_nv = h->intr_nv;
  return a;
}

static int GroupReduce(pp_ba_impl* h, double* ptr, int64_t count, int op) {
  if (h->comm) return CommAllReduce(h->comm, ptr, count, op, h->stream);      // stream-ordered: no host synchronisation
  if (!h->allreduce) return PP_OK;
  PP_HIP_TRY(hipStreamSynchronize(h->stream));
  const int rc = h->allreduce(h->allreduce_ctx, ptr, count, op);
  if (rc) { SetLastError("allreduce callback returned %d", rc); return PP_ERR_INVALID; }
  return PP_OK;
}
bool BaInGroup(const pp_ba_impl* h) { return InGroup(h); }
int BaGroupReduce(pp_ba_impl* h, double* ptr, int64_t count, int op) { return GroupReduce(h, ptr, count, op); }
// several GroupReduce calls of one exchange become ONE RCCL launch (no-ops with a host callback)
static int GroupBegin(pp_ba_impl* h) { return h->comm ? CommGroupStart() : PP_OK; }
static int GroupEnd(pp_ba_impl* h) { return h->comm ? CommGroupEnd() : PP_OK; }
// one exchange = GroupBegin .. GroupEnd; an error return in between must not leave the ncclGroupStart open (the communicator
// would be unusable for every later call): the scope closes it on every path
struct GroupScope {
  pp_ba_impl* h; bool open = false;
  explicit GroupScope(pp_ba_impl* hh) : h(hh) {}
  int Begin() { const int rc = GroupBegin(h); open = rc == PP_OK; return rc; }
  int End() { open = false; return GroupEnd(h); }
  ~GroupScope() { if (open) (void)GroupEnd(h); }
};
// The failure bits (d_flag[0]: 1 = pivot, 2 = point block, 4 = a bounded wait of the one-launch factorisation ran out) are rank-local,
// the decisions they drive (invalid step, retry with per-column launches) issue collectives: every rank of a group must take the same
// one.  The bits travel as a double through the unused device slot kTicketSlot, MAX-reduced with the scalars of the trial step.
__global__ void k_flag_to_scalar(const int32_t* __restrict__ flag, double* __restrict__ slot) { *slot = (double)(*flag & 7); }
__global__ void k_scalar_to_flag(const double* __restrict__ slot, int32_t* __restrict__ flag) { const int v = (int)*slot; if (v) atomicOr(flag, v); }

// lower triangle + rhs row of S (rows 0 .. n_red, row r = r + 1 entries) <-> a contiguous buffer: the group exchange moves
// (n+1)(n+2)/2 doubles (36 MB at 500 images) instead of the (n+1) x N rectangle (72 MB)
__global__ __launch_bounds__(256) void k_pack_lower(const double* __restrict__ S, int N, int rows, double* __restrict__ packed, int unpack, double* __restrict__ Sout) {
  const int r = blockIdx.y;
  const size_t base = (size_t)r * (r + 1) / 2;
  for (int c = blockIdx.x * 256 + threadIdx.x; c <= r && r < rows; c += gridDim.x * 256) {
    if (unpack) Sout[(size_t)r * N + c] = packed[base + c];
    else packed[base + c] = S[(size_t)r * N + c];
  }
}

// the same for a block-sparse system (a group whose ranks were created with the union co-visibility: the same tile map everywhere): only the non-zero
// 64 x 64 tiles travel - 12 MB instead of 36 at banded cfg 3 ("exploit block sparsity of S when cameras don't co-observe")
__global__ __launch_bounds__(256) void k_pack_tiles(const double* __restrict__ S, int N, const int32_t* __restrict__ tiles, double* __restrict__ packed, int unpack, double* __restrict__ Sout) {
  const int ti = tiles[2 * blockIdx.x], tj = tiles[2 * blockIdx.x + 1];
  double2* pk = reinterpret_cast<double2*>(packed + (size_t)blockIdx.x * 64 * 64);
  for (int idx = threadIdx.x; idx < 64 * 32; idx += 256) {
    const size_t at = ((size_t)ti * 64 + (idx >> 5)) * N + (size_t)tj * 64 + 2 * (idx & 31);
    if (unpack) *reinterpret_cast<double2*>(Sout + at) = pk[idx];
    else pk[idx] = *reinterpret_cast<const double2*>(S + at);
  }
}

// K1 (Jacobian) + K2 at the current parameters; leaves cost in scal[kCost]
// fold_cost: the cost partials are summed by the LaunchNorms call that follows (fold = 1) instead of a kernel of their own
static int EvaluateAndReduce(pp_ba_impl* h, bool fold_cost = false) {
  hipStream_t s = h->stream;
  int rc = LaunchEval(h, 0, h->NI > 0 ? 1 : 0, true, h->poses, h->points, fold_cost ? nullptr : h->scal + kCost, /*compact_cam=*/true);
  if (rc) return rc;
  hipLaunchKernelGGL(k_reduce, dim3(h->C + CeilDiv(4 * (int64_t)h->P, 256)), dim3(256), 0, s, h->C, h->P, h->pose_start, h->pose_obs, h->pt_start, h->pt_obs, h->Jpose,
                     h->Jpoint, h->r, h->U, h->gc, h->V, h->gp);
  PP_HIP_TRY(hipGetLastError());
  if ((rc = IntrSumsAfterEval(h))) return rc;
  if (InGroup(h)) {      // the per-pose blocks (and the intrinsics sums) of all shards: one exchange
    GroupScope g(h);
    if ((rc = g.Begin())) return rc;
    if ((rc = GroupReduce(h, h->U, 36 * (int64_t)h->C, PP_REDUCE_SUM))) return rc;
    if ((rc = GroupReduce(h, h->gc, (int64_t)h->n_red, PP_REDUCE_SUM))) return rc;
    if (h->NI > 0 && (rc = GroupReduce(h, h->cnI, (int64_t)h->NI, PP_REDUCE_SUM))) return rc;
    if (!fold_cost && (rc = GroupReduce(h, h->scal + kCost, 1, PP_REDUCE_SUM))) return rc;
    if ((rc = g.End())) return rc;
  }
  return PP_OK;
}

// fold: 0 nothing; 1 K1's cost partials -> scal[kCost]; 2 K1's cost partials -> scal[kCostCand] and the model-cost partials
// -> scal[kModelChange] (the trial step)
static int LaunchNorms(pp_ba_impl* h, bool with_step, int fold = 0, double* host_slot = nullptr, unsigned long long ticket = 0) {
  const int pose_blocks = std::min(8, CeilDiv(h->C, 256));
  const int nblk = 64 + pose_blocks;      // (<= 256: the last block combines one partial per thread; norm_part holds 3 x 256)
  const double* model_partials = h->partials + h->partials_stride;
  const int n_trial = fold == 2 && h->trial_partials > 0 ? h->trial_partials : h->num_partials;      // (k_step_points sums per point workgroup)
  hipLaunchKernelGGL(k_norms_partial, dim3(nblk + fold), dim3(256), 0, h->stream, h->C, h->P, h->poses, h->points, h->gc, h->gp, h->scale_c,
                     h->scale_p, with_step ? h->step_c : nullptr, with_step ? h->step_p : nullptr, h->norm_part, h->d_flag + 2, h->scal, nblk,
                     fold ? h->partials : nullptr, fold == 2 ? n_trial : h->num_partials, fold == 2 ? h->scal + kCostCand : h->scal + kCost,
                     fold == 2 ? model_partials : nullptr, n_trial, h->scal + kModelChange, h->NI > 0 ? nullptr : host_slot, h->NI > 0 ? 0ull : ticket,
                     h->group_rank == 0 ? 1 : 0, pose_blocks);
  if (h->NI > 0)
    hipLaunchKernelGGL(k_norms_intr, dim3(1), dim3(kNormsIntrThreads), 0, h->stream, h->K, h->C, h->intr_off, h->intr_nv, h->cam_np, h->intr, h->gc, h->scale_c,
                       with_step ? h->step_c : nullptr, h->scal, h->group_rank == 0 ? 1 : 0, host_slot, ticket);
  PP_HIP_TRY(hipGetLastError());
  if (InGroup(h)) {
    // one exchange for the scalars of this evaluation: the gradient max norm as a max; |step|^2 and |x|^2 as sums (the point
    // parts are sharded, the replicated pose / intrinsics parts were counted on rank 0 only: parameter_tolerance sees the
    // same global norms on every rank); the sums this call folded (cost, or candidate cost + model cost change) as sums
    int rc;
    if (with_step) hipLaunchKernelGGL(k_flag_to_scalar, dim3(1), dim3(1), 0, h->stream, h->d_flag, h->scal + kTicketSlot);
    {
      GroupScope g(h);
      if ((rc = g.Begin())) return rc;
      if ((rc = GroupReduce(h, h->scal + kGradMax, 1, PP_REDUCE_MAX))) return rc;
      if ((rc = GroupReduce(h, h->scal + kStepNorm2, 2, PP_REDUCE_SUM))) return rc;          // kStepNorm2, kXNorm2 are adjacent
      if (fold == 1 && (rc = GroupReduce(h, h->scal + kCost, 1, PP_REDUCE_SUM))) return rc;
      if (fold == 2 && (rc = GroupReduce(h, h->scal + kCostCand, 2, PP_REDUCE_SUM))) return rc;   // kCostCand, kModelChange are adjacent
      if (with_step && (rc = GroupReduce(h, h->scal + kTicketSlot, 1, PP_REDUCE_MAX))) return rc;    // the failure bits: every rank sees the worst
      if ((rc = g.End())) return rc;
    }
    if (with_step) hipLaunchKernelGGL(k_scalar_to_flag, dim3(1), dim3(1), 0, h->stream, h->scal + kTicketSlot, h->d_flag);
    PP_HIP_TRY(hipGetLastError());
  }
  return PP_OK;
}

// assemble the damped reduced system for `radius` into S (lower triangle + rhs row)
static int AssembleReducedSystem(pp_ba_impl* h, double radius, bool refresh_diagonal = false, double dmin = 0.0, double dmax = 0.0) {
  hipStream_t s = h->stream;
  // The factorisation overwrites S with L (fill-in included), so blocks without a pair list must be cleared again;
  // when every block has one (dense scenes), the assembly kernels rewrite the whole lower triangle and the padding
  // rows keep their zeros (cleared once at allocation): no 72 MB clear, no read-modify-write in k_schur_pairs.
  // (variable intrinsics: beside their images' pose columns they are part of the images' blocks, all of which are rewritten (k_schur_wide_*); behind the pose
  // columns - the vectors' order - their rows are cleared, a few rows, and the pose part is stored as without them; other layouts clear S)
  const bool store_blocks = h->pairs_complete && !h->iterative && !InGroup(h) && (h->NI == 0 || h->intr_wide_nv > 0 || h->spos_identity);
  const int zero_tiles = (!store_blocks && !h->iterative && SparseActive(h)) ? h->num_nz_tiles : 0;      // (only the tiles anything is written to: k_prepare's third role)
  if (!store_blocks && !h->iterative && !zero_tiles) PP_HIP_TRY(hipMemsetAsync(h->S, 0, sizeof(double) * (size_t)h->N * h->N, s));
  if (store_blocks && h->NI > 0 && h->spos_identity) PP_HIP_TRY(hipMemsetAsync(h->S + (size_t)6 * h->C * h->N, 0, sizeof(double) * (size_t)h->NI * h->N, s));
  {      // (V + D^2 / radius)^-1, V^-1 b_p per point and the per-observation records: one launch
    const int point_blocks = CeilDiv(refresh_diagonal ? std::max(h->P, 6 * h->C) : h->P, 256);
    const dim3 grid(point_blocks + h->num_partials + zero_tiles);
    if (refresh_diagonal)
      hipLaunchKernelGGL(k_prepare<true>, grid, dim3(256), 0, s, point_blocks, h->P, h->V, h->gp, h->scale_p, h->diag_p, h->point_const, 1.0 / radius, h->Vinv, h->vb,
                         h->d_flag, h->C, h->U, h->scale_c, dmin, dmax, h->diag_c, h->M, h->obs_pose, h->obs_point, h->Jpose, h->Jpoint, h->JpS, h->num_partials, h->S, h->N,
                         (const int32_t*)h->nz_tile_list);
    else
      hipLaunchKernelGGL(k_prepare<false>, grid, dim3(256), 0, s, point_blocks, h->P, h->V, h->gp, h->scale_p, h->diag_p, h->point_const, 1.0 / radius, h->Vinv, h->vb,
                         h->d_flag, h->C, h->U, h->scale_c, dmin, dmax, h->diag_c, h->M, h->obs_pose, h->obs_point, h->Jpose, h->Jpoint, h->JpS, h->num_partials, h->S, h->N,
                         (const int32_t*)h->nz_tile_list);
  }
  SchurArgs a = MakeSchurArgs(h, radius);
  if (h->iterative) { a.Sd = h->pcg_Sd; a.rhs_out = h->pcg_b; }
  if (h->iterative) {      // the diagonal blocks (preconditioner) and the reduced right-hand side; S itself is applied from the records
    hipLaunchKernelGGL(k_schur_self_rhs, dim3(h->C), dim3(256), 0, s, a, h->JpS);
    PP_HIP_TRY(hipGetLastError());
    if (InGroup(h)) {      // every shard's part of the diagonal blocks and of the right-hand side (rank 0 carries U + D^2 and -g)
      GroupScope g(h);
      int rc;
      if ((rc = g.Begin())) return rc;
      if ((rc = GroupReduce(h, h->pcg_Sd, 36 * (int64_t)h->C, PP_REDUCE_SUM))) return rc;
      if ((rc = GroupReduce(h, h->pcg_b, 6 * (int64_t)h->C, PP_REDUCE_SUM))) return rc;
      if ((rc = g.End())) return rc;
    }
    { const int rc = IntrAssemble(h, 1.0 / radius, a.add_diagonal); if (rc) return rc; }      // (variable intrinsics: their diagonal blocks and their part of the right-hand side)
    if (InGroup(h) && h->NI > 0) {      // the per-camera sums of every shard: the compact diagonal blocks (the preconditioner's) and the intrinsics rows of the right-hand side
      GroupScope g(h);
      int rc;
      if ((rc = g.Begin())) return rc;
      if ((rc = GroupReduce(h, h->pcg_Scomp, 12 * (int64_t)h->NI, PP_REDUCE_SUM))) return rc;
      if ((rc = GroupReduce(h, h->pcg_b + 6 * (size_t)h->C, (int64_t)h->NI, PP_REDUCE_SUM))) return rc;
      if ((rc = g.End())) return rc;
    }
    return PP_OK;
  }
  if (h->pairs_chunked && h->num_pairs > 0) {      // long lists (few images, many shared points): chunks of the lists, then the blocks from their chunks
    static const bool split = []() { const char* e = std::getenv("PPSFM_BA_CHUNK_SPLIT"); return e && std::atoi(e) != 0; }();
    if (split) {
      hipLaunchKernelGGL(k_schur_chunks, dim3(CeilDiv(h->small_num_chunks, 40)), dim3(256), 0, s, h->JpS, h->small_num_chunks, h->small_chunk, h->pair_entries, h->small_partials);
      hipLaunchKernelGGL(k_schur_self_rhs, dim3(h->C), dim3(256), 0, s, a, h->JpS);
    } else
    hipLaunchKernelGGL(k_schur_self_chunks, dim3(h->C + CeilDiv(h->small_num_chunks, 40)), dim3(256), 0, s, a, h->JpS, h->small_num_chunks, h->small_chunk, h->pair_entries,
                       h->small_partials);
    const dim3 grid(CeilDiv(36 * h->num_pairs, 256));
    if (store_blocks) hipLaunchKernelGGL(k_schur_chunk_reduce<true>, grid, dim3(256), 0, s, a, h->num_pairs, h->pair_ij, h->small_pair_chunk, h->small_partials);
    else hipLaunchKernelGGL(k_schur_chunk_reduce<false>, grid, dim3(256), 0, s, a, h->num_pairs, h->pair_ij, h->small_pair_chunk, h->small_partials);
  } else if (h->intr_wide_nv > 0) {      // every image's 6 + n_v columns as one block: the pose gather with wider rows
    { const int rc = IntrScaledJacobians(h); if (rc) return rc; }
    const int W = 6 + h->intr_wide_nv;
    const dim3 gp((unsigned)CeilDiv(h->num_pairs, (int64_t)(4 * (64 / W))));
#define PP_WIDE(NV) do { \
      hipLaunchKernelGGL(k_schur_wide_self<NV>, dim3(h->C), dim3(256), 0, s, a, h->JpS, h->JkS_intr, h->pose_camera, h->intr_off); \
      if (h->num_pairs > 0) { \
        if (store_blocks) hipLaunchKernelGGL((k_schur_wide_pairs<NV, true>), gp, dim3(256), 0, s, a, h->JpS, h->JkS_intr, h->num_pairs, h->pair_start, h->pair_ij, h->pair_entries); \
        else hipLaunchKernelGGL((k_schur_wide_pairs<NV, false>), gp, dim3(256), 0, s, a, h->JpS, h->JkS_intr, h->num_pairs, h->pair_start, h->pair_ij, h->pair_entries); } } while (0)
    switch (h->intr_wide_nv) { case 2: PP_WIDE(2); break; case 4: PP_WIDE(4); break; case 6: PP_WIDE(6); break; default: PP_WIDE(8); break; }
#undef PP_WIDE
  } else if (store_blocks && h->num_pairs > 0) {
    hipLaunchKernelGGL(k_schur_blocks, dim3(h->C + CeilDiv(h->num_pairs, 40)), dim3(256), 0, s, a, h->JpS, h->num_pairs, h->pair_start, h->pair_ij,
                       h->pair_entries);
  } else {
    hipLaunchKernelGGL(k_schur_self_rhs, dim3(h->C), dim3(256), 0, s, a, h->JpS);
    if (h->num_pairs > 0)
      hipLaunchKernelGGL(k_schur_pairs<false>, dim3(CeilDiv(h->num_pairs, 40)), dim3(256), 0, s, a, h->JpS, h->num_pairs, h->pair_start, h->pair_ij,
                         h->pair_entries);
  }
  PP_HIP_TRY(hipGetLastError());
  { const int rc = IntrAssemble(h, 1.0 / radius, a.add_diagonal); if (rc) return rc; }
  if (InGroup(h) && SparseActive(h) && h->nz_tile_list && h->num_nz_tiles > 0) {      // the non-zero tiles, packed (every rank has the group's tile map)
    const int64_t count = (int64_t)h->num_nz_tiles * 64 * 64;
    if (h->Spack_cap < count) { if (h->Spack) PoolDeviceFree(h->Spack); h->Spack = nullptr; h->Spack_cap = 0; const int rc = HandleAlloc(&h->Spack, (size_t)count); if (rc) return rc; h->Spack_cap = count; }
    hipLaunchKernelGGL(k_pack_tiles, dim3(h->num_nz_tiles), dim3(256), 0, s, h->S, h->N, (const int32_t*)h->nz_tile_list, h->Spack, 0, (double*)nullptr);
    const int rc = GroupReduce(h, h->Spack, count, PP_REDUCE_SUM);
    if (rc) return rc;
    hipLaunchKernelGGL(k_pack_tiles, dim3(h->num_nz_tiles), dim3(256), 0, s, (const double*)nullptr, h->N, (const int32_t*)h->nz_tile_list, h->Spack, 1, h->S);
    PP_HIP_TRY(hipGetLastError());
  } else if (InGroup(h)) {      // lower triangle + rhs row, packed: half the bytes of the rectangle on the wire (RCCL or the host callback)
    const int rows = h->n_red + 1;
    const int64_t count = (int64_t)rows * (rows + 1) / 2;
    if (h->Spack_cap < count) { if (h->Spack) PoolDeviceFree(h->Spack); h->Spack = nullptr; h->Spack_cap = 0; const int rc = HandleAlloc(&h->Spack, (size_t)count); if (rc) return rc; h->Spack_cap = count; }
    const dim3 grid(std::max(1, std::min(64, CeilDiv(rows, 256))), rows);
    hipLaunchKernelGGL(k_pack_lower, grid, dim3(256), 0, s, h->S, h->N, rows, h->Spack, 0, (double*)nullptr);
    const int rc = GroupReduce(h, h->Spack, count, PP_REDUCE_SUM);
    if (rc) return rc;
    hipLaunchKernelGGL(k_pack_lower, grid, dim3(256), 0, s, (const double*)nullptr, h->N, rows, h->Spack, 1, h->S);
    PP_HIP_TRY(hipGetLastError());
  }
  return PP_OK;
}

static int ReadScalars(pp_ba_impl* h) {
  PP_HIP_TRY(hipMemcpyAsync(h->h_scal, h->scal, sizeof(double) * kNumScalars, hipMemcpyDeviceToHost, h->stream));   // the flag rides in the last slot
  PP_HIP_TRY(hipStreamSynchronize(h->stream));
  return PP_OK;
}
// The norms kernel of a trial step writes the scalars and then a ticket into the pinned slot; the host polls the ticket.
// The poll is a busy spin only for as long as a cfg-3 sized iteration lasts (PPSFM_TICKET_SPIN_US, default 1500 us, 0 = never
// spin): with larger reduced systems (C ~ 5000: ~1 s per factorisation) a spinning caller thread per handle would burn a host
// core per GPU, so after that the thread sleeps between polls (50 us naps: < 0.1 % of such an iteration).  Bounded: after ~2 s
// without the ticket the stream is synchronised once and the ticket re-checked (a failed launch would otherwise wait forever).
static int WaitTicket(pp_ba_impl* h, unsigned long long ticket) {
  static const long spin_us = []() { const char* e = std::getenv("PPSFM_TICKET_SPIN_US"); return e ? std::atol(e) : 1500L; }();
  const volatile unsigned long long* t = reinterpret_cast<const volatile unsigned long long*>(h->h_scal) + kTicketSlot;
  const auto t0 = std::chrono::steady_clock::now();
  bool synced = false;
  for (unsigned spins = 0; *t != ticket; ++spins) {
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
    __builtin_ia32_pause();
#endif
    if ((spins & 0xFF) != 0xFF) continue;
    const auto waited = std::chrono::steady_clock::now() - t0;
    if (waited > std::chrono::seconds(2) && !synced) {
      PP_HIP_TRY(hipStreamSynchronize(h->stream));
      synced = true;
      if (*t != ticket) { SetLastError("pp_ba_solve: the trial step's scalars never arrived"); return PP_ERR_HIP; }
      break;
    }
    if (waited > std::chrono::microseconds(spin_us)) std::this_thread::sleep_for(std::chrono::microseconds(50));
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  return PP_OK;
}
static int32_t HostFlag(const pp_ba_impl* h) { int32_t f; std::memcpy(&f, h->h_scal + kNumScalars - 1, sizeof(f)); return f; }

// pp_ba_options::phase_timings: HIP events between the phases of an iteration.  Off by default: every record is a barrier
// packet on the stream (~5 us each, ~7 per iteration measured on MI355X).
struct PhaseTimer {
  pp_ba_impl* h; bool on; int n = 0; int phase[8];
  PhaseTimer(pp_ba_impl* hh, bool enabled) : h(hh), on(enabled) { if (on) (void)hipEventRecord(h->tev[0], h->stream); }
  void Mark(int ph) { if (on && n < 7) { phase[n] = ph; ++n; (void)hipEventRecord(h->tev[n], h->stream); } }
  void Collect() {
    if (!on) return;
    if (n > 0) (void)hipEventSynchronize(h->tev[n]);     // the host may be ahead of the last mark (it polls a ticket, not the stream)
    for (int i = 0; i < n; ++i) {
      float ms = 0;
      if (hipEventElapsedTime(&ms, h->tev[i], h->tev[i + 1]) == hipSuccess) { h->timings_ms[phase[i]] += ms; h->timing_calls[phase[i]] += 1; }
    }
    n = 0;
  }
};

}  // namespace ppsfm

using namespace ppsfm;

extern "C" {

// What every rank of a point-sharded group must agree on before it exchanges anything: the layout of the reduced system (order, width, tile map) decides
// the COUNT of every all-reduce of a solve (num_nz_tiles x 4096 doubles or the packed triangle) - ranks that disagree hang in RCCL or sum mismatched tiles.
// 52 bits of an FNV-1a hash, i.e. exactly representable as the double that travels.
static double GroupStructureHash(const pp_ba_impl* h) {
  uint64_t x = 1469598103934665603ull;
  auto mix = [&](const void* p, size_t n) { const uint8_t* b = static_cast<const uint8_t*>(p); for (size_t i = 0; i < n; ++i) { x ^= b[i]; x *= 1099511628211ull; } };
  const int32_t head[8] = {h->C, h->n_red, h->N, h->NI, h->structure_from_covisibility ? 1 : 0, (h->sparse_tiles && h->structure_from_covisibility) ? 1 : 0,
                           (h->sparse_tiles && h->structure_from_covisibility) ? h->num_nz_tiles : 0, h->iterative ? 1 : 0};
  mix(head, sizeof(head));
  if (h->sparse_tiles && h->structure_from_covisibility) mix(h->tile_nz.data(), h->tile_nz.size());      // (outside that case a group factorises the dense system: the map is not used)
  mix(h->pose_new_of_old.data(), h->pose_new_of_old.size() * sizeof(int32_t));
  return (double)(x >> 12);
}

// The attach calls are COLLECTIVE over the group: one MAX all-reduce of three doubles through the exchange that is about to be attached - [refusal of any
// rank, structure hash, -structure hash] - so that the ranks refuse TOGETHER (one returning an error while the others enter a collective is a hang) and a
// group whose handles lay out the reduced system differently is refused before its first exchange.  A group of one rank exchanges nothing.
static int GroupAgreeOnRefusal(pp_ba_impl* h, int* bad, int* mismatch, pp_allreduce_fn fn, void* ctx, pp_comm_handle comm, int group_size) {
  *mismatch = 0;
  if (group_size <= 1) return PP_OK;
  PP_HIP_TRY(hipSetDevice(h->device));
  const double hash = GroupStructureHash(h);
  double v[3] = {*bad ? 1.0 : 0.0, hash, -hash};
  if (!h->attach_slot) { const int rc = HandleAlloc(&h->attach_slot, 4); if (rc) return rc; }      // (a slot of its own: nothing of a solve lives here)
  double* slot = h->attach_slot;
  PP_HIP_TRY(hipMemcpyAsync(slot, v, sizeof(v), hipMemcpyHostToDevice, h->stream));
  PP_HIP_TRY(hipStreamSynchronize(h->stream));
  int rc = PP_OK;
  if (comm) rc = pp_comm_allreduce(comm, slot, 3, PP_REDUCE_MAX);
  else if (fn) { rc = fn(ctx, slot, 3, PP_REDUCE_MAX); if (rc) { SetLastError("pp_ba_set_allreduce: the reduction callback returned %d", rc); rc = PP_ERR_HIP; } }
  if (rc) return rc;
  PP_HIP_TRY(hipMemcpy(v, slot, sizeof(v), hipMemcpyDeviceToHost));
  *bad = v[0] != 0.0 ? 1 : 0;
  *mismatch = (v[1] != -v[2]) ? 1 : 0;      // max(hash) != min(hash)
  return PP_OK;
}
#define PP_GROUP_MISMATCH_TEXT "the handles of this group do not lay out the reduced camera system alike (image count, variable intrinsics, internal image order or " \
                               "the block-sparse tile map differ between ranks - e.g. one rank was created without pp_ba_problem_desc::covisibility or with a stale one): " \
                               "every rank of a point-sharded group passes the same order and, with PP_ORDERING_AUTO, the group's union co-visibility"

int pp_ba_set_allreduce(pp_ba_handle h, pp_allreduce_fn fn, void* ctx, int32_t group_rank, int32_t group_size) try {
  PP_REQUIRE(h, "pp_ba_set_allreduce: null handle");
  PP_REQUIRE(group_size >= 1 && group_rank >= 0 && group_rank < group_size, "pp_ba_set_allreduce: bad group");
  if (fn) {
    // every rank must lay out the exchanged system alike: the caller's order, or an order taken from the co-visibility every rank was given.  The verdict is
    // exchanged before anybody refuses, so that the ranks of a group fail TOGETHER instead of one returning an error while the others enter a collective
    int bad = (!h->pose_new_of_old.empty() && !h->structure_from_covisibility) ? 1 : 0, mismatch = 0;
    const int rc = GroupAgreeOnRefusal(h, &bad, &mismatch, fn, ctx, nullptr, group_size);
    if (rc) return rc;
    PP_REQUIRE(bad || !mismatch, "pp_ba_set_allreduce: " PP_GROUP_MISMATCH_TEXT);
    PP_REQUIRE(!bad, "pp_ba_set_allreduce: a handle of this group renumbered its images from its own shard's co-visibility (pp_ba_problem_desc::ordering = AUTO "
               "without ::covisibility); the handles of a point-sharded group keep the caller's order (PP_ORDERING_NATURAL) or are all created with the group's union "
               "co-visibility, so that every rank lays out the exchanged system alike");
  }
  h->allreduce = fn; h->allreduce_ctx = ctx; h->comm = nullptr;
  h->group_rank = fn ? group_rank : 0; h->group_size = fn ? group_size : 1;
  // a host callback is where other host threads do device-wide things (allocate, synchronize) while this handle would be capturing
  // its factorisation graph, and the callback's own synchronisation per LM iteration dwarfs what the graph saves: enqueue eagerly
  if (fn) {
    h->chol_aux.use_graph = false;
    if (h->chol_aux.graph_exec) { (void)hipGraphExecDestroy(h->chol_aux.graph_exec); h->chol_aux.graph_exec = nullptr; }
  }
  return ApplyLinearSolverStructure(h);      // (a block-sparse tile map made from the shard's own observations is rank-local: not used inside a group)
} PP_API_CATCH("pp_ba_set_allreduce")

int pp_ba_set_communicator(pp_ba_handle h, pp_comm_handle comm) try {
  PP_REQUIRE(h, "pp_ba_set_communicator: null handle");
  PP_REQUIRE(!comm || comm->device == h->device, "pp_ba_set_communicator: the communicator lives on device %d, the handle on device %d", comm ? comm->device : -1, h->device);
  if (comm) {
    int bad = (!h->pose_new_of_old.empty() && !h->structure_from_covisibility) ? 1 : 0, mismatch = 0;      // (see pp_ba_set_allreduce: the ranks of a group refuse together)
    const int rc = GroupAgreeOnRefusal(h, &bad, &mismatch, nullptr, nullptr, comm, comm->size);
    if (rc) return rc;
    PP_REQUIRE(bad || !mismatch, "pp_ba_set_communicator: " PP_GROUP_MISMATCH_TEXT);
    PP_REQUIRE(!bad, "pp_ba_set_communicator: a handle of this group renumbered its images from its own shard's co-visibility (pp_ba_problem_desc::ordering = AUTO "
               "without ::covisibility); the handles of a point-sharded group keep the caller's order (PP_ORDERING_NATURAL) or are all created with the group's union "
               "co-visibility, so that every rank lays out the exchanged system alike");
  }
  h->comm = comm; h->allreduce = nullptr; h->allreduce_ctx = nullptr;
  h->group_rank = comm ? comm->rank : 0; h->group_size = comm ? comm->size : 1;
  return ApplyLinearSolverStructure(h);
} PP_API_CATCH("pp_ba_set_communicator")

int pp_ba_get_structure(pp_ba_handle h, int32_t* info) try {
  PP_REQUIRE(h && info, "pp_ba_get_structure: null argument");
  const int T = ((h->n_red + 1 + 63) / 64);
  info[0] = T * (T + 1) / 2;
  info[1] = h->nnz_tiles_natural >= 0 ? h->nnz_tiles_natural : h->num_nz_tiles;
  info[2] = h->num_nz_tiles;
  info[3] = h->pose_new_of_old.empty() ? 0 : 1;
  info[4] = SparseActive(h) ? 1 : 0;
  info[5] = h->iterative ? 1 : 0;
  // the chains of the one-launch factorisation (several: a nested-dissection order whose sub-trees are factorised side by side) and its chain steps
  info[6] = 1; info[7] = T;
  if (SparseActive(h) && T >= 4 && T <= 128 && !h->tile_nz.empty()) {
    if (h->structure_steps < 0) h->structure_steps = CholeskyChainSteps(T, h->tile_nz.data(), &h->structure_chains);      // (planned once per handle; the plan itself is cached per tile map)
    info[6] = h->structure_chains; info[7] = h->structure_steps;
  }
  return PP_OK;
} PP_API_CATCH("pp_ba_get_structure")

int pp_ba_get_trace(pp_ba_handle h, double* trace, int32_t capacity_rows, int32_t* num_rows) try {
  PP_REQUIRE(h && num_rows, "pp_ba_get_trace: null argument");
  const int rows = (int)(h->trace.size() / 7);
  *num_rows = rows;
  if (trace) std::memcpy(trace, h->trace.data(), sizeof(double) * 7 * (size_t)std::min(rows, capacity_rows));
  return PP_OK;
} PP_API_CATCH("pp_ba_get_trace")

int pp_ba_get_timings(pp_ba_handle h, double* ms, int32_t* calls) try {
  PP_REQUIRE(h && ms && calls, "pp_ba_get_timings: null argument");
  for (int i = 0; i < PP_BA_T_COUNT; ++i) { ms[i] = h->timings_ms[i]; calls[i] = h->timing_calls[i]; }
  return PP_OK;
} PP_API_CATCH("pp_ba_get_timings")

int pp_ba_reduced_system(pp_ba_handle h, const pp_ba_options* o, double radius, int32_t* n_out, double* S, double* rhs, int64_t capacity) try {
  PP_REQUIRE(h && o && n_out && radius > 0, "pp_ba_reduced_system: bad argument");
  PP_REQUIRE(!h->iterative, "pp_ba_reduced_system: an iterative (ITERATIVE_SCHUR) handle never forms the reduced system - create the handle with "
             "pp_ba_problem_desc::linear_solver = PP_LINEAR_SOLVER_DIRECT (or PPSFM_BA_LINEAR_SOLVER=direct) for the direct solve");
  PP_HIP_TRY(hipSetDevice(h->device));
  int rc;
  if ((rc = BaEnsureJacobianBuffers(h, 0, h->NI > 0 ? 1 : 0))) return rc;
  if ((rc = EnsureSolverBuffers(h))) return rc;
  PP_HIP_TRY(hipMemsetAsync(h->d_flag, 0, sizeof(int32_t), h->stream));
  if ((rc = EvaluateAndReduce(h))) return rc;
  const int grid = CeilDiv(std::max<int64_t>(6 * (int64_t)h->C, 3 * (int64_t)h->P), 256);
  hipLaunchKernelGGL(k_jacobi_scale, dim3(grid), dim3(256), 0, h->stream, h->C, h->P, h->U, h->V, h->pose_const, h->tvec_mask, h->point_const,
                     o->jacobi_scaling, h->scale_c, h->scale_p);
  hipLaunchKernelGGL(k_lm_diagonal, dim3(grid), dim3(256), 0, h->stream, h->C, h->P, h->U, h->V, h->scale_c, h->scale_p, o->min_lm_diagonal,
                     o->max_lm_diagonal, h->diag_c, h->diag_p);
  if ((rc = IntrScale(h, o->jacobi_scaling))) return rc;
  if ((rc = IntrDiagonal(h, o->min_lm_diagonal, o->max_lm_diagonal))) return rc;
  if ((rc = AssembleReducedSystem(h, radius))) return rc;
  const int n = h->n_red;
  *n_out = n;
  if (S) {
    PP_REQUIRE(capacity >= (int64_t)n * n, "pp_ba_reduced_system: capacity %lld < %lld", (long long)capacity, (long long)n * n);
    std::vector<double> full((size_t)h->N * h->N);
    PP_HIP_TRY(hipMemcpyAsync(full.data(), h->S, sizeof(double) * full.size(), hipMemcpyDeviceToHost, h->stream));
    PP_HIP_TRY(hipStreamSynchronize(h->stream));
    // (the caller's image order: column c of the output sits at internal position at(c) when pp_ba_create renumbered the images)
    // and the position of a vector column in S: pp_ba_impl::spos)
    const bool perm = !h->pose_new_of_old.empty();
    auto at = [&](int c) { const int v = (perm && c < 6 * h->C) ? 6 * h->pose_new_of_old[c / 6] + c % 6 : c; return h->spos_host.empty() ? v : h->spos_host[v]; };
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < n; ++j) {
        const int a = at(i), b = at(j);
        S[(size_t)i * n + j] = (b <= a) ? full[(size_t)a * h->N + b] : full[(size_t)b * h->N + a];
      }
    if (rhs) for (int j = 0; j < n; ++j) rhs[j] = full[(size_t)n * h->N + at(j)];
  }
  return PP_OK;
} PP_API_CATCH("pp_ba_reduced_system")

int pp_ba_solve(pp_ba_handle h, const pp_ba_options* o, pp_ba_summary* sum) try {
  PP_REQUIRE(h && o && sum, "pp_ba_solve: null argument");
  PP_REQUIRE(o->max_num_iterations >= 0 && o->initial_trust_region_radius > 0, "pp_ba_solve: bad options");
  PP_HIP_TRY(hipSetDevice(h->device));
  const auto t_start = std::chrono::steady_clock::now();
  std::memset(sum, 0, sizeof(*sum));
  int rc;
  if ((rc = BaEnsureJacobianBuffers(h, 0, h->NI > 0 ? 1 : 0))) return rc;
  if ((rc = EnsureSolverBuffers(h))) return rc;
  hipStream_t s = h->stream;
  h->trace.clear();
  h->linear_solver_iterations = 0;
  for (int i = 0; i < PP_BA_T_COUNT; ++i) { h->timings_ms[i] = 0; h->timing_calls[i] = 0; }
  PP_HIP_TRY(hipMemsetAsync(h->d_flag, 0, 4 * sizeof(int32_t), s));     // failure bits, the Cholesky token, the norms kernel's block counter
  PP_HIP_TRY(hipMemsetAsync(h->step_c, 0, sizeof(double) * h->N, s));
  PP_HIP_TRY(hipEventRecord(h->ev0, s));
  const bool phase_timings = o->phase_timings != 0;
  PhaseTimer timer(h, phase_timings);

  const int grid_cp = CeilDiv(std::max<int64_t>(6 * (int64_t)h->C, 3 * (int64_t)h->P), 256);
  const int grid_obs = h->num_partials;

  // iteration 0: evaluate, Jacobi scale, gradient norm
  const bool fold = h->allreduce == nullptr;     // (a host-callback all-reduce needs the sums before the norms kernel; RCCL reduces what the norms kernel folded)
  if ((rc = EvaluateAndReduce(h, fold))) return rc;
  timer.Mark(PP_BA_T_EVAL);
  hipLaunchKernelGGL(k_jacobi_scale, dim3(grid_cp), dim3(256), 0, s, h->C, h->P, h->U, h->V, h->pose_const, h->tvec_mask, h->point_const,
                     o->jacobi_scaling, h->scale_c, h->scale_p);
  if ((rc = IntrScale(h, o->jacobi_scaling))) return rc;
  // the scalars reach the host without the copy engine when nothing else touches them after the norms kernel (see below)
  const bool direct0 = !InGroup(h) && h->h_scal_dev != nullptr;
  if (direct0) {
    const unsigned long long ticket0 = ++h->ticket_seq;
    if ((rc = LaunchNorms(h, false, fold ? 1 : 0, h->h_scal_dev, ticket0))) return rc;
    if ((rc = WaitTicket(h, ticket0))) return rc;
  } else {
    if ((rc = LaunchNorms(h, false, fold ? 1 : 0))) return rc;
    if ((rc = ReadScalars(h))) return rc;
  }
  timer.Collect();
  double cost = h->h_scal[kCost], gmax = h->h_scal[kGradMax];
  sum->initial_cost = cost;
  sum->num_residuals = (int32_t)(2 * h->M);
  double radius = o->initial_trust_region_radius, decrease_factor = 2.0;
  bool reuse_diagonal = false, last_successful = true;
  int invalid = 0;
  auto push = [&](double c, double dc, double g, double sn, double rel, double rad, int ok) {
    const double row[7] = {c, dc, g, sn, rel, rad, (double)ok};
    h->trace.insert(h->trace.end(), row, row + 7);
  };
  push(cost, 0, gmax, 0, 0, radius, 1);
  sum->termination = PP_TERM_NO_CONVERGENCE;
  if (!std::isfinite(cost)) { sum->termination = PP_TERM_FAILURE; SetLastError("pp_ba_solve: initial cost is not finite"); }
  // ceres::IterationCallback: called with the last trace row; true = the callback ended the solve (termination set)
  int cb_iteration = 0;
  auto user_callback = [&]() -> bool {
    if (!o->iteration_callback) return false;
    const double* row = h->trace.data() + h->trace.size() - 7;
    pp_ba_iteration_summary it;
    it.iteration = cb_iteration++; it.step_is_successful = row[6] != 0.0;
    it.cost = row[0]; it.cost_change = row[1]; it.gradient_max_norm = row[2]; it.step_norm = row[3]; it.relative_decrease = row[4];
    it.trust_region_radius = row[5];
    const int32_t r = o->iteration_callback(o->iteration_callback_ctx, &it);
    if (r == PP_SOLVER_ABORT) { sum->termination = PP_TERM_USER_FAILURE; return true; }
    if (r == PP_SOLVER_TERMINATE_SUCCESSFULLY) { sum->termination = PP_TERM_USER_SUCCESS; return true; }
    return false;
  };
  bool user_stop = sum->termination != PP_TERM_FAILURE && user_callback();

  // After an accepted step the evaluation at the new point (cost, gradient max-norm) is only ENQUEUED: its two scalars
  // are first needed after the next trial step's own read-back, so a successful iteration synchronises with the
  // host once, not twice.  The gradient-tolerance test is applied when they arrive; if it fires, the trial step that
  // was computed speculatively is simply dropped (the trial point lives in separate buffers), so the sequence of
  // accepted points and the termination are exactly those of the eager loop.
  bool pending = false;
  double* h_eval = h->h_scal + kNumScalars;     // one of two slots (the next evaluation is enqueued before this one is consumed)
  int eval_slot = 0;
  // Speculative acceptance.  Almost every trial step is accepted, and the accept decision needs a host round trip
  // (~30 us of idle GPU, after which the host has to catch up launching ~15 kernels).  So the accept path — make the
  // candidate the current point, evaluate + reduce there, norms, read-back — is enqueued BEFORE the host waits for the
  // trial step's scalars; the wait is on an event recorded right after their copy, not on the stream.  The candidate
  // becomes current by swapping buffer pointers (no copies).  If the step turns out rejected / invalid, the pointers are
  // swapped back and the evaluation at the old point is enqueued again (its Jacobians were overwritten): the sequence of
  // accepted points, costs and radii is that of the eager loop.  Not used with a group all-reduce (host callbacks).
  const bool speculate = h->allreduce == nullptr;
  // the norms kernel hands the scalars to the pinned host slot itself (no copy-engine hop) when nothing else touches them
  // after it: no group all-reduce, no intrinsics norms kernel
  const bool direct = speculate && !InGroup(h) && h->h_scal_dev != nullptr;
  auto swap_points = [&]() {
    std::swap(h->poses, h->poses_c); std::swap(h->points, h->points_c);
    if (h->NI > 0) std::swap(h->intr, h->intr_c);
  };
  auto enqueue_evaluation = [&]() -> int {      // at h->poses / h->points: K1 + K2, norms, scalars -> the free host slot
    int r;
    if (phase_timings) PP_HIP_TRY(hipEventRecord(h->tev_eval[0], s));
    if ((r = EvaluateAndReduce(h, fold))) return r;
    if (phase_timings) PP_HIP_TRY(hipEventRecord(h->tev_eval[1], s));
    eval_slot ^= 1;
    if ((r = LaunchNorms(h, false, fold ? 1 : 0, direct ? h->h_scal_dev + kNumScalars * (1 + eval_slot) : nullptr))) return r;
    if (!direct) PP_HIP_TRY(hipMemcpyAsync(h->h_scal + kNumScalars * (1 + eval_slot), h->scal, sizeof(double) * kNumScalars, hipMemcpyDeviceToHost, s));
    return PP_OK;
  };
  auto resolve = [&]() {   // requires the stream to be synchronised past the enqueued evaluation
    cost = h_eval[kCost]; gmax = h_eval[kGradMax];
    double* row = h->trace.data() + h->trace.size() - 7;
    row[0] = cost; row[2] = gmax;
    float ms = 0;
    if (phase_timings && hipEventElapsedTime(&ms, h->tev_eval[0], h->tev_eval[1]) == hipSuccess) { h->timings_ms[PP_BA_T_EVAL] += ms; h->timing_calls[PP_BA_T_EVAL] += 1; }
    pending = false;
  };
  // (variable intrinsics take the fused step kernel - k_step_points<true>, the trial intrinsics applied before it - and keep the separate kernels otherwise)
  const bool fused_step_allowed = !(getenv("PPSFM_BA_FUSED_STEP") && atoi(getenv("PPSFM_BA_FUSED_STEP")) == 0);
  const bool fused_trial_cost = fold && (h->NI == 0 || fused_step_allowed) && !(getenv("PPSFM_BA_FUSED_TRIAL_COST") && atoi(getenv("PPSFM_BA_FUSED_TRIAL_COST")) == 0);
  const bool fused_step = fused_trial_cost && fused_step_allowed;
  for (int iter = 1; sum->termination != PP_TERM_FAILURE && !user_stop; ++iter) {
    if (pending && (iter > o->max_num_iterations || radius < o->min_trust_region_radius)) {
      PP_HIP_TRY(hipStreamSynchronize(s));
      resolve();
    }
    if (!pending && last_successful && gmax <= o->gradient_tolerance) { sum->termination = PP_TERM_CONVERGENCE; break; }
    if (iter > o->max_num_iterations) { sum->termination = PP_TERM_NO_CONVERGENCE; break; }
    if (radius < o->min_trust_region_radius) { sum->termination = PP_TERM_CONVERGENCE; break; }

    PhaseTimer t2(h, phase_timings);
    if (!reuse_diagonal && (rc = IntrDiagonal(h, o->min_lm_diagonal, o->max_lm_diagonal))) return rc;
    if ((rc = AssembleReducedSystem(h, radius, !reuse_diagonal, o->min_lm_diagonal, o->max_lm_diagonal))) return rc;
    t2.Mark(PP_BA_T_SCHUR);
    if (h->iterative) {
      int cg = 0;
      if ((rc = PcgSolve(h, radius, o->max_linear_solver_iterations, o->eta, &cg))) return rc;
      h->linear_solver_iterations += cg;
    } else {
      // (the solution comes out in the reduced system's column order: the vectors' order unless the intrinsics sit beside their images' pose columns)
      if ((rc = CholeskySolveAugmented(h->S, h->N, h->n_red, h->Linv, h->Lfac, h->spos_identity ? h->step_c : h->step_s, h->d_flag, s, &h->chol_aux))) return rc;
      if (!h->spos_identity) hipLaunchKernelGGL(k_gather_step, dim3(CeilDiv(h->n_red, 256)), dim3(256), 0, s, h->n_red, h->spos, h->step_s, h->step_c);
    }
    t2.Mark(PP_BA_T_CHOLESKY);
    reuse_diagonal = true;
    StepArgs sa = MakeStepArgs(h);
    // the cost at the trial point inside k_model_cost_apply (one launch less) whenever its partials are summed by the norms kernel anyway
    if (fused_trial_cost) {
      sa.cand_partials = h->partials; sa.la = h->la; sa.lb = h->lb; sa.lc = h->lc; sa.poses = h->poses; sa.points = h->points; sa.intr = h->intr;
      sa.loss_type = h->loss_type; sa.loss_scale = h->loss_scale;
    }
    h->trial_partials = 0;
    if (fused_step) {      // point steps, model cost change, trial point and its cost: one pass over the observations
      const int point_blocks = CeilDiv(4 * (int64_t)h->P, 256);
      h->trial_partials = point_blocks;
      if (h->NI > 0) {
        hipLaunchKernelGGL(k_apply_intr, dim3(CeilDiv(h->K * kCamStride, 256)), dim3(256), 0, s, h->K, h->C, h->intr_off, h->intr_col, h->intr, h->scale_c, h->step_c, h->intr_c);
        sa.intr = h->intr_c;      // (the trial residuals are taken with the trial intrinsics)
        hipLaunchKernelGGL(k_step_points<true>, dim3(point_blocks + CeilDiv(h->C, 256)), dim3(256), 0, s, sa, point_blocks, h->poses, h->points, h->poses_c, h->points_c);
      } else
      hipLaunchKernelGGL(k_step_points<false>, dim3(point_blocks + CeilDiv(h->C, 256)), dim3(256), 0, s, sa, point_blocks, h->poses, h->points, h->poses_c, h->points_c);
    } else {
      hipLaunchKernelGGL(k_backsub_points, dim3(CeilDiv(4 * (int64_t)h->P, 256)), dim3(256), 0, s, sa);
      hipLaunchKernelGGL(k_model_cost_apply, dim3(grid_obs + CeilDiv(std::max(h->C, h->P), 256)), dim3(256), 0, s, sa, grid_obs, h->poses, h->points,
                         h->poses_c, h->points_c);
    }
    if (!fold) hipLaunchKernelGGL(k_sum, dim3(1), dim3(256), 0, s, sa.partials, grid_obs, h->scal + kModelChange);
    t2.Mark(PP_BA_T_BACKSUB);
    if (h->NI > 0 && !fused_step)
      hipLaunchKernelGGL(k_apply_intr, dim3(CeilDiv(h->K * kCamStride, 256)), dim3(256), 0, s, h->K, h->C, h->intr_off, h->intr_col, h->intr, h->scale_c,
                         h->step_c, h->intr_c);
    if (!fused_trial_cost && (rc = LaunchCostOnly(h, h->poses_c, h->points_c, h->NI > 0 ? h->intr_c : nullptr, fold ? nullptr : h->scal + kCostCand))) return rc;
    PP_HIP_TRY(hipGetLastError());
    if (h->allreduce) {      // (host callback: the two sums exist before the norms kernel here; with RCCL the norms call reduces what it folded)
      if ((rc = GroupReduce(h, h->scal + kCostCand, 2, PP_REDUCE_SUM))) return rc;   // kCostCand, kModelChange are adjacent
    }
    const unsigned long long ticket = direct ? ++h->ticket_seq : 0;
    if ((rc = LaunchNorms(h, true, fold ? 2 : 0, direct ? h->h_scal_dev : nullptr, ticket))) return rc;
    t2.Mark(PP_BA_T_UPDATE_COST);
    bool speculated = false;
    double* h_eval_prev = h_eval;                 // where a pending evaluation (the previous accepted step's) arrives
    if (speculate) {
      if (!direct) {
        PP_HIP_TRY(hipMemcpyAsync(h->h_scal, h->scal, sizeof(double) * kNumScalars, hipMemcpyDeviceToHost, s));
        PP_HIP_TRY(hipEventRecord(h->ev_readback, s));
      }
      swap_points();
      if ((rc = enqueue_evaluation())) return rc;
      speculated = true;
      if (direct) { if ((rc = WaitTicket(h, ticket))) return rc; }
      else PP_HIP_TRY(hipEventSynchronize(h->ev_readback));
    } else {
      if ((rc = ReadScalars(h))) return rc;
    }
    // leaves the speculated state: the old point is current again; `reevaluate` restores its Jacobians and sums
    auto undo_speculation = [&](bool reevaluate) -> int {
      if (!speculated) return PP_OK;
      speculated = false;
      swap_points();
      if (!reevaluate) return PP_OK;
      const int r = enqueue_evaluation();       // result identical to what `cost` / `gmax` already hold: never consumed
      return r;
    };
    t2.Collect();
    if (pending) {
      h_eval = h_eval_prev;
      resolve();
      if (gmax <= o->gradient_tolerance) {   // drops the speculative trial step
        if ((rc = undo_speculation(false))) return rc;
        sum->termination = PP_TERM_CONVERGENCE; break;
      }
    }

    const double model_change = h->h_scal[kModelChange], ccost = h->h_scal[kCostCand];
    const double step_norm = std::sqrt(h->h_scal[kStepNorm2]), x_norm = std::sqrt(h->h_scal[kXNorm2]);
    bool valid = HostFlag(h) == 0 && std::isfinite(model_change) && model_change > 0.0 && std::isfinite(step_norm);
    if (HostFlag(h) != 0) PP_HIP_TRY(hipMemsetAsync(h->d_flag, 0, sizeof(int32_t), s));
    if ((HostFlag(h) & 4) && h->chol_aux.mode != 0) {
      // a bounded wait of the one-launch factorisation (k_cholesky_tasks) ran out: nothing wrong with the system - the same step again
      // with one launch per block column, which this handle then stays with
      h->chol_aux.mode = 0;
      ++h->chol_aux.fallbacks;
      if (h->chol_aux.graph_exec) { (void)hipGraphExecDestroy(h->chol_aux.graph_exec); h->chol_aux.graph_exec = nullptr; }
      PP_HIP_TRY(hipMemsetAsync(h->S, 0, sizeof(double) * (size_t)h->N * h->N, s));      // (the assembly relies on the zero padding it never rewrites; the aborted run may have touched it)
      if ((rc = undo_speculation(true))) return rc;
      --iter;
      continue;
    }
    if (!valid) {
      ++invalid;
      if (invalid >= o->max_num_consecutive_invalid_steps) {
        if ((rc = undo_speculation(false))) return rc;
        sum->termination = PP_TERM_FAILURE;
        SetLastError("pp_ba_solve: %d consecutive invalid steps (linear system not positive definite or step without model decrease)", invalid);
        break;
      }
      if ((rc = undo_speculation(true))) return rc;
      radius /= decrease_factor; decrease_factor *= 2.0;
      push(cost, 0, gmax, 0, 0, radius, 0);
      ++sum->num_unsuccessful_steps; last_successful = false;
      user_stop = user_callback();
      continue;
    }
    invalid = 0;
    if (step_norm <= o->parameter_tolerance * (x_norm + o->parameter_tolerance)) {
      if ((rc = undo_speculation(false))) return rc;
      sum->termination = PP_TERM_CONVERGENCE; break;
    }
    const double cost_change = cost - ccost;
    if (std::fabs(cost_change) <= o->function_tolerance * cost) {
      if ((rc = undo_speculation(false))) return rc;
      sum->termination = PP_TERM_CONVERGENCE; break;
    }
    const double rel = cost_change / model_change;
    if (rel > o->min_relative_decrease) {
      if (!speculated) {     // the candidate becomes the current point; its evaluation is consumed with the next read-back
        swap_points();
        if ((rc = enqueue_evaluation())) return rc;
      }
      h_eval = h->h_scal + kNumScalars * (1 + eval_slot);
      pending = true;
      cost = ccost;     // provisional (the candidate evaluation); replaced by the re-evaluated cost when it arrives
      radius = radius / std::fmax(1.0 / 3.0, 1.0 - std::pow(2.0 * rel - 1.0, 3));
      radius = std::fmin(o->max_trust_region_radius, radius);
      decrease_factor = 2.0; reuse_diagonal = false;
      ++sum->num_successful_steps; last_successful = true;
      push(cost, cost_change, gmax, step_norm, rel, radius, 1);
      if (o->iteration_callback) {     // the callback sees the cost / gradient norm AT the accepted point: wait for its evaluation
        PP_HIP_TRY(hipStreamSynchronize(s));
        resolve();
      }
    } else {
      if ((rc = undo_speculation(true))) return rc;
      radius /= decrease_factor; decrease_factor *= 2.0; reuse_diagonal = true;
      ++sum->num_unsuccessful_steps; last_successful = false;
      push(cost, cost_change, gmax, step_norm, rel, radius, 0);
    }
    user_stop = user_callback();
  }
  PP_HIP_TRY(hipEventRecord(h->ev1, s));
  PP_HIP_TRY(hipEventSynchronize(h->ev1));
  if (pending) resolve();
  float ms = 0;
  PP_HIP_TRY(hipEventElapsedTime(&ms, h->ev0, h->ev1));
  sum->final_cost = cost;
  sum->num_iterations = sum->num_successful_steps + sum->num_unsuccessful_steps;
  sum->device_time_s = ms * 1e-3;
  sum->total_time_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
  const int neff = h->num_effective_pose_point + h->NI;     // (three synchronous read-backs of the masks per solve before: ~50 us)
  sum->num_effective_parameters = neff;
  sum->linear_solver_iterations = h->linear_solver_iterations;
  sum->linear_solver = h->iterative ? PP_LINSOLVE_PCG : h->chol_aux.last_used < 0 ? (SparseActive(h) ? PP_LINSOLVE_CHOLESKY_SPARSE : (h->Lfac ? PP_LINSOLVE_CHOLESKY_TASKS : PP_LINSOLVE_CHOLESKY_COLUMNS)) : h->chol_aux.last_used;
  sum->cholesky_fallbacks = h->chol_aux.fallbacks;
  return sum->termination == PP_TERM_FAILURE ? PP_ERR_NUMERIC : PP_OK;
} PP_API_CATCH("pp_ba_solve")

}  // extern "C"

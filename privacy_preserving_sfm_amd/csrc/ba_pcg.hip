// K3-it — matrix-free preconditioned conjugate gradients on the point-Schur complement of the camera system.
//
// Replaces the linear solve Ceres performs for bundle adjustments of more than 1000 images, where the reference switches to
// ITERATIVE_SCHUR with the SCHUR_JACOBI preconditioner (src/optim/bundle_adjustment.cc:283-286; iteration cap
// max_linear_solver_iterations = 200, bundle_adjustment.h:87).  Ceres is third-party and absent: the loop restates its published
// ConjugateGradientsSolver (*** parity unpinned ***, like the LM driver): x0 = 0, explicit residual every 10th iteration,
// termination on the quadratic-model criterion  i (Q_i - Q_{i-1}) / Q_i < eta  (the residual criterion is switched off by the
// trust-region strategy), preconditioner = inverses of the diagonal blocks of S per PARAMETER BLOCK (rotation tangent 3x3, translation
// 3x3).  oracle/bundle_adjustment.h SchurJacobiConjugateGradients is the CPU restatement the tests compare with.
//
// The reduced system S = U + D_c^2 - W (V + D_p^2)^-1 W^T is never formed (at 5000 images it would be 7 GB and a second of
// factorisation): S v is applied from the per-observation records k_prepare builds anyway ([T_o | J_c,o s_c | J_p,o], 192 bytes):
//   k_pcg_points   per point (four lanes):   a_p = sum_{o in p} J_p,o^T (J^_c,o v_c(o))
//   k_pcg_images   per image (one workgroup): (S v)_c = sum_{o in c} J^_c,o^T (J^_c,o v_c - T_o a_p(o)) + d_c v_c, and v_c . (S v)_c
// both HBM/L2-bound gathers of 144 of a record's 192 bytes per observation (algorithmic: 2 x 144 B per observation and product).
// The vector updates, dot products (fixed order: deterministic) and the termination test of an iteration: k_pcg_wide_a / _b (many
// workgroups, two launches), or k_pcg_vec (ONE workgroup, one launch: the first form, kept behind PPSFM_PCG_WIDE=0 for the tests).  The host enqueues a few iterations at a time and reads the
// state back; once the loop has ended the kernels already in the stream return at their first instruction.
// DEFAULT outside point-sharded groups: three launches per iteration (k_pcg_points_dir, k_pcg_images_dir, k_pcg_step - see below): the
// product kernels take the decision and form the direction themselves.  PPSFM_PCG_FUSED=0 / PPSFM_PCG_WIDE=0 select the older forms.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <thread>

#include "ba_impl.hpp"
#include "resource_pool.hpp"

namespace ppsfm {

constexpr int kVecThreads = 1024;
constexpr int kResidualResetPeriod = 10;      // ConjugateGradientsSolver::Options::residual_reset_period
enum { kPcgRunning = 0, kPcgConverged = 1, kPcgNoConvergence = 2, kPcgFailure = 3 };

// sum over the workgroup, the same order every time: wave butterflies, then the sixteen wave totals in wave order
__device__ __forceinline__ double BlockSum(double v, double* red) {
  v = WaveSum(v);
  __syncthreads();      // (red may still be read from the previous call)
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  double s = 0.0;
#pragma unroll
  for (int w = 0; w < kVecThreads / 64; ++w) s += red[w];
  return s;
}

// inverse of the two 3x3 diagonal blocks (rotation tangent, translation) of every image's 6x6 diagonal block of S
__global__ __launch_bounds__(256) void k_pcg_block_inverse(int C, const double* __restrict__ Sd, double* __restrict__ binv, int32_t* __restrict__ flag) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= 2 * C) return;
  const int c = i >> 1, h = i & 1;
  const double* B = Sd + 36 * (size_t)c + 21 * h;      // rows 3h.., columns 3h..
  const double a = B[0], b = B[1], cc = B[2], d = B[7], e = B[8], f = B[14];
  const double c00 = d * f - e * e, c01 = cc * e - b * f, c02 = b * e - cc * d;
  const double det = a * c00 + b * c01 + cc * c02;
  if (!(det > 0.0) || !isfinite(det)) atomicOr(flag, 1);
  const double id = 1.0 / det;
  double* o = binv + 18 * (size_t)c + 9 * h;
  o[0] = c00 * id; o[1] = c01 * id; o[2] = c02 * id;
  o[3] = c01 * id; o[4] = (a * f - cc * cc) * id; o[5] = (b * cc - a * e) * id;
  o[6] = c02 * id; o[7] = (b * cc - a * e) * id; o[8] = (a * d - b * b) * id;
}

__global__ __launch_bounds__(256) void k_pcg_points(int P, const int32_t* __restrict__ pt_start, const int32_t* __restrict__ pt_obs,
                                                    const int32_t* __restrict__ obs_pose, const double* __restrict__ rec, const double* __restrict__ v,
                                                    double* __restrict__ a, const PcgState* __restrict__ st, const double2* __restrict__ tk) {
  if (st->done) return;
  const int gid = blockIdx.x * 256 + threadIdx.x;
  const int p = gid >> 2, q = gid & 3;
  double acc[3] = {0.0, 0.0, 0.0};
  if (p < P) {
    for (int e = pt_start[p] + q; e < pt_start[p + 1]; e += 4) {
      const int o = pt_obs[e];
      const int c = obs_pose[o];
      const double2* rj = reinterpret_cast<const double2*>(RecJ(rec, (size_t)o));      // J_pose,o s_c (2 x 6) then J_pt,o (2 x 3): 144 contiguous bytes
      double jp[12], jx[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) { const double2 t = rj[i]; jp[2 * i] = t.x; jp[2 * i + 1] = t.y; }
#pragma unroll
      for (int i = 0; i < 3; ++i) { const double2 t = rj[6 + i]; jx[2 * i] = t.x; jx[2 * i + 1] = t.y; }
      double m0 = 0.0, m1 = 0.0;
      if (tk) { const double2 t = tk[o]; m0 = t.x; m1 = t.y; }      // (variable intrinsics: J^_k,o v_k)
#pragma unroll
      for (int j = 0; j < 6; ++j) { const double d = v[6 * (size_t)c + j]; m0 += jp[j] * d; m1 += jp[6 + j] * d; }
#pragma unroll
      for (int k = 0; k < 3; ++k) acc[k] += jx[k] * m0 + jx[3 + k] * m1;
    }
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {      // the four lanes of a point: fixed two-step butterfly
    acc[k] += __shfl_xor(acc[k], 1);
    acc[k] += __shfl_xor(acc[k], 2);
  }
  if (p < P && q == 0) { a[3 * (size_t)p] = acc[0]; a[3 * (size_t)p + 1] = acc[1]; a[3 * (size_t)p + 2] = acc[2]; }
}

__global__ __launch_bounds__(256) void k_pcg_images(int C, const int32_t* __restrict__ pose_start, const int32_t* __restrict__ pose_obs,
                                                    const int32_t* __restrict__ obs_point, const double* __restrict__ rec, const double* __restrict__ v,
                                                    const double* __restrict__ a, const double* __restrict__ scale_c, const double* __restrict__ diag_c,
                                                    double inv_radius, double* __restrict__ out, double* __restrict__ dotp, const PcgState* __restrict__ st,
                                                    int add_diagonal, const double2* __restrict__ tk, double2* __restrict__ w_out) {
  if (st->done) return;
  __shared__ double red[4][6];
  const int c = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  double vc[6];
#pragma unroll
  for (int j = 0; j < 6; ++j) vc[j] = v[6 * (size_t)c + j];
  double acc[6] = {0, 0, 0, 0, 0, 0};
  for (int e = pose_start[c] + (int)threadIdx.x; e < pose_start[c + 1]; e += 256) {
    const int o = pose_obs[e];
    const int p = obs_point[o];
    const double2* rt = reinterpret_cast<const double2*>(RecT(rec, (size_t)o));      // T_o (2 x 3) then J_pose,o s_c (2 x 6): 144 contiguous bytes
    double t[6], jp[12];
#pragma unroll
    for (int i = 0; i < 3; ++i) { const double2 x = rt[i]; t[2 * i] = x.x; t[2 * i + 1] = x.y; }
#pragma unroll
    for (int i = 0; i < 6; ++i) { const double2 x = rt[3 + i]; jp[2 * i] = x.x; jp[2 * i + 1] = x.y; }
    const double a0 = a[3 * (size_t)p], a1 = a[3 * (size_t)p + 1], a2 = a[3 * (size_t)p + 2];
    double m0 = -(t[0] * a0 + t[1] * a1 + t[2] * a2), m1 = -(t[3] * a0 + t[4] * a1 + t[5] * a2);
    if (tk) { const double2 tt = tk[o]; m0 += tt.x; m1 += tt.y; }      // (variable intrinsics: J^_k,o v_k)
#pragma unroll
    for (int j = 0; j < 6; ++j) { m0 += jp[j] * vc[j]; m1 += jp[6 + j] * vc[j]; }
    if (w_out) w_out[o] = make_double2(m0, m1);      // (what the per-camera sums J^_k^T m take)
#pragma unroll
    for (int j = 0; j < 6; ++j) acc[j] += jp[j] * m0 + jp[6 + j] * m1;
  }
#pragma unroll
  for (int j = 0; j < 6; ++j) acc[j] = WaveSum(acc[j]);
  if (lane == 0) {
#pragma unroll
    for (int j = 0; j < 6; ++j) red[wv][j] = acc[j];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double dot = 0.0;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const double s = scale_c[6 * (size_t)c + j];
      // constant column: identity row (as the assembled system has it); in a point-sharded group the diagonal term is added once, by rank 0
      const double d = add_diagonal ? ((s == 0.0) ? 1.0 : diag_c[6 * (size_t)c + j] * inv_radius) : 0.0;
      const double qv = ((red[0][j] + red[1][j]) + red[2][j]) + red[3][j] + d * vc[j];
      out[6 * (size_t)c + j] = qv;
      dot += vc[j] * qv;
    }
    if (dotp) dotp[c] = dot;      // (a group sums the products of all ranks first: v . S v is taken from the reduced vector then)
  }
}

// z = M^-1 r for this thread's elements i = tid, tid + 1024, ..;  returns this thread's part of r . z
__device__ __forceinline__ double Precondition(int n, const double* __restrict__ binv, const double* __restrict__ r, double* __restrict__ z) {
  double part = 0.0;
  for (int i = threadIdx.x; i < n; i += kVecThreads) {
    const int c = i / 6, j = i - 6 * c, h = j / 3, jj = j - 3 * h;
    const double* B = binv + 18 * (size_t)c + 9 * h + 3 * jj;
    const double* rr = r + 6 * (size_t)c + 3 * h;
    const double zv = B[0] * rr[0] + B[1] * rr[1] + B[2] * rr[2];
    z[i] = zv;
    part += r[i] * zv;
  }
  return part;
}

// One launch per step of the loop, ONE workgroup.  mode 0: start (x = 0, r = b, first direction).  mode 1: after q = S p of iteration
// `it`: step length, x, r (or, every 10th iteration, x only: the host then applies S to x and calls mode 2), termination test, next
// direction.  mode 2: r = b - S x (in `q`), termination test, next direction.
__global__ __launch_bounds__(kVecThreads) void k_pcg_vec(int mode, int it, int n, int C, const double* __restrict__ b, double* __restrict__ x, double* __restrict__ r,
                                                         double* __restrict__ z, double* __restrict__ p, const double* __restrict__ q,
                                                         const double* __restrict__ binv, const double* __restrict__ dotp, PcgState* __restrict__ st,
                                                         double eta, int max_iterations, int32_t* __restrict__ flag) {
  __shared__ double red[kVecThreads / 64];
  __shared__ PcgState s_in;
  const int tid = threadIdx.x;
  if (tid == 0) s_in = *st;
  __syncthreads();
  // every thread works on its OWN copy of the state (all of them compute the same scalars from the same block sums, so all take the
  // same branches); thread 0 writes the new state back at the end.  Nothing is read from the shared copy after this line: a thread
  // that is ahead must not change what a slower wavefront is still about to read.
  PcgState s = s_in;
  if (mode != 0 && (s.done || s.iter != it)) return;
  auto finish = [&](int status) {      // all threads call with the same value
    if (tid == 0) { s.done = 1; s.status = status; *st = s; if (status == kPcgFailure) atomicOr(flag, 1); }
  };
  if (mode == 0) {
    double nb = 0.0;
    for (int i = tid; i < n; i += kVecThreads) { const double bv = b[i]; x[i] = 0.0; r[i] = bv; nb += bv * bv; }
    nb = BlockSum(nb, red);
    s.iter = 1; s.done = 0; s.status = kPcgRunning; s.rho = 1.0; s.Q0 = 0.0; s.norm_b = sqrt(nb); s.alpha = 0.0;
    if (!(nb > 0.0)) { s.iter = 0; finish(nb == 0.0 ? kPcgConverged : kPcgFailure); return; }      // |b| = 0: x = 0 is the solution (NaN: failure)
  } else if (mode == 1) {
    double pq = 0.0;
    if (dotp) { for (int c = tid; c < C; c += kVecThreads) pq += dotp[c]; }
    else { for (int i = tid; i < n; i += kVecThreads) pq += p[i] * q[i]; }      // a group: q is the all-reduced product
    pq = BlockSum(pq, red);
    if (!(pq > 0.0) || isinf(pq)) { finish(isnan(pq) ? kPcgFailure : kPcgNoConvergence); return; }      // indefinite direction: the iterate so far is the answer
    const double alpha = s.rho / pq;
    if (isinf(alpha)) { finish(kPcgFailure); return; }
    const bool reset = (it % kResidualResetPeriod) == 0;
    for (int i = tid; i < n; i += kVecThreads) { x[i] += alpha * p[i]; if (!reset) r[i] -= alpha * q[i]; }
    if (reset) { if (tid == 0) { s.alpha = alpha; *st = s; } return; }
  } else {
    for (int i = tid; i < n; i += kVecThreads) r[i] = b[i] - q[i];
  }
  if (mode != 0) {
    double q1 = 0.0;
    for (int i = tid; i < n; i += kVecThreads) q1 -= x[i] * (b[i] + r[i]);
    q1 = BlockSum(q1, red);
    const double zeta = it * (q1 - s.Q0) / q1;
    if (zeta < eta) { finish(kPcgConverged); return; }
    if (it >= max_iterations) { finish(kPcgNoConvergence); return; }
    s.Q0 = q1; s.iter = it + 1;
  }
  // next direction (the block sum's barriers also order this workgroup's writes of r before the reads below)
  const double rho = BlockSum(Precondition(n, binv, r, z), red);
  if (rho == 0.0 || isinf(rho) || isnan(rho)) { finish(kPcgFailure); return; }
  double beta = 0.0;
  if (mode != 0) {
    beta = rho / s.rho;
    if (beta == 0.0 || isinf(beta) || isnan(beta)) { finish(kPcgFailure); return; }
  }
  for (int i = tid; i < n; i += kVecThreads) p[i] = (mode == 0) ? z[i] : z[i] + beta * p[i];
  s.rho = rho;
  if (tid == 0) *st = s;
}

// ---- the vector step over MANY workgroups (k_pcg_vec, one workgroup, is 12.5 us at 1100 images and 30 us at 4000 - the longest kernel
// of an iteration there).  Two launches instead of one, a thread per 3 x 3 parameter block:
//   k_pcg_wide_a   alpha = rho / (p . S p) from the per-image parts (every workgroup sums them itself, in the same order: the same
//                  bits everywhere, no exchange), x += alpha p, r -= alpha q (or the explicit residual), z = M^-1 r, and this
//                  workgroup's parts of Q = -x . (b + r) / ... and of r . z
//   k_pcg_wide_b   every workgroup sums the parts, takes the termination decision of k_pcg_vec (same tests, same order) and updates its
//                  slice of the direction p = z + beta p
// The state is read by every workgroup of a launch and written by workgroup 0 of k_pcg_wide_b: it ping-pongs between two copies (a
// workgroup dispatched late must not find the NEXT iteration's state), the host tracks which one is current.
constexpr int kWideThreads = 256;
// Variable intrinsics on the iterative path (bundle_adjustment.cc:283-286 with :490-528): the NI intrinsics columns follow the 6 C pose columns in
// every vector of the loop.  Their part of the operator, matrix-free like the rest:
//   k_pcg_cam_t    t_o = J^_k,o v_k per observation (the scaled compact intrinsics Jacobians k_intr_prepare builds)
//   k_pcg_points / k_pcg_images add t_o to J^_c,o v_c; the image kernel also stores m_o = J^_c,o v_c + t_o - T_o a_p
//   k_pcg_cam_q    (S v)_k = sum_{o of camera k} J^_k,o^T m_o + d_k v_k: per-camera chunks of the observations (the lists of k_intr_sums), then one
//                  wavefront per camera adds the chunks in order and forms the camera's part of v . S v
// The preconditioner's intrinsics blocks (SCHUR_JACOBI: one block per parameter block) are the diagonal blocks S_kk, assembled per trial radius
// from the (k, k) pair lists alone (IntrAssemble into pcg_Scomp) and inverted by k_pcg_intr_inverse; the vector step treats a camera's block in
// one thread (workgroups behind the pose workgroups of k_pcg_wide_a / _b).  The four-launch form of the loop runs these (7 launches per iteration).
struct PcgIntr {
  int K = 0, pose_blocks = 1 << 30;      // cameras; the workgroups before the intrinsics' ones
  const int32_t* intr_off = nullptr; const int32_t* intr_nv = nullptr;
  const double* binvI = nullptr;         // [NI][12]: row i of the intrinsics columns holds its block's row of the inverse
};
__global__ __launch_bounds__(256) void k_pcg_cam_t(int64_t M, int C, const int32_t* __restrict__ obs_cam, const int32_t* __restrict__ intr_off,
                                                   const int32_t* __restrict__ intr_nv, const double* __restrict__ JkS, const double* __restrict__ v,
                                                   double2* __restrict__ tk, const PcgState* __restrict__ st) {
  if (st->done) return;
  const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (o >= M) return;
  const int k = obs_cam[o] >> 4;
  const int off = intr_off[k];
  double t0 = 0.0, t1 = 0.0;
  if (off >= 0) {
    const int nv = intr_nv[k];
    const double* j = JkS + (size_t)2 * kCamStride * o;
    const double* vk = v + 6 * (size_t)C + off;
    for (int c = 0; c < nv; ++c) { const double d = vk[c]; t0 += j[c] * d; t1 += j[kCamStride + c] * d; }
  }
  tk[o] = make_double2(t0, t1);
}
__global__ __launch_bounds__(256) void k_pcg_cam_q(const int32_t* __restrict__ chunk, const int32_t* __restrict__ cam_obs, const double* __restrict__ JkS,
                                                   const double2* __restrict__ w, double* __restrict__ partial, const PcgState* __restrict__ st) {
  if (st->done) return;
  __shared__ double red[4][12];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int e0 = chunk[3 * blockIdx.x + 1], e1 = chunk[3 * blockIdx.x + 2];
  double acc[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) acc[i] = 0.0;
  for (int e = e0 + (int)threadIdx.x; e < e1; e += 256) {
    const int o = cam_obs[e];
    const double* j = JkS + (size_t)2 * kCamStride * o;
    const double2 m = w[o];
#pragma unroll
    for (int c = 0; c < 12; ++c) acc[c] += j[c] * m.x + j[kCamStride + c] * m.y;
  }
#pragma unroll
  for (int i = 0; i < 12; ++i) acc[i] = WaveSum(acc[i]);
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < 12; ++i) red[wv][i] = acc[i];
  }
  __syncthreads();
  if (threadIdx.x < 12) partial[(size_t)blockIdx.x * 24 + threadIdx.x] = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
}
// one wavefront per camera: the chunk sums in chunk order, the damping term, the camera's part of v . S v
__global__ __launch_bounds__(64) void k_pcg_cam_q_reduce(int C, const int32_t* __restrict__ cam_chunk, const int32_t* __restrict__ intr_off, const int32_t* __restrict__ intr_nv,
                                                         const double* __restrict__ partial, const double* __restrict__ v, const double* __restrict__ diag_c, double inv_radius,
                                                         double* __restrict__ out, double* __restrict__ dotp, const PcgState* __restrict__ st, int add_diag) {      // add_diag: this rank carries the damping term (rank 0 of a group)
  if (st->done) return;
  const int k = blockIdx.x, t = threadIdx.x;
  const int off = intr_off[k];
  double term = 0.0;
  if (off >= 0 && t < intr_nv[k]) {
    double s = 0.0;
    int c = cam_chunk[k];
    const int ce = cam_chunk[k + 1];
    for (; c + 8 <= ce; c += 8) {      // eight partials in flight, added in chunk order (one shared camera at 1100 images: 86 chunks, 15.8 us one load at a time)
      double v8[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v8[u] = partial[(size_t)(c + u) * 24 + t];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v8[u];
    }
    for (; c < ce; ++c) s += partial[(size_t)c * 24 + t];
    const size_t idx = 6 * (size_t)C + off + t;
    const double qv = add_diag ? s + diag_c[idx] * inv_radius * v[idx] : s;
    out[idx] = qv;
    term = v[idx] * qv;
  }
  double dot = 0.0;
#pragma unroll
  for (int j = 0; j < 12; ++j) dot += __shfl(term, j);
  if (t == 0) dotp[C + k] = dot;
}
// inverses of the intrinsics' diagonal blocks (Gauss-Jordan on the symmetric positive definite block, as the oracle does): a thread per camera
__global__ __launch_bounds__(64) void k_pcg_intr_inverse(int K, const int32_t* __restrict__ intr_off, const int32_t* __restrict__ intr_nv, const double* __restrict__ Scomp,
                                                         double* __restrict__ binvI, int32_t* __restrict__ flag) {
  const int k = blockIdx.x * 64 + threadIdx.x;
  if (k >= K) return;
  const int off = intr_off[k];
  if (off < 0) return;
  const int m = intr_nv[k];
  double A[12][12], I[12][12];
  for (int i = 0; i < m; ++i)
    for (int j = 0; j < m; ++j) { A[i][j] = Scomp[12 * (size_t)(off + i) + j]; I[i][j] = i == j ? 1.0 : 0.0; }
  bool bad = false;
  for (int c = 0; c < m; ++c) {
    const double piv = A[c][c];
    if (!(piv > 0.0) || !isfinite(piv)) { bad = true; break; }
    for (int j = 0; j < m; ++j) { A[c][j] /= piv; I[c][j] /= piv; }
    for (int r = 0; r < m; ++r) {
      if (r == c) continue;
      const double f = A[r][c];
      for (int j = 0; j < m; ++j) { A[r][j] -= f * A[c][j]; I[r][j] -= f * I[c][j]; }
    }
  }
  if (bad) atomicOr(flag, 1);
  for (int i = 0; i < m; ++i)
    for (int j = 0; j < 12; ++j) binvI[12 * (size_t)(off + i) + j] = (j < m && !bad) ? I[i][j] : 0.0;
}
// per-image parts of p . q for a point-sharded group, where q is the all-reduced product (k_pcg_images' own parts would be this rank's only)
__global__ __launch_bounds__(256) void k_pcg_dot(int C, const double* __restrict__ p, const double* __restrict__ q, double* __restrict__ dotp,
                                                 const PcgState* __restrict__ st, int K, const int32_t* __restrict__ intr_off, const int32_t* __restrict__ intr_nv) {      // K > 0: the cameras' parts behind the images'
  if (st->done) return;
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C + K) return;
  double d = 0.0;
  if (c < C) {
#pragma unroll
    for (int j = 0; j < 6; ++j) d += p[6 * (size_t)c + j] * q[6 * (size_t)c + j];
  } else {
    const int k = c - C, off = intr_off[k];
    if (off >= 0) for (int j = 0; j < intr_nv[k]; ++j) d += p[6 * (size_t)C + off + j] * q[6 * (size_t)C + off + j];
  }
  dotp[c] = d;
}
__device__ __forceinline__ double WideBlockSum(double v, double* red) {      // fixed order, result in every thread
  v = WaveSum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return ((red[0] + red[1]) + red[2]) + red[3];
}
// mode 0: start (x = 0, r = b); 1: after q = S p of iteration `it`; 2: after q = S x (explicit residual).  part: 4 doubles per workgroup
// [Q part (mode 0: |b|^2 part) | r . z part | status | alpha]
__global__ __launch_bounds__(kWideThreads) void k_pcg_wide_a(int mode, int it, int C, const double* __restrict__ b, double* __restrict__ x, double* __restrict__ r,
                                                             double* __restrict__ z, const double* __restrict__ p, const double* __restrict__ q,
                                                             const double* __restrict__ binv, const double* __restrict__ dotp, const PcgState* __restrict__ st,
                                                             double* __restrict__ part, PcgIntr in) {
  __shared__ double red[4];
  const PcgState s = *st;
  if (mode != 0 && (s.done || s.iter != it)) return;
  const int tid = threadIdx.x;
  double alpha = 0.0;
  int status = kPcgRunning;
  if (mode == 1) {
    double pq = 0.0;
    for (int c = tid; c < C + in.K; c += kWideThreads) pq += dotp[c];      // (per image, then per camera with variable intrinsics)
    pq = WideBlockSum(pq, red);
    if (!(pq > 0.0) || isinf(pq)) status = isnan(pq) ? kPcgFailure : kPcgNoConvergence;      // indefinite direction: the iterate so far is the answer
    else { alpha = s.rho / pq; if (isinf(alpha)) status = kPcgFailure; }
  }
  const bool reset = mode == 1 && (it % kResidualResetPeriod) == 0;
  const int g = blockIdx.x * kWideThreads + tid;      // parameter block: rows 3 g .. 3 g + 2
  double s0 = 0.0, s1 = 0.0;
  if ((int)blockIdx.x >= in.pose_blocks) {
    // ---- the intrinsics blocks (variable intrinsics): a thread per camera, the rows of its block one after the other
    const int k = ((int)blockIdx.x - in.pose_blocks) * kWideThreads + tid;
    const int off = k < in.K ? in.intr_off[k] : -1;
    if (off >= 0 && status == kPcgRunning) {
      const int nv = in.intr_nv[k];
      const size_t i0 = 6 * (size_t)C + off;
      double rn[12], xn[12];
      for (int j = 0; j < nv; ++j) {
        const double bj = b[i0 + j];
        if (mode == 0) { xn[j] = 0.0; rn[j] = bj; x[i0 + j] = 0.0; s0 += bj * bj; }
        else if (mode == 1) { xn[j] = x[i0 + j] + alpha * p[i0 + j]; x[i0 + j] = xn[j]; rn[j] = reset ? 0.0 : r[i0 + j] - alpha * q[i0 + j]; }
        else { xn[j] = x[i0 + j]; rn[j] = bj - q[i0 + j]; }
        if (!reset && mode != 0) s0 -= xn[j] * (bj + rn[j]);
      }
      if (!reset) {
        for (int a = 0; a < nv; ++a) {
          const double* B = in.binvI + 12 * (size_t)(off + a);
          double zv = 0.0;
          for (int j = 0; j < nv; ++j) zv += B[j] * rn[j];
          z[i0 + a] = zv; r[i0 + a] = rn[a];
          s1 += rn[a] * zv;
        }
      }
    }
  } else if (g < 2 * C && status == kPcgRunning) {
    const size_t i = 3 * (size_t)g;
    double xv[3], rv[3], bv[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) bv[k] = b[i + k];
    if (mode == 0) {
#pragma unroll
      for (int k = 0; k < 3; ++k) { xv[k] = 0.0; rv[k] = bv[k]; x[i + k] = 0.0; s0 += bv[k] * bv[k]; }
    } else if (mode == 1) {
#pragma unroll
      for (int k = 0; k < 3; ++k) { xv[k] = x[i + k] + alpha * p[i + k]; x[i + k] = xv[k]; }
      if (!reset) {
#pragma unroll
        for (int k = 0; k < 3; ++k) rv[k] = r[i + k] - alpha * q[i + k];
      }
    } else {
#pragma unroll
      for (int k = 0; k < 3; ++k) { xv[k] = x[i + k]; rv[k] = bv[k] - q[i + k]; }
    }
    if (!reset) {
      if (mode != 0) {
#pragma unroll
        for (int k = 0; k < 3; ++k) s0 -= xv[k] * (bv[k] + rv[k]);
      }
      const double* B = binv + 9 * (size_t)g;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const double zv = B[3 * k] * rv[0] + B[3 * k + 1] * rv[1] + B[3 * k + 2] * rv[2];
        z[i + k] = zv; r[i + k] = rv[k];
        s1 += rv[k] * zv;
      }
    }
  }
  s0 = WideBlockSum(s0, red);
  s1 = WideBlockSum(s1, red);
  if (tid == 0) { double* o = part + 4 * (size_t)blockIdx.x; o[0] = s0; o[1] = s1; o[2] = (double)status; o[3] = alpha; }
}
// mode 0: start; 1: an iteration's end; 3: after the x-only step of a residual-reset iteration (only carries a failure of that step over)
__global__ __launch_bounds__(kWideThreads) void k_pcg_wide_b(int mode, int it, int C, int G, const double* __restrict__ z, double* __restrict__ p,
                                                             const double* __restrict__ part, const PcgState* __restrict__ st, PcgState* __restrict__ st_out,
                                                             double eta, int max_iterations, int32_t* __restrict__ flag, PcgIntr in) {
  __shared__ double red[4];
  PcgState s = *st;
  const int tid = threadIdx.x;
  const bool writer = blockIdx.x == 0 && tid == 0;
  if (mode != 0 && (s.done || s.iter != it)) { if (writer) *st_out = s; return; }      // (both copies say "done" from here on)
  auto finish = [&](int status) {
    if (writer) { s.done = 1; s.status = status; *st_out = s; if (status == kPcgFailure) atomicOr(flag, 1); }
  };
  double a0 = 0.0, a1 = 0.0;
  for (int w = tid; w < G; w += kWideThreads) { a0 += part[4 * (size_t)w]; a1 += part[4 * (size_t)w + 1]; }
  a0 = WideBlockSum(a0, red);
  a1 = WideBlockSum(a1, red);
  const int status = (int)part[2];
  s.alpha = part[3];
  if (status != kPcgRunning) { finish(status); return; }
  if (mode == 3) { if (writer) *st_out = s; return; }
  double beta = 0.0;
  if (mode == 0) {
    s.iter = 1; s.done = 0; s.status = kPcgRunning; s.rho = 1.0; s.Q0 = 0.0; s.norm_b = sqrt(a0); s.alpha = 0.0;
    if (!(a0 > 0.0)) { s.iter = 0; finish(a0 == 0.0 ? kPcgConverged : kPcgFailure); return; }
  } else {
    const double zeta = it * (a0 - s.Q0) / a0;
    if (zeta < eta) { finish(kPcgConverged); return; }
    if (it >= max_iterations) { finish(kPcgNoConvergence); return; }
    s.Q0 = a0; s.iter = it + 1;
  }
  const double rho = a1;
  if (rho == 0.0 || isinf(rho) || isnan(rho)) { finish(kPcgFailure); return; }
  if (mode != 0) {
    beta = rho / s.rho;
    if (beta == 0.0 || isinf(beta) || isnan(beta)) { finish(kPcgFailure); return; }
  }
  const int g = blockIdx.x * kWideThreads + tid;
  if ((int)blockIdx.x >= in.pose_blocks) {      // the intrinsics blocks
    const int k = ((int)blockIdx.x - in.pose_blocks) * kWideThreads + tid;
    const int off = k < in.K ? in.intr_off[k] : -1;
    if (off >= 0) {
      const size_t i0 = 6 * (size_t)C + off;
      for (int j = 0; j < in.intr_nv[k]; ++j) p[i0 + j] = (mode == 0) ? z[i0 + j] : z[i0 + j] + beta * p[i0 + j];
    }
  } else if (g < 2 * C) {
    const size_t i = 3 * (size_t)g;
#pragma unroll
    for (int k = 0; k < 3; ++k) p[i + k] = (mode == 0) ? z[i + k] : z[i + k] + beta * p[i + k];
  }
  s.rho = rho;
  if (writer) *st_out = s;
}

// ---- THREE launches per iteration (the default outside point-sharded groups).  k_pcg_wide_b is gone: the decision it took and the
// direction it wrote are taken by the product kernels themselves.
//   k_pcg_points_dir   every wavefront sums the parts of Q and r . z k_pcg_step left (the same order everywhere: the same bits, no
//                      exchange), takes the termination decision and beta, and forms the direction of the images it touches on the
//                      fly: p_c = z_c + beta p_old,c (both vectors are L2 resident: 53 KB at 1100 images)
//   k_pcg_images_dir   the same decision, p_c once more for ITS image - stored: the direction buffer ping-pongs, like the state that workgroup 0
//                      writes - then (S p)_c and the image's part of p . S p
//   k_pcg_step         alpha, x, r, z = M^-1 r, parts (k_pcg_wide_a with every load issued before the first workgroup sum; at the
//                      start of a solve it also inverts the diagonal blocks: k_pcg_block_inverse folded in)
// and the index walks are one level shorter: (observation, image) pairs per point-list entry and (observation, point) pairs per image-list
// entry (k_pcg_entries, built once per handle), eight lanes per point when the tracks are long enough.  The kernels of an iteration are
// bound by the LATENCY of their dependent loads (344 + 1100 + 9 workgroups at 1100 images, a few microseconds each), not by bytes.
struct PcgDecision { PcgState s; double beta; int run; int failed; };

// The decision k_pcg_wide_b took after iteration it_prev (0: the start of a solve), by one wavefront for itself.
__device__ __forceinline__ PcgDecision PcgDecide(int it_prev, int G, const double* __restrict__ part, const PcgState* __restrict__ st, double eta,
                                                 int max_iterations) {
  PcgDecision d;
  d.s = *st; d.beta = 0.0; d.run = 0; d.failed = 0;
  const int lane = threadIdx.x & 63;
  double a0 = 0.0, a1 = 0.0;
  for (int w = lane; w < G; w += 64) { a0 += part[4 * (size_t)w]; a1 += part[4 * (size_t)w + 1]; }
  const int status = (int)part[2];
  const double alpha = part[3];
  if (it_prev != 0 && (d.s.done || d.s.iter != it_prev)) return d;      // (ended before: both copies of the state keep saying so)
  a0 = WaveSum(a0); a1 = WaveSum(a1);
  auto finish = [&](int st_code) { d.s.done = 1; d.s.status = st_code; d.failed = st_code == kPcgFailure; };
  d.s.alpha = alpha;
  if (status != kPcgRunning) { finish(status); return d; }
  if (it_prev == 0) {
    d.s.iter = 1; d.s.done = 0; d.s.status = kPcgRunning; d.s.rho = 1.0; d.s.Q0 = 0.0; d.s.norm_b = sqrt(a0); d.s.alpha = 0.0;
    if (!(a0 > 0.0)) { d.s.iter = 0; finish(a0 == 0.0 ? kPcgConverged : kPcgFailure); return d; }
  } else {
    const double zeta = it_prev * (a0 - d.s.Q0) / a0;
    if (zeta < eta) { finish(kPcgConverged); return d; }
    if (it_prev >= max_iterations) { finish(kPcgNoConvergence); return d; }
    d.s.Q0 = a0; d.s.iter = it_prev + 1;
  }
  const double rho = a1;
  if (rho == 0.0 || isinf(rho) || isnan(rho)) { finish(kPcgFailure); return d; }
  if (it_prev != 0) {
    d.beta = rho / d.s.rho;
    if (d.beta == 0.0 || isinf(d.beta) || isnan(d.beta)) { finish(kPcgFailure); return d; }
  }
  d.s.rho = rho;
  d.run = 1;
  return d;
}
// the x-only step of a residual-reset iteration left (status, alpha) in the parts: carries a failure of that step over (k_pcg_wide_b's mode 3)
__device__ __forceinline__ PcgDecision PcgCarry(int it, const double* __restrict__ part, const PcgState* __restrict__ st) {
  PcgDecision d;
  d.s = *st; d.beta = 0.0; d.run = 0; d.failed = 0;
  const int status = (int)part[2];
  const double alpha = part[3];
  if (d.s.done || d.s.iter != it) return d;
  d.s.alpha = alpha;
  if (status != kPcgRunning) { d.s.done = 1; d.s.status = status; d.failed = status == kPcgFailure; return d; }
  d.run = 1;
  return d;
}
// has the loop ended before this launch (the test PcgDecide / PcgCarry begin with)?  One load, asked for together with the list entry: a launch enqueued
// beyond the loop's end leaves after ONE round trip instead of after the record fetch it would never use
__device__ __forceinline__ bool PcgEnded(const PcgState* __restrict__ st, int it, bool dir) {
  const int2 w = *reinterpret_cast<const int2*>(&st->iter);      // (iter, done)
  return (!dir || it != 0) && (w.y != 0 || w.x != it);
}
__device__ __forceinline__ double PcgDirection(bool first, double beta, double z, double p_old) { return first ? z : fma(beta, p_old, z); }

__global__ __launch_bounds__(256) void k_pcg_entries(int64_t M, const int32_t* __restrict__ pt_obs, const int32_t* __restrict__ obs_pose,
                                                     const int32_t* __restrict__ pose_obs, const int32_t* __restrict__ obs_point,
                                                     int2* __restrict__ pt_entry, int2* __restrict__ pose_entry) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= M) return;
  const int o = pt_obs[e], o2 = pose_obs[e];
  pt_entry[e] = make_int2(o, obs_pose[o]);
  pose_entry[e] = make_int2(o2, obs_point[o2]);
}

// the decision alone, for the host's look at the loop: writes the state as it will be after the decision on iteration `it`
// (`out` is pinned host memory: the state, then - released at system scope - the ticket the host spins on: a stream synchronisation wakes the
// caller ~10 us after the kernel has ended, and every microsecond of it is an idle device in the middle of an LM iteration)
__global__ __launch_bounds__(64) void k_pcg_decide(int it, int G, const double* __restrict__ part, const PcgState* __restrict__ st, PcgState* __restrict__ out,
                                                   double eta, int max_iterations, int32_t* __restrict__ flag, int32_t ticket) {
  const PcgDecision d = PcgDecide(it, G, part, st, eta, max_iterations);
  if (threadIdx.x == 0) {
    PcgState o = d.s;
    o.pad_ = 0;
    *out = o;
    if (d.failed) atomicOr(flag, 1);
    __hip_atomic_store(&out->pad_, ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// kDir: v = the direction z + beta p_old after the decision on iteration `it` (= the one before); otherwise v = the vector given (x of a residual-reset iteration `it`)
template <bool kDir, int kLanes>
__global__ __launch_bounds__(256) void k_pcg_points_dir(int P, const int32_t* __restrict__ pt_start, const int2* __restrict__ pt_entry, const double* __restrict__ rec,
                                                        const double* __restrict__ v, const double* __restrict__ p_old, double* __restrict__ a,
                                                        const PcgState* __restrict__ st, const double* __restrict__ part, int G, int it, double eta, int max_iterations) {
  const int gid = blockIdx.x * 256 + threadIdx.x;
  const int p = gid / kLanes, q = gid % kLanes;
  int e0 = 0, e1 = 0;
  if (p < P) { e0 = pt_start[p] + q; e1 = pt_start[p + 1]; }
  const bool first = kDir && it == 0;
  // The first entry of the lane and everything it points at are requested BEFORE the decision (whose loads and wavefront sums then run
  // beside them instead of in front of them); a lane without an entry reads entry 0 and drops it.
  double jp[12], jx[6], vc[6], po[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  auto fetch = [&](int e) {
    const int2 en = pt_entry[e];
    const int o = en.x, c = en.y;
    const double2* rj = reinterpret_cast<const double2*>(RecJ(rec, (size_t)o));      // J_pose,o s_c (2 x 6) then J_pt,o (2 x 3): 144 contiguous bytes
#pragma unroll
    for (int j = 0; j < 6; ++j) vc[j] = v[6 * (size_t)c + j];
    if (kDir && !first) {
#pragma unroll
      for (int j = 0; j < 6; ++j) po[j] = p_old[6 * (size_t)c + j];
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) { const double2 t = rj[i]; jp[2 * i] = t.x; jp[2 * i + 1] = t.y; }
#pragma unroll
    for (int i = 0; i < 3; ++i) { const double2 t = rj[6 + i]; jx[2 * i] = t.x; jx[2 * i + 1] = t.y; }
  };
  fetch(e0 < e1 ? e0 : 0);
  const PcgDecision d = kDir ? PcgDecide(it, G, part, st, eta, max_iterations) : PcgCarry(it, part, st);
  if (!d.run) return;
  double acc[3] = {0.0, 0.0, 0.0};
  for (int e = e0; e < e1; e += kLanes) {
    if (e != e0) fetch(e);
    double m0 = 0.0, m1 = 0.0;
#pragma unroll
    for (int j = 0; j < 6; ++j) { const double dv = kDir ? PcgDirection(first, d.beta, vc[j], first ? 0.0 : po[j]) : vc[j]; m0 += jp[j] * dv; m1 += jp[6 + j] * dv; }
#pragma unroll
    for (int k = 0; k < 3; ++k) acc[k] += jx[k] * m0 + jx[3 + k] * m1;
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {      // the lanes of a point: fixed butterfly
#pragma unroll
    for (int off = 1; off < kLanes; off <<= 1) acc[k] += __shfl_xor(acc[k], off);
  }
  if (p < P && q == 0) { a[3 * (size_t)p] = acc[0]; a[3 * (size_t)p + 1] = acc[1]; a[3 * (size_t)p + 2] = acc[2]; }
}

template <bool kDir>
__global__ __launch_bounds__(256) void k_pcg_images_dir(int C, const int32_t* __restrict__ pose_start, const int2* __restrict__ pose_entry, const double* __restrict__ rec,
                                                        const double* __restrict__ v, const double* __restrict__ p_old, double* __restrict__ p_new,
                                                        const double* __restrict__ a, const double* __restrict__ scale_c, const double* __restrict__ diag_c, double inv_radius,
                                                        double* __restrict__ out, double* __restrict__ dotp, const PcgState* __restrict__ st, PcgState* __restrict__ st_out,
                                                        const double* __restrict__ part, int G, int it, double eta, int max_iterations, int32_t* __restrict__ flag) {
  __shared__ double red[4][6];
  const int c = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  // everything that does not depend on the decision is requested first
  const int e0 = pose_start[c] + (int)threadIdx.x, e1 = pose_start[c + 1];
  double vc[6], po[6];
#pragma unroll
  for (int j = 0; j < 6; ++j) { vc[j] = v[6 * (size_t)c + j]; po[j] = (kDir && it != 0) ? p_old[6 * (size_t)c + j] : 0.0; }
  double sc = 1.0, dg = 0.0;
  if (threadIdx.x < 6) { sc = scale_c[6 * (size_t)c + threadIdx.x]; dg = diag_c[6 * (size_t)c + threadIdx.x]; }
  // the thread's first entry and what it points at: requested before the decision, whose loads and wavefront sums then run beside them
  double t[6], jp[12], a0, a1, a2;
  auto fetch = [&](int e) {
    const int2 en = pose_entry[e];
    const int o = en.x, p = en.y;
    const double2* rt = reinterpret_cast<const double2*>(RecT(rec, (size_t)o));      // T_o (2 x 3) then J_pose,o s_c (2 x 6): 144 contiguous bytes
    a0 = a[3 * (size_t)p]; a1 = a[3 * (size_t)p + 1]; a2 = a[3 * (size_t)p + 2];
#pragma unroll
    for (int i = 0; i < 3; ++i) { const double2 x = rt[i]; t[2 * i] = x.x; t[2 * i + 1] = x.y; }
#pragma unroll
    for (int i = 0; i < 6; ++i) { const double2 x = rt[3 + i]; jp[2 * i] = x.x; jp[2 * i + 1] = x.y; }
  };
  fetch(e0 < e1 ? e0 : pose_start[0]);
  const PcgDecision d = kDir ? PcgDecide(it, G, part, st, eta, max_iterations) : PcgCarry(it, part, st);
  if (c == 0 && threadIdx.x == 0) { *st_out = d.s; if (d.failed) atomicOr(flag, 1); }
  if (!d.run) return;
  const bool first = kDir && it == 0;
  if (kDir) {
#pragma unroll
    for (int j = 0; j < 6; ++j) vc[j] = PcgDirection(first, d.beta, vc[j], po[j]);
    if (threadIdx.x < 6) {
      double mine = vc[0];
#pragma unroll
      for (int j = 1; j < 6; ++j) mine = ((int)threadIdx.x == j) ? vc[j] : mine;
      p_new[6 * (size_t)c + threadIdx.x] = mine;
    }
  }
  double acc[6] = {0, 0, 0, 0, 0, 0};
  for (int e = e0; e < e1; e += 256) {
    if (e != e0) fetch(e);
    double m0 = -(t[0] * a0 + t[1] * a1 + t[2] * a2), m1 = -(t[3] * a0 + t[4] * a1 + t[5] * a2);
#pragma unroll
    for (int j = 0; j < 6; ++j) { m0 += jp[j] * vc[j]; m1 += jp[6 + j] * vc[j]; }
#pragma unroll
    for (int j = 0; j < 6; ++j) acc[j] += jp[j] * m0 + jp[6 + j] * m1;
  }
#pragma unroll
  for (int j = 0; j < 6; ++j) acc[j] = WaveSum(acc[j]);
  if (lane == 0) {
#pragma unroll
    for (int j = 0; j < 6; ++j) red[wv][j] = acc[j];
  }
  __syncthreads();
  if (threadIdx.x < 64) {      // lanes 0..5: one component each (constant column: identity row, as the assembled system has it), then lane 0 adds v . S v in component order
    const int j = threadIdx.x < 6 ? (int)threadIdx.x : 0;
    double mine = vc[0];
#pragma unroll
    for (int k = 1; k < 6; ++k) mine = (j == k) ? vc[k] : mine;
    const double dd = (sc == 0.0) ? 1.0 : dg * inv_radius;
    const double qv = ((red[0][j] + red[1][j]) + red[2][j]) + red[3][j] + dd * mine;
    if (threadIdx.x < 6) out[6 * (size_t)c + j] = qv;
    const double term = mine * qv;
    double dot = 0.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) dot += __shfl(term, k);
    if (threadIdx.x == 0) dotp[c] = dot;
  }
}

// ---- the product kernels with COALESCED record reads.  A lane that reads its own record takes nine 16-byte pieces at a 192-byte stride: 64
// distinct lines per wave-level load, ~15 line requests per record and product - the texture-address path is what bounds k_pcg_points_dir /
// k_pcg_images_dir, not bytes and not latency.  Here FOUR lanes share a record: in load i lane q takes piece 4 i + q, so a quad reads 64 contiguous
// bytes per load and the record in three.  What a lane holds (pieces of 2 doubles: T = pieces 0-2, J_pose row 0 = 3-5, row 1 = 6-8, J_point = 9-11):
//   lane 0: T[0:2]   Jc0[2:4]  Jc1[4:6]          lane 2: T[4:6]    Jc1[0:2]  Jx[2:4]
//   lane 1: T[2:4]   Jc0[4:6]  Jx[0:2]           lane 3: Jc0[0:2]  Jc1[2:4]  Jx[4:6]
// every lane forms the terms of the two-vector m = J_c v (- T a) it has the factors for, the quad adds them up (two DPP moves), and every lane adds
// the products of m with ITS pieces of J_point (points) / J_pose (images) to private sums that are sorted into components once, at the end.
__device__ __forceinline__ double QuadSum(double v) {
  v += DppMove<0xB1>(v);       // quad_perm [1,0,3,2]
  v += DppMove<0x4E>(v);       // quad_perm [2,3,0,1]
  return v;
}
struct QuadRecord { double2 r0, r1, r2; };
__device__ __forceinline__ QuadRecord LoadQuadRecord(const double* __restrict__ rec, int o, int q) {
  const double2* base = reinterpret_cast<const double2*>(rec + kRecStride * (size_t)o);
  QuadRecord r;
  r.r0 = base[q]; r.r1 = base[4 + q]; r.r2 = base[8 + q];
  return r;
}

// (workgroups of 1024: at 256 threads the 2750 workgroups of 1100 images / 22 000 points took ~5 us to DISPATCH - a launch whose workgroups return at once was 6.2 us)
constexpr int kPointsQThreads = 1024;
template <bool kDir, int kQuads>      // kQuads quads (records in flight) per point: 2, 4 or 8
__global__ __launch_bounds__(kPointsQThreads) void k_pcg_points_q(int P, const int32_t* __restrict__ pt_start, const int2* __restrict__ pt_entry, const double* __restrict__ rec,
                                                                  const double* __restrict__ v, const double* __restrict__ p_old, double* __restrict__ a,
                                                                  const PcgState* __restrict__ st, const double* __restrict__ part, int G, int it, double eta, int max_iterations) {
  const int gid = blockIdx.x * kPointsQThreads + threadIdx.x;
  const int q = gid & 3, quad = gid >> 2;
  const int p = quad / kQuads, slot = quad % kQuads;
  int e0 = 0, e1 = 0;
  if (p < P) { e0 = pt_start[p] + slot; e1 = pt_start[p + 1]; }
  const bool first = kDir && it == 0;
  // the pairs of the direction this lane multiplies with: lane 0: (2,3) and (4,5); lane 1: (4,5); lane 2: (0,1); lane 3: (0,1) and (2,3)
  const int ta = q == 0 ? 1 : (q == 1 ? 2 : 0), tb = q == 0 ? 2 : (q == 3 ? 1 : ta);
  QuadRecord r;
  double2 va, vb, pa = make_double2(0.0, 0.0), pb = make_double2(0.0, 0.0);
  auto fetch_at = [&](const int2 en) {
    r = LoadQuadRecord(rec, en.x, q);
    const double2* vv = reinterpret_cast<const double2*>(v + 6 * (size_t)en.y);
    va = vv[ta]; vb = vv[tb];
    if (kDir && !first) { const double2* pp = reinterpret_cast<const double2*>(p_old + 6 * (size_t)en.y); pa = pp[ta]; pb = pp[tb]; }
  };
  auto fetch = [&](int e) { fetch_at(pt_entry[e]); };
  {      // the first entry and "has the loop ended" travel together; the record fetch goes out before the decision, whose loads and wavefront sums run beside it
    const int2 en0 = pt_entry[e0 < e1 ? e0 : 0];
    if (PcgEnded(st, it, kDir)) return;
    fetch_at(en0);
  }
  const PcgDecision d = kDir ? PcgDecide(it, G, part, st, eta, max_iterations) : PcgCarry(it, part, st);
  if (!d.run) return;
  double acc0 = 0.0, acc1 = 0.0;      // lane 1: components 0, 1 (m0); lane 2: 2 (m0), 0 (m1); lane 3: 1, 2 (m1)
  for (int e = e0; e < e1; e += kQuads) {
    if (e != e0) fetch(e);
    if (kDir) {
      va.x = PcgDirection(first, d.beta, va.x, pa.x); va.y = PcgDirection(first, d.beta, va.y, pa.y);
      vb.x = PcgDirection(first, d.beta, vb.x, pb.x); vb.y = PcgDirection(first, d.beta, vb.y, pb.y);
    }
    const double2 fa = q == 3 ? r.r0 : r.r1, fb = q == 0 ? r.r2 : r.r1;      // the J_pose pieces that go with va / vb
    const double da = fa.x * va.x + fa.y * va.y, db = fb.x * vb.x + fb.y * vb.y;
    const double m0 = QuadSum(q == 2 ? 0.0 : da);                              // lanes 0, 1, 3 hold row 0's pieces in `fa`
    const double m1 = QuadSum((q == 2 ? da : 0.0) + ((q == 0 || q == 3) ? db : 0.0));
    // J_point^T m: lane 1 has (x00, x01), lane 2 (x02, x10), lane 3 (x11, x12)
    acc0 += r.r2.x * (q == 1 || q == 2 ? m0 : m1);
    acc1 += r.r2.y * (q == 1 ? m0 : m1);
  }
  // sort the private sums into components: c0 = lane1.acc0 + lane2.acc1, c1 = lane1.acc1 + lane3.acc0, c2 = lane2.acc0 + lane3.acc1
  double c[3];
  c[0] = (q == 1 ? acc0 : 0.0) + (q == 2 ? acc1 : 0.0);
  c[1] = (q == 1 ? acc1 : 0.0) + (q == 3 ? acc0 : 0.0);
  c[2] = (q == 2 ? acc0 : 0.0) + (q == 3 ? acc1 : 0.0);
#pragma unroll
  for (int k = 0; k < 3; ++k) {      // the 4 kQuads lanes of a point: fixed order - quad, half row, row, (two rows)
    c[k] = QuadSum(c[k]);
    if (kQuads >= 2) c[k] += DppMove<0x141>(c[k]);      // row_half_mirror
    if (kQuads >= 4) c[k] += DppMove<0x140>(c[k]);      // row_mirror: every lane holds the sum of its 16-lane row
    if (kQuads == 8) c[k] += __shfl_xor(c[k], 16);
  }
  if (p < P && slot == 0 && q == 0) { a[3 * (size_t)p] = c[0]; a[3 * (size_t)p + 1] = c[1]; a[3 * (size_t)p + 2] = c[2]; }
}

// (256 threads = 64 quads per image.  1024 threads - a few hundred observations in ONE round of dependent loads - was 25 us against 11: at 74 VGPRs one
// such workgroup fills a CU, and 1100 of them run in four rounds)
constexpr int kImagesQThreads = 256;
template <bool kDir>
__global__ __launch_bounds__(kImagesQThreads) void k_pcg_images_q(int C, const int32_t* __restrict__ pose_start, const int2* __restrict__ pose_entry, const double* __restrict__ rec,
                                                      const double* __restrict__ v, const double* __restrict__ p_old, double* __restrict__ p_new,
                                                      const double* __restrict__ a, const double* __restrict__ scale_c, const double* __restrict__ diag_c, double inv_radius,
                                                      double* __restrict__ out, double* __restrict__ dotp, const PcgState* __restrict__ st, PcgState* __restrict__ st_out,
                                                      const double* __restrict__ part, int G, int it, double eta, int max_iterations, int32_t* __restrict__ flag) {
  __shared__ double red[kImagesQThreads / 64][6];
  const int c = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6, q = threadIdx.x & 3;
  const int e0 = pose_start[c] + (int)(threadIdx.x >> 2), e1 = pose_start[c + 1];
  double vc[6], po[6];
#pragma unroll
  for (int j = 0; j < 6; ++j) { vc[j] = v[6 * (size_t)c + j]; po[j] = (kDir && it != 0) ? p_old[6 * (size_t)c + j] : 0.0; }
  double sc = 1.0, dg = 0.0;
  if (threadIdx.x < 6) { sc = scale_c[6 * (size_t)c + threadIdx.x]; dg = diag_c[6 * (size_t)c + threadIdx.x]; }
  QuadRecord r;
  double a0, a1, a2;
  auto fetch_at = [&](const int2 en) {
    r = LoadQuadRecord(rec, en.x, q);      // (lanes 1-3 do not use their third piece: J_point)
    a0 = a[3 * (size_t)en.y]; a1 = a[3 * (size_t)en.y + 1]; a2 = a[3 * (size_t)en.y + 2];
  };
  auto fetch = [&](int e) { fetch_at(pose_entry[e]); };
  {
    const int2 en0 = pose_entry[e0 < e1 ? e0 : pose_start[0]];
    if (PcgEnded(st, it, kDir)) {      // (both copies of the state keep saying so)
      if (c == 0 && threadIdx.x == 0) *st_out = *st;
      return;
    }
    fetch_at(en0);
  }
  const PcgDecision d = kDir ? PcgDecide(it, G, part, st, eta, max_iterations) : PcgCarry(it, part, st);
  if (c == 0 && threadIdx.x == 0) { *st_out = d.s; if (d.failed) atomicOr(flag, 1); }
  if (!d.run) return;
  const bool first = kDir && it == 0;
  if (kDir) {
#pragma unroll
    for (int j = 0; j < 6; ++j) vc[j] = PcgDirection(first, d.beta, vc[j], po[j]);
    if (threadIdx.x < 6) {
      double mine = vc[0];
#pragma unroll
      for (int j = 1; j < 6; ++j) mine = ((int)threadIdx.x == j) ? vc[j] : mine;
      p_new[6 * (size_t)c + threadIdx.x] = mine;
    }
  }
  // this lane's pairs of v_c: the one that goes with its first J_pose piece and the one that goes with its second
  //   lane 0: Jc0[2:4] (v 2,3), Jc1[4:6] (v 4,5)   lane 1: Jc0[4:6] (v 4,5), -   lane 2: Jc1[0:2] (v 0,1), -   lane 3: Jc0[0:2] (v 0,1), Jc1[2:4] (v 2,3)
  const double vax = q == 0 ? vc[2] : (q == 1 ? vc[4] : vc[0]), vay = q == 0 ? vc[3] : (q == 1 ? vc[5] : vc[1]);
  const double vbx = q == 0 ? vc[4] : vc[2], vby = q == 0 ? vc[5] : vc[3];
  double u0 = 0.0, u1 = 0.0, u2 = 0.0, u3 = 0.0;
  for (int e = e0; e < e1; e += kImagesQThreads / 4) {
    if (e != e0) fetch(e);
    const double2 fa = q == 3 ? r.r0 : r.r1, fb = q == 0 ? r.r2 : r.r1;
    const double da = fa.x * vax + fa.y * vay, db = fb.x * vbx + fb.y * vby;
    // T a: lane 0 (T00, T01) -> m0; lane 1 (T02 -> m0, T10 -> m1); lane 2 (T11, T12) -> m1
    const double t0 = q == 0 ? r.r0.x * a0 + r.r0.y * a1 : (q == 1 ? r.r0.x * a2 : 0.0);
    const double t1 = q == 1 ? r.r0.y * a0 : (q == 2 ? r.r0.x * a1 + r.r0.y * a2 : 0.0);
    const double m0 = QuadSum((q == 2 ? 0.0 : da) - t0);
    const double m1 = QuadSum((q == 2 ? da : 0.0) + ((q == 0 || q == 3) ? db : 0.0) - t1);
    const double ma = q == 2 ? m1 : m0;      // the row `fa` belongs to
    u0 += fa.x * ma; u1 += fa.y * ma;
    u2 += fb.x * m1; u3 += fb.y * m1;        // (lanes 1, 2: not used below)
  }
  double acc[6];
  acc[0] = (q == 2 || q == 3) ? u0 : 0.0; acc[1] = (q == 2 || q == 3) ? u1 : 0.0;
  acc[2] = (q == 0 ? u0 : 0.0) + (q == 3 ? u2 : 0.0); acc[3] = (q == 0 ? u1 : 0.0) + (q == 3 ? u3 : 0.0);
  acc[4] = (q == 1 ? u0 : 0.0) + (q == 0 ? u2 : 0.0); acc[5] = (q == 1 ? u1 : 0.0) + (q == 0 ? u3 : 0.0);
#pragma unroll
  for (int j = 0; j < 6; ++j) acc[j] = WaveSumDpp(acc[j]);
  if (lane == 0) {
#pragma unroll
    for (int j = 0; j < 6; ++j) red[wv][j] = acc[j];
  }
  __syncthreads();
  if (threadIdx.x < 64) {
    const int j = threadIdx.x < 6 ? (int)threadIdx.x : 0;
    double mine = vc[0];
#pragma unroll
    for (int k = 1; k < 6; ++k) mine = (j == k) ? vc[k] : mine;
    const double dd = (sc == 0.0) ? 1.0 : dg * inv_radius;
    double rs = 0.0;
#pragma unroll
    for (int w = 0; w < kImagesQThreads / 64; ++w) rs += red[w][j];      // wave order
    const double qv = rs + dd * mine;
    if (threadIdx.x < 6) out[6 * (size_t)c + j] = qv;
    const double term = mine * qv;
    double dot = 0.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) dot += __shfl(term, k);
    if (threadIdx.x == 0) dotp[c] = dot;
  }
}

// k_pcg_wide_a with the loads first, and the block inverses at the start of a solve (mode 0)
__global__ __launch_bounds__(kWideThreads) void k_pcg_step(int mode, int it, int C, const double* __restrict__ Sd, const double* __restrict__ b, double* __restrict__ x,
                                                           double* __restrict__ r, double* __restrict__ z, const double* __restrict__ p, const double* __restrict__ q,
                                                           double* __restrict__ binv, const double* __restrict__ dotp, const PcgState* __restrict__ st,
                                                           double* __restrict__ part, int32_t* __restrict__ flag) {
  __shared__ double red[4];
  __shared__ double red2[4][2];
  const int tid = threadIdx.x;
  const int g = blockIdx.x * kWideThreads + tid;      // parameter block: rows 3 g .. 3 g + 2
  const bool active = g < 2 * C;
  const size_t i = 3 * (size_t)(active ? g : 0);
  // ---- every load of the step (none depends on alpha) ----
  PcgState s;
  if (mode != 0) s = *st;
  double pq = 0.0;
  if (mode == 1) { for (int c = tid; c < C; c += kWideThreads) pq += dotp[c]; }
  double bv[3], xv[3] = {0, 0, 0}, rv[3] = {0, 0, 0}, pv[3] = {0, 0, 0}, qv[3] = {0, 0, 0}, B[9];
#pragma unroll
  for (int k = 0; k < 3; ++k) bv[k] = b[i + k];
  if (mode != 0) {
#pragma unroll
    for (int k = 0; k < 3; ++k) { xv[k] = x[i + k]; qv[k] = q[i + k]; }
#pragma unroll
    for (int k = 0; k < 9; ++k) B[k] = binv[9 * (size_t)(active ? g : 0) + k];
  }
  if (mode == 1) {
#pragma unroll
    for (int k = 0; k < 3; ++k) { pv[k] = p[i + k]; rv[k] = r[i + k]; }
  }
  if (mode == 0) {      // the inverse of this parameter block's 3 x 3 diagonal block of S (rotation tangent / translation), as k_pcg_block_inverse forms it
    const int c = (active ? g : 0) >> 1, hh = g & 1;
    const double* D = Sd + 36 * (size_t)c + 21 * hh;
    const double a = D[0], bb = D[1], cc = D[2], dd = D[7], e = D[8], f = D[14];
    const double c00 = dd * f - e * e, c01 = cc * e - bb * f, c02 = bb * e - cc * dd;
    const double det = a * c00 + bb * c01 + cc * c02;
    if (active && (!(det > 0.0) || !isfinite(det))) atomicOr(flag, 1);
    const double id = 1.0 / det;
    B[0] = c00 * id; B[1] = c01 * id; B[2] = c02 * id;
    B[3] = c01 * id; B[4] = (a * f - cc * cc) * id; B[5] = (bb * cc - a * e) * id;
    B[6] = c02 * id; B[7] = (bb * cc - a * e) * id; B[8] = (a * dd - bb * bb) * id;
    if (active) {
#pragma unroll
      for (int k = 0; k < 9; ++k) binv[9 * (size_t)g + k] = B[k];
    }
  }
  if (mode != 0 && (s.done || s.iter != it)) return;
  double alpha = 0.0;
  int status = kPcgRunning;
  if (mode == 1) {
    pq = WideBlockSum(pq, red);
    if (!(pq > 0.0) || isinf(pq)) status = isnan(pq) ? kPcgFailure : kPcgNoConvergence;      // indefinite direction: the iterate so far is the answer
    else { alpha = s.rho / pq; if (isinf(alpha)) status = kPcgFailure; }
  }
  const bool reset = mode == 1 && (it % kResidualResetPeriod) == 0;
  double s0 = 0.0, s1 = 0.0;
  if (active && status == kPcgRunning) {
    if (mode == 0) {
#pragma unroll
      for (int k = 0; k < 3; ++k) { xv[k] = 0.0; rv[k] = bv[k]; x[i + k] = 0.0; s0 += bv[k] * bv[k]; }
    } else if (mode == 1) {
#pragma unroll
      for (int k = 0; k < 3; ++k) { xv[k] = xv[k] + alpha * pv[k]; x[i + k] = xv[k]; }
      if (!reset) {
#pragma unroll
        for (int k = 0; k < 3; ++k) rv[k] = rv[k] - alpha * qv[k];
      }
    } else {
#pragma unroll
      for (int k = 0; k < 3; ++k) rv[k] = bv[k] - qv[k];
    }
    if (!reset) {
      if (mode != 0) {
#pragma unroll
        for (int k = 0; k < 3; ++k) s0 -= xv[k] * (bv[k] + rv[k]);
      }
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const double zv = B[3 * k] * rv[0] + B[3 * k + 1] * rv[1] + B[3 * k + 2] * rv[2];
        z[i + k] = zv; r[i + k] = rv[k];
        s1 += rv[k] * zv;
      }
    }
  }
  s0 = WaveSum(s0); s1 = WaveSum(s1);      // (both sums behind ONE pair of barriers; the order of k_pcg_wide_a's two workgroup sums)
  __syncthreads();
  if ((tid & 63) == 0) { red2[tid >> 6][0] = s0; red2[tid >> 6][1] = s1; }
  __syncthreads();
  if (tid == 0) {
    double* o = part + 4 * (size_t)blockIdx.x;
    o[0] = ((red2[0][0] + red2[1][0]) + red2[2][0]) + red2[3][0]; o[1] = ((red2[0][1] + red2[1][1]) + red2[2][1]) + red2[3][1]; o[2] = (double)status; o[3] = alpha;
  }
}

int PcgEnsureBuffers(pp_ba_impl* h) {
  if (h->pcg_state) return PP_OK;
  const size_t n = (size_t)h->n_red;      // 6 C pose columns, then the NI variable intrinsics
  int rc;
#define A(ptr, cnt) if ((rc = HandleAlloc(&h->ptr, (size_t)(cnt)))) return rc
  A(pcg_Sd, 36 * (size_t)h->C); A(pcg_binv, 18 * (size_t)h->C); A(pcg_b, n); A(pcg_r, n); A(pcg_z, n); A(pcg_p, 2 * n); A(pcg_q, n);      // (pcg_p: two copies, the fused direction update ping-pongs)
  A(pcg_a, 3 * (size_t)h->P); A(pcg_dot, (size_t)h->C + (h->NI > 0 ? h->K : 0));
  A(pcg_part, 4 * (size_t)(CeilDiv(2 * (int64_t)h->C, kWideThreads) + (h->NI > 0 ? CeilDiv(h->K, kWideThreads) : 0)));
  if (h->NI > 0) { A(pcg_tk, 2 * (size_t)h->M); A(pcg_w, 2 * (size_t)h->M); A(pcg_Scomp, 12 * (size_t)h->NI); A(pcg_binvI, 12 * (size_t)h->NI); }
  { const int rcp = PoolDeviceAlloc(reinterpret_cast<void**>(&h->pcg_state), 2 * sizeof(PcgState)); if (rcp) return rcp; }      // (two copies: the many-workgroup vector step ping-pongs)
#undef A
  { const int rcp = PoolPinnedAlloc(reinterpret_cast<void**>(&h->pcg_state_host), sizeof(PcgState)); if (rcp) return rcp; }
  // the list entries with what they point at beside them (one dependent load less per product kernel)
  { int rcp = PoolDeviceAlloc(reinterpret_cast<void**>(&h->pcg_pt_entry), std::max<size_t>(1, (size_t)h->M) * sizeof(int2)); if (rcp) return rcp;
    rcp = PoolDeviceAlloc(reinterpret_cast<void**>(&h->pcg_pose_entry), std::max<size_t>(1, (size_t)h->M) * sizeof(int2)); if (rcp) return rcp; }
  if (h->M > 0) {
    hipLaunchKernelGGL(k_pcg_entries, dim3(CeilDiv(h->M, (int64_t)256)), dim3(256), 0, h->stream, h->M, h->pt_obs, h->obs_pose, h->pose_obs, h->obs_point,
                       reinterpret_cast<int2*>(h->pcg_pt_entry), reinterpret_cast<int2*>(h->pcg_pose_entry));
    PP_HIP_TRY(hipGetLastError());
  }
  return PP_OK;
}

void PcgFreeBuffers(pp_ba_impl* h) {
  double** bufs[] = {&h->pcg_Sd, &h->pcg_binv, &h->pcg_b, &h->pcg_r, &h->pcg_z, &h->pcg_p, &h->pcg_q, &h->pcg_a, &h->pcg_dot, &h->pcg_part, &h->pcg_tk, &h->pcg_w, &h->pcg_Scomp, &h->pcg_binvI};
  for (double** b : bufs) { if (*b) PoolDeviceFree(*b); *b = nullptr; }
  if (h->pcg_state) PoolDeviceFree(h->pcg_state);
  if (h->pcg_state_host) PoolPinnedFree(h->pcg_state_host);
  if (h->pcg_pt_entry) PoolDeviceFree(h->pcg_pt_entry);
  if (h->pcg_pose_entry) PoolDeviceFree(h->pcg_pose_entry);
  h->pcg_state = nullptr; h->pcg_state_host = nullptr; h->pcg_pt_entry = nullptr; h->pcg_pose_entry = nullptr;
}

// S x = b for the system k_schur_self_rhs (compact) + k_prepare have set up for `radius`; x -> h->step_c (scaled space).
// Synchronises the stream (the loop's length is data dependent).  *iterations: conjugate-gradient iterations run.
// the host's wait for k_pcg_decide: a busy spin on the ticket for as long as such a look can reasonably take (PPSFM_TICKET_SPIN_US, default 1500 us,
// 0 = never spin), then naps; after 2 s the stream is synchronised once (a failed launch would otherwise wait forever)
static int WaitPcgTicket(pp_ba_impl* h, int32_t ticket) {
  static const long spin_us = []() { const char* e = std::getenv("PPSFM_TICKET_SPIN_US"); return e ? std::atol(e) : 1500L; }();
  const volatile int32_t* t = &h->pcg_state_host->pad_;
  if (spin_us == 0) { PP_HIP_TRY(hipStreamSynchronize(h->stream)); }
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned spins = 0; *t != ticket; ++spins) {
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
    __builtin_ia32_pause();
#endif
    if ((spins & 0xFF) != 0xFF) continue;
    const auto waited = std::chrono::steady_clock::now() - t0;
    if (waited > std::chrono::seconds(2)) {
      PP_HIP_TRY(hipStreamSynchronize(h->stream));
      if (*t != ticket) { SetLastError("pp_ba_solve: the conjugate-gradient state never arrived"); return PP_ERR_HIP; }
      break;
    }
    if (waited > std::chrono::microseconds(spin_us)) std::this_thread::sleep_for(std::chrono::microseconds(50));
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  return PP_OK;
}

static int PcgFinishCount(pp_ba_impl* h, int* iterations) {
  const PcgState* hs = h->pcg_state_host;
  if (iterations) *iterations = hs->iter;
  static const bool log = getenv("PPSFM_PCG_LOG") != nullptr;      // (tools: the iteration counts a look schedule has to predict)
  if (log) fprintf(stderr, "pcg: %d iterations (previous solve %d)\n", hs->iter, h->pcg_last_iterations);
  h->pcg_last_iterations = hs->iter;
  return PP_OK;
}

// The three-launch loop (the default outside point-sharded groups).
static int PcgFusedRun(pp_ba_impl* h, double inv_radius, int max_iterations, double eta, int* iterations) {
  hipStream_t s = h->stream;
  const int n = 6 * h->C, C = h->C;
  const int G = CeilDiv(2 * (int64_t)C, kWideThreads);
  const int cap = std::max(1, max_iterations);
  PcgState* hs = h->pcg_state_host;
  int batch = h->pcg_last_iterations > 0 ? std::min(32, h->pcg_last_iterations + 1) : 8;
  int next_look = batch;
  // ---- three launches per iteration: the product kernels take the decision and form the direction themselves ----
  const int2* pt_entry = reinterpret_cast<const int2*>(h->pcg_pt_entry);
  const int2* pose_entry = reinterpret_cast<const int2*>(h->pcg_pose_entry);
  const bool eight = h->M > (int64_t)h->P * 9 / 2;      // eight lanes per point when the mean track is longer than 4.5
  const char* quad_env = getenv("PPSFM_PCG_QUAD");
  const bool quad = !(quad_env && atoi(quad_env) == 0);      // four lanes per record, coalesced reads (0: a lane per record)
  int cur = 0, pc = 0;      // current copy of the state / of the direction
  auto step = [&](int mode, int it) {
    hipLaunchKernelGGL(k_pcg_step, dim3(G), dim3(kWideThreads), 0, s, mode, it, C, h->pcg_Sd, h->pcg_b, h->step_c, h->pcg_r, h->pcg_z, h->pcg_p + (size_t)pc * n, h->pcg_q,
                       h->pcg_binv, h->pcg_dot, h->pcg_state + cur, h->pcg_part, h->d_flag);
  };
  auto product = [&](bool dir, const double* v, int it) {      // dir: pcg_q = S (z + beta p) after the decision on iteration `it`; else pcg_q = S v
    const double* p_old = h->pcg_p + (size_t)pc * n;
    double* p_new = h->pcg_p + (size_t)(pc ^ 1) * n;
    const dim3 gp(CeilDiv((eight ? 8 : 4) * (int64_t)h->P, 256)), gi(C), b(256);
#define PP_POINTS(D, L) hipLaunchKernelGGL((k_pcg_points_dir<D, L>), gp, b, 0, s, h->P, h->pt_start, pt_entry, h->JpS, v, p_old, h->pcg_a, h->pcg_state + cur, h->pcg_part, G, \
                                         it, eta, max_iterations)
#define PP_IMAGES(D) hipLaunchKernelGGL((k_pcg_images_dir<D>), gi, b, 0, s, C, h->pose_start, pose_entry, h->JpS, v, p_old, p_new, h->pcg_a, h->scale_c, h->diag_c, inv_radius, \
                                      h->pcg_q, h->pcg_dot, h->pcg_state + cur, h->pcg_state + (cur ^ 1), h->pcg_part, G, it, eta, max_iterations, h->d_flag)
#define PP_POINTS_Q(D, L) hipLaunchKernelGGL((k_pcg_points_q<D, L>), gq, dim3(kPointsQThreads), 0, s, h->P, h->pt_start, pt_entry, h->JpS, v, p_old, h->pcg_a, h->pcg_state + cur, h->pcg_part, G, \
                                           it, eta, max_iterations)
#define PP_IMAGES_Q(D) hipLaunchKernelGGL((k_pcg_images_q<D>), gi, dim3(kImagesQThreads), 0, s, C, h->pose_start, pose_entry, h->JpS, v, p_old, p_new, h->pcg_a, h->scale_c, h->diag_c, inv_radius, \
                                        h->pcg_q, h->pcg_dot, h->pcg_state + cur, h->pcg_state + (cur ^ 1), h->pcg_part, G, it, eta, max_iterations, h->d_flag)
    if (quad) {
      // quads per point (a record each per round): 2 up to a mean track of 2.5, 4 up to 4.5, 8 beyond (two records in flight per quad with half
      // the quads: 11.3 us against 9.1 - the early exit waits for both)
      const int qp = eight ? 8 : (h->M > (int64_t)h->P * 5 / 2 ? 4 : 2);
      const dim3 gq(CeilDiv(4 * qp * (int64_t)h->P, kPointsQThreads));
      if (dir) { if (qp == 8) PP_POINTS_Q(true, 8); else if (qp == 4) PP_POINTS_Q(true, 4); else PP_POINTS_Q(true, 2); PP_IMAGES_Q(true); pc ^= 1; }
      else { if (qp == 8) PP_POINTS_Q(false, 8); else if (qp == 4) PP_POINTS_Q(false, 4); else PP_POINTS_Q(false, 2); PP_IMAGES_Q(false); }
    } else if (dir) { if (eight) PP_POINTS(true, 8); else PP_POINTS(true, 4); PP_IMAGES(true); pc ^= 1; }
    else { if (eight) PP_POINTS(false, 8); else PP_POINTS(false, 4); PP_IMAGES(false); }
#undef PP_POINTS
#undef PP_IMAGES
#undef PP_POINTS_Q
#undef PP_IMAGES_Q
    cur ^= 1;
  };
  hs->done = 0; hs->iter = 0; hs->status = kPcgRunning;
  step(0, 0);
  for (int it = 1; it <= cap; ++it) {
    product(true, h->pcg_z, it - 1);
    step(1, it);
    if (it % kResidualResetPeriod == 0) { product(false, h->step_c, it); step(2, it); }
    if (it == next_look || it == cap) {
      // (the state goes straight into the pinned block the host reads: no copy in between)
      const int32_t ticket = ++h->pcg_ticket == 0 ? ++h->pcg_ticket : h->pcg_ticket;
      hs->pad_ = 0;
      hipLaunchKernelGGL(k_pcg_decide, dim3(1), dim3(64), 0, s, it, G, h->pcg_part, h->pcg_state + cur, hs, eta, max_iterations, h->d_flag, ticket);
      PP_HIP_TRY(hipGetLastError());
      { const int rcw = WaitPcgTicket(h, ticket); if (rcw) return rcw; }
      if (hs->done) break;
      batch = it == next_look && next_look > batch ? std::min(32, batch * 2) : 2;
      next_look = it + batch;
    }
  }
  return PcgFinishCount(h, iterations);
}

int PcgSolve(pp_ba_impl* h, double radius, int max_iterations, double eta, int* iterations) {
  hipStream_t s = h->stream;
  const int n = h->n_red, C = h->C;
  const double inv_radius = 1.0 / radius;
  const bool group = BaInGroup(h);
  const bool intr = h->NI > 0;      // variable intrinsics: the four-launch form with the per-camera kernels (see PcgIntr)
  const int Gp = CeilDiv(2 * (int64_t)C, kWideThreads);
  const int G = Gp + (intr ? CeilDiv(h->K, kWideThreads) : 0);
  PcgIntr in;
  if (intr) { in.K = h->K; in.pose_blocks = Gp; in.intr_off = h->intr_off; in.intr_nv = h->intr_nv; in.binvI = h->pcg_binvI; }
  const int cap = std::max(1, max_iterations);
  // iterations enqueued before the first look at the state: what the previous solve of this handle needed, + 1 (consecutive LM iterations take about
  // the same number of CG iterations - ~6 at 1100 images with eta = 0.1 - and every iteration enqueued beyond the end is launches of kernels that
  // return at once plus the wait for them: a fixed first batch of 8 was ~10 % of such an LM iteration); then doubling
  int batch = h->pcg_last_iterations > 0 ? std::min(32, h->pcg_last_iterations + 1) : 8;
  int next_look = batch;
  PcgState* hs = h->pcg_state_host;
  hs->done = 0; hs->iter = 0; hs->status = kPcgRunning;
  const char* fused_env = getenv("PPSFM_PCG_FUSED");
  const char* wide_env0 = getenv("PPSFM_PCG_WIDE");
  if (!group && !intr && !(fused_env && atoi(fused_env) == 0) && !(wide_env0 && atoi(wide_env0) == 0)) {
    return PcgFusedRun(h, inv_radius, max_iterations, eta, iterations);
  }
  hipLaunchKernelGGL(k_pcg_block_inverse, dim3(CeilDiv(2 * C, 256)), dim3(256), 0, s, C, h->pcg_Sd, h->pcg_binv, h->d_flag);
  if (intr) hipLaunchKernelGGL(k_pcg_intr_inverse, dim3(CeilDiv(h->K, 64)), dim3(64), 0, s, h->K, h->intr_off, h->intr_nv, h->pcg_Scomp, h->pcg_binvI, h->d_flag);
  // the vector step: many workgroups (k_pcg_wide_a / _b, two launches) - 1100 images: 2460 -> 2860 LM it/s against the one-workgroup
  // kernel (12.5 us per step; 29.6 us at 4000 images, where it was the longest kernel of an iteration, against 6.1 + 4.4 us); 600 images
  // +6 %.  In a point-sharded group p . S p belongs to the all-reduced product: k_pcg_dot forms its per-image parts after the exchange.
  // PPSFM_PCG_WIDE = 0 forces the one-workgroup kernel (tests compare the two).
  const char* wide_env = getenv("PPSFM_PCG_WIDE");
  const bool wide = intr || !(wide_env && atoi(wide_env) == 0);
  int cur = 0;      // which copy of the state is current (wide: ping-pong; otherwise always 0)
  auto wide_a = [&](int mode, int it) {
    hipLaunchKernelGGL(k_pcg_wide_a, dim3(G), dim3(kWideThreads), 0, s, mode, it, C, h->pcg_b, h->step_c, h->pcg_r, h->pcg_z, h->pcg_p, h->pcg_q, h->pcg_binv, h->pcg_dot,
                       h->pcg_state + cur, h->pcg_part, in);
  };
  auto wide_b = [&](int mode, int it) {
    hipLaunchKernelGGL(k_pcg_wide_b, dim3(G), dim3(kWideThreads), 0, s, mode, it, C, G, h->pcg_z, h->pcg_p, h->pcg_part, h->pcg_state + cur, h->pcg_state + (cur ^ 1), eta,
                       max_iterations, h->d_flag, in);
    cur ^= 1;
  };
  if (wide) { wide_a(0, 0); wide_b(0, 0); }
  else hipLaunchKernelGGL(k_pcg_vec, dim3(1), dim3(kVecThreads), 0, s, 0, 0, n, C, h->pcg_b, h->step_c, h->pcg_r, h->pcg_z, h->pcg_p, h->pcg_q, h->pcg_binv, h->pcg_dot,
                          h->pcg_state, eta, max_iterations, h->d_flag);
  // A point-sharded group (pp_ba_set_communicator / pp_ba_set_allreduce): every rank applies S to the same vector with ITS points'
  // observations (a point's observations all live on its owner, so the partial products simply add up), the products are summed over the
  // group - 6 C doubles per product, 24 KB at 500 images, against the 36 MB lower triangle the direct solver exchanges per LM iteration -
  // and every rank runs the same vector updates on the same data: identical decisions, identical iterates, no further exchange.
  double* dotp = group ? nullptr : h->pcg_dot;
  int rc_group = PP_OK;
  auto apply = [&](const double* v) {      // pcg_q = S v (and pcg_dot = the per-image parts of v . S v)
    const double2* tk = intr ? reinterpret_cast<const double2*>(h->pcg_tk) : nullptr;
    double2* w = intr ? reinterpret_cast<double2*>(h->pcg_w) : nullptr;
    if (intr) hipLaunchKernelGGL(k_pcg_cam_t, dim3(CeilDiv(h->M, (int64_t)256)), dim3(256), 0, s, h->M, C, h->obs_cam, h->intr_off, h->intr_nv, h->JkS_intr, v,
                                 reinterpret_cast<double2*>(h->pcg_tk), h->pcg_state + cur);
    hipLaunchKernelGGL(k_pcg_points, dim3(CeilDiv(4 * (int64_t)h->P, 256)), dim3(256), 0, s, h->P, h->pt_start, h->pt_obs, h->obs_pose, h->JpS, v, h->pcg_a, h->pcg_state + cur, tk);
    hipLaunchKernelGGL(k_pcg_images, dim3(C), dim3(256), 0, s, C, h->pose_start, h->pose_obs, h->obs_point, h->JpS, v, h->pcg_a, h->scale_c, h->diag_c, inv_radius,
                       h->pcg_q, dotp, h->pcg_state + cur, h->group_rank == 0 ? 1 : 0, tk, w);
    if (intr) {      // the intrinsics rows of the product and the cameras' parts of v . S v
      if (h->isum_num_chunks > 0)
        hipLaunchKernelGGL(k_pcg_cam_q, dim3((unsigned)h->isum_num_chunks), dim3(256), 0, s, h->isum_chunk, h->cam_obs, h->JkS_intr, (const double2*)w, h->isum_partial,
                           h->pcg_state + cur);
      hipLaunchKernelGGL(k_pcg_cam_q_reduce, dim3(h->K), dim3(64), 0, s, C, h->isum_cam_chunk, h->intr_off, h->intr_nv, h->isum_partial, v, h->diag_c, inv_radius, h->pcg_q,
                         h->pcg_dot, h->pcg_state + cur, h->group_rank == 0 ? 1 : 0);
    }
    if (group && rc_group == PP_OK) rc_group = BaGroupReduce(h, h->pcg_q, n, PP_REDUCE_SUM);
    // (the all-reduced product carries the intrinsics rows as well: n = 6 C + NI doubles; the cameras' parts of v . S v follow the images')
    if (group && wide) hipLaunchKernelGGL(k_pcg_dot, dim3(CeilDiv(C + (intr ? h->K : 0), 256)), dim3(256), 0, s, C, v, h->pcg_q, h->pcg_dot, h->pcg_state + cur, intr ? h->K : 0,
                                          (const int32_t*)h->intr_off, (const int32_t*)h->intr_nv);
  };
  for (int it = 1; it <= cap; ++it) {
    apply(h->pcg_p);
    if (rc_group) { (void)hipStreamSynchronize(s); return rc_group; }      // (what is already enqueued reads the buffers the caller may free next)
    const bool reset = it % kResidualResetPeriod == 0;
    if (wide) { wide_a(1, it); wide_b(reset ? 3 : 1, it); }
    else hipLaunchKernelGGL(k_pcg_vec, dim3(1), dim3(kVecThreads), 0, s, 1, it, n, C, h->pcg_b, h->step_c, h->pcg_r, h->pcg_z, h->pcg_p, h->pcg_q, h->pcg_binv, (const double*)dotp,
                            h->pcg_state, eta, max_iterations, h->d_flag);
    if (reset) {
      apply(h->step_c);
      if (rc_group) { (void)hipStreamSynchronize(s); return rc_group; }
      if (wide) { wide_a(2, it); wide_b(1, it); }
      else hipLaunchKernelGGL(k_pcg_vec, dim3(1), dim3(kVecThreads), 0, s, 2, it, n, C, h->pcg_b, h->step_c, h->pcg_r, h->pcg_z, h->pcg_p, h->pcg_q, h->pcg_binv, h->pcg_dot,
                              h->pcg_state, eta, max_iterations, h->d_flag);
    }
    if (it == next_look || it == cap) {
      PP_HIP_TRY(hipGetLastError());
      PP_HIP_TRY(hipMemcpyAsync(hs, h->pcg_state + cur, sizeof(PcgState), hipMemcpyDeviceToHost, s));
      PP_HIP_TRY(hipStreamSynchronize(s));
      if (hs->done) break;
      batch = it == next_look && next_look > batch ? std::min(32, batch * 2) : 2;      // (longer than last time: look again soon, then less and less often)
      if (group) batch = 1;      // (in a group every iteration enqueued beyond the end is a real collective of 6 C doubles on every rank: look after each one)
      next_look = it + batch;
    }
  }
  if (!hs->done) {      // (cap not a multiple of the batch and the loop still running: cannot happen - the cap ends it - but never trust a loop)
    PP_HIP_TRY(hipMemcpyAsync(hs, h->pcg_state + cur, sizeof(PcgState), hipMemcpyDeviceToHost, s));
    PP_HIP_TRY(hipStreamSynchronize(s));
  }
  if (iterations) *iterations = hs->iter;
  h->pcg_last_iterations = hs->iter;
  return PP_OK;
}

}  // namespace ppsfm

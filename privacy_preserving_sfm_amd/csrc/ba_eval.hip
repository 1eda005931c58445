// K1 — batched residual + Jacobian evaluation of the line-to-point reprojection cost, and the
// pp_ba_handle lifecycle.
//
// Replaces, for all M residual blocks at once, what Ceres does by calling
// AutoDiffCostFunction<BundleAdjustment[ConstantPose]LineCostFunction<CameraModel>,...>::Evaluate
// once per block from its thread pool (reference src/base/cost_functions.h:55-60, :130-137, call
// sites src/optim/bundle_adjustment.cc:381-415, :470-486).
//
// Roofline: HBM.  Algorithmic bytes per observation: 60 B in (line 24, two int32
// indices 8 (+4 amortised), point gather 24, pose/intrinsics amortised over the image's
// observations) + 160 B out (r 16, J_pose 2x6 96, J_point 2x3 48) = 220 B.
// Mapping: one lane per observation, 256-lane workgroups (4 wavefronts), >= 3 workgroups per CU at
// the 200k-observation size.  Line coefficients are SoA streams (coalesced 512 B / wavefront /
// stream).  Poses (56 B) and points (24 B) are gathered; with observations grouped by image the pose
// gather is wave-uniform and served by L1/L2.
#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <numeric>
#include <thread>

#include "ba_impl.hpp"
#include "resource_pool.hpp"
#include "line_residual.hpp"

namespace ppsfm {

struct EvalArgs {
  int64_t M;
  const double *la, *lb, *lc;
  const int32_t *obs_pose, *obs_point, *obs_cam;   // obs_cam = (intrinsics index << 4) | camera model id
  const double *poses, *points, *intr;
  double *r, *Jpose, *Jpoint, *Jcam;
  // Jcam rows: ambient (2 x kCamStride per observation: the C ABI's layout) when cam_col is null; otherwise COMPACT - only the variable parameters, column
  // cam_col[camera][parameter] (< cam_stride) of rows cam_stride wide: what the solver's own evaluations write (32 bytes per observation for two
  // variable parameters instead of 192, and contiguous across the lanes of a wavefront)
  const int32_t* cam_col;
  int cam_stride;
  double* partials;
  int loss_type;
  double loss_scale;
};

__device__ __forceinline__ void BlockPartialSum(double v, double* partials) {
  __shared__ double wsum[4];
  v = WaveSum(v);
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) wsum[wave] = v;
  __syncthreads();
  if (threadIdx.x == 0) partials[blockIdx.x] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
}

// MODE 0: residual/cost only; 1: tangent pose Jacobian (2x6); 2: ambient pose Jacobian (2x7)
// The Jacobian rows (96/112 + 48 bytes per observation) are staged through LDS so that the workgroup
// writes its contiguous 36 KB slab with fully coalesced 16-byte stores instead of 9 strided stores per lane.
template <int MODE, bool WANT_CAM, bool LOSS_CORRECT>
__global__ __launch_bounds__(256) void k_line_eval(EvalArgs a) {
  constexpr int JW = MODE == 2 ? 14 : 12;
  __shared__ __attribute__((aligned(16))) double sJp[MODE == 0 ? 2 : 256 * JW];
  __shared__ __attribute__((aligned(16))) double sJx[MODE == 0 ? 2 : 256 * 6];
  const int tid = threadIdx.x;
  const int64_t o0 = (int64_t)blockIdx.x * 256;
  const int64_t o = o0 + tid;
  double half_rho = 0.0;
  if (o < a.M) {
    // independent loads first: indices, line, then the gathers they feed
    const int c = a.obs_pose[o], p = a.obs_point[o], ck = a.obs_cam[o];
    const double la = a.la[o], lb = a.lb[o], lc = a.lc[o];
    const int model = ck & 15;
    const double* cam = a.intr + (size_t)kCamStride * (ck >> 4);
    const double* pose = a.poses + (size_t)7 * c;
    const double q[4] = {pose[0], pose[1], pose[2], pose[3]};
    const double t[3] = {pose[4], pose[5], pose[6]};
    const double X[3] = {a.points[3 * (size_t)p], a.points[3 * (size_t)p + 1], a.points[3 * (size_t)p + 2]};
    if (MODE == 0) {
      double r[2];
      LineResidualOnly(model, cam, q, t, X, la, lb, lc, r);
      double rho0, rho1;
      LossRho(a.loss_type, a.loss_scale, r[0] * r[0] + r[1] * r[1], &rho0, &rho1);
      half_rho = 0.5 * rho0;
      if (a.r) { a.r[2 * o] = r[0]; a.r[2 * o + 1] = r[1]; }
    } else {
      LineObsJac J;
      LineResidualJacobian<MODE == 2>(model, cam, q, t, X, la, lb, lc, &J);
      double rho0, rho1;
      LossRho(a.loss_type, a.loss_scale, J.r[0] * J.r[0] + J.r[1] * J.r[1], &rho0, &rho1);
      half_rho = 0.5 * rho0;
      const double sr = LOSS_CORRECT ? sqrt(rho1) : 1.0;  // Ceres Corrector with alpha = 0 (rho'' <= 0)
      double2* r2 = reinterpret_cast<double2*>(a.r);
      r2[o] = make_double2(sr * J.r[0], sr * J.r[1]);
      double2* jp = reinterpret_cast<double2*>(sJp + JW * tid);
      if (MODE == 1) {
        jp[0] = make_double2(sr * J.Jrot[0], sr * J.Jrot[1]);
        jp[1] = make_double2(sr * J.Jrot[2], sr * J.Jt[0]);
        jp[2] = make_double2(sr * J.Jt[1], sr * J.Jt[2]);
        jp[3] = make_double2(sr * J.Jrot[3], sr * J.Jrot[4]);
        jp[4] = make_double2(sr * J.Jrot[5], sr * J.Jt[3]);
        jp[5] = make_double2(sr * J.Jt[4], sr * J.Jt[5]);
      } else {
        jp[0] = make_double2(sr * J.Jq[0], sr * J.Jq[1]);
        jp[1] = make_double2(sr * J.Jq[2], sr * J.Jq[3]);
        jp[2] = make_double2(sr * J.Jt[0], sr * J.Jt[1]);
        jp[3] = make_double2(sr * J.Jt[2], sr * J.Jq[4]);
        jp[4] = make_double2(sr * J.Jq[5], sr * J.Jq[6]);
        jp[5] = make_double2(sr * J.Jq[7], sr * J.Jt[3]);
        jp[6] = make_double2(sr * J.Jt[4], sr * J.Jt[5]);
      }
      double2* jx = reinterpret_cast<double2*>(sJx + 6 * tid);
      jx[0] = make_double2(sr * J.JX[0], sr * J.JX[1]);
      jx[1] = make_double2(sr * J.JX[2], sr * J.JX[3]);
      jx[2] = make_double2(sr * J.JX[4], sr * J.JX[5]);
      if (WANT_CAM) {
        double jl[2 * kCamStride];
#pragma unroll
        for (int i = 0; i < 2 * kCamStride; ++i) jl[i] = 0.0;
        LineResidualCameraJacobian(model, cam, q, t, X, la, lb, lc, jl, kCamStride);
        if (LOSS_CORRECT) {
#pragma unroll
          for (int i = 0; i < 2 * kCamStride; ++i) jl[i] *= sr;
        }
        if (a.cam_col) {
          const int W = a.cam_stride;
          double* jc = a.Jcam + (size_t)2 * W * o;
          double2* z = reinterpret_cast<double2*>(jc);
          for (int i = 0; i < W; ++i) z[i] = make_double2(0.0, 0.0);      // (a camera with fewer variable parameters than the widest one)
          const int32_t* col = a.cam_col + (size_t)kCamStride * (ck >> 4);
#pragma unroll
          for (int i = 0; i < kCamStride; ++i) { const int cc = col[i]; if (cc >= 0) { jc[cc] = jl[i]; jc[W + cc] = jl[kCamStride + i]; } }
        } else {
          double2* jc = reinterpret_cast<double2*>(a.Jcam + (size_t)2 * kCamStride * o);
#pragma unroll
          for (int i = 0; i < kCamStride; ++i) jc[i] = make_double2(jl[2 * i], jl[2 * i + 1]);
        }
      }
    }
  }
  if (MODE != 0) {
    __syncthreads();
    const int64_t left = a.M - o0;
    const int nobs = left < 256 ? (int)left : 256;
    {  // J_pose slab: nobs * JW doubles, contiguous in global memory
      const int n2 = nobs * JW / 2;   // double2 chunks
      double2* dst = reinterpret_cast<double2*>(a.Jpose + (size_t)JW * o0);
      const double2* src = reinterpret_cast<const double2*>(sJp);
#pragma unroll
      for (int it = 0; it < (256 * JW / 2 + 255) / 256; ++it) {
        const int idx = it * 256 + tid;
        if (idx < n2) { const double2 v = src[idx]; __builtin_nontemporal_store(v.x, &dst[idx].x); __builtin_nontemporal_store(v.y, &dst[idx].y); }
      }
    }
    {
      const int n2 = nobs * 3;
      double2* dst = reinterpret_cast<double2*>(a.Jpoint + (size_t)6 * o0);
      const double2* src = reinterpret_cast<const double2*>(sJx);
#pragma unroll
      for (int it = 0; it < 3; ++it) {
        const int idx = it * 256 + tid;
        if (idx < n2) { const double2 v = src[idx]; __builtin_nontemporal_store(v.x, &dst[idx].x); __builtin_nontemporal_store(v.y, &dst[idx].y); }
      }
    }
  }
  BlockPartialSum(half_rho, a.partials);
}

// fixed-order final reduction of the per-block partial sums (deterministic cost)
__global__ __launch_bounds__(256) void k_sum_partials(const double* partials, int n, double* out) {
  __shared__ double sh[256];
  double acc = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) acc += partials[i];
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) *out = sh[0];
}

static EvalArgs MakeArgs(pp_ba_impl* h, const double* poses, const double* points, const double* intr = nullptr) {
  EvalArgs a;
  a.M = h->M; a.la = h->la; a.lb = h->lb; a.lc = h->lc;
  a.obs_pose = h->obs_pose; a.obs_point = h->obs_point; a.obs_cam = h->obs_cam;
  a.poses = poses; a.points = points; a.intr = intr ? intr : h->intr;
  a.r = h->r; a.Jpose = h->Jpose; a.Jpoint = h->Jpoint; a.Jcam = h->Jcam; a.cam_col = nullptr; a.cam_stride = kCamStride;
  a.partials = h->partials; a.loss_type = h->loss_type; a.loss_scale = h->loss_scale;
  return a;
}

int BaEnsureJacobianBuffers(pp_ba_impl* h, int jac_mode, int want_cam) {
  const int width = jac_mode == 1 ? 14 : 12;
  if (!h->Jpose || h->jpose_width < width) {
    if (h->Jpose) PoolDeviceFree(h->Jpose);
    h->Jpose = nullptr;
    int rc = HandleAlloc(&h->Jpose, (size_t)h->M * width);
    if (rc) return rc;
    h->jpose_width = width;
  }
  if (want_cam && !h->Jcam) {
    int rc = HandleAlloc(&h->Jcam, (size_t)h->M * 2 * kCamStride);
    if (rc) return rc;
  }
  return PP_OK;
}

int LaunchEval(pp_ba_impl* h, int jac_mode, int want_cam, bool loss_correct, const double* poses, const double* points,
               double* cost_slot, bool compact_cam) {
  EvalArgs a = MakeArgs(h, poses, points);
  if (want_cam) {      // (the readers of Jcam - IntrSumsAfterEval, IntrScaledJacobians - are told which layout the last evaluation left)
    h->jcam_compact = compact_cam && h->NI > 0;
    if (h->jcam_compact) { a.cam_col = h->intr_col; a.cam_stride = h->jcam_stride; }
  }
  const int grid = h->num_partials;
  hipStream_t s = h->stream;
  if (jac_mode == 0) {
    if (want_cam) { if (loss_correct) hipLaunchKernelGGL((k_line_eval<1, true, true>), dim3(grid), dim3(256), 0, s, a); else hipLaunchKernelGGL((k_line_eval<1, true, false>), dim3(grid), dim3(256), 0, s, a); }
    else { if (loss_correct) hipLaunchKernelGGL((k_line_eval<1, false, true>), dim3(grid), dim3(256), 0, s, a); else hipLaunchKernelGGL((k_line_eval<1, false, false>), dim3(grid), dim3(256), 0, s, a); }
  } else {
    if (want_cam) hipLaunchKernelGGL((k_line_eval<2, true, false>), dim3(grid), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((k_line_eval<2, false, false>), dim3(grid), dim3(256), 0, s, a);
  }
  if (cost_slot) hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(256), 0, s, h->partials, grid, cost_slot);
  PP_HIP_TRY(hipGetLastError());
  return PP_OK;
}

int LaunchCostOnly(pp_ba_impl* h, const double* poses, const double* points, const double* intr, double* cost_slot) {
  EvalArgs a = MakeArgs(h, poses, points, intr);
  a.r = nullptr;
  const int grid = h->num_partials;
  hipLaunchKernelGGL((k_line_eval<0, false, false>), dim3(grid), dim3(256), 0, h->stream, a);
  if (cost_slot) hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(256), 0, h->stream, h->partials, grid, cost_slot);   // else: summed by the caller's next kernel
  PP_HIP_TRY(hipGetLastError());
  return PP_OK;
}

// residuals (not loss-corrected) + cost, no Jacobians: what Ceres asks for at a trial point
static int LaunchResidualsOnly(pp_ba_impl* h, const double* poses, const double* points, double* cost_slot) {
  EvalArgs a = MakeArgs(h, poses, points);
  const int grid = h->num_partials;
  hipLaunchKernelGGL((k_line_eval<0, false, false>), dim3(grid), dim3(256), 0, h->stream, a);
  if (cost_slot) hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(256), 0, h->stream, h->partials, grid, cost_slot);
  PP_HIP_TRY(hipGetLastError());
  return PP_OK;
}

}  // namespace ppsfm

using namespace ppsfm;

extern "C" {

int pp_ba_destroy(pp_ba_handle h) try {
  if (!h) return PP_OK;
  (void)hipSetDevice(h->device);
  void* bufs[] = {h->la, h->lb, h->lc, h->obs_cam, h->obs_pose, h->obs_point, h->pose_camera, h->camera_model, h->pose_const,
                  h->tvec_mask, h->point_const, h->pt_start, h->pt_obs, h->pose_start, h->pose_obs, h->pair_start,
                  h->pair_ij, h->pair_entries, h->poses, h->points, h->intr, h->poses_c, h->points_c, h->r, h->Jpose,
                  h->Jpoint, h->Jcam, h->partials, h->U, h->gc, h->V, h->gp, h->Vinv, h->vb, h->scale_c, h->scale_p,
                  h->diag_c, h->diag_p, h->S, h->Linv, h->Lfac, h->step_c, h->step_p, h->scal, h->JpS, h->Q, h->norm_part,
                  h->intr_c, h->cam_np, h->intr_off, h->intr_nv, h->intr_col, h->cam_start, h->cam_obs, h->gen_pair, h->gen_pair_chunk, h->gen_chunk,
                  h->gen_multi, h->gen_grp_start, h->gen_grp_obs, h->gen_L, h->kk_entries, h->kk_pair, h->kk_pair_chunk, h->kk_chunk, h->kk_multi, h->kk_partial, h->gen_entries, h->isum_chunk, h->isum_cam_chunk, h->gen_partial, h->isum_partial, h->cnI, h->JkS_intr, h->Spack, h->nz_tile_list,
                  h->small_chunk, h->small_pair_chunk, h->small_partials, h->spos, h->step_s, h->attach_slot};
  if (h->stream) (void)hipStreamSynchronize(h->stream);      // nothing of this handle is in flight when its blocks go back to the pool (resource_pool.hpp)
  for (void* b : bufs) if (b) PoolDeviceFree(b);
  CholeskyAuxDestroy(&h->chol_aux);
  PcgFreeBuffers(h);
  for (int i = 0; i < 8; ++i) if (h->tev[i]) PoolEventRelease(h->tev[i], true);
  for (int i = 0; i < 2; ++i) if (h->tev_eval[i]) PoolEventRelease(h->tev_eval[i], true);
  if (h->h_scal) PoolPinnedFree(h->h_scal);
  { void* pins[] = {h->pin_r, h->pin_jpose, h->pin_jpoint, h->pin_jcam}; for (void* b : pins) if (b) (void)hipHostFree(b); }
  if (h->ev_readback) PoolEventRelease(h->ev_readback, false);
  if (h->ev0) PoolEventRelease(h->ev0, true);
  if (h->ev1) PoolEventRelease(h->ev1, true);
  if (h->stream) PoolStreamRelease(h->stream);
  delete h;
  return PP_OK;
} PP_API_CATCH("pp_ba_destroy")

int pp_ba_create(const pp_ba_problem_desc* d, int device, pp_ba_handle* out) try {
  PP_REQUIRE(d && out, "pp_ba_create: null argument");
  *out = nullptr;
  PP_REQUIRE(d->num_poses > 0 && d->num_points > 0 && d->num_cameras > 0 && d->num_obs > 0,
             "pp_ba_create: empty problem (poses %d, points %d, cameras %d, obs %lld)", d->num_poses, d->num_points,
             d->num_cameras, (long long)d->num_obs);
  PP_REQUIRE(d->lines && d->obs_pose && d->obs_point && d->pose_camera && d->camera_model, "pp_ba_create: null array");
  PP_REQUIRE(d->loss_type >= 0 && d->loss_type <= 2 && d->loss_scale >= 0, "pp_ba_create: bad loss");
  PP_REQUIRE(d->num_obs < (int64_t)1 << 31, "pp_ba_create: more than 2^31 observations");
  PP_REQUIRE(d->ordering >= PP_ORDERING_DEFAULT && d->ordering <= PP_ORDERING_AUTO, "pp_ba_create: unknown ordering %d", d->ordering);      // (every check of the descriptor comes before the device is touched)
  const int C = d->num_poses, P = d->num_points, K = d->num_cameras;
  const int64_t M = d->num_obs;
  PP_REQUIRE(K < (1 << 26), "pp_ba_create: too many intrinsics blocks");
  for (int k = 0; k < K; ++k) PP_REQUIRE(CameraNumParams(d->camera_model[k]) > 0, "pp_ba_create: unknown camera model %d", d->camera_model[k]);
  for (int c = 0; c < C; ++c) PP_REQUIRE(d->pose_camera[c] >= 0 && d->pose_camera[c] < K, "pp_ba_create: pose_camera[%d] out of range", c);
  for (int64_t o = 0; o < M; ++o) {
    PP_REQUIRE(d->obs_pose[o] >= 0 && d->obs_pose[o] < C && d->obs_point[o] >= 0 && d->obs_point[o] < P,
               "pp_ba_create: observation %lld indexes out of range", (long long)o);
    const double nrm = std::sqrt(d->lines[3 * o] * d->lines[3 * o] + d->lines[3 * o + 1] * d->lines[3 * o + 1]);
    // CHECK_NEAR(norm, 1.0, 1e-6) of the reference (cost_functions.h:51-52, bundle_adjustment.cc:373)
    PP_REQUIRE(std::fabs(nrm - 1.0) <= 1e-6, "pp_ba_create: line %lld is not normalised (|(a,b)| = %.9g)", (long long)o, nrm);
  }
  // variable intrinsics: compact columns, block k at intr_off[k] (oracle/bundle_adjustment.h BuildLayout; reference
  // bundle_adjustment.cc:490-528: constant camera unless a refine flag is set, SubsetParameterization otherwise)
  std::vector<int32_t> intr_off(K, -1), intr_nv(K, 0), intr_col((size_t)K * kCamStride, -1);
  int nv_widest = 0;      // the most variable parameters any camera has (the row width of the solver's compact camera Jacobians)
  int NI = 0;
  if (d->camera_const_mask) {
    // a block is part of the problem if an image references it (the same on every rank of a point-sharded group,
    // whose shards hold different observations)
    std::vector<char> cam_used(K, 0);
    for (int c = 0; c < C; ++c) cam_used[d->pose_camera[c]] = 1;
    for (int k = 0; k < K; ++k) {
      if (!cam_used[k]) continue;
      const int np = CameraNumParams(d->camera_model[k]);
      int nv = 0;
      for (int j = 0; j < np; ++j) if (!((d->camera_const_mask[k] >> j) & 1)) intr_col[(size_t)k * kCamStride + j] = nv++;
      if (nv > 0) { intr_off[k] = NI; intr_nv[k] = nv; NI += nv; } nv_widest = std::max(nv_widest, nv);
    }
  }
  int ndev = 0;
  PP_HIP_TRY(hipGetDeviceCount(&ndev));
  PP_REQUIRE(device >= 0 && device < ndev, "pp_ba_create: device %d of %d", device, ndev);
  PP_HIP_TRY(hipSetDevice(device));

  // ---- camera ordering of the reduced system (what Ceres' SPARSE_SCHUR does before it factorises, bundle_adjustment.cc:279-282) -------
  // The images are renumbered INTERNALLY (pose index = position of its six columns in the reduced system) when that makes the tile
  // structure of the factor sparser; every per-image input / output of the C ABI (pp_ba_set/get_parameters, pp_ba_reduced_system) is
  // in the caller's order.  old_of_new empty = the caller's order.
  const auto t_create0 = std::chrono::steady_clock::now();
  const bool create_dbg = std::getenv("PPSFM_CREATE_DEBUG") != nullptr;      // (stderr: where the host time of this create goes)
  auto lap = [&, last = t_create0](const char* what) mutable {
    if (!create_dbg) return;
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "ppsfm: create %-34s %.3f ms\n", what, std::chrono::duration<double, std::milli>(now - last).count());
    last = now;
  };
  pp_ba_impl* h = new pp_ba_impl();
  OnUnwind unwind{[&] { pp_ba_destroy(h); }};      // (a std::bad_alloc of the host builders below must not leak the handle's device memory)
  // (a handle whose order and tile structure come from the caller's co-visibility - the union over the shards of a point-sharded group - lays out the
  // exchanged system like every other rank that was given the same matrix: it may join a group renumbered and block-sparse)
  h->structure_from_covisibility = d->covisibility != nullptr;
  h->device = device; h->C = C; h->P = P; h->K = K; h->M = M;
  h->loss_type = d->loss_type; h->loss_scale = d->loss_scale;
  h->NI = NI; h->n_red = 6 * C + NI; h->intrinsics_variable = NI > 0;
  h->jcam_stride = std::max(2, (nv_widest + 1) & ~1);
  {
    // linear solver of the reduced camera system, chosen before the structure is built as BundleAdjuster::Solve does
    // (bundle_adjustment.cc:273-286): ITERATIVE_SCHUR above 1000 images.  PPSFM_BA_LINEAR_SOLVER=direct|iterative overrides (tools / tests).
    // Variable intrinsics ride along: their columns follow the pose columns in the conjugate-gradient vectors, their part of the operator is applied
    // from the per-observation intrinsics Jacobians, their diagonal blocks (the preconditioner's) are assembled from the (k, k) pair lists alone.
    int ls = d->linear_solver;
    if (const char* e = std::getenv("PPSFM_BA_LINEAR_SOLVER")) ls = (e[0] == 'i' || e[0] == 'I') ? PP_LINEAR_SOLVER_ITERATIVE_SCHUR : ((e[0] == 'd' || e[0] == 'D') ? PP_LINEAR_SOLVER_DIRECT : ls);
    h->iterative = ls == PP_LINEAR_SOLVER_ITERATIVE_SCHUR || (ls == PP_LINEAR_SOLVER_AUTO && C > PP_MAX_NUM_IMAGES_DIRECT_SOLVER);
  }
  const bool iterative = h->iterative;
  int rc = PP_OK;
#define TRY(x) do { rc = (x); if (rc) { pp_ba_destroy(h); return rc; } } while (0)
#define TRYH(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { SetLastError("%s: %s", #x, hipGetErrorString(e_)); pp_ba_destroy(h); return PP_ERR_HIP; } } while (0)
  TRY(PoolStreamAcquire(&h->stream));
  TRY(PoolEventAcquire(&h->ev0, true));
  TRY(PoolEventAcquire(&h->ev1, true));
  hipStream_t s = h->stream;
  lap("handle, stream, events");

  // ---- the by-point lists (no image order in them) ---------------------------------------------------------------------------------------------------
  std::vector<uint8_t> point_const(P, 0);
  if (d->point_const) std::memcpy(point_const.data(), d->point_const, P);
  std::vector<int32_t> pt_start(P + 1, 0), pt_obs(M);      // CSR by point (counting sort keeps observation order inside a group)
  for (int64_t o = 0; o < M; ++o) pt_start[d->obs_point[o] + 1]++;
  for (int p = 0; p < P; ++p) pt_start[p + 1] += pt_start[p];
  {
    std::vector<int32_t> fp(pt_start.begin(), pt_start.end() - 1);
    for (int64_t o = 0; o < M; ++o) pt_obs[fp[d->obs_point[o]]++] = (int32_t)o;
  }
  lap("CSR by point");
  // On the device when the problem is large enough to pay for the launches (pair_lists.hip): the by-point lists go up first - the co-visibility graph the
  // image order is chosen on comes from them (in the caller's numbering), then the Schur pair lists (in the order chosen).
  bool lists_on_device = !iterative && PairListsOnDeviceEligible(C, M);
  std::vector<uint64_t> graph_bits;
  double graph_ms = 0;
  if (lists_on_device) {
    TRY(HandleAlloc(&h->obs_pose, M)); TRY(HandleAlloc(&h->obs_point, M)); TRY(HandleAlloc(&h->pose_const, C)); TRY(HandleAlloc(&h->point_const, P));
    TRY(HandleAlloc(&h->pt_start, P + 1)); TRY(HandleAlloc(&h->pt_obs, M));
    TRY(Upload(h->obs_point, d->obs_point, M, s)); TRY(Upload(h->point_const, point_const.data(), P, s));
    TRY(Upload(h->pt_start, pt_start.data(), P + 1, s)); TRY(Upload(h->pt_obs, pt_obs.data(), M, s));
    if (ppsfm::OrderingReadsObservations(d, NI)) {
      const auto tg = std::chrono::steady_clock::now();
      std::vector<uint8_t> fixed(C, 0);
      if (d->pose_const && (iterative || ppsfm::PrivateIntrinsicsColumns(d) == 0)) std::memcpy(fixed.data(), d->pose_const, C);      // (as ChooseImageOrdering's fixed_image)
      TRY(Upload(h->obs_pose, d->obs_pose, M, s)); TRY(Upload(h->pose_const, fixed.data(), C, s));
      TRY(CoVisibilityOnDevice(C, M, h->pt_start, h->pt_obs, h->obs_pose, h->obs_point, h->pose_const, h->point_const, s, &graph_bits));
      graph_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tg).count();
      lap("co-visibility graph (device)");
    }
  }

  // ---- camera ordering of the reduced system (what Ceres' SPARSE_SCHUR does before it factorises, bundle_adjustment.cc:279-282) -------
  // The images are renumbered INTERNALLY (pose index = position of its six columns in the reduced system) when that makes the tile
  // structure of the factor sparser; every per-image input / output of the C ABI (pp_ba_set/get_parameters, pp_ba_reduced_system) is
  // in the caller's order.  old_of_new empty = the caller's order.
  ppsfm::ImageOrdering ord = ppsfm::ChooseImageOrdering(d, NI, graph_bits.empty() ? nullptr : graph_bits.data());
  const double ordering_ms = graph_ms + ord.plan_ms;
  std::vector<int32_t> old_of_new, new_of_old;
  old_of_new.swap(ord.old_of_new); new_of_old.swap(ord.new_of_old);
  const int nnz_natural = ord.nnz_natural, nnz_ordered = ord.nnz_ordered;
  const bool reordered = !old_of_new.empty();
  lap("image order");
  // the problem in internal image order (views of the caller's arrays when nothing moved)
  std::vector<int32_t> obs_pose_perm, pose_camera_perm;
  std::vector<uint8_t> pose_const_perm, tvec_mask_perm;
  if (reordered) {
    obs_pose_perm.resize(M); pose_camera_perm.resize(C);
    for (int64_t o = 0; o < M; ++o) obs_pose_perm[o] = new_of_old[d->obs_pose[o]];
    for (int c = 0; c < C; ++c) pose_camera_perm[new_of_old[c]] = d->pose_camera[c];
    if (d->pose_const) { pose_const_perm.resize(C); for (int c = 0; c < C; ++c) pose_const_perm[new_of_old[c]] = d->pose_const[c]; }
    if (d->tvec_const_mask) { tvec_mask_perm.resize(C); for (int c = 0; c < C; ++c) tvec_mask_perm[new_of_old[c]] = d->tvec_const_mask[c]; }
  }
  const int32_t* in_obs_pose = reordered ? obs_pose_perm.data() : d->obs_pose;
  const int32_t* in_pose_camera = reordered ? pose_camera_perm.data() : d->pose_camera;
  const uint8_t* in_pose_const = reordered ? (d->pose_const ? pose_const_perm.data() : nullptr) : d->pose_const;
  const uint8_t* in_tvec_mask = reordered ? (d->tvec_const_mask ? tvec_mask_perm.data() : nullptr) : d->tvec_const_mask;
  h->pose_old_of_new = old_of_new; h->pose_new_of_old = new_of_old;
  h->nnz_tiles_natural = nnz_natural; h->nnz_tiles_ordered = nnz_ordered;
  // columns of the reduced system: the vectors' order (pose c at 6c, intrinsics block k at 6C + intr_off[k]) unless every image carries its own variable
  // intrinsics, which then sit beside its pose columns (image_ordering.hip PrivateIntrinsicsColumns; internal image order)
  const int nv_private = iterative ? 0 : ppsfm::PrivateIntrinsicsColumns(d);
  const int W6 = 6 + nv_private;
  h->spos_identity = nv_private == 0;
  {
    const char* e = std::getenv("PPSFM_BA_INTR_WIDE");      // 0: the general block-pair lists (ba_intr.hip) also for per-image intrinsics (tests / comparisons)
    h->intr_wide_nv = (nv_private >= 2 && nv_private <= 8 && !(e && std::atoi(e) == 0)) ? nv_private : 0;
  }
  h->spos_host.resize((size_t)h->n_red);
  for (int v = 0; v < h->n_red; ++v) h->spos_host[v] = v;
  if (nv_private)
    for (int i = 0; i < C; ++i) {
      const int k = in_pose_camera[i];
      for (int j = 0; j < 6; ++j) h->spos_host[6 * i + j] = W6 * i + j;
      for (int j = 0; j < nv_private; ++j) h->spos_host[6 * C + intr_off[k] + j] = W6 * i + 6 + j;
    }

  // ---- host-side structure building ------------------------------------------------------
  std::vector<double> la(M), lb(M), lc(M);
  for (int64_t o = 0; o < M; ++o) { la[o] = d->lines[3 * o]; lb[o] = d->lines[3 * o + 1]; lc[o] = d->lines[3 * o + 2]; }
  std::vector<int32_t> obs_cam(M);
  for (int64_t o = 0; o < M; ++o) { const int k = in_pose_camera[in_obs_pose[o]]; obs_cam[o] = (k << 4) | d->camera_model[k]; }
  std::vector<uint8_t> pose_const(C, 0), tvec_mask(C, 0);
  if (in_pose_const) std::memcpy(pose_const.data(), in_pose_const, C);
  if (in_tvec_mask) std::memcpy(tvec_mask.data(), in_tvec_mask, C);
  h->host_pose_const = pose_const;
  // which images have columns in the reduced system at all: those with a variable pose - and every image when each carries variable intrinsics of its own
  // beside its pose columns (its block pairs with the images it shares points with exist whatever its pose is; the pose rows of a constant pose are zeros)
  const std::vector<uint8_t> list_const = nv_private > 0 ? std::vector<uint8_t>(C, 0) : pose_const;
  std::vector<int32_t> pose_start(C + 1, 0), pose_obs(M);      // CSR by image
  for (int64_t o = 0; o < M; ++o) pose_start[in_obs_pose[o] + 1]++;
  for (int c = 0; c < C; ++c) pose_start[c + 1] += pose_start[c];
  {
    std::vector<int32_t> fc(pose_start.begin(), pose_start.end() - 1);
    for (int64_t o = 0; o < M; ++o) pose_obs[fc[in_obs_pose[o]]++] = (int32_t)o;
  }
  lap("line streams, CSR by image");
  // block-pair entry lists of the reduced camera matrix (lower triangle, variable poses/points only): for every pair of variable images (ci >= cj) that
  // share a variable point, the (observation of ci, observation of cj) pairs, lists in (ci, cj) order, a list's entries in (oi, oj) order.
  // Built per problem structure, i.e. once per BA call of an incremental mapper (src/sfm/incremental_mapper.cc:893-936): ROW BY ROW (round 5; rounds 1-4
  // walked the points twice through a C x C table of counters and sorted 16-byte entries - 7.6 ms of an 11 ms create at 500 images / 200k observations) -
  // image ci's observations in order, each with the other observers of its point: the row's counters are C ints (cache resident), the rows are independent
  // (a few host threads share them), and the entries come out in list order without a sort.
  std::vector<int32_t> pair_start, pair_ij, pair_entries;
  int64_t total_entries = 0;
  {
    // the entry count grows with the SQUARE of the track lengths (a track of L variable observers gives L (L - 1) / 2 entries, up to L (L - 1) when images
    // repeat) while every offset into the lists is 32-bit: count in 64 bits first and refuse what does not fit
    int64_t bound = 0;
    for (int p = 0; p < P && !iterative; ++p) {
      if (point_const[p]) continue;
      int64_t nv = 0;
      for (int e = pt_start[p]; e < pt_start[p + 1]; ++e) nv += list_const[in_obs_pose[pt_obs[e]]] ? 0 : 1;
      bound += nv * (nv - 1);               // (a track that sees ONE image nv times lists both orders of every pair)
    }
    if (bound >= ((int64_t)1 << 31) - 1) {
      SetLastError("pp_ba_create: %lld Schur pair entries (sum over points of track^2 / 2) exceed the 32-bit pair lists", (long long)bound);
      pp_ba_destroy(h);
      return PP_ERR_INVALID;
    }
  }
  // The pair lists on the device (the by-point lists are there): the lists' 3 ints per list come back, the entries never leave the device.  A structure with a list too long for the device's per-list sort takes the host builder below.
  int32_t* dev_entries = nullptr;
  const bool arrays_on_device = lists_on_device;      // (the by-point lists, obs_pose / obs_point and the constant flags are on the device already - also when the host builder takes over below)
  if (lists_on_device) {
    TRY(Upload(h->obs_pose, in_obs_pose, M, s)); TRY(Upload(h->pose_const, list_const.data(), C, s));      // (the order chosen)
    bool fallback = false;
    rc = BuildPairListsOnDevice(C, M, h->pt_start, h->pt_obs, h->obs_pose, h->obs_point, h->pose_const, h->point_const, s, &dev_entries, &total_entries, &pair_start, &pair_ij, &fallback);
    if (rc && !fallback) { pp_ba_destroy(h); return rc; }
    if (fallback) { lists_on_device = false; rc = PP_OK; total_entries = 0; pair_start.clear(); pair_ij.clear(); }
    else h->pair_entries = dev_entries;
    if (nv_private > 0) TRY(Upload(h->pose_const, pose_const.data(), C, s));      // (the handle's array says which POSES are constant)
  }
  if (!iterative && !lists_on_device) {      // (an iterative handle applies S from the records: no pair lists)
    // (pair_lists.hip BuildPairListsOnHost: buckets per row image + a counting sort per row, on a few host threads)
    BuildPairListsOnHost(C, P, M, pt_start.data(), pt_obs.data(), in_obs_pose, list_const.data(), point_const.data(), 0, [&](const char* what) { lap(what); },
                         &total_entries, &pair_start, &pair_ij, &pair_entries);
  } else if (iterative) {
    pair_start.assign(1, 0);
  }
  lap("pair lists");
  h->num_pairs = (int64_t)pair_start.size() - 1; h->num_entries = total_entries;
  const auto t_create2 = std::chrono::steady_clock::now();
  {
    // Tile structure of the reduced camera system (64x64 tiles of its lower triangle): which tiles the co-visibility puts an
    // entry in, closed under the fill-in of the factorisation.  When a good part of them stays empty (a sequence: images only
    // share points with their neighbours) the assembly, the factorisation and the back substitution skip them - what the
    // reference gets from Ceres' SPARSE_SCHUR above 50 images (src/optim/bundle_adjustment.cc:275-286).  PPSFM_BA_SPARSE=0 disables.
    const int Nn = ((h->n_red + 1 + 63) / 64) * 64, Tt = Nn / 64;
    std::vector<uint8_t> nz((size_t)Tt * Tt, 0);
    int64_t marked = 0;
    const int64_t image_rows = (W6 * C - 1) / 64 + 1, all_tiles = image_rows * (image_rows + 1) / 2;      // the tiles the images' columns can reach
    auto mark = [&](int r0, int r1, int c0, int c1) {
      for (int ti = r0 / 64; ti <= r1 / 64; ++ti)
        for (int tj = c0 / 64; tj <= c1 / 64; ++tj) if (tj <= ti && !nz[(size_t)ti * Tt + tj]) { nz[(size_t)ti * Tt + tj] = 1; ++marked; }
    };
    // (W6 columns per image: its pose and, when every image carries its own variable intrinsics, those beside it - coupled with the same images as the pose)
    for (int c = 0; c < C; ++c) mark(W6 * c, W6 * c + W6 - 1, W6 * c, W6 * c + W6 - 1);
    // (a dense co-visibility has every tile after a fraction of its 125 000 pairs: the walk stops there)
    for (size_t i = 0; i + 1 < pair_ij.size() && marked < all_tiles; i += 2) mark(W6 * pair_ij[i], W6 * pair_ij[i] + W6 - 1, W6 * pair_ij[i + 1], W6 * pair_ij[i + 1] + W6 - 1);
    if (d->covisibility) {
      // a pair of THIS shard that the given matrix lacks: the matrix is not the group's union (stale, partial, another scene's) and the other ranks - who
      // only have the matrix - would lay out another tile map than this one: refuse here instead of exchanging differently sized systems later
      for (size_t i = 0; i + 1 < pair_ij.size(); i += 2) {
        const int oi = reordered ? old_of_new[pair_ij[i]] : pair_ij[i], oj = reordered ? old_of_new[pair_ij[i + 1]] : pair_ij[i + 1];
        if (oi != oj && !d->covisibility[(size_t)oi * C + oj] && !d->covisibility[(size_t)oj * C + oi]) {
          SetLastError("pp_ba_create: images %d and %d share a point of this shard but pp_ba_problem_desc::covisibility has no entry for them - the matrix must be "
                       "the union over the group's shards (pp_ba_covisibility of every rank, element-wise MAX)", oi, oj);
          pp_ba_destroy(h);
          return PP_ERR_INVALID;
        }
      }
    }
    if (d->covisibility)      // (the union over a group's shards: tiles other ranks' points fill, in the internal order)
      for (int i = 1; i < C; ++i) {
        if (nv_private == 0 && d->pose_const && d->pose_const[i]) continue;
        const int ni = reordered ? new_of_old[i] : i;
        for (int j = 0; j < i; ++j)
          if ((d->covisibility[(size_t)i * C + j] || d->covisibility[(size_t)j * C + i]) && !(nv_private == 0 && d->pose_const && d->pose_const[j])) {
            const int nj = reordered ? new_of_old[j] : j, hi = std::max(ni, nj), lo = std::min(ni, nj);
            mark(W6 * hi, W6 * hi + W6 - 1, W6 * lo, W6 * lo + W6 - 1);
          }
      }
    if (NI > nv_private * C) mark(W6 * C, h->n_red - 1, 0, h->n_red - 1);      // the shared intrinsics rows couple with every image
    mark(h->n_red, h->n_red, 0, h->n_red);                        // the right-hand side's row
    const int nnz = SymbolicTileFill(Tt, nz.data());
    const char* e = std::getenv("PPSFM_BA_SPARSE");
    h->sparse_tiles = !iterative && !(e && std::atoi(e) == 0) && Tt >= 8 && (int64_t)nnz * 10 <= (int64_t)Tt * (Tt + 1) / 2 * 7;      // (variable intrinsics: their rows are dense, the pose part keeps its structure - an arrow)
    h->tile_nz.swap(nz);
    h->num_nz_tiles = nnz;
  }
  {
    // the factorisation overwrites S with L, fill-in included, so a block of two variable poses that share no point must be
    // cleared before every assembly: give it an EMPTY list (k_schur_pairs then stores zeros).  With every such block listed
    // and no same-image pair (which accumulates into a diagonal block), k_schur_pairs stores instead of read-modify-write
    // and S needs no per-iteration clear.
    lap("tile map");
    bool same = false;
    for (size_t i = 0; i + 1 < pair_ij.size(); i += 2) same = same || pair_ij[i] == pair_ij[i + 1];
    h->pairs_complete = !same && !h->sparse_tiles;      // (block-sparse: no empty lists; the non-zero tiles are cleared per assembly instead)
    if (h->pairs_complete && h->num_pairs > 0) {
      std::vector<int32_t> var;      // the variable images, ascending
      for (int c = 0; c < C; ++c) if (!list_const[c]) var.push_back(c);
      const size_t V = var.size(), npairs = V * (V - 1) / 2;
      std::vector<int32_t> start2(npairs + 1), ij2(2 * npairs);
      size_t src = 0, at = 0;
      const size_t np0 = (size_t)h->num_pairs;
      const int32_t* pij = pair_ij.data();
      for (size_t a = 1; a < V; ++a) {
        const int ci = var[a];
        for (size_t b = 0; b < a; ++b, ++at) {
          const int cj = var[b];
          const bool hit = src < np0 && pij[2 * src] == ci && pij[2 * src + 1] == cj;
          start2[at] = src < np0 ? pair_start[src] : (int32_t)total_entries;      // (an empty list starts where the next non-empty one does)
          src += hit ? 1 : 0;
          ij2[2 * at] = ci; ij2[2 * at + 1] = cj;
        }
      }
      start2[npairs] = (int32_t)total_entries;
      // an empty list starts where the next non-empty one does, so consecutive differences are still the lengths
      pair_start.swap(start2); pair_ij.swap(ij2);
      h->num_pairs = (int64_t)pair_start.size() - 1;
    }
  }
  lap("empty lists of a complete system");
  {
    // k_schur_pairs walks ten lists per wavefront in lock step: order the pairs by list length (longest first) so
    // that the lists sharing a wavefront have equal lengths; pair_start becomes (first, last+1) per pair
    const size_t np = (size_t)h->num_pairs;
    std::vector<int32_t> order(np);
    {   // stable counting sort by list length, longest first
      int32_t max_len = 0;
      for (size_t i = 0; i < np; ++i) max_len = std::max(max_len, pair_start[i + 1] - pair_start[i]);
      std::vector<int64_t> pos((size_t)max_len + 2, 0);
      for (size_t i = 0; i < np; ++i) ++pos[(size_t)(max_len - (pair_start[i + 1] - pair_start[i])) + 1];
      for (size_t l = 0; l + 1 < pos.size(); ++l) pos[l + 1] += pos[l];
      for (size_t i = 0; i < np; ++i) order[(size_t)pos[(size_t)(max_len - (pair_start[i + 1] - pair_start[i]))]++] = (int32_t)i;
    }
    // L2 locality: pairs grouped by STRIPS of 8 column images (all rows), strip t handled by the workgroups that land on XCD
    // t % 8 (workgroups are dealt round-robin by blockIdx; 40 pairs per workgroup): the records of the strip's 8 images
    // (0.6 MB) stay in that XCD's 4 MB L2 while the row side streams through once.  Measured on cfg 3 (Schur phase = the two
    // prepare kernels + the gather, us): 16x16-image tiles in row-major order 97.2, in column-major order 93.8, strips of 8
    // or 4 images 89.1, of 16 images 94.7, of 32 images 98.3.  With the tiles the gather's L2 hit rate was 56 % of 7.8 M
    // requests and 4 M 64-byte requests went to the fabric (rocprofv3 TCC_HIT/MISS, TCC_EA0_RDREQ/WRREQ).
    if (np > 0) {
      std::vector<std::vector<int32_t>> bucket(8);
      {
        // order is by length (desc); a stable counting sort by strip keeps that inside a strip
        const int ts = 3;
        auto strip_of = [&](int32_t id) { return (size_t)(pair_ij[2 * id + 1] >> ts); };
        std::vector<int64_t> pos(((size_t)C >> ts) + 2, 0);
        size_t per_bucket[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int32_t id : order) { const size_t t = strip_of(id); ++pos[t + 1]; ++per_bucket[t & 7]; }
        for (size_t t = 0; t + 1 < pos.size(); ++t) pos[t + 1] += pos[t];
        std::vector<int32_t> by_strip(np);
        for (int32_t id : order) by_strip[(size_t)pos[strip_of(id)]++] = id;
        for (int x = 0; x < 8; ++x) bucket[(size_t)x].reserve(per_bucket[x]);
        for (int32_t id : by_strip) bucket[strip_of(id) & 7].push_back(id);
      }
      std::vector<size_t> at(8, 0);
      size_t out = 0;
      const int first_xcd = C & 7;      // k_schur_blocks: the pair workgroups follow C per-image workgroups
      for (size_t wg = 0; out < np; ++wg) {
        int x = (int)((first_xcd + wg) & 7);
        for (int tries = 0; tries < 8 && at[(size_t)x] >= bucket[(size_t)x].size(); ++tries) x = (x + 1) & 7;
        for (int k2 = 0; k2 < 40 && at[(size_t)x] < bucket[(size_t)x].size(); ++k2) order[out++] = bucket[(size_t)x][at[(size_t)x]++];
      }
    }
    std::vector<int32_t> range(2 * np), ij(2 * np);
    for (size_t i = 0; i < np; ++i) {
      range[2 * i] = pair_start[order[i]]; range[2 * i + 1] = pair_start[order[i] + 1];
      ij[2 * i] = pair_ij[2 * order[i]]; ij[2 * i + 1] = pair_ij[2 * order[i] + 1];
    }
    if (np == 0) range.assign(2, 0);
    pair_start.swap(range); pair_ij.swap(ij);
  }

  lap("lists by length and strip");
  // ---- the pair lists in chunks (long lists) ----------------------------------------------
  std::vector<int32_t> small_chunk, small_pair_chunk;
  // A pair list is walked entry by entry with a dependent gather each (~0.7 us): lists of more than 64 entries are always cut into chunks of 16
  // (deterministic partial blocks + one reduction); a problem too small to fill the chip (the mapper's local bundle adjustment: 20 images /
  // 2000 observations walk 40-entry lists for 26 us with 3 % of the lanes) cuts lists of more than 12 entries into chunks of 8.
  int32_t chunk_len = 16;      // (32 until the sequence scenes were measured: cfg-3 size, window 40 - lists of ~35 entries - Schur phase 105 us with 32, 95 with 16, 93 with 8, 98 with 4: tools/chunk_len_probe.sh)
  {
    int32_t longest = 0;
    int64_t total = 0;
    for (size_t i = 0; i < (size_t)h->num_pairs; ++i) { const int32_t len = pair_start[2 * i + 1] - pair_start[2 * i]; longest = std::max(longest, len); total += len; }
    const bool latency_bound = total <= 65536 && longest > 12;
    h->pairs_chunked = !iterative && NI == 0 && (longest > 64 || latency_bound) && !(std::getenv("PPSFM_BA_CHUNKED_PAIRS") && std::atoi(std::getenv("PPSFM_BA_CHUNKED_PAIRS")) == 0);
    if (latency_bound) chunk_len = 8;
    if (const char* e = std::getenv("PPSFM_BA_CHUNK_LEN")) chunk_len = std::max(1, std::atoi(e));      // (experiments: tools/nd_probe.py)
  }
  const bool want_chunks = h->pairs_chunked;
  if (want_chunks) {
    const size_t np = (size_t)h->num_pairs;
    small_pair_chunk.assign(np + 1, 0);
    {
      size_t count = 0;
      for (size_t i = 0; i < np; ++i) count += (size_t)((pair_start[2 * i + 1] - pair_start[2 * i] + chunk_len - 1) / chunk_len);
      small_chunk.reserve(3 * count);
    }
    for (size_t i = 0; i < np; ++i) {
      small_pair_chunk[i] = (int32_t)(small_chunk.size() / 3);
      for (int32_t e = pair_start[2 * i]; e < pair_start[2 * i + 1]; e += chunk_len) {
        small_chunk.push_back((int32_t)i); small_chunk.push_back(e); small_chunk.push_back(std::min(e + chunk_len, pair_start[2 * i + 1]));
      }
    }
    small_pair_chunk[np] = (int32_t)(small_chunk.size() / 3);
    h->small_num_chunks = (int)(small_chunk.size() / 3);
    // L2 locality of the chunk kernel.  k_schur_self_chunks gives a workgroup 40 CHUNKS, so the strip order above (made for 40 PAIRS per workgroup) no longer
    // lines a strip up with an XCD: at banded cfg 3 (lists of ~35 entries, three chunks each) a strip's pairs landed on three XCDs and every XCD read most
    // records - FETCH_SIZE 172 MB per launch against 38 MB of records, L2 hit rate 45 % (profiles/r06_band_pmc.json).  A sequence's pairs lie in a band:
    // the chunks are PROCESSED in the order of their pair's column image, cut into eight equal runs, run x on the workgroups that land on XCD x (dealt 40
    // chunks at a time, as the dispatcher deals workgroups) - an XCD then works through one contiguous range of column images with their partners (the next
    // window of row images) and a record is read by at most two XCDs.  A chunk keeps its id (entry 0 of its triple): its partial block is written where
    // k_schur_chunk_reduce expects it, so the sums and their bits are unchanged.  Small problems keep the natural order (nothing to gain below a few MB).
    const size_t nch = small_chunk.size() / 3;
    const bool xcd_order = nch >= 8 * 40 * 4 && !(std::getenv("PPSFM_BA_CHUNK_XCD") && std::atoi(std::getenv("PPSFM_BA_CHUNK_XCD")) == 0);
    for (size_t q = 0; q < nch; ++q) small_chunk[3 * q] = (int32_t)q;      // (entry 0: the chunk's id = where its partial block goes)
    if (xcd_order) {
      std::vector<int32_t> key_of[2] = {std::vector<int32_t>(nch), std::vector<int32_t>(nch)};      // [0] row image (minor key), [1] column image (major key) of a chunk's pair
      for (size_t i = 0; i < np; ++i)
        for (int32_t q = small_pair_chunk[i]; q < small_pair_chunk[i + 1]; ++q) { key_of[0][(size_t)q] = pair_ij[2 * i]; key_of[1][(size_t)q] = pair_ij[2 * i + 1]; }
      // by column image, then by row image, a pair's chunks in order: two stable counting sorts, the minor key first (a comparison sort with this
      // indirect key cost 1.5 ms of a 5.5 ms create at banded cfg 3 - a third more `structure` time than the whole round-5 create spent there)
      std::vector<int32_t> by_col(nch), tmp(nch);
      {
        std::vector<int32_t> cnt((size_t)C + 1);
        for (int pass = 0; pass < 2; ++pass) {
          std::fill(cnt.begin(), cnt.end(), 0);
          const int32_t* keys = key_of[pass].data();
          auto key = [&](int32_t q) { return (size_t)keys[(size_t)q]; };
          if (pass == 0) { for (size_t q = 0; q < nch; ++q) ++cnt[key((int32_t)q) + 1]; }
          else { for (size_t q = 0; q < nch; ++q) ++cnt[key(tmp[q]) + 1]; }
          for (int c = 0; c < C; ++c) cnt[(size_t)c + 1] += cnt[(size_t)c];
          if (pass == 0) { for (size_t q = 0; q < nch; ++q) tmp[(size_t)cnt[key((int32_t)q)]++] = (int32_t)q; }
          else { for (size_t q = 0; q < nch; ++q) by_col[(size_t)cnt[key(tmp[q])]++] = tmp[q]; }
        }
      }
      std::vector<int32_t> out(3 * nch);
      size_t w = 0;
      const size_t per = (nch + 7) / 8;
      size_t at[8];
      for (int x = 0; x < 8; ++x) at[x] = std::min(nch, per * (size_t)x);
      const int first_xcd = C & 7;      // (the chunk workgroups follow C per-image workgroups)
      size_t done = 0;
      for (size_t wg = 0; done < nch; ++wg) {
        int x = (int)((first_xcd + wg) & 7);
        for (int tries = 0; tries < 8 && at[x] >= std::min(nch, per * (size_t)(x + 1)); ++tries) x = (x + 1) & 7;
        for (int k2 = 0; k2 < 40 && at[x] < std::min(nch, per * (size_t)(x + 1)); ++k2, ++done) {
          const size_t q = (size_t)by_col[at[x]++];
          out[w++] = small_chunk[3 * q]; out[w++] = small_chunk[3 * q + 1]; out[w++] = small_chunk[3 * q + 2];
        }
      }
      small_chunk.swap(out);
    }
  }

  // ---- variable intrinsics: CSR by intrinsics block, generic block-pair lists with chunks --------------------
  struct DiagLists { std::vector<int32_t> entries, pair, pair_chunk, chunk, multi; int64_t num_groups = 0; };
  DiagLists kk;      // (direct handles with variable intrinsics: the diagonal blocks' lists)
  std::vector<int32_t> cam_start(K + 1, 0), cam_obs, gen_pair, gen_pair_chunk, gen_chunk, gen_entries, gen_multi, gen_grp_start, gen_grp_obs, isum_chunk, isum_cam_chunk;
  if (NI > 0) {
    cam_obs.resize(M);
    for (int64_t o = 0; o < M; ++o) cam_start[in_pose_camera[in_obs_pose[o]] + 1]++;
    for (int k = 0; k < K; ++k) cam_start[k + 1] += cam_start[k];
    { std::vector<int32_t> f(cam_start.begin(), cam_start.end() - 1); for (int64_t o = 0; o < M; ++o) cam_obs[f[in_pose_camera[in_obs_pose[o]]]++] = (int32_t)o; }
    isum_cam_chunk.push_back(0);
    for (int k = 0; k < K; ++k) {
      if (intr_off[k] >= 0)
        for (int e = cam_start[k]; e < cam_start[k + 1]; e += kIsumChunk) { isum_chunk.push_back(k); isum_chunk.push_back(e); isum_chunk.push_back(std::min(e + kIsumChunk, cam_start[k + 1])); }
      isum_cam_chunk.push_back((int32_t)(isum_chunk.size() / 3));
    }
    // The DIAGONAL blocks S_kk of the intrinsics (an iterative handle assembles nothing else - they are its preconditioner, everything else is applied
    // from the records; a direct handle takes them out of the entry lists below).  They factor:
    auto build_diag = [&](DiagLists& dl) {
      //   S_kk = sum_o J_k,o^T J_k,o - sum_{(p, k)} L R,  L = sum_{o in (p,k)} J_k,o^T T_o,  R = sum_{o in (p,k)} X_o^T J_k,o
      // over the GROUPS (p, k) = the observations of point p taken with camera k - linear in the observations where the pair list of a shared
      // camera is quadratic in the track lengths (k_intr_kk).  Groups sorted by camera, chunks of ~320 observations (a workgroup each), one pair per variable camera.
      struct Grp { int32_t k, p, e0, e1; };
      std::vector<Grp> grps;
      std::vector<std::pair<int32_t, int32_t>> ko;      // (camera, observation) of one track
      std::vector<int32_t> flat;                          // observations, group after group (in point order first)
      for (int p = 0; p < P; ++p) {
        ko.clear();
        for (int e = pt_start[p]; e < pt_start[p + 1]; ++e) {
          const int32_t o = pt_obs[e]; const int k = in_pose_camera[in_obs_pose[o]];
          if (intr_off[k] >= 0) ko.push_back({k, o});
        }
        std::stable_sort(ko.begin(), ko.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
        for (size_t a = 0; a < ko.size();) {
          size_t b = a;
          while (b < ko.size() && ko[b].first == ko[a].first) ++b;
          grps.push_back({ko[a].first, p, (int32_t)flat.size(), (int32_t)(flat.size() + (b - a))});
          for (size_t t = a; t < b; ++t) flat.push_back(ko[t].second);
          a = b;
        }
      }
      std::stable_sort(grps.begin(), grps.end(), [](const Grp& a, const Grp& b) { return a.k < b.k; });
      const size_t G = grps.size();
      dl.entries.assign(G + 1 + flat.size(), 0);      // [group starts | observations by group]
      { size_t pos = 0; for (size_t g = 0; g < G; ++g) { dl.entries[g] = (int32_t)pos; for (int32_t e = grps[g].e0; e < grps[g].e1; ++e) dl.entries[G + 1 + pos++] = flat[e]; } dl.entries[G] = (int32_t)pos; }
      dl.num_groups = (int64_t)G;
      dl.pair_chunk.push_back(0);
      size_t g = 0;
      for (int k = 0; k < K; ++k) {
        if (intr_off[k] < 0) continue;
        const int pair_id = (int)(dl.pair.size() / 4);
        dl.pair.push_back(6 * C + intr_off[k]); dl.pair.push_back(intr_nv[k]); dl.pair.push_back(6 * C + intr_off[k]); dl.pair.push_back(intr_nv[k] | (1 << 8));
        while (g < G && grps[g].k < k) ++g;
        size_t g0 = g; int64_t nobs = 0;
        for (; g < G && grps[g].k == k; ++g) {
          nobs += grps[g].e1 - grps[g].e0;
          if (nobs >= 320) { dl.chunk.push_back(pair_id); dl.chunk.push_back((int32_t)g0); dl.chunk.push_back((int32_t)(g + 1)); g0 = g + 1; nobs = 0; }
        }
        if (g0 < g) { dl.chunk.push_back(pair_id); dl.chunk.push_back((int32_t)g0); dl.chunk.push_back((int32_t)g); }
        dl.pair_chunk.push_back((int32_t)(dl.chunk.size() / 3));
      }
    };
    auto finish_diag = [&](DiagLists& dl) {
      const int64_t np = (int64_t)(dl.pair.size() / 4);
      for (int64_t pr = 0; pr < np; ++pr) if (dl.pair_chunk[pr + 1] - dl.pair_chunk[pr] != 1) dl.multi.push_back((int32_t)pr);
    };
    if (iterative) {
      DiagLists dl;
      build_diag(dl);
      gen_entries = dl.entries; gen_pair = dl.pair; gen_pair_chunk = dl.pair_chunk; gen_chunk = dl.chunk; h->gen_num_groups = dl.num_groups;
    } else if (h->intr_wide_nv > 0) {
      // every image carries its own intrinsics beside its pose columns: its 6 + n_v columns are ONE block, assembled by the pose blocks' own gather over the
      // pair lists with wider rows (k_schur_wide_self / k_schur_wide_pairs, ba_solver.hip) - no lists of their own
      gen_pair_chunk.push_back(0);
    } else {
    // FACTORED entries.  The intrinsics rows of S are  S_AB = sum_o J_A,o^T J_B,o - sum_{(oi, oj) sharing a point} J_A,oi^T T_oi X_oj^T J_B,oj  with A
    // an intrinsics block; the sum over oi does not depend on B or oj:  L_(p,A) = sum_{oi in (p,A)} J_A,oi^T T_oi  (n_v x 3, k_intr_L, per trial radius)
    // over the GROUP (p, A) = the observations of point p taken with camera A.  An entry is (group, oj [, oj belongs to the group: the direct term
    // rides on it]): sum_p (groups of p) x (observations of p) entries - linear in the track length for a camera shared by all images, where the
    // (oi, oj) lists were quadratic (500 images, tracks of 8, one camera: 3.2 M -> 0.4 M entries); a camera per image keeps its count.
    // Row block = the group's camera; column block = pose of oj (kind 0) or intrinsics of oj (kind 1, lower triangle k(oj) <= k(group); the diagonal
    // pair takes every oj of the group = the full block).  A CONSTANT point has T = 0: only its direct terms are listed.
    struct GEntry { int64_t key; int32_t oi, oj; };      // oi = group, oj = observation | (member of the group) << 31
    std::vector<GEntry> ge;
    std::vector<int32_t> grp_start(1, 0), grp_obs;
    std::vector<int32_t> grp_cam;
    {
      std::vector<std::pair<int32_t, int32_t>> ko;
      for (int p = 0; p < P; ++p) {
        ko.clear();
        for (int e = pt_start[p]; e < pt_start[p + 1]; ++e) {
          const int32_t o = pt_obs[e]; const int k = in_pose_camera[in_obs_pose[o]];
          if (intr_off[k] >= 0) ko.push_back({k, o});
        }
        std::stable_sort(ko.begin(), ko.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
        const size_t first_group = grp_cam.size();
        for (size_t a = 0; a < ko.size();) {
          size_t b = a;
          while (b < ko.size() && ko[b].first == ko[a].first) ++b;
          grp_cam.push_back(ko[a].first);
          for (size_t t = a; t < b; ++t) grp_obs.push_back(ko[t].second);
          grp_start.push_back((int32_t)grp_obs.size());
          a = b;
        }
        for (size_t g = first_group; g < grp_cam.size(); ++g) {
          const int ka = grp_cam[g];
          for (int f = pt_start[p]; f < pt_start[p + 1]; ++f) {
            const int32_t oj = pt_obs[f]; const int cj = in_obs_pose[oj]; const int kb = in_pose_camera[cj];
            const bool same = kb == ka;
            if (point_const[p] && !same) continue;
            const int32_t code = oj | (same ? (int32_t)0x80000000 : 0);
            if (!pose_const[cj]) ge.push_back({((int64_t)ka * 2 + 0) * (int64_t)(C + K) + cj, (int32_t)g, code});
            if (intr_off[kb] >= 0 && kb < ka) ge.push_back({((int64_t)ka * 2 + 1) * (int64_t)(C + K) + kb, (int32_t)g, code});      // (kb == ka: the diagonal block, k_intr_kk's)
          }
        }
      }
    }
    h->gen_num_groups = (int64_t)grp_cam.size();
    gen_grp_start = grp_start; gen_grp_obs = grp_obs;
    build_diag(kk); finish_diag(kk);
    std::sort(ge.begin(), ge.end(), [](const GEntry& a, const GEntry& b) {
      if (a.key != b.key) return a.key < b.key;
      if (a.oi != b.oi) return a.oi < b.oi;
      return (a.oj & 0x7fffffff) < (b.oj & 0x7fffffff);
    });
    // every variable block needs its own diagonal pair (it carries the damping) even without a local observation
    {
      std::vector<char> has_obs(K, 0);
      for (int64_t o = 0; o < M; ++o) has_obs[in_pose_camera[in_obs_pose[o]]] = 1;
      (void)has_obs;      // (the diagonal pairs - they carry the damping, also of a block without a local observation - are k_intr_kk's: build_diag lists every variable block)
      std::sort(ge.begin(), ge.end(), [](const GEntry& a, const GEntry& b) {
        if (a.key != b.key) return a.key < b.key;
        if (a.oi != b.oi) return a.oi < b.oi;
        return a.oj < b.oj;
      });
    }
    gen_entries.resize(2 * ge.size());
    gen_pair_chunk.push_back(0);
    size_t e = 0;
    while (e < ge.size()) {
      size_t f = e;
      while (f < ge.size() && ge[f].key == ge[e].key) ++f;
      const int64_t key = ge[e].key;
      const int col = (int)(key % (C + K)), kind = (int)((key / (C + K)) & 1), ka = (int)(key / (C + K) / 2);
      const int pair_id = (int)(gen_pair.size() / 4);
      gen_pair.push_back(6 * C + intr_off[ka]); gen_pair.push_back(intr_nv[ka]);
      if (kind == 0) { gen_pair.push_back(6 * col); gen_pair.push_back(6); }
      else { gen_pair.push_back(6 * C + intr_off[col]); gen_pair.push_back(intr_nv[col] | (1 << 8)); }
      if (ge[e].oi >= 0)    // (a block without observations has the placeholder entry only: a pair with no chunk)
        for (size_t c0 = e; c0 < f; c0 += kGenChunk) { gen_chunk.push_back(pair_id); gen_chunk.push_back((int32_t)c0); gen_chunk.push_back((int32_t)std::min(c0 + kGenChunk, f)); }
      gen_pair_chunk.push_back((int32_t)(gen_chunk.size() / 3));
      for (size_t g = e; g < f; ++g) { gen_entries[2 * g] = ge[g].oi; gen_entries[2 * g + 1] = ge[g].oj; }
      e = f;
    }
    }
    h->gen_num_pairs = (int64_t)(gen_pair.size() / 4); h->gen_num_chunks = (int64_t)(gen_chunk.size() / 3);
    for (int64_t pr = 0; pr < h->gen_num_pairs; ++pr) if (gen_pair_chunk[pr + 1] - gen_pair_chunk[pr] != 1) gen_multi.push_back((int32_t)pr);
    h->gen_num_multi = (int64_t)gen_multi.size();
    h->isum_num_chunks = (int64_t)(isum_chunk.size() / 3);
  }

  lap("chunks, intrinsics lists");
  const auto t_create3 = std::chrono::steady_clock::now();
  // ---- device allocation + upload --------------------------------------------------------------
  TRY(HandleAlloc(&h->la, M)); TRY(HandleAlloc(&h->lb, M)); TRY(HandleAlloc(&h->lc, M));
  if (!arrays_on_device) { TRY(HandleAlloc(&h->obs_pose, M)); TRY(HandleAlloc(&h->obs_point, M)); }
  TRY(HandleAlloc(&h->obs_cam, M));
  TRY(HandleAlloc(&h->pose_camera, C)); TRY(HandleAlloc(&h->camera_model, K));
  if (!arrays_on_device) { TRY(HandleAlloc(&h->pose_const, C)); TRY(HandleAlloc(&h->point_const, P)); TRY(HandleAlloc(&h->pt_start, P + 1)); TRY(HandleAlloc(&h->pt_obs, M)); }
  TRY(HandleAlloc(&h->tvec_mask, C));
  TRY(HandleAlloc(&h->pose_start, C + 1)); TRY(HandleAlloc(&h->pose_obs, M));
  TRY(HandleAlloc(&h->pair_start, pair_start.size())); TRY(HandleAlloc(&h->pair_ij, std::max<size_t>(pair_ij.size(), 2)));
  if (!h->pair_entries) TRY(HandleAlloc(&h->pair_entries, std::max<size_t>(pair_entries.size(), 2)));      // (built on the device: already there)
  TRY(HandleAlloc(&h->poses, (size_t)7 * C)); TRY(HandleAlloc(&h->points, (size_t)3 * P)); TRY(HandleAlloc(&h->intr, (size_t)kCamStride * K));
  TRY(HandleAlloc(&h->poses_c, (size_t)7 * C)); TRY(HandleAlloc(&h->points_c, (size_t)3 * P)); TRY(HandleAlloc(&h->intr_c, (size_t)kCamStride * K));
  TRY(HandleAlloc(&h->cam_np, K));
  TRY(HandleAlloc(&h->intr_off, K)); TRY(HandleAlloc(&h->intr_nv, K)); TRY(HandleAlloc(&h->intr_col, (size_t)K * kCamStride));
  if (NI > 0) {
    TRY(HandleAlloc(&h->cam_start, K + 1)); TRY(HandleAlloc(&h->cam_obs, M));
    TRY(HandleAlloc(&h->gen_pair, std::max<size_t>(gen_pair.size(), 4))); TRY(HandleAlloc(&h->gen_pair_chunk, gen_pair_chunk.size()));
    TRY(HandleAlloc(&h->gen_chunk, gen_chunk.size())); TRY(HandleAlloc(&h->gen_entries, std::max<size_t>(gen_entries.size(), 2)));
    TRY(HandleAlloc(&h->gen_multi, std::max<size_t>(gen_multi.size(), 1)));
    if (!iterative) {
      h->kk_num_groups = kk.num_groups; h->kk_num_pairs = (int64_t)(kk.pair.size() / 4); h->kk_num_chunks = (int64_t)(kk.chunk.size() / 3); h->kk_num_multi = (int64_t)kk.multi.size();
      TRY(HandleAlloc(&h->kk_entries, std::max<size_t>(kk.entries.size(), 1))); TRY(HandleAlloc(&h->kk_pair, std::max<size_t>(kk.pair.size(), 1)));
      TRY(HandleAlloc(&h->kk_pair_chunk, std::max<size_t>(kk.pair_chunk.size(), 1))); TRY(HandleAlloc(&h->kk_chunk, std::max<size_t>(kk.chunk.size(), 1)));
      TRY(HandleAlloc(&h->kk_multi, std::max<size_t>(kk.multi.size(), 1))); TRY(HandleAlloc(&h->kk_partial, 144 * std::max<size_t>((size_t)h->kk_num_chunks, 1)));
      TRY(Upload(h->kk_entries, kk.entries.data(), kk.entries.size(), s)); TRY(Upload(h->kk_pair, kk.pair.data(), kk.pair.size(), s));
      TRY(Upload(h->kk_pair_chunk, kk.pair_chunk.data(), kk.pair_chunk.size(), s)); TRY(Upload(h->kk_chunk, kk.chunk.data(), kk.chunk.size(), s));
      TRY(Upload(h->kk_multi, kk.multi.data(), kk.multi.size(), s));
      TRY(HandleAlloc(&h->gen_grp_start, std::max<size_t>(gen_grp_start.size(), 1))); TRY(HandleAlloc(&h->gen_grp_obs, std::max<size_t>(gen_grp_obs.size(), 1)));
      TRY(HandleAlloc(&h->gen_L, 36 * std::max<size_t>((size_t)h->gen_num_groups, 1)));
    }
    TRY(HandleAlloc(&h->isum_chunk, isum_chunk.size())); TRY(HandleAlloc(&h->isum_cam_chunk, isum_cam_chunk.size()));
    TRY(HandleAlloc(&h->gen_partial, (size_t)std::max<int64_t>(h->gen_num_chunks, 1) * 144)); TRY(HandleAlloc(&h->isum_partial, (size_t)std::max<int64_t>(h->isum_num_chunks, 1) * 24));
    TRY(HandleAlloc(&h->cnI, (size_t)NI)); TRY(HandleAlloc(&h->JkS_intr, (size_t)M * 2 * kCamStride));
    TRYH(hipMemsetAsync(h->JkS_intr, 0, sizeof(double) * (size_t)M * 2 * kCamStride, s));      // (k_intr_prepare only ever writes a camera's variable columns)
  }
  TRY(HandleAlloc(&h->r, (size_t)2 * M)); TRY(HandleAlloc(&h->Jpoint, (size_t)6 * M));
  h->num_partials = CeilDiv(M, 256);
  h->partials_stride = std::max(std::max(h->num_partials, 4096), CeilDiv(4 * (int64_t)P, 256));      // (k_step_points: one partial per 64 points)
  TRY(HandleAlloc(&h->partials, 2 * (size_t)h->partials_stride));     // K1's cost partials, then the model-cost partials
  // the int32 flag words live in the last scalar slot (+ one more double), so ONE copy of kNumScalars doubles reads back the
  // scalars and the failure flag
  TRY(HandleAlloc(&h->scal, kNumScalars + 1));
  h->d_flag = reinterpret_cast<int32_t*>(h->scal + kNumScalars - 1);
  TRY(PoolPinnedAlloc(reinterpret_cast<void**>(&h->h_scal), sizeof(double) * 3 * kNumScalars));   // read-back + two evaluation slots
  std::memset(h->h_scal, 0, sizeof(double) * 3 * kNumScalars);     // the ticket slot starts at 0 = "no ticket"
  if (hipHostGetDevicePointer(reinterpret_cast<void**>(&h->h_scal_dev), h->h_scal, 0) != hipSuccess) { h->h_scal_dev = nullptr; (void)hipGetLastError(); }
  TRYH(hipMemsetAsync(h->scal, 0, sizeof(double) * (kNumScalars + 1), s));

  TRY(Upload(h->la, la.data(), M, s)); TRY(Upload(h->lb, lb.data(), M, s)); TRY(Upload(h->lc, lc.data(), M, s));
  if (!arrays_on_device) { TRY(Upload(h->obs_pose, in_obs_pose, M, s)); TRY(Upload(h->obs_point, d->obs_point, M, s)); }
  TRY(Upload(h->obs_cam, obs_cam.data(), M, s));
  TRY(Upload(h->pose_camera, in_pose_camera, C, s)); TRY(Upload(h->camera_model, d->camera_model, K, s));
  if (!arrays_on_device) { TRY(Upload(h->pose_const, pose_const.data(), C, s)); TRY(Upload(h->point_const, point_const.data(), P, s)); }
  TRY(Upload(h->tvec_mask, tvec_mask.data(), C, s));
  {  // effective parameters (tangent dimensions of the variable blocks): fixed with the masks, reported by every solve
    int neff = 0;
    for (int c = 0; c < C; ++c) if (!pose_const[c]) neff += 6 - __builtin_popcount(tvec_mask[c] & 7);
    for (int p = 0; p < P; ++p) if (!point_const[p]) neff += 3;
    h->num_effective_pose_point = neff;
  }
  if (!arrays_on_device) { TRY(Upload(h->pt_start, pt_start.data(), P + 1, s)); TRY(Upload(h->pt_obs, pt_obs.data(), M, s)); }
  TRY(Upload(h->pose_start, pose_start.data(), C + 1, s)); TRY(Upload(h->pose_obs, pose_obs.data(), M, s));
  TRY(Upload(h->pair_start, pair_start.data(), pair_start.size(), s));
  TRY(Upload(h->pair_ij, pair_ij.data(), pair_ij.size(), s));
  if (!lists_on_device) TRY(Upload(h->pair_entries, pair_entries.data(), pair_entries.size(), s));
  { std::vector<int32_t> np(K); for (int k = 0; k < K; ++k) np[k] = CameraNumParams(d->camera_model[k]); TRY(Upload(h->cam_np, np.data(), K, s)); TRYH(hipStreamSynchronize(s)); }
  TRY(HandleAlloc(&h->spos, std::max<size_t>(h->spos_host.size(), 1))); TRY(Upload(h->spos, h->spos_host.data(), h->spos_host.size(), s));
  TRY(Upload(h->intr_off, intr_off.data(), K, s)); TRY(Upload(h->intr_nv, intr_nv.data(), K, s)); TRY(Upload(h->intr_col, intr_col.data(), intr_col.size(), s));
  if (NI > 0) {
    TRY(Upload(h->cam_start, cam_start.data(), K + 1, s)); TRY(Upload(h->cam_obs, cam_obs.data(), M, s));
    TRY(Upload(h->gen_pair, gen_pair.data(), gen_pair.size(), s)); TRY(Upload(h->gen_pair_chunk, gen_pair_chunk.data(), gen_pair_chunk.size(), s));
    TRY(Upload(h->gen_chunk, gen_chunk.data(), gen_chunk.size(), s)); TRY(Upload(h->gen_entries, gen_entries.data(), gen_entries.size(), s));
    TRY(Upload(h->gen_multi, gen_multi.data(), gen_multi.size(), s));
    if (!iterative) { TRY(Upload(h->gen_grp_start, gen_grp_start.data(), gen_grp_start.size(), s)); TRY(Upload(h->gen_grp_obs, gen_grp_obs.data(), gen_grp_obs.size(), s)); }
    TRY(Upload(h->isum_chunk, isum_chunk.data(), isum_chunk.size(), s)); TRY(Upload(h->isum_cam_chunk, isum_cam_chunk.data(), isum_cam_chunk.size(), s));
  }
  if (want_chunks) {
    TRY(HandleAlloc(&h->small_chunk, std::max<size_t>(small_chunk.size(), 3))); TRY(HandleAlloc(&h->small_pair_chunk, small_pair_chunk.size()));
    TRY(HandleAlloc(&h->small_partials, 36 * std::max<size_t>((size_t)h->small_num_chunks, 1)));
    TRY(Upload(h->small_chunk, small_chunk.data(), small_chunk.size(), s)); TRY(Upload(h->small_pair_chunk, small_pair_chunk.data(), small_pair_chunk.size(), s));
  }
  TRYH(hipStreamSynchronize(s));  // host staging vectors die at scope exit
#undef TRY
#undef TRYH
  {
    const auto t_create4 = std::chrono::steady_clock::now();
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    // (the image order: the graph's pass on the device + ChooseImageOrdering; the handle, the by-point lists and their upload count with the pair lists)
    h->create_ms[0] = ordering_ms; h->create_ms[1] = ms(t_create0, t_create2) - ordering_ms; h->create_ms[2] = ms(t_create2, t_create3);
    h->create_ms[3] = ms(t_create3, t_create4); h->create_ms[4] = 0; h->create_ms[5] = ms(t_create0, t_create4);
  }
  *out = h;
  return PP_OK;
} PP_API_CATCH("pp_ba_create")

int pp_ba_get_create_profile(pp_ba_handle h, double* ms) try {
  PP_REQUIRE(h && ms, "pp_ba_get_create_profile: null argument");
  for (int i = 0; i < 6; ++i) ms[i] = h->create_ms[i];
  ms[4] = h->chol_aux.plan_ms;      // (the task plan is made with the solver buffers, at the first solve or attach)
  return PP_OK;
} PP_API_CATCH("pp_ba_get_create_profile")

int pp_ba_covisibility(const pp_ba_problem_desc* d, uint8_t* out) try {
  PP_REQUIRE(d && out && d->obs_pose && d->obs_point, "pp_ba_covisibility: null argument");
  const int C = d->num_poses, P = d->num_points;
  const int64_t M = d->num_obs;
  PP_REQUIRE(C > 0 && P > 0 && M >= 0, "pp_ba_covisibility: empty problem");
  if (d->camera_const_mask) {      // (PrivateIntrinsicsColumns walks the cameras of the images: the same checks as pp_ba_create / pp_ba_plan_ordering)
    PP_REQUIRE(d->pose_camera && d->camera_model && d->num_cameras > 0, "pp_ba_covisibility: camera_const_mask without pose_camera / camera_model");
    for (int k = 0; k < d->num_cameras; ++k) PP_REQUIRE(CameraNumParams(d->camera_model[k]) > 0, "pp_ba_covisibility: unknown camera model %d", d->camera_model[k]);
    for (int c = 0; c < C; ++c) PP_REQUIRE(d->pose_camera[c] >= 0 && d->pose_camera[c] < d->num_cameras, "pp_ba_covisibility: pose_camera[%d] out of range", c);
  }
  for (int64_t o = 0; o < M; ++o)
    PP_REQUIRE(d->obs_pose[o] >= 0 && d->obs_pose[o] < C && d->obs_point[o] >= 0 && d->obs_point[o] < P, "pp_ba_covisibility: observation %lld indexes out of range", (long long)o);
  std::memset(out, 0, (size_t)C * C);
  std::vector<int32_t> ps(P + 1, 0), po(M);
  for (int64_t o = 0; o < M; ++o) ps[d->obs_point[o] + 1]++;
  for (int p = 0; p < P; ++p) ps[p + 1] += ps[p];
  { std::vector<int32_t> f(ps.begin(), ps.end() - 1); for (int64_t o = 0; o < M; ++o) po[f[d->obs_point[o]]++] = d->obs_pose[o]; }
  const uint8_t* fixed = (d->camera_const_mask && ppsfm::PrivateIntrinsicsColumns(d) > 0) ? nullptr : d->pose_const;      // (intrinsics of its own beside the pose: every image has columns)
  for (int p = 0; p < P; ++p) {
    if (d->point_const && d->point_const[p]) continue;
    for (int a = ps[p]; a < ps[p + 1]; ++a) {
      const int ca = po[a];
      if (fixed && fixed[ca]) continue;
      for (int b = ps[p]; b < a; ++b) {
        const int cb = po[b];
        if (cb == ca || (fixed && fixed[cb])) continue;
        out[(size_t)ca * C + cb] = 1; out[(size_t)cb * C + ca] = 1;
      }
    }
  }
  return PP_OK;
} PP_API_CATCH("pp_ba_covisibility")

int pp_ba_set_parameters(pp_ba_handle h, const double* poses, const double* points, const double* intr) try {
  PP_REQUIRE(h, "pp_ba_set_parameters: null handle");
  PP_HIP_TRY(hipSetDevice(h->device));
  std::vector<double> staged;      // (the caller's image order -> the handle's)
  if (poses && !h->pose_new_of_old.empty()) {
    staged.resize((size_t)7 * h->C);
    for (int c = 0; c < h->C; ++c) std::memcpy(&staged[(size_t)7 * h->pose_new_of_old[c]], poses + (size_t)7 * c, 7 * sizeof(double));
    poses = staged.data();
  }
  if (poses && (int)h->host_pose_const.size() == h->C)
    for (int c = 0; c < h->C; ++c) {
      // "CostFunction assumes unit quaternions" (bundle_adjustment.cc:354-355: AddImageToProblem normalises first): the Jacobian on the rotation tangent of a
      // VARIABLE pose is exact for unit q only - a caller that skipped the normalisation is told so instead of being given other steps than Ceres'
      // (a constant pose only enters through the rotate-point polynomial, as in the reference; NaN passes and fails the solve as before)
      const double* q = poses + (size_t)7 * c;
      const double n2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
      PP_REQUIRE(h->host_pose_const[c] || !(std::fabs(n2 - 1.0) > 1e-6),
                 "pp_ba_set_parameters: the quaternion of a variable pose is not of unit length (|q|^2 = %.9g at internal image %d); normalise it as "
                 "BundleAdjuster::AddImageToProblem does (Image::NormalizeQvec)", n2, c);
    }
  if (poses) { int rc = Upload(h->poses, poses, (size_t)7 * h->C, h->stream); if (rc) return rc; }
  if (points) { int rc = Upload(h->points, points, (size_t)3 * h->P, h->stream); if (rc) return rc; }
  if (intr) { int rc = Upload(h->intr, intr, (size_t)kCamStride * h->K, h->stream); if (rc) return rc; }
  PP_HIP_TRY(hipStreamSynchronize(h->stream));
  return PP_OK;
} PP_API_CATCH("pp_ba_set_parameters")

int pp_ba_get_parameters(pp_ba_handle h, double* poses, double* points, double* intr) try {
  PP_REQUIRE(h, "pp_ba_get_parameters: null handle");
  PP_HIP_TRY(hipSetDevice(h->device));
  std::vector<double> staged;
  const bool perm = poses && !h->pose_new_of_old.empty();
  if (perm) staged.resize((size_t)7 * h->C);
  if (poses) { int rc = Download(perm ? staged.data() : poses, h->poses, (size_t)7 * h->C, h->stream); if (rc) return rc; }
  if (points) { int rc = Download(points, h->points, (size_t)3 * h->P, h->stream); if (rc) return rc; }
  if (intr) { int rc = Download(intr, h->intr, (size_t)kCamStride * h->K, h->stream); if (rc) return rc; }
  PP_HIP_TRY(hipStreamSynchronize(h->stream));
  if (perm) for (int c = 0; c < h->C; ++c) std::memcpy(poses + (size_t)7 * c, &staged[(size_t)7 * h->pose_new_of_old[c]], 7 * sizeof(double));
  return PP_OK;
} PP_API_CATCH("pp_ba_get_parameters")

int pp_ba_eval(pp_ba_handle h, int jac_mode, int want_cam, double* residuals_out, double* jpose_out, double* jpoint_out,
               double* jcam_out, double* cost_out) try {
  PP_REQUIRE(h, "pp_ba_eval: null handle");
  PP_REQUIRE(jac_mode == 0 || jac_mode == 1, "pp_ba_eval: jac_mode must be 0 (tangent) or 1 (ambient)");
  PP_HIP_TRY(hipSetDevice(h->device));
  int rc = BaEnsureJacobianBuffers(h, jac_mode, want_cam || jcam_out != nullptr);
  if (rc) return rc;
  const int cam = (want_cam || jcam_out) ? 1 : 0;
  rc = LaunchEval(h, jac_mode, cam, false, h->poses, h->points, h->scal + kCost);
  if (rc) return rc;
  const int width = jac_mode == 1 ? 14 : 12;
  if (residuals_out) { rc = Download(residuals_out, h->r, (size_t)2 * h->M, h->stream); if (rc) return rc; }
  if (jpose_out) { rc = Download(jpose_out, h->Jpose, (size_t)width * h->M, h->stream); if (rc) return rc; }
  if (jpoint_out) { rc = Download(jpoint_out, h->Jpoint, (size_t)6 * h->M, h->stream); if (rc) return rc; }
  if (jcam_out) { rc = Download(jcam_out, h->Jcam, (size_t)2 * kCamStride * h->M, h->stream); if (rc) return rc; }
  if (cost_out) { rc = Download(cost_out, h->scal + kCost, 1, h->stream); if (rc) return rc; }
  PP_HIP_TRY(hipStreamSynchronize(h->stream));
  return PP_OK;
} PP_API_CATCH("pp_ba_eval")

int pp_ba_eval_host_view(pp_ba_handle h, int jac_mode, int want_cam, int want_jacobians, const double** residuals, const double** jpose,
                         const double** jpoint, const double** jcam, double* cost_out) try {
  PP_REQUIRE(h && residuals, "pp_ba_eval_host_view: null argument");
  PP_REQUIRE(jac_mode == 0 || jac_mode == 1, "pp_ba_eval_host_view: jac_mode must be 0 (tangent) or 1 (ambient)");
  PP_HIP_TRY(hipSetDevice(h->device));
  const int cam = want_cam ? 1 : 0, width = jac_mode == 1 ? 14 : 12;
  int rc = BaEnsureJacobianBuffers(h, jac_mode, cam);
  if (rc) return rc;
  const size_t M = (size_t)h->M;
  if (!h->pin_r) PP_HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&h->pin_r), sizeof(double) * 2 * M));
  if (want_jacobians) {
    if (h->pin_jpose && h->pin_width < width) { (void)hipHostFree(h->pin_jpose); h->pin_jpose = nullptr; }
    if (!h->pin_jpose) { PP_HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&h->pin_jpose), sizeof(double) * width * M)); h->pin_width = width; }
    if (!h->pin_jpoint) PP_HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&h->pin_jpoint), sizeof(double) * 6 * M));
    if (cam && !h->pin_jcam) PP_HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&h->pin_jcam), sizeof(double) * 2 * kCamStride * M));
  }
  // a cost-only evaluation (Ceres asks for residuals without Jacobians at every trial point) runs K1's cost-only variant and
  // moves 16 B per observation instead of 220 B+
  if (want_jacobians) rc = LaunchEval(h, jac_mode, cam, false, h->poses, h->points, h->scal + kCost);
  else rc = LaunchResidualsOnly(h, h->poses, h->points, h->scal + kCost);
  if (rc) return rc;
  rc = Download(h->pin_r, h->r, 2 * M, h->stream); if (rc) return rc;
  if (want_jacobians) {
    rc = Download(h->pin_jpose, h->Jpose, (size_t)width * M, h->stream); if (rc) return rc;
    rc = Download(h->pin_jpoint, h->Jpoint, 6 * M, h->stream); if (rc) return rc;
    if (cam) { rc = Download(h->pin_jcam, h->Jcam, (size_t)2 * kCamStride * M, h->stream); if (rc) return rc; }
  }
  if (cost_out) { rc = Download(cost_out, h->scal + kCost, 1, h->stream); if (rc) return rc; }
  PP_HIP_TRY(hipStreamSynchronize(h->stream));
  *residuals = h->pin_r;
  if (jpose) *jpose = want_jacobians ? h->pin_jpose : nullptr;
  if (jpoint) *jpoint = want_jacobians ? h->pin_jpoint : nullptr;
  if (jcam) *jcam = (want_jacobians && cam) ? h->pin_jcam : nullptr;
  return PP_OK;
} PP_API_CATCH("pp_ba_eval_host_view")

int pp_ba_eval_device(pp_ba_handle h, int jac_mode, int want_cam, int repeat, float* ms_per_launch) try {
  PP_REQUIRE(h && repeat > 0, "pp_ba_eval_device: bad argument");
  PP_REQUIRE(jac_mode == 0 || jac_mode == 1, "pp_ba_eval_device: jac_mode must be 0 or 1");
  PP_HIP_TRY(hipSetDevice(h->device));
  int rc = BaEnsureJacobianBuffers(h, jac_mode, want_cam);
  if (rc) return rc;
  PP_HIP_TRY(hipEventRecord(h->ev0, h->stream));
  for (int i = 0; i < repeat; ++i) {
    rc = LaunchEval(h, jac_mode, want_cam, false, h->poses, h->points, nullptr);
    if (rc) return rc;
  }
  PP_HIP_TRY(hipEventRecord(h->ev1, h->stream));
  PP_HIP_TRY(hipEventSynchronize(h->ev1));
  float ms = 0;
  PP_HIP_TRY(hipEventElapsedTime(&ms, h->ev0, h->ev1));
  if (ms_per_launch) *ms_per_launch = ms / repeat;
  return PP_OK;
} PP_API_CATCH("pp_ba_eval_device")

}  // extern "C"

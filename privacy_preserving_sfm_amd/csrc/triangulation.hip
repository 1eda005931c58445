// K8 — batched robust track triangulation: thousands of tiny, independent LORANSAC problems
// (one per track) — the mapper's IncrementalTriangulator calls EstimateTriangulation per track
// (reference src/sfm/incremental_triangulator.cc:214, 536).
//   EstimateTriangulation, TriangulationEstimator::{Estimate, Residuals}      src/estimators/triangulation.cc:55-149
//   TriangulateMultiViewPoint (null vector of the K x 4 system [l_i^T P_i])    src/base/triangulation.cc:41-57
//   LORANSAC<..., InlierSupportMeasurer, CombinationSampler>::Estimate          src/optim/loransac.h:88-235
//   CombinationSampler: the 3-combinations in lexicographic order               src/optim/combination_sampler.cc:41-70
//   residuals: squared pixel line error / squared angular line error            src/base/projection.cc:161-203, 238-262
// One LANE per track: the whole RANSAC (it draws no random numbers) runs in the lane; the null vector of a minimal
// sample is the vector of signed 3x3 minors, the local optimisation's is the smallest eigenvector of the 4x4 Gram
// matrix (cyclic Jacobi in registers).  The current inlier flags of a track live in its slice of the output mask.
#include <algorithm>
#include <cfloat>

#include "camera_models.hpp"
#include "common.hpp"
#include "ransac_host.hpp"
#include "small_eigen.hpp"

namespace ppsfm {

struct TriArgs {
  int T;
  const int32_t *track_start, *obs_view, *view_camera, *camera_model, *cam_size;
  const double *lines, *P, *centers, *intr;
  double min_tri_angle, max_residual, confidence, multiplier;
  int residual_type;
  unsigned long long min_num_trials, max_num_trials;
  uint8_t *success, *mask;
  double* xyz;
  int32_t* num_trials;
};

__device__ __forceinline__ double TriProjZ(const double* P, const double* X) { return P[8] * X[0] + P[9] * X[1] + P[10] * X[2] + P[11]; }

__device__ __forceinline__ double TriResidual(const TriArgs& a, int o, const double* X) {
  const int v = a.obs_view[o], k = a.view_camera[v];
  const double* P = a.P + 12 * (size_t)v;
  const double* l = a.lines + 3 * (size_t)o;
  const double* cam = a.intr + (size_t)kCamStride * k;
  const int model = a.camera_model[k];
  const double w = (double)a.cam_size[2 * k], h = (double)a.cam_size[2 * k + 1];
  const double r0 = P[0] * X[0] + P[1] * X[1] + P[2] * X[2] + P[3], r1 = P[4] * X[0] + P[5] * X[1] + P[6] * X[2] + P[7], r2 = TriProjZ(P, X);
  if (a.residual_type == 1) {          // CalculateSquaredLineReprojectionError
    if (r2 < DBL_EPSILON) return DBL_MAX;
    const double inv = 1.0 / r2, u = inv * r0, vv = inv * r1;
    const double alpha = l[0] * u + l[1] * vv + l[2];
    double ix, iy, jx, jy;
    WorldToImage<double, double>(model, cam, u, vv, &ix, &iy);
    if (!(ix >= 0 && ix < w && iy >= 0 && iy < h)) return DBL_MAX;
    WorldToImage<double, double>(model, cam, u - l[0] * alpha, vv - l[1] * alpha, &jx, &jy);
    return (ix - jx) * (ix - jx) + (iy - jy) * (iy - jy);
  }
  // CalculateNormalizedLineAngularError, squared
  if (r2 < 0.0) return DBL_MAX;
  double ix, iy;
  WorldToImage<double, double>(model, cam, r0 / r2, r1 / r2, &ix, &iy);
  if (ix < 0 || ix >= w || iy < 0 || iy >= h) return DBL_MAX;
  const double nl = sqrt(l[0] * l[0] + l[1] * l[1] + l[2] * l[2]), nr = sqrt(r0 * r0 + r1 * r1 + r2 * r2);
  const double ang = fabs(1.57079632679489661923 - acos(fabs((l[0] * r0 + l[1] * r1 + l[2] * r2) / (nl * nr))));
  return ang * ang;
}

__device__ __forceinline__ double TriAngle(const double* c1, const double* c2, const double* X) {
  double b2 = 0, r1 = 0, r2 = 0;
#pragma unroll
  for (int i = 0; i < 3; ++i) { b2 += (c1[i] - c2[i]) * (c1[i] - c2[i]); r1 += (X[i] - c1[i]) * (X[i] - c1[i]); r2 += (X[i] - c2[i]) * (X[i] - c2[i]); }
  const double den = 2.0 * sqrt(r1 * r2);
  if (den == 0.0) return 0.0;
  const double ang = fabs(acos((r1 + r2 - b2) / den));
  return fmin(ang, 3.14159265358979323846 - ang);
}

__device__ __forceinline__ void TriRow(const TriArgs& a, int o, double row[4]) {
  const double* P = a.P + 12 * (size_t)a.obs_view[o];
  const double* l = a.lines + 3 * (size_t)o;
#pragma unroll
  for (int c = 0; c < 4; ++c) row[c] = l[0] * P[c] + l[1] * P[4 + c] + l[2] * P[8 + c];
}
__device__ __forceinline__ double Det3(const double* a, const double* b, const double* c, int i0, int i1, int i2) {
  return a[i0] * (b[i1] * c[i2] - b[i2] * c[i1]) - a[i1] * (b[i0] * c[i2] - b[i2] * c[i0]) + a[i2] * (b[i0] * c[i1] - b[i1] * c[i0]);
}

// residual pass over the track: support of X; with `flags` the per-observation inlier flags are written
__device__ __forceinline__ void TriSupport(const TriArgs& a, int e0, int n, const double* X, unsigned long long* num_inliers, double* residual_sum, uint8_t* flags) {
  unsigned long long cnt = 0;
  double sum = 0.0;
  for (int i = 0; i < n; ++i) {
    const double r = TriResidual(a, e0 + i, X);
    const bool in = r <= a.max_residual;
    if (in) { ++cnt; sum += r; }
    if (flags) flags[e0 + i] = in ? 1 : 0;
  }
  *num_inliers = cnt; *residual_sum = sum;
}

__device__ __forceinline__ unsigned long long TriNumTrials(unsigned long long num_inliers, unsigned long long num_samples, double confidence, double multiplier) {
  const double ratio = (double)num_inliers / (double)num_samples;      // RANSAC::ComputeNumTrials (optim/ransac.h:158-176)
  const double nom = 1.0 - confidence;
  if (nom <= 0) return 0xFFFFFFFFFFFFFFFFull;
  const double denom = 1.0 - pow(ratio, 3.0);
  if (denom <= 0) return 1;
  const double v = ceil(log(nom) / log(denom) * multiplier);
  // zero inliers give log(1) = 0 in the denominator, i.e. -inf: the reference's static_cast<size_t> of that is undefined
  // behaviour which on its x86-64 hosts yields 2^63 ("never abort"); a GPU conversion would saturate to 0 and abort at once
  if (!(v >= 0.0 && v < 1.8e19)) return 0x8000000000000000ull;
  return (unsigned long long)v;
}

__global__ __launch_bounds__(64) void k_triangulate_tracks(TriArgs a) {
  const int t = blockIdx.x * 64 + threadIdx.x;
  if (t >= a.T) return;
  const int e0 = a.track_start[t], n = a.track_start[t + 1] - e0;
  a.success[t] = 0; a.num_trials[t] = 0;
  a.xyz[3 * t] = a.xyz[3 * t + 1] = a.xyz[3 * t + 2] = 0.0;
  for (int i = 0; i < n; ++i) a.mask[e0 + i] = 0;
  if (n < 3) return;                                                   // triangulation.cc:124-125
  unsigned long long best_inl = 0;
  double best_sum = DBL_MAX, best[3] = {0, 0, 0};
  const unsigned long long nck = (unsigned long long)n * (n - 1) * (n - 2) / 6;
  const unsigned long long max_trials = a.max_num_trials < nck ? a.max_num_trials : nck;
  unsigned long long dyn = max_trials, trials = 0;
  bool abort = false;
  int c0 = 0, c1 = 1, c2 = 2;
  for (trials = 0; trials < max_trials; ++trials) {
    if (abort) { trials += 1; break; }
    const int s0 = c0, s1 = c1, s2 = c2;
    if (c2 + 1 < n) ++c2;                                              // next 3-combination, lexicographic, wrapping
    else if (c1 + 2 < n) { ++c1; c2 = c1 + 1; }
    else if (c0 + 3 < n) { ++c0; c1 = c0 + 1; c2 = c0 + 2; }
    else { c0 = 0; c1 = 1; c2 = 2; }
    // Estimate on the minimal sample: null vector of the 3 x 4 system = signed 3x3 minors
    double ra[4], rb[4], rc[4];
    TriRow(a, e0 + s0, ra); TriRow(a, e0 + s1, rb); TriRow(a, e0 + s2, rc);
    const double h0 = Det3(ra, rb, rc, 1, 2, 3), h1 = -Det3(ra, rb, rc, 0, 2, 3), h2 = Det3(ra, rb, rc, 0, 1, 3), h3 = -Det3(ra, rb, rc, 0, 1, 2);
    double X[3] = {h0 / h3, h1 / h3, h2 / h3};
    const int sv[3] = {a.obs_view[e0 + s0], a.obs_view[e0 + s1], a.obs_view[e0 + s2]};
    bool ok = true;
#pragma unroll
    for (int i = 0; i < 3; ++i) ok = ok && (TriProjZ(a.P + 12 * (size_t)sv[i], X) >= DBL_EPSILON);
    if (!ok) continue;
    ok = TriAngle(a.centers + 3 * (size_t)sv[1], a.centers + 3 * (size_t)sv[0], X) >= a.min_tri_angle ||
         TriAngle(a.centers + 3 * (size_t)sv[2], a.centers + 3 * (size_t)sv[0], X) >= a.min_tri_angle ||
         TriAngle(a.centers + 3 * (size_t)sv[2], a.centers + 3 * (size_t)sv[1], X) >= a.min_tri_angle;
    if (!ok) continue;
    unsigned long long inl; double sum;
    TriSupport(a, e0, n, X, &inl, &sum, a.mask);
    if (inl > best_inl || (inl == best_inl && sum < best_sum)) {
      best_inl = inl; best_sum = sum; best[0] = X[0]; best[1] = X[1]; best[2] = X[2];
      if (inl > 3) {                                                   // local optimisation on the inliers (loransac.h:157-187)
        double S[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) S[e] = 0.0;
        for (int i = 0; i < n; ++i) {
          if (!a.mask[e0 + i]) continue;
          double row[4];
          TriRow(a, e0 + i, row);
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) S[4 * r + c] += row[r] * row[c];
        }
        double hv[4];
        SmallestEigenvector<4>(S, hv);
        double L[3] = {hv[0] / hv[3], hv[1] / hv[3], hv[2] / hv[3]};
        bool lok = true;
        for (int i = 0; i < n && lok; ++i) if (a.mask[e0 + i]) lok = TriProjZ(a.P + 12 * (size_t)a.obs_view[e0 + i], L) >= DBL_EPSILON;
        if (lok) {
          lok = false;
          for (int i = 0; i < n && !lok; ++i) {
            if (!a.mask[e0 + i]) continue;
            for (int j = 0; j < i; ++j)
              if (a.mask[e0 + j] && TriAngle(a.centers + 3 * (size_t)a.obs_view[e0 + i], a.centers + 3 * (size_t)a.obs_view[e0 + j], L) >= a.min_tri_angle) { lok = true; break; }
          }
        }
        if (lok) {
          unsigned long long linl; double lsum;
          TriSupport(a, e0, n, L, &linl, &lsum, nullptr);
          if (linl > best_inl || (linl == best_inl && lsum < best_sum)) { best_inl = linl; best_sum = lsum; best[0] = L[0]; best[1] = L[1]; best[2] = L[2]; }
        }
      }
      dyn = TriNumTrials(best_inl, (unsigned long long)n, a.confidence, a.multiplier);
    }
    if (trials >= dyn && trials >= a.min_num_trials) abort = true;
  }
  a.num_trials[t] = (int32_t)(trials > 0x7FFFFFFFull ? 0x7FFFFFFFull : trials);
  a.xyz[3 * t] = best[0]; a.xyz[3 * t + 1] = best[1]; a.xyz[3 * t + 2] = best[2];
  if (best_inl < 3) { for (int i = 0; i < n; ++i) a.mask[e0 + i] = 0; return; }
  a.success[t] = 1;
  unsigned long long inl; double sum;
  TriSupport(a, e0, n, best, &inl, &sum, a.mask);                        // the reported inlier mask (loransac.h:213-233)
}

}  // namespace ppsfm

using namespace ppsfm;

extern "C" int pp_triangulate_tracks(int device, int32_t num_tracks, const int32_t* track_start, const double* lines, const int32_t* obs_view, int32_t num_views,
                                     const double* proj_matrices, const double* proj_centers, const int32_t* view_camera, int32_t num_cameras,
                                     const int32_t* camera_model, const double* intr, const int32_t* cam_size, const pp_triangulation_options* o, uint8_t* success,
                                     double* xyz, uint8_t* inlier_mask, int32_t* num_trials, float* device_ms) try {
  PP_REQUIRE(num_tracks >= 0 && num_views > 0 && num_cameras > 0 && o && (num_tracks == 0 || (track_start && lines && obs_view && success && xyz && inlier_mask && num_trials)) &&
                 proj_matrices && proj_centers && view_camera && camera_model && intr && cam_size,
             "pp_triangulate_tracks: bad argument");
  PP_REQUIRE(o->min_tri_angle >= 0 && o->ransac.max_error > 0 && (o->residual_type == 0 || o->residual_type == 1), "pp_triangulate_tracks: bad options");
  if (num_tracks == 0) return PP_OK;
  const int64_t N = track_start[num_tracks];
  for (int64_t i = 0; i < N; ++i) PP_REQUIRE(obs_view[i] >= 0 && obs_view[i] < num_views, "pp_triangulate_tracks: view index out of range");
  for (int v = 0; v < num_views; ++v) PP_REQUIRE(view_camera[v] >= 0 && view_camera[v] < num_cameras, "pp_triangulate_tracks: camera index out of range");
  for (int k = 0; k < num_cameras; ++k) PP_REQUIRE(pp_camera_num_params(camera_model[k]) > 0, "pp_triangulate_tracks: unknown camera model");
  int ndev = 0;
  PP_HIP_TRY(hipGetDeviceCount(&ndev));
  PP_REQUIRE(device >= 0 && device < ndev, "pp_triangulate_tracks: device %d of %d", device, ndev);
  PP_HIP_TRY(hipSetDevice(device));
  TriArgs a{};
  int32_t *d_ts = nullptr, *d_ov = nullptr, *d_vc = nullptr, *d_cm = nullptr, *d_cs = nullptr, *d_nt = nullptr;
  double *d_l = nullptr, *d_P = nullptr, *d_c = nullptr, *d_in = nullptr, *d_xyz = nullptr;
  uint8_t *d_s = nullptr, *d_m = nullptr;
  hipStream_t s = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  int rc = PP_OK;
  auto cleanup = [&]() {
    void* b[] = {d_ts, d_ov, d_vc, d_cm, d_cs, d_nt, d_l, d_P, d_c, d_in, d_xyz, d_s, d_m};
    for (void* p : b) if (p) (void)hipFree(p);
    if (ev0) (void)hipEventDestroy(ev0);
    if (ev1) (void)hipEventDestroy(ev1);
    if (s) (void)hipStreamDestroy(s);
  };
  OnUnwind unwind{[&] { cleanup(); }};
#define TRY(x) do { rc = (x); if (rc) { cleanup(); return rc; } } while (0)
#define TRYH(x) do { if ((x) != hipSuccess) { SetLastError("pp_triangulate_tracks: %s failed", #x); cleanup(); return PP_ERR_HIP; } } while (0)
  TRYH(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); TRYH(hipEventCreate(&ev0)); TRYH(hipEventCreate(&ev1));
  TRY(DeviceAlloc(&d_ts, (size_t)num_tracks + 1)); TRY(DeviceAlloc(&d_ov, (size_t)N)); TRY(DeviceAlloc(&d_vc, (size_t)num_views)); TRY(DeviceAlloc(&d_cm, (size_t)num_cameras));
  TRY(DeviceAlloc(&d_cs, (size_t)2 * num_cameras)); TRY(DeviceAlloc(&d_nt, (size_t)num_tracks)); TRY(DeviceAlloc(&d_l, (size_t)3 * N)); TRY(DeviceAlloc(&d_P, (size_t)12 * num_views));
  TRY(DeviceAlloc(&d_c, (size_t)3 * num_views)); TRY(DeviceAlloc(&d_in, (size_t)kCamStride * num_cameras)); TRY(DeviceAlloc(&d_xyz, (size_t)3 * num_tracks));
  TRY(DeviceAlloc(&d_s, (size_t)num_tracks)); TRY(DeviceAlloc(&d_m, (size_t)std::max<int64_t>(N, 1)));
  TRY(Upload(d_ts, track_start, (size_t)num_tracks + 1, s)); TRY(Upload(d_ov, obs_view, (size_t)N, s)); TRY(Upload(d_vc, view_camera, (size_t)num_views, s));
  TRY(Upload(d_cm, camera_model, (size_t)num_cameras, s)); TRY(Upload(d_cs, cam_size, (size_t)2 * num_cameras, s)); TRY(Upload(d_l, lines, (size_t)3 * N, s));
  TRY(Upload(d_P, proj_matrices, (size_t)12 * num_views, s)); TRY(Upload(d_c, proj_centers, (size_t)3 * num_views, s)); TRY(Upload(d_in, intr, (size_t)kCamStride * num_cameras, s));
  a.T = num_tracks; a.track_start = d_ts; a.obs_view = d_ov; a.view_camera = d_vc; a.camera_model = d_cm; a.cam_size = d_cs; a.lines = d_l; a.P = d_P; a.centers = d_c; a.intr = d_in;
  a.min_tri_angle = o->min_tri_angle; a.max_residual = o->ransac.max_error * o->ransac.max_error; a.confidence = o->ransac.confidence;
  a.multiplier = o->ransac.dyn_num_trials_multiplier; a.residual_type = o->residual_type; a.min_num_trials = o->ransac.min_num_trials;
  {  // RANSAC ctor: cap max_num_trials from the a-priori inlier ratio (optim/ransac.h:149-155)
    const uint64_t cap = ComputeNumTrials((uint64_t)(o->ransac.min_inlier_ratio * 100000), 100000, o->ransac.confidence, o->ransac.dyn_num_trials_multiplier, 3);
    a.max_num_trials = std::min<uint64_t>(o->ransac.max_num_trials, cap);
  }
  a.success = d_s; a.mask = d_m; a.xyz = d_xyz; a.num_trials = d_nt;
  TRYH(hipEventRecord(ev0, s));
  hipLaunchKernelGGL(k_triangulate_tracks, dim3(CeilDiv(num_tracks, 64)), dim3(64), 0, s, a);
  TRYH(hipGetLastError());
  TRYH(hipEventRecord(ev1, s));
  TRY(Download(success, d_s, (size_t)num_tracks, s)); TRY(Download(xyz, d_xyz, (size_t)3 * num_tracks, s)); TRY(Download(inlier_mask, d_m, (size_t)N, s));
  TRY(Download(num_trials, d_nt, (size_t)num_tracks, s));
  TRYH(hipStreamSynchronize(s));
  if (device_ms) { float ms = 0; TRYH(hipEventElapsedTime(&ms, ev0, ev1)); *device_ms = ms; }
#undef TRY
#undef TRYH
  cleanup();
  return PP_OK;
} PP_API_CATCH("pp_triangulate_tracks")

// Observation / point filters that the mapper runs after every bundle adjustment, on the data
// a pp_ba_handle already holds (lines, indices, poses, points, intrinsics):
//   Reconstruction::FilterPoints3D = FilterPoints3DWithLargeReprojectionError + FilterPoints3DWithSmallTriangulationAngle
//                                           reference src/base/reconstruction.cc:425-439, 594-719
//   Reconstruction::FilterObservationsWithNegativeDepth                            :441-460
//   CalculateSquaredLineReprojectionError (pixel-space line error, cheirality and in-image gates)   src/base/projection.cc:153-203
//   CalculateTriangulationAngle src/base/triangulation.cc:59-82; ProjectionCenterFromPose src/base/pose.cc:94-101
// K7a k_filter_obs    one lane per observation: squared pixel error (DBL_MAX when gated) + depth flag      (HBM bound, K1-like)
// K7b k_filter_points one lane per point over its track (CSR by point): the deletion rules of the two point filters
// The reference deletes through Reconstruction::DeletePoint3D / DeleteObservation; here deletions come back as masks.
#include <cfloat>

#include "ba_impl.hpp"
#include "camera_models.hpp"

namespace ppsfm {

__device__ __forceinline__ void QuatToRotNormalized(const double* q_in, double R[9]) {   // QuaternionToRotationMatrix(NormalizeQuaternion(q))
  const double n = sqrt(q_in[0] * q_in[0] + q_in[1] * q_in[1] + q_in[2] * q_in[2] + q_in[3] * q_in[3]);
  const double w = q_in[0] / n, x = q_in[1] / n, y = q_in[2] / n, z = q_in[3] / n;
  R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - z * w); R[2] = 2 * (x * z + y * w);
  R[3] = 2 * (x * y + z * w); R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - x * w);
  R[6] = 2 * (x * z - y * w); R[7] = 2 * (y * z + x * w); R[8] = 1 - 2 * (x * x + y * y);
}

__global__ __launch_bounds__(256) void k_filter_obs(int64_t M, const double* __restrict__ la, const double* __restrict__ lb, const double* __restrict__ lc,
                                                    const int32_t* __restrict__ obs_pose, const int32_t* __restrict__ obs_point, const int32_t* __restrict__ obs_cam,
                                                    const double* __restrict__ poses, const double* __restrict__ points, const double* __restrict__ intr,
                                                    const int32_t* __restrict__ cam_size, double* __restrict__ err2, uint8_t* __restrict__ negative) {
  const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (o >= M) return;
  const int c = obs_pose[o], p = obs_point[o], ck = obs_cam[o];
  const int model = ck & 15, k = ck >> 4;
  const double* pose = poses + 7 * (size_t)c;
  const double X0 = points[3 * (size_t)p], X1 = points[3 * (size_t)p + 1], X2 = points[3 * (size_t)p + 2];
  double R[9];
  QuatToRotNormalized(pose, R);
  const double pz = R[6] * X0 + R[7] * X1 + R[8] * X2 + pose[6];
  if (negative) negative[o] = !(pz >= DBL_EPSILON);       // HasPointPositiveDepth
  if (!err2) return;
  double e = DBL_MAX;
  if (!(pz < DBL_EPSILON)) {
    const double px = R[0] * X0 + R[1] * X1 + R[2] * X2 + pose[4], py = R[3] * X0 + R[4] * X1 + R[5] * X2 + pose[5];
    const double inv = 1.0 / pz;
    const double u = inv * px, v = inv * py;
    const double a = la[o], b = lb[o];
    const double alpha = a * u + b * v + lc[o];
    const double lu = u - a * alpha, lv = v - b * alpha;
    const double* cam = intr + (size_t)kCamStride * k;
    double ix, iy;
    WorldToImage<double, double>(model, cam, u, v, &ix, &iy);
    if (ix >= 0 && ix < (double)cam_size[2 * k] && iy >= 0 && iy < (double)cam_size[2 * k + 1]) {
      double jx, jy;
      WorldToImage<double, double>(model, cam, lu, lv, &jx, &jy);
      e = (ix - jx) * (ix - jx) + (iy - jy) * (iy - jy);
    }
  }
  err2[o] = e;
}

__global__ __launch_bounds__(256) void k_proj_centers(int C, const double* __restrict__ poses, double* __restrict__ centers) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  double R[9];
  QuatToRotNormalized(poses + 7 * (size_t)c, R);
  const double* t = poses + 7 * (size_t)c + 4;
#pragma unroll
  for (int i = 0; i < 3; ++i) centers[3 * (size_t)c + i] = -(R[i] * t[0] + R[3 + i] * t[1] + R[6 + i] * t[2]);
}

__device__ __forceinline__ double TriangulationAngle(const double* c1, const double* c2, double X0, double X1, double X2) {
  const double b2 = (c1[0] - c2[0]) * (c1[0] - c2[0]) + (c1[1] - c2[1]) * (c1[1] - c2[1]) + (c1[2] - c2[2]) * (c1[2] - c2[2]);
  const double r1 = (X0 - c1[0]) * (X0 - c1[0]) + (X1 - c1[1]) * (X1 - c1[1]) + (X2 - c1[2]) * (X2 - c1[2]);
  const double r2 = (X0 - c2[0]) * (X0 - c2[0]) + (X1 - c2[1]) * (X1 - c2[1]) + (X2 - c2[2]) * (X2 - c2[2]);
  const double den = 2.0 * sqrt(r1 * r2);
  if (den == 0.0) return 0.0;
  const double ang = fabs(acos((r1 + r2 - b2) / den));
  return fmin(ang, 3.14159265358979323846 - ang);
}

// per point: both point filters over its track (the observations of the point in problem order)
__global__ __launch_bounds__(256) void k_filter_points(int P, const int32_t* __restrict__ pt_start, const int32_t* __restrict__ pt_obs, const int32_t* __restrict__ obs_pose,
                                                       const uint8_t* __restrict__ obs_aligned, const double* __restrict__ err2, const double* __restrict__ centers,
                                                       const double* __restrict__ points, const uint8_t* __restrict__ subset, double max2, double min_rad,
                                                       uint8_t* __restrict__ obs_deleted, uint8_t* __restrict__ point_deleted, double* __restrict__ point_error,
                                                       unsigned long long* __restrict__ counters) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= P) return;
  const int e0 = pt_start[p], e1 = pt_start[p + 1], len = e1 - e0;
  point_deleted[p] = 0; point_error[p] = -1.0;
  if (subset && !subset[p]) return;
  auto delete_point = [&](unsigned long long filtered) {
    point_deleted[p] = 1;
    for (int e = e0; e < e1; ++e) obs_deleted[pt_obs[e]] = 1;
    atomicAdd(&counters[0], filtered);
    atomicAdd(&counters[1], 1ull);
  };
  bool non_aligned = false;
  for (int e = e0; e < e1; ++e) non_aligned = non_aligned || !obs_aligned[pt_obs[e]];
  if (!non_aligned || len < 3) { delete_point((unsigned long long)len); return; }       // reconstruction.cc:673-689
  double sum = 0.0;
  int ndel = 0;
  for (int e = e0; e < e1; ++e) {
    const double v = err2[pt_obs[e]];
    if (v > max2) ++ndel; else sum += sqrt(v);
  }
  if (ndel >= len - 3) { delete_point((unsigned long long)len); return; }                 // :705-707 (a track of exactly 3 never survives)
  for (int e = e0; e < e1; ++e) if (err2[pt_obs[e]] > max2) obs_deleted[pt_obs[e]] = 1;
  atomicAdd(&counters[0], (unsigned long long)ndel);
  point_error[p] = sum / (double)(len - ndel);
  const double X0 = points[3 * (size_t)p], X1 = points[3 * (size_t)p + 1], X2 = points[3 * (size_t)p + 2];
  bool keep = false;
  for (int i1 = e0; i1 < e1 && !keep; ++i1) {
    const int o1 = pt_obs[i1];
    if (err2[o1] > max2) continue;
    for (int i2 = e0; i2 < i1; ++i2) {
      const int o2 = pt_obs[i2];
      if (err2[o2] > max2) continue;
      if (TriangulationAngle(centers + 3 * (size_t)obs_pose[o1], centers + 3 * (size_t)obs_pose[o2], X0, X1, X2) >= min_rad) { keep = true; break; }
    }
  }
  if (!keep) { delete_point(1ull); }                                                      // :649-652 counts the point once
}

}  // namespace ppsfm

using namespace ppsfm;

extern "C" {

int pp_ba_filter_points(pp_ba_handle h, const pp_filter_options* o, const uint8_t* obs_aligned, const int32_t* cam_size, const uint8_t* point_subset,
                        uint8_t* obs_deleted, uint8_t* point_deleted, double* point_error, pp_filter_report* rep) try {
  PP_REQUIRE(h && o && cam_size && obs_deleted && point_deleted && point_error && rep, "pp_ba_filter_points: null argument");
  PP_REQUIRE(o->max_reproj_error >= 0 && o->min_tri_angle_deg >= 0, "pp_ba_filter_points: bad options");
  PP_HIP_TRY(hipSetDevice(h->device));
  hipStream_t s = h->stream;
  const int64_t M = h->M;
  const int P = h->P, C = h->C, K = h->K;
  double *err2 = nullptr, *centers = nullptr, *perr = nullptr;
  uint8_t *d_al = nullptr, *d_sub = nullptr, *d_od = nullptr, *d_pd = nullptr;
  int32_t* d_cs = nullptr;
  unsigned long long* d_cnt = nullptr;
  int rc = PP_OK;
  auto cleanup = [&]() { void* b[] = {err2, centers, perr, d_al, d_sub, d_od, d_pd, d_cs, d_cnt}; for (void* p : b) if (p) (void)hipFree(p); };
  OnUnwind unwind{[&] { cleanup(); }};
#define TRY(x) do { rc = (x); if (rc) { cleanup(); return rc; } } while (0)
#define TRYH(x) do { if ((x) != hipSuccess) { SetLastError("pp_ba_filter_points: %s failed", #x); cleanup(); return PP_ERR_HIP; } } while (0)
  TRY(DeviceAlloc(&err2, (size_t)M)); TRY(DeviceAlloc(&centers, (size_t)3 * C)); TRY(DeviceAlloc(&perr, (size_t)P));
  TRY(DeviceAlloc(&d_al, (size_t)M)); TRY(DeviceAlloc(&d_od, (size_t)M)); TRY(DeviceAlloc(&d_pd, (size_t)P)); TRY(DeviceAlloc(&d_cs, (size_t)2 * K));
  TRY(DeviceAlloc(&d_cnt, 2));
  std::vector<uint8_t> al(M, 0);
  if (obs_aligned) al.assign(obs_aligned, obs_aligned + M);
  TRY(Upload(d_al, al.data(), (size_t)M, s)); TRY(Upload(d_cs, cam_size, (size_t)2 * K, s));
  if (point_subset) { TRY(DeviceAlloc(&d_sub, (size_t)P)); TRY(Upload(d_sub, point_subset, (size_t)P, s)); }
  TRYH(hipMemsetAsync(d_od, 0, (size_t)M, s)); TRYH(hipMemsetAsync(d_cnt, 0, 2 * sizeof(unsigned long long), s));
  hipLaunchKernelGGL(k_filter_obs, dim3(CeilDiv(M, 256)), dim3(256), 0, s, M, h->la, h->lb, h->lc, h->obs_pose, h->obs_point, h->obs_cam, h->poses, h->points, h->intr, d_cs,
                     err2, (uint8_t*)nullptr);
  hipLaunchKernelGGL(k_proj_centers, dim3(CeilDiv(C, 256)), dim3(256), 0, s, C, h->poses, centers);
  hipLaunchKernelGGL(k_filter_points, dim3(CeilDiv(P, 256)), dim3(256), 0, s, P, h->pt_start, h->pt_obs, h->obs_pose, d_al, err2, centers, h->points, d_sub,
                     o->max_reproj_error * o->max_reproj_error, o->min_tri_angle_deg * 3.14159265358979323846 / 180.0, d_od, d_pd, perr, d_cnt);
  TRYH(hipGetLastError());
  unsigned long long cnt[2] = {0, 0};
  TRY(Download(obs_deleted, d_od, (size_t)M, s)); TRY(Download(point_deleted, d_pd, (size_t)P, s)); TRY(Download(point_error, perr, (size_t)P, s));
  TRY(Download(cnt, d_cnt, 2, s));
  TRYH(hipStreamSynchronize(s));
#undef TRY
#undef TRYH
  cleanup();
  rep->num_filtered = (int64_t)cnt[0];
  rep->num_points_deleted = (int64_t)cnt[1];
  rep->num_observations_deleted = 0;
  for (int64_t i = 0; i < M; ++i) rep->num_observations_deleted += obs_deleted[i];
  return PP_OK;
} PP_API_CATCH("pp_ba_filter_points")

int pp_ba_filter_negative_depth(pp_ba_handle h, uint8_t* obs_negative, int64_t* num_filtered) try {
  PP_REQUIRE(h && obs_negative && num_filtered, "pp_ba_filter_negative_depth: null argument");
  PP_HIP_TRY(hipSetDevice(h->device));
  uint8_t* d = nullptr;
  int rc = DeviceAlloc(&d, (size_t)h->M); if (rc) return rc;
  hipLaunchKernelGGL(k_filter_obs, dim3(CeilDiv(h->M, 256)), dim3(256), 0, h->stream, h->M, h->la, h->lb, h->lc, h->obs_pose, h->obs_point, h->obs_cam, h->poses, h->points,
                     h->intr, (const int32_t*)nullptr, (double*)nullptr, d);
  if (hipGetLastError() != hipSuccess) { (void)hipFree(d); SetLastError("pp_ba_filter_negative_depth: launch failed"); return PP_ERR_HIP; }
  rc = Download(obs_negative, d, (size_t)h->M, h->stream);
  if (!rc && hipStreamSynchronize(h->stream) != hipSuccess) rc = PP_ERR_HIP;
  (void)hipFree(d);
  if (rc) return rc;
  *num_filtered = 0;
  for (int64_t i = 0; i < h->M; ++i) *num_filtered += obs_negative[i];
  return PP_OK;
} PP_API_CATCH("pp_ba_filter_negative_depth")

}  // extern "C"

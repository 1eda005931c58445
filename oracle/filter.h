// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// CPU restatement of the observation / point filters that run after every bundle adjustment:
//   CalculateSquaredLineReprojectionError   reference src/base/projection.cc:153-203
//   HasPointPositiveDepth                   src/base/projection.cc:280-284
//   CalculateTriangulationAngle             src/base/triangulation.cc:59-82
//   ProjectionCenterFromPose                src/base/pose.cc:94-101
//   Reconstruction::FilterPoints3D = FilterPoints3DWithLargeReprojectionError + ...SmallTriangulationAngle,
//   FilterObservationsWithNegativeDepth     src/base/reconstruction.cc:425-460, 594-719
// on the flat problem form (observations grouped per point = the track); deletions are reported as masks.
#pragma once
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <vector>

#include "camera_models.h"

namespace oracle {

inline void QuatToRot(const double q_in[4], double R[9]) {   // QuaternionToRotationMatrix of the NORMALISED quaternion (base/pose.cc)
  const double n = std::sqrt(q_in[0] * q_in[0] + q_in[1] * q_in[1] + q_in[2] * q_in[2] + q_in[3] * q_in[3]);
  const double w = q_in[0] / n, x = q_in[1] / n, y = q_in[2] / n, z = q_in[3] / n;
  R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - z * w); R[2] = 2 * (x * z + y * w);
  R[3] = 2 * (x * y + z * w); R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - x * w);
  R[6] = 2 * (x * z - y * w); R[7] = 2 * (y * z + x * w); R[8] = 1 - 2 * (x * x + y * y);
}

// projection.cc:161-203; DBL_MAX behind the camera or outside the image
inline double SquaredLineReprojectionErrorPx(const double l[3], const double X[3], const double pose[7], int model, const double* params, double width,
                                             double height) {
  double R[9]; QuatToRot(pose, R);
  const double pz = R[6] * X[0] + R[7] * X[1] + R[8] * X[2] + pose[6];
  if (pz < DBL_EPSILON) return DBL_MAX;
  const double px = R[0] * X[0] + R[1] * X[1] + R[2] * X[2] + pose[4], py = R[3] * X[0] + R[4] * X[1] + R[5] * X[2] + pose[5];
  const double inv = 1.0 / pz;
  const double u = inv * px, v = inv * py;
  const double alpha = l[0] * u + l[1] * v + l[2];
  const double lu = u - l[0] * alpha, lv = v - l[1] * alpha;
  double ix, iy, jx, jy;
  WorldToImage<double>(model, params, u, v, &ix, &iy);
  if (!(ix >= 0 && ix < width && iy >= 0 && iy < height)) return DBL_MAX;
  WorldToImage<double>(model, params, lu, lv, &jx, &jy);
  return (ix - jx) * (ix - jx) + (iy - jy) * (iy - jy);
}

inline double TriangulationAngle(const double c1[3], const double c2[3], const double X[3]) {
  double b2 = 0, r1 = 0, r2 = 0;
  for (int i = 0; i < 3; ++i) { b2 += (c1[i] - c2[i]) * (c1[i] - c2[i]); r1 += (X[i] - c1[i]) * (X[i] - c1[i]); r2 += (X[i] - c2[i]) * (X[i] - c2[i]); }
  const double den = 2.0 * std::sqrt(r1 * r2);
  if (den == 0.0) return 0.0;
  const double ang = std::fabs(std::acos((r1 + r2 - b2) / den));
  return std::fmin(ang, M_PI - ang);
}

struct FilterResult { int64_t num_filtered = 0; };

// M observations; track of point p = its observations in input order.  point_subset may be null (all points).
inline int64_t FilterPoints3D(int64_t M, int P, int C, const double* lines, const int32_t* obs_pose, const int32_t* obs_point, const uint8_t* obs_aligned,
                              const int32_t* pose_camera, const int32_t* camera_model, const int32_t* cam_size /*K x 2*/, const double* poses, const double* points,
                              const double* intr, int cam_stride, double max_reproj_error, double min_tri_angle_deg, const uint8_t* point_subset,
                              uint8_t* obs_deleted, uint8_t* point_deleted, double* point_error) {
  std::vector<std::vector<int64_t>> track(P);
  for (int64_t o = 0; o < M; ++o) track[obs_point[o]].push_back(o);
  for (int64_t o = 0; o < M; ++o) obs_deleted[o] = 0;
  std::vector<double> centers(3 * (size_t)C);
  for (int c = 0; c < C; ++c) {   // ProjectionCenterFromPose: -R^T t
    double R[9]; QuatToRot(poses + 7 * c, R);
    const double* t = poses + 7 * c + 4;
    for (int i = 0; i < 3; ++i) centers[3 * c + i] = -(R[i] * t[0] + R[3 + i] * t[1] + R[6 + i] * t[2]);
  }
  const double max2 = max_reproj_error * max_reproj_error, min_rad = min_tri_angle_deg * M_PI / 180.0;
  int64_t num_filtered = 0;
  for (int p = 0; p < P; ++p) {
    point_deleted[p] = 0; point_error[p] = -1.0;
    if (point_subset && !point_subset[p]) continue;
    const std::vector<int64_t>& tr = track[p];
    const size_t len = tr.size();
    // FilterPoints3DWithLargeReprojectionError (:658-719)
    bool non_aligned = false;
    for (int64_t o : tr) if (!obs_aligned[o]) non_aligned = true;
    if (!non_aligned || len < 3) { point_deleted[p] = 1; num_filtered += (int64_t)len; for (int64_t o : tr) obs_deleted[o] = 1; continue; }
    double sum = 0; size_t ndel = 0;
    std::vector<char> del(len, 0);
    for (size_t e = 0; e < len; ++e) {
      const int64_t o = tr[e]; const int c = obs_pose[o], k = pose_camera[c];
      const double e2 = SquaredLineReprojectionErrorPx(lines + 3 * o, points + 3 * p, poses + 7 * c, camera_model[k], intr + (size_t)cam_stride * k, cam_size[2 * k],
                                                       cam_size[2 * k + 1]);
      if (e2 > max2) { del[e] = 1; ++ndel; } else sum += std::sqrt(e2);
    }
    if (ndel >= len - 3) { point_deleted[p] = 1; num_filtered += (int64_t)len; for (int64_t o : tr) obs_deleted[o] = 1; continue; }
    num_filtered += (int64_t)ndel;
    for (size_t e = 0; e < len; ++e) if (del[e]) obs_deleted[tr[e]] = 1;
    point_error[p] = sum / (double)(len - ndel);
    // FilterPoints3DWithSmallTriangulationAngle (:594-656) on the remaining track
    bool keep = false;
    for (size_t i1 = 0; i1 < len && !keep; ++i1) {
      if (del[i1]) continue;
      for (size_t i2 = 0; i2 < i1; ++i2) {
        if (del[i2]) continue;
        if (TriangulationAngle(&centers[3 * obs_pose[tr[i1]]], &centers[3 * obs_pose[tr[i2]]], points + 3 * p) >= min_rad) { keep = true; break; }
      }
    }
    if (!keep) { point_deleted[p] = 1; num_filtered += 1; for (int64_t o : tr) obs_deleted[o] = 1; }
  }
  return num_filtered;
}

inline int64_t FilterObservationsWithNegativeDepth(int64_t M, const int32_t* obs_pose, const int32_t* obs_point, const double* poses, const double* points,
                                                   uint8_t* obs_negative) {
  int64_t n = 0;
  for (int64_t o = 0; o < M; ++o) {
    const double* pose = poses + 7 * obs_pose[o]; const double* X = points + 3 * obs_point[o];
    double R[9]; QuatToRot(pose, R);
    const double z = R[6] * X[0] + R[7] * X[1] + R[8] * X[2] + pose[6];
    obs_negative[o] = !(z >= DBL_EPSILON);
    n += obs_negative[o];
  }
  return n;
}

}  // namespace oracle

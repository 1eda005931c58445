// K4 / K5 — absolute pose from six line/point pairs: batched minimal solver, batched scoring, and
// the RANSAC driver that keeps the reference's sequential semantics.
//
//   RANSAC<P6LEstimator>::Estimate              reference src/optim/ransac.h:178-278
//   P6LEstimator::Residuals -> ComputeSquaredLineReprojectionError   src/estimators/utils.cc:40-89
//   InlierSupportMeasurer::{Evaluate, Compare}  src/optim/support_measurement.cc:36-60
//   EstimateAbsolutePoseFromLines               src/estimators/pose.cc:52-94 (host mirror: ppsfm/pose.hpp)
//
// K4 (scoring) roofline: fp64 VALU.  The correspondences (N x 48 B, SoA) are shared by every
// hypothesis: the valid models of all hypotheses are flattened into one list, a workgroup of eight
// wavefronts stages 512-correspondence tiles in LDS (double-buffered) and every wavefront scores two
// models held in scalar registers against the tile, so a correspondence is read from L2 once per 16
// model evaluations; inliers are counted with ballot+popcount, the residual sum with a fixed
// butterfly.  ~30 flop + 1 IEEE division per (model, correspondence).
// The per-correspondence arithmetic reproduces estimators/utils.cc:70-84 operation by operation with
// FMA contraction disabled, so `r <= max_residual` decides bit-identically to the CPU reference.
#include <algorithm>
#include <cstdlib>
#include <chrono>
#include <cfloat>
#include <cstring>
#include <vector>

#include "common.hpp"
#include "p6l_device.hpp"
#include "ransac_host.hpp"

struct pp_pose_impl {
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  int32_t n = 0;
  double *l0 = nullptr, *l1 = nullptr, *l2 = nullptr, *x0 = nullptr, *x1 = nullptr, *x2 = nullptr;
  uint8_t* aligned = nullptr;
  // work buffers sized for `cap_hyp` hypotheses
  int64_t cap_hyp = 0;
  uint32_t* samples = nullptr;
  double* models = nullptr;       // cap_hyp x 8 x 12
  int32_t* num_models = nullptr;  // cap_hyp
  uint32_t* inliers = nullptr;    // cap_hyp x 8
  double* sums = nullptr;         // cap_hyp x 8
  double* residuals = nullptr;    // n (single model) — grown on demand
  int64_t cap_res = 0;
  unsigned long long* best_key = nullptr;  // per-block best candidates
  int32_t* flat = nullptr;                 // cap_hyp x 8 flat model slots
  int32_t* flat_total = nullptr;
  int32_t* flat_blocks = nullptr;          // 1024 per-block sums / offsets of the flat list
  std::vector<uint32_t> h_samples;
  uint32_t h_samples_seed = 0;
  // pinned host mirrors of one RANSAC chunk (pp_pose_ransac reads every chunk back: pageable copies were ~0.4 ms per chunk)
  int64_t cap_pin = 0;
  uint32_t *pin_samples = nullptr, *pin_inl = nullptr;
  int32_t* pin_nm = nullptr;
  double *pin_sm = nullptr, *pin_mdl = nullptr;
  int64_t last_hyp = 0;     // hypotheses of the last pp_pose_hypotheses call (their scores are still on the device)
};

namespace ppsfm {


struct CorrData {
  const double *l0, *l1, *l2, *x0, *x1, *x2;
  int32_t n;
};

// 1.0 / pz, correctly rounded, for DBL_EPSILON < pz < 2^1000.  The compiler's IEEE fp64 division is
// v_div_scale x2, v_rcp, 6 fma/mul, v_div_fmas, v_div_fixup; for a numerator of 1.0 and a denominator in that
// range both v_div_scale are the identity, v_div_fmas is a plain fma and v_div_fixup passes its argument
// through, so the same Newton steps on the unscaled operands give the same bits with 7 instructions instead of
// 12 (v_rcp_f64 issues at a third of the fma rate, tools/valu_rate_bench.hip).  Lanes with pz <= DBL_EPSILON or
// NaN may return anything: the scoring kernel discards their value.  k_score_flat proves pz < 2^1000 from
// bounds on the model row and on the points (kDepthRowBound, kPointBound) and takes the full division otherwise.
__device__ __forceinline__ double ReciprocalOfDepth(double pz) {
  double r = __builtin_amdgcn_rcp(pz);
  r = __builtin_fma(r, __builtin_fma(-pz, r, 1.0), r);
  r = __builtin_fma(r, __builtin_fma(-pz, r, 1.0), r);
  return __builtin_fma(__builtin_fma(-pz, r, 1.0), r, r);
}
constexpr double kDepthRowBound = 0x1p900;   // |P[8..11]| <= 2^900 and |X| <= 2^90  =>  |pz| <= 2^992
constexpr double kPointBound = 0x1p90;

// the arithmetic of estimators/utils.cc:70-84 in the reference's association, no contraction
template <bool kFullDivision>
__device__ __forceinline__ void LineErrorTerms(const double* __restrict__ P, double X0, double X1, double X2, double L0, double L1,
                                               double L2, double* pz_out, double* sq_out) {
#pragma clang fp contract(off)
  const double pz = P[8] * X0 + P[9] * X1 + P[10] * X2 + P[11];
  const double px = P[0] * X0 + P[1] * X1 + P[2] * X2 + P[3];
  const double py = P[4] * X0 + P[5] * X1 + P[6] * X2 + P[7];
  const double inv = kFullDivision ? 1.0 / pz : ReciprocalOfDepth(pz);
  const double res = px * L0 * inv + py * L1 * inv + L2;
  *pz_out = pz;
  *sq_out = res * res;
}

__device__ __forceinline__ double ReadLaneF64(double v, int src_lane) {     // src_lane wave-uniform
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, src_lane);
  hi = __builtin_amdgcn_readlane(hi, src_lane);
  return __hiloint2double(hi, lo);
}

// estimators/utils.cc:70-84, exact association; returns DBL_MAX behind the camera
__device__ __forceinline__ double SquaredLineError(const double* __restrict__ P, double X0, double X1, double X2, double L0,
                                                   double L1, double L2) {
  double pz, sq;
  LineErrorTerms<true>(P, X0, X1, X2, L0, L1, L2, &pz, &sq);
  return (pz > DBL_EPSILON) ? sq : DBL_MAX;
}

// ---- K4: flat model list + LDS-tiled correspondences ----------------------------------------------
// exclusive scan of num_models -> flat list of model slots (h*8+s), total in *total_out.  Three small launches: per-block
// counts (block b owns the hypotheses [b, b+1) * span), a one-block scan of those, then every block scans its own span again
// and writes its slots (one block walking a million hypotheses took 2.1 ms).
constexpr int kFlattenThreads = 1024;
__device__ __forceinline__ int FlattenLocalScan(int64_t h0, int64_t h1, const int32_t* __restrict__ num_models, int* part, int* local_out) {
  const int tid = threadIdx.x;
  const int64_t chunk = (h1 - h0 + kFlattenThreads - 1) / kFlattenThreads;
  const int64_t t0 = h0 + tid * chunk, t1 = (t0 + chunk < h1) ? t0 + chunk : h1;
  int local = 0;
  for (int64_t h = t0; h < t1; ++h) local += num_models[h];
  part[tid] = local;
  __syncthreads();
  for (int off = 1; off < kFlattenThreads; off <<= 1) {   // Hillis-Steele inclusive scan
    const int v = (tid >= off) ? part[tid - off] : 0;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  *local_out = local;
  return part[tid] - local;          // exclusive prefix of this thread within the block
}
__global__ __launch_bounds__(kFlattenThreads) void k_flatten_count(int64_t num_hyp, int64_t span, const int32_t* __restrict__ num_models, int32_t* __restrict__ block_sums) {
  __shared__ int part[kFlattenThreads];
  const int64_t h0 = blockIdx.x * span, h1 = (h0 + span < num_hyp) ? h0 + span : num_hyp;
  int local;
  FlattenLocalScan(h0, h1 > h0 ? h1 : h0, num_models, part, &local);
  if (threadIdx.x == kFlattenThreads - 1) block_sums[blockIdx.x] = part[kFlattenThreads - 1];
}
__global__ __launch_bounds__(kFlattenThreads) void k_flatten_offsets(int nblocks, int32_t* __restrict__ block_sums, int32_t* __restrict__ total_out) {
  __shared__ int part[kFlattenThreads];
  const int tid = threadIdx.x;
  const int v0 = tid < nblocks ? block_sums[tid] : 0;
  part[tid] = v0;
  __syncthreads();
  for (int off = 1; off < kFlattenThreads; off <<= 1) {
    const int v = (tid >= off) ? part[tid - off] : 0;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  if (tid < nblocks) block_sums[tid] = part[tid] - v0;     // exclusive
  if (tid == kFlattenThreads - 1) *total_out = part[kFlattenThreads - 1];
}
__global__ __launch_bounds__(kFlattenThreads) void k_flatten_fill(int64_t num_hyp, int64_t span, const int32_t* __restrict__ num_models,
                                                                  const int32_t* __restrict__ block_offsets, int32_t* __restrict__ flat) {
  __shared__ int part[kFlattenThreads];
  const int64_t h0 = blockIdx.x * span, h1 = (h0 + span < num_hyp) ? h0 + span : num_hyp;
  int local;
  int pos = block_offsets[blockIdx.x] + FlattenLocalScan(h0, h1 > h0 ? h1 : h0, num_models, part, &local);
  const int64_t chunk = ((h1 > h0 ? h1 : h0) - h0 + kFlattenThreads - 1) / kFlattenThreads;
  const int64_t t0 = h0 + threadIdx.x * chunk, t1 = (t0 + chunk < h1) ? t0 + chunk : h1;
  for (int64_t h = t0; h < t1; ++h) {
    const int nm = num_models[h];
    for (int sidx = 0; sidx < nm; ++sidx) flat[pos++] = (int32_t)(h * 8 + sidx);
  }
}
// the identity list (models stored contiguously) — pp_pose_score
__global__ __launch_bounds__(256) void k_flat_identity(int num, int32_t* __restrict__ flat, int32_t* __restrict__ total_out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < num) flat[i] = i;
  if (i == 0) *total_out = num;
}

constexpr int kTile = 512;       // correspondences per LDS tile (= workgroup size)
constexpr int kFlatMPW = 2;      // models per wavefront, held in scalar registers

// acc += r on the lanes of `mask` only: the addition runs under a narrowed EXEC mask instead of selecting
// (r or 0) first (a 64-bit select is two v_cndmask_b32).
__device__ __forceinline__ void MaskedAdd(double& acc, double r, unsigned long long mask) {
  unsigned long long saved;
  asm volatile("s_and_saveexec_b64 %0, %3\n\tv_add_f64 %1, %1, %2\n\ts_mov_b64 exec, %0"
               : "=&s"(saved), "+v"(acc)
               : "v"(r), "s"(mask)
               : "scc");   // s_and_saveexec writes SCC
}

// 64 correspondences (one per lane) of an LDS tile against the kFlatMPW models of this wavefront.
// kFast (max_residual < DBL_MAX, i.e. a point behind the camera is never an inlier): the DBL_MAX select of
// SquaredLineError becomes part of the inlier predicate, which is identical for every input including NaN;
// the predicate lives in a scalar register pair, the count is its population count (scalar unit) and the
// residual sum is accumulated under EXEC = predicate.  The arithmetic producing `sq` is SquaredLineError's.
template <bool kFast, bool kCheckValid, bool kFullDivision>
__device__ __forceinline__ void ScoreSubTile(const double (*__restrict__ tile)[kTile], int li, bool valid,
                                             const double (&P)[kFlatMPW][12], double max_residual, uint32_t (&cnt)[kFlatMPW],
                                             double (&acc)[kFlatMPW]) {
  const double X0 = tile[0][li], X1 = tile[1][li], X2 = tile[2][li];
  const double L0 = tile[3][li], L1 = tile[4][li], L2 = tile[5][li];
#pragma unroll
  for (int j = 0; j < kFlatMPW; ++j) {
    if (kFast) {
      double pz, sq;
      LineErrorTerms<kFullDivision>(P[j], X0, X1, X2, L0, L1, L2, &pz, &sq);
      // one ballot per comparison, combined as 64-bit scalars (a ballot of the combined bool goes through a
      // v_cndmask/v_cmp_ne pair)
      unsigned long long mask = __builtin_amdgcn_ballot_w64(pz > DBL_EPSILON) & __builtin_amdgcn_ballot_w64(sq <= max_residual);
      if (kCheckValid) mask &= __builtin_amdgcn_ballot_w64(valid);
      cnt[j] += (uint32_t)__popcll(mask);     // wave-uniform: stays on the scalar unit
      MaskedAdd(acc[j], sq, mask);
    } else {
      const double r = SquaredLineError(P[j], X0, X1, X2, L0, L1, L2);
      const bool in = (!kCheckValid || valid) && (r <= max_residual);
      cnt[j] += in ? 1u : 0u;
      acc[j] += in ? r : 0.0;
    }
  }
}

template <bool kFast, bool kFullDivision>
__device__ __forceinline__ void ScoreTile(const double (*__restrict__ tile)[kTile], int lane, int base, int n,
                                          const double (&P)[kFlatMPW][12], double max_residual, uint32_t (&cnt)[kFlatMPW],
                                          double (&acc)[kFlatMPW]) {
  if (base + kTile <= n) {          // full tile: no bounds predicate
#pragma unroll 2
    for (int sub = 0; sub < kTile / 64; ++sub)
      ScoreSubTile<kFast, false, kFullDivision>(tile, sub * 64 + lane, true, P, max_residual, cnt, acc);
  } else {
#pragma unroll 2
    for (int sub = 0; sub < kTile / 64; ++sub) {
      const int li = sub * 64 + lane;
      ScoreSubTile<kFast, true, kFullDivision>(tile, li, base + li < n, P, max_residual, cnt, acc);
    }
  }
}

// One wavefront scores kFlatMPW consecutive entries of the flat model list; the 8 wavefronts of a
// workgroup walk the correspondences together, tile by tile, through a double-buffered LDS tile
// (6 SoA streams x 512 doubles = 24 KB per buffer), so every 48-byte correspondence is fetched from
// L2 once per 16 models.  Models live in SGPRs (wave-uniform), the per-correspondence arithmetic is
// SquaredLineError (bit-exact).  kFast: see ScoreSubTile and ReciprocalOfDepth; the general variant counts by
// lane accumulation + butterfly and is only launched when max_residual >= DBL_MAX.  Lane l accumulates the
// correspondences l, l+64, l+128, ... in index order, then WaveSum: the order of the residual sum is fixed.
template <bool kFast>
__global__ __launch_bounds__(512) void k_score_flat(CorrData d, const int32_t* __restrict__ flat, const int32_t* __restrict__ total_ptr,
                                                    const double* __restrict__ models, double max_residual,
                                                    uint32_t* __restrict__ inliers, double* __restrict__ sums) {
  __shared__ double tile[2][6][kTile];
  __shared__ int large_point;      // set (never cleared) once a staged point exceeds kPointBound
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int total = *total_ptr;
  const int m_base = (blockIdx.x * 8 + wave) * kFlatMPW;
  if (blockIdx.x * 8 * kFlatMPW >= total) return;     // whole workgroup idle (uniform)
  double P[kFlatMPW][12];
  int slot[kFlatMPW];
  bool large_row = false;          // wave-uniform: a depth row entry above kDepthRowBound (or NaN)
#pragma unroll
  for (int j = 0; j < kFlatMPW; ++j) {
    const int m = (m_base + j < total) ? m_base + j : total - 1;
    slot[j] = __builtin_amdgcn_readfirstlane(flat[m]);
    const double* src = models + (size_t)slot[j] * 12;
#pragma unroll
    for (int e = 0; e < 12; ++e) P[j][e] = src[e];
#pragma unroll
    for (int e = 8; e < 12; ++e) large_row = large_row || !(fabs(P[j][e]) <= kDepthRowBound);
  }
  uint32_t cnt[kFlatMPW];
  double acc[kFlatMPW];
#pragma unroll
  for (int j = 0; j < kFlatMPW; ++j) { cnt[j] = 0; acc[j] = 0.0; }
  const int n = d.n;
  const int ntiles = (n + kTile - 1) / kTile;
  double pre[6];
  {
    const int i = tid;
    const bool ok = i < n;
    pre[0] = ok ? d.x0[i] : 0.0; pre[1] = ok ? d.x1[i] : 0.0; pre[2] = ok ? d.x2[i] : 0.0;
    pre[3] = ok ? d.l0[i] : 0.0; pre[4] = ok ? d.l1[i] : 0.0; pre[5] = ok ? d.l2[i] : 0.0;
  }
  if (tid == 0) large_point = 0;
  __syncthreads();
#pragma unroll
  for (int c = 0; c < 6; ++c) tile[0][c][tid] = pre[c];
  if (!(fabs(pre[0]) <= kPointBound && fabs(pre[1]) <= kPointBound && fabs(pre[2]) <= kPointBound)) large_point = 1;
  __syncthreads();
  for (int t = 0; t < ntiles; ++t) {
    const int buf = t & 1;
    if (t + 1 < ntiles) {   // prefetch the next tile into registers while this one is scored
      const int i = (t + 1) * kTile + tid;
      const bool ok = i < n;
      pre[0] = ok ? d.x0[i] : 0.0; pre[1] = ok ? d.x1[i] : 0.0; pre[2] = ok ? d.x2[i] : 0.0;
      pre[3] = ok ? d.l0[i] : 0.0; pre[4] = ok ? d.l1[i] : 0.0; pre[5] = ok ? d.l2[i] : 0.0;
    }
    const bool full_division = !kFast || large_row || __builtin_amdgcn_readfirstlane(large_point) != 0;
    if (full_division) ScoreTile<kFast, true>(tile[buf], lane, t * kTile, n, P, max_residual, cnt, acc);
    else ScoreTile<kFast, false>(tile[buf], lane, t * kTile, n, P, max_residual, cnt, acc);
    if (t + 1 < ntiles) {
#pragma unroll
      for (int c = 0; c < 6; ++c) tile[buf ^ 1][c][tid] = pre[c];
      if (!(fabs(pre[0]) <= kPointBound && fabs(pre[1]) <= kPointBound && fabs(pre[2]) <= kPointBound)) large_point = 1;
    }
    __syncthreads();
  }
#pragma unroll
  for (int j = 0; j < kFlatMPW; ++j) {
    uint32_t c = cnt[j];
    if (!kFast) {
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) c += __shfl_xor((int)c, off, 64);
    }
    const double sm = WaveSum(acc[j]);
    if (lane == 0 && m_base + j < total) { inliers[slot[j]] = c; sums[slot[j]] = sm; }
  }
}

// full residual vectors: one lane per (model, correspondence)
__global__ __launch_bounds__(256) void k_residuals(CorrData d, int num, const double* __restrict__ models, double* __restrict__ out) {
  const int m = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (m >= num || i >= d.n) return;
  double P[12];
#pragma unroll
  for (int e = 0; e < 12; ++e) P[e] = models[(size_t)m * 12 + e];
  out[(size_t)m * d.n + i] = SquaredLineError(P, d.x0[i], d.x1[i], d.x2[i], d.l0[i], d.l1[i], d.l2[i]);
}

// exact sequential-order support (support_measurement.cc:40-47): the fp64 additions happen in the reference's order.
// One WAVEFRONT per model: the 64 lanes compute the residuals of 64 consecutive correspondences in parallel, then the
// inlier residuals are added one by one in index order (v_readlane, wave-uniform accumulator) — the order of the sum is the
// sequential loop's, its cost is not (one lane recomputing a division per step took 0.1 us per correspondence).
__global__ __launch_bounds__(256) void k_support_sequential(CorrData d, int num, const double* __restrict__ models, double max_residual,
                                                            uint32_t* __restrict__ inliers, double* __restrict__ sums) {
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= num) return;
  double P[12];
#pragma unroll
  for (int e = 0; e < 12; ++e) P[e] = models[(size_t)m * 12 + e];
  uint32_t c = 0;
  double acc = 0.0;
  for (int base = 0; base < d.n; base += 64) {
    const int i = base + lane;
    double r = DBL_MAX;
    bool in = false;
    if (i < d.n) { r = SquaredLineError(P, d.x0[i], d.x1[i], d.x2[i], d.l0[i], d.l1[i], d.l2[i]); in = r <= max_residual; }
    unsigned long long mask = __builtin_amdgcn_ballot_w64(in);
    c += (uint32_t)__popcll(mask);
    while (mask) {                       // wave-uniform loop over the inliers of this group, in index order
      const int l = __builtin_ctzll(mask);
      mask &= mask - 1;
      acc += ReadLaneF64(r, l);
    }
  }
  if (lane == 0) { inliers[m] = c; sums[m] = acc; }
}

// K5: one lane per hypothesis
__global__ __launch_bounds__(64) void k_p6l(CorrData d, const uint8_t* __restrict__ aligned, int64_t num_hyp,
                                            const uint32_t* __restrict__ samples, double* __restrict__ models,
                                            int32_t* __restrict__ num_models) {
  const int64_t h = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (h >= num_hyp) return;
  double L[18], X[18];
  bool all_aligned = aligned != nullptr;
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const uint32_t id = samples[6 * h + i];
    L[3 * i] = d.l0[id]; L[3 * i + 1] = d.l1[id]; L[3 * i + 2] = d.l2[id];
    X[3 * i] = d.x0[id]; X[3 * i + 1] = d.x1[id]; X[3 * i + 2] = d.x2[id];
    if (aligned) all_aligned = all_aligned && aligned[id] != 0;
  }
  double out[96];
  const int n = P6LDevice(L, X, all_aligned, out, nullptr, nullptr);
  num_models[h] = n;
  double* dst = models + (size_t)h * 96;
#pragma unroll
  for (int s = 0; s < 8; ++s)
    if (s < n)
#pragma unroll
      for (int e = 0; e < 12; ++e) dst[12 * s + e] = out[12 * s + e];
}

__global__ __launch_bounds__(64) void k_re3q3(int64_t num, const double* __restrict__ coeffs, double* __restrict__ sols,
                                              int32_t* __restrict__ nsol) {
  const int64_t h = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (h >= num) return;
  double c[30], s[24];
#pragma unroll
  for (int i = 0; i < 30; ++i) c[i] = coeffs[30 * h + i];
#pragma unroll
  for (int i = 0; i < 24; ++i) s[i] = 0.0;
  const int n = Re3q3Device(c, s, true, nullptr);
  nsol[h] = n;
#pragma unroll
  for (int i = 0; i < 24; ++i) sols[24 * h + i] = s[i];
}

// best (num_inliers, -tree sum, lowest flat index) over all scored models: per-block candidates
__global__ __launch_bounds__(256) void k_best_candidates(int64_t num_hyp, const int32_t* __restrict__ num_models,
                                                         const uint32_t* __restrict__ inliers, const double* __restrict__ sums,
                                                         unsigned long long* __restrict__ out /* per block: {inl<<32|~?}, idx */) {
  // key ordering: more inliers, then smaller sum, then smaller flat index
  __shared__ uint32_t s_inl[256];
  __shared__ double s_sum[256];
  __shared__ unsigned long long s_idx[256];
  uint32_t bi = 0; double bs = DBL_MAX; unsigned long long bidx = ~0ull;
  for (int64_t h = (int64_t)blockIdx.x * 256 + threadIdx.x; h < num_hyp; h += (int64_t)gridDim.x * 256) {
    const int nm = num_models[h];
    for (int m = 0; m < nm; ++m) {
      const uint32_t c = inliers[h * 8 + m]; const double s = sums[h * 8 + m];
      const unsigned long long idx = (unsigned long long)h * 8 + m;
      if (c > bi || (c == bi && (s < bs || (s == bs && idx < bidx)))) { bi = c; bs = s; bidx = idx; }
    }
  }
  s_inl[threadIdx.x] = bi; s_sum[threadIdx.x] = bs; s_idx[threadIdx.x] = bidx;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) {
      const uint32_t c = s_inl[threadIdx.x + st]; const double s = s_sum[threadIdx.x + st]; const unsigned long long idx = s_idx[threadIdx.x + st];
      const uint32_t c0 = s_inl[threadIdx.x]; const double s0 = s_sum[threadIdx.x]; const unsigned long long i0 = s_idx[threadIdx.x];
      if (c > c0 || (c == c0 && (s < s0 || (s == s0 && idx < i0)))) { s_inl[threadIdx.x] = c; s_sum[threadIdx.x] = s; s_idx[threadIdx.x] = idx; }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out[3 * blockIdx.x] = s_inl[0];
    out[3 * blockIdx.x + 1] = (unsigned long long)__double_as_longlong(s_sum[0]);
    out[3 * blockIdx.x + 2] = s_idx[0];
  }
}

static CorrData Corr(const pp_pose_impl* h) { return CorrData{h->l0, h->l1, h->l2, h->x0, h->x1, h->x2, h->n}; }

// score h->flat_total models listed in h->flat
static void LaunchScoreFlat(pp_pose_impl* h, int64_t max_models, double max_residual) {
  const int64_t wg = CeilDiv(max_models, 8 * kFlatMPW);
  if (max_residual < DBL_MAX)
    hipLaunchKernelGGL(k_score_flat<true>, dim3((unsigned)wg), dim3(512), 0, h->stream, Corr(h), h->flat, h->flat_total, h->models,
                       max_residual, h->inliers, h->sums);
  else   // DBL_MAX <= max_residual (or NaN): points behind the camera count as inliers with r = DBL_MAX
    hipLaunchKernelGGL(k_score_flat<false>, dim3((unsigned)wg), dim3(512), 0, h->stream, Corr(h), h->flat, h->flat_total, h->models,
                       max_residual, h->inliers, h->sums);
}


static int EnsureCapacity(pp_pose_impl* h, int64_t hyp) {
  h->last_hyp = 0;      // every user of the work buffers passes here first: the scores of an earlier pp_pose_hypotheses are gone
  if (hyp <= h->cap_hyp) return PP_OK;
  void* old[] = {h->samples, h->models, h->num_models, h->inliers, h->sums, h->flat};
  for (void* p : old) if (p) (void)hipFree(p);
  h->samples = nullptr; h->models = nullptr; h->num_models = nullptr; h->inliers = nullptr; h->sums = nullptr; h->flat = nullptr;
  h->cap_hyp = 0; h->last_hyp = 0;
  int rc;
  if ((rc = DeviceAlloc(&h->samples, (size_t)hyp * 6))) return rc;
  if ((rc = DeviceAlloc(&h->models, (size_t)hyp * 96))) return rc;
  if ((rc = DeviceAlloc(&h->num_models, (size_t)hyp))) return rc;
  if ((rc = DeviceAlloc(&h->inliers, (size_t)hyp * 8))) return rc;
  if ((rc = DeviceAlloc(&h->sums, (size_t)hyp * 8))) return rc;
  if ((rc = DeviceAlloc(&h->flat, (size_t)hyp * 8))) return rc;
  if (!h->flat_total && (rc = DeviceAlloc(&h->flat_total, 4))) return rc;
  if (!h->flat_blocks && (rc = DeviceAlloc(&h->flat_blocks, 1024))) return rc;
  h->cap_hyp = hyp;
  return PP_OK;
}

static int EnsurePinned(pp_pose_impl* h, int64_t hyp) {
  if (hyp <= h->cap_pin) return PP_OK;
  void* old[] = {h->pin_samples, h->pin_inl, h->pin_nm, h->pin_sm, h->pin_mdl};
  for (void* p : old) if (p) (void)hipHostFree(p);
  h->pin_samples = nullptr; h->pin_inl = nullptr; h->pin_nm = nullptr; h->pin_sm = nullptr; h->pin_mdl = nullptr; h->cap_pin = 0;
  PP_HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&h->pin_samples), sizeof(uint32_t) * 6 * (size_t)hyp));
  PP_HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&h->pin_inl), sizeof(uint32_t) * 8 * (size_t)hyp));
  PP_HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&h->pin_nm), sizeof(int32_t) * (size_t)hyp));
  PP_HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&h->pin_sm), sizeof(double) * 8 * (size_t)hyp));
  PP_HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&h->pin_mdl), sizeof(double) * 96 * (size_t)hyp));
  h->cap_pin = hyp;
  return PP_OK;
}

// solve + score `count` hypotheses whose samples are already in h->samples
static int SolveAndScore(pp_pose_impl* h, int64_t count, double max_residual) {
  hipLaunchKernelGGL(k_p6l, dim3(CeilDiv(count, 64)), dim3(64), 0, h->stream, Corr(h), h->aligned, count, h->samples, h->models, h->num_models);
  {   // flat list of the returned models
    const int nblocks = (int)std::min<int64_t>(kFlattenThreads, CeilDiv(count, 1024));
    const int64_t span = CeilDiv(count, nblocks);
    hipLaunchKernelGGL(k_flatten_count, dim3(nblocks), dim3(kFlattenThreads), 0, h->stream, count, span, h->num_models, h->flat_blocks);
    hipLaunchKernelGGL(k_flatten_offsets, dim3(1), dim3(kFlattenThreads), 0, h->stream, nblocks, h->flat_blocks, h->flat_total);
    hipLaunchKernelGGL(k_flatten_fill, dim3(nblocks), dim3(kFlattenThreads), 0, h->stream, count, span, h->num_models, h->flat_blocks, h->flat);
  }
  // grid sized for the worst case (8 models per hypothesis); workgroups beyond the flat total exit at once
  LaunchScoreFlat(h, count * 8, max_residual);
  PP_HIP_TRY(hipGetLastError());
  return PP_OK;
}

}  // namespace ppsfm

using namespace ppsfm;

extern "C" {

static void ApplyAberthKnob() {
  const char* e = std::getenv("PPSFM_ABERTH_SWEEPS");
  if (!e) return;
  int v = std::atoi(e);
  if (v < 1) v = 1;
  (void)hipMemcpyToSymbol(HIP_SYMBOL(ppsfm::g_aberth_sweeps), &v, sizeof(int));
}

int pp_pose_destroy(pp_pose_handle h) try {
  if (!h) return PP_OK;
  (void)hipSetDevice(h->device);
  void* bufs[] = {h->l0, h->l1, h->l2, h->x0, h->x1, h->x2, h->aligned, h->samples, h->models, h->num_models, h->inliers,
                  h->sums, h->residuals, h->best_key, h->flat, h->flat_total, h->flat_blocks};
  for (void* b : bufs) if (b) (void)hipFree(b);
  { void* pins[] = {h->pin_samples, h->pin_inl, h->pin_nm, h->pin_sm, h->pin_mdl}; for (void* b : pins) if (b) (void)hipHostFree(b); }
  if (h->ev0) (void)hipEventDestroy(h->ev0);
  if (h->ev1) (void)hipEventDestroy(h->ev1);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
  return PP_OK;
} PP_API_CATCH("pp_pose_destroy")

int pp_pose_create(int32_t n, const double* lines2D, const double* points3D, const uint8_t* aligned, int device,
                   pp_pose_handle* out) try {
  PP_REQUIRE(out, "pp_pose_create: null out");
  *out = nullptr;
  PP_REQUIRE(n >= 0 && (n == 0 || (lines2D && points3D)), "pp_pose_create: bad argument");
  int ndev = 0;
  PP_HIP_TRY(hipGetDeviceCount(&ndev));
  PP_REQUIRE(device >= 0 && device < ndev, "pp_pose_create: device %d of %d", device, ndev);
  PP_HIP_TRY(hipSetDevice(device));
  ApplyAberthKnob();
  pp_pose_impl* h = new pp_pose_impl();
  OnUnwind unwind{[&] { pp_pose_destroy(h); }};
  h->device = device; h->n = n;
  int rc = PP_OK;
#define TRY(x) do { rc = (x); if (rc) { pp_pose_destroy(h); return rc; } } while (0)
#define TRYH(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { SetLastError("%s: %s", #x, hipGetErrorString(e_)); pp_pose_destroy(h); return PP_ERR_HIP; } } while (0)
  TRYH(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
  TRYH(hipEventCreate(&h->ev0)); TRYH(hipEventCreate(&h->ev1));
  const size_t nn = std::max(n, 1);
  std::vector<double> soa(6 * nn, 0.0);
  for (int i = 0; i < n; ++i)
    for (int c = 0; c < 3; ++c) { soa[c * nn + i] = lines2D[3 * i + c]; soa[(3 + c) * nn + i] = points3D[3 * i + c]; }
  TRY(DeviceAlloc(&h->l0, nn)); TRY(DeviceAlloc(&h->l1, nn)); TRY(DeviceAlloc(&h->l2, nn));
  TRY(DeviceAlloc(&h->x0, nn)); TRY(DeviceAlloc(&h->x1, nn)); TRY(DeviceAlloc(&h->x2, nn));
  double* dst[6] = {h->l0, h->l1, h->l2, h->x0, h->x1, h->x2};
  for (int c = 0; c < 6; ++c) TRY(Upload(dst[c], soa.data() + c * nn, nn, h->stream));
  if (aligned && n > 0) { TRY(DeviceAlloc(&h->aligned, nn)); TRY(Upload(h->aligned, aligned, (size_t)n, h->stream)); }
  TRY(DeviceAlloc(&h->best_key, 3 * 1024));
  TRYH(hipStreamSynchronize(h->stream));
#undef TRY
#undef TRYH
  *out = h;
  return PP_OK;
} PP_API_CATCH("pp_pose_create")

int pp_pose_residuals(pp_pose_handle h, int32_t num_models, const double* models, double* residuals_out) try {
  PP_REQUIRE(h && num_models >= 0 && (num_models == 0 || (models && residuals_out)), "pp_pose_residuals: bad argument");
  if (num_models == 0 || h->n == 0) return PP_OK;
  PP_HIP_TRY(hipSetDevice(h->device));
  const int64_t need = (int64_t)num_models * h->n;
  if (need > h->cap_res) {
    if (h->residuals) (void)hipFree(h->residuals);
    h->residuals = nullptr; h->cap_res = 0;
    int rc = DeviceAlloc(&h->residuals, (size_t)need); if (rc) return rc;
    h->cap_res = need;
  }
  int rc = EnsureCapacity(h, CeilDiv(num_models, 8) + 1); if (rc) return rc;
  rc = Upload(h->models, models, (size_t)num_models * 12, h->stream); if (rc) return rc;
  hipLaunchKernelGGL(k_residuals, dim3(CeilDiv(h->n, 256), num_models), dim3(256), 0, h->stream, Corr(h), num_models, h->models, h->residuals);
  PP_HIP_TRY(hipGetLastError());
  rc = Download(residuals_out, h->residuals, (size_t)need, h->stream); if (rc) return rc;
  PP_HIP_TRY(hipStreamSynchronize(h->stream));
  return PP_OK;
} PP_API_CATCH("pp_pose_residuals")

static int ScoreImpl(pp_pose_handle h, int32_t num_models, const double* models, double max_residual, uint32_t* num_inliers,
                     double* residual_sum, bool sequential) {
  PP_REQUIRE(h && num_models >= 0 && (num_models == 0 || (models && num_inliers && residual_sum)), "pp_pose_score: bad argument");
  if (num_models == 0) return PP_OK;
  PP_HIP_TRY(hipSetDevice(h->device));
  int rc = EnsureCapacity(h, CeilDiv(num_models, 8) + 1); if (rc) return rc;
  rc = Upload(h->models, models, (size_t)num_models * 12, h->stream); if (rc) return rc;
  if (sequential)
    hipLaunchKernelGGL(k_support_sequential, dim3(CeilDiv(num_models, 4)), dim3(256), 0, h->stream, Corr(h), num_models, h->models, max_residual, h->inliers, h->sums);
  else {   // the RANSAC scoring kernel itself, on the identity list
    hipLaunchKernelGGL(k_flat_identity, dim3(CeilDiv(num_models, 256)), dim3(256), 0, h->stream, num_models, h->flat, h->flat_total);
    LaunchScoreFlat(h, num_models, max_residual);
  }
  PP_HIP_TRY(hipGetLastError());
  rc = Download(num_inliers, h->inliers, (size_t)num_models, h->stream); if (rc) return rc;
  rc = Download(residual_sum, h->sums, (size_t)num_models, h->stream); if (rc) return rc;
  PP_HIP_TRY(hipStreamSynchronize(h->stream));
  return PP_OK;
}

int pp_pose_score(pp_pose_handle h, int32_t num_models, const double* models, double max_residual, uint32_t* num_inliers,
                  double* residual_sum) try {
  return ScoreImpl(h, num_models, models, max_residual, num_inliers, residual_sum, false);
} PP_API_CATCH("pp_pose_score")
int pp_pose_support_sequential(pp_pose_handle h, int32_t num_models, const double* models, double max_residual,
                               uint32_t* num_inliers, double* residual_sum) try {
  return ScoreImpl(h, num_models, models, max_residual, num_inliers, residual_sum, true);
} PP_API_CATCH("pp_pose_support_sequential")

int pp_pose_p6l_batch(pp_pose_handle h, int64_t num_hyp, const uint32_t* samples, double* models_out, int32_t* num_models_out) try {
  PP_REQUIRE(h && num_hyp >= 0 && (num_hyp == 0 || (samples && models_out && num_models_out)), "pp_pose_p6l_batch: bad argument");
  if (num_hyp == 0) return PP_OK;
  for (int64_t i = 0; i < 6 * num_hyp; ++i) PP_REQUIRE(samples[i] < (uint32_t)h->n, "pp_pose_p6l_batch: sample index out of range");
  PP_HIP_TRY(hipSetDevice(h->device));
  int rc = EnsureCapacity(h, num_hyp); if (rc) return rc;
  rc = Upload(h->samples, samples, (size_t)num_hyp * 6, h->stream); if (rc) return rc;
  PP_HIP_TRY(hipMemsetAsync(h->models, 0, sizeof(double) * 96 * (size_t)num_hyp, h->stream));
  hipLaunchKernelGGL(k_p6l, dim3(CeilDiv(num_hyp, 64)), dim3(64), 0, h->stream, Corr(h), h->aligned, num_hyp, h->samples, h->models, h->num_models);
  PP_HIP_TRY(hipGetLastError());
  rc = Download(models_out, h->models, (size_t)num_hyp * 96, h->stream); if (rc) return rc;
  rc = Download(num_models_out, h->num_models, (size_t)num_hyp, h->stream); if (rc) return rc;
  PP_HIP_TRY(hipStreamSynchronize(h->stream));
  return PP_OK;
} PP_API_CATCH("pp_pose_p6l_batch")

int pp_re3q3_batch(int64_t num, const double* coeffs, double* solutions, int32_t* num_solutions, int device) try {
  PP_REQUIRE(num >= 0 && (num == 0 || (coeffs && solutions && num_solutions)), "pp_re3q3_batch: bad argument");
  if (num == 0) return PP_OK;
  PP_HIP_TRY(hipSetDevice(device));
  double *dc = nullptr, *ds = nullptr; int32_t* dn = nullptr;
  int rc;
  if ((rc = DeviceAlloc(&dc, (size_t)num * 30)) || (rc = DeviceAlloc(&ds, (size_t)num * 24)) || (rc = DeviceAlloc(&dn, (size_t)num))) {
    if (dc) (void)hipFree(dc); if (ds) (void)hipFree(ds); if (dn) (void)hipFree(dn);
    return rc;
  }
  hipError_t e = hipMemcpy(dc, coeffs, sizeof(double) * 30 * num, hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_re3q3, dim3(CeilDiv(num, 64)), dim3(64), 0, 0, num, dc, ds, dn);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpy(solutions, ds, sizeof(double) * 24 * num, hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(num_solutions, dn, sizeof(int32_t) * num, hipMemcpyDeviceToHost);
  (void)hipFree(dc); (void)hipFree(ds); (void)hipFree(dn);
  if (e != hipSuccess) { SetLastError("pp_re3q3_batch: %s", hipGetErrorString(e)); return PP_ERR_HIP; }
  return PP_OK;
} PP_API_CATCH("pp_re3q3_batch")

int pp_pose_hypotheses(pp_pose_handle h, int64_t num_hyp, const uint32_t* samples, uint32_t seed, double max_residual,
                       pp_ransac_report* rep) try {
  PP_REQUIRE(h && rep && num_hyp > 0, "pp_pose_hypotheses: bad argument");
  PP_REQUIRE(h->n >= 6, "pp_pose_hypotheses: fewer than 6 correspondences");
  PP_HIP_TRY(hipSetDevice(h->device));
  const auto t0 = std::chrono::steady_clock::now();
  std::memset(rep, 0, sizeof(*rep));
  int rc = EnsureCapacity(h, num_hyp); if (rc) return rc;
  if (samples) {
    rc = Upload(h->samples, samples, (size_t)num_hyp * 6, h->stream); if (rc) return rc;
  } else {
    if ((int64_t)h->h_samples.size() != num_hyp * 6 || h->h_samples_seed != seed) {
      h->h_samples.resize((size_t)num_hyp * 6);
      h->h_samples_seed = seed;
      RandomSampler sampler(6, seed); sampler.Initialize((uint32_t)h->n);
      for (int64_t i = 0; i < num_hyp; ++i) sampler.Sample(h->h_samples.data() + 6 * i);
    }
    rc = Upload(h->samples, h->h_samples.data(), (size_t)num_hyp * 6, h->stream); if (rc) return rc;
  }
  PP_HIP_TRY(hipEventRecord(h->ev0, h->stream));
  rc = SolveAndScore(h, num_hyp, max_residual); if (rc) return rc;
  const int nblk = 1024;
  hipLaunchKernelGGL(k_best_candidates, dim3(nblk), dim3(256), 0, h->stream, num_hyp, h->num_models, h->inliers, h->sums, h->best_key);
  PP_HIP_TRY(hipGetLastError());
  PP_HIP_TRY(hipEventRecord(h->ev1, h->stream));
  std::vector<unsigned long long> cand(3 * nblk);
  rc = Download(cand.data(), h->best_key, cand.size(), h->stream); if (rc) return rc;
  std::vector<int32_t> nm((size_t)num_hyp);
  rc = Download(nm.data(), h->num_models, (size_t)num_hyp, h->stream); if (rc) return rc;
  PP_HIP_TRY(hipStreamSynchronize(h->stream));
  uint32_t bi = 0; double bs = DBL_MAX; unsigned long long bidx = ~0ull;
  for (int b = 0; b < nblk; ++b) {
    const uint32_t c = (uint32_t)cand[3 * b]; double s; std::memcpy(&s, &cand[3 * b + 1], 8); const unsigned long long idx = cand[3 * b + 2];
    if (idx == ~0ull) continue;
    if (c > bi || (c == bi && (s < bs || (s == bs && idx < bidx)))) { bi = c; bs = s; bidx = idx; }
  }
  uint64_t total_models = 0;
  for (int64_t i = 0; i < num_hyp; ++i) total_models += (uint64_t)nm[i];
  rep->hypotheses_evaluated = (uint64_t)num_hyp; rep->models_scored = total_models; rep->num_trials = (uint64_t)num_hyp;
  h->last_hyp = num_hyp;
  if (bidx != ~0ull) {
    rep->success = bi >= 6; rep->num_inliers = bi; rep->residual_sum = bs;
    rep->best_trial = (int64_t)(bidx / 8); rep->best_model_index = (int32_t)(bidx % 8);
    PP_HIP_TRY(hipMemcpy(rep->model, h->models + bidx * 12, sizeof(double) * 12, hipMemcpyDeviceToHost));
  }
  float ms = 0; PP_HIP_TRY(hipEventElapsedTime(&ms, h->ev0, h->ev1));
  rep->device_time_s = ms * 1e-3;
  rep->total_time_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  return PP_OK;
} PP_API_CATCH("pp_pose_hypotheses")

int pp_pose_last_scores(pp_pose_handle h, int64_t num_hyp, int32_t* num_models, uint32_t* num_inliers, double* residual_sum) try {
  PP_REQUIRE(h && num_hyp > 0, "pp_pose_last_scores: bad argument");
  PP_REQUIRE(num_hyp <= h->last_hyp, "pp_pose_last_scores: the last pp_pose_hypotheses call scored %lld hypotheses, %lld asked for",
             (long long)h->last_hyp, (long long)num_hyp);
  PP_HIP_TRY(hipSetDevice(h->device));
  int rc;
  if (num_models && (rc = Download(num_models, h->num_models, (size_t)num_hyp, h->stream))) return rc;
  if (num_inliers && (rc = Download(num_inliers, h->inliers, (size_t)num_hyp * 8, h->stream))) return rc;
  if (residual_sum && (rc = Download(residual_sum, h->sums, (size_t)num_hyp * 8, h->stream))) return rc;
  PP_HIP_TRY(hipStreamSynchronize(h->stream));
  return PP_OK;
} PP_API_CATCH("pp_pose_last_scores")

int pp_pose_ransac(pp_pose_handle h, const pp_ransac_options* o, pp_ransac_report* rep, uint8_t* inlier_mask) try {
  PP_REQUIRE(h && o && rep, "pp_pose_ransac: null argument");
  // RANSACOptions::Check (optim/ransac.h:68-75)
  PP_REQUIRE(o->max_error > 0 && o->min_inlier_ratio >= 0 && o->min_inlier_ratio <= 1 && o->confidence >= 0 && o->confidence <= 1 &&
             o->min_num_trials <= o->max_num_trials, "pp_pose_ransac: RANSACOptions::Check failed");
  PP_HIP_TRY(hipSetDevice(h->device));
  const auto t0 = std::chrono::steady_clock::now();
  std::memset(rep, 0, sizeof(*rep));
  rep->best_trial = -1; rep->best_model_index = -1;
  rep->residual_sum = DBL_MAX;  // InlierSupportMeasurer::Support default (support_measurement.h:44-50)
  const int n = h->n;
  if (inlier_mask && n > 0) std::memset(inlier_mask, 0, (size_t)n);
  const int kMin = 6;
  // ctor: a-priori cap of max_num_trials (optim/ransac.h:149-155)
  uint64_t max_num_trials = o->max_num_trials;
  {
    const uint64_t kNumSamples = 100000;
    const uint64_t dyn = ComputeNumTrials((uint64_t)(o->min_inlier_ratio * kNumSamples), kNumSamples, o->confidence,
                                          o->dyn_num_trials_multiplier, kMin);
    max_num_trials = std::min(max_num_trials, dyn);
  }
  if (n < kMin) { rep->total_time_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); return PP_OK; }

  const double max_residual = o->max_error * o->max_error;
  RandomSampler sampler(kMin, o->seed);
  sampler.Initialize((uint32_t)n);
  uint64_t dyn_max_num_trials = max_num_trials;
  // best support so far; the exact (sequential-order) residual sum is computed lazily, only when a
  // comparison actually needs it (a tie in num_inliers, or the final report)
  uint64_t best_inl = 0; double best_sum = DBL_MAX; bool best_sum_exact = true; bool have_best = false;
  double best_model[12] = {0};
  bool abort = false;
  uint64_t trial = 0;
  uint32_t chunk = o->chunk_trials ? o->chunk_trials : 1024;
  uint32_t* hs = nullptr; int32_t* nm = nullptr; uint32_t* inl = nullptr; double* sm = nullptr; double* mdl = nullptr;   // pinned mirrors of a chunk
  std::vector<size_t> cand; std::vector<double> cand_models, cand_sums, exact_of;
  double dev_s = 0;

  auto exact_support = [&](const double* model, uint32_t* c, double* s) -> int {
    int rc = Upload(h->models, model, 12, h->stream); if (rc) return rc;
    hipLaunchKernelGGL(k_support_sequential, dim3(1), dim3(256), 0, h->stream, Corr(h), 1, h->models, max_residual, h->inliers, h->sums);
    PP_HIP_TRY(hipGetLastError());
    rc = Download(c, h->inliers, 1, h->stream); if (rc) return rc;
    rc = Download(s, h->sums, 1, h->stream); if (rc) return rc;
    PP_HIP_TRY(hipStreamSynchronize(h->stream));
    return PP_OK;
  };

  while (trial < max_num_trials && !abort) {
    // speculate: never beyond the static cap; shrink towards the dynamic bound once it is known
    uint64_t want = std::min<uint64_t>(chunk, max_num_trials - trial);
    if (dyn_max_num_trials < max_num_trials) {
      const uint64_t floor_trials = std::max<uint64_t>(dyn_max_num_trials, o->min_num_trials);
      if (floor_trials > trial) want = std::min<uint64_t>(want, floor_trials - trial + 1);
      else want = std::min<uint64_t>(want, 64);
    }
    want = std::max<uint64_t>(want, 1);
    int rc = EnsureCapacity(h, (int64_t)std::max<uint64_t>(want, 64)); if (rc) return rc;
    rc = EnsurePinned(h, (int64_t)std::max<uint64_t>(want, chunk)); if (rc) return rc;
    hs = h->pin_samples; nm = h->pin_nm; inl = h->pin_inl; sm = h->pin_sm; mdl = h->pin_mdl;
    for (uint64_t i = 0; i < want; ++i) sampler.Sample(hs + 6 * i);
    rc = Upload(h->samples, hs, want * 6, h->stream); if (rc) return rc;
    PP_HIP_TRY(hipEventRecord(h->ev0, h->stream));
    rc = SolveAndScore(h, (int64_t)want, max_residual); if (rc) return rc;
    PP_HIP_TRY(hipEventRecord(h->ev1, h->stream));
    rc = Download(nm, h->num_models, want, h->stream); if (rc) return rc;
    rc = Download(inl, h->inliers, want * 8, h->stream); if (rc) return rc;
    rc = Download(sm, h->sums, want * 8, h->stream); if (rc) return rc;
    rc = Download(mdl, h->models, want * 96, h->stream); if (rc) return rc;
    PP_HIP_TRY(hipStreamSynchronize(h->stream));
    float ms = 0; PP_HIP_TRY(hipEventElapsedTime(&ms, h->ev0, h->ev1)); dev_s += ms * 1e-3;
    rep->hypotheses_evaluated += want;

    // The exact (sequential-order) residual sums the replay can ask for, in ONE batch: a model can only become the best or
    // tie with it if its inlier count reaches the running maximum of the counts before it, and that maximum follows from
    // the counts alone.  (One launch + one read-back per chunk instead of a round trip per tie.)
    exact_of.assign(want * 8, 0.0);
    {
      cand.clear(); cand_models.clear();
      const bool carry = have_best && !best_sum_exact;
      if (carry) cand_models.insert(cand_models.end(), best_model, best_model + 12);
      uint64_t run = have_best ? best_inl : 0; bool hb = have_best;
      for (uint64_t i = 0; i < want; ++i)
        for (int m = 0; m < nm[i]; ++m) {
          const uint64_t c = inl[i * 8 + m];
          if (!hb || c >= run) {
            cand.push_back(i * 8 + m);
            cand_models.insert(cand_models.end(), mdl + (i * 8 + m) * 12, mdl + (i * 8 + m) * 12 + 12);
            if (!hb || c > run) run = c;
            hb = true;
          }
        }
      const size_t K = cand_models.size() / 12;
      if (K > 0) {
        rc = Upload(h->models, cand_models.data(), cand_models.size(), h->stream); if (rc) return rc;
        hipLaunchKernelGGL(k_support_sequential, dim3(CeilDiv((int64_t)K, 4)), dim3(256), 0, h->stream, Corr(h), (int)K, h->models, max_residual, h->inliers, h->sums);
        PP_HIP_TRY(hipGetLastError());
        cand_sums.resize(K);
        rc = Download(cand_sums.data(), h->sums, K, h->stream); if (rc) return rc;
        PP_HIP_TRY(hipStreamSynchronize(h->stream));
        size_t k0 = 0;
        if (carry) { best_sum = cand_sums[0]; best_sum_exact = true; k0 = 1; }
        for (size_t k = 0; k < cand.size(); ++k) exact_of[cand[k]] = cand_sums[k0 + k];
      }
    }
    // replay of optim/ransac.h:213-249 in trial order
    for (uint64_t i = 0; i < want; ++i, ++trial) {
      if (trial >= max_num_trials) break;
      if (abort) break;
      const int nmod = nm[i];
      for (int m = 0; m < nmod; ++m) {
        rep->models_scored += 1;
        const uint64_t c = inl[i * 8 + m];
        bool better = false;
        if (!have_best) {
          // Compare(support, default Support{0, DBL_MAX}): more inliers, or 0 inliers with sum 0 < DBL_MAX
          better = true;
        } else if (c > best_inl) {
          better = true;
        } else if (c == best_inl) {
          // tie on the count: the reference compares the SEQUENTIAL residual sums
          uint32_t ce;
          if (!best_sum_exact) { rc = exact_support(best_model, &ce, &best_sum); if (rc) return rc; best_sum_exact = true; }   // (not reached: batch above)
          const double se = exact_of[i * 8 + m];
          if (se < best_sum) better = true;
        }
        if (better) {
          have_best = true; best_inl = c;
          best_sum = exact_of[i * 8 + m]; best_sum_exact = true;
          std::memcpy(best_model, mdl + (i * 8 + m) * 12, sizeof(best_model));
          rep->best_trial = (int64_t)trial; rep->best_model_index = m;
          dyn_max_num_trials = ComputeNumTrials(best_inl, (uint64_t)n, o->confidence, o->dyn_num_trials_multiplier, kMin);
        }
        if (trial >= dyn_max_num_trials && trial >= o->min_num_trials) { abort = true; break; }
      }
    }
  }
  // num_trials bookkeeping of the reference loop (optim/ransac.h:213-218): an abort raised in trial t
  // ends the run with num_trials = t + 2 (for-increment, then the `if (abort) num_trials += 1`), or
  // t + 1 when t + 1 already equals max_num_trials
  // (after an abort in trial t the replay loop above leaves `trial` == t + 1)
  rep->num_trials = trial;
  if (abort && trial < max_num_trials) rep->num_trials = trial + 1;

  if (have_best && !best_sum_exact) { uint32_t ce; int rc = exact_support(best_model, &ce, &best_sum); if (rc) return rc; best_inl = ce; }
  rep->num_inliers = best_inl; rep->residual_sum = have_best ? best_sum : DBL_MAX;
  std::memcpy(rep->model, best_model, sizeof(best_model));
  rep->device_time_s = dev_s;
  if (best_inl >= (uint64_t)kMin) {
    rep->success = 1;
    if (inlier_mask) {
      // final mask: residuals of the winner, `<=` threshold (optim/ransac.h:265-275)
      std::vector<double> res((size_t)n);
      int rc = pp_pose_residuals(h, 1, best_model, res.data()); if (rc) return rc;
      for (int i = 0; i < n; ++i) inlier_mask[i] = res[i] <= max_residual ? 1 : 0;
    }
  }
  rep->total_time_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  return PP_OK;
} PP_API_CATCH("pp_pose_ransac")

}  // extern "C"

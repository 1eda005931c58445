// FourView2dEstimator::LeastSquares on the device (reference src/init/sfm2d.cc:42-175, 469-489):
//   bundle_adjust2d    cameras 1..3 + the sample's points, camera 0 constant, (cos, sin) pairs and the translation of
//                      camera 1 under HomogeneousVectorParameterization(2);          k_fv2d_bundle
//   optimize_points2d  all points, cameras constant                                  k_fv2d_points
// The reference hands both to Ceres (third party, absent from /root/reference, version unpinned: PARITY UNPINNED).  As in
// ba_solver.hip the solver restates Ceres' published trust-region Levenberg-Marquardt (Jacobi scaling fixed at the
// start, clamped LM diagonal / radius, radius /= max(1/3, 1-(2 rho-1)^3) resp. /2,/4.., tolerances 1e-10, 50
// iterations, exact Schur solve).  Both problems are tiny and strictly sequential across LM iterations, so each is ONE
// workgroup that keeps the whole iteration loop on the device: points are strided over 256 lanes, every reduction is a
// wave butterfly + fixed-order LDS sum (deterministic), the 8x8 reduced camera system is solved redundantly by every lane.
#pragma once
#include "common.hpp"

namespace ppsfm {

// sum of K values over the NW wavefronts of the workgroup, result in every thread (fixed order)
template <int K, int NW = 4>
__device__ __forceinline__ void BlockSumN(double (&v)[K], double* lds) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] = WaveSum(v[k]);
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < K; ++k) lds[wv * K + k] = v[k];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < K; ++k) {
    double part[NW / 4];
#pragma unroll
    for (int g = 0; g < NW / 4; ++g) part[g] = (lds[(4 * g) * K + k] + lds[(4 * g + 1) * K + k]) + (lds[(4 * g + 2) * K + k] + lds[(4 * g + 3) * K + k]);
    double t = part[0];
#pragma unroll
    for (int g = 1; g < NW / 4; ++g) t += part[g];
    v[k] = t;
  }
  __syncthreads();
}
template <int NW = 4>
__device__ __forceinline__ double BlockMax(double v, double* lds) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
  if (lane == 0) lds[wv] = v;
  __syncthreads();
  v = lds[0];
#pragma unroll
  for (int w = 1; w < NW; ++w) v = fmax(v, lds[w]);
  __syncthreads();
  return v;
}

// The same for a workgroup of ONE wavefront (the LO-MSAC samples are at most 35 points: three quarters of a 256-lane workgroup had nothing to add and every
// sum still paid a six-step ds_bpermute butterfly - 53 of them per LM iteration of the bundle refinement): the lanes' values go to LDS, lane k adds up
// column k over the first `rows` lanes in lane order, the totals come back as broadcast reads.  lds: K x 65 + K doubles.
template <int K>
__device__ __forceinline__ void WaveSumLds(double (&v)[K], double* lds, int rows) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int k = 0; k < K; ++k) lds[k * 65 + lane] = v[k];
  __syncthreads();      // (one wavefront: an LDS fence)
  if (lane < K) {
    double t = 0.0;
    for (int l = 0; l < rows; ++l) t += lds[lane * 65 + l];
    lds[K * 65 + lane] = t;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] = lds[K * 65 + k];
  __syncthreads();
}
template <int NW, int K>
__device__ __forceinline__ void GroupSumN(double (&v)[K], double* lds, int rows) {
  if constexpr (NW == 1) WaveSumLds<K>(v, lds, rows); else BlockSumN<K, NW>(v, lds);
}
template <int NW>
__device__ __forceinline__ double GroupMax(double v, double* lds) {
  if constexpr (NW == 1) return WaveMaxDpp(v); else return BlockMax<NW>(v, lds);
}

// residual of BundleAdjustment2DCostFunction (sfm2d.cc:42-76) and its derivatives wrt (q0,q1,t0,t1,X0,X1)
__device__ __forceinline__ double Residual2d(const double* q, const double* t, double X0, double X1, double xa, double xb, double* d) {
  const double p0 = q[0] * X0 - q[1] * X1 + t[0], p1 = q[1] * X0 + q[0] * X1 + t[1];
  if (d) {
    const double a = 1.0 / p1, b = -p0 / (p1 * p1);
    d[0] = a * X0 + b * X1; d[1] = -a * X1 + b * X0; d[2] = a; d[3] = b;
    d[4] = a * q[0] + b * q[1]; d[5] = -a * q[1] + b * q[0];
  }
  return p0 / p1 - xa / xb;
}

// the same residual with the observation's ratio xa / xb precomputed and ONE reciprocal (1 / p1) per evaluation;
// d = derivatives wrt the point (X0, X1) only.  Used by k_fv2d_points, whose per-iteration cost is its divisions.
__device__ __forceinline__ double ResidualPoint2d(const double* q, const double* t, double X0, double X1, double ratio, double* d) {
  const double p0 = q[0] * X0 - q[1] * X1 + t[0], p1 = q[1] * X0 + q[0] * X1 + t[1];
  const double a = 1.0 / p1, u = p0 * a;
  if (d) { const double b = -u * a; d[0] = a * q[0] + b * q[1]; d[1] = -a * q[1] + b * q[0]; }
  return u - ratio;
}

// HomogeneousVectorParameterization of size 2 (Ceres): Householder frame of x, Plus rotates x by |delta| / 2
__device__ __forceinline__ void Householder2(const double* x, double* v, double* beta) {
  const double sigma = x[0] * x[0];
  v[0] = x[0]; v[1] = 1.0; *beta = 0.0;
  const double pivot = x[1];
  if (sigma <= 2.220446049250313e-16) { if (pivot < 0.0) *beta = 2.0; return; }
  const double mu = sqrt(pivot * pivot + sigma);
  const double vp = (pivot <= 0.0) ? pivot - mu : -sigma / (pivot + mu);
  *beta = 2.0 * vp * vp / (sigma + vp * vp);
  v[0] /= vp;
}
__device__ __forceinline__ void HomogeneousPlus2(const double* x, double delta, double* out) {
  const double nd = fabs(delta);
  if (nd == 0.0) { out[0] = x[0]; out[1] = x[1]; return; }
  const double half = 0.5 * nd;
  const double y0 = 0.5 * (sin(half) / half) * delta, y1 = cos(half);
  double v[2], beta;
  Householder2(x, v, &beta);
  const double nx = sqrt(x[0] * x[0] + x[1] * x[1]);
  const double vy = v[0] * y0 + v[1] * y1;
  out[0] = nx * (y0 - v[0] * beta * vy); out[1] = nx * (y1 - v[1] * beta * vy);
}
__device__ __forceinline__ void HomogeneousJacobian2(const double* x, double* J) {
  double v[2], beta;
  Householder2(x, v, &beta);
  const double nx = sqrt(x[0] * x[0] + x[1] * x[1]);
  J[0] = 0.5 * nx * (1.0 - beta * v[0] * v[0]); J[1] = 0.5 * nx * (-beta * v[1] * v[0]);
}

struct TrustRegionState {   // Ceres TrustRegionMinimizer + LevenbergMarquardtStrategy bookkeeping (uniform across the workgroup)
  double radius, decrease;
  int invalid;
  __device__ void Accept(double rel) { const double v = 2.0 * rel - 1.0; radius = fmin(1e16, radius / fmax(1.0 / 3.0, 1.0 - v * v * v)); decrease = 2.0; }   // pow(v, 3) up to an ulp, without libm's register footprint
  __device__ void Reject() { radius /= decrease; decrease *= 2.0; }
};

// ---- optimize_points2d: all n points, cameras constant, one joint LM (block-diagonal 2x2 system) --------------------
// X (n x 2) in/out; scratch: scale (n x 2), Xc (n x 2), ratio (4 x n: xa / xb of every observation), xch (the exchange slots below).
// An LM iteration is ~450 dependent fp64 instructions per point and the iterations are strictly sequential (one joint trust
// region: every iteration ends in seven sums over ALL points), so the loop is bound by the issue rate of the lanes it runs on.
// One workgroup (512 lanes, four points per lane at n = 2000) took 12 us per iteration, 424 us per call.  Now the points are
// spread over up to kPointsMaxGroups workgroups of 256 lanes (one point per lane at n = 2000) that exchange their seven partial
// sums per iteration through poisoned slots in global memory - one store and one polled load per iteration, no counter:
//   slot set (iteration mod 3), one row of 8 doubles per workgroup, preset to the all-ones NaN pattern by the host;
//   iteration i: a workgroup stores its partial sums into set i mod 3 and polls the whole set until no value is the pattern, then
//   re-poisons its row of set (i+2) mod 3 (= the rows of iteration i-1, which every workgroup has finished reading: they all wrote
//   iteration i after it); everybody then adds the rows in workgroup order (deterministic, the same in every workgroup).
// The workgroups must be co-resident (at most 16 x 256 lanes: they are, unless the chip is full of kernels that wait for this one).
constexpr int kPointsThreads = 256;
constexpr int kPointsMaxGroups = 16;
constexpr int kPointsSlotDoubles = 3 * kPointsMaxGroups * 8;
__device__ __forceinline__ void ExchangeSums(double (&s)[6], double& gmax, double* __restrict__ xch, int iter, double* lds) {
  const int G = gridDim.x;
  if (G == 1) return;
  const unsigned long long kPattern = 0xFFFFFFFFFFFFFFFFull;
  double* mine = xch + ((size_t)(iter % 3) * kPointsMaxGroups + blockIdx.x) * 8;
  const double* set = xch + (size_t)(iter % 3) * kPointsMaxGroups * 8;
  // slot 7 of the first row is never a sum: it is the group's ABORT word (pattern = running).  A workgroup whose wait ran out clears
  // it, every poll loop looks at it now and then: all workgroups leave the iteration with NaN sums together instead of each waiting
  // out its own bound on a peer that has already gone.
  unsigned long long* abort_word = reinterpret_cast<unsigned long long*>(xch) + 7;
  if (threadIdx.x < 7) {
    double v = threadIdx.x < 6 ? s[threadIdx.x] : gmax;
    unsigned long long bits = (unsigned long long)__double_as_longlong(v);
    if (bits == kPattern) bits = 0x7FF8000000000000ull;      // a NaN sum stays a NaN, never the pattern
    // ordering: a peer that sees all seven values of this row may run ahead into iteration i+1 and must then find this workgroup's row
    // of set (i+1) mod 3 poisoned, not the sums of iteration i-2.  That row was re-poisoned at the END of the previous exchange (below),
    // a whole pass over the points ago; the wait makes it certain without a release fence (which would write back the L2).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(mine) + threadIdx.x, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (threadIdx.x < G * 8 && (threadIdx.x & 7) < 7) {      // lane (g, k) polls value k of workgroup g
    const unsigned long long* p = reinterpret_cast<const unsigned long long*>(set) + threadIdx.x;
    unsigned long long bits = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int spins = 0; bits == kPattern; ++spins) {
      if (spins >= (1 << 22)) { __hip_atomic_store(abort_word, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
      if ((spins & 255) == 255 && __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != kPattern) break;
      __builtin_amdgcn_s_sleep(1);
      bits = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    lds[threadIdx.x] = __longlong_as_double((long long)bits);      // (a timeout / an abort leaves the NaN pattern: every sum turns NaN, the loop ends as invalid)
  }
  __syncthreads();
  // AFTER the barrier every poll lane of this workgroup (both polling wavefronts: workgroups 0..7 and 8..15) has seen iteration i of
  // every peer, so every peer has finished reading the rows of iteration i-1 = set (i+2) mod 3: this workgroup's row there is
  // re-poisoned now, for iteration i+2.  (Before the barrier only wavefront 0's polls - workgroups 0..7 - were ordered before it.)
  if (threadIdx.x < 7)
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(xch + ((size_t)((iter + 2) % 3) * kPointsMaxGroups + blockIdx.x) * 8) + threadIdx.x, kPattern, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
  for (int k = 0; k < 6; ++k) { double t = 0.0; for (int g = 0; g < G; ++g) t += lds[8 * g + k]; s[k] = t; }
  { double t = 0.0; for (int g = 0; g < G; ++g) t = fmax(t, lds[8 * g + 6]); gmax = t; }
  __syncthreads();
}

__global__ __launch_bounds__(kPointsThreads) void k_fv2d_points(int n, const double* __restrict__ x, const double* __restrict__ cams, double* __restrict__ X,
                                                                double* __restrict__ scale, double* __restrict__ Xc, double* __restrict__ ratio,
                                                                double* __restrict__ xch) {
  constexpr int NW = kPointsThreads / 64;
  __shared__ double lds[kPointsMaxGroups * 8];
  const double kTol = 1e-10;
  const int first_point = blockIdx.x * kPointsThreads + threadIdx.x, stride = gridDim.x * kPointsThreads;
  double q[4][2], t[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i) { q[i][0] = cams[6 * i]; q[i][1] = cams[6 * i + 3]; t[i][0] = cams[6 * i + 2]; t[i][1] = cams[6 * i + 5]; }
  for (int j = first_point; j < n; j += stride)
#pragma unroll
    for (int i = 0; i < 4; ++i) ratio[(size_t)i * n + j] = x[((size_t)i * n + j) * 2] / x[((size_t)i * n + j) * 2 + 1];
  TrustRegionState tr{1e4, 2.0, 0};
  bool last_ok = true, first = true;
  for (int iter = 1;; ++iter) {
    // one pass: H, g at X; (first: Jacobi scale); LM step for the current radius; model change; candidate; candidate cost
    double s[6] = {0, 0, 0, 0, 0, 0};     // cost, model, |step|^2, |x|^2, candidate cost, invalid count
    double gmax = 0.0;
    for (int j = first_point; j < n; j += stride) {
      const double X0 = X[2 * (size_t)j], X1 = X[2 * (size_t)j + 1];
      double rt[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) rt[i] = ratio[(size_t)i * n + j];
      double h00 = 0, h01 = 0, h11 = 0, g0 = 0, g1 = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        double d[2];
        const double r = ResidualPoint2d(q[i], t[i], X0, X1, rt[i], d);
        s[0] += 0.5 * r * r;
        h00 += d[0] * d[0]; h01 += d[0] * d[1]; h11 += d[1] * d[1]; g0 += d[0] * r; g1 += d[1] * r;
      }
      gmax = fmax(gmax, fmax(fabs(g0), fabs(g1)));
      double s0, s1;
      if (first) { s0 = 1.0 / (1.0 + sqrt(h00)); s1 = 1.0 / (1.0 + sqrt(h11)); scale[2 * (size_t)j] = s0; scale[2 * (size_t)j + 1] = s1; }
      else { s0 = scale[2 * (size_t)j]; s1 = scale[2 * (size_t)j + 1]; }
      const double dg0 = fmin(fmax(s0 * s0 * h00, 1e-6), 1e32), dg1 = fmin(fmax(s1 * s1 * h11, 1e-6), 1e32);
      const double a = s0 * s0 * h00 + dg0 / tr.radius, b = s0 * s1 * h01, c = s1 * s1 * h11 + dg1 / tr.radius;
      const double det = a * c - b * b;
      if (!(det > 0.0)) { s[5] += 1.0; continue; }
      const double r0 = -s0 * g0, r1 = -s1 * g1;
      const double inv_det = 1.0 / det;
      const double e0 = s0 * (c * r0 - b * r1) * inv_det, e1 = s1 * (a * r1 - b * r0) * inv_det;
      s[1] -= g0 * e0 + g1 * e1 + 0.5 * (h00 * e0 * e0 + 2.0 * h01 * e0 * e1 + h11 * e1 * e1);
      s[2] += e0 * e0 + e1 * e1; s[3] += X0 * X0 + X1 * X1;
      const double c0 = X0 + e0, c1 = X1 + e1;
      Xc[2 * (size_t)j] = c0; Xc[2 * (size_t)j + 1] = c1;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const double r = ResidualPoint2d(q[i], t[i], c0, c1, rt[i], nullptr);
        s[4] += 0.5 * r * r;
      }
    }
    first = false;
    BlockSumN<6, NW>(s, lds);
    gmax = BlockMax<NW>(gmax, lds);
    ExchangeSums(s, gmax, xch, iter, lds);
    if (last_ok && gmax <= kTol) break;
    if (iter > 50 || tr.radius < 1e-32) break;
    if (s[5] > 0.0 || !(s[1] > 0.0)) { if (++tr.invalid >= 5) break; tr.Reject(); last_ok = false; continue; }
    tr.invalid = 0;
    if (sqrt(s[2]) <= kTol * (sqrt(s[3]) + kTol)) break;
    const double change = s[0] - s[4];
    if (fabs(change) <= kTol * s[0]) break;
    const double rel = change / s[1];
    if (rel > 1e-3) {
      for (int j = first_point; j < n; j += stride) { X[2 * (size_t)j] = Xc[2 * (size_t)j]; X[2 * (size_t)j + 1] = Xc[2 * (size_t)j + 1]; }      // (each lane its own points: no barrier needed)
      tr.Accept(rel); last_ok = true;
    } else { tr.Reject(); last_ok = false; }
  }
}

// ---- the same loop with ONE point per lane, held in registers (n <= kPointsWaveMax) -------------------------------------------------------------
// k_fv2d_points above spends most of an iteration OUTSIDE the arithmetic: per iteration it re-reads X, the ratios and the scale from global memory, stores the
// candidate, crosses six workgroup barriers (two block sums, the exchange, whose poll lanes each wait for their own word) - 4.7 us per iteration at n = 2000
// for ~0.8 us of fp64 issue, and with outlier tracks in the data the joint problem never meets its 1e-10 tolerances: every call runs the full 50 iterations
// (tools/fourview_lomsac_probe.py; a numpy replica of the loop on the bench scene shows relative decreases of 0.2 - 0.75 to the end).  Here a point lives
// in its lane's registers for the whole loop (X, candidate, scale, four ratios), the seven per-iteration values are reduced per wavefront by DPP moves, per
// workgroup through LDS (one barrier), and between the G workgroups as above - three row sets in rotation, poisoned by the bundle kernel before the launch
// and by their owner two iterations ahead - but polled by ONE wavefront with every load of a round in flight together, and summed once per workgroup.
// A lane without a point contributes zeros.  The sums are taken in another order than k_fv2d_points takes them: equal to rounding, and the same on
// every path of the library (pp_fourview2d_least_squares, the LO-MSAC replay and the RansacLib adaptor all launch this kernel).
constexpr int kPointsWaveMax = 4096;                      // at most 16 workgroups of 256 lanes
__global__ __launch_bounds__(256) void k_fv2d_points_reg(int n, const double* __restrict__ x, const double* __restrict__ cams, double* __restrict__ X, double* __restrict__ xch,
                                                         int32_t* __restrict__ iterations_out) {
  const double kTol = 1e-10;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, G = gridDim.x, j = blockIdx.x * 256 + threadIdx.x;
  const bool has = j < n;
  __shared__ double wsum[16 * 8];      // the sixteen 16-lane rows' values
  __shared__ double rows[kPointsMaxGroups * 8];      // the polled rows of an iteration
  __shared__ double tot[8];
  double q[4][2], t[4][2], rt[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { q[i][0] = cams[6 * i]; q[i][1] = cams[6 * i + 3]; t[i][0] = cams[6 * i + 2]; t[i][1] = cams[6 * i + 5]; }
#pragma unroll
  for (int i = 0; i < 4; ++i) rt[i] = has ? x[((size_t)i * n + j) * 2] / x[((size_t)i * n + j) * 2 + 1] : 0.0;
  double X0 = has ? X[2 * (size_t)j] : 0.0, X1 = has ? X[2 * (size_t)j + 1] : 0.0, c0 = X0, c1 = X1, s0 = 0.0, s1 = 0.0;
  TrustRegionState tr{1e4, 2.0, 0};
  bool last_ok = true;
  const unsigned long long kPattern = 0xFFFFFFFFFFFFFFFFull;
  unsigned long long* xw = reinterpret_cast<unsigned long long*>(xch);
  unsigned long long* abort_word = xw + 7;      // (value 7 of row 0 of set 0 is never a sum: pattern = running)
  // The residuals at a candidate ARE the residuals of the next iteration when the candidate is accepted: one evaluation per view and iteration (with its
  // derivatives) instead of two - the lane keeps cost / normal equations of the current point (cur) and of the candidate (cnd); the values, their order of
  // summation inside the lane and so the bits are those of the two-evaluation loop.
  struct Local { double cost, h00, h01, h11, g0, g1; };
  auto evaluate = [&](double P0, double P1) {
    Local L{0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      double d[2];
      const double r = ResidualPoint2d(q[i], t[i], P0, P1, rt[i], d);
      L.cost += 0.5 * r * r;
      L.h00 += d[0] * d[0]; L.h01 += d[0] * d[1]; L.h11 += d[1] * d[1]; L.g0 += d[0] * r; L.g1 += d[1] * r;
    }
    return L;
  };
  Local cur{0, 0, 0, 0, 0, 0}, cnd{0, 0, 0, 0, 0, 0};
  if (has) { cur = evaluate(X0, X1); s0 = 1.0 / (1.0 + sqrt(cur.h00)); s1 = 1.0 / (1.0 + sqrt(cur.h11)); }
  int iter = 1;
  for (;; ++iter) {
    double s[7] = {0, 0, 0, 0, 0, 0, 0};     // cost, model, |step|^2, |x|^2, candidate cost, invalid count, gradient max
    if (has) {
      const double h00 = cur.h00, h01 = cur.h01, h11 = cur.h11, g0 = cur.g0, g1 = cur.g1;
      s[0] = cur.cost;
      s[6] = fmax(fabs(g0), fabs(g1));
      const double dg0 = fmin(fmax(s0 * s0 * h00, 1e-6), 1e32), dg1 = fmin(fmax(s1 * s1 * h11, 1e-6), 1e32);
      const double a = s0 * s0 * h00 + dg0 / tr.radius, b = s0 * s1 * h01, c = s1 * s1 * h11 + dg1 / tr.radius;
      const double det = a * c - b * b;
      if (!(det > 0.0)) s[5] = 1.0;
      else {
        const double r0 = -s0 * g0, r1 = -s1 * g1;
        const double inv_det = 1.0 / det;
        const double e0 = s0 * (c * r0 - b * r1) * inv_det, e1 = s1 * (a * r1 - b * r0) * inv_det;
        s[1] = -(g0 * e0 + g1 * e1 + 0.5 * (h00 * e0 * e0 + 2.0 * h01 * e0 * e1 + h11 * e1 * e1));
        s[2] = e0 * e0 + e1 * e1; s[3] = X0 * X0 + X1 * X1;
        c0 = X0 + e0; c1 = X1 + e1;
        cnd = evaluate(c0, c1);
        s[4] = cnd.cost;
      }
    }
    // wavefront: DPP moves leave every 16-lane row with its own sums (six) and maximum; lane k < 7 of every row hands value k to LDS - sixteen partial
    // values per quantity and workgroup, added up below in a fixed order (no v_readlane: they cost as much as the arithmetic of an iteration)
#pragma unroll
    for (int k = 0; k < 6; ++k) s[k] = RowSumDpp(s[k]);
    s[6] = RowMaxDpp(s[6]);
    if ((lane & 15) < 7) {
      double v = s[0];
#pragma unroll
      for (int k = 1; k < 7; ++k) v = (lane & 15) == k ? s[k] : v;
      wsum[(wv * 4 + (lane >> 4)) * 8 + (lane & 15)] = v;
    }
    __syncthreads();
    if (wv == 0) {
      // workgroup: the sixteen rows in order
      double mine = 0.0;
      if (lane < 6) { for (int r = 0; r < 16; ++r) mine += wsum[r * 8 + lane]; }
      else if (lane == 6) { for (int r = 0; r < 16; ++r) mine = fmax(mine, wsum[r * 8 + 6]); }
      if (G == 1) { if (lane < 7) tot[lane] = mine; }
      else {
        const int set = iter % 3;
        if (lane < 7) {
          unsigned long long bits = (unsigned long long)__double_as_longlong(mine);
          if (bits == kPattern) bits = 0x7FF8000000000000ull;      // a NaN sum stays a NaN, never the pattern
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // (this row's re-poisoning of two iterations ago has left; see ExchangeSums)
          __hip_atomic_store(xw + ((size_t)set * kPointsMaxGroups + blockIdx.x) * 8 + lane, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const unsigned long long* cur = xw + (size_t)set * kPointsMaxGroups * 8;
        // words lane and lane + 64 of the G rows (G <= 16: two per lane), both in flight; value 7 of a row is not polled
        const int w0 = lane, w1 = lane + 64;
        const bool p0 = w0 < G * 8 && (w0 & 7) != 7, p1 = w1 < G * 8 && (w1 & 7) != 7;
        unsigned long long b0 = 0, b1 = 0;
        for (int spins = 0;; ++spins) {
          if (p0) b0 = __hip_atomic_load(cur + w0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (p1) b1 = __hip_atomic_load(cur + w1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const bool missing = (p0 && b0 == kPattern) || (p1 && b1 == kPattern);
          if (!__any(missing)) break;
          if (spins >= (1 << 22)) { if (lane == 0) __hip_atomic_store(abort_word, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
          if ((spins & 255) == 255 && __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != kPattern) break;
        }
        // (a timeout / an abort leaves the NaN pattern in a row: the sums turn NaN, the loop ends as invalid - in every workgroup)
        if (w0 < G * 8) rows[w0] = p0 ? __longlong_as_double((long long)b0) : 0.0;
        if (w1 < G * 8) rows[w1] = p1 ? __longlong_as_double((long long)b1) : 0.0;
        // this wavefront has seen iteration `iter` of every peer: the peers have finished reading the rows of iteration iter - 1 = set (iter + 2) % 3
        if (lane < 7) __hip_atomic_store(xw + ((size_t)((iter + 2) % 3) * kPointsMaxGroups + blockIdx.x) * 8 + lane, kPattern, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0): the rows are in LDS (one wavefront wrote them, the same one reads them)
        __builtin_amdgcn_wave_barrier();
        if (lane < 6) { double tsum = 0.0; for (int g = 0; g < G; ++g) tsum += rows[8 * g + lane]; tot[lane] = tsum; }
        else if (lane == 6) { double tmax = 0.0; for (int g = 0; g < G; ++g) tmax = fmax(tmax, rows[8 * g + 6]); tot[6] = tmax; }
      }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 7; ++k) s[k] = tot[k];
    if (last_ok && s[6] <= kTol) break;
    if (iter > 50 || tr.radius < 1e-32) break;
    if (s[5] > 0.0 || !(s[1] > 0.0)) { if (++tr.invalid >= 5) break; tr.Reject(); last_ok = false; continue; }
    tr.invalid = 0;
    if (sqrt(s[2]) <= kTol * (sqrt(s[3]) + kTol)) break;
    const double change = s[0] - s[4];
    if (fabs(change) <= kTol * s[0]) break;
    const double rel = change / s[1];
    if (rel > 1e-3) { X0 = c0; X1 = c1; cur = cnd; tr.Accept(rel); last_ok = true; }
    else { tr.Reject(); last_ok = false; }
  }
  if (has) { X[2 * (size_t)j] = X0; X[2 * (size_t)j + 1] = X1; }
  if (iterations_out && blockIdx.x == 0 && threadIdx.x == 0) *iterations_out = iter;
}

// ---- bundle_adjust2d: cameras 1..3 and the m sample points -----------------------------------------------------------
// camera tangent columns: cam1 (q, t) = 0,1; cam2 (q, t0, t1) = 2,3,4; cam3 = 5,6,7.  One point's Jacobian pieces:
struct PointJac2d {
  double r[4];        // residuals in views 0..3
  double jc[4][3];    // camera-side tangent Jacobian of view i (cam1 uses 2 entries, cam2/3 use 3, cam0 none)
  double jp[4][2];    // point-side Jacobian of view i
};
__device__ __forceinline__ void EvalPoint2d(const double (*q)[2], const double (*t)[2], const double (*jq)[2], const double* jt1, double X0, double X1,
                                            const double* __restrict__ x, int n, int idx, PointJac2d* P) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    double d[6];
    P->r[i] = Residual2d(q[i], t[i], X0, X1, x[((size_t)i * n + idx) * 2], x[((size_t)i * n + idx) * 2 + 1], d);
    P->jp[i][0] = d[4]; P->jp[i][1] = d[5];
    P->jc[i][0] = d[0] * jq[i][0] + d[1] * jq[i][1];
    if (i == 1) { P->jc[i][1] = d[2] * jt1[0] + d[3] * jt1[1]; P->jc[i][2] = 0.0; }
    else { P->jc[i][1] = d[2]; P->jc[i][2] = d[3]; }
  }
}

// cams (24, in/out), sample (m indices into the n tracks), X (n x 2, the model's points; only the sample's entries are
// read and written), scratch scale_p / Xc (n x 2, indexed like X)
// Prologue (the launches a LeastSquares call no longer needs): the refined model is a NEW model - its cameras start as a copy of cams_src (24) and its
// points as a copy of Xsrc (n x 2; either may alias the destination) - and the exchange slots of the point kernel that follows in the stream are
// preset to their "not written yet" pattern (xch, xch_doubles; null: none).
// NW = 4: 256 lanes (samples of any size); NW = 1: ONE wavefront for samples of at most 64 points (every LeastSquaresFit of a local optimisation), whose sums go
// through WaveSumLds.  The two take their sums in different orders (equal to rounding); which one runs depends on the sample size alone, on every path.
template <int NW>
__global__ __launch_bounds__(64 * NW) void k_fv2d_bundle(int n, const double* __restrict__ x, int m, const int32_t* __restrict__ sample, const double* cams_src, double* cams,
                                                         const double* Xsrc, double* X, double* __restrict__ scale_p, double* __restrict__ Xc,
                                                         double* __restrict__ xch, int xch_doubles) {
  constexpr int kT = 64 * NW;
  if (Xsrc != X) {      // eight loads in flight per lane (the two arrays may be the same one: no restrict, so the compiler would wait for every store)
    for (int i0 = 0; i0 < 2 * n; i0 += 8 * kT) {
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { const int i = i0 + u * kT + (int)threadIdx.x; v[u] = i < 2 * n ? Xsrc[i] : 0.0; }
#pragma unroll
      for (int u = 0; u < 8; ++u) { const int i = i0 + u * kT + (int)threadIdx.x; if (i < 2 * n) X[i] = v[u]; }
    }
  }
  if (cams_src != cams && threadIdx.x < 24) cams[threadIdx.x] = cams_src[threadIdx.x];
  if (xch) for (int i = threadIdx.x; i < xch_doubles; i += kT) reinterpret_cast<unsigned long long*>(xch)[i] = 0xFFFFFFFFFFFFFFFFull;
  __syncthreads();      // (one workgroup: every copy above is visible to every thread below)
  if (m < 10) return;      // "only bundle when there are enough points to make it worthwhile" (sfm2d.cc:126-127)
  __shared__ double lds[NW == 1 ? 53 * 65 + 53 : 4 * 53];
  const int rows = m < 64 ? m : 64;      // (NW = 1: the lanes that hold a point)
  const double kTol = 1e-10;
  const int coff[4] = {0, 0, 2, 5}, cw[4] = {0, 2, 3, 3};
  double q[4][2], t[4][2], qc[4][2], tc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i) { q[i][0] = cams[6 * i]; q[i][1] = cams[6 * i + 3]; t[i][0] = cams[6 * i + 2]; t[i][1] = cams[6 * i + 5]; }
  double scale_c[8];
  TrustRegionState tr{1e4, 2.0, 0};
  bool last_ok = true, first = true;
  for (int iter = 1;; ++iter) {
    double jq[4][2], jt1[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) HomogeneousJacobian2(q[i], jq[i]);
    HomogeneousJacobian2(t[1], jt1);
    // ---- pass 0 (first iteration only): squared column norms for the Jacobi scaling
    if (first) {
      double cn[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (int e = threadIdx.x; e < m; e += kT) {
        const int idx = sample[e];
        PointJac2d P;
        EvalPoint2d(q, t, jq, jt1, X[2 * (size_t)idx], X[2 * (size_t)idx + 1], x, n, idx, &P);
        double v0 = 0, v1 = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          v0 += P.jp[i][0] * P.jp[i][0]; v1 += P.jp[i][1] * P.jp[i][1];
#pragma unroll
          for (int a = 0; a < 3; ++a) if (a < cw[i]) cn[coff[i] + a] += P.jc[i][a] * P.jc[i][a];
        }
        scale_p[2 * (size_t)idx] = 1.0 / (1.0 + sqrt(v0)); scale_p[2 * (size_t)idx + 1] = 1.0 / (1.0 + sqrt(v1));
      }
      GroupSumN<NW>(cn, lds, rows);
#pragma unroll
      for (int a = 0; a < 8; ++a) scale_c[a] = 1.0 / (1.0 + sqrt(cn[a]));
      first = false;
    }
    // ---- pass 1: cost, gradient, reduced camera system for the current radius
    // acc: [0] cost, [1..8] g_c, [9..16] Schur rhs correction, [17..52] lower triangle of (U_s - Schur correction)
    double acc[53];
#pragma unroll
    for (int k = 0; k < 53; ++k) acc[k] = 0.0;
    double gpmax = 0.0, ucn[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int e = threadIdx.x; e < m; e += kT) {
      const int idx = sample[e];
      PointJac2d P;
      EvalPoint2d(q, t, jq, jt1, X[2 * (size_t)idx], X[2 * (size_t)idx + 1], x, n, idx, &P);
      const double s0 = scale_p[2 * (size_t)idx], s1 = scale_p[2 * (size_t)idx + 1];
      double v00 = 0, v01 = 0, v11 = 0, g0 = 0, g1 = 0;
      double W[8][2];
#pragma unroll
      for (int a = 0; a < 8; ++a) { W[a][0] = 0.0; W[a][1] = 0.0; }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc[0] += 0.5 * P.r[i] * P.r[i];
        v00 += P.jp[i][0] * P.jp[i][0]; v01 += P.jp[i][0] * P.jp[i][1]; v11 += P.jp[i][1] * P.jp[i][1];
        g0 += P.jp[i][0] * P.r[i]; g1 += P.jp[i][1] * P.r[i];
#pragma unroll
        for (int a = 0; a < 3; ++a)
          if (a < cw[i]) {
            const int ca = coff[i] + a;
            const double js = P.jc[i][a] * scale_c[ca];
            acc[1 + ca] += P.jc[i][a] * P.r[i];
            ucn[ca] += P.jc[i][a] * P.jc[i][a];
            W[ca][0] += js * P.jp[i][0] * s0; W[ca][1] += js * P.jp[i][1] * s1;
#pragma unroll
            for (int b = 0; b <= a; ++b) {     // U is block diagonal by camera
              const int cb = coff[i] + b;
              acc[17 + ca * (ca + 1) / 2 + cb] += js * P.jc[i][b] * scale_c[cb];
            }
          }
      }
      gpmax = fmax(gpmax, fmax(fabs(g0), fabs(g1)));
      const double dg0 = fmin(fmax(s0 * s0 * v00, 1e-6), 1e32), dg1 = fmin(fmax(s1 * s1 * v11, 1e-6), 1e32);
      const double a_ = s0 * s0 * v00 + dg0 / tr.radius, b_ = s0 * s1 * v01, c_ = s1 * s1 * v11 + dg1 / tr.radius;
      const double det = a_ * c_ - b_ * b_;
      const double i00 = c_ / det, i01 = -b_ / det, i11 = a_ / det;
      const double gs0 = s0 * g0, gs1 = s1 * g1;
      const double y0 = i00 * gs0 + i01 * gs1, y1 = i01 * gs0 + i11 * gs1;     // Vd^-1 g_p,s
#pragma unroll
      for (int a = 0; a < 8; ++a) {
        const double wa0 = W[a][0] * i00 + W[a][1] * i01, wa1 = W[a][0] * i01 + W[a][1] * i11;   // (W Vd^-1)[a]
        acc[9 + a] += W[a][0] * y0 + W[a][1] * y1;
#pragma unroll
        for (int b = 0; b <= a; ++b) acc[17 + a * (a + 1) / 2 + b] -= wa0 * W[b][0] + wa1 * W[b][1];
      }
    }
    GroupSumN<NW>(acc, lds, rows);
    GroupSumN<NW>(ucn, lds, rows);
    gpmax = GroupMax<NW>(gpmax, lds);
    // gradient max-norm: ||x - Plus(x, -g)||_inf over the blocks
    double gmax = gpmax;
#pragma unroll
    for (int i = 1; i < 4; ++i) {
      double o[2];
      HomogeneousPlus2(q[i], -acc[1 + coff[i]], o);
      gmax = fmax(gmax, fmax(fabs(q[i][0] - o[0]), fabs(q[i][1] - o[1])));
      if (i == 1) { HomogeneousPlus2(t[1], -acc[1 + 1], o); gmax = fmax(gmax, fmax(fabs(t[1][0] - o[0]), fabs(t[1][1] - o[1]))); }
      else gmax = fmax(gmax, fmax(fabs(acc[1 + coff[i] + 1]), fabs(acc[1 + coff[i] + 2])));
    }
    if (last_ok && gmax <= kTol) break;
    if (iter > 50 || tr.radius < 1e-32) break;
    // reduced system (every lane solves it): S = U_s + D_c - corr, rhs = -g_c,s + corr_rhs
    double S[8][8], rhs[8], dc[8];
    bool valid = true;
#pragma unroll
    for (int a = 0; a < 8; ++a) {
#pragma unroll
      for (int b = 0; b <= a; ++b) S[a][b] = acc[17 + a * (a + 1) / 2 + b];
      S[a][a] += fmin(fmax(scale_c[a] * scale_c[a] * ucn[a], 1e-6), 1e32) / tr.radius;
      rhs[a] = -scale_c[a] * acc[1 + a] + acc[9 + a];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {       // Cholesky, then forward and back substitution
      double dgn = S[j][j];
#pragma unroll
      for (int k = 0; k < j; ++k) dgn -= S[j][k] * S[j][k];
      if (!(dgn > 0.0)) { valid = false; dgn = 1.0; }
      const double l = sqrt(dgn);
      S[j][j] = l;
#pragma unroll
      for (int i2 = j + 1; i2 < 8; ++i2) {
        double v = S[i2][j];
#pragma unroll
        for (int k = 0; k < j; ++k) v -= S[i2][k] * S[j][k];
        S[i2][j] = v / l;
      }
    }
#pragma unroll
    for (int a = 0; a < 8; ++a) { double v = rhs[a]; for (int k = 0; k < a; ++k) v -= S[a][k] * dc[k]; dc[a] = v / S[a][a]; }
#pragma unroll
    for (int a = 7; a >= 0; --a) { double v = dc[a]; for (int k = a + 1; k < 8; ++k) v -= S[k][a] * dc[k]; dc[a] = v / S[a][a]; }
    double del_c[8];
#pragma unroll
    for (int a = 0; a < 8; ++a) del_c[a] = scale_c[a] * dc[a];
    // candidate cameras
#pragma unroll
    for (int i = 0; i < 4; ++i) { qc[i][0] = q[i][0]; qc[i][1] = q[i][1]; tc[i][0] = t[i][0]; tc[i][1] = t[i][1]; }
#pragma unroll
    for (int i = 1; i < 4; ++i) {
      HomogeneousPlus2(q[i], del_c[coff[i]], qc[i]);
      if (i == 1) HomogeneousPlus2(t[1], del_c[1], tc[1]);
      else { tc[i][0] = t[i][0] + del_c[coff[i] + 1]; tc[i][1] = t[i][1] + del_c[coff[i] + 2]; }
    }
    // ---- pass 2: point steps, model cost change, candidate cost
    double s2[4] = {0, 0, 0, 0};    // model, |step_p|^2, |x_p|^2, candidate cost
    for (int e = threadIdx.x; e < m; e += kT) {
      const int idx = sample[e];
      const double X0 = X[2 * (size_t)idx], X1 = X[2 * (size_t)idx + 1];
      PointJac2d P;
      EvalPoint2d(q, t, jq, jt1, X0, X1, x, n, idx, &P);
      const double s0 = scale_p[2 * (size_t)idx], s1 = scale_p[2 * (size_t)idx + 1];
      double v00 = 0, v01 = 0, v11 = 0, g0 = 0, g1 = 0, w0 = 0, w1 = 0;
      double jd[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        v00 += P.jp[i][0] * P.jp[i][0]; v01 += P.jp[i][0] * P.jp[i][1]; v11 += P.jp[i][1] * P.jp[i][1];
        g0 += P.jp[i][0] * P.r[i]; g1 += P.jp[i][1] * P.r[i];
        double jc_d = 0.0;
#pragma unroll
        for (int a = 0; a < 3; ++a) if (a < cw[i]) jc_d += P.jc[i][a] * del_c[coff[i] + a];
        jd[i] = jc_d;
        w0 += P.jp[i][0] * jc_d; w1 += P.jp[i][1] * jc_d;      // J_p^T J_c delta_c
      }
      const double dg0 = fmin(fmax(s0 * s0 * v00, 1e-6), 1e32), dg1 = fmin(fmax(s1 * s1 * v11, 1e-6), 1e32);
      const double a_ = s0 * s0 * v00 + dg0 / tr.radius, b_ = s0 * s1 * v01, c_ = s1 * s1 * v11 + dg1 / tr.radius;
      const double det = a_ * c_ - b_ * b_;
      const double r0 = -s0 * (g0 + w0), r1 = -s1 * (g1 + w1);
      const double e0 = s0 * (c_ * r0 - b_ * r1) / det, e1 = s1 * (a_ * r1 - b_ * r0) / det;
#pragma unroll
      for (int i = 0; i < 4; ++i) { const double j_d = jd[i] + P.jp[i][0] * e0 + P.jp[i][1] * e1; s2[0] -= j_d * (P.r[i] + 0.5 * j_d); }
      s2[1] += e0 * e0 + e1 * e1; s2[2] += X0 * X0 + X1 * X1;
      const double c0 = X0 + e0, c1 = X1 + e1;
      Xc[2 * (size_t)idx] = c0; Xc[2 * (size_t)idx + 1] = c1;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const double r = Residual2d(qc[i], tc[i], c0, c1, x[((size_t)i * n + idx) * 2], x[((size_t)i * n + idx) * 2 + 1], nullptr);
        s2[3] += 0.5 * r * r;
      }
    }
    GroupSumN<NW>(s2, lds, rows);
    if (!valid || !(s2[0] > 0.0)) { if (++tr.invalid >= 5) break; tr.Reject(); last_ok = false; continue; }
    tr.invalid = 0;
    double sn = s2[1], xn = s2[2];
#pragma unroll
    for (int a = 0; a < 8; ++a) sn += del_c[a] * del_c[a];
#pragma unroll
    for (int i = 1; i < 4; ++i) xn += q[i][0] * q[i][0] + q[i][1] * q[i][1] + t[i][0] * t[i][0] + t[i][1] * t[i][1];
    if (sqrt(sn) <= kTol * (sqrt(xn) + kTol)) break;
    const double change = acc[0] - s2[3];
    if (fabs(change) <= kTol * acc[0]) break;
    const double rel = change / s2[0];
    if (rel > 1e-3) {
#pragma unroll
      for (int i = 1; i < 4; ++i) { q[i][0] = qc[i][0]; q[i][1] = qc[i][1]; t[i][0] = tc[i][0]; t[i][1] = tc[i][1]; }
      for (int e = threadIdx.x; e < m; e += kT) { const int idx = sample[e]; X[2 * (size_t)idx] = Xc[2 * (size_t)idx]; X[2 * (size_t)idx + 1] = Xc[2 * (size_t)idx + 1]; }
      __syncthreads();
      tr.Accept(rel); last_ok = true;
    } else { tr.Reject(); last_ok = false; }
  }
  if (threadIdx.x == 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      cams[6 * i] = q[i][0]; cams[6 * i + 1] = -q[i][1]; cams[6 * i + 3] = q[i][1]; cams[6 * i + 4] = q[i][0];
      cams[6 * i + 2] = t[i][0]; cams[6 * i + 5] = t[i][1];
    }
  }
}

}  // namespace ppsfm

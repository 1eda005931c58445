// Building blocks of the blocked fp64 Cholesky on 64x64 LDS tiles (row stride kLS), shared by the dense factorisation (cholesky.hip)
// and k_small_cholesky: the 16-column register panel, the in-block trailing
// update and tile inverses on the matrix cores (PotrfPanels), tile <-> LDS moves, the product-form triangular solve (SolveTile) and
// the rank-64 tile updates.  See cholesky.hip for the measurements behind each of them.
#pragma once
#include <type_traits>
#include <utility>

#include "ba_impl.hpp"

namespace ppsfm {

constexpr int kNB = 64;
constexpr int kDeferAbove = 280;      // super-tiles in a launch above which the far ones are visited every other launch with two panels
                                      // (measured: pays from ~T = 50 block columns on; at T = 47 the two-pass visit costs what it saves)
constexpr int kNumCUs = 256;          // MI355X; only used to size the tile-queue part of a launch
constexpr int kPanelThreads = 1024;   // 16 wavefronts: one 16x16 tile of a 64x64 block per wavefront
typedef double v4f64 __attribute__((ext_vector_type(4)));

// phase stamps for tools/chol_phase_bench.hip (compiled out of the library)
#ifdef PP_CHOL_TRACE
__device__ long long g_chol_trace[32];
__device__ long long g_chol_trace2[32][128];   // the same stamps per step of the task mode's chain (g_chol_step = its current step)
__device__ int g_chol_step;
#define PP_CHOL_PHASE(i) do { if (threadIdx.x == 0 && blockIdx.x == 0) { const long long t_ = wall_clock64(); g_chol_trace[i] = t_; g_chol_trace2[i][g_chol_step & 127] = t_; } } while (0)
#define PP_CHOL_STAMP(i) do { if (threadIdx.x == 0) g_chol_trace[i] = wall_clock64(); } while (0)
// switches for timing experiments on the task mode's chain run ALONE (tools/chol_task_trace.hip ... iso; results are garbage then)
__device__ int g_chol_exp;
#define PP_EXP(bit) (g_chol_exp & (bit))
// arrival of every wavefront of the chain workgroup at the barriers of PotrfPanels (last step / launch wins)
__device__ long long g_wave_arrive[12][16];
__device__ long long g_bs_trace[5][128];      // back substitution: per block / pair (index = lowest block): entry, far terms done, newest input seen, published
#define PP_BS_STAMP(slot, blk) do { if (threadIdx.x == 0) g_bs_trace[slot][(blk) & 127] = wall_clock64(); } while (0)
__device__ int g_arrive_step = -1;      // >= 0: only that step of the task mode's chain is recorded
#define PP_WAVE_ARRIVE(b) do { if ((threadIdx.x & 63) == 0 && blockIdx.x == 0 && (g_arrive_step < 0 || g_arrive_step == g_chol_step)) g_wave_arrive[b][threadIdx.x >> 6] = wall_clock64(); } while (0)
// per launch k: chain entry / exit and the latest exit of any workgroup
__device__ long long g_chol_launch[3][64];
#define PP_CHOL_LAUNCH(slot, k) do { if (threadIdx.x == 0 && (k) < 64) atomicMax((unsigned long long*)&g_chol_launch[slot][k], (unsigned long long)wall_clock64()); } while (0)
// role mask for timing experiments: bit 0 chain, 1 prep pair, 2 triangular solves, 3 trailing update (results are garbage then)
__device__ int g_chol_skip;
#define PP_CHOL_SKIPPED(bit) (g_chol_skip & (1 << (bit)))
// task mode (tools/chol_task_trace.hip): per step k, slot -> latest (max) or earliest (min) stamp over the workgroups that hit it
__device__ long long g_task_trace[24][128];
__device__ int g_dbg_mismatch[16];
__device__ int g_burn_stop;               // contention experiment (see k_cholesky_tasks): set by the chain when it is done
__device__ unsigned g_burn_hwid[512];    // HW_ID | XCC_ID << 16 of the chain (slot 0) and of the busy workgroups
__device__ long long g_spare_wait[3][128];   // chain, per step: ticks wavefront 4 waited after the last panel; state of the two fetches when it got there (X * 4 + D: 1 in flight, 2 in LDS); when
__device__ unsigned long long g_wait_missing[128];   // front update of step k: slots (bits 0-4: ver, rows 2I, 2I+1, 2J, 2J+1) its last polling round still waited for | rounds << 8
__device__ long long g_chain_phase[8][128];     // chain workgroup, thread 0: phase boundaries of step k
__device__ long long g_chain_clk[128];           // shader-clock counter at the start of step k (with the 100 MHz stamps: the clock the chain runs at)
#define PP_CHAIN_PHASE(slot, k) do { if (threadIdx.x == 0 && (k) < 128) { g_chain_phase[slot][k] = wall_clock64(); if ((slot) == 0) g_chain_clk[k] = clock64(); } } while (0)
#define PP_TASK_MAX(slot, k) do { if (threadIdx.x == 0 && (k) < 128) atomicMax((unsigned long long*)&g_task_trace[slot][k], (unsigned long long)wall_clock64()); } while (0)
#define PP_TASK_MIN(slot, k) do { if (threadIdx.x == 0 && (k) < 128) atomicMin((unsigned long long*)&g_task_trace[slot][k], (unsigned long long)wall_clock64()); } while (0)
#ifdef PP_CHOL_NO_STAMPS      // the variables stay (the tools read them), the stamps go: the chain at its production speed
#undef PP_CHOL_PHASE
#undef PP_CHOL_STAMP
#undef PP_WAVE_ARRIVE
#undef PP_CHOL_LAUNCH
#undef PP_CHAIN_PHASE
#undef PP_TASK_MAX
#undef PP_TASK_MIN
#define PP_TASK_MAX(slot, k) do { } while (0)
#define PP_TASK_MIN(slot, k) do { } while (0)
#define PP_CHAIN_PHASE(slot, k) do { } while (0)
#define PP_CHOL_LAUNCH(slot, k) do { } while (0)
#define PP_CHOL_PHASE(i) do { } while (0)
#define PP_CHOL_STAMP(i) do { } while (0)
#define PP_WAVE_ARRIVE(b) do { } while (0)
#endif
#else
#define PP_BS_STAMP(slot, blk) do { } while (0)
#define PP_TASK_MAX(slot, k) do { } while (0)
#define PP_TASK_MIN(slot, k) do { } while (0)
#define PP_CHAIN_PHASE(slot, k) do { } while (0)
#define PP_CHOL_LAUNCH(slot, k) do { } while (0)
#define PP_CHOL_SKIPPED(bit) false
#define PP_CHOL_PHASE(i) do { } while (0)
#define PP_CHOL_STAMP(i) do { } while (0)
#define PP_WAVE_ARRIVE(b) do { } while (0)
#define PP_EXP(bit) false
#endif

// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double ReadLane(double v, int src_lane) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, src_lane);
  hi = __builtin_amdgcn_readlane(hi, src_lane);
  return __hiloint2double(hi, lo);
}

// acc + sum_kk a[kk] x b[kk] over 16 k-slices of a 16x16 tile product.  A dependent v_mfma_f64_16x16x4 (same
// accumulator) was measured at 78 ns on MI355X against ~27 ns issue, so the K loop runs on FOUR independent
// partial accumulators that are summed at the end instead of one 16-deep dependent chain.
__device__ __forceinline__ v4f64 MfmaK16(const double (&a)[16], const double (&b)[16], v4f64 acc) {
  v4f64 p1 = (v4f64){0.0, 0.0, 0.0, 0.0}, p2 = p1, p3 = p1;
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[kk], b[kk], acc, 0, 0, 0);
    p1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[4 + kk], b[4 + kk], p1, 0, 0, 0);
    p2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[8 + kk], b[8 + kk], p2, 0, 0, 0);
    p3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[12 + kk], b[12 + kk], p3, 0, 0, 0);
  }
  return (acc + p1) + (p2 + p3);
}
// the same for a K = 4-slice product (one 16x16x16): four single MFMAs, no dependent pair
__device__ __forceinline__ v4f64 MfmaK4(const double (&a)[4], const v4f64& b, v4f64 acc) {
  const v4f64 z = (v4f64){0.0, 0.0, 0.0, 0.0};
  const v4f64 p0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0], b[0], acc, 0, 0, 0);
  const v4f64 p1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[1], b[1], z, 0, 0, 0);
  const v4f64 p2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[2], b[2], z, 0, 0, 0);
  const v4f64 p3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[3], b[3], z, 0, 0, 0);
  return (p0 + p1) + (p2 + p3);
}

// 1 / sqrt(d) for a finite positive d: v_rsq_f64 and the device library's one correction step, without its
// v_cmp_class / v_cndmask pair for 0 and inf (v_cndmask issues at a third of the fma rate and sits on the pivot chain)
__device__ __forceinline__ double RsqrtPositive(double d) {
  const double r = __builtin_amdgcn_rsq(d);
  const double e = fma(-d * r, r, 1.0);
  return fma(r * e, fma(e, 0.375, 0.5), r);
}

constexpr int kLS = kNB + 2;  // LDS row stride (doubles): conflict-free for the MFMA operand pattern

// one 16-column panel of the 64x64 diagonal block, unblocked and entirely in the registers of ONE
// wavefront (lane = row): the pivot and the multipliers travel by v_readlane, no barrier, no LDS
template <int P>
__device__ __forceinline__ void PotrfPanel16(double* A, double* inv_diag, int lane, int32_t* flag) {
  constexpr int c0 = 16 * P;
  double a[16];
#pragma unroll
  for (int jj = 0; jj < 16; ++jj) a[jj] = A[lane * kLS + c0 + jj];
  // The 16 columns are ONE basic block (no pivot branch, no special-case selects), so the scheduler starts column
  // jj+1's pivot / rsqrt under column jj's updates: 2.5 -> 1.9 us per panel.  A non-positive or NaN pivot gives a
  // NaN reciprocal root (RsqrtPositive) that makes column jj, every column it updates and so every later pivot of the
  // panel NaN; it is detected once, afterwards, from the LAST reciprocal root.
  double invs[16];
#pragma unroll
  for (int jj = 0; jj < 16; ++jj) {
    const double d = ReadLane(a[jj], c0 + jj);
    const double inv = RsqrtPositive(d);
    a[jj] *= inv;                       // lane c0+jj now holds sqrt(d)
    invs[jj] = inv;
#pragma unroll
    for (int cc = jj + 1; cc < 16; ++cc) {
      const double s = ReadLane(a[jj], c0 + cc);
      a[cc] = fma(-a[jj], s, a[cc]);
    }
  }
  if (lane == 0) {
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) inv_diag[c0 + jj] = invs[jj];
    if (!(invs[15] < 1.7976931348623157e308)) atomicOr(flag, 1);
  }
  if (lane >= c0) {
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) A[lane * kLS + c0 + jj] = a[jj];
  }
}

// (Measured and dropped, round 3: the multipliers through LDS instead of v_readlane - a finished column stored with one ds_write_b64 and
// read back as uniform-address broadcasts, the next column's multiplier still by v_readlane.  Straight: 2.08 us per panel (every column
// waits out the LDS round trip); software-pipelined one column deep: 1.72-1.9 us against 1.64 us - the LDS instructions cost the
// wavefront as many issue slots as the v_readlane pairs they replace.  tools/ab_phase.sh.)
// The LAST panel (no rows below the diagonal tile) also builds the INVERSE of its 16x16 tile in the same pass, by
// column-oriented forward substitution: step jj needs column jj of the factor, i.e. exactly the multipliers the panel update
// uses anyway, so it costs one extra fma per (jj, cc) pair; the 16-step substitution (InverseDiag16, 1.8 us on one wavefront)
// that followed the last panel on the chain's critical path is gone.  The earlier panels keep the separate inverse (it runs
// beside the next panel there; folded into all four panels it cost more than it saved).
// Layout: lane l holds row l & 15 of the tile (t) and column l & 15 of its inverse (x) - all four 16-lane rows of the wavefront
// alike, which costs no instruction and makes every pivot and multiplier a DPP row broadcast (row_newbcast, lane N of each
// 16-lane row: the one DPP control the 64-bit VALU has on gfx950; v_mov_b64 and v_fmac_f64 take it).  A multiplier then costs
// no instruction of its own: 2 x 3.5 ns per pair against 11.4 ns with a v_readlane pair feeding two fmas (a wavefront alone
// on its SIMD issues one fp64 instruction per 3.5 ns, tools/valu_rate_bench.hip): 2.44 -> 2.12 us for this panel.  With rows
// below the tile the same layout needs a second register set for them (two fmacs per pair: no gain over v_readlane, measured),
// so panels 0-2 keep lane = row.  A DPP read needs 2 wait states after the VALU write of its source and the compiler does not
// see inside inline asm: the s_nop travels with the producer (ScaleForBroadcast) or with the read (RowBroadcast).
template <int N>
__device__ __forceinline__ double RowBroadcast(double v) {
  double r;
  asm("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v), "n"(N));
  return r;
}
__device__ __forceinline__ double ScaleForBroadcast(double v, double s) {
  double r;
  asm("v_mul_f64 %0, %1, %2\n\ts_nop 1" : "=v"(r) : "v"(v), "v"(s));
  return r;
}
template <int N, bool kFreshSource = false>      // kFreshSource: bsrc may have been written by the instruction just before (2 wait states)
__device__ __forceinline__ void SubMulRowBroadcast(double& acc, double bsrc, double own) {   // acc -= (lane N of the row's bsrc) * own
  if (kFreshSource) asm("s_nop 1\n\tv_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(bsrc), "v"(own), "n"(N));
  else asm("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(bsrc), "v"(own), "n"(N));
}
template <int JJ, int... CC>
__device__ __forceinline__ double LastPanelColumn(double (&t)[16], double (&x)[16], std::integer_sequence<int, CC...>) {
  const double inv = RsqrtPositive(RowBroadcast<JJ>(t[JJ]));
  t[JJ] = ScaleForBroadcast(t[JJ], inv);      // lane JJ of each row now holds sqrt(d)
  x[JJ] *= inv;                               // row JJ of T^-1, this lane's column
  ((SubMulRowBroadcast<JJ + 1 + CC>(t[JJ + 1 + CC], t[JJ], t[JJ]), SubMulRowBroadcast<JJ + 1 + CC>(x[JJ + 1 + CC], t[JJ], x[JJ])), ...);
  return inv;
}
template <int... JJ>
__device__ __forceinline__ double LastPanelColumns(double (&t)[16], double (&x)[16], std::integer_sequence<int, JJ...>) {
  double inv = 0.0;
  ((inv = LastPanelColumn<JJ>(t, x, std::make_integer_sequence<int, 15 - JJ>())), ...);
  return inv;
}
__device__ __forceinline__ void PotrfLastPanelWithInverse(double* A, double* M, int lane, int32_t* flag) {
  constexpr int c0 = 48;
  const int r = lane & 15;
  double t[16], x[16];
#pragma unroll
  for (int jj = 0; jj < 16; ++jj) { t[jj] = A[(c0 + r) * kLS + c0 + jj]; x[jj] = (r == jj) ? 1.0 : 0.0; }
  const double last_inv = LastPanelColumns(t, x, std::make_integer_sequence<int, 16>());   // NaN if any pivot was bad (see PotrfPanel16)
  if (lane < 16) {
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) { A[(c0 + r) * kLS + c0 + jj] = t[jj]; M[(c0 + jj) * kLS + c0 + r] = x[jj]; }
  }
  if (lane == 0 && !(last_inv < 1.7976931348623157e308)) atomicOr(flag, 1);
}

// inverse of the factored diagonal 16x16 tile P into tile (P,P) of M by column-oriented forward substitution on ONE
// wavefront: lane l carries column l & 15 of T^-1 (16 running sums, so each step's dependent chain is one multiply + one
// fma) and row l & 15 of the tile; the multiplier L[r][q] of a step is lane r of the register holding column q - a DPP row
// broadcast inside the fma (see above): 136 fp64 instructions, ~0.6 us (1.8 us with a v_readlane pair per multiplier).
template <int Q, int... R>
__device__ __forceinline__ void InverseStep(const double (&a)[16], double (&x)[16], double inv_q, std::integer_sequence<int, R...>) {
  x[Q] *= inv_q;
  // a[Q] comes straight from an LDS load; the leading s_nop of the first use covers a register copy the compiler might place before it
  (SubMulRowBroadcast<Q + 1 + R, R == 0>(x[Q + 1 + R], a[Q], x[Q]), ...);
}
template <int... Q>
__device__ __forceinline__ void InverseSteps(const double (&a)[16], double (&x)[16], const double (&inv)[16], std::integer_sequence<int, Q...>) {
  (InverseStep<Q>(a, x, inv[Q], std::make_integer_sequence<int, 15 - Q>()), ...);
}
template <int P>
__device__ __forceinline__ void InverseDiag16(const double* A, const double* inv_diag, double* M, int lane) {
  constexpr int t0 = 16 * P;
  const int r = lane & 15;
  double a[16], x[16], inv[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) { a[q] = A[(t0 + r) * kLS + t0 + q]; inv[q] = inv_diag[t0 + q]; x[q] = (q == r) ? 1.0 : 0.0; }
  InverseSteps(a, x, inv, std::make_integer_sequence<int, 16>());
  if (lane < 16) {
#pragma unroll
    for (int q = 0; q < 16; ++q) M[(t0 + q) * kLS + t0 + r] = x[q];
  }
}

// 16x16x16 tile products on LDS tiles (row stride kLS).  Operand conventions of v_mfma_f64_16x16x4:
//   A operand: lane l holds A[l&15][4kk + (l>>4)];  B operand: lane l holds B[4kk + (l>>4)][l&15];  D: reg i <-> C[(l>>4)+4i][l&15]
__device__ __forceinline__ v4f64 TileMulAB(const double* At, const double* Bt, v4f64 acc, int lr, int g) {   // acc + At * Bt
  double av[4];
  v4f64 bv;
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) { av[kk] = At[lr * kLS + 4 * kk + g]; bv[kk] = Bt[(4 * kk + g) * kLS + lr]; }
  return MfmaK4(av, bv, acc);
}
__device__ __forceinline__ v4f64 TileMulABt(const double* At, const double* Bt, v4f64 acc, int lr, int g) {  // acc + At * Bt^T
  double av[4];
  v4f64 bv;
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) { av[kk] = At[lr * kLS + 4 * kk + g]; bv[kk] = Bt[lr * kLS + 4 * kk + g]; }
  return MfmaK4(av, bv, acc);
}
__device__ __forceinline__ v4f64 TileNegMulAD(const double* At, const v4f64& d, int lr, int g) {   // -(At * d), d in the D layout
  double av[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) av[kk] = -At[lr * kLS + 4 * kk + g];
  return MfmaK4(av, d, (v4f64){0.0, 0.0, 0.0, 0.0});
}
__device__ __forceinline__ void TileStoreD(double* Ct, const v4f64& c, int lr, int g) {
#pragma unroll
  for (int i = 0; i < 4; ++i) Ct[(g + 4 * i) * kLS + lr] = c[i];
}
#define PP_TILE(buf, ti, tj) ((buf) + (16 * (ti)) * kLS + 16 * (tj))

// rank-16 update of the 16x16 tiles right of panel P on the matrix cores, one tile per wavefront
template <int P>
__device__ __forceinline__ void PotrfTrailing16(double* A, int lane, int w) {
  constexpr int ntile = (3 - P) * (4 - P) / 2;
  if (w >= ntile) return;
  const int lr = lane & 15, g = lane >> 4;
  // enumerate (ti, tj), P < tj <= ti <= 3, row by row
  int ti = P + 1, rem = w;
  while (rem > ti - (P + 1)) { rem -= ti - P; ++ti; }
  const int tj = P + 1 + rem;
  v4f64 acc;
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = PP_TILE(A, ti, tj)[(g + 4 * i) * kLS + lr];
  double av[4];
  v4f64 bv;
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) { av[kk] = -PP_TILE(A, ti, P)[lr * kLS + 4 * kk + g]; bv[kk] = PP_TILE(A, tj, P)[lr * kLS + 4 * kk + g]; }
  acc = MfmaK4(av, bv, acc);
  TileStoreD(PP_TILE(A, ti, tj), acc, lr, g);
}

// 64x64 tile <-> LDS (row stride kLS), 16-byte global accesses, every load in flight before the first LDS store
__device__ __forceinline__ double2 TileLoad2(const double* __restrict__ src, int ld, int tid, int it) {
  const int idx = tid + kPanelThreads * it, r = idx >> 5, c2 = idx & 31;
  return *reinterpret_cast<const double2*>(src + (size_t)r * ld + 2 * c2);
}
__device__ __forceinline__ void TileStore2(double* dst, int tid, int it, double2 v) {
  const int idx = tid + kPanelThreads * it, r = idx >> 5, c2 = idx & 31;
  *reinterpret_cast<double2*>(dst + r * kLS + 2 * c2) = v;
}
// The same 16 bytes read COHERENTLY at agent scope (two 8-byte sc1 loads: they are served below the per-XCD L2s, which are not
// coherent with each other inside a kernel).  Task mode (k_cholesky_tasks) reads every tile that another workgroup of the SAME
// launch has rewritten this way; tiles that are written exactly once per launch (the solved tiles in their own array, the M_k,
// the per-step staging tiles) cannot be stale in any L2 and keep the plain, L2-cached loads.
__device__ __forceinline__ double LoadCoherent(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <bool kCoh>
__device__ __forceinline__ double2 TileLoad2T(const double* __restrict__ src, int ld, int tid, int it) {
  if (!kCoh) return TileLoad2(src, ld, tid, it);
  const int idx = tid + kPanelThreads * it, r = idx >> 5, c2 = idx & 31;
  const double* p = src + (size_t)r * ld + 2 * c2;
  return make_double2(LoadCoherent(p), LoadCoherent(p + 1));
}
template <bool kCoh>
__device__ __forceinline__ void LoadTileT(double* dst, const double* __restrict__ src, int ld, int tid) {
  const double2 a0 = TileLoad2T<kCoh>(src, ld, tid, 0), a1 = TileLoad2T<kCoh>(src, ld, tid, 1);
  TileStore2(dst, tid, 0, a0); TileStore2(dst, tid, 1, a1);
}
template <bool kCoh>
__device__ __forceinline__ double LoadS(const double* p) { return kCoh ? LoadCoherent(p) : *p; }
__device__ __forceinline__ void LoadTile(double* dst, const double* __restrict__ src, int ld, int tid) {
  const double2 a0 = TileLoad2(src, ld, tid, 0), a1 = TileLoad2(src, ld, tid, 1);
  TileStore2(dst, tid, 0, a0); TileStore2(dst, tid, 1, a1);
}
__device__ __forceinline__ void LoadTiles2(double* d0, const double* __restrict__ s0, double* d1, const double* __restrict__ s1, int ld, int tid) {
  const double2 a0 = TileLoad2(s0, ld, tid, 0), a1 = TileLoad2(s0, ld, tid, 1), b0 = TileLoad2(s1, ld, tid, 0), b1 = TileLoad2(s1, ld, tid, 1);
  TileStore2(d0, tid, 0, a0); TileStore2(d0, tid, 1, a1); TileStore2(d1, tid, 0, b0); TileStore2(d1, tid, 1, b1);
}
__device__ __forceinline__ void LoadTiles4(double* d0, const double* __restrict__ s0, double* d1, const double* __restrict__ s1, double* d2,
                                           const double* __restrict__ s2, double* d3, const double* __restrict__ s3, int ld, int tid) {
  const double2 a0 = TileLoad2(s0, ld, tid, 0), a1 = TileLoad2(s0, ld, tid, 1), b0 = TileLoad2(s1, ld, tid, 0), b1 = TileLoad2(s1, ld, tid, 1);
  const double2 c0 = TileLoad2(s2, ld, tid, 0), c1 = TileLoad2(s2, ld, tid, 1), e0 = TileLoad2(s3, ld, tid, 0), e1 = TileLoad2(s3, ld, tid, 1);
  TileStore2(d0, tid, 0, a0); TileStore2(d0, tid, 1, a1); TileStore2(d1, tid, 0, b0); TileStore2(d1, tid, 1, b1);
  TileStore2(d2, tid, 0, c0); TileStore2(d2, tid, 1, c1); TileStore2(d3, tid, 0, e0); TileStore2(d3, tid, 1, e1);
}
// Stores of a k_column_step workgroup go THROUGH the L2 (agent-scope store): dirty lines left in an XCD's L2 are written
// back at the kernel boundary, which is on the critical path of the factorisation.
__device__ __forceinline__ void StoreThrough(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void StoreTile(double* __restrict__ dst, const double* src, int ld, int tid) {
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int idx = tid + kPanelThreads * it, r = idx >> 5, c2 = idx & 31;
    const double2 v = *reinterpret_cast<const double2*>(src + r * kLS + 2 * c2);
    StoreThrough(dst + (size_t)r * ld + 2 * c2, v.x);
    StoreThrough(dst + (size_t)r * ld + 2 * c2 + 1, v.y);
  }
}

// The factorisation proper of a 64x64 block held in LDS buffer A (lower triangle valid) together with M = L^-1
// (LDS buffer, must be zero on entry):
//   sequential 16-column panels on wavefront 0, in-block trailing updates one tile per wavefront;
//   while wavefront 0 is in panel P+1, wavefront 15 inverts diagonal tile P; the off-diagonal tiles
//     M_Pj = -M_PP (sum_{m=j}^{P-1} L_Pm M_mj)
//   are built by otherwise idle wavefronts as early as their inputs exist (never on wavefront 0's SIMD during a
//   panel); only the products with the last two tile inverses remain after panel 3
// `side(w)` is run by wavefronts 1..15 during panel 0 (they idle there), `side1(w)` by wavefronts 1..14 during panel 1.
struct NoSideJob { __device__ void operator()(int) const {} };
// `spare(phase)`: the wavefronts that have nothing left to do once panel 1 is over (4 and 7..14) run it beside panel 2 (phase 0),
// beside panel 3 (phase 1) and once more after the last barrier (phase 2, may block) - the task mode's chain fetches the next step's X and D tiles there.  They take a branch of
// their own that only mirrors the remaining barriers, so whatever registers the job keeps between its two calls are not live
// through wavefront 0's panels (kept in the common path they pushed the 128-VGPR kernel into spills).  It may not block:
// wavefront 0 meets the others at the barrier after each panel.
template <typename Side, typename Side1, typename Spare = NoSideJob>
__device__ __forceinline__ void PotrfPanels(double* A, double* M, double* inv_diag, int32_t* __restrict__ flag, int lane, int w, Side side, Side1 side1,
                                            Spare spare = Spare()) {
  constexpr int kInvWave = kPanelThreads / 64 - 1;
  __shared__ int m22_ready;      // set by the inverting wavefront during panel 3 (see there); cleared here, barriers follow
#define PP_PANEL16(P) PotrfPanel16<P>(A, inv_diag, lane, flag)
  const int lr = lane & 15, g = lane >> 4;
  const v4f64 zero = (v4f64){0.0, 0.0, 0.0, 0.0};
  if (w == kInvWave && lane == 0) __hip_atomic_store(&m22_ready, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  if (w == 0) { __builtin_amdgcn_s_setprio(3); PP_PANEL16(0); }
  else side(w);
  PP_WAVE_ARRIVE(0); __syncthreads();
  PP_CHOL_PHASE(3);
  PotrfTrailing16<0>(A, lane, w);
  PP_WAVE_ARRIVE(1); __syncthreads();
  PP_CHOL_PHASE(4);
  if (w == 0) PP_PANEL16(1);
  else if (w == kInvWave) InverseDiag16<0>(A, inv_diag, M, lane);
  else side1(w);
  PP_WAVE_ARRIVE(2); __syncthreads();
  PP_CHOL_PHASE(5);
  if (!std::is_same<Spare, NoSideJob>::value && (w == 4 || (w >= 7 && w < kInvWave))) {
    PP_WAVE_ARRIVE(3); __syncthreads();      // after the trailing update of panel 1 (wavefronts 0..2)
    spare(0);             // beside panel 2
    PP_WAVE_ARRIVE(4); __syncthreads();      // after panel 2
    PP_WAVE_ARRIVE(5); __syncthreads();      // after the trailing update of panel 2
    spare(1);             // beside panel 3
    PP_WAVE_ARRIVE(6); __syncthreads();      // after panel 3
    PP_WAVE_ARRIVE(7); __syncthreads();      // after the last products of M
    spare(2);             // after the last panel: this one may block (everything the job keeps in registers stays inside this branch)
    return;
  }
  PotrfTrailing16<1>(A, lane, w);
  PP_WAVE_ARRIVE(3); __syncthreads();
  PP_CHOL_PHASE(6);
  if (w == 0) PP_PANEL16(2);
  if (w == kInvWave) InverseDiag16<1>(A, inv_diag, M, lane);
  PP_WAVE_ARRIVE(4); __syncthreads();
  PP_CHOL_PHASE(7);
  PotrfTrailing16<2>(A, lane, w);
  if (w == 1) {   // M_10 = -M_11 (L_10 M_00)
    const v4f64 t = TileMulAB(PP_TILE(A, 1, 0), PP_TILE(M, 0, 0), zero, lr, g);
    TileStoreD(PP_TILE(M, 1, 0), TileNegMulAD(PP_TILE(M, 1, 1), t, lr, g), lr, g);
  }
  PP_WAVE_ARRIVE(5); __syncthreads();
  PP_CHOL_PHASE(8);
  // during panel 3 (which also builds M_33): the inner sums of rows 2 and 3 that do not need M_22 (being inverted by
  // wavefront 15 now); then, as soon as that wavefront announces M_22 through an LDS flag (no workgroup barrier can be used
  // while wavefront 0 is in the panel), the rest of row 2 and the M_22 part of row 3.  After the panel one round of products
  // per row-3 tile remains.
  v4f64 t = zero;
  if (w == 0) PotrfLastPanelWithInverse(A, M, lane, flag);
  if (w == kInvWave) {
    InverseDiag16<2>(A, inv_diag, M, lane);
    if (lane == 0) __hip_atomic_store(&m22_ready, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  if (w == 1) { t = TileMulAB(PP_TILE(A, 2, 0), PP_TILE(M, 0, 0), zero, lr, g); t = TileMulAB(PP_TILE(A, 2, 1), PP_TILE(M, 1, 0), t, lr, g); }   // for M_20
  if (w == 2) t = TileMulAB(PP_TILE(A, 2, 1), PP_TILE(M, 1, 1), zero, lr, g);                                                                      // for M_21
  if (w == 3) { t = TileMulAB(PP_TILE(A, 3, 0), PP_TILE(M, 0, 0), zero, lr, g); t = TileMulAB(PP_TILE(A, 3, 1), PP_TILE(M, 1, 0), t, lr, g); }   // for M_30
  if (w == 5) t = TileMulAB(PP_TILE(A, 3, 1), PP_TILE(M, 1, 1), zero, lr, g);                                                                      // for M_31
  if (w == 1 || w == 2 || w == 6) {
    while (__hip_atomic_load(&m22_ready, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) __builtin_amdgcn_s_sleep(2);
    if (w == 6) t = TileMulAB(PP_TILE(A, 3, 2), PP_TILE(M, 2, 2), zero, lr, g);                                 // for M_32
    else TileStoreD(PP_TILE(M, 2, w - 1), TileNegMulAD(PP_TILE(M, 2, 2), t, lr, g), lr, g);                     // M_2j = -M_22 t
  }
  PP_WAVE_ARRIVE(6); __syncthreads();
  PP_CHOL_PHASE(9);
  if (w == 3) t = TileMulAB(PP_TILE(A, 3, 2), PP_TILE(M, 2, 0), t, lr, g);
  if (w == 5) t = TileMulAB(PP_TILE(A, 3, 2), PP_TILE(M, 2, 1), t, lr, g);
  if (w == 3 || w == 5 || w == 6) TileStoreD(PP_TILE(M, 3, w == 3 ? 0 : (w == 5 ? 1 : 2)), TileNegMulAD(PP_TILE(M, 3, 3), t, lr, g), lr, g);
  PP_WAVE_ARRIVE(7); __syncthreads();
}

__device__ __forceinline__ void ZeroTile(double* dst, int tid) {
  for (int idx = tid; idx < kNB * kLS / 2; idx += kPanelThreads) reinterpret_cast<double2*>(dst)[idx] = make_double2(0.0, 0.0);
}
// the same with the zero made in place: inside the task mode's k-loop the compiler keeps a loop-invariant zero quad for the stores
// above, spills it, and reloads it from scratch per store - behind an s_waitcnt vmcnt(0) that also waits for the wavefront's
// outstanding mailbox stores (0.8 us per step on the chain)
__device__ __forceinline__ void ZeroTileFresh(double* dst, int tid) {
  double z = 0.0;
  asm volatile("" : "+v"(z));
  for (int idx = tid; idx < kNB * kLS / 2; idx += kPanelThreads) reinterpret_cast<double2*>(dst)[idx] = make_double2(z, z);
}

// X (LDS, 64x64) -> tile (s, ct) of X M^T = sum_{kt <= ct} X[s][kt] M[ct][kt]^T, D layout
// (all operand loads of the tile's 1..4 products are issued before the first MFMA and the MFMAs run on four independent
// partial accumulators: the wavefront with four products was a chain of four load -> MFMA -> add round trips, 2.0 us)
template <int NK>
__device__ __forceinline__ v4f64 SolveTileN(const double* X, const double* M, int s, int ct, int lr, int g) {
  double av[4 * NK], bv[4 * NK];
#pragma unroll
  for (int kt = 0; kt < NK; ++kt)
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) { av[4 * kt + kk] = PP_TILE(X, s, kt)[lr * kLS + 4 * kk + g]; bv[4 * kt + kk] = PP_TILE(M, ct, kt)[lr * kLS + 4 * kk + g]; }
  const v4f64 z = (v4f64){0.0, 0.0, 0.0, 0.0};
  v4f64 p[4] = {z, z, z, z};
#pragma unroll
  for (int i = 0; i < 4 * NK; ++i) p[i & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[i], bv[i], p[i & 3], 0, 0, 0);
  return (p[0] + p[1]) + (p[2] + p[3]);
}
__device__ __forceinline__ v4f64 SolveTile(const double* X, const double* M, int s, int ct, int lr, int g) {
  switch (ct) {      // wave-uniform
    case 0: return SolveTileN<1>(X, M, s, ct, lr, g);
    case 1: return SolveTileN<2>(X, M, s, ct, lr, g);
    case 2: return SolveTileN<3>(X, M, s, ct, lr, g);
    default: return SolveTileN<4>(X, M, s, ct, lr, g);
  }
}
// tile (ti, tj) of X (LDS, in place) -= A_ti B_tj^T   (K = 64)
__device__ __forceinline__ void UpdateTileInPlace(double* X, const double* A, const double* B, int ti, int tj, int lr, int g) {
  v4f64 x;
#pragma unroll
  for (int i = 0; i < 4; ++i) x[i] = PP_TILE(X, ti, tj)[(g + 4 * i) * kLS + lr];
  double av[16], bv[16];
#pragma unroll
  for (int kk = 0; kk < 16; ++kk) { av[kk] = -A[(16 * ti + lr) * kLS + 4 * kk + g]; bv[kk] = B[(16 * tj + lr) * kLS + 4 * kk + g]; }
  x = MfmaK16(av, bv, x);
  TileStoreD(PP_TILE(X, ti, tj), x, lr, g);
}

// tile (ti, tj) held in registers (D layout) -= A_ti B_tj^T  (K = 64), operands in LDS
__device__ __forceinline__ v4f64 UpdateTileRegs(v4f64 x, const double* A, const double* B, int ti, int tj, int lr, int g) {
  double av[16], bv[16];
#pragma unroll
  for (int kk = 0; kk < 16; ++kk) { av[kk] = -A[(16 * ti + lr) * kLS + 4 * kk + g]; bv[kk] = B[(16 * tj + lr) * kLS + 4 * kk + g]; }
  return MfmaK16(av, bv, x);
}


// The whole solve of an augmented system of one or two block columns (N = 64 or 128: 6 C + 1 <= 128, at most 21 images) on LDS tiles
// [A11 | A21 | A22 | M] (row stride kLS, lower triangle valid, row n = the right-hand side, identity padding below it): blocked Cholesky with
// the forward substitution folded in, then the back substitution with the explicit block inverses
//   x2 = M22^T y2,   x1 = M11^T (y1 - L21^T x2)         (y = row n of the factor; M22 is built in L11's buffer, which nothing needs again)
// x -> xs (LDS, 2 x 64 doubles) and x_out[0 .. n).  One workgroup of kPanelThreads; a bad pivot sets bit 0 of *flag.
// Used by k_small_cholesky (cholesky.hip): the reduced system of the
// mapper's local bundle adjustment (src/sfm/incremental_mapper.cc:813-858) in ONE launch instead of k_potrf64 + k_column_step + k_backsub_all.
__device__ __forceinline__ void SmallFactorSolveTiles(double* tiles, double* inv_diag, double* xs, double* ys, int32_t* flag, int T, int n, double* __restrict__ x_out) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  double* A11 = tiles; double* A21 = tiles + kNB * kLS; double* A22 = tiles + 2 * kNB * kLS; double* Mb = tiles + 3 * kNB * kLS;
  for (int blk = 0; blk < T; ++blk) {      // (one copy of the panels in the code: the tile pointers are run-time values)
    double* A = blk ? A22 : A11;
    double* Md = blk ? A11 : Mb;
    if (blk) {
      const int s = w & 3, ct = w >> 2, lr = lane & 15, g = lane >> 4;
      const v4f64 x = SolveTile(A21, Mb, s, ct, lr, g);      // L21 = A21 M11^T
      __syncthreads();
      TileStoreD(PP_TILE(A21, s, ct), x, lr, g);
      __syncthreads();
      if (w < 10) {                                          // A22 -= L21 L21^T (the lower triangle of 16x16 pieces)
        int di = 0, rem = w;
        while (rem > di) { rem -= di + 1; ++di; }
        UpdateTileInPlace(A22, A21, A21, di, rem, lr, g);
      }
    }
    ZeroTile(Md, tid);
    __syncthreads();
    PotrfPanels(A, Md, inv_diag, flag, lane, w, NoSideJob(), NoSideJob());
  }
  const int m2 = T == 2 ? n - kNB : 0, n1 = T == 2 ? kNB : n;
  if (T == 2) {
    if (tid < m2) ys[kNB + tid] = A22[m2 * kLS + tid];
    if (tid < kNB) ys[tid] = A21[m2 * kLS + tid];
    __syncthreads();
    if (tid < m2) { double v = 0.0; for (int i = tid; i < m2; ++i) v += A11[i * kLS + tid] * ys[kNB + i]; xs[kNB + tid] = v; }
    __syncthreads();
    if (tid < kNB) { double v = ys[tid]; for (int i = 0; i < m2; ++i) v -= A21[i * kLS + tid] * xs[kNB + i]; ys[tid] = v; }
    __syncthreads();
  } else {
    if (tid < n1) ys[tid] = A11[n * kLS + tid];
    __syncthreads();
  }
  if (tid < n1) { double v = 0.0; for (int i = tid; i < n1; ++i) v += Mb[i * kLS + tid] * ys[i]; xs[tid] = v; }
  __syncthreads();
  if (tid < n) x_out[tid] = xs[tid];
  __syncthreads();      // (also publishes x_out to the workgroup)
}

}  // namespace ppsfm

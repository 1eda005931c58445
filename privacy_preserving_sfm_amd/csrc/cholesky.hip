// K3b — dense fp64 Cholesky solve of the reduced camera system on the matrix cores.
//
// Replaces the linear solve Ceres performs inside ceres::Solve for the Schur-reduced camera
// system (DENSE_SCHUR / SPARSE_SCHUR chosen at reference src/optim/bundle_adjustment.cc:275-286).
//
// Layout: S is N x N row-major, N a multiple of 64, only the lower triangle is referenced.  Row
// `rhs_row` (= 6C) holds the right-hand side b (augmented system [S b; b' BIG]): its Cholesky
// factor's row rhs_row is y = L^-1 b, i.e. the forward substitution is performed by the
// factorisation itself.  Rows beyond rhs_row are identity padding.
//
// Two launch structures over the same work items (same arithmetic per 16x16 piece, bitwise equal results up to 48 block columns):
//   task mode     (k_cholesky_tasks, default up to 64 block columns) the whole factorisation in ONE launch: a persistent chain
//                 workgroup + one workgroup per item of a priority-sorted task list, per-tile dependency counters, mailbox
//                 hand-offs - see "task mode" below; a block-sparse system whose elimination tree has independent sub-trees (a nested-dissection
//                 order of the cameras, ba_eval.hip DissectBand) gets a chain workgroup per sub-tree - see "ChainRanges";
//   column mode   (k_column_step, above 64 block columns and for a block-sparse system) one launch per block column:
// Right-looking blocked algorithm, 64 x 64 blocks, ONE launch per block column (k_column_step, see there):
// a chain workgroup (solve of the tile X left of the next diagonal block, that block's update by X X', its
// factorisation and the 64x64 INVERSE of the new diagonal factor) runs in the same grid as a prep workgroup (applies
// the panel-k update to the NEXT launch's chain inputs, X into a staging tile) and the bulk work that only depends
// on earlier launches (triangular solves of column k as plain products with L_kk^-1, trailing updates of
// panel k-1).  A cross-queue event edge was measured at ~9 us and a kernel boundary at ~1.7 us on MI355X, so the
// overlap is expressed INSIDE a grid and a step of the critical path pays one boundary.
// One CU sustains 307 GFLOP/s of v_mfma_f64_16x16x4 (26.6 ns per MFMA per SIMD, = the fp64 vector rate), so the
// chain workgroup is sized at 16 wavefronts and every product is spread one 16x16 tile per wavefront.
//   PotrfPanel16    16-column panel in the REGISTERS of one wavefront (lane = row, pivots/multipliers by
//                   v_readlane, no barriers); the last panel and the 16x16 tile inverses use DPP row broadcasts
//                   (PotrfLastPanelWithInverse, InverseDiag16)
//   PotrfPanels     panels on wavefront 0; in-block trailing updates one tile per wavefront; the other wavefronts
//                   build L^-1 (16x16 tile inverses by substitution, off-diagonal tiles by MFMA products) while
//                   wavefront 0 is in the next panel
//   k_backsub_all   the whole back substitution in one launch, block j waiting on the x_k (k > j) it needs
// Roofline: the trailing update is fp64-MFMA bound (n^3/3 flop); the chain workgroup is latency bound.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include "ba_impl.hpp"
#include "chol_block.hpp"
#include "resource_pool.hpp"

namespace ppsfm {

// SEVERAL CHAINS (a block-sparse system whose elimination tree has independent sub-trees - a nested-dissection order of the cameras: the leaves are
// factorised side by side, the separators last).  A chain is a run of consecutive block columns [begin, end) whose tiles (k+1,k) / (k+2,k) exist; it
// STARTS at a block column whose rows k, k+1, k+2 have nothing left of column k (no panel ever touches the three tiles of its first step: k_potrf64
// factorises every chain's first diagonal block) and a chain that is followed by another one STOPS after the step that produces M_(end-1): its last
// block column is solved by solve tasks alone (rows >= end + 3: the separators), and `post` is what it stores into sol[end - 1] when the solved tile
// (end-1,end-2) is in L.  Workgroup c of k_cholesky_tasks runs chain c; the task list follows.
constexpr int kMaxChains = 16;
struct ChainRanges { int32_t n; int32_t begin[kMaxChains]; int32_t end[kMaxChains]; int32_t post[kMaxChains]; };
inline ChainRanges OneChain(int T) { ChainRanges cr; std::memset(&cr, 0, sizeof(cr)); cr.n = 1; cr.end[0] = T; return cr; }

// the FIRST diagonal block of every chain (workgroup c: block column cr.begin[c]; one chain: block column 0): factor in place, emit its inverse
// (row-major 64x64) to that block column's slot of Minv, copy the tile below it to its X slot
// Lout: where the factored block goes (S itself in the per-column mode, the solved-tile array in task mode);
// ctr / nctr: the progress counters of task mode, reset by workgroup 0;  workgroups cr.n .. (task mode only): preset the mailbox slots
// [mail, mail + mail_doubles) to the "not written yet" pattern, except the M and X slots of the chains' first block columns
__global__ __launch_bounds__(kPanelThreads) void k_potrf64(double* __restrict__ S, int ld, double* __restrict__ Minv, double* __restrict__ xs, int32_t* __restrict__ flag,
                                                           double* __restrict__ x_out, double* Lout, int32_t* __restrict__ ctr, int nctr, double* mail,
                                                           long long mail_doubles, ChainRanges cr) {
  __shared__ __attribute__((aligned(16))) double smem[2 * kNB * kLS];
  __shared__ double inv_diag[kNB];
  double* A = smem;
  double* M = smem + kNB * kLS;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  // workgroups 0 .. cr.n - 1: the first diagonal block of chain c (block column kb: M_kb and the X slot of step kb are written here, not preset)
  if ((int)blockIdx.x >= cr.n) {
    // slot by slot (64 x 64 doubles each; the X slots follow the T slots of M): whether a slot is one of a chain's first step is decided once per slot
    const int nslots = (int)(mail_doubles / (kNB * kNB)), xslot0 = (int)((xs - mail) / (kNB * kNB));
    const double pattern = __longlong_as_double(-1ll);
    for (int sl = (int)blockIdx.x - cr.n; sl < nslots; sl += (int)gridDim.x - cr.n) {
      bool keep = false;
#pragma unroll
      for (int c = 0; c < kMaxChains; ++c) keep = keep || (c < cr.n && (sl == cr.begin[c] || sl == xslot0 + cr.begin[c]));
      if (keep) continue;
      double* dst = mail + (size_t)sl * kNB * kNB;
#pragma unroll
      for (int it = 0; it < 4; ++it) StoreThrough(dst + tid + kPanelThreads * it, pattern);
    }
    return;
  }
  int kb = 0;
#pragma unroll
  for (int c = 0; c < kMaxChains; ++c) if (c == (int)blockIdx.x) kb = cr.begin[c];
  if (blockIdx.x == 0) {
    if (tid == 0) __hip_atomic_store(flag + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // the PrepD -> PrepX token of k_column_step
    for (int i = tid; i < nctr; i += kPanelThreads) __hip_atomic_store(ctr + i, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // progress counters (task mode)
    // the back substitution's hand-off buffer starts as 'not ready' (k_backsub_all); x_out may be null (factorisation only)
    if (x_out) for (int i = tid; i < ld; i += kPanelThreads) __hip_atomic_store(reinterpret_cast<unsigned long long*>(x_out + i), 0xFFFFFFFFFFFFFFFFull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  const size_t diag = (size_t)kb * kNB * ld + (size_t)kb * kNB;
  LoadTile(A, S + diag, ld, tid);
  ZeroTile(M, tid);
  __syncthreads();
  PotrfPanels(A, M, inv_diag, flag, lane, w, NoSideJob(), NoSideJob());
  StoreTile(Lout + diag, A, ld, tid);    // the strictly upper part of a diagonal block is never read
  StoreTile(Minv + (size_t)kb * kNB * kNB, M, kNB, tid);
  if (ld > kNB * (kb + 1)) {      // staging copy of tile (kb+1,kb) for the chain's first step (see k_column_step)
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int idx = tid + kPanelThreads * it, r = idx >> 5, c2 = idx & 31;
      *reinterpret_cast<double2*>(xs + (size_t)kb * kNB * kNB + r * kNB + 2 * c2) = *reinterpret_cast<const double2*>(S + diag + (size_t)(kNB + r) * ld + 2 * c2);
    }
  }
}

// lower-triangular tile index t -> (row, col), row >= col
__device__ __forceinline__ void TriIndex(int t, int* row, int* col) {
  int r = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
  while ((r + 1) * (r + 2) / 2 <= t) ++r;
  while (r * (r + 1) / 2 > t) --r;
  *row = r; *col = t - r * (r + 1) / 2;
}

// Trailing update by block column kp of the region below / right of block (kp+2, kp+2):  C -= A_i A_j^T, in 128x128
// SUPER-TILES (2x2 blocks): the workgroup stages the two 128x64 operand panels in LDS (135 KB) and each of its 16
// wavefronts owns a 32x32 piece of C = 2x2 MFMA tiles, so every operand fetched from LDS feeds two MFMAs and every operand
// byte fetched from L2 serves two block rows / columns.  Why: with one 64x64 tile per workgroup the update moved 128 KB
// per 0.5 MFLOP and the early launches (k <~ 15, ~1000 tiles) were bound by memory and by the dispatcher (1000 sixteen-
// wavefront workgroups took 19 us to START at k = 2), 20-27 us against a 15 us chain.  The accumulators start at zero and
// the C piece, whose loads are issued before the products, is combined at the end (no exposed global round trip).
// Super-tile u -> TriIndex(u + 1) = (I, J): super-tile (0, 0) is exactly the three tiles the chain / prep workgroups own.
// Workgroup q of nW takes u = q, q + nW, ...
// Deferred pairs (early launches, where this update is bound by the read-modify-write of C in HBM/MALL): in the first
// launch of a pair the tiles of block columns >= skip_from are left out, in the second one the tiles of block columns
// >= double_from (the same set, aligned to a super-column there) take the panels kp-1 AND kp in one visit — C moves once
// for 128 columns of update.  Nothing reads those tiles in between (they are >= 5 block columns ahead of the front).
template <bool kTask>      // kTask: operand panels come from the solved-tile array L (write-once), C is read coherently from S
__device__ __forceinline__ void SyrkSuperTiles(double* S, const double* L, int ld, int kp, int T, int q, int nW, int skip_from,
                                               int double_from, double* As, double* Bs) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int lr = lane & 15, lk = lane >> 4;
  const int wi = w >> 2, wj = w & 3;
  const int k1 = kp + 2, nb = T - k1, ns = (nb + 1) / 2, nsup = ns * (ns + 1) / 2 - 1;
  for (int u = q; u < nsup; u += nW) {
    int I, J;
    TriIndex(u + 1, &I, &J);
    const int bi0 = k1 + 2 * I, bj0 = k1 + 2 * J;
    if (bj0 >= skip_from) continue;                                  // the whole super-tile is deferred
    const bool has_i1 = bi0 + 1 < T, has_j1 = bj0 + 1 < T;
    const bool two = bj0 >= double_from;                             // workgroup-uniform: double_from starts a super-column
    // this wavefront's block and 32x32 piece
    const int bi = bi0 + (wi >> 1), bj = bj0 + (wj >> 1);
    const bool valid = bi < T && bj < T && bi >= bj && bj < skip_from;
    const size_t cbase = (size_t)bi * kNB * ld + (size_t)bj * kNB + (size_t)(32 * (wi & 1) + lk) * ld + 32 * (wj & 1) + lr;
    const v4f64 z = (v4f64){0.0, 0.0, 0.0, 0.0};
    v4f64 c[2][2], p[2][2] = {{z, z}, {z, z}};
    for (int pass = two ? 0 : 1; pass < 2; ++pass) {
      const size_t col = (size_t)(kp - 1 + pass) * kNB;
      __syncthreads();                    // the previous pass's / super-tile's (or the previous role's) LDS reads are done
      {
        const double* a0 = L + (size_t)bi0 * kNB * ld + col;
        const double* b0 = L + (size_t)bj0 * kNB * ld + col;
        const double* a1 = has_i1 ? a0 + (size_t)kNB * ld : a0;      // a missing block: any valid address, its results are not stored
        const double* b1 = has_j1 ? b0 + (size_t)kNB * ld : b0;
        LoadTiles4(As, a0, As + kNB * kLS, a1, Bs, b0, Bs + kNB * kLS, b1, ld, tid);
      }
      if (valid && pass == (two ? 0 : 1)) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 4; ++i) c[a][b][i] = LoadS<kTask>(S + cbase + (size_t)(16 * a + 4 * i) * ld + 16 * b);
      }
      __syncthreads();
      if (valid) {
        const double* ar = As + (32 * wi + lr) * kLS + lk;
        const double* br = Bs + (32 * wj + lr) * kLS + lk;
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
          const double a0v = ar[4 * kk], a1v = ar[16 * kLS + 4 * kk], b0v = br[4 * kk], b1v = br[16 * kLS + 4 * kk];
          p[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0v, b0v, p[0][0], 0, 0, 0);
          p[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0v, b1v, p[0][1], 0, 0, 0);
          p[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1v, b0v, p[1][0], 0, 0, 0);
          p[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1v, b1v, p[1][1], 0, 0, 0);
        }
      }
    }
    if (valid) {
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int i = 0; i < 4; ++i)      // write-through: 30 MB of dirty lines per launch would otherwise be flushed at the kernel boundary, on the critical path
            __hip_atomic_store(S + cbase + (size_t)(16 * a + 4 * i) * ld + 16 * b, c[a][b][i] - p[a][b][i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// ---- one launch per block column ------------------------------------------------------------------------
// Launch k (k = 0 .. T-2).  Before it: L_kk and M_k = L_kk^-1 are final, block column k-1 is solved, every tile (i,j),
// j >= k, carries the trailing updates of panels 0 .. k-2, and the two tiles the chain needs — X = (k+1,k) and
// D = (k+1,k+1) — already carry panel k-1 as well (prepared by the previous launch); X is read from a staging copy
// xs[k & 1] so that nobody reads a tile another workgroup of the same launch overwrites.
// The launch holds four kinds of workgroup, all depending on EARLIER launches only:
//   chain (blockIdx 0):  X <- X M_k^T (stored to S);  D -= X X^T, factor D, store L_{k+1,k+1} and M_{k+1}
//   prep  (blockIdx 1 = PrepX, 2 = PrepD, when block row k+2 exists): everything the NEXT chain needs, redundantly where
//          necessary (PrepBody):
//          A_{k+1,k} = X M_k^T and A_{k+2,k} = ((k+2,k) - A_{k+2,k-1} A_{k,k-1}^T) M_k^T (stored: it is also column k's tile),
//          X' = (k+2,k+1) - A_{k+2,k-1} A_{k+1,k-1}^T - A_{k+2,k} A_{k+1,k}^T  (stored to S and to xs[(k+1) & 1]),
//          D' = (k+2,k+2) - A_{k+2,k-1} A_{k+2,k-1}^T - A_{k+2,k} A_{k+2,k}^T
//   trsm tiles  (i >= k+3):  tile (i,k) -= A_{i,k-1} A_{k,k-1}^T, times M_k^T, store
//   trailing update (SyrkSuperTiles): tiles (i,j), i >= j >= k+1, except the three the chain and the prep workgroups own:
//          -= A_{i,k-1} A_{j,k-1}^T, one 128x128 super-tile per workgroup
// so the critical path of a step is: load X, M_k, D -> one product -> one rank-64 update of D's first block column ->
// the panels -> ONE kernel boundary.  Everything else (the prep pair, the solves, the trailing update) runs beside it.
// The triangular solve is a plain product with the explicit inverse of the 64x64 diagonal factor: 10 independent
// 16x16x16 products per 16-row strip instead of a 7-stage dependent substitution chain, and the back substitution gets
// its L_kk^-1 for free.

template <bool kTask>      // kTask: the unsolved tile is read coherently from S, solved tiles live in L (legacy: L == S)
__device__ __forceinline__ void TrsmTileBody(double* S, double* L, int ld, int k, int i, const double* __restrict__ Minv, double* BX,
                                             double* Mk, double* B1, double* B2) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, lr = lane & 15, g = lane >> 4;
  const size_t dbase = (size_t)k * kNB * ld + (size_t)k * kNB, pbase = (size_t)i * kNB * ld + (size_t)k * kNB;
  const double* mk = Minv + (size_t)k * kNB * kNB;
  if (k > 0) {
    // X, A_{i,k-1}, A_{k,k-1} (row stride ld) and M_k (row stride 64)
    if (kTask) { LoadTileT<true>(BX, S + pbase, ld, tid); LoadTile(B1, L + pbase - kNB, ld, tid); }
    else LoadTiles2(BX, S + pbase, B1, S + pbase - kNB, ld, tid);
    LoadTile(B2, L + dbase - kNB, ld, tid);
    LoadTileT<kTask>(Mk, mk, kNB, tid);      // (task mode: M_k's slot is a mailbox other workgroups have polled - its lines may sit stale in an L2)
    __syncthreads();
    UpdateTileInPlace(BX, B1, B2, w >> 2, w & 3, lr, g);
  } else {
    LoadTileT<kTask>(BX, S + pbase, ld, tid);
    LoadTile(Mk, mk, kNB, tid);
  }
  __syncthreads();
  const int s = w & 3, ct = w >> 2;    // SIMD (w & 3) gets one tile of every column tile: balanced MFMA load
  const v4f64 x = SolveTile(BX, Mk, s, ct, lr, g);
#pragma unroll
  for (int r = 0; r < 4; ++r) StoreThrough(L + pbase + (size_t)(16 * s + g + 4 * r) * ld + 16 * ct + lr, x[r]);
}

__device__ __forceinline__ void ChainBody(double* __restrict__ S, int ld, int k, int T, double* __restrict__ Minv, const double* __restrict__ xs_k,
                                          int32_t* __restrict__ flag, double* BX, double* Mk, double* BD, double* BS, double* inv_diag) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, lr = lane & 15, g = lane >> 4;
  const size_t xbase = (size_t)(k + 1) * kNB * ld + (size_t)k * kNB;       // X
  const size_t nbase = xbase + kNB;                                        // D
  // D tiles: the first block column (needed by panel 0) on wavefronts 0..3; the six others on wavefronts
  // 5,6,7,9,10,11, which finish them while wavefront 0 is already in panel 0 — none of them shares wavefront 0's
  // SIMD (w & 3 == 0), whose issue slots the panel needs
  const bool dlate = w >= 5 && w < 12 && (w & 3) != 0;
  const bool dwave = w < 4 || dlate;
  int dti = w & 3, dtj = 0;
  if (dlate) {   // wavefronts 5,6,7: (1,1) (2,1) (3,1);  9,10,11: (2,2) (3,2) (3,3)
    if (w < 8) { dti = w - 4; dtj = 1; } else { dti = w == 9 ? 2 : 3; dtj = w == 11 ? 3 : 2; }
  }
  PP_CHOL_PHASE(0);
  v4f64 d = (v4f64){0.0, 0.0, 0.0, 0.0};
  if (dwave) {
#pragma unroll
    for (int i = 0; i < 4; ++i) d[i] = S[nbase + (size_t)(16 * dti + g + 4 * i) * ld + 16 * dtj + lr];
  }
  LoadTiles2(BX, xs_k, Mk, Minv + (size_t)k * kNB * kNB, kNB, tid);          // both row-major 64x64, row stride 64
  __syncthreads();
  PP_CHOL_PHASE(1);
  PP_CHOL_PHASE(2);
  const int s = w & 3, ct = w >> 2;      // SIMD (w & 3) gets one tile of every column tile: balanced MFMA load
  const v4f64 x = SolveTile(BX, Mk, s, ct, lr, g);
  TileStoreD(PP_TILE(BS, s, ct), x, lr, g);   // solved X
  __syncthreads();
  ZeroTile(BX, tid);                          // BX becomes M_{k+1} (every read of it is done)
  PP_CHOL_PHASE(12);
  if (w < 4) {   // first block column of D -= X X^T
    d = UpdateTileRegs(d, BS, BS, dti, dtj, lr, g);
    TileStoreD(PP_TILE(BD, dti, dtj), d, lr, g);
  }
  __syncthreads();
  PP_CHOL_PHASE(13);
  // Beside panel 0: the D tiles of block column 1 (the next panel's) on wavefronts 5,6,7 - one per SIMD other than wavefront 0's -
  // while wavefronts 9,10,11 only park their raw D tiles (2,2) (3,2) (3,3) in LDS; beside panel 1 those three tiles get their
  // X X^T (nothing reads them before the trailing update that follows panel 1).  With all six tiles beside panel 0 that phase
  // lasted 1.8 us for a 1.45 us panel (two 16-MFMA tiles per SIMD).  The solved X goes back to S from wavefronts that idle
  // beside panel 0 (issued later, its write-through stores were still in flight at the end of the workgroup).
  auto side = [&](int wv) {
    if (dlate) {
      if (dtj == 1) d = UpdateTileRegs(d, BS, BS, dti, dtj, lr, g);
      TileStoreD(PP_TILE(BD, dti, dtj), d, lr, g);
    } else if ((wv & 3) != 0) {
      const int p = (wv < 4 ? wv - 1 : wv - 10) * 64 + lane;     // wavefronts 1,2,3,13,14,15
      for (int idx = p; idx < 2048; idx += 384) {
        const int r = idx >> 5, c2 = idx & 31;
        const double2 v = *reinterpret_cast<const double2*>(BS + r * kLS + 2 * c2);
        StoreThrough(S + xbase + (size_t)r * ld + 2 * c2, v.x);
        StoreThrough(S + xbase + (size_t)r * ld + 2 * c2 + 1, v.y);
      }
    }
  };
  auto side1 = [&](int wv) {
    if (dlate && dtj != 1) UpdateTileInPlace(BD, BS, BS, dti, dtj, lr, g);
  };
  PotrfPanels(BD, BX, inv_diag, flag, lane, w, side, side1);
  PP_CHOL_PHASE(10);
  StoreTile(S + nbase, BD, ld, tid);
  StoreTile(Minv + (size_t)(k + 1) * kNB * kNB, BX, kNB, tid);
  PP_CHOL_PHASE(11);
}

// The two workgroups that prepare the NEXT launch's chain inputs (see above); require k + 2 < T.  Both need the solved
// tile A_{k+2,k} and compute it themselves (an update + a solve: ~3.6 us of MFMA) rather than hand it over inside the
// launch; one workgroup doing everything was ~11 us of MFMA work on one CU and had become longer than the chain.
//   PrepX: A_{k+2,k} (stored), A_{k+1,k}, X' = (k+2,k+1) - A_{k+2,k-1} A_{k+1,k-1}^T - A_{k+2,k} A_{k+1,k}^T
//   PrepD: A_{k+2,k},           D' = (k+2,k+2) - A_{k+2,k-1} A_{k+2,k-1}^T - A_{k+2,k} A_{k+2,k}^T
// Every global load is issued before the first product (tiles that have no free LDS buffer yet wait in registers).
template <bool kIsX>
__device__ __forceinline__ void PrepBody(double* __restrict__ S, int ld, int k, const double* __restrict__ Minv, const double* __restrict__ xs_k,
                                         double* __restrict__ xs_next, int32_t* __restrict__ flag, double* Ba, double* Bb, double* Bc, double* Bm) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, lr = lane & 15, g = lane >> 4;
  const int ti = w >> 2, tj = w & 3, s = w & 3, ct = w >> 2;
  const bool prev = k > 0;
  const size_t row_k = (size_t)k * kNB * ld, row_k1 = (size_t)(k + 1) * kNB * ld, row_k2 = (size_t)(k + 2) * kNB * ld;
  const size_t col_km1 = (size_t)(k - 1) * kNB, col_k = (size_t)k * kNB, col_k1 = (size_t)(k + 1) * kNB, col_k2 = (size_t)(k + 2) * kNB;
  // D' tile of this wavefront (PrepD, wavefronts 0..9: the lower triangle of 16x16 tiles)
  int di = 0, dj = 0;
  if (!kIsX) { int rem = w; while (rem > di) { rem -= di + 1; ++di; } dj = rem; }
  const bool has_out = kIsX || w < 10;
  const size_t obase = kIsX ? row_k2 + col_k1 + (size_t)(16 * ti) * ld + 16 * tj : row_k2 + col_k2 + (size_t)(16 * di) * ld + 16 * dj;
  // ---- all global loads
  if (prev) LoadTiles2(Ba, S + row_k + col_km1, Bb, S + row_k2 + col_km1, ld, tid);     // A_{k,k-1}, A_{k+2,k-1}
  LoadTile(Bc, S + row_k2 + col_k, ld, tid);
  LoadTile(Bm, Minv + (size_t)k * kNB * kNB, kNB, tid);
  double2 x0, x1, q0, q1;     // PrepX: the staging copy of X = (k+1,k) and A_{k+1,k-1}
  if (kIsX) {
    x0 = TileLoad2(xs_k, kNB, tid, 0); x1 = TileLoad2(xs_k, kNB, tid, 1);
    if (prev) { q0 = TileLoad2(S + row_k1 + col_km1, ld, tid, 0); q1 = TileLoad2(S + row_k1 + col_km1, ld, tid, 1); }
  }
  v4f64 out = (v4f64){0.0, 0.0, 0.0, 0.0};
  if (has_out) {
#pragma unroll
    for (int i = 0; i < 4; ++i) out[i] = S[obase + (size_t)(g + 4 * i) * ld + lr];
  }
  __syncthreads();
  // PrepX overwrites tile (k+2,k) with its solved form while PrepD reads the unsolved one: PrepD's loads have all
  // returned here (they went through registers into LDS), which it announces in flag[1]; PrepX waits for that token
  // before its store (~4 us later; both workgroups are resident from the start of the launch: blockIdx 1 and 2)
  if (!kIsX && tid == 0) __hip_atomic_store(flag + 1, k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // ---- A_{k+2,k}: panel k-1 update, then the solve
  if (prev) { UpdateTileInPlace(Bc, Bb, Ba, ti, tj, lr, g); __syncthreads(); }
  v4f64 x = SolveTile(Bc, Bm, s, ct, lr, g);
  if (!kIsX && prev && has_out) out = UpdateTileRegs(out, Bb, Bb, di, dj, lr, g);   // independent of the solve: fills its latency
  __syncthreads();
  TileStoreD(PP_TILE(Bc, s, ct), x, lr, g);
  if (kIsX) {
    // A_{k+1,k} from the staging copy of X (the chain workgroup stores it to S)
    TileStore2(Ba, tid, 0, x0); TileStore2(Ba, tid, 1, x1);
    if (tid == 0) {
      int spins = 0;
      while (__hip_atomic_load(flag + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < k + 1 && spins < (1 << 20)) { __builtin_amdgcn_s_sleep(1); ++spins; }
      if (spins >= (1 << 20)) atomicOr(flag, 2);      // never observed; reported as a failed factorisation instead of a silent race
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) StoreThrough(S + row_k2 + col_k + (size_t)(16 * s + g + 4 * r) * ld + 16 * ct + lr, x[r]);
    x = SolveTile(Ba, Bm, s, ct, lr, g);
    __syncthreads();
    TileStoreD(PP_TILE(Ba, s, ct), x, lr, g);
    if (prev) { TileStore2(Bm, tid, 0, q0); TileStore2(Bm, tid, 1, q1); }      // A_{k+1,k-1} (M_k is no longer needed)
    __syncthreads();
    if (prev) out = UpdateTileRegs(out, Bb, Bm, ti, tj, lr, g);
    out = UpdateTileRegs(out, Bc, Ba, ti, tj, lr, g);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      StoreThrough(S + obase + (size_t)(g + 4 * i) * ld + lr, out[i]);
      StoreThrough(xs_next + (16 * ti + g + 4 * i) * kNB + 16 * tj + lr, out[i]);
    }
  } else {
    __syncthreads();
    if (has_out) {
      out = UpdateTileRegs(out, Bc, Bc, di, dj, lr, g);
#pragma unroll
      for (int i = 0; i < 4; ++i) StoreThrough(S + obase + (size_t)(g + 4 * i) * ld + lr, out[i]);
    }
  }
}

// Block-sparse systems (lists != nullptr; see BuildSparseLists): the launch holds a solve workgroup only for the rows whose tile
// (i,k) is structurally non-zero in the factor, and an update workgroup only for the super-tiles that panel k-1 actually
// touches - lists[trsm_off ..) = the nT_list rows, lists[syrk_off ..) = one super-tile index u per remaining workgroup.
__global__ __launch_bounds__(kPanelThreads) void k_column_step(double* __restrict__ S, int ld, int k, int T, double* __restrict__ Minv, double* __restrict__ xs,
                                                               int32_t* __restrict__ flag, int skip_from, int double_from, const int32_t* __restrict__ lists,
                                                               int trsm_off, int nT_list, int syrk_off) {
  __shared__ __attribute__((aligned(16))) double smem[4 * kNB * kLS];   // registers already limit a CU to one such workgroup
  __shared__ double inv_diag[kNB];
  const int b = blockIdx.x;
  const int n_prep = (k + 2 < T) ? 2 : 0;
  const int nT = lists ? nT_list : (T - k - 3 > 0 ? T - k - 3 : 0);   // rows k+3 .. T-1 (row k+2 belongs to the prep workgroups)
  double* xs_k = xs + (size_t)(k & 1) * kNB * kNB;
  double* xs_next = xs + (size_t)((k + 1) & 1) * kNB * kNB;
  if (b == 0) {
    if (PP_CHOL_SKIPPED(0)) return;
    PP_CHOL_STAMP(20);
    PP_CHOL_LAUNCH(0, k);
    ChainBody(S, ld, k, T, Minv, xs_k, flag, smem, smem + kNB * kLS, smem + 2 * kNB * kLS, smem + 3 * kNB * kLS, inv_diag);
    PP_CHOL_STAMP(21);
    PP_CHOL_LAUNCH(1, k);
  } else if (b <= n_prep) {
    if (PP_CHOL_SKIPPED(1)) return;
    if (b == 1) {
      PP_CHOL_STAMP(16);
      PrepBody<true>(S, ld, k, Minv, xs_k, xs_next, flag, smem, smem + kNB * kLS, smem + 2 * kNB * kLS, smem + 3 * kNB * kLS);
      PP_CHOL_STAMP(17);
    } else {
      PP_CHOL_STAMP(24);
      PrepBody<false>(S, ld, k, Minv, xs_k, xs_next, flag, smem, smem + kNB * kLS, smem + 2 * kNB * kLS, smem + 3 * kNB * kLS);
      PP_CHOL_STAMP(25);
    }
  } else if (b - n_prep <= nT) {
    if (PP_CHOL_SKIPPED(2)) return;
    if (b == 1 + n_prep) PP_CHOL_STAMP(22);
    const int row = lists ? lists[trsm_off + (b - n_prep - 1)] : k + 2 + (b - n_prep);
    TrsmTileBody<false>(S, S, ld, k, row, Minv, smem, smem + kNB * kLS, smem + 2 * kNB * kLS, smem + 3 * kNB * kLS);
    if (b == 1 + n_prep) PP_CHOL_STAMP(23);
  }
  else if (k >= 1) {
    // trailing update by panel k-1 of the region below (k+1,k+1), except the three tiles the chain / prep workgroups own
    const int nW = (int)gridDim.x - 1 - n_prep - nT, q = b - 1 - n_prep - nT;
    if (PP_CHOL_SKIPPED(3)) return;
    if (q == nW - 1) PP_CHOL_STAMP(18);
    if (lists) SyrkSuperTiles<false>(S, S, ld, k - 1, T, lists[syrk_off + q], 1 << 30, skip_from, double_from, smem, smem + 2 * kNB * kLS);
    else SyrkSuperTiles<false>(S, S, ld, k - 1, T, q, nW, skip_from, double_from, smem, smem + 2 * kNB * kLS);
    if (q == nW - 1) PP_CHOL_STAMP(19);
  }
  PP_CHOL_LAUNCH(2, k);
}

__host__ __device__ inline int BacksubNumPairs(int T) { return T >= 7 ? (T - 3) / 2 : 0; }      // blocks 0 .. 2 npairs - 1 in pairs, the 3 or 4 above singly

// C (16 x 16 piece (ti, tj), D layout) = A B over K = 64, A and B 64 x 64 tiles in LDS (row stride kLS), four partial accumulators
__device__ __forceinline__ v4f64 TileProduct64(const double* A, const double* B, int ti, int tj, int lr, int g) {
  double av[16], bv[16];
#pragma unroll
  for (int kk = 0; kk < 16; ++kk) { av[kk] = A[(16 * ti + lr) * kLS + 4 * kk + g]; bv[kk] = B[(4 * kk + g) * kLS + 16 * tj + lr]; }
  return MfmaK16(av, bv, (v4f64){0.0, 0.0, 0.0, 0.0});
}
__device__ __forceinline__ void StoreTileRegs(double* __restrict__ dst, int ld, const v4f64& c, int ti, int tj, int lr, int g) {
#pragma unroll
  for (int i = 0; i < 4; ++i) dst[(size_t)(16 * ti + g + 4 * i) * ld + 16 * tj + lr] = c[i];
}

// Pair gp, part 0: P_g (stored) and Z rows of block 2g+2, columns a; part 1: P_g again (not stored: two products are cheaper than a
// hand-over) and Z rows of block 2g+3, columns a; part 2: Z columns b of both rows.  The top pair has no Z.  `fetch_m(dst, k)` brings
// M_k into an LDS tile: a plain load after the factorisation (k_backsub_prepare), a mailbox fetch inside it (the task mode's
// kTaskPairPrep: the same arithmetic, hidden behind the factorisation).  Returns false if a bounded wait gave up.
template <typename FetchM>
__device__ __forceinline__ bool PairPrepBody(int gp, int part, int T, const double* __restrict__ L, int ld, double* __restrict__ Pw, double* __restrict__ Zw,
                                             double* B0, double* B1, double* B2, double* B3, FetchM fetch_m) {
  const int npairs = BacksubNumPairs(T);
  const int a = 2 * gp, b = a + 1, r0 = a + 2, r1 = a + 3;
  const bool has_z = gp + 1 < npairs;
  if (part > 0 && !has_z) return true;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, lr = lane & 15, g = lane >> 4, ti = w >> 2, tj = w & 3;
  const size_t tile = (size_t)kNB * kNB;
  auto Ltile = [&](int i, int j) { return L + (size_t)i * kNB * ld + (size_t)j * kNB; };
  double* Z = Zw + (size_t)gp * 4 * tile;      // 128 x 128, row-major, row stride 128
  if (part == 2) {
    LoadTiles2(B2, Ltile(r0, b), B3, Ltile(r1, b), ld, tid);
    if (!fetch_m(B1, b)) return false;
    __syncthreads();
    StoreTileRegs(Z + kNB, 2 * kNB, TileProduct64(B2, B1, ti, tj, lr, g), ti, tj, lr, g);
    StoreTileRegs(Z + (size_t)kNB * 2 * kNB + kNB, 2 * kNB, TileProduct64(B3, B1, ti, tj, lr, g), ti, tj, lr, g);
    return true;
  }
  LoadTile(B0, Ltile(b, a), ld, tid);
  // the tiles of the second half travel with the first half's (registers until their LDS buffers are free)
  const int r = part == 0 ? r0 : r1;
  double2 la0 = make_double2(0.0, 0.0), la1 = la0, lb0 = la0, lb1 = la0;
  if (has_z) { la0 = TileLoad2(Ltile(r, a), ld, tid, 0); la1 = TileLoad2(Ltile(r, a), ld, tid, 1); lb0 = TileLoad2(Ltile(r, b), ld, tid, 0); lb1 = TileLoad2(Ltile(r, b), ld, tid, 1); }
  if (!fetch_m(B1, a) || !fetch_m(B2, b)) return false;
  __syncthreads();
  const v4f64 t1 = TileProduct64(B0, B1, ti, tj, lr, g);      // L_ba M_a
  __syncthreads();
  TileStoreD(PP_TILE(B0, ti, tj), t1, lr, g);
  if (has_z) { TileStore2(B3, tid, 0, lb0); TileStore2(B3, tid, 1, lb1); }
  __syncthreads();
  const v4f64 pv = -TileProduct64(B2, B0, ti, tj, lr, g);     // P = -M_b (L_ba M_a)
  if (part == 0) StoreTileRegs(Pw + (size_t)gp * tile, kNB, pv, ti, tj, lr, g);
  if (!has_z) return true;
  __syncthreads();
  TileStoreD(PP_TILE(B0, ti, tj), pv, lr, g);
  TileStore2(B2, tid, 0, la0); TileStore2(B2, tid, 1, la1);
  __syncthreads();
  // L_ra M_a + L_rb P, one product after the other: with both in one expression the scheduler fetched the operands of both (128 VGPRs)
  // before the first MFMA - the three spilled registers that were the only scratch use of the 128-VGPR kernels
  const v4f64 z1 = TileProduct64(B2, B1, ti, tj, lr, g);
  __builtin_amdgcn_sched_barrier(0);
  const v4f64 z = z1 + TileProduct64(B3, B0, ti, tj, lr, g);
  StoreTileRegs(Z + (size_t)(part == 0 ? 0 : kNB) * 2 * kNB, 2 * kNB, z, ti, tj, lr, g);
  return true;
}

// the per-column launch structure: one launch between the factorisation and the solve, grid = 3 npairs
__global__ __launch_bounds__(kPanelThreads) void k_backsub_prepare(const double* __restrict__ L, int ld, int T, const double* __restrict__ Linv,
                                                                   double* __restrict__ Pw, double* __restrict__ Zw) {
  __shared__ __attribute__((aligned(16))) double smem[4 * kNB * kLS];
  auto fetch_m = [&](double* dst, int k) { LoadTile(dst, Linv + (size_t)k * kNB * kNB, kNB, threadIdx.x); return true; };
  (void)PairPrepBody(blockIdx.x / 3, blockIdx.x % 3, T, L, ld, Pw, Zw, smem, smem + kNB * kLS, smem + 2 * kNB * kLS, smem + 3 * kNB * kLS, fetch_m);
}

// ---- task mode: the whole factorisation in ONE launch ------------------------------------------------------------------
// The per-column launches pay, on the critical path of every block column, a kernel boundary (~2.6 us of a ~15 us step) and a
// cold reload of M_k / X / D by a freshly dispatched chain workgroup, and every launch waits for its slowest workgroup.
// Here the same work items are ONE grid:
//   workgroup 0      the CHAIN, persistent: loops over the block columns; M_k never leaves LDS, the next step's X and D
//                    tiles are fetched by wavefronts that idle through the panels, nothing waits for the step's own stores
//   workgroups 1..   one TASK each, read from a list the host builds once per matrix size: PrepX(k), PrepD(k), the
//                    triangular solve of tile (i,k), the update of one 128x128 super-tile by panel k-1.
// Dependencies are tracked per TILE, not per step, so the tiles near the diagonal (which the chain needs next) run ahead of
// the far ones instead of waiting for the whole trailing update of the previous column (launches 1-10 of the per-column mode
// are bound by that update: 18-30 us against a ~12 us chain):
//   sol[i]      number of solved columns of block row i (tile (i,c) solved for every c < sol[i]); set by the solve task of
//               (i,c) and by PrepX(c) for rows c+1 (the chain's tile, which it copies to L) and c+2
//   ver[I][J]   number of panels applied to super-tile (I,J) = block rows 2I,2I+1 x block columns 2J,2J+1 (a FIXED grid:
//               one counter follows a super-tile through all its updates)
// (No counter follows the chain: whatever it produces is taken from a mailbox, see below.)
// The task list is sorted by a priority that is also a topological order: key = k + 0.3 x (distance of the super-column from the
// front) for an update, slightly less than k for PrepX / PrepD / a solve of step k.  A task only waits on the chain
// (resident from the first cycle) and on tasks EARLIER in the list; every XCD dispatches its share of the grid in increasing
// block index, so the lowest unfinished task is always resident with its inputs complete: the grid cannot deadlock however
// few workgroups fit the chip.  Every wait is bounded all the same: a timeout fails the factorisation (error bit 4), the other
// waits see the bit and leave, the host repeats the solve with per-column launches and stays with them.
// Hand-offs FROM and TO the chain (chain -> PrepX / PrepD / solve tasks, PrepX / PrepD -> chain) do not go through a counter at all: a memory round trip is
// 0.7 us idle and 1.5-2.5 us under load, and "store, wait for the acknowledgement, set a counter, poll it, load the data" is
// four of them per direction.  Instead every such tile has a MAILBOX slot of its own per step (M_k, the chain's X and D
// inputs, the solved X tile), preset to an all-ones NaN pattern by k_potrf64's side workgroups; the producer just stores, the
// consumer loads the data with agent-scope loads until no element is the pattern: one round trip per direction.
// Coherence without fences (the XCD L2s are not coherent inside a kernel, MI355X_MICROARCH.md "Correctness boundaries"):
//   * everything a workgroup hands to another one is stored write-through (agent-scope stores);
//   * tiles of S that are REWRITTEN during the launch (the running Schur complement) and the mailboxes (read before and
//     after they are written) are read with agent-scope loads;
//   * data written exactly ONCE per launch and only read after a counter says so can not be stale in any L2 and is read with
//     plain, L2-cached loads: the solved tiles live in their own array L (the factor ends up there; the back substitution reads
//     it from there), and so does M_k for the solve tasks.
constexpr int kMaxSteps = 128;       // block columns the counter arrays hold (N <= 8192)
constexpr int kMaxSuper = kMaxSteps / 2 + 1;
constexpr int kScratchCounters = 2048;      // counters of the per-chain accumulation sequences (several chains, see ChainRanges), handed out by the host
enum { cSol0 = 8, cVer0 = cSol0 + kMaxChains * kMaxSteps, cSub0 = cVer0 + kMaxSuper * kMaxSuper, cScratch0 = cSub0 + kMaxSuper * kMaxSuper,
       kNumCounters = cScratch0 + kScratchCounters };      // sol: one set of row counters per chain (cSol0 + chain x kMaxSteps + row)
static_assert(kNumCounters * sizeof(int32_t) <= 8192 * sizeof(double), "counters exceed their part of the workspace (CholeskyWorkspaceDoubles)");
enum { kTaskPrepX = 1, kTaskPrepD = 2, kTaskSolve = 3, kTaskUpdate = 4, kTaskPairPrep = 5, kTaskMerge = 6 };      // pair prep: a = pair, b = part (paired back substitution)
// solve: a = block row; update: a = I, b = J | part << 8 | parts << 12 | target << 16: a PART of super-tile (I,J) - parts = 2: block row
// 2I + part (both block columns); parts = 4: the one 64x64 tile (2I + part / 2, 2J + part % 2).  The part that brings the
// super-tile's sub-counter to `target` (the parts listed for it so far) moves its ver counter.
// w0, w1: the values the task's ver counters must have reached (the panels an EXISTING earlier task applies; in a dense system k - 1).  In a
// block-sparse system a panel only touches the super-tiles whose tiles it couples, so "every panel below k" becomes "the last panel below k that has
// an update task for this super-tile" - which only the host, who lists the tasks, knows.
// w2: the value the row counter sol[] of the row the task SOLVES a tile of must have reached - the row's previous structurally non-zero column, solved:
// the solves of a row stay in column order (a counter value then says "every non-zero column below it is solved"), whichever columns exist.
// Values a task STORES into a counter come from the host as well (they were k / k + 1 while the block columns were eliminated in index order):
//   PrepX / PrepD  a = the value "column k-1 of a row is solved" (0 for the first step of a chain), b = "column k is solved"
//   solve          w1 = "column k is solved" (stored into sol[i])
//   update         w1 = the value the super-tile's ver counter takes once every part of this panel is applied, w2 = "column k-1 is solved"
// flags: bit 0 = k is the FIRST block column of a chain (nothing pending from a column k-1; M_k is k_potrf64's); bits 4..7 = the chain whose row counters the
// task waits for / moves (the chain of block column k; of column k-1 for an update)
// update / merge tasks: cidx / sidx = the ver / sub counter of the sequence the task belongs to (absolute index), zsel = -1: the tiles of S themselves, >= 0:
// the chain whose scratch tiles the task accumulates into (update) or adds to S (merge: cidx = the super-tile's own ver counter, sidx = the scratch sequence's, w2 = the
// value that one must have reached), mask = bits 0..3: tiles of the super-tile nothing has been accumulated into yet (update: taken as zero instead of read;
// merge: the tiles to add)
// slot[q]: where tile q (2 x row + column) of the super-tile lives in the scratch pool (64 x 64 doubles per slot, row stride 64) when zsel >= 0
struct ChainTask { int32_t type, k, a, b, w0, w1, w2, flags, cidx, sidx, zsel, mask, slot[4]; };
constexpr int kPartsTwoPanels = 8;      // `parts` of an update task that applies panels k-1 and k to its whole super-tile (far from the front)
constexpr int kSpinBound = 1 << 21;
// Super-tile columns this far right of the front are updated whole, nearer ones in two halves.  Halves keep the per-super-tile
// sequence of updates shorter than a step of the chain (they cannot fall behind), whole super-tiles move the least operand bytes
// per flop: the smaller the matrix, the more the chain bounds the time and the further out halves pay.  Measured optimum
// (tools/chol_time.py with PPSFM_CHOL_WHOLE_FROM), round 2, priority slope 0.5: 12 at 47 block columns (0.73 against 0.77 ms with 6), 9 at 63,
// 6 at 79, 3 at 94.  Round 3 (tools/sched_sweep.sh, the knobs swept on one box): what the chain still waited for in steps 8-18 of a
// 47-column factorisation (~45 us in all) was the BULK - every CU busy with updates, the front updates of the step dispatched late - and
// not the position of PrepX / PrepD in the list (moving them one or two steps ahead changed nothing); and the bulk of the early steps is
// bound by its TRAFFIC (~800 tiles per step x ~100 KB per tile and panel = 6 TB/s).  So: (a) a flatter priority (far updates deferred
// by 0.3 instead of 0.5 steps per super-column: less of the far work piles up behind the front later on; steeper ones are much worse -
// 0.75: 804 us, 1.0: 887 us at 47 columns) with whole super-tiles five columns nearer: 743 -> 728 us (factorisation + back
// substitution in the tool); (b) far super-tiles take TWO panels per task (UpdateSuperTile<true>: C read and written once per two
// steps, the second panel's operands in flight under the first panel's products), and with that "far" starts three super-columns from
// the front: 47 columns 728 -> 706 us, 63: 1262 -> 1120, 79: 2135 -> 1765 (whole_from 2), 16 - 32 columns unchanged.
constexpr double kUpdateSlope = 0.3;
static int WholeFrom(int T) { return T >= 56 ? 2 : 3; }
constexpr unsigned long long kPoison = 0xFFFFFFFFFFFFFFFFull;

// mailboxes of one factorisation: 64x64 row-major slots (stride 64), one per step
struct Mailboxes {
  double* Minv;   // [T]    M_k       chain(k-1) -> PrepX(k), PrepD(k), the solve tasks of column k
  double* xs;     // [T+1]  X of chain(k) = tile (k+1,k), panels <= k-1 applied          PrepX(k-1) -> chain(k)
  double* ds;     // [T+1]  D of chain(k) = tile (k+1,k+1), panels <= k-1 applied        PrepD(k-1) -> chain(k)
  double* xsol;   // [T+1]  the solved tile (k+1,k)                                      chain(k) -> PrepX(k)
  double* Pw;     // pair inverses and pair couplings of the paired back substitution (kTaskPairPrep -> k_backsub_pairs)
  double* Zw;
};

__device__ __forceinline__ void WaitOwnStores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// every store of a task is a write-through store: once this workgroup's have been acknowledged a counter may move
__device__ __forceinline__ void TaskStoresDone() {
  WaitOwnStores();
  __syncthreads();
}
__device__ __forceinline__ bool IsPoison(double v) { return (unsigned long long)__double_as_longlong(v) == kPoison; }
// a result that happens to be the mailbox pattern (only a NaN can be) is stored as another NaN
__device__ __forceinline__ void StoreMail(double* p, double v) { StoreThrough(p, IsPoison(v) ? __longlong_as_double(0x7FF8000000000000ll) : v); }

// All threads call; true once every one of the n <= 6 counters has reached its value.  Lane i of wavefront 0 polls counter i,
// so the wait costs ONE memory round trip, not one per counter; the other wavefronts wait at the barrier.  *s_failed is sticky
// (zeroed by thread 0 at kernel start, before the first barrier), so one barrier suffices.
// (scalars, not arrays: a lane-indexed array of pointers ends up in scratch memory)
struct WaitList {
  const int32_t *p0 = nullptr, *p1 = nullptr, *p2 = nullptr, *p3 = nullptr, *p4 = nullptr;
  int n0 = 0, n1 = 0, n2 = 0, n3 = 0, n4 = 0;      // <= 0: nothing to wait for in this slot
};
__device__ __forceinline__ bool TaskWait(const WaitList& wl, int32_t* flag, int* s_failed, unsigned long long* last_missing = nullptr) {
  if (threadIdx.x < 64) {
    const int l = threadIdx.x;
    const int32_t* p = l == 0 ? wl.p0 : (l == 1 ? wl.p1 : (l == 2 ? wl.p2 : (l == 3 ? wl.p3 : wl.p4)));
    const int need = l == 0 ? wl.n0 : (l == 1 ? wl.n1 : (l == 2 ? wl.n2 : (l == 3 ? wl.n3 : wl.n4)));
    bool mine = l >= 5 || need <= 0;
    for (int spins = 0;; ++spins) {
      if (last_missing && l == 0) *last_missing = __ballot(!mine) | ((unsigned long long)spins << 8);      // (trace builds: what the last round still waited for)
      if (!mine) mine = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= need;
      if (__all(mine)) break;
      bool give_up = spins >= kSpinBound;
      if ((spins & 1023) == 1023) give_up = give_up || (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 4);   // somebody else gave up
      if (give_up) { if (l == 0) { atomicOr(flag, 4); *s_failed = 1; } break; }
      __builtin_amdgcn_s_sleep(1);
    }
  }
  __syncthreads();
  return *s_failed == 0;
}

// A whole mailbox tile -> LDS (row stride kLS), all 1024 threads.  One attempt with everybody (agent-scope loads; the common case:
// the tile has been there for a while - ONE memory round trip); if any element still is the pattern, ONE wavefront samples 64
// elements of the slot (one per row) until none is, the others sleep at the barrier (a workgroup that polls with all its threads
// reads 32 KB per round from below the L2s), then everybody tries again.  Bounded; false (after a barrier) if the wait gave up.
// `deposit` runs once, right after the first attempt's loads have been issued: the place for the LDS stores of tiles whose loads
// the caller issued just before the call - all of them share one memory round trip instead of queueing up behind each other.
// kEager (the two prep workgroups, whose answer the chain is waiting for): every thread polls its own elements instead - the tile
// is in LDS one round trip after it became visible, not two.
struct NoDeposit { __device__ void operator()() const {} };
template <bool kEager = false, typename Deposit = NoDeposit>
__device__ __forceinline__ bool FetchMailTile(double* dst, const double* __restrict__ src, int tid, int32_t* flag, int* s_failed, Deposit deposit = Deposit()) {
  const int r0 = tid >> 5, c2 = tid & 31;       // rows r0 and r0 + 32, columns 2 c2, 2 c2 + 1
  const double* p0 = src + (size_t)r0 * kNB + 2 * c2;
  const double* p1 = p0 + (size_t)32 * kNB;
  const double* probe = src + (size_t)(tid & 63) * kNB + (((tid & 63) * 5) & 63);
  bool mine = false;
  for (int spins = 0;; ++spins) {
    if (!mine) {
      const double a = LoadCoherent(p0), b = LoadCoherent(p0 + 1), c = LoadCoherent(p1), d = LoadCoherent(p1 + 1);
      if (spins == 0) deposit();
      if (!IsPoison(a) && !IsPoison(b) && !IsPoison(c) && !IsPoison(d)) {
        *reinterpret_cast<double2*>(dst + r0 * kLS + 2 * c2) = make_double2(a, b);
        *reinterpret_cast<double2*>(dst + (r0 + 32) * kLS + 2 * c2) = make_double2(c, d);
        mine = true;
      }
    }
    if (kEager) {
      if (mine) break;
      bool give_up = spins >= kSpinBound;
      if ((spins & 1023) == 1023) give_up = give_up || (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 4);
      if (give_up) { atomicOr(flag, 4); *s_failed = 1; break; }
      continue;
    }
    if (__syncthreads_and(mine)) break;        // (also the barrier that publishes the LDS tile)
    if (tid < 64) {
      for (int inner = 0;; ++inner) {
        if (__all(!IsPoison(LoadCoherent(probe)))) break;
        bool give_up = inner >= kSpinBound || spins >= 4096;
        if ((inner & 1023) == 1023) give_up = give_up || (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 4);
        if (give_up) { if (tid == 0) { atomicOr(flag, 4); *s_failed = 1; } break; }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    __syncthreads();
    if (*s_failed) break;
  }
  if (kEager) __syncthreads();
  return *s_failed == 0;
}

// The chain workgroup.  Per step k: X M_k^T, D -= X X^T, the panels (M_{k+1} built beside them), stores.  What keeps a step at
// the length of its arithmetic:
//   * M_k never leaves LDS: bufM holds it (built by the previous step), bufX takes X and is rebuilt into M_{k+1}; the two swap;
//   * the NEXT step's X and D tiles come out of their mailboxes through the nine wavefronts that idle through panels 2 and 3:
//     loads issued beside panel 2, looked at beside panel 3 (re-issued if the tile was not there yet) and after the last panel
//     - nothing blocks while wavefront 0 is in a panel - and written to LDS (X into the then free M_k buffer, D into the
//     buffer of the solved X, dead after panel 1); only after the last panel do they wait, bounded, for PrepX / PrepD;
//   * nothing waits for the step's own stores: the solved X goes out beside panel 0, M_{k+1} after the last panel, both to
//     mailboxes their consumers poll (no acknowledgement, no counter).
// Every pointer and thread index is laundered through an empty asm per step: inlined into the k-loop, the loop-invariant
// arithmetic the compiler hoists pushed the 128-VGPR body into spills; a real call cost a 48-register save / restore per step.
template <typename P>
__device__ __forceinline__ P* Launder(P* p) { long long z = 0; asm volatile("" : "+s"(z)); return p + z; }      // (an opaque zero OFFSET: the pointer keeps its address space - laundering the pointer itself makes every access through it a FLAT one)

// kb, ke: the chain's block columns [kb, ke) (0, T for the only chain of a system); post: see ChainRanges
__device__ __forceinline__ void ChainLoop(double* S_, double* L_, int ld_, int T, Mailboxes mb_, int32_t* flag_, int32_t* ctr_, double* smem_, double* inv_diag_,
                                          int* s_failed, int kb, int ke, int post, int solbase) {
  int swap = 0;
  // X of the first step (k_potrf64's staging copy), M_kb and the raw D of the first step
  LoadTile(smem_, mb_.xs + (size_t)kb * kNB * kNB, kNB, threadIdx.x);
  LoadTile(smem_ + kNB * kLS, mb_.Minv + (size_t)kb * kNB * kNB, kNB, threadIdx.x);
  LoadTile(smem_ + 3 * kNB * kLS, S_ + (size_t)(kb + 1) * kNB * ld_ + (size_t)(kb + 1) * kNB, ld_, threadIdx.x);
  const int wave_index = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  for (int k = kb; k + 1 < ke; ++k) {
    double* S = Launder(S_); double* L = Launder(L_); int32_t* flag = Launder(flag_);
    double* smem = smem_; double* inv_diag = inv_diag_;      // LDS: compile-time addresses - laundering them would turn every LDS access into a FLAT one
    double* mbM = Launder(mb_.Minv); const double* mbX = Launder(mb_.xs); const double* mbD = Launder(mb_.ds); double* mbS = Launder(mb_.xsol);
    int ld = ld_; asm volatile("" : "+s"(ld));
    // the thread index, re-derived per step from the lane count of the wavefront and its index in the workgroup (an SGPR): taken
    // from threadIdx.x it was one more VGPR live around the whole loop - the one the compiler spilled, and reloaded from scratch in
    // four places of the step (every index below is re-derived per step: none stays live around the loop)
    int tid;
    {
      int wv = wave_index; asm volatile("" : "+s"(wv));
      int ln; asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
      tid = wv * 64 + ln;
    }
    (void)S;
    const int lane = tid & 63, w = tid >> 6, lr = lane & 15, g = lane >> 4;
    const bool dlate = w >= 5 && w < 12 && (w & 3) != 0;
    const bool dwave = w < 4 || dlate;
    int dti = w & 3, dtj = 0;
    if (dlate) { if (w < 8) { dti = w - 4; dtj = 1; } else { dti = w == 9 ? 2 : 3; dtj = w == 11 ? 3 : 2; } }     // as in ChainBody
    // the wavefronts that idle through panels 2 and 3 (PotrfPanels: 4, 7..14) and do NOT share wavefront 0's SIMD (w & 3 == 0): 7, 9, 10, 11, 13, 14
    // -> 0..5.  Wavefronts 4, 8, 12 fetched too at first: behind wavefront 0's raised priority they got through their 16 load instructions
    // 0.85 us AFTER the panel had ended (barrier arrivals of step 20: 6.72 / 7.08 / 7.20 us against 6.36 for wavefront 0 and 5.85 for the
    // six others) and slowed the panel itself (1.88 against 1.52 us): the panel-2 phase lasted 2.4 us.
#ifdef PP_FETCH_9WAVES      // (A/B switch for tools/chol_task_trace.hip: the former assignment)
    const int spare_rank = w == 4 ? 0 : w - 6;
    constexpr int kFetchWaves = 9, kFetchRounds = 4;
#else
    const int spare_rank = (w & 3) == 0 ? -1 : (w == 7 ? 0 : (w < 12 ? w - 8 : w - 9));
    constexpr int kFetchWaves = 6, kFetchRounds = 6;       // 32 row pairs over six wavefronts
#endif
    double* bufX = smem + (swap ? kNB * kLS : 0);
    double* bufM = smem + (swap ? 0 : kNB * kLS);
    double* BD = smem + 2 * kNB * kLS;
    double* BS = smem + 3 * kNB * kLS;
    PP_TASK_MAX(1, k);
#ifdef PP_CHOL_TRACE
    if (tid == 0) g_chol_step = k;
#endif
    const size_t xbase = (size_t)(k + 1) * kNB * ld + (size_t)k * kNB;
    PP_CHAIN_PHASE(0, k);
    PP_WAVE_ARRIVE(8);
    __syncthreads();          // X is in bufX, D in BS (fetched during the previous step, or above)
    PP_CHAIN_PHASE(1, k);
    if (*s_failed) return;
    v4f64 d = (v4f64){0.0, 0.0, 0.0, 0.0};
    if (dwave) {
#pragma unroll
      for (int i = 0; i < 4; ++i) d[i] = PP_TILE(BS, dti, dtj)[(g + 4 * i) * kLS + lr];
    }
    const int s = w & 3, ct = w >> 2;
    const v4f64 x = SolveTile(bufX, bufM, s, ct, lr, g);
    PP_WAVE_ARRIVE(9);
    __syncthreads();          // every wavefront has its D tile out of BS
    TileStoreD(PP_TILE(BS, s, ct), x, lr, g);
    // the solved X straight from the registers to its mailbox (PrepX(k), PrepX(k+1), the solve tasks of column k+1 poll it): PrepX(k)'s
    // answer - the next step's X - is due before this step's last panel ends, and every microsecond the tile leaves later comes back
    // as a wait of the spare wavefronts after that panel
    if (!PP_EXP(1)) {
      double* mail = mbS + (size_t)k * kNB * kNB;
#pragma unroll
      for (int r = 0; r < 4; ++r) StoreMail(mail + (size_t)(16 * s + g + 4 * r) * kNB + 16 * ct + lr, x[r]);
    }
    PP_CHAIN_PHASE(2, k);
    PP_WAVE_ARRIVE(10);
    __syncthreads();
    PP_CHAIN_PHASE(3, k);
    ZeroTileFresh(bufX, tid);
    if (w < 4) {
      d = UpdateTileRegs(d, BS, BS, dti, dtj, lr, g);
      TileStoreD(PP_TILE(BD, dti, dtj), d, lr, g);
    }
    PP_WAVE_ARRIVE(11);
    __syncthreads();
    PP_CHAIN_PHASE(4, k);
    const bool has_next = k + 2 < ke;
    auto side = [&](int wv) {
      if (dlate) {
        if (dtj == 1) d = UpdateTileRegs(d, BS, BS, dti, dtj, lr, g);
        TileStoreD(PP_TILE(BD, dti, dtj), d, lr, g);
      } else if ((wv & 3) != 0) {
        // wavefronts 1,2,3,13,14,15 (idle beside panel 0).  The LAST step
        // has no PrepX task that copies its solved X from the mailbox to L (the back substitution reads it there): stored here.
        if (!has_next && !PP_EXP(2)) {
          const int p = (wv < 4 ? wv - 1 : wv - 10) * 64 + lane;
          for (int idx = p; idx < 2048; idx += 384) {
            const int r = idx >> 5, c2 = idx & 31;
            const double2 v = *reinterpret_cast<const double2*>(BS + r * kLS + 2 * c2);
            StoreThrough(L + xbase + (size_t)r * ld + 2 * c2, v.x);
            StoreThrough(L + xbase + (size_t)r * ld + 2 * c2 + 1, v.y);
          }
        }
      }
    };
    auto side1 = [&](int) {
      if (dlate && dtj != 1) UpdateTileInPlace(BD, BS, BS, dti, dtj, lr, g);
    };
    // the next step's X and D tiles: mailbox -> LDS by the nine spare wavefronts, BOTH tiles in flight together and WITHOUT passing
    // through registers: `global_load_lds_dwordx4` (gfx950) writes 16 bytes per lane straight into LDS at M0 + 16 x lane, agent scope
    // (sc1).  Through registers one tile at a time was all the spare branch could hold (the values that live across PotrfPanels
    // leave it ~20 VGPRs: two tiles' worth spilled, and a spill after a load is a blocking wait) - and then the D request only left
    // when X had arrived, a memory round trip after the last panel whenever X missed the first look.
    // A row of the padded LDS tile is 32 lanes x 16 bytes, so a wavefront loads two rows with two half-wave instructions (the second
    // one's M0 points 512 bytes before its row: lanes 32..63 write at M0 + 512 ..).  Arrival is checked by reading the tile back from
    // LDS (after s_waitcnt vmcnt(0)): if an element still is the mailbox pattern, the wavefront's share is requested again.
    //   per tile: 0 = not requested, 1 = in flight, 2 = in LDS;  stage 4 = both in LDS
    const double* srcX = mbX + (size_t)(k + 1) * kNB * kNB;
    const double* srcD = mbD + (size_t)(k + 1) * kNB * kNB;
    int sx = 0, sd = 0;
    int stage = (has_next && spare_rank >= 0) ? 0 : 4;
    auto issue = [&](const double* src, double* dst) {
#pragma unroll
      for (int j = 0; j < kFetchRounds; ++j) {
        const int pair = spare_rank + kFetchWaves * j;       // rows 2 pair, 2 pair + 1 (wave-uniform)
        if (pair < 32) {
          const int r = 2 * pair + (lane >> 5), c = lane & 31;
          const double* g = src + (size_t)r * kNB + 2 * c;
          // both half-wave loads in ONE asm block with its own EXEC halves: written as `if (lane < 32) load(row) else load(row + 1)` with
          // the builtin, the compiler merged the two into one instruction whose LDS base came from lane 0 (rows 2 pair + 1 landed 16
          // bytes off)
          const int lo = __builtin_amdgcn_readfirstlane((int)(size_t)(__attribute__((address_space(3))) void*)(dst + (2 * pair) * kLS));
          const int hi = __builtin_amdgcn_readfirstlane((int)(size_t)(__attribute__((address_space(3))) void*)(dst + (2 * pair + 1) * kLS - 64));
          unsigned long long saved;
          int saved_m0;      // (M0 is a reserved register: saved and restored rather than declared clobbered)
          asm volatile("s_mov_b64 %[sv], exec\n\t"
                       "s_mov_b32 %[m0s], m0\n\t"
                       "s_mov_b32 m0, %[lo]\n\t"
                       "s_mov_b32 exec_lo, -1\n\t"
                       "s_mov_b32 exec_hi, 0\n\t"
                       "global_load_lds_dwordx4 %[g], off sc1\n\t"
                       "s_mov_b32 m0, %[hi]\n\t"
                       "s_mov_b32 exec_lo, 0\n\t"
                       "s_mov_b32 exec_hi, -1\n\t"
                       "global_load_lds_dwordx4 %[g], off sc1\n\t"
                       "s_mov_b32 m0, %[m0s]\n\t"
                       "s_mov_b64 exec, %[sv]"
                       : [sv] "=&s"(saved), [m0s] "=&s"(saved_m0) : [g] "v"(g), [lo] "s"(lo), [hi] "s"(hi) : "memory");
        }
      }
    };
    auto arrived = [&](const double* dst) -> bool {      // wave-uniform: every element of this wavefront's share is there
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      bool ok = true;
#pragma unroll
      for (int j = 0; j < kFetchRounds; ++j) {
        const int pair = spare_rank + kFetchWaves * j;
        if (pair < 32) {
          const int r = 2 * pair + (lane >> 5), c = lane & 31;
          const double2 v = *reinterpret_cast<const double2*>(dst + r * kLS + 2 * c);      // (re-read each time: the asm above clobbers memory)
          ok = ok && !IsPoison(v.x) && !IsPoison(v.y);
        }
      }
      return __all(ok);
    };
    auto advance = [&]() {      // one move; X -> the M_k buffer (free since the solve), D -> BS (free since panel 1)
      if (stage == 4) return;
      const bool look_x = sx == 1, look_d = sd == 1;
      bool have_x = sx == 2, have_d = sd == 2;
      if (look_x) have_x = arrived(bufM);
      if (look_d) have_d = arrived(BS);
      if (!have_x) { issue(srcX, bufM); sx = 1; } else sx = 2;
      if (!have_d) { issue(srcD, BS); sd = 1; } else sd = 2;
      if (sx == 2 && sd == 2) stage = 4;
    };
    // beside panel 2: both requested; beside panel 3: what has arrived goes to LDS, the rest is requested again; after the last
    // panel: whatever has not arrived yet is waited for (bounded)
    auto spare_job = [&](int phase) {
      if (phase < 2) { if (!PP_EXP(8)) advance(); return; }
#ifdef PP_CHOL_TRACE
      const long long wait_t0 = wall_clock64();      // (wavefront 4 only writes it: how long the step waited for its next inputs after the last panel)
      const int wait_stage0 = sx * 4 + sd;
#endif
      for (int spins = 0; stage != 4; ++spins) {
        advance();
        if (stage == 4) break;
        bool give_up = spins >= kSpinBound;
        if ((spins & 255) == 255) give_up = give_up || (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 4);
        if (give_up) { if (lane == 0) { atomicOr(flag, 4); *s_failed = 1; } break; }
      }
#ifdef PP_CHOL_TRACE
      if (w == 7 && lane == 0 && k < 128) { g_spare_wait[0][k] = wall_clock64() - wait_t0; g_spare_wait[1][k] = wait_stage0; g_spare_wait[2][k] = wait_t0; }
      if (has_next && spare_rank >= 0 && PP_EXP(16)) {      // check (switch 16 of tools/chol_task_trace.hip): what sits in LDS against what the mailbox holds now
        for (int which = 0; which < 2; ++which) {
          const double* src = which ? srcD : srcX; const double* dst = which ? BS : bufM;
          for (int j = 0; j < kFetchRounds; ++j) {
            const int pair = spare_rank + kFetchWaves * j;
            if (pair < 32) {
              const int r = 2 * pair + (lane >> 5), c = lane & 31;
              const double a = dst[r * kLS + 2 * c], b = dst[r * kLS + 2 * c + 1];
              const double ga = LoadCoherent(src + (size_t)r * kNB + 2 * c), gb = LoadCoherent(src + (size_t)r * kNB + 2 * c + 1);
              if (__double_as_longlong(a) != __double_as_longlong(ga) || __double_as_longlong(b) != __double_as_longlong(gb)) atomicAdd(&g_dbg_mismatch[which * 8 + (lane >> 5) * 4 + (j & 3)], 1);
            }
          }
        }
      }
#endif
    };
    PotrfPanels(BD, bufX, inv_diag, flag, lane, w, side, side1, spare_job);
    PP_CHAIN_PHASE(5, k);
    {      // M_{k+1} -> its mailbox (PrepX(k+1) / PrepD(k+1) and the solve tasks of column k+1 are polling it): two 16-byte pieces per thread
      double* mail = mbM + (size_t)(k + 1) * kNB * kNB;
#pragma unroll
      for (int it = 0; it < 2 && !PP_EXP(4); ++it) {
        const int idx = tid + kPanelThreads * it, r = idx >> 5, c2 = idx & 31;
        const double2 v = *reinterpret_cast<const double2*>(bufX + r * kLS + 2 * c2);
        StoreMail(mail + (size_t)r * kNB + 2 * c2, v.x);
        StoreMail(mail + (size_t)r * kNB + 2 * c2 + 1, v.y);
      }
    }
    if (k + 2 == T) StoreTile(L + xbase + kNB, BD, ld, tid);      // the last diagonal block holds part of the right-hand side's row: the back substitution reads it
    PP_CHAIN_PHASE(6, k);
    PP_TASK_MAX(2, k);
    swap ^= 1;
  }
  // the last step's stores (nothing waits for them inside the kernel - unless another chain follows: the update tasks of panel ke - 2 read the solved tile
  // (ke-1,ke-2) from L once the row's counter says so)
  TaskStoresDone();
  if (ke < T && threadIdx.x == 0) __hip_atomic_store(ctr_ + solbase + (ke - 1), post, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ int32_t* VerCounter(int32_t* ctr, int I, int J) { return ctr + cVer0 + I * kMaxSuper + J; }

// PrepX(k) / PrepD(k): everything that does not need M_k (the panel k-1 updates of the three tiles) is done before the
// workgroup looks for M_k's mailbox; after it has arrived one solve and one rank-64 update remain.  PrepX takes the solved tile
// (k+1,k) from the chain's mailbox instead of solving it a second time.  Results for the chain go to the xs / ds mailboxes;
// solved tiles go to L, so PrepX never overwrites what PrepD still reads.
template <bool kIsX>
__device__ __forceinline__ void PrepTask(double* S, double* L, int ld, int k, Mailboxes mb, int32_t* __restrict__ flag, int32_t* __restrict__ ctr, int* s_failed,
                                         double* Ba, double* Bb, double* Bc, double* Bm, int w0, int w1, int w2, bool far_nz, bool first, int done_km1, int done_k, int solbase) {
  // far_nz: tile (k+2,k-1) is structurally non-zero; first: k starts a chain; done_km1 / done_k: the counter values "column k-1 / k of a row is solved"
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, lr = lane & 15, g = lane >> 4;
  const int ti = w >> 2, tj = w & 3, s = w & 3, ct = w >> 2;
  const bool prev = !first;
  const size_t row_k1 = (size_t)(k + 1) * kNB * ld, row_k2 = (size_t)(k + 2) * kNB * ld;
  const size_t col_km1 = (size_t)(k - 1) * kNB, col_k = (size_t)k * kNB, col_k1 = (size_t)(k + 1) * kNB, col_k2 = (size_t)(k + 2) * kNB;
  int di = 0, dj = 0;
  if (!kIsX) { int rem = w; while (rem > di) { rem -= di + 1; ++di; } dj = rem; }
  const bool has_out = kIsX || w < 10;
  const size_t obase = kIsX ? row_k2 + col_k1 + (size_t)(16 * ti) * ld + 16 * tj : row_k2 + col_k2 + (size_t)(16 * di) * ld + 16 * dj;
  // ---- phase A: the tiles with the panels <= k-2 applied, column k-1 of rows k+1, k+2 solved
  PP_TASK_MAX(kIsX ? 19 : 20, k);
  if (prev) {
    WaitList wl;
    wl.p0 = VerCounter(ctr, (k + 2) >> 1, k >> 1); wl.n0 = w0;                                             // tile (k+2,k)
    wl.p1 = VerCounter(ctr, (k + 2) >> 1, kIsX ? (k + 1) >> 1 : (k + 2) >> 1); wl.n1 = w1;                // the output tile
    wl.p3 = ctr + solbase + (k + 2); wl.n3 = kIsX ? w2 : (far_nz ? done_km1 : 0);      // (PrepX moves this counter: behind the row's previous non-zero column, w2 >= k if far_nz)
    // PrepX: A_{k+1,k-1} = PrepX(k-1)'s solved tile, in the SAME wait (round 5).  The wait above ends with the solve task of tile (k+2,k-1), ~6 us after
    // M_{k-1} exists; PrepX(k-1) moves this counter ~1.5 us earlier, so asking for it here costs nothing and saves the second wait's round trip and the
    // tile's own load round trip below: phase A was 8.7 us (tools/chol_task_trace, round 5), the PrepX(k-1) -> PrepX(k) -> chain cycle 14 us per step
    if (kIsX) { wl.p2 = ctr + solbase + (k + 1); wl.n2 = done_km1; }
    if (!TaskWait(wl, flag, s_failed)) return;
  }
  PP_TASK_MAX(kIsX ? 3 : 11, k);
  v4f64 out = (v4f64){0.0, 0.0, 0.0, 0.0};
  if (has_out) {
#pragma unroll
    for (int i = 0; i < 4; ++i) out[i] = LoadCoherent(S + obase + (size_t)(g + 4 * i) * ld + lr);
  }
  // (k+2,k) with the panels <= k-2 applied, A_{k+2,k-1}, and A_{k,k-1} = the solved X tile of chain(k-1) from its mailbox (in there
  // since early in that step): every load of phase A in flight together
  const double2 c0 = TileLoad2T<true>(S + row_k2 + col_k, ld, tid, 0), c1 = TileLoad2T<true>(S + row_k2 + col_k, ld, tid, 1);
  if (prev) {
    // (a structurally zero tile (k+2,k-1) was never solved: its place in L holds nothing - an exact zero tile instead)
    const double2 zz = make_double2(0.0, 0.0);
    const double2 b0 = far_nz ? TileLoad2(L + row_k2 + col_km1, ld, tid, 0) : zz, b1 = far_nz ? TileLoad2(L + row_k2 + col_km1, ld, tid, 1) : zz;
    // (PrepX: A_{k+1,k-1} rides in the same round trip, into the buffer M_k takes later)
    const double2 m0 = kIsX ? TileLoad2(L + row_k1 + col_km1, ld, tid, 0) : zz, m1 = kIsX ? TileLoad2(L + row_k1 + col_km1, ld, tid, 1) : zz;
    auto deposit = [&]() {
      TileStore2(Bc, tid, 0, c0); TileStore2(Bc, tid, 1, c1); TileStore2(Bb, tid, 0, b0); TileStore2(Bb, tid, 1, b1);
      if (kIsX) { TileStore2(Bm, tid, 0, m0); TileStore2(Bm, tid, 1, m1); }
    };
    if (!FetchMailTile<false>(Ba, mb.xsol + (size_t)(k - 1) * kNB * kNB, tid, flag, s_failed, deposit)) return;
    UpdateTileInPlace(Bc, Bb, Ba, ti, tj, lr, g);
    if (kIsX) out = UpdateTileRegs(out, Bb, Bm, ti, tj, lr, g);
    else if (has_out) out = UpdateTileRegs(out, Bb, Bb, di, dj, lr, g);
    __syncthreads();                                                                 // Bm's readers are done before M_k lands in it
  }
  // ---- phase B: M_k (its mailbox; step 0's is k_potrf64's)
  PP_TASK_MAX(kIsX ? 12 : 13, k);
  if (prev) { if (!FetchMailTile<true>(Bm, mb.Minv + (size_t)k * kNB * kNB, tid, flag, s_failed)) return; }
  else { TileStore2(Bc, tid, 0, c0); TileStore2(Bc, tid, 1, c1); LoadTile(Bm, mb.Minv + (size_t)k * kNB * kNB, kNB, tid); __syncthreads(); }
  PP_TASK_MAX(kIsX ? 4 : 14, k);
  const v4f64 x = SolveTile(Bc, Bm, s, ct, lr, g);                                   // A_{k+2,k}
  __syncthreads();
  TileStoreD(PP_TILE(Bc, s, ct), x, lr, g);
  if (kIsX) {
#pragma unroll
    for (int r = 0; r < 4; ++r) StoreThrough(L + row_k2 + col_k + (size_t)(16 * s + g + 4 * r) * ld + 16 * ct + lr, x[r]);
    if (!FetchMailTile<true>(Ba, mb.xsol + (size_t)k * kNB * kNB, tid, flag, s_failed)) return;      // the solved tile (k+1,k), stored by chain(k) beside its first panel
    TaskStoresDone();      // (the fetch above was a memory round trip: the stores of A_{k+2,k} have been acknowledged)
    // row k+2: column k solved (in L) - PrepX(k+1)'s phase A waits for it: this counter, not the X below, is on the longest cycle of the factorisation
    // (round 5: with the product and the X stores ahead of it the chain waited ~2 us per step for its next X - 787 against 721 us)
    if (tid == 0) __hip_atomic_store(ctr + solbase + (k + 2), done_k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    out = UpdateTileRegs(out, Bc, Ba, ti, tj, lr, g);
    double* mail = mb.xs + (size_t)(k + 1) * kNB * kNB;
#pragma unroll
    for (int i = 0; i < 4; ++i) StoreMail(mail + (16 * ti + g + 4 * i) * kNB + 16 * tj + lr, out[i]);
    PP_TASK_MAX(5, k);
    StoreTile(L + row_k1 + col_k, Ba, ld, tid);      // the chain's solved tile (k+1,k) -> L (the chain itself only fills the mailbox: one store set less beside its first panel)
    TaskStoresDone();
    if (tid == 0) __hip_atomic_store(ctr + solbase + (k + 1), done_k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // row k+1: column k solved (in L)
  } else {
    __syncthreads();
    double* mail = mb.ds + (size_t)(k + 1) * kNB * kNB;
    if (has_out) {
      out = UpdateTileRegs(out, Bc, Bc, di, dj, lr, g);
#pragma unroll
      for (int i = 0; i < 4; ++i) StoreMail(mail + (16 * di + g + 4 * i) * kNB + 16 * dj + lr, out[i]);
    } else {      // the strictly upper 16x16 tiles of D are never read by the factorisation, but the chain waits for the WHOLE slot
      int ui = 0, uj = 1, rem = w - 10;      // wavefronts 10..15 -> (0,1) (0,2) (0,3) (1,2) (1,3) (2,3)
      while (rem >= 3 - ui) { rem -= 3 - ui; ++ui; }
      uj = ui + 1 + rem;
#pragma unroll
      for (int i = 0; i < 4; ++i) StoreThrough(mail + (16 * ui + g + 4 * i) * kNB + 16 * uj + lr, 0.0);
    }
    PP_TASK_MAX(6, k);
  }
}

// update of ONE super-tile (block rows 2I, 2I+1 x block columns 2J, 2J+1 of the FIXED grid) by panel kp at step k = kp + 1:
// the tiles of the region below / right of (k+1,k+1) except the three the chain and the prep tasks own at this step; per tile
// the same arithmetic, in the same order, as SyrkSuperTiles (each wavefront a 32x32 piece = 2x2 MFMA tiles, 16 k-slices)
// kTwo: panels kp AND kp + 1 in one pass over the super-tile (a super-tile far from the front: every tile of it takes both): the
// operands of the second panel are requested before the first panel's products and wait in registers, C is read and written once -
// (c - p_kp) - p_kp+1, the bits of two single passes.  The early steps of a factorisation are bound by the traffic of the updates
// (~100 KB moved per 64x64 tile and panel, ~800 tiles per step); a far super-tile has steps of slack for the second panel's solves.
template <bool kTwo>
__device__ __forceinline__ void UpdateSuperTile(double* S, const double* L, int ld, int kp, int T, int I, int J, double* As, double* Bs, const uint8_t* __restrict__ nz, int fresh_mask = 0, double* zpool = nullptr, const int32_t* slot = nullptr) {
  // zpool / slot: the tiles are accumulated in scratch tiles (slot[2 x row + column] of the pool, row stride 64) instead of S; fresh_mask: tiles that count as zero
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int lr = lane & 15, lk = lane >> 4;
  const int wi = w >> 2, wj = w & 3;
  const int k = kp + 1, bi0 = 2 * I, bj0 = 2 * J;
  const int bi = bi0 + (wi >> 1), bj = bj0 + (wj >> 1);
  const bool front = (bi == k + 1 && bj == k + 1) || (bi == k + 2 && (bj == k + 1 || bj == k + 2));
  bool valid = bi < T && bj < T && bi >= bj && bj >= k + 1 && !front;
  if (valid && nz) valid = nz[(size_t)bi * T + kp] && nz[(size_t)bj * T + kp];      // (block-sparse: the panel only touches the tiles it couples; the others' operands were never solved)
  const size_t cbase = (size_t)bi * kNB * ld + (size_t)bj * kNB + (size_t)(32 * (wi & 1) + lk) * ld + 32 * (wj & 1) + lr;
  const v4f64 z = (v4f64){0.0, 0.0, 0.0, 0.0};
  v4f64 c[2][2], p[2][2] = {{z, z}, {z, z}};
  const size_t col = (size_t)kp * kNB;
  const int ra1 = bi0 + 1 < T ? bi0 + 1 : bi0, rb1 = bj0 + 1 < T ? bj0 + 1 : bj0;      // a missing block: any valid address, its results are not stored
  const double* s0 = L + (size_t)bi0 * kNB * ld + col; const double* s1 = L + (size_t)ra1 * kNB * ld + col;
  const double* s2 = L + (size_t)bj0 * kNB * ld + col; const double* s3 = L + (size_t)rb1 * kNB * ld + col;
  LoadTiles4(As, s0, As + kNB * kLS, s1, Bs, s2, Bs + kNB * kLS, s3, ld, tid);
  const int tq = 2 * (wi >> 1) + (wj >> 1);
  const bool fresh = (fresh_mask >> tq) & 1;
  double* Cw = S + cbase;
  size_t ldc = (size_t)ld;
  if (zpool && valid) { Cw = zpool + (size_t)slot[tq] * kNB * kNB + (size_t)(32 * (wi & 1) + lk) * kNB + 32 * (wj & 1) + lr; ldc = kNB; }
  if (valid) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int i = 0; i < 4; ++i) c[a][b][i] = fresh ? 0.0 : LoadCoherent(Cw + (size_t)(16 * a + 4 * i) * ldc + 16 * b);
  }
  double2 nx[kTwo ? 8 : 1];
  if (kTwo) {      // the second panel's four operand tiles: in flight under the first panel's products
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      nx[it] = TileLoad2(s0 + kNB, ld, tid, it); nx[2 + it] = TileLoad2(s1 + kNB, ld, tid, it);
      nx[4 + it] = TileLoad2(s2 + kNB, ld, tid, it); nx[6 + it] = TileLoad2(s3 + kNB, ld, tid, it);
    }
  }
  __syncthreads();
  const double* ar = As + (32 * wi + lr) * kLS + lk;
  const double* br = Bs + (32 * wj + lr) * kLS + lk;
  auto products = [&]() {
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      const double a0v = ar[4 * kk], a1v = ar[16 * kLS + 4 * kk], b0v = br[4 * kk], b1v = br[16 * kLS + 4 * kk];
      p[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0v, b0v, p[0][0], 0, 0, 0);
      p[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0v, b1v, p[0][1], 0, 0, 0);
      p[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1v, b0v, p[1][0], 0, 0, 0);
      p[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1v, b1v, p[1][1], 0, 0, 0);
    }
  };
  if (valid) products();
  if (kTwo) {
    if (valid) {
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) { c[a][b] = c[a][b] - p[a][b]; p[a][b] = z; }
    }
    __syncthreads();      // every wavefront is done with the first panel's operands
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      TileStore2(As, tid, it, nx[it]); TileStore2(As + kNB * kLS, tid, it, nx[2 + it]);
      TileStore2(Bs, tid, it, nx[4 + it]); TileStore2(Bs + kNB * kLS, tid, it, nx[6 + it]);
    }
    __syncthreads();
    if (valid) products();
  }
  if (valid) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          __hip_atomic_store(Cw + (size_t)(16 * a + 4 * i) * ldc + 16 * b, c[a][b][i] - p[a][b][i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// The update of block row bi of a super-tile - block columns bj0 .. bj0 + nb - 1, nb <= 2 - by panel kp on all 16 wavefronts (one 16x16
// piece per wavefront and tile).  A whole super-tile on one workgroup is 11-12 us (its fp64 MFMA rate) plus the hand-over to the next
// panel's update of the same super-tile - as long as a step of the chain: the updates fell further behind with every step.  Halves
// (nb = 2) take 6 us; the super-tiles the next step's PrepX / PrepD wait for are done as four single tiles (nb = 1) on four CUs, 4 us.
// Per 16x16 piece the same arithmetic in the same order as SyrkSuperTiles (one accumulator over the 16 k-slices, then c - p).
__device__ __forceinline__ void UpdateTilesTask(double* S, const double* L, int ld, int kp, int bi, int bj0, bool valid0, bool valid1, double* At, double* Bt, bool fresh0 = false, bool fresh1 = false, double* z0 = nullptr, double* z1 = nullptr) {      // z0 / z1: scratch tiles (row stride 64) that take the place of the two tiles of S
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, lr = lane & 15, lk = lane >> 4;
  const int ti = w >> 2, tj = w & 3;
  const size_t col = (size_t)kp * kNB;
  const size_t cbase = (size_t)bi * kNB * ld + (size_t)bj0 * kNB + (size_t)(16 * ti + lk) * ld + 16 * tj + lr;
  const v4f64 z = (v4f64){0.0, 0.0, 0.0, 0.0};
  {
    const double2 a0 = TileLoad2(L + (size_t)bi * kNB * ld + col, ld, tid, 0), a1 = TileLoad2(L + (size_t)bi * kNB * ld + col, ld, tid, 1);
    double2 b0 = make_double2(0.0, 0.0), b1 = b0, e0 = b0, e1 = b0;
    if (valid0) { b0 = TileLoad2(L + (size_t)bj0 * kNB * ld + col, ld, tid, 0); b1 = TileLoad2(L + (size_t)bj0 * kNB * ld + col, ld, tid, 1); }
    if (valid1) { e0 = TileLoad2(L + (size_t)(bj0 + 1) * kNB * ld + col, ld, tid, 0); e1 = TileLoad2(L + (size_t)(bj0 + 1) * kNB * ld + col, ld, tid, 1); }
    TileStore2(At, tid, 0, a0); TileStore2(At, tid, 1, a1);
    if (valid0) { TileStore2(Bt, tid, 0, b0); TileStore2(Bt, tid, 1, b1); }
    if (valid1) { TileStore2(Bt + kNB * kLS, tid, 0, e0); TileStore2(Bt + kNB * kLS, tid, 1, e1); }
  }
  const size_t zoff = (size_t)(16 * ti + lk) * kNB + 16 * tj + lr;
  double* C0 = z0 ? z0 + zoff : S + cbase; double* C1 = z1 ? z1 + zoff : S + cbase + kNB;
  const size_t ld0 = z0 ? (size_t)kNB : (size_t)ld, ld1 = z1 ? (size_t)kNB : (size_t)ld;
  v4f64 c0 = z, c1 = z, p0 = z, p1 = z;
  if (valid0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) c0[i] = fresh0 ? 0.0 : LoadCoherent(C0 + (size_t)(4 * i) * ld0);
  }
  if (valid1) {
#pragma unroll
    for (int i = 0; i < 4; ++i) c1[i] = fresh1 ? 0.0 : LoadCoherent(C1 + (size_t)(4 * i) * ld1);
  }
  __syncthreads();
  const double* ar = At + (16 * ti + lr) * kLS + lk;
  const double* br = Bt + (16 * tj + lr) * kLS + lk;
  if (valid0 && valid1) {
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      const double av = ar[4 * kk];
      p0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, br[4 * kk], p0, 0, 0, 0);
      p1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, br[kNB * kLS + 4 * kk], p1, 0, 0, 0);
    }
  } else {      // one tile: its operand is in the slot of the tile that is updated (see the loads above)
    const double* b1 = br + (valid1 ? kNB * kLS : 0);
    v4f64 p = z;
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) p = __builtin_amdgcn_mfma_f64_16x16x4f64(ar[4 * kk], b1[4 * kk], p, 0, 0, 0);
    if (valid1) p1 = p; else p0 = p;
  }
  if (valid0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) __hip_atomic_store(C0 + (size_t)(4 * i) * ld0, c0[i] - p0[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (valid1) {
#pragma unroll
    for (int i = 0; i < 4; ++i) __hip_atomic_store(C1 + (size_t)(4 * i) * ld1, c1[i] - p1[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// Solve task of tile (i,k), i >= k+3.  Nothing of it waits for a counter the chain moves: the solved tile (k,k-1) comes out of
// chain(k-1)'s mailbox (in there ~5 us into that step) and M_k out of its mailbox (stored at the end of that step), so the pending
// panel k-1 update runs during chain(k-1) and the solve starts one memory round trip after M_k exists.
__device__ __forceinline__ bool SolveTask(double* S, double* L, int ld, int k, int i, Mailboxes mb, int32_t* __restrict__ flag, int* s_failed,
                                          double* BX, double* Mk, double* B1, double* B2, bool prev_nz, bool first) {      // prev_nz: tile (i,k-1) is structurally non-zero; first: k starts a chain
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, lr = lane & 15, g = lane >> 4;
  const size_t pbase = (size_t)i * kNB * ld + (size_t)k * kNB;
  const double2 x0 = TileLoad2T<true>(S + pbase, ld, tid, 0), x1 = TileLoad2T<true>(S + pbase, ld, tid, 1);
  if (!first) {
    const double2 zz = make_double2(0.0, 0.0);      // (a structurally zero tile (i,k-1) was never solved: an exact zero tile, the pending update adds nothing)
    const double2 a0 = prev_nz ? TileLoad2(L + pbase - kNB, ld, tid, 0) : zz, a1 = prev_nz ? TileLoad2(L + pbase - kNB, ld, tid, 1) : zz;
    auto deposit = [&]() { TileStore2(BX, tid, 0, x0); TileStore2(BX, tid, 1, x1); TileStore2(B1, tid, 0, a0); TileStore2(B1, tid, 1, a1); };
    if (!FetchMailTile<false>(B2, mb.xsol + (size_t)(k - 1) * kNB * kNB, tid, flag, s_failed, deposit)) return false;
    UpdateTileInPlace(BX, B1, B2, w >> 2, w & 3, lr, g);
    __syncthreads();
    // (the tiles of rows k+3, k+4 are what PrepX(k+1) / PrepX(k+2) and the front updates wait for: their workgroups poll M_k with every thread - the tile is in
    // LDS one round trip after it becomes visible, not two)
    if (i <= k + 4) { if (!FetchMailTile<true>(Mk, mb.Minv + (size_t)k * kNB * kNB, tid, flag, s_failed)) return false; }
    else if (!FetchMailTile(Mk, mb.Minv + (size_t)k * kNB * kNB, tid, flag, s_failed)) return false;
  } else {
    TileStore2(BX, tid, 0, x0); TileStore2(BX, tid, 1, x1);
    LoadTile(Mk, mb.Minv + (size_t)k * kNB * kNB, kNB, tid);      // k_potrf64's, from the previous launch
    __syncthreads();
  }
  const int s = w & 3, ct = w >> 2;
  const v4f64 x = SolveTile(BX, Mk, s, ct, lr, g);
#pragma unroll
  for (int r = 0; r < 4; ++r) StoreThrough(L + pbase + (size_t)(16 * s + g + 4 * r) * ld + 16 * ct + lr, x[r]);
  return true;
}

// nz (may be null = dense): T x T bytes, the structurally non-zero tiles of the factor (closed under fill-in, the two sub-diagonals the chain and the
// prep tasks own included): tasks only exist for those, and a task skips operands that are not (they were never solved)
__global__ __launch_bounds__(kPanelThreads) void k_cholesky_tasks(double* S, double* L, int ld, int T, Mailboxes mb, int32_t* __restrict__ flag,
                                                                  int32_t* __restrict__ ctr, const ChainTask* __restrict__ tasks, const uint8_t* __restrict__ nz, ChainRanges cr, double* Z) {      // Z: the scratch tile pool of several chains
  __shared__ __attribute__((aligned(16))) double smem[4 * kNB * kLS];
  __shared__ double inv_diag[kNB];
  __shared__ int s_failed;      // sticky: a wait of this workgroup ran into its bound
  const int b = blockIdx.x;
  if (threadIdx.x == 0) s_failed = 0;
  if (b < cr.n) {
#ifdef PP_CHOL_TRACE
    if (threadIdx.x == 0) {
      unsigned id; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
      unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      g_burn_hwid[0] = (id & 0xffff) | (xcc << 16);
    }
#endif
    int kb = 0, ke = T, post = 0;      // (selects, not cr.begin[b]: a dynamically indexed kernel argument is copied to scratch memory)
#pragma unroll
    for (int c = 0; c < kMaxChains; ++c) if (c == b) { kb = cr.begin[c]; ke = cr.end[c]; post = cr.post[c]; }
    ChainLoop(S, L, ld, T, mb, flag, ctr, smem, inv_diag, &s_failed, kb, ke, post, cSol0 + b * kMaxSteps);
#ifdef PP_CHOL_TRACE
    if (threadIdx.x == 0) __hip_atomic_store(&g_burn_stop, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
    return;
  }
#ifdef PP_CHOL_TRACE
  if (tasks == nullptr) {
    // contention experiment (tools/chol_task_trace.hip ... iso N): workgroups that keep their CUs busy with ~30 KB of LDS-only code
    // (the block factorisation, on garbage) while the chain runs alone - no memory traffic, no dependence on the chain
    if (threadIdx.x == 0) {
      unsigned id; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
      unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      if (b < 512) g_burn_hwid[b] = (id & 0xffff) | (xcc << 16);
    }
    __shared__ int burn_flag[4];
    for (int i = threadIdx.x; i < 4 * kNB * kLS; i += kPanelThreads) smem[i] = (i % 67 == 0) ? 64.0 : 0.001 * (i % 13);
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    while (__hip_atomic_load(&g_burn_stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
      PotrfPanels(smem + 2 * kNB * kLS, smem, inv_diag, (int32_t*)burn_flag, lane, w, NoSideJob(), NoSideJob());
      __syncthreads();
    }
    return;
  }
#endif
  const ChainTask t = tasks[b - cr.n];
  const int k = t.k;
  const bool first = (t.flags & 1) != 0;
  const int solbase = cSol0 + ((t.flags >> 4) & 15) * kMaxSteps;
  double* B0 = smem; double* B1 = smem + kNB * kLS; double* B2 = smem + 2 * kNB * kLS; double* B3 = smem + 3 * kNB * kLS;
  auto tile_nz = [&](int r, int c) { return !nz || nz[(size_t)r * T + c] != 0; };
  if (t.type == kTaskPrepX || t.type == kTaskPrepD) {
    const bool far_nz = !first && tile_nz(k + 2, k - 1);
    if (t.type == kTaskPrepX) PrepTask<true>(S, L, ld, k, mb, flag, ctr, &s_failed, B0, B1, B2, B3, t.w0, t.w1, t.w2, far_nz, first, t.a, t.b, solbase);
    else PrepTask<false>(S, L, ld, k, mb, flag, ctr, &s_failed, B0, B1, B2, B3, t.w0, t.w1, t.w2, far_nz, first, t.a, t.b, solbase);
    return;
  }
  if (t.type == kTaskPairPrep) {
    // P and Z of pair t.a for the paired back substitution: the solved tiles (b,a), (a+2,{a,b}), (a+3,{a,b}) and the inverses M_a, M_b
    const int gp = t.a, a = 2 * gp, b = a + 1;
    const bool has_z = gp + 1 < BacksubNumPairs(T);
    WaitList wl;
    wl.p0 = ctr + cSol0 + b; wl.n0 = a + 1;
    if (has_z) { wl.p1 = ctr + cSol0 + (a + 2); wl.n1 = a + 2; wl.p2 = ctr + cSol0 + (a + 3); wl.n2 = a + 2; }
    if (!TaskWait(wl, flag, &s_failed)) return;
    int* sf = &s_failed;
    auto fetch_m = [&](double* dst, int kk) { return FetchMailTile<false>(dst, mb.Minv + (size_t)kk * kNB * kNB, (int)threadIdx.x, flag, sf); };
    (void)PairPrepBody(gp, t.b, T, L, ld, mb.Pw, mb.Zw, B0, B1, B2, B3, fetch_m);
    return;
  }
  if (t.type == kTaskSolve) {
    // tile (i,k), i >= k+3: M_k (chain(k-1)), the solved tiles (k,k-1) and (i,k-1), the panels <= k-2 applied to (i,k)
    const int i = t.a;
    const bool prev_nz = !first && tile_nz(i, k - 1);
    WaitList wl;
    wl.p1 = ctr + solbase + i; wl.n1 = t.w2;      // (the row's previous non-zero column of this chain: k if tile (i,k-1) is one)
    wl.p3 = VerCounter(ctr, i >> 1, k >> 1); wl.n3 = t.w0;
    if (!TaskWait(wl, flag, &s_failed)) return;
    PP_TASK_MIN(7, k);
    if (!SolveTask(S, L, ld, k, i, mb, flag, &s_failed, B0, B1, B2, B3, prev_nz, first)) return;
    TaskStoresDone();
    if (threadIdx.x == 0) __hip_atomic_store(ctr + solbase + i, t.w1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    PP_TASK_MAX(9, k);
    return;
  }
  if (t.type == kTaskMerge) {
    // what another chain has accumulated for super-tile (I,J) in its scratch array is added to the tiles themselves: one step of the super-tile's own sequence
    const int I = t.a, J = t.b & 255;
    WaitList wl;
    wl.p0 = ctr + t.cidx; wl.n0 = t.w0;
    wl.p1 = ctr + t.sidx; wl.n1 = t.w2;
    if (!TaskWait(wl, flag, &s_failed)) return;
#pragma unroll      // (constant indices into t.slot: a dynamically indexed member would put the whole task record into scratch memory)
    for (int q = 0; q < 4; ++q) {
      if (!((t.mask >> q) & 1)) continue;
      const size_t base = (size_t)(2 * I + (q >> 1)) * kNB * ld + (size_t)(2 * J + (q & 1)) * kNB;
      const double* Zt = Z + (size_t)t.slot[q] * kNB * kNB;
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int idx = (int)threadIdx.x + kPanelThreads * it, r = idx >> 6, c = idx & 63;
        const size_t o = base + (size_t)r * ld + c;
        __hip_atomic_store(S + o, LoadCoherent(S + o) + LoadCoherent(Zt + r * kNB + c), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    TaskStoresDone();
    if (threadIdx.x == 0) __hip_atomic_store(ctr + t.cidx, t.w1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  {
    // super-tile (I,J) by panel k-1: column k-1 of its block rows solved, the panels <= k-2 applied to it
    const int I = t.a, J = t.b & 255, part = (t.b >> 8) & 15, parts = (t.b >> 12) & 15, target = t.b >> 16;
    const bool two = parts == kPartsTwoPanels;      // the whole super-tile by panels k-1 AND k: column k solved as well, ver moves by two
    WaitList wl;
    wl.p0 = ctr + t.cidx; wl.n0 = t.w0;
    auto row_slot = [&](int row, bool distinct, const int32_t** p, int* n) {      // column k-1 (and k) of a block row this task reads
      const bool used = distinct && row < T && row >= k + 1 && tile_nz(row, k - 1);
      *p = ctr + solbase + (used ? row : 0); *n = used ? (two ? t.w2 + 1 : t.w2) : 0;
    };
    const int bi = 2 * I + (parts == 2 ? part : part >> 1), bj0 = 2 * J + (parts == 2 ? 0 : part & 1), nb = parts == 2 ? 2 : 1;
    if (parts == 1 || two) {
      row_slot(2 * I, true, &wl.p1, &wl.n1); row_slot(2 * I + 1, true, &wl.p2, &wl.n2);
      row_slot(2 * J, J != I, &wl.p3, &wl.n3); row_slot(2 * J + 1, J != I, &wl.p4, &wl.n4);
    } else {
      row_slot(bi, true, &wl.p1, &wl.n1);
      row_slot(bj0, bj0 != bi, &wl.p3, &wl.n3); row_slot(bj0 + 1, nb == 2 && bj0 + 1 != bi, &wl.p4, &wl.n4);
    }
#ifdef PP_CHOL_TRACE
    const bool front = (I == (k + 3) >> 1) && (J == (k + 1) >> 1 || J == (k + 3) >> 1);      // the super-tiles PrepX(k+1) / PrepD(k+1) wait for
    const int fs = J == (k + 1) >> 1 ? 16 : 21;
    if (front) PP_TASK_MAX(fs, k);
    unsigned long long* wait_trace = (front && fs == 16 && part == 0 && k < 128) ? &g_wait_missing[k] : nullptr;
    if (!TaskWait(wl, flag, &s_failed, wait_trace)) return;
    if (false)
#endif
    if (!TaskWait(wl, flag, &s_failed)) return;
    PP_TASK_MIN(10, k);
#ifdef PP_CHOL_TRACE
    if (front) PP_TASK_MAX(fs + 1, k);
#endif
    auto valid = [&](int r, int c) {
      const bool own = (r == k + 1 && c == k + 1) || (r == k + 2 && (c == k + 1 || c == k + 2));      // the chain's / prep's three tiles
      return r < T && c < T && r >= c && c >= k + 1 && !own && tile_nz(r, k - 1) && tile_nz(c, k - 1);
    };
    const bool v0 = valid(bi, bj0), v1 = nb == 2 && valid(bi, bj0 + 1);
    const int q0 = 2 * (bi - 2 * I) + (bj0 - 2 * J);
    const int ts0 = t.slot[0], ts1 = t.slot[1], ts2 = t.slot[2], ts3 = t.slot[3];
    auto slot_of = [=](int q) { return q == 0 ? ts0 : (q == 1 ? ts1 : (q == 2 ? ts2 : ts3)); };      // (no dynamic index into the task record: it would live in scratch memory)
    const bool zs = t.zsel >= 0;      // (another chain's super-tile: into this chain's scratch tiles)
    if (parts == 1) UpdateSuperTile<false>(S, L, ld, k - 1, T, I, J, B0, B2, nz, t.mask, zs ? Z : nullptr, tasks[b - cr.n].slot);
    else if (two) UpdateSuperTile<true>(S, L, ld, k - 1, T, I, J, B0, B2, nullptr);      // (two panels per task: dense systems only)
    else if (v0 || v1) UpdateTilesTask(S, L, ld, k - 1, bi, bj0, v0, v1, B0, B2, (t.mask >> q0) & 1, (t.mask >> (q0 + 1)) & 1,
                                       zs && v0 ? Z + (size_t)slot_of(q0) * kNB * kNB : nullptr, zs && v1 ? Z + (size_t)slot_of(q0 + 1) * kNB * kNB : nullptr);
    TaskStoresDone();
    if (threadIdx.x == 0 && __hip_atomic_fetch_add(ctr + t.sidx, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1 == target)
      __hip_atomic_store(ctr + t.cidx, t.w1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    PP_TASK_MAX(8, k);
#ifdef PP_CHOL_TRACE
    if (front) PP_TASK_MAX(fs + 2, k);
#endif
  }
}

// Back substitution L^T x = y (y = row rhs_row of the factor) in ONE launch: workgroup j owns the 64 unknowns of
// block j, applies  y_j -= L[k-block, j-block]^T x_k  for k = T-1 .. j+1 as the x_k arrive, then solves its block
// with the precomputed L_jj^-1 and publishes x_j.  The 47 dependent launches of a per-block kernel cost ~5.7 us
// each (launch boundary + two dependent global round trips); here a step of the chain is one 8-byte-granule
// hand-off (x values are their own ready flags: the buffer is preset to an all-ones NaN pattern and written with
// write-through agent-scope stores, read with agent-scope loads — MI355X_MICROARCH.md, inter-workgroup visibility).
// A workgroup only ever waits on HIGHER block indices, which are given the lower blockIdx (dispatched first),
// so the wait cannot deadlock even if not all workgroups are resident; every spin is bounded.
constexpr unsigned long long kNotReady = 0xFFFFFFFFFFFFFFFFull;
// nz (may be null): T x T bytes, nz[k T + j] = tile (k,j) of the factor is structurally non-zero - the others are skipped (a
// block-sparse factor: block j only waits for the x_k it is coupled to)
__global__ __launch_bounds__(256) void k_backsub_all(const double* __restrict__ S, int ld, int T, int rhs_row, const double* __restrict__ Linv,
                                                     double* x_out, int32_t* __restrict__ flag, const uint8_t* __restrict__ nz) {
  __shared__ double part[4][kNB];
  __shared__ double ys[kNB];
  const int tid = threadIdx.x, c = tid & 63, q = tid >> 6;
  const int j = T - 1 - (int)blockIdx.x;
  // this thread's slice of L_jj^-1: rows 16q .. 16q+15, column c
  double linv[16];
  {
    const double* Lb = Linv + (size_t)j * kNB * kNB;
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) linv[rr] = Lb[(16 * q + rr) * kNB + c];
  }
  double acc = 0.0;
  bool dead = false;    // a lane whose wait timed out stops waiting: the solve is reported invalid instead of hanging
  for (int k = T - 1; k > j; --k) {
    if (nz && !nz[(size_t)k * T + j]) continue;
    double lt[16];
    const double* Lt = S + ((size_t)k * kNB + 16 * q) * ld + (size_t)j * kNB + c;
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) lt[rr] = Lt[(size_t)rr * ld];
    // lanes 0..15 of every wavefront poll the 16 values of x_k this wavefront needs
    double xv = 0.0;
    if (c < 16) {
      unsigned long long* src = reinterpret_cast<unsigned long long*>(x_out + (size_t)k * kNB + 16 * q + c);
      unsigned long long bits = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      int spins = 0;
      while (bits == kNotReady && !dead && spins < (1 << 18)) {
        __builtin_amdgcn_s_sleep(1);
        bits = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ++spins;
      }
      if (bits == kNotReady) { if (!dead) atomicOr(flag, 4); dead = true; bits = 0ull; }
      xv = __longlong_as_double((long long)bits);
    }
    double p0 = 0.0, p1 = 0.0, p2 = 0.0, p3 = 0.0;      // four short chains instead of one 16-deep one (this is the hop's critical path)
#pragma unroll
    for (int rr = 0; rr < 16; rr += 4) {
      p0 = fma(lt[rr], ReadLane(xv, rr), p0); p1 = fma(lt[rr + 1], ReadLane(xv, rr + 1), p1);
      p2 = fma(lt[rr + 2], ReadLane(xv, rr + 2), p2); p3 = fma(lt[rr + 3], ReadLane(xv, rr + 3), p3);
    }
    acc += (p0 + p1) + (p2 + p3);
  }
  part[q][c] = acc;
  __syncthreads();
  if (tid < kNB) {
    const int col = j * kNB + tid;
    const double y = (col < rhs_row) ? S[(size_t)rhs_row * ld + col] : 0.0;   // padding / the rhs row's own diagonal carry no unknown
    ys[tid] = y - ((part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid]));
  }
  __syncthreads();
  {  // x[c] = sum_r Linv[r][c] * y[r]  (L^-T y), 4 partial sums over r
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
    for (int rr = 0; rr < 16; rr += 4) {
      s0 = fma(linv[rr], ys[16 * q + rr], s0); s1 = fma(linv[rr + 1], ys[16 * q + rr + 1], s1);
      s2 = fma(linv[rr + 2], ys[16 * q + rr + 2], s2); s3 = fma(linv[rr + 3], ys[16 * q + rr + 3], s3);
    }
    part[q][c] = (s0 + s1) + (s2 + s3);
  }
  __syncthreads();
  if (tid < kNB) {
    const double v = (part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid]);
    unsigned long long bits = (unsigned long long)__double_as_longlong(v);
    if (bits == kNotReady) bits = 0x7FF8000000000000ull;    // a NaN result stays a NaN, never the not-ready pattern
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(x_out + (size_t)j * kNB + tid), bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// ---- paired back substitution (dense systems) --------------------------------------------------------------------------------
// k_backsub_all pays one memory hand-off (~1.0 us: the store's visibility + a poll round trip) and two dependent mat-vec stages per
// 64-block: 47 x 1.3 us = 62 us.  Here two consecutive blocks a = 2g, b = 2g + 1 form a PAIR solved by one workgroup in ONE stage
// after its newest inputs arrive:
//     x_ab = G^T u  -  Z^T x_new,        G = [M_a 0; P M_b] = the inverse of the pair's 128 x 128 diagonal factor, P = -M_b L_ba M_a,
//                                        u = y_ab - sum over the blocks BEYOND the next pair of L_k,ab^T x_k      (known a hop earlier),
//                                        Z = L_new,ab G   (128 x 128; new = the pair above, blocks 2g+2, 2g+3)
// so the critical path of a hop is the hand-off + one 128 x 128 mat-vec, and there are half as many hops: 22 pairs + the three or four
// top blocks (solved singly, as in k_backsub_all: their inverses are the last thing the factorisation produces) = ~33 us.
// P and Z are 8 products of 64 x 64 tiles per pair, computed by k_backsub_prepare in one launch between the factorisation and the
// solve (three workgroups per pair, ~4 us): the same kernels in both launch structures, so their solutions stay bitwise equal.
// Block-sparse systems keep k_backsub_all (it skips the structurally zero tiles).
constexpr int kPairThreads = 1024;
__device__ __forceinline__ double PollReady(const double* p, int32_t* flag, bool* dead) {
  unsigned long long* src = const_cast<unsigned long long*>(reinterpret_cast<const unsigned long long*>(p));
  unsigned long long bits = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  int spins = 0;
  while (bits == kNotReady && !*dead && spins < (1 << 18)) {
    __builtin_amdgcn_s_sleep(1);
    bits = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ++spins;
  }
  if (bits == kNotReady) { if (!*dead) atomicOr(flag, 4); *dead = true; bits = 0ull; }
  return __longlong_as_double((long long)bits);
}
// two values per lane with BOTH requests in flight before either is looked at (two PollReady calls in a row are two round trips)
__device__ __forceinline__ void PollReady2(const double* pa, const double* pb, int32_t* flag, bool* dead, double* va, double* vb) {
  unsigned long long* sa = const_cast<unsigned long long*>(reinterpret_cast<const unsigned long long*>(pa));
  unsigned long long* sb = const_cast<unsigned long long*>(reinterpret_cast<const unsigned long long*>(pb));
  unsigned long long a = __hip_atomic_load(sa, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), b = __hip_atomic_load(sb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  int spins = 0;
  while ((a == kNotReady || b == kNotReady) && !*dead && spins < (1 << 18)) {
    __builtin_amdgcn_s_sleep(1);
    a = __hip_atomic_load(sa, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); b = __hip_atomic_load(sb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ++spins;
  }
  if (a == kNotReady || b == kNotReady) { if (!*dead) atomicOr(flag, 4); *dead = true; a = 0ull; b = 0ull; }
  *va = __longlong_as_double((long long)a); *vb = __longlong_as_double((long long)b);
}
__device__ __forceinline__ void PublishX(double* p, double v) {
  unsigned long long bits = (unsigned long long)__double_as_longlong(v);
  if (bits == kNotReady) bits = 0x7FF8000000000000ull;    // a NaN result stays a NaN, never the not-ready pattern
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// grid = nsingles + npairs: the top blocks first (T-1 downwards), then the pairs from the top one down - a workgroup only waits for
// workgroups with a lower blockIdx.  Thread (c, q): c = tid & 127 the unknown inside the pair (block a: 0..63, block b: 64..127),
// q = tid >> 7 = 0..7 the slice of rows it sums over.
__global__ __launch_bounds__(kPairThreads) void k_backsub_pairs(const double* __restrict__ S, int ld, int T, int rhs_row, const double* __restrict__ Linv,
                                                                const double* __restrict__ Pw, const double* __restrict__ Zw, double* x_out,
                                                                int32_t* __restrict__ flag) {
  __shared__ double part[8][2 * kNB];
  __shared__ double us[2 * kNB];
  const int tid = threadIdx.x, lane = tid & 63, c = tid & 127, q = tid >> 7;
  const int npairs = BacksubNumPairs(T), nsingles = T - 2 * npairs;
  const size_t tile = (size_t)kNB * kNB;
  bool dead = false;
  if ((int)blockIdx.x < nsingles) {
    // ---- one of the top blocks, as k_backsub_all (threads (c < 64, q): 8 rows each)
    const int j = T - 1 - (int)blockIdx.x;
    const bool act = c < kNB;
    PP_BS_STAMP(0, j);
    double linv[8];
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) linv[rr] = act ? Linv[(size_t)j * tile + (size_t)(8 * q + rr) * kNB + c] : 0.0;
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) asm volatile("" : "+v"(linv[rr]));      // (in registers before the waiting starts, see the pairs below)
    double acc = 0.0;
    for (int k = T - 1; k > j; --k) {
      double lt[8];
#pragma unroll
      for (int rr = 0; rr < 8; ++rr) lt[rr] = act ? S[((size_t)k * kNB + 8 * q + rr) * ld + (size_t)j * kNB + c] : 0.0;
      const double xv = (lane < 8 && act) ? PollReady(x_out + (size_t)k * kNB + 8 * q + lane, flag, &dead) : 0.0;      // (`act` is wave-uniform: the idle half does not poll)
      double p0 = 0.0, p1 = 0.0;
#pragma unroll
      for (int rr = 0; rr < 8; rr += 2) { p0 = fma(lt[rr], ReadLane(xv, rr), p0); p1 = fma(lt[rr + 1], ReadLane(xv, rr + 1), p1); }
      acc += p0 + p1;
    }
    part[q][c] = acc;
    __syncthreads();
    if (tid < kNB) {
      const int col = j * kNB + tid;
      const double y = (col < rhs_row) ? S[(size_t)rhs_row * ld + col] : 0.0;
      double sum = 0.0;
#pragma unroll
      for (int qq = 0; qq < 8; ++qq) sum += part[qq][tid];
      us[tid] = y - sum;
    }
    __syncthreads();
    {
      double s0 = 0.0, s1 = 0.0;
#pragma unroll
      for (int rr = 0; rr < 8; rr += 2) { s0 = fma(linv[rr], us[8 * q + rr], s0); s1 = fma(linv[rr + 1], us[8 * q + rr + 1], s1); }
      part[q][c] = s0 + s1;
    }
    __syncthreads();
    if (tid < kNB) {
      double v = 0.0;
#pragma unroll
      for (int qq = 0; qq < 8; ++qq) v += part[qq][tid];
      PublishX(x_out + (size_t)j * kNB + tid, v);
    }
    PP_BS_STAMP(3, j);
    return;
  }
  // ---- a pair
  const int gp = npairs - 1 - ((int)blockIdx.x - nsingles);
  const int a = 2 * gp;
  const bool has_z = gp + 1 < npairs;
  const int first_far = has_z ? a + 4 : a + 2;      // blocks >= first_far enter through their tiles of L; the pair above through Z
  PP_BS_STAMP(0, a);
  // G = [M_a 0; P M_b] goes to LDS (96 KB; the workgroup has the CU to itself anyway), this thread's slice of Z (rows 16q .. 16q+15 of
  // 128, column c) to registers: Z is what the hop's critical path multiplies with, and it must be THERE before the waiting starts -
  // left to itself the compiler sinks the loads to their first use, behind the poll of the newest input (2.7 us per hop).  With G's
  // slices in registers as well the kernel spilled (128 VGPRs at 1024 threads): every use of a spilled slice was a scratch load.
  __shared__ double Gs[3 * kNB * kNB];      // M_a | P | M_b, row-major 64 x 64 each
  double z[16];
  double y_rhs = 0.0;      // this thread's entry of the right-hand side (threads 0..127), fetched now: it is needed on the two-hop cycle below
  if (tid < 2 * kNB) { const int col = a * kNB + tid; y_rhs = (col < rhs_row) ? S[(size_t)rhs_row * ld + col] : 0.0; }
  {
    const double* Ma = Linv + (size_t)a * tile;
    const double* P = Pw + (size_t)gp * tile;
    for (int i = tid; i < (int)tile; i += kPairThreads) { Gs[i] = Ma[i]; Gs[tile + i] = P[i]; Gs[2 * tile + i] = Ma[tile + i]; }
    if (has_z) {
      const double* Z = Zw + (size_t)gp * 4 * tile;
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) z[rr] = Z[(size_t)(16 * q + rr) * 2 * kNB + c];
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) asm volatile("" : "+v"(z[rr]));
    }
    asm volatile("" : "+v"(y_rhs));
  }
  // Far terms, one ARRIVAL at a time: the blocks above arrive singly while they are the top singles and two at a time below (a pair
  // publishes both its blocks at once).  Both blocks of an arrival are polled together (lanes 0..7 / 8..15: ONE memory round trip) and
  // the tile slices of the next arrival are requested before this one's are used.  This loop and the G^T u stage after it are on a
  // two-hop cycle (x of pair g+2 -> far terms and G^T u of pair g -> ready for x of pair g+1): block by block with a poll round trip and a
  // tile fetch per block, and the right-hand side fetched only when needed, the cycle was 5.8 us and a hop 2.9 us (tools/chol_task_trace.hip, PP_BS_TRACE).
  double acc = 0.0;
  auto load_slice = [&](int k, double (&lt)[8]) {
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) lt[rr] = S[((size_t)k * kNB + 8 * q + rr) * ld + (size_t)a * kNB + c];
  };
  auto use_slice = [&](const double (&lt)[8], const double* xb) {
    double p0 = 0.0, p1 = 0.0;
#pragma unroll
    for (int rr = 0; rr < 8; rr += 2) { p0 = fma(lt[rr], xb[8 * q + rr], p0); p1 = fma(lt[rr + 1], xb[8 * q + rr + 1], p1); }
    acc += p0 + p1;
  };
  // Only wavefront 0 polls (lane = unknown; two values per lane for a two-block arrival) and hands the values over through LDS: with
  // every wavefront of every waiting workgroup polling its own slice (22 x 16 wavefronts on the same few lines, each poll a transaction
  // below the L2s) the slowest of a workgroup's sixteen polls came back 1-2 us after the fastest, and the barrier waits for the slowest.
  __shared__ double xs[2][2 * kNB];      // double-buffered: the next arrival may be written while slow wavefronts still read this one
  {
    const int top_pairs_block = 2 * npairs - 1;      // highest block that belongs to a pair
    int k = T - 1, buf = 0;
    while (k >= first_far) {
      const int n = k > top_pairs_block ? 1 : 2;
      double cur0[8], cur1[8];      // the tile slices travel with the poll (both a memory round trip)
      load_slice(k, cur0);
      if (n == 2) load_slice(k - 1, cur1);
      if (tid < kNB) {
        if (n == 2) PollReady2(x_out + (size_t)k * kNB + tid, x_out + (size_t)(k - 1) * kNB + tid, flag, &dead, &xs[buf][tid], &xs[buf][kNB + tid]);
        else xs[buf][tid] = PollReady(x_out + (size_t)k * kNB + tid, flag, &dead);
      }
      __syncthreads();
      use_slice(cur0, xs[buf]);
      if (n == 2) use_slice(cur1, xs[buf] + kNB);
      k -= n; buf ^= 1;
    }
  }
  PP_BS_STAMP(1, a);
  part[q][c] = acc;
  __syncthreads();
  if (tid < 2 * kNB) {
    double sum = 0.0;
#pragma unroll
    for (int qq = 0; qq < 8; ++qq) sum += part[qq][tid];
    us[tid] = y_rhs - sum;
  }
  __syncthreads();
  {      // G^T u:  x_a = M_a^T u_a + P^T u_b,  x_b = M_b^T u_b  (this thread: rows 8q .. 8q+7)
    double s0 = 0.0, s1 = 0.0;
    if (c < kNB) {
#pragma unroll
      for (int rr = 0; rr < 8; ++rr) { const int r = 8 * q + rr; s0 = fma(Gs[r * kNB + c], us[r], s0); s1 = fma(Gs[tile + r * kNB + c], us[kNB + r], s1); }
    } else {
#pragma unroll
      for (int rr = 0; rr < 8; ++rr) { const int r = 8 * q + rr; s1 = fma(Gs[2 * tile + r * kNB + (c - kNB)], us[kNB + r], s1); }
    }
    part[q][c] = s0 + s1;
  }
  __syncthreads();
  double v = 0.0;
  if (tid < 2 * kNB) {
#pragma unroll
    for (int qq = 0; qq < 8; ++qq) v += part[qq][tid];
  }
  if (has_z) {
    PP_BS_STAMP(4, a);
    if (tid < kNB) {      // the pair above: wavefront 0 polls its 128 values
      PollReady2(x_out + (size_t)(a + 2) * kNB + tid, x_out + (size_t)(a + 3) * kNB + tid, flag, &dead, &xs[0][tid], &xs[0][kNB + tid]);
    }
    __syncthreads();      // (also: part is reused below)
    PP_BS_STAMP(2, a);
    double p0 = 0.0, p1 = 0.0, p2 = 0.0, p3 = 0.0;
#pragma unroll
    for (int rr = 0; rr < 16; rr += 4) {
      p0 = fma(z[rr], xs[0][16 * q + rr], p0); p1 = fma(z[rr + 1], xs[0][16 * q + rr + 1], p1);
      p2 = fma(z[rr + 2], xs[0][16 * q + rr + 2], p2); p3 = fma(z[rr + 3], xs[0][16 * q + rr + 3], p3);
    }
    part[q][c] = (p0 + p1) + (p2 + p3);
    __syncthreads();
    if (tid < 2 * kNB) {
      double sum = 0.0;
#pragma unroll
      for (int qq = 0; qq < 8; ++qq) sum += part[qq][tid];
      v -= sum;
    }
  }
  if (tid < 2 * kNB) PublishX(x_out + (size_t)a * kNB + tid, v);
  PP_BS_STAMP(3, a);
}

// (Four consecutive blocks per workgroup - 12 hand-offs between workgroups instead of 47, the hops inside a group through LDS - was
// measured at 98 us against 62 us: the x_k of the group above arrive as a burst, and the 4 x 4 tiles they multiply (512 KB) have no
// place on the CU to wait in, so their loads queue up behind each other on the critical path; see DESIGN.md.)
// Symbolic Cholesky on the tile graph: eliminating block column k couples every pair of rows that have a non-zero tile in it.
int SymbolicTileFill(int T, uint8_t* nz) {
  for (int i = 0; i < T; ++i) nz[(size_t)i * T + i] = 1;
  std::vector<int> rows;
  for (int k = 0; k < T; ++k) {
    rows.clear();
    for (int i = k + 1; i < T; ++i) if (nz[(size_t)i * T + k]) rows.push_back(i);
    for (size_t a = 0; a < rows.size(); ++a)
      for (size_t b = 0; b <= a; ++b) nz[(size_t)rows[a] * T + rows[b]] = 1;
  }
  int count = 0;
  for (int i = 0; i < T; ++i) for (int j = 0; j <= i; ++j) count += nz[(size_t)i * T + j] ? 1 : 0;
  return count;
}

// Block-sparse structure: tile_nz (T x T, lower triangle, row-major; the caller has already closed it under the fill-in of
// the factorisation) -> per launch k the rows of the solve workgroups and the super-tiles of the update workgroups.
// Layout of aux->sparse_host: [T+1 offsets of the row lists | T+1 offsets of the super-tile lists | the lists]; the same
// array on the device, plus the T x T byte map for the back substitution.
static int EnsureSparseLists(CholeskyAux* aux, int T, hipStream_t strm) {
  if (!aux->tile_nz || aux->tile_T != T) return PP_OK;
  if (aux->sparse_lists && aux->sparse_T == T) return PP_OK;
  if (aux->sparse_lists) { (void)hipFree(aux->sparse_lists); aux->sparse_lists = nullptr; }
  if (aux->sparse_nz) { (void)hipFree(aux->sparse_nz); aux->sparse_nz = nullptr; }
  const uint8_t* nz = aux->tile_nz;
  auto has = [&](int i, int j) { return i < T && j < T && nz[(size_t)i * T + j] != 0; };
  std::vector<int32_t> rows, sups, row_off(T + 1, 0), sup_off(T + 1, 0);
  for (int k = 0; k + 1 < T; ++k) {
    row_off[k] = (int32_t)rows.size(); sup_off[k] = (int32_t)sups.size();
    for (int i = k + 3; i < T; ++i) if (has(i, k)) rows.push_back(i);
    if (k >= 1) {
      const int kp = k - 1, k1 = kp + 2, nb = T - k1, ns = (nb + 1) / 2, nsup = ns * (ns + 1) / 2 - 1;
      for (int u = 0; u < nsup; ++u) {
        int I = (int)((std::sqrt(8.0 * (u + 1) + 1.0) - 1.0) * 0.5);      // TriIndex(u + 1)
        while ((I + 1) * (I + 2) / 2 <= u + 1) ++I;
        while (I * (I + 1) / 2 > u + 1) --I;
        const int J = u + 1 - I * (I + 1) / 2;
        bool any = false;
        for (int q = 0; q < 4; ++q) {
          const int bi = k1 + 2 * I + (q >> 1), bj = k1 + 2 * J + (q & 1);
          any = any || (bi < T && bj < T && bi >= bj && has(bi, kp) && has(bj, kp));
        }
        if (any) sups.push_back(u);
      }
    }
  }
  for (int k = T - 1; k <= T; ++k) { row_off[k] = (int32_t)rows.size(); sup_off[k] = (int32_t)sups.size(); }
  aux->sparse_host.clear();
  aux->sparse_host.insert(aux->sparse_host.end(), row_off.begin(), row_off.end());
  aux->sparse_host.insert(aux->sparse_host.end(), sup_off.begin(), sup_off.end());
  const int base_rows = (int)aux->sparse_host.size();
  aux->sparse_host.insert(aux->sparse_host.end(), rows.begin(), rows.end());
  const int base_sups = (int)aux->sparse_host.size();
  aux->sparse_host.insert(aux->sparse_host.end(), sups.begin(), sups.end());
  aux->sparse_base_rows = base_rows; aux->sparse_base_sups = base_sups;
  PP_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&aux->sparse_lists), sizeof(int32_t) * std::max<size_t>(aux->sparse_host.size(), 1)));
  PP_HIP_TRY(hipMemcpyAsync(aux->sparse_lists, aux->sparse_host.data(), sizeof(int32_t) * aux->sparse_host.size(), hipMemcpyHostToDevice, strm));
  PP_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&aux->sparse_nz), (size_t)T * T));
  PP_HIP_TRY(hipMemcpyAsync(aux->sparse_nz, nz, (size_t)T * T, hipMemcpyHostToDevice, strm));
  PP_HIP_TRY(hipStreamSynchronize(strm));      // (the caller's stream, not the legacy one: another host thread may be capturing its own factorisation)
  aux->sparse_T = T;
  return PP_OK;
}

// The task list of task mode for T block columns: PrepX / PrepD / solve / update tasks sorted by priority (see above); built once
// per matrix size, outside any stream capture.
constexpr int kTaskAutoMaxT = kMaxSteps;      // (round 2: 88 - equal at n = 6000, slower at 8000; with the two-panel updates of round 3: n = 3000 0.68 against 0.83 ms,
                                              // 4000 1.08 / 1.36, 6000 2.57 / 3.00, 8000 5.53 / 5.96 - tools/chol_time.py)
static bool UseTasks(int mode, int T) { return T >= 4 && T <= kMaxSteps && (mode == 1 || (mode == 2 && T <= kTaskAutoMaxT)); }

// The chains of a tile map (see ChainRanges), the map the one-launch mode works with, and the ORDER in which its block columns are eliminated:
//   map      the caller's (already closed under fill-in) plus, inside every chain, the two sub-diagonals - the tiles the chain and the prep tasks own at
//            every step whether anything couples them or not - closed under fill-in again (a no-op for a band of at least two tiles)
//   time[k]  length of the longest dependency path below block column k (k for one chain): columns of different chains with the same time are
//            eliminated side by side
//   rho1[k]  1 + the rank of k in the order (time, k): the value that says "column k is done" in a counter.  Every counter is moved by tasks that wait
//            for each other in this order, so "counter >= rho1[k]" means k's contribution and every earlier one are in (k + 1 for one chain).
struct ChainPlan {
  ChainRanges cr;
  std::vector<uint8_t> map;      // empty: dense
  std::vector<int> time, rho1, chain_of;
};
static ChainPlan PlanChains(int T, const uint8_t* nz, int max_chains = kMaxChains) {
  ChainPlan p;
  std::memset(&p.cr, 0, sizeof(p.cr));
  std::vector<int> starts{0};
  if (const char* e = getenv("PPSFM_CHOL_CHAINS")) max_chains = std::max(1, std::min(kMaxChains, atoi(e)));
  if (nz) {
    for (int k = 3; k + 4 <= T && (int)starts.size() < max_chains; ++k) {
      if (k - starts.back() < 3) continue;
      bool empty = true;
      for (int r = k; r <= k + 2 && empty; ++r)
        for (int c = 0; c < k && empty; ++c) empty = nz[(size_t)r * T + c] == 0;
      if (empty) starts.push_back(k);
    }
  }
  p.cr.n = (int)starts.size();
  p.chain_of.assign(T, 0);
  for (int c = 0; c < p.cr.n; ++c) {
    p.cr.begin[c] = starts[c]; p.cr.end[c] = c + 1 < p.cr.n ? starts[c + 1] : T;
    for (int k = p.cr.begin[c]; k < p.cr.end[c]; ++k) p.chain_of[k] = c;
  }
  if (nz) {
    p.map.assign(nz, nz + (size_t)T * T);
    for (int c = 0; c < p.cr.n; ++c)
      for (int k = p.cr.begin[c]; k < p.cr.end[c]; ++k)
        for (int i = k; i < p.cr.end[c] && i <= k + 2; ++i) p.map[(size_t)i * T + k] = 1;
    (void)SymbolicTileFill(T, p.map.data());
  }
  p.time.assign(T, 0);
  for (int k = 0; k < T; ++k) {
    // (the panels of ANOTHER chain reach the tiles of this column's tasks - rows k .. k+2: PrepX / PrepD(k) finish tiles of row k+2 - when that chain is
    // through: its merge tasks are listed a step behind the solves of its last block column, and they must be listed before this column's tasks)
    int t = 0;
    const int ck = p.chain_of[k];
    for (int j = 0; j < k; ++j) {
      if (!nz) { t = std::max(t, p.time[j] + 1); continue; }
      if (p.chain_of[j] == ck) { if (p.map[(size_t)k * T + j]) t = std::max(t, p.time[j] + 1); continue; }
      for (int r = k; r <= k + 2 && r < p.cr.end[ck]; ++r) if (p.map[(size_t)r * T + j]) t = std::max(t, p.time[p.cr.end[p.chain_of[j]] - 1] + 2);
    }
    p.time[k] = t;
  }
  std::vector<int> order(T);
  for (int k = 0; k < T; ++k) order[k] = k;
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return p.time[a] < p.time[b]; });
  p.rho1.assign(T, 0);
  for (int r = 0; r < T; ++r) p.rho1[order[r]] = r + 1;
  for (int c = 0; c < p.cr.n; ++c) p.cr.post[c] = p.cr.end[c] < T ? p.rho1[p.cr.end[c] - 2] : 0;
  return p;
}

// The task list of a plan.  A solve task exists per non-zero tile below the two sub-diagonals, an update task per super-tile and panel that couples one of
// its tiles; the values that depend on which tasks exist and on the elimination order (ChainTask::w0, w1, w2, a, b) are computed here.  The block columns
// are visited in the plan's order; the update tasks of panel k-1 are listed with block column k ("step k"), those of a stopping chain's last panel
// (end-1) in a pseudo step of their own (k = end) behind it.
// Priorities: a task's key is the one of the single-chain list with the step's TIME in place of its index - and never below the key of anything the task
// waits for (the task that stored the counter value it waits for, the prep tasks of the chain step whose mailbox it reads): tasks are generated in an
// order in which every task only waits for earlier ones, so one pass suffices and the sorted list is a topological order by construction (for one chain
// no key is ever raised: the list is what it was).
struct TaskListInfo { bool fits = true; int scratch_tiles = 0; };      // fits: the scratch sequences found counters; scratch_tiles: slots of the scratch tile pool
static std::vector<ChainTask> BuildTaskList(int T, const ChainPlan& plan, TaskListInfo* info = nullptr) {
  struct Item { double key; ChainTask t; };
  std::vector<Item> items;
  const uint8_t* nz = plan.map.empty() ? nullptr : plan.map.data();
  auto has = [&](int i, int j) { return i < T && j < T && (!nz || nz[(size_t)i * T + j] != 0); };
  const int whole_from = getenv("PPSFM_CHOL_WHOLE_FROM") ? atoi(getenv("PPSFM_CHOL_WHOLE_FROM")) : WholeFrom(T);
  const int nch = plan.cr.n;
  const double kNone = -1e30;
  // one SEQUENCE of updates per super-tile and accumulation target: the tiles themselves (panels of the chain that owns the super-tile's columns) or the
  // scratch array of another chain c (its panels; added to the tiles by one merge task when chain c is through).  Per sequence: ver / sub counter,
  // parts listed, the value of ver once the tasks listed so far are done, the key of the last task, the tiles touched so far (scratch: what is not zero yet)
  struct Seq { int cidx = 0, sidx = 0, listed = 0, post = 0, touched = 0; double key = -1e30; int slot[4] = {-1, -1, -1, -1}; };      // slot: the scratch tiles of a scratch sequence's four tiles
  std::vector<Seq> own(kMaxSuper * kMaxSuper);
  for (int I = 0; I < kMaxSuper; ++I) for (int J = 0; J < kMaxSuper; ++J) { own[I * kMaxSuper + J].cidx = cVer0 + I * kMaxSuper + J; own[I * kMaxSuper + J].sidx = cSub0 + I * kMaxSuper + J; }
  std::vector<std::vector<std::pair<int, Seq>>> scratch(nch);      // per chain: (I * kMaxSuper + J, sequence)
  int scratch_used = 0, slots_used = 0;
  bool ok = true;
  auto owner = [&](int J) { return plan.chain_of[std::min(2 * J, T - 1)]; };      // (a super-tile column that straddles two chains: its second block column starts a chain and never takes a panel)
  auto seq_of = [&](int c, int I, int J) -> Seq& {
    if (c == owner(J)) return own[I * kMaxSuper + J];
    for (auto& e : scratch[c]) if (e.first == I * kMaxSuper + J) return e.second;
    Seq q;
    if (scratch_used + 2 > kScratchCounters) ok = false; else { q.cidx = cScratch0 + scratch_used; q.sidx = cScratch0 + scratch_used + 1; scratch_used += 2; }
    scratch[c].push_back({I * kMaxSuper + J, q});
    return scratch[c].back().second;
  };
  std::vector<int> solpost((size_t)nch * (T + 4), 0);              // per chain: value of the row's sol counter once the solves listed so far are done
  std::vector<double> rowkey((size_t)nch * (T + 4), kNone);        // key of the last task that moves it
  std::vector<double> tilekey((size_t)(T + 4) * (T + 4), kNone);  // key of the task that solves tile (row, column)
  std::vector<double> stepkey(T + 4, kNone);                      // key of the last prep task chain step s takes its inputs from (a chain's first step: none)
  auto tk_ = [&](int r, int c) -> double& { return tilekey[(size_t)r * (T + 4) + c]; };
  auto sp = [&](int c, int row) -> int& { return solpost[(size_t)c * (T + 4) + row]; };
  auto rk = [&](int c, int row) -> double& { return rowkey[(size_t)c * (T + 4) + row]; };
  auto raised = [](double desired, std::initializer_list<double> deps) { double k = desired; for (double d : deps) k = std::max(k, d); return k; };
  const bool two_panels = !nz && !(getenv("PPSFM_CHOL_TWO_PANELS") && atoi(getenv("PPSFM_CHOL_TWO_PANELS")) == 0);      // (two panels per task: dense systems)
  const double slope = getenv("PPSFM_CHOL_SLOPE") ? atof(getenv("PPSFM_CHOL_SLOPE")) : kUpdateSlope;
  auto vp = [&](int I, int J) -> int& { return own[I * kMaxSuper + J].post; };
  auto vk = [&](int I, int J) -> double& { return own[I * kMaxSuper + J].key; };
  const std::vector<int>& time = plan.time;
  const std::vector<int>& rho1 = plan.rho1;
  auto own3 = [](int k, int r, int c) { return (r == k + 1 && c == k + 1) || (r == k + 2 && (c == k + 1 || c == k + 2)); };
  // the update tasks of panel k - 1 at (pseudo) step k, which happens at time ts
  auto list_updates = [&](int k, int ts, bool pseudo) {
    const int pc = plan.chain_of[k - 1], fl = pc << 4;
    for (int J = (k + 1) / 2; 2 * J < T; ++J)
      for (int I = J; 2 * I < T; ++I) {
        int tiles = 0;      // the tiles of the region below / right of (k+1,k+1) that are not one of the chain's / prep's three and that panel k-1 couples (the device's `valid`)
        for (int q = 0; q < 4; ++q) {
          const int bi = 2 * I + (q >> 1), bj = 2 * J + (q & 1);
          if (bi < T && bj < T && bi >= bj && bj >= k + 1 && !own3(k, bi, bj) && has(bi, k - 1) && has(bj, k - 1)) tiles |= 1 << q;
        }
        if (!tiles) continue;
        Seq& sq = seq_of(pc, I, J);
        const bool into_scratch = pc != owner(J);
        const int zsel = into_scratch ? pc : -1;
        // what the task waits for: the sequence's previous update, column k-1 of the block rows it reads
        double dep = sq.key;
        for (int row : {2 * I, 2 * I + 1, 2 * J, 2 * J + 1})
          if (row < T && row >= k + 1 && has(row, k - 1)) dep = std::max(dep, tk_(row, k - 1));
        // the time at which the super-tile's columns become the front, in steps from now (2J - (k+1) for one chain)
        const int tJ = std::min(time[2 * J], 2 * J + 1 < T ? time[2 * J + 1] : time[2 * J]);
        const int Jt = std::max(tJ / 2, (ts + 1) / 2);
        // in parts (UpdateTilesTask): four single tiles for the super-tiles PrepX(k+1) / PrepD(k+1) wait for, two block rows otherwise
        const bool front = !pseudo && I == (k + 3) / 2 && (J == I - 1 || J == I);
        const bool far = Jt - (ts + 1) / 2 >= whole_from;
        // far at the next step too: steps k (odd) and k + 1 in one task, listed where step k + 1's update would be
        const bool far_next = two_panels && k + 2 < T && J - (k + 2) / 2 >= whole_from;
        if (far && (k & 1) == 0 && two_panels) continue;      // (the odd step before it took this one along: far at k => far at k - 1)
        if (far && far_next && (k & 1) == 1) {
          sq.listed += 1;
          for (int row : {2 * I, 2 * I + 1, 2 * J, 2 * J + 1}) if (row < T && row >= k + 1) dep = std::max(dep, tk_(row, k));      // (column k as well)
          const double key = raised((k + 1) + slope * (J - 0.5 * (k + 2)), {dep});
          items.push_back({key, {kTaskUpdate, k, I, J | (kPartsTwoPanels << 12) | (sq.listed << 16), sq.post, k + 1, k, 0, sq.cidx, sq.sidx, -1, 0, {0, 0, 0, 0}}});
          sq.post = k + 1; sq.key = key;
          continue;
        }      // (whole: the least operand traffic per flop; a far super-tile has steps of slack.  A lower
               // threshold for the first steps, where the bulk is the bound: +-1 %, not kept)
        const int parts = front ? 4 : (far ? 1 : 2);      // (four tiles also for the next ring of super-tiles, other slopes of the priority: measured, no gain)
        sq.listed += parts;
        const double dist = std::max(0.5 * tJ - 0.5 * (ts + 1), -0.5);
        const double key = raised(front ? ts - 0.2 : ts + slope * dist, {dep});
        const int post = std::max(rho1[k - 1], sq.post + 1);      // (k for one chain; several sequences and merges move a separator's counters)
        const int fresh = into_scratch ? (tiles & ~sq.touched) : 0;
        if (into_scratch) for (int q = 0; q < 4; ++q) if (((fresh >> q) & 1) && sq.slot[q] < 0) sq.slot[q] = slots_used++;
        for (int q = 0; q < parts; ++q)
          items.push_back({key, {kTaskUpdate, k, I, J | (q << 8) | (parts << 12) | (sq.listed << 16), sq.post, post, rho1[k - 1], fl, sq.cidx, sq.sidx, zsel, fresh,
                                 {sq.slot[0], sq.slot[1], sq.slot[2], sq.slot[3]}}});
        sq.post = post; sq.key = key; sq.touched |= tiles;
      }
  };
  // chain c is through (its last panel's updates are listed): what it accumulated for other chains' super-tiles joins their own sequences
  auto list_merges = [&](int c, int ts) {
    for (auto& e : scratch[c]) {
      const int I = e.first / kMaxSuper, J = e.first % kMaxSuper;
      Seq& z = e.second;
      Seq& o = own[e.first];
      const double key = raised(ts + 0.05, {z.key, o.key});
      const int post = o.post + 1;
      items.push_back({key, {kTaskMerge, plan.cr.end[c], I, J, o.post, post, z.post, 0, o.cidx, z.cidx, c, z.touched, {z.slot[0], z.slot[1], z.slot[2], z.slot[3]}}});
      o.post = post; o.key = key;
    }
  };
  struct Event { int t, kind, k; };
  std::vector<Event> events;
  for (int k = 0; k + 1 < T; ++k) events.push_back({time[k], 0, k});
  for (int c = 0; c + 1 < nch; ++c) events.push_back({time[plan.cr.end[c] - 1] + 1, 1, plan.cr.end[c]});
  std::stable_sort(events.begin(), events.end(), [](const Event& a, const Event& b) { return a.t != b.t ? a.t < b.t : (a.kind != b.kind ? a.kind < b.kind : a.k < b.k); });
  for (const Event& ev : events) {
    const int k = ev.k;
    if (ev.kind == 1) { list_updates(k, ev.t, true); list_merges(plan.chain_of[k - 1], ev.t); continue; }
    const int c = plan.chain_of[k], e = plan.cr.end[c];
    const bool first = k == plan.cr.begin[c];
    const int fl = (first ? 1 : 0) | (c << 4), tk = time[k];
    const int prev_done = first ? 0 : rho1[k - 1];
    const double step_prev = first ? kNone : stepkey[k - 1];      // M_k and the solved tile (k,k-1): chain step k-1
    if (k + 2 < e) {
      // (their ver waits only exist behind a chain's first step: PrepTask's `prev`)
      const int I2 = (k + 2) >> 1;
      const int wx0 = !first ? vp(I2, k >> 1) : 0, wx1 = !first ? vp(I2, (k + 1) >> 1) : 0, wd1 = !first ? vp(I2, (k + 2) >> 1) : 0;
      const double far_key = !first && has(k + 2, k - 1) ? tk_(k + 2, k - 1) : kNone;
      const double kx = first ? tk - 0.4 : raised(tk - 0.4, {vk(I2, k >> 1), vk(I2, (k + 1) >> 1), rk(c, k + 2), rk(c, k + 1), step_prev, stepkey[k]});
      const double kd = first ? tk - 0.4 : raised(tk - 0.4, {vk(I2, k >> 1), vk(I2, (k + 2) >> 1), far_key, step_prev});
      // PrepX: a = what sol[k+1] must have reached (column k-1 solved - by PrepX(k-1)), w2 = the same for sol[k+2];  PrepD: a = "column k-1 of row k+2 is solved"
      items.push_back({kx, {kTaskPrepX, k, first ? 0 : sp(c, k + 1), rho1[k], wx0, wx1, sp(c, k + 2), fl, 0, 0, -1, 0, {0, 0, 0, 0}}});
      items.push_back({kd, {kTaskPrepD, k, prev_done, rho1[k], wx0, wd1, 0, fl, 0, 0, -1, 0, {0, 0, 0, 0}}});
      tk_(k + 2, k) = kx; tk_(k + 1, k) = kx; rk(c, k + 2) = kx; rk(c, k + 1) = kx;
      stepkey[k + 1] = std::max(std::max(kx, kd), stepkey[k]);
      sp(c, k + 2) = rho1[k]; sp(c, k + 1) = rho1[k];
    } else if (k + 1 < e) {      // the last step of a chain that stops: the chain itself stores the tile and moves the counter
      tk_(k + 1, k) = stepkey[k]; rk(c, k + 1) = std::max(rk(c, k + 1), stepkey[k]);
      sp(c, k + 1) = rho1[k];
    }
    for (int i = k + 3; i < T; ++i)
      if (has(i, k)) {
        const double key = raised(tk - 0.3, {vk(i >> 1, k >> 1), rk(c, i), step_prev});
        items.push_back({key, {kTaskSolve, k, i, 0, vp(i >> 1, k >> 1), rho1[k], sp(c, i), fl, 0, 0, -1, 0, {0, 0, 0, 0}}});
        sp(c, i) = rho1[k]; tk_(i, k) = key; rk(c, i) = key;
      }
    if (!first) list_updates(k, tk, false);
  }
  // the pair inverses / couplings of the paired back substitution (dense systems): off every critical path, behind the tasks of step 2g + 2
  if (!nz)
    for (int gp = 0; gp < BacksubNumPairs(T); ++gp)
      for (int part = 0; part < (gp + 1 < BacksubNumPairs(T) ? 3 : 1); ++part) items.push_back({2 * gp + 2.2, {kTaskPairPrep, 2 * gp + 2, gp, part, 0, 0, 0, 0, 0, 0, -1, 0, {0, 0, 0, 0}}});
  std::stable_sort(items.begin(), items.end(), [](const Item& a, const Item& b) { return a.key < b.key; });
  std::vector<ChainTask> list(items.size());
  for (size_t i = 0; i < items.size(); ++i) list[i] = items[i].t;
  if (info) { info->fits = ok; info->scratch_tiles = slots_used; }
  return list;
}

// Replay of a list on the host (what tests/test_cholesky_task_order.py does for a set of shapes, here for the structure at hand):
//   * every counter value a task waits for has been stored by a task EARLIER in the list (or by a chain whose inputs were), and every counter only
//     grows - what makes the one launch free of deadlocks however few workgroups are resident;
//   * every tile has received exactly the panels that couple it when a task consumes it, every operand is solved, every non-zero tile gets solved.
static bool TaskListWaitsAreMet(int T, const ChainPlan& plan, const std::vector<ChainTask>& list) {
  const uint8_t* nz = plan.map.empty() ? nullptr : plan.map.data();
  auto has = [&](int i, int j) { return i < T && j < T && (!nz || nz[(size_t)i * T + j] != 0); };
  const int nch = plan.cr.n;
  std::vector<int> ctr(kNumCounters, 0);      // the device's counters
  auto ver = [&](int I, int J) -> int& { return ctr[cVer0 + I * kMaxSuper + J]; };
  auto sol = [&](int c, int row) -> int& { return ctr[cSol0 + c * kMaxSteps + row]; };
  std::vector<char> px(T + 2, 0), pd(T + 2, 0), solved((size_t)T * T, 0);
  struct Bits { uint64_t w[2] = {0, 0}; bool operator==(const Bits& o) const { return w[0] == o.w[0] && w[1] == o.w[1]; } bool none() const { return !w[0] && !w[1]; } };
  std::vector<Bits> applied((size_t)T * T);
  std::vector<std::vector<Bits>> zapplied(nch, std::vector<Bits>((size_t)T * T));      // per chain: what sits in its scratch tiles
  std::vector<std::vector<int>> zslot(nch, std::vector<int>((size_t)T * T, -1));        // ... and where: a slot of the pool per (chain, tile), nobody else's
  std::vector<char> slot_taken;
  auto bit = [](Bits* b, int p) { b->w[p >> 6] |= 1ull << (p & 63); };
  auto coupling = [&](int r, int c, int below) {      // the panels p < below that couple tile (r,c)
    Bits b;
    for (int p = 0; p < below && p < c; ++p) if (has(r, p) && has(c, p)) bit(&b, p);
    return b;
  };
  auto own = [](int k, int r, int c) { return (r == k + 1 && c == k + 1) || (r == k + 2 && (c == k + 1 || c == k + 2)); };
  // chain step s (the solve of tile (s+1,s), M_(s+1)) can run: its inputs come from k_potrf64 (a chain's first step) or from PrepX / PrepD(s-1), and step s-1 ran
  auto can_run = [&](int s) {
    const int c = plan.chain_of[s], b = plan.cr.begin[c];
    if (s + 1 >= plan.cr.end[c]) return false;
    for (int q = b + 1; q <= s; ++q) if (!px[q - 1] || !pd[q - 1]) return false;
    return true;
  };
  auto chain_stores = [&](int row, int col) {      // tile (row,col) is the last solved tile of a chain that stops, and that step can run
    const int c = plan.chain_of[col];
    return plan.cr.end[c] < T && row == plan.cr.end[c] - 1 && col == row - 1 && can_run(col);
  };
  for (const ChainTask& t : list) {
    const int k = t.k;
    const bool first = (t.flags & 1) != 0;
    const int fc = (t.flags >> 4) & 15;
    if (t.type == kTaskPrepX || t.type == kTaskPrepD) {
      const bool X = t.type == kTaskPrepX;
      const int oc = X ? k + 1 : k + 2;
      if (fc != plan.chain_of[k] || k + 2 >= plan.cr.end[fc] || first != (k == plan.cr.begin[fc])) return false;
      if (!first) {
        if (ver((k + 2) >> 1, k >> 1) < t.w0) return false;
        if (ver((k + 2) >> 1, oc >> 1) < t.w1) return false;
        const bool far = has(k + 2, k - 1);
        if (sol(fc, k + 2) < (X ? t.w2 : (far ? t.a : 0))) return false;
        if (X && sol(fc, k + 1) < t.a) return false;
        if (!can_run(k - 1)) return false;      // M_k, the solved tile (k,k-1)
        if (far && !solved[(size_t)(k + 2) * T + k - 1]) return false;
        if (X && !solved[(size_t)(k + 1) * T + k - 1]) return false;
      }
      // the update tasks have applied every panel below k-1 (k-1 and k the task applies itself; a chain's first step: there are none at all)
      if (!(applied[(size_t)(k + 2) * T + k] == coupling(k + 2, k, first ? k : k - 1))) return false;
      if (!(applied[(size_t)(k + 2) * T + oc] == coupling(k + 2, oc, first ? k : k - 1))) return false;
      if (X) {
        if (!can_run(k)) return false;          // the solved tile (k+1,k)
        if (sol(fc, k + 2) >= t.b || sol(fc, k + 1) >= t.b) return false;
        sol(fc, k + 2) = t.b; sol(fc, k + 1) = t.b;
        solved[(size_t)(k + 2) * T + k] = 1; solved[(size_t)(k + 1) * T + k] = 1;
        px[k] = 1;
      } else pd[k] = 1;
    } else if (t.type == kTaskSolve) {
      const int i = t.a;
      if (fc != plan.chain_of[k] || i < k + 3 || i >= T || !has(i, k) || first != (k == plan.cr.begin[fc])) return false;
      if (sol(fc, i) < t.w2 || ver(i >> 1, k >> 1) < t.w0) return false;
      if (!first && (!can_run(k - 1) || (has(i, k - 1) && !solved[(size_t)i * T + k - 1]))) return false;
      if (!(applied[(size_t)i * T + k] == coupling(i, k, first ? k : k - 1))) return false;
      if (sol(fc, i) >= t.w1) return false;
      sol(fc, i) = t.w1;
      solved[(size_t)i * T + k] = 1;
    } else if (t.type == kTaskMerge) {
      const int I = t.a, J = t.b & 255, c = t.zsel;
      if (c < 0 || c >= nch || t.cidx != cVer0 + I * kMaxSuper + J) return false;
      if (ctr[t.cidx] < t.w0 || ctr[t.sidx] < t.w2) return false;
      for (int q = 0; q < 4; ++q) {
        const int r = 2 * I + (q >> 1), cc = 2 * J + (q & 1);
        if (r >= T || cc >= T) { if ((t.mask >> q) & 1) return false; continue; }
        Bits& z = zapplied[c][(size_t)r * T + cc];
        if (((t.mask >> q) & 1) != (z.none() ? 0 : 1)) return false;
        if (!z.none() && t.slot[q] != zslot[c][(size_t)r * T + cc]) return false;      // ... from the scratch tile they were accumulated in
        Bits& a = applied[(size_t)r * T + cc];
        if ((a.w[0] & z.w[0]) || (a.w[1] & z.w[1])) return false;
        a.w[0] |= z.w[0]; a.w[1] |= z.w[1];
        z = Bits();
      }
      if (ctr[t.cidx] >= t.w1) return false;
      ctr[t.cidx] = t.w1;
    } else if (t.type == kTaskUpdate) {
      const int I = t.a, J = t.b & 255, part = (t.b >> 8) & 15, parts = (t.b >> 12) & 15, target = t.b >> 16;
      const bool two = parts == kPartsTwoPanels;
      if (fc != plan.chain_of[k - 1] || t.zsel >= nch || (t.zsel >= 0 && t.zsel != fc)) return false;
      if (t.zsel < 0 && (t.cidx != cVer0 + I * kMaxSuper + J || t.sidx != cSub0 + I * kMaxSuper + J)) return false;
      if (ctr[t.cidx] < t.w0) return false;
      auto row_ok = [&](int row, bool distinct) {
        if (!(distinct && row < T && row >= k + 1 && has(row, k - 1))) return true;
        int have = sol(fc, row);
        if (chain_stores(row, k - 1)) have = std::max(have, plan.cr.post[fc]);
        return have >= (two ? t.w2 + 1 : t.w2);
      };
      const int bi = 2 * I + (parts == 2 ? part : part >> 1), bj0 = 2 * J + (parts == 2 ? 0 : part & 1), nb = parts == 2 ? 2 : 1;
      bool ok;
      if (parts == 1 || two) ok = row_ok(2 * I, true) && row_ok(2 * I + 1, true) && row_ok(2 * J, J != I) && row_ok(2 * J + 1, J != I);
      else ok = row_ok(bi, true) && row_ok(bj0, bj0 != bi) && row_ok(bj0 + 1, nb == 2 && bj0 + 1 != bi);
      if (!ok) return false;
      // the tiles it updates (the device's `valid`), panel k-1 (and k: two)
      for (int q = 0; q < 4; ++q) {
        const int r = 2 * I + (q >> 1), c = 2 * J + (q & 1);
        const bool mine = (parts == 1 || two) || (parts == 2 ? (q >> 1) == part : q == part);
        for (int kk = k; kk <= (two ? k + 1 : k); ++kk) {
          const bool valid = r < T && c < T && r >= c && c >= kk + 1 && !own(kk, r, c) && has(r, kk - 1) && has(c, kk - 1);
          if (!mine || !valid) continue;
          for (int row : {r, c}) if (!solved[(size_t)row * T + kk - 1] && !chain_stores(row, kk - 1) && !(two && kk == k + 1)) return false;
          // a panel goes to the tile itself exactly when its chain owns the tile's block column
          if ((t.zsel < 0) != (plan.chain_of[kk - 1] == plan.chain_of[c])) return false;
          Bits& a = t.zsel < 0 ? applied[(size_t)r * T + c] : zapplied[t.zsel][(size_t)r * T + c];
          if (t.zsel >= 0 && (((t.mask >> q) & 1) != (a.none() ? 1 : 0))) return false;      // taken as zero exactly when nothing has been accumulated yet
          if (t.zsel >= 0) {
            int& zs = zslot[t.zsel][(size_t)r * T + c];
            if (t.slot[q] < 0) return false;
            if (zs < 0) {
              if ((int)slot_taken.size() <= t.slot[q]) slot_taken.resize(t.slot[q] + 1, 0);
              if (slot_taken[t.slot[q]]) return false;
              slot_taken[t.slot[q]] = 1; zs = t.slot[q];
            } else if (zs != t.slot[q]) return false;
          }
          if (a.w[(kk - 1) >> 6] >> ((kk - 1) & 63) & 1) return false;
          bit(&a, kk - 1);
        }
      }
      if (++ctr[t.sidx] == target) {
        if (ctr[t.cidx] >= t.w1) return false;
        ctr[t.cidx] = t.w1;
      }
    }
  }
  // every non-zero tile below the diagonal is solved, every tile got the panels that couple it (those its own tasks apply aside), nothing is left in a scratch array
  for (int c = 0; c + 1 < T; ++c)
    for (int r = c + 1; r < T; ++r) {
      if (!has(r, c)) continue;
      const int e = plan.cr.end[plan.chain_of[c]];
      if (r == c + 1 && r < e) { if (!solved[(size_t)r * T + c] && !(c + 2 >= e && can_run(c))) return false; }
      else if (!solved[(size_t)r * T + c]) return false;
    }
  for (int c = 1; c < T; ++c)
    for (int r = c; r < T; ++r) {
      for (int ch = 0; ch < nch; ++ch) if (!zapplied[ch][(size_t)r * T + c].none()) return false;
      if (!has(r, c)) continue;
      Bits want = coupling(r, c, c);
      auto clear = [&](int p) { if (p >= 0) want.w[p >> 6] &= ~(1ull << (p & 63)); };
      clear(c - 1); if (r <= c + 1) clear(c - 2); if (r == c) clear(c - 3);
      if (!(applied[(size_t)r * T + c] == want)) return false;
    }
  return true;
}

// plan + list for a tile map (null: dense); a plan of several chains whose list does not pass the replay falls back to ONE chain (the former behaviour)
static ChainPlan PlanAndList(int T, const uint8_t* nz, std::vector<ChainTask>* list, bool* verified = nullptr, int* scratch_tiles = nullptr) {
  ChainPlan plan = PlanChains(T, nz);
  TaskListInfo info;
  *list = BuildTaskList(T, plan, &info);
  bool ok = info.fits && TaskListWaitsAreMet(T, plan, *list);
  if (!ok && plan.cr.n > 1) {
    fprintf(stderr, "ppsfm: the task list of %d chains over %d block columns did not pass its replay - one chain\n", plan.cr.n, T);
    plan = PlanChains(T, nz, 1);
    *list = BuildTaskList(T, plan, &info);
    ok = TaskListWaitsAreMet(T, plan, *list);
  }
  if (getenv("PPSFM_CHOL_PLAN_PRINT")) {      // (debugging aid: the chains and the closed tile map of the structure at hand)
    fprintf(stderr, "ppsfm plan: T %d, %d chains:", T, plan.cr.n);
    for (int c = 0; c < plan.cr.n; ++c) fprintf(stderr, " [%d,%d)", plan.cr.begin[c], plan.cr.end[c]);
    fprintf(stderr, "\n");
    for (int r = 0; r < T && !plan.map.empty(); ++r) { for (int c = 0; c <= r; ++c) fputc(plan.map[(size_t)r * T + c] ? '#' : '.', stderr); fputc('\n', stderr); }
  }
  if (verified) *verified = ok;
  if (scratch_tiles) *scratch_tiles = info.scratch_tiles;
  return plan;
}

int CholeskyPlanSteps(int T, const uint8_t* nz, int* chains) {
  if (chains) *chains = 1;
  if (!nz || T < 4 || T > kMaxSteps) return T;
  const ChainPlan plan = PlanChains(T, nz);
  if (chains) *chains = plan.cr.n;
  int steps = 0;
  for (int k = 0; k < T; ++k) steps = std::max(steps, plan.time[k] + 1);
  return steps;
}


static std::recursive_mutex g_setup_mutex;
// Plans are kept process-wide, keyed on the tile map: the mapper builds a new BundleAdjuster per global bundle adjustment (src/sfm/incremental_mapper.cc:893-936)
// and an unchanged structure (the dense list of a size; the same sequence a second time) must not pay the plan, the list and its replay again.
struct CachedPlan { int T; std::vector<uint8_t> key; std::string knobs; ChainPlan plan; std::vector<ChainTask> list; bool verified; int scratch_tiles; };
// the environment switches PlanChains / BuildTaskList read (tests and tools turn them): part of the key - a plan made under other switches is another plan
// (without this the switches stopped doing anything once a matrix's plan was cached, and the tests that compare them compared a plan with itself)
static std::string PlanKnobs() {
  std::string k;
  for (const char* name : {"PPSFM_CHOL_CHAINS", "PPSFM_CHOL_WHOLE_FROM", "PPSFM_CHOL_TWO_PANELS", "PPSFM_CHOL_SLOPE"}) { const char* e = getenv(name); k += e ? e : ""; k += '|'; }
  return k;
}
static std::vector<CachedPlan> g_plan_cache;      // (guarded by g_setup_mutex, most recently used last)
static constexpr size_t kPlanCacheEntries = 16, kPlanCacheBytes = (size_t)32 << 20;
static const CachedPlan& PlanCached(int T, const uint8_t* nz) {
  const size_t bytes = nz ? (size_t)T * T : 0;
  const std::string knobs = PlanKnobs();
  for (size_t i = g_plan_cache.size(); i-- > 0;) {
    CachedPlan& c = g_plan_cache[i];
    if (c.T == T && c.key.size() == bytes && c.knobs == knobs && (bytes == 0 || std::memcmp(c.key.data(), nz, bytes) == 0)) {
      if (i + 1 != g_plan_cache.size()) std::rotate(g_plan_cache.begin() + i, g_plan_cache.begin() + i + 1, g_plan_cache.end());
      return g_plan_cache.back();
    }
  }
  // (least recently used first; an entry is its list - 64 bytes per task, ~5 000 tasks at 47 dense block columns, ~100 000 at 128 - plus its maps: bounded by
  // entries AND bytes)
  auto bytes_of = [](const CachedPlan& e) { return e.list.size() * sizeof(ChainTask) + e.key.size() + e.plan.map.size(); };
  size_t held = 0;
  for (const CachedPlan& e : g_plan_cache) held += bytes_of(e);
  while (!g_plan_cache.empty() && (g_plan_cache.size() >= kPlanCacheEntries || held > kPlanCacheBytes)) { held -= bytes_of(g_plan_cache.front()); g_plan_cache.erase(g_plan_cache.begin()); }
  g_plan_cache.emplace_back();
  CachedPlan& c = g_plan_cache.back();
  c.T = T; c.knobs = knobs;
  if (nz) c.key.assign(nz, nz + bytes);
  c.plan = PlanAndList(T, nz, &c.list, &c.verified, &c.scratch_tiles);
  return c;
}

int CholeskyChainSteps(int T, const uint8_t* nz, int* chains) {
  if (chains) *chains = 1;
  if (!nz || T < 4 || T > kMaxSteps) return T;
  std::lock_guard<std::recursive_mutex> lock(g_setup_mutex);
  const CachedPlan& cp = PlanCached(T, nz);      // (plan + list + replay, once per tile map: pp_ba_create's winner is what EnsureTaskList asks for next)
  if (chains) *chains = cp.plan.cr.n;
  int steps = 0;
  for (int k = 0; k < T; ++k) steps = std::max(steps, cp.plan.time[k] + 1);
  return steps;
}

static int EnsureTaskList(CholeskyAux* aux, int T, hipStream_t strm) {
  const bool block_sparse = aux->tile_nz && aux->tile_T == T;
  const uint8_t* src = block_sparse ? aux->tile_nz : nullptr;
  if (aux->tasks_T == T && aux->tasks_src_nz == src && (aux->tasks || aux->tasks_rejected)) return PP_OK;
  const auto t0 = std::chrono::steady_clock::now();
  if (aux->tasks) { (void)hipFree(aux->tasks); aux->tasks = nullptr; }
  if (aux->tasks_nz) { (void)hipFree(aux->tasks_nz); aux->tasks_nz = nullptr; }
  std::lock_guard<std::recursive_mutex> lock(g_setup_mutex);
  const CachedPlan& cp = PlanCached(T, src);
  const ChainPlan& plan = cp.plan;
  aux->tasks_T = T;
  aux->tasks_src_nz = src;
  aux->tasks_rejected = !cp.verified;
  if (!cp.verified) {
    // a list whose host replay finds a wait that no earlier task meets is not launched at all (it would run into the device's bounded waits):
    // this structure is factorised by per-column launches over its tile lists
    fprintf(stderr, "ppsfm: the one-launch task list of %d block columns (%s) did not pass its replay - per-column launches for this structure\n", T, src ? "block-sparse" : "dense");
    aux->num_tasks = 0;
    aux->plan_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return PP_OK;
  }
  if (block_sparse) {
    PP_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&aux->tasks_nz), plan.map.size()));
    PP_HIP_TRY(hipMemcpyAsync(aux->tasks_nz, plan.map.data(), plan.map.size(), hipMemcpyHostToDevice, strm));
  }
  // (measured and dropped in round 6, PPSFM_CHOL_QUIET_XCD: an empty task at every list position whose workgroup lands on the chain's XCD - a quieter L2 /
  // fabric port for the chain, a seventh less of the chip for the bulk: 705-720 -> 721-731 us per factorisation + back substitution, 809 -> 824 us per LM iteration)
  const std::vector<ChainTask>* to_upload = &cp.list;
  PP_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&aux->tasks), sizeof(ChainTask) * to_upload->size()));
  // (on the caller's stream, not the legacy one: another host thread may be capturing its own factorisation just now)
  PP_HIP_TRY(hipMemcpyAsync(aux->tasks, to_upload->data(), sizeof(ChainTask) * to_upload->size(), hipMemcpyHostToDevice, strm));
  PP_HIP_TRY(hipStreamSynchronize(strm));
  aux->num_tasks = (int)to_upload->size();
  // several chains: the pool of 64 x 64 scratch tiles in which a chain accumulates for another chain's tiles
  if (cp.scratch_tiles > aux->scratch_tiles) {
    if (aux->scratch) { PoolDeviceFree(aux->scratch); aux->scratch = nullptr; aux->scratch_tiles = 0; }      // (recycled blocks: resource_pool.hpp)
    { const int rc = PoolDeviceAlloc(reinterpret_cast<void**>(&aux->scratch), sizeof(double) * (size_t)cp.scratch_tiles * kNB * kNB); if (rc) return rc; }
    aux->scratch_tiles = cp.scratch_tiles;
  }
  static_assert(sizeof(aux->chains) == sizeof(ChainRanges), "CholeskyAux::chains holds a ChainRanges");
  std::memcpy(aux->chains, &plan.cr, sizeof(ChainRanges));
  aux->critical_path = 0;
  for (int k = 0; k < T; ++k) aux->critical_path = std::max(aux->critical_path, plan.time[k] + 1);
  if (getenv("PPSFM_CHOL_DEBUG") && plan.cr.n > 1) {
    fprintf(stderr, "ppsfm: %d chains over %d block columns, %d tasks, %d scratch tiles:", plan.cr.n, T, (int)cp.list.size(), cp.scratch_tiles);
    for (int c = 0; c < plan.cr.n; ++c) fprintf(stderr, " [%d,%d) t=%d..%d", plan.cr.begin[c], plan.cr.end[c], plan.time[plan.cr.begin[c]], plan.time[plan.cr.end[c] - 1]);
    fprintf(stderr, "\n");
  }
  aux->plan_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  return PP_OK;
}

// the back substitution: pairs for a dense factor (k_backsub_prepare + k_backsub_pairs), block by block for a block-sparse one or a
// small system.  Lw: where the factor's solved tiles live (S in per-column mode, the solved-tile array in task mode).
static void LaunchBacksub(const double* Lw, int N, int T, int rhs_row, double* Linv_ws, double* x_out, int32_t* d_flag, hipStream_t s, const uint8_t* nz,
                          bool prepared = false) {
  const int npairs = BacksubNumPairs(T);
  const char* pairs_env = getenv("PPSFM_BACKSUB_PAIRS");      // 0: block by block also for dense systems (what a block-sparse system always takes; tests compare the two)
  const bool pairs_off = pairs_env && atoi(pairs_env) == 0;
  if (nz || npairs == 0 || pairs_off) {
    hipLaunchKernelGGL(k_backsub_all, dim3(T), dim3(256), 0, s, Lw, N, T, rhs_row, (const double*)Linv_ws, x_out, d_flag, nz);
    return;
  }
  const size_t tile = (size_t)kNB * kNB;
  double* Pw = Linv_ws + (size_t)(4 * T + 3) * tile + 8192;      // behind the mailboxes and the counters (CholeskyWorkspaceDoubles)
  double* Zw = Pw + (size_t)npairs * tile;
  if (!prepared) hipLaunchKernelGGL(k_backsub_prepare, dim3(3 * npairs), dim3(kPanelThreads), 0, s, Lw, N, T, (const double*)Linv_ws, Pw, Zw);
  hipLaunchKernelGGL(k_backsub_pairs, dim3(T - 2 * npairs + npairs), dim3(kPairThreads), 0, s, Lw, N, T, rhs_row, (const double*)Linv_ws, (const double*)Pw,
                     (const double*)Zw, x_out, d_flag);
}

// A system of one or two block columns (at most 21 images: the mapper's local bundle adjustment) in ONE launch of one workgroup: the tiles
// come into LDS, SmallFactorSolveTiles does the rest.  (k_potrf64 + k_column_step + k_backsub_all: three launches, ~40 us of a ~105 us LM
// iteration at 20 images.)  Neither the factor nor the block inverses leave the CU: nothing after the solve reads them.
__global__ __launch_bounds__(kPanelThreads) void k_small_cholesky(const double* __restrict__ S, int N, int rhs_row, double* __restrict__ x_out, int32_t* __restrict__ flag) {
  __shared__ __attribute__((aligned(16))) double tiles[4 * kNB * kLS];
  __shared__ double inv_diag[kNB], xs[2 * kNB], ys[2 * kNB];
  const int tid = threadIdx.x, T = N / kNB;
  LoadTile(tiles, S, N, tid);
  if (T == 2) LoadTiles2(tiles + kNB * kLS, S + (size_t)kNB * N, tiles + 2 * kNB * kLS, S + (size_t)kNB * N + kNB, N, tid);
  __syncthreads();
  SmallFactorSolveTiles(tiles, inv_diag, xs, ys, flag, T, rhs_row, x_out);
}
static bool UseSmallCholesky(int N) {
  static const bool enabled = []() { const char* e = getenv("PPSFM_CHOL_SMALL"); return !(e && atoi(e) == 0); }();
  return enabled && N <= 2 * kNB;
}

// enqueue the whole factorisation + solve on stream s
// Linv_ws: [0, N*64) L_kk^-1 (row-major 64x64) of every diagonal block = the M_k mailboxes (solves + back substitution), then T+1
// staging slots for the chain's X tile (per-column mode uses two of them in turn; task mode: one mailbox per step), then the D and
// solved-X mailboxes of task mode (T+1 slots each), then the progress counters of task mode.
// Lfac (may be null): the solved-tile array of task mode, N x N like S.
static int EnqueueCholesky(double* S, int N, int rhs_row, double* Linv_ws, double* Lfac, double* x_out, int32_t* d_flag, hipStream_t s, CholeskyAux* aux) {
  const int T = N / kNB;
  const size_t tile = (size_t)kNB * kNB;
  if (UseSmallCholesky(N) && x_out) {
    if (aux) aux->last_used = PP_LINSOLVE_CHOLESKY_COLUMNS;
    hipLaunchKernelGGL(k_small_cholesky, dim3(1), dim3(kPanelThreads), 0, s, (const double*)S, N, rhs_row, x_out, d_flag);
    PP_HIP_TRY(hipGetLastError());
    return PP_OK;
  }
  Mailboxes mb;
  mb.Minv = Linv_ws; mb.xs = Linv_ws + (size_t)T * tile; mb.ds = mb.xs + (size_t)(T + 1) * tile; mb.xsol = mb.ds + (size_t)(T + 1) * tile;
  double* xs = mb.xs;
  int32_t* ctr = reinterpret_cast<int32_t*>(mb.xsol + (size_t)(T + 1) * tile);
  mb.Pw = Linv_ws + (size_t)(4 * T + 3) * tile + 8192;      // behind the mailboxes and the counters (CholeskyWorkspaceDoubles)
  mb.Zw = mb.Pw + (size_t)BacksubNumPairs(T) * tile;
  const bool block_sparse = aux && aux->tile_nz && aux->tile_T == T;
  const bool tasks = aux && UseTasks(aux->mode, T) && Lfac && aux->tasks && aux->tasks_T == T && aux->tasks_src_nz == (block_sparse ? aux->tile_nz : nullptr);
  const bool sparse = !tasks && aux && aux->sparse_lists && aux->sparse_T == T;      // (per-column launches over the non-zero tiles: above 128 block columns, or after a fallback)
  if (aux) aux->last_used = block_sparse ? PP_LINSOLVE_CHOLESKY_SPARSE : (tasks ? PP_LINSOLVE_CHOLESKY_TASKS : PP_LINSOLVE_CHOLESKY_COLUMNS);
  ChainRanges cr = OneChain(T);
  if (tasks) std::memcpy(&cr, aux->chains, sizeof(cr));
  hipLaunchKernelGGL(k_potrf64, dim3(tasks ? 64 + cr.n : 1), dim3(kPanelThreads), 0, s, S, N, Linv_ws, xs, d_flag, x_out, tasks ? Lfac : S, ctr, (int)kNumCounters, Linv_ws,
                     (long long)((size_t)(4 * T + 3) * tile), cr);
  if (tasks) {
    // ONE launch: workgroup 0 = the chain, then the task list (see k_cholesky_tasks)
    // (test hook: with half of the task list missing the chain's wait for a prep task runs into its bound - the host must then repeat
    // the solve with per-column launches, tests/test_gpu_bundle_adjustment.py::test_task_mode_timeout_falls_back_to_column_launches)
    const int grid_tasks = aux->test_drop_tasks ? aux->num_tasks / 2 : aux->num_tasks;
    const uint8_t* nz = block_sparse ? (const uint8_t*)aux->tasks_nz : (const uint8_t*)nullptr;
    hipLaunchKernelGGL(k_cholesky_tasks, dim3(cr.n + grid_tasks), dim3(kPanelThreads), 0, s, S, Lfac, N, T, mb, d_flag, ctr, aux->tasks, nz, cr, aux->scratch);
    LaunchBacksub(Lfac, N, T, rhs_row, Linv_ws, x_out, d_flag, s, nz, /*prepared=*/true);      // (dense: the kTaskPairPrep tasks of the launch above; block-sparse: block by block over the non-zero tiles)
    PP_HIP_TRY(hipGetLastError());
    return PP_OK;
  }
  // per-column mode: P(0); then ONE launch per block column: chain || prep (next chain's inputs) || trsm tiles of column k || syrk tiles of panel k-1.
  const int kNever = 1 << 30;
  int pending_double = kNever;      // set by the first launch of a deferred pair for the second one
  for (int k = 0; k + 1 < T; ++k) {
    const int n_prep = (k + 2 < T) ? 2 : 0, nT = std::max(T - k - 3, 0), nb = T - k - 1;
    // trailing update by panel k-1: 128x128 super-tiles of the region below (k+1,k+1) without the first one; as many
    // workgroups as fill the chip next to the chain, prep and trsm workgroups (one 16-wavefront workgroup per CU)
    const int ns = (nb + 1) / 2, nsup = (k >= 1) ? ns * (ns + 1) / 2 - 1 : 0;
    const int nW = std::min(nsup, 4 * kNumCUs);      // one super-tile per workgroup (those beyond the CU count start as trsm workgroups retire)
    // deferred pairs (see SyrkSuperTiles): while a launch has more super-tiles than the chip has CUs, every other launch
    // leaves the far block columns to the next one, which applies two panels to them at once
    int skip_from = kNever, double_from = pending_double;
    pending_double = kNever;
    if (sparse) {      // block-sparse: only the structurally non-zero tiles get a workgroup (no deferred pairs)
      const int32_t* h = aux->sparse_host.data();
      const int nTs = h[k + 1] - h[k], nWs = h[T + 1 + k + 1] - h[T + 1 + k];
      hipLaunchKernelGGL(k_column_step, dim3(1 + n_prep + nTs + nWs), dim3(kPanelThreads), 0, s, S, N, k, T, Linv_ws, xs, d_flag, kNever, kNever,
                         (const int32_t*)aux->sparse_lists, aux->sparse_base_rows + h[k], nTs, aux->sparse_base_sups + h[T + 1 + k]);
      continue;
    }
    if (double_from == kNever && k >= 1 && nsup > kDeferAbove && k + 6 < T && k + 2 < T - 1) { skip_from = k + 6; pending_double = k + 6; }
    hipLaunchKernelGGL(k_column_step, dim3(1 + n_prep + nT + nW), dim3(kPanelThreads), 0, s, S, N, k, T, Linv_ws, xs, d_flag, skip_from, double_from,
                       (const int32_t*)nullptr, 0, 0, 0);
  }
  LaunchBacksub(S, N, T, rhs_row, Linv_ws, x_out, d_flag, s, sparse ? (const uint8_t*)aux->sparse_nz : (const uint8_t*)nullptr);
  PP_HIP_TRY(hipGetLastError());
  return PP_OK;
}

// The launch structure is static for a given (S, N, ...): ~190 dependent launches on two streams.  It is
// captured ONCE into a hipGraph and replayed per LM iteration (host launch cost would otherwise bound
// the ~35 us steps of the critical path).  Falls back to eager enqueueing if capture is unavailable.
// Allocations, uploads and the graph capture of one handle must not run beside the capture of another host thread's handle
// (a hipMalloc from thread B invalidates thread A's capture in progress): one process-wide lock around both.
std::recursive_mutex& DeviceSetupMutex() { return g_setup_mutex; }

bool CholeskyWantsFactorArray(const CholeskyAux* aux, int N) { return aux && UseTasks(aux->mode, N / kNB); }

// the per-size device lists (task list, block-sparse lists): at buffer set-up, so that a solve allocates nothing
int CholeskyPrepare(CholeskyAux* aux, int N, bool has_factor_array, hipStream_t s) {
  if (!aux) return PP_OK;
  std::lock_guard<std::recursive_mutex> lock(g_setup_mutex);
  if (aux->tile_nz) { const int rc = EnsureSparseLists(aux, N / kNB, s); if (rc) return rc; }
  if (has_factor_array && UseTasks(aux->mode, N / kNB)) { const int rc = EnsureTaskList(aux, N / kNB, s); if (rc) return rc; }      // (dense or block-sparse: one launch up to 128 block columns)
  return PP_OK;
}

int CholeskySolveAugmented(double* S, int N, int rhs_row, double* Linv_ws, double* Lfac, double* x_out, int32_t* d_flag, hipStream_t s, CholeskyAux* aux) {
  { const int rc = CholeskyPrepare(aux, N, Lfac != nullptr, s); if (rc) return rc; }      // (a no-op after the first call for this size)
  // task mode is three launches: nothing to gain from a graph, and a capture is one thing less that can collide with whatever
  // other host threads do on the device meanwhile (a device-wide synchronize in another thread fails while any stream captures)
  const bool three_launches = aux && Lfac && UseTasks(aux->mode, N / kNB);
  if (aux && aux->use_graph && !three_launches && !(UseSmallCholesky(N) && x_out)) {
    const bool same = aux->graph_exec && aux->g_S == S && aux->g_N == N && aux->g_rhs == rhs_row && aux->g_Linv == Linv_ws &&
                      aux->g_x == x_out && aux->g_flag == d_flag && aux->g_stream == s && aux->g_mode == aux->mode && aux->g_Lfac == Lfac &&
                      aux->g_sparse == (aux->sparse_lists != nullptr);
    if (!same) {
      std::lock_guard<std::recursive_mutex> lock(g_setup_mutex);
      if (aux->graph_exec) { (void)hipGraphExecDestroy(aux->graph_exec); aux->graph_exec = nullptr; }
      hipGraph_t graph = nullptr;
      if (hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed) == hipSuccess) {
        const int rc = EnqueueCholesky(S, N, rhs_row, Linv_ws, Lfac, x_out, d_flag, s, aux);
        const hipError_t e = hipStreamEndCapture(s, &graph);
        if (rc == PP_OK && e == hipSuccess && graph && hipGraphInstantiate(&aux->graph_exec, graph, nullptr, nullptr, 0) == hipSuccess) {
          aux->g_S = S; aux->g_N = N; aux->g_rhs = rhs_row; aux->g_Linv = Linv_ws; aux->g_x = x_out; aux->g_flag = d_flag; aux->g_stream = s;
          aux->g_mode = aux->mode; aux->g_Lfac = Lfac; aux->g_sparse = aux->sparse_lists != nullptr;
        } else {
          aux->graph_exec = nullptr;
          aux->use_graph = false;   // do not retry
          (void)hipGetLastError();
        }
        if (graph) (void)hipGraphDestroy(graph);
      } else {
        aux->use_graph = false;
        (void)hipGetLastError();
      }
    }
    if (aux->graph_exec) {
      PP_HIP_TRY(hipGraphLaunch(aux->graph_exec, s));
      return PP_OK;
    }
  }
  return EnqueueCholesky(S, N, rhs_row, Linv_ws, Lfac, x_out, d_flag, s, aux);
}

int CholeskyAuxCreate(CholeskyAux* aux) {
  // PPSFM_CHOL_MODE: "columns" = one launch per block column; "tasks" = the whole factorisation as one launch (k_cholesky_tasks,
  // bit-identical results); unset / "auto": tasks up to kTaskAutoMaxT block columns (0.73 against 0.84 ms at n = 3000, 0.35 against
  // 0.40 ms at n = 1500, 1.24 against 1.38 ms at n = 4030), per-column launches above (there the trailing update is the bound and
  // the per-column grid runs it in bigger, better balanced pieces)
  if (aux->mode < 0) {
    const char* e = getenv("PPSFM_CHOL_MODE");
    aux->mode = !e ? 2 : ((e[0] == 't' || e[0] == '1') ? 1 : ((e[0] == 'c' || e[0] == '0') ? 0 : 2));
  }
  { const char* e = getenv("PPSFM_CHOL_GRAPH"); if (e && atoi(e) == 0) aux->use_graph = false; }
  { const char* e = getenv("PPSFM_CHOL_TEST_DROP_TASKS"); aux->test_drop_tasks = e && atoi(e) != 0; }
  return PP_OK;
}
void CholeskyAuxDestroy(CholeskyAux* aux) {
  if (aux->graph_exec) (void)hipGraphExecDestroy(aux->graph_exec);
  aux->graph_exec = nullptr;
  if (aux->tasks) (void)hipFree(aux->tasks);
  if (aux->tasks_nz) (void)hipFree(aux->tasks_nz);
  if (aux->scratch) PoolDeviceFree(aux->scratch);
  aux->scratch = nullptr; aux->scratch_tiles = 0;
  aux->tasks = nullptr; aux->tasks_T = 0; aux->tasks_nz = nullptr; aux->tasks_src_nz = nullptr;
  if (aux->sparse_lists) (void)hipFree(aux->sparse_lists);
  if (aux->sparse_nz) (void)hipFree(aux->sparse_nz);
  aux->sparse_lists = nullptr; aux->sparse_nz = nullptr; aux->sparse_T = 0;
}

}  // namespace ppsfm

using namespace ppsfm;

extern "C" int pp_cholesky_task_list(int32_t block_columns, int32_t* tasks, int64_t capacity, int64_t* count) try {
  PP_REQUIRE(block_columns >= 4 && block_columns <= kMaxSteps && count && (tasks || capacity == 0), "pp_cholesky_task_list: bad argument");
  std::vector<ChainTask> list;
  (void)PlanAndList(block_columns, nullptr, &list);
  *count = (int64_t)list.size();
  for (int64_t i = 0; i < (int64_t)list.size() && i < capacity; ++i) {
    tasks[4 * i] = list[i].type; tasks[4 * i + 1] = list[i].k; tasks[4 * i + 2] = list[i].a; tasks[4 * i + 3] = list[i].b;
  }
  return PP_OK;
} PP_API_CATCH("pp_cholesky_task_list")

extern "C" int pp_cholesky_task_list_sparse(int32_t block_columns, const uint8_t* tile_nz, uint8_t* map_out, int32_t* tasks, int64_t capacity, int64_t* count) try {
  PP_REQUIRE(block_columns >= 4 && block_columns <= kMaxSteps && count && tile_nz && (tasks || capacity == 0), "pp_cholesky_task_list_sparse: bad argument");
  const int T = block_columns;
  std::vector<uint8_t> closed(tile_nz, tile_nz + (size_t)T * T);
  (void)SymbolicTileFill(T, closed.data());
  std::vector<ChainTask> list;
  const ChainPlan plan = PlanChains(T, closed.data(), 1);      // ONE chain (pp_cholesky_task_plan: as many as the structure has)
  list = BuildTaskList(T, plan);
  if (map_out) std::memcpy(map_out, plan.map.data(), plan.map.size());
  *count = (int64_t)list.size();
  for (int64_t i = 0; i < (int64_t)list.size() && i < capacity; ++i) {
    const ChainTask& t = list[i];
    const int32_t row[7] = {t.type, t.k, t.a, t.b, t.w0, t.w1, t.w2};
    std::memcpy(tasks + 7 * i, row, sizeof(row));
  }
  return PP_OK;
} PP_API_CATCH("pp_cholesky_task_list_sparse")

extern "C" int pp_cholesky_task_plan(int32_t block_columns, const uint8_t* tile_nz, int32_t max_chains, uint8_t* map_out, int32_t* tasks, int64_t capacity,
                                     int64_t* count, int32_t* chains_out, int32_t* time_out, int32_t* rho1_out, int32_t* verified) try {
  PP_REQUIRE(block_columns >= 4 && block_columns <= kMaxSteps && count && tile_nz && (tasks || capacity == 0), "pp_cholesky_task_plan: bad argument");
  const int T = block_columns;
  std::vector<uint8_t> closed(tile_nz, tile_nz + (size_t)T * T);
  (void)SymbolicTileFill(T, closed.data());
  const ChainPlan plan = PlanChains(T, closed.data(), max_chains > 0 ? max_chains : kMaxChains);
  TaskListInfo info;
  const std::vector<ChainTask> list = BuildTaskList(T, plan, &info);
  if (verified) *verified = (info.fits && TaskListWaitsAreMet(T, plan, list)) ? 1 : 0;
  if (map_out) std::memcpy(map_out, plan.map.data(), plan.map.size());
  if (chains_out) {
    chains_out[0] = plan.cr.n;
    for (int c = 0; c < kMaxChains; ++c) { chains_out[1 + 3 * c] = plan.cr.begin[c]; chains_out[2 + 3 * c] = plan.cr.end[c]; chains_out[3 + 3 * c] = plan.cr.post[c]; }
  }
  for (int k = 0; k < T; ++k) { if (time_out) time_out[k] = plan.time[k]; if (rho1_out) rho1_out[k] = plan.rho1[k]; }
  *count = (int64_t)list.size();
  for (int64_t i = 0; i < (int64_t)list.size() && i < capacity; ++i) std::memcpy(tasks + 16 * i, &list[i], 16 * sizeof(int32_t));
  return PP_OK;
} PP_API_CATCH("pp_cholesky_task_plan")

extern "C" int pp_dense_cholesky_solve(int32_t n, const double* A, const double* b, double* x, int device, int32_t repeat,
                                       float* ms_per_solve) try {
  PP_REQUIRE(n > 0 && A && b && x && repeat >= 1, "pp_dense_cholesky_solve: bad argument");
  PP_HIP_TRY(hipSetDevice(device));
  const int N = ((n + 1 + 63) / 64) * 64;
  std::vector<double> h((size_t)N * N, 0.0);
  for (int i = 0; i < n; ++i)
    for (int j = 0; j <= i; ++j) h[(size_t)i * N + j] = A[(size_t)i * n + j];
  for (int j = 0; j < n; ++j) h[(size_t)n * N + j] = b[j];
  h[(size_t)n * N + n] = 1e100;
  for (int i = n + 1; i < N; ++i) h[(size_t)i * N + i] = 1.0;
  double *dS = nullptr, *dS0 = nullptr, *dLinv = nullptr, *dx = nullptr, *dL = nullptr;
  int32_t* dflag = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  hipStream_t strm = nullptr;
  CholeskyAux aux;
  int rc = PP_OK;
  auto cleanup = [&]() {
    CholeskyAuxDestroy(&aux);
    if (strm) (void)hipStreamDestroy(strm);
    if (dS) (void)hipFree(dS); if (dL) (void)hipFree(dL); if (dS0) (void)hipFree(dS0); if (dLinv) (void)hipFree(dLinv); if (dx) (void)hipFree(dx); if (dflag) (void)hipFree(dflag);
    if (e0) (void)hipEventDestroy(e0); if (e1) (void)hipEventDestroy(e1);
  };
  OnUnwind unwind{[&] { cleanup(); }};
#define TRYH(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { SetLastError("%s: %s", #expr, hipGetErrorString(e_)); cleanup(); return PP_ERR_HIP; } } while (0)
  if ((rc = DeviceAlloc(&dS, (size_t)N * N)) || (rc = DeviceAlloc(&dS0, (size_t)N * N)) || (rc = DeviceAlloc(&dLinv, CholeskyWorkspaceDoubles(N))) ||
      (rc = DeviceAlloc(&dx, (size_t)N)) || (rc = DeviceAlloc(&dflag, 4)) || (rc = DeviceAlloc(&dL, (size_t)N * N))) { cleanup(); return rc; }
  TRYH(hipEventCreate(&e0)); TRYH(hipEventCreate(&e1));
  TRYH(hipStreamCreateWithFlags(&strm, hipStreamNonBlocking));
  if ((rc = CholeskyAuxCreate(&aux))) { cleanup(); return rc; }
  // block-sparse input (PPSFM_CHOL_SPARSE=0 disables): tiles of the lower triangle that are entirely zero and stay zero in the
  // factor get no workgroup (the reference switches to SPARSE_SCHUR above 50 images, src/optim/bundle_adjustment.cc:275-286)
  std::vector<uint8_t> tile_nz;
  {
    const char* e = getenv("PPSFM_CHOL_SPARSE");
    const int T = N / kNB;
    if (!(e && atoi(e) == 0) && T >= 4) {
      tile_nz.assign((size_t)T * T, 0);
      for (int i = 0; i < N; ++i)
        for (int j = 0; j <= i; ++j)
          if (h[(size_t)i * N + j] != 0.0) tile_nz[(size_t)(i / kNB) * T + j / kNB] = 1;
      const int nnz = SymbolicTileFill(T, tile_nz.data());
      if (nnz < T * (T + 1) / 2) { aux.tile_nz = tile_nz.data(); aux.tile_T = T; }
    }
  }
  TRYH(hipMemcpy(dS0, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice));
  TRYH(hipMemset(dflag, 0, sizeof(int32_t) * 4));
  // per-solve times; the MEDIAN is reported.  This entry point creates its stream (a new hardware queue), its buffers and the kernels'
  // scratch per call, and the first dispatches on a fresh queue now and then take tens of milliseconds (observed: the untimed warm-up at
  // 70 ms instead of ~2, flag clear, three times a timed solve at 62-69 ms in ~60 calls of five to ten solves; never in 4000 LM
  // iterations on a handle, whose stream and buffers persist): one such solve must not pass for the factorisation's time.
  std::vector<float> times;
  for (int it = -1; it < repeat; ++it) {   // it = -1: untimed warm-up (graph capture + instantiate)
    if (it == -1 && repeat == 1) continue;
    TRYH(hipMemcpy(dS, dS0, sizeof(double) * h.size(), hipMemcpyDeviceToDevice));
    TRYH(hipDeviceSynchronize());
    TRYH(hipEventRecord(e0, strm));
    rc = CholeskySolveAugmented(dS, N, n, dLinv, dL, dx, dflag, strm, &aux);
    if (rc) { cleanup(); return rc; }
    TRYH(hipEventRecord(e1, strm));
    TRYH(hipEventSynchronize(e1));
    float ms = 0; TRYH(hipEventElapsedTime(&ms, e0, e1)); if (it >= 0) times.push_back(ms);
    if (getenv("PPSFM_CHOL_DEBUG_SLOW") && ms > 5.0f) {      // (how the outliers described at `times` were caught)
      int32_t f4[4] = {0, 0, 0, 0};
      (void)hipMemcpy(f4, dflag, sizeof(f4), hipMemcpyDeviceToHost);
      fprintf(stderr, "SLOW dense solve: n=%d it=%d ms=%.3f mode=%d last_used=%d flag=%d %d %d %d\n", n, it, ms, aux.mode, aux.last_used, f4[0], f4[1], f4[2], f4[3]);
    }
    if (aux.mode != 0) {      // a bounded wait of the one-launch factorisation ran out (bit 4): once more, with per-column launches from here on
      int32_t f = 0;
      TRYH(hipMemcpy(&f, dflag, sizeof(f), hipMemcpyDeviceToHost));
      if (f & 4) {
        aux.mode = 0;
        if (aux.graph_exec) { (void)hipGraphExecDestroy(aux.graph_exec); aux.graph_exec = nullptr; }
        TRYH(hipMemset(dflag, 0, sizeof(int32_t) * 4));
        if (it >= 0) times.pop_back();
        --it;
      }
    }
  }
  int32_t flag = 0;
  TRYH(hipMemcpy(&flag, dflag, sizeof(flag), hipMemcpyDeviceToHost));
  TRYH(hipMemcpy(x, dx, sizeof(double) * n, hipMemcpyDeviceToHost));
#undef TRYH
  cleanup();
  if (ms_per_solve) {
    std::sort(times.begin(), times.end());
    const size_t m = times.size();
    *ms_per_solve = m == 0 ? 0.0f : (m & 1 ? times[m / 2] : 0.5f * (times[m / 2 - 1] + times[m / 2]));
  }
  if (flag) { SetLastError("pp_dense_cholesky_solve: matrix is not positive definite"); return PP_ERR_NUMERIC; }
  return PP_OK;
} PP_API_CATCH("pp_dense_cholesky_solve")

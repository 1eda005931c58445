// Variable camera intrinsics inside the device LM solver (reference src/optim/bundle_adjustment.cc:490-528:
// refine_focal_length / refine_principal_point / refine_extra_params make a camera block variable, the constant
// parameters of such a block are held by a SubsetParameterization).
//
// The variable parameters of all intrinsics blocks form NI compact columns after the 6C pose columns of the reduced
// system.  Their rows of S are assembled by the same deterministic GATHER as the pose blocks, generalised to block
// pairs (row block = an intrinsics block of width <= 12; column block = a pose block or an intrinsics block):
//   S_AB = sum_o J_A,o^T J_B,o - sum_{(oi,oj) sharing a point} J_A,oi^T G_oi,oj J_B,oj ,  G = J_pt,oi (V+D^2)^-1 J_pt,oj^T
//   with G = T_oi X_oj^T (T_o = J_pt,o (V+D^2)^-1, X_o = J_pt,o: both in the per-observation records).
// FACTORED (round 4; rounds 1-3 listed the (oi, oj) entries, quadratic in the track length for a shared camera): the sum over oi does not
// depend on B or oj,  L_(p,A) = sum_{oi in (p,A)} J_A,oi^T T_oi  over the GROUP (p, A) = the observations of point p taken with camera A
// (k_intr_L), and the list of a pair holds (group, oj [, member]) entries: row a of (L X_oj^T - [member] J_A,oj^T) J_B,oj (k_schur_gen) - the
// direct term J^T J rides on the group's own observations.  Lists are cut into chunks of kGenChunk entries that run in parallel; a second kernel
// adds the chunk results of a pair in list order (deterministic, no atomics), a pair of one chunk is finished by its chunk.  The DIAGONAL blocks
// S_AA = sum_o J^T J - sum_(p,A) L R, R = sum_{o in (p,A)} X_o^T J_A,o, are assembled per group (k_intr_kk; the only blocks an iterative handle
// assembles: its preconditioner).  Column norms, gradient and right-hand side of the intrinsics columns are per-camera sums over its
// observations, chunked the same way.
#include "ba_impl.hpp"

namespace ppsfm {

// ---- per-camera sums over the observations of an intrinsics block -------------------------------------------
// MODE 0 (after an evaluation):  sum_o Jk[:,j]^2  and  sum_o Jk[:,j] . r_o           (unscaled ambient J, 12 + 12 sums)
// MODE 1 (per trial radius):     sum_o JkS[:,j] . (J_pt,o (s_p * vb_p))              (compact scaled J, 12 sums)
// `stride`: row width of Jk (kCamStride: ambient rows or JkS; less: the solver's compact camera Jacobians, EvalArgs::cam_col - columns beyond it do not exist)
template <int MODE>
__global__ __launch_bounds__(256) void k_intr_sums(const int32_t* __restrict__ chunk, const int32_t* __restrict__ cam_obs, const int32_t* __restrict__ obs_point,
                                                   const double* __restrict__ Jk, const double* __restrict__ r, const double* __restrict__ Jpoint,
                                                   const double* __restrict__ scale_p, const double* __restrict__ vb, double* __restrict__ partial, int stride) {
  constexpr int NS = MODE == 0 ? 24 : 12;
  __shared__ double red[4][NS];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int e0 = chunk[3 * blockIdx.x + 1], e1 = chunk[3 * blockIdx.x + 2];
  double acc[NS];
#pragma unroll
  for (int i = 0; i < NS; ++i) acc[i] = 0.0;
  for (int e = e0 + (int)threadIdx.x; e < e1; e += 256) {
    const int o = cam_obs[e];
    const double* j = Jk + (size_t)2 * stride * o;
    double v0, v1;
    if (MODE == 0) { v0 = r[2 * (size_t)o]; v1 = r[2 * (size_t)o + 1]; }
    else {
      const int p = obs_point[o];
      const double* jx = Jpoint + 6 * (size_t)o;
      const double w0 = scale_p[3 * p] * vb[3 * (size_t)p], w1 = scale_p[3 * p + 1] * vb[3 * (size_t)p + 1], w2 = scale_p[3 * p + 2] * vb[3 * (size_t)p + 2];
      v0 = jx[0] * w0 + jx[1] * w1 + jx[2] * w2; v1 = jx[3] * w0 + jx[4] * w1 + jx[5] * w2;
    }
#pragma unroll
    for (int c = 0; c < 12; ++c) {
      if (c >= stride) break;
      const double a = j[c], b = j[stride + c];
      if (MODE == 0) { acc[c] += a * a + b * b; acc[12 + c] += a * v0 + b * v1; }
      else acc[c] += a * v0 + b * v1;
    }
  }
#pragma unroll
  for (int i = 0; i < NS; ++i) acc[i] = WaveSum(acc[i]);
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < NS; ++i) red[wv][i] = acc[i];
  }
  __syncthreads();
  if (threadIdx.x < NS) partial[(size_t)blockIdx.x * 24 + threadIdx.x] = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
}

// one wavefront per intrinsics block: chunk partials added in chunk order
template <int MODE>
__global__ __launch_bounds__(64) void k_intr_sums_reduce(int C, const int32_t* __restrict__ cam_chunk, const int32_t* __restrict__ intr_off,
                                                         const int32_t* __restrict__ intr_nv, const int32_t* __restrict__ intr_col, const double* __restrict__ partial,
                                                         double* __restrict__ cnI, double* __restrict__ gc, const double* __restrict__ scale_c,
                                                         double* __restrict__ S, int N, int rhs_row, int add_diagonal, double* __restrict__ rhs_out,
                                                         const int32_t* __restrict__ spos, int compact) {
  const int k = blockIdx.x, t = threadIdx.x;
  const int off = intr_off[k];
  if (off < 0) return;
  constexpr int NS = MODE == 0 ? 24 : 12;
  if (t >= NS) return;
  double s = 0.0;
  {
    int c = cam_chunk[k];
    const int ce = cam_chunk[k + 1];
    for (; c + 8 <= ce; c += 8) {      // eight partials in flight, added in chunk order (86 chunks for a camera shared by 1100 images: 22 us one load at a time)
      double v8[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v8[u] = partial[(size_t)(c + u) * 24 + t];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v8[u];
    }
    for (; c < ce; ++c) s += partial[(size_t)c * 24 + t];
  }
  if (MODE == 0) {
    const int col = compact ? ((t % 12) < intr_nv[k] ? (t % 12) : -1) : intr_col[k * kCamStride + (t % 12)];    // ambient parameter -> compact column (compact sums: already there)
    if (col < 0) return;
    if (t < 12) cnI[off + col] = s; else gc[6 * C + off + col] = s;
  } else {
    // compact column t of the block (JkS is already compact and scaled)
    if (t >= intr_nv[k]) return;
    const int idx = 6 * C + off + t;
    const double own = add_diagonal ? -scale_c[idx] * gc[idx] : 0.0;
    if (rhs_out) rhs_out[idx] = own - s;      // (an iterative handle: the right-hand side vector of its conjugate gradients; there is no S)
    else S[(size_t)rhs_row * N + spos[idx]] = own - s;
  }
}

__global__ __launch_bounds__(256) void k_intr_scale(int C, int NI, int jacobi, const double* __restrict__ cnI, double* __restrict__ scale_c) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < NI) scale_c[6 * C + i] = jacobi ? 1.0 / (1.0 + sqrt(cnI[i])) : 1.0;
}
__global__ __launch_bounds__(256) void k_intr_diag(int C, int NI, double dmin, double dmax, const double* __restrict__ cnI, const double* __restrict__ scale_c,
                                                   double* __restrict__ diag_c) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < NI) { const double s = scale_c[6 * C + i]; diag_c[6 * C + i] = fmin(fmax(s * s * cnI[i], dmin), dmax); }
}

// JkS[o] = compact, scaled intrinsics Jacobian of observation o: rows of 12, columns [0, nv) used, zero beyond.  Only the used columns are written: the rest
// of the array is zero since pp_ba_create (a camera's nv never changes) - 32 instead of 192 bytes per observation at two variable parameters.
__global__ __launch_bounds__(256) void k_intr_prepare(int64_t M, int C, const int32_t* __restrict__ obs_cam, const int32_t* __restrict__ intr_off,
                                                      const int32_t* __restrict__ intr_col, const double* __restrict__ Jcam,
                                                      const double* __restrict__ scale_c, double* __restrict__ JkS, int compact_stride, const int32_t* __restrict__ intr_nv) {
  const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (o >= M) return;
  const int k = obs_cam[o] >> 4;
  const int off = intr_off[k];
  if (off < 0) return;
  double* dst = JkS + (size_t)2 * kCamStride * o;
  if (compact_stride > 0) {      // compact rows (the solver's evaluations): the columns are in place
    const double* j = Jcam + (size_t)2 * compact_stride * o;
    const int nv = intr_nv[k];
    for (int col = 0; col < nv; ++col) { const double s = scale_c[6 * C + off + col]; dst[col] = j[col] * s; dst[kCamStride + col] = j[compact_stride + col] * s; }
  } else {
    const double* j = Jcam + (size_t)2 * kCamStride * o;
    for (int c = 0; c < kCamStride; ++c) {
      const int col = intr_col[k * kCamStride + c];
      if (col >= 0) { const double s = scale_c[6 * C + off + col]; dst[col] = j[c] * s; dst[kCamStride + col] = j[kCamStride + c] * s; }
    }
  }
}

// L_g = sum_{o in group g} J^_k,o^T T_o  (n_v x 3; row a in lane a): twelve lanes per group, five groups per wavefront.  Per trial radius (T holds (V + D)^-1).
__global__ __launch_bounds__(256) void k_intr_L(int64_t num_groups, const int32_t* __restrict__ grp_start, const int32_t* __restrict__ grp_obs,
                                                const double* __restrict__ rec, const double* __restrict__ JkS, double* __restrict__ Lbuf) {
  const int lane = threadIdx.x & 63;
  const int slot = lane / 12, ar = lane % 12;
  const int64_t g = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 5 + slot;
  if (slot >= 5 || g >= num_groups) return;
  double l0 = 0.0, l1 = 0.0, l2 = 0.0;
  for (int e = grp_start[g]; e < grp_start[g + 1]; ++e) {
    const int o = grp_obs[e];
    const double2* q = reinterpret_cast<const double2*>(RecT(rec, (size_t)o));
    const double2 t0 = q[0], t1 = q[1], t2 = q[2];      // T rows (t0.x t0.y t1.x | t1.y t2.x t2.y)
    const double ja = JkS[(size_t)2 * kCamStride * o + ar], jb = JkS[(size_t)2 * kCamStride * o + kCamStride + ar];
    l0 += ja * t0.x + jb * t1.y; l1 += ja * t0.y + jb * t2.x; l2 += ja * t1.x + jb * t2.y;
  }
  double* dst = Lbuf + 36 * (size_t)g + 3 * ar;
  dst[0] = l0; dst[1] = l1; dst[2] = l2;
}

// ---- generic block pairs: TWELVE lanes per chunk (lane = row of the <=12-row block), five chunks per wavefront ----
// entries in FACTORED form (pp_ba_create): (group g, observation oj [| member of g]) - row a of  L_g X_oj^T - [member] J^_k,oj^T  times J_B,oj
// where a finished block goes (shared by the chunk kernels - a pair of ONE chunk is finished by the chunk itself - and k_schur_gen_reduce)
struct GenTarget {
  const int32_t* pair_chunk; const double* diag_c; double inv_radius; int add_diagonal; double* S; int N; int compact_base;
  const int32_t* spos;      // position in S of every vector column (pp_ba_impl::spos); a block above the diagonal there is stored transposed
};
__device__ __forceinline__ void StoreGenEntry(const GenTarget& g, const int32_t* __restrict__ pair, int pr, int a, int b, double sum) {
  const int roff = pair[4 * pr], rw = pair[4 * pr + 1], coff = pair[4 * pr + 2], cw = pair[4 * pr + 3] & 255;
  if (a >= rw || b >= cw) return;
  double v = -sum;
  if (roff == coff && a == b && g.add_diagonal) v += g.diag_c[roff + a] * g.inv_radius;
  // compact_base >= 0 (an iterative handle, diagonal pairs only): row i of the intrinsics columns holds its block's row, twelve wide
  if (g.compact_base >= 0) g.S[(size_t)(roff - g.compact_base + a) * 12 + b] = v;
  else {
    const int sr = g.spos[roff] + a, sc = g.spos[coff] + b;
    if (sr >= sc) g.S[(size_t)sr * g.N + sc] = v; else g.S[(size_t)sc * g.N + sr] = v;      // (the lower triangle is what the factorisation reads)
  }
}
__global__ __launch_bounds__(256) void k_schur_gen(int64_t num_chunks, const int32_t* __restrict__ chunk, const int32_t* __restrict__ pair,
                                                   const int32_t* __restrict__ entries, const double* __restrict__ rec, const double* __restrict__ JkS,
                                                   const double* __restrict__ Lbuf, double* __restrict__ partial, GenTarget tg) {
  const int lane = threadIdx.x & 63;
  const int slot = lane / 12, ar = lane % 12;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t ch = wave * 5 + slot;
  if (slot >= 5 || ch >= num_chunks) return;
  const int pr = chunk[3 * ch], e0 = chunk[3 * ch + 1], e1 = chunk[3 * ch + 2];
  const bool col_intr = (pair[4 * pr + 3] >> 8) != 0;
  double acc[12];
#pragma unroll
  for (int b = 0; b < 12; ++b) acc[b] = 0.0;
  for (int e = e0; e < e1; ++e) {
    const int2 en = *reinterpret_cast<const int2*>(entries + 2 * (size_t)e);
    const int2 oo = make_int2(en.x, en.y & 0x7fffffff);      // (group, observation)
    const double2* qj = reinterpret_cast<const double2*>(RecX(rec, (size_t)oo.y));
    const double2 x0 = qj[0], x1 = qj[1], x2 = qj[2];      // X rows (x0.x x0.y x1.x | x1.y x2.x x2.y)
    const double* lg = Lbuf + 36 * (size_t)oo.x + 3 * ar;
    const double l0 = lg[0], l1 = lg[1], l2 = lg[2];
    double h0 = l0 * x0.x + l1 * x0.y + l2 * x1.x, h1 = l0 * x1.y + l1 * x2.x + l2 * x2.y;      // row a of L_g X_oj^T
    if (en.y < 0) { h0 -= JkS[(size_t)2 * kCamStride * oo.y + ar]; h1 -= JkS[(size_t)2 * kCamStride * oo.y + kCamStride + ar]; }      // the direct term J^T J rides on the group's own observations
    if (col_intr) {
      const double2* pj = reinterpret_cast<const double2*>(JkS + (size_t)2 * kCamStride * oo.y);
#pragma unroll
      for (int b2 = 0; b2 < 6; ++b2) {
        const double2 u = pj[b2], v = pj[6 + b2];
        acc[2 * b2] += h0 * u.x + h1 * v.x; acc[2 * b2 + 1] += h0 * u.y + h1 * v.y;
      }
    } else {
      const double2* pj = reinterpret_cast<const double2*>(RecJ(rec, (size_t)oo.y));
#pragma unroll
      for (int b2 = 0; b2 < 3; ++b2) {
        const double2 u = pj[b2], v = pj[3 + b2];
        acc[2 * b2] += h0 * u.x + h1 * v.x; acc[2 * b2 + 1] += h0 * u.y + h1 * v.y;
      }
    }
  }
  if (tg.pair_chunk[pr + 1] - tg.pair_chunk[pr] == 1) {      // the pair's only chunk: the block is finished here (a camera per image: 250 000 such pairs)
#pragma unroll
    for (int b = 0; b < 12; ++b) StoreGenEntry(tg, pair, pr, ar, b, acc[b]);
    return;
  }
  double* dst = partial + (size_t)ch * 144 + ar * 12;
#pragma unroll
  for (int b = 0; b < 12; ++b) dst[b] = acc[b];
}

// ---- the DIAGONAL blocks alone (iterative handles: the preconditioner's intrinsics blocks), from (point, camera) groups ----
//   S_kk = sum_o J_k,o^T J_k,o - sum_{groups (p,k)} L R,   L = sum_{o in group} J_k,o^T T_o  (n_v x 3),   R = sum_{o in group} X_o^T J_k,o  (3 x n_v)
// - what the (k, k) pair list adds up entry by entry (sum over (oi, oj) of J^T T_oi X_oj^T J, minus the identity on (o, o)), factored: linear in
// the observations where the list of a camera shared by every image is quadratic in the track lengths (1.4 M entries at 1100 images / tracks of
// 8: k_schur_gen 330 us + 5500 chunks to add up; here 176 k observations).  Twelve lanes per group as in k_schur_gen (lane a = row a of
// the block); the partial blocks NEGATED so that k_schur_gen_reduce's "- sum" gives S_kk.  T_o = 0 for a constant point: its group adds J^T J only.
__global__ __launch_bounds__(256) void k_intr_kk(int64_t num_chunks, const int32_t* __restrict__ chunk, const int32_t* __restrict__ grp_start,
                                                 const int32_t* __restrict__ grp_obs, const double* __restrict__ rec, const double* __restrict__ JkS,
                                                 double* __restrict__ partial, const int32_t* __restrict__ pair, GenTarget tg) {
  // one WORKGROUP per chunk (~320 observations): its twenty 12-lane slots take the chunk's groups in turn (slot s: groups g0 + s, g0 + s + 20, ..),
  // their blocks are added in slot order through LDS - one partial block per chunk for k_schur_gen_reduce (a slot per 64-observation chunk walked
  // its observations for 97 us with two wavefronts per CU, and left 2750 partial blocks to add up)
  __shared__ double blocks[20][144];
  const int lane = threadIdx.x & 63;
  const int slot = lane / 12, ar = lane % 12;
  const int sid = (int)(threadIdx.x >> 6) * 5 + slot;
  const bool on = slot < 5;      // (lanes 60-63 walk no groups but take part in the shuffles below)
  const int64_t ch = blockIdx.x;
  const int g0 = chunk[3 * ch + 1], g1 = chunk[3 * ch + 2];
  const int base = 12 * slot;
  double acc[12];
#pragma unroll
  for (int b = 0; b < 12; ++b) acc[b] = 0.0;
  const int rounds = (g1 - g0 + 19) / 20;
  for (int gi = 0; gi < rounds; ++gi) {
    const int g = g0 + sid + 20 * gi;
    const bool live = on && g < g1;
    double l0 = 0.0, l1 = 0.0, l2 = 0.0, r0 = 0.0, r1 = 0.0, r2 = 0.0;
    if (live) {
      for (int e = grp_start[g]; e < grp_start[g + 1]; ++e) {
        const int o = grp_obs[e];
        const double2* q = reinterpret_cast<const double2*>(RecT(rec, (size_t)o));
        const double2* x = reinterpret_cast<const double2*>(RecX(rec, (size_t)o));
        const double2 t0 = q[0], t1 = q[1], t2 = q[2];      // T rows (t0.x t0.y t1.x | t1.y t2.x t2.y)
        const double2 x0 = x[0], x1 = x[1], x2 = x[2];      // X rows (x0.x x0.y x1.x | x1.y x2.x x2.y)
        const double* j = JkS + (size_t)2 * kCamStride * o;
        const double ja = j[ar], jb = j[kCamStride + ar];
        l0 += ja * t0.x + jb * t1.y; l1 += ja * t0.y + jb * t2.x; l2 += ja * t1.x + jb * t2.y;
        r0 += x0.x * ja + x1.y * jb; r1 += x0.y * ja + x2.x * jb; r2 += x1.x * ja + x2.y * jb;
        const double2* jj = reinterpret_cast<const double2*>(j);
#pragma unroll
        for (int b2 = 0; b2 < 6; ++b2) {
          const double2 u = jj[b2], v = jj[6 + b2];
          acc[2 * b2] -= ja * u.x + jb * v.x; acc[2 * b2 + 1] -= ja * u.y + jb * v.y;      // (negated: the direct term enters S with a plus)
        }
      }
    }
#pragma unroll
    for (int b = 0; b < 12; ++b) {      // + L R: row a of L (this lane) with column b of R (lane b of the slot)
      const double rb0 = __shfl(r0, base + b, 64), rb1 = __shfl(r1, base + b, 64), rb2 = __shfl(r2, base + b, 64);
      if (live) acc[b] += l0 * rb0 + l1 * rb1 + l2 * rb2;
    }
  }
  if (on) {
#pragma unroll
    for (int b = 0; b < 12; ++b) blocks[sid][ar * 12 + b] = acc[b];
  }
  __syncthreads();
  if (threadIdx.x < 144) {
    double sum = 0.0;
#pragma unroll
    for (int q = 0; q < 20; ++q) sum += blocks[q][threadIdx.x];
    const int pr = chunk[3 * ch];
    if (tg.pair_chunk[pr + 1] - tg.pair_chunk[pr] == 1) StoreGenEntry(tg, pair, pr, threadIdx.x / 12, threadIdx.x % 12, sum);      // (the pair's only chunk)
    else partial[(size_t)ch * 144 + threadIdx.x] = sum;
  }
}

// one workgroup per pair: S block = -(sum of the chunk results in list order) (+ D^2 / radius on the diagonal of an
// intrinsics block's own pair, added once per group by the rank that owns the damping)
template <int kGroups>      // 7: 1024 threads, seven groups of 144 (pairs with thousands of chunks); 1: 256 threads, one plain loop (many pairs of a few chunks)
__global__ __launch_bounds__(kGroups == 7 ? 1024 : 256) void k_schur_gen_reduce(const int32_t* __restrict__ multi, const int32_t* __restrict__ pair,
                                                                                const double* __restrict__ partial, GenTarget tg) {
  // one workgroup per pair that is NOT finished by its only chunk: S block = -(sum of the chunk results in list order) (+ D^2 / radius on the
  // diagonal of an intrinsics block's own pair, added once per group by the rank that owns the damping).  Group g adds a contiguous part of the
  // pair's chunks (eight loads in flight), thread t < 144 then adds the group sums in group order: up to kGroups chunks the chunk order of a plain loop.
  __shared__ double sums[kGroups][144];
  const int pr = multi[blockIdx.x], g = threadIdx.x / 144, t = threadIdx.x % 144;
  const int c0 = tg.pair_chunk[pr], c1 = tg.pair_chunk[pr + 1];
  const int per = (c1 - c0 + kGroups - 1) / kGroups;
  if (g < kGroups) {
    double s = 0.0;
    int c = c0 + g * per;
    const int ce = c + per < c1 ? c + per : c1;
    for (; c + 8 <= ce; c += 8) {
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = partial[(size_t)(c + u) * 144 + t];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; c < ce; ++c) s += partial[(size_t)c * 144 + t];
    sums[g][t] = s;
  }
  __syncthreads();
  if (g != 0) return;
  double s = 0.0;
#pragma unroll
  for (int u = 0; u < kGroups; ++u) s += sums[u][t];
  StoreGenEntry(tg, pair, pr, t / 12, t % 12, s);
}

// ---- host launchers ----------------------------------------------------------------------------------------
int IntrSumsAfterEval(pp_ba_impl* h) {
  if (h->NI == 0) return PP_OK;
  hipStream_t s = h->stream;
  if (h->isum_num_chunks > 0)
    hipLaunchKernelGGL(k_intr_sums<0>, dim3((unsigned)h->isum_num_chunks), dim3(256), 0, s, h->isum_chunk, h->cam_obs, h->obs_point, h->Jcam, h->r, h->Jpoint,
                     h->scale_p, h->vb, h->isum_partial, h->jcam_compact ? h->jcam_stride : kCamStride);
  hipLaunchKernelGGL(k_intr_sums_reduce<0>, dim3(h->K), dim3(64), 0, s, h->C, h->isum_cam_chunk, h->intr_off, h->intr_nv, h->intr_col, h->isum_partial, h->cnI, h->gc,
                     h->scale_c, h->S, h->N, h->n_red, 0, (double*)nullptr, (const int32_t*)h->spos, h->jcam_compact ? 1 : 0);
  PP_HIP_TRY(hipGetLastError());
  return PP_OK;
}
int IntrScale(pp_ba_impl* h, int jacobi) {
  if (h->NI == 0) return PP_OK;
  hipLaunchKernelGGL(k_intr_scale, dim3(CeilDiv(h->NI, 256)), dim3(256), 0, h->stream, h->C, h->NI, jacobi, h->cnI, h->scale_c);
  PP_HIP_TRY(hipGetLastError());
  return PP_OK;
}
int IntrDiagonal(pp_ba_impl* h, double dmin, double dmax) {
  if (h->NI == 0) return PP_OK;
  hipLaunchKernelGGL(k_intr_diag, dim3(CeilDiv(h->NI, 256)), dim3(256), 0, h->stream, h->C, h->NI, dmin, dmax, h->cnI, h->scale_c, h->diag_c);
  PP_HIP_TRY(hipGetLastError());
  return PP_OK;
}
// the compact scaled intrinsics Jacobians of the current linearisation (what every assembly of the intrinsics' blocks reads)
int IntrScaledJacobians(pp_ba_impl* h) {
  if (h->NI == 0) return PP_OK;
  hipLaunchKernelGGL(k_intr_prepare, dim3(h->num_partials), dim3(256), 0, h->stream, h->M, h->C, h->obs_cam, h->intr_off, h->intr_col, h->Jcam, h->scale_c, h->JkS_intr,
                     h->jcam_compact ? h->jcam_stride : 0, (const int32_t*)h->intr_nv);
  PP_HIP_TRY(hipGetLastError());
  return PP_OK;
}
int IntrAssemble(pp_ba_impl* h, double inv_radius, int add_diagonal) {
  if (h->NI == 0 || h->intr_wide_nv > 0) return PP_OK;      // (a camera per image beside its pose columns: assembled with the pose blocks, ba_solver.hip k_schur_wide_*)
  hipStream_t s = h->stream;
  { const int rc = IntrScaledJacobians(h); if (rc) return rc; }
  if (h->isum_num_chunks > 0)
    hipLaunchKernelGGL(k_intr_sums<1>, dim3((unsigned)h->isum_num_chunks), dim3(256), 0, s, h->isum_chunk, h->cam_obs, h->obs_point, h->JkS_intr, h->r, h->Jpoint,
                     h->scale_p, h->vb, h->isum_partial, kCamStride);
  hipLaunchKernelGGL(k_intr_sums_reduce<1>, dim3(h->K), dim3(64), 0, s, h->C, h->isum_cam_chunk, h->intr_off, h->intr_nv, h->intr_col, h->isum_partial, h->cnI, h->gc,
                     h->scale_c, h->S, h->N, h->n_red, add_diagonal, h->iterative ? h->pcg_b : (double*)nullptr, (const int32_t*)h->spos, 0);
  GenTarget tg;
  tg.pair_chunk = h->gen_pair_chunk; tg.diag_c = h->diag_c; tg.inv_radius = inv_radius; tg.add_diagonal = add_diagonal;
  tg.S = h->iterative ? h->pcg_Scomp : h->S; tg.N = h->N; tg.compact_base = h->iterative ? 6 * h->C : -1; tg.spos = h->spos;
  if (h->gen_num_chunks > 0 && h->iterative)      // the diagonal blocks alone, from (point, camera) groups
    hipLaunchKernelGGL(k_intr_kk, dim3((unsigned)h->gen_num_chunks), dim3(256), 0, s, h->gen_num_chunks, h->gen_chunk, h->gen_entries,
                       h->gen_entries + h->gen_num_groups + 1, h->JpS, h->JkS_intr, h->gen_partial, h->gen_pair, tg);
  else if (h->gen_num_chunks > 0) {
    hipLaunchKernelGGL(k_intr_L, dim3(CeilDiv(h->gen_num_groups, (int64_t)20)), dim3(256), 0, s, h->gen_num_groups, h->gen_grp_start, h->gen_grp_obs, h->JpS, h->JkS_intr, h->gen_L);
    hipLaunchKernelGGL(k_schur_gen, dim3(CeilDiv(h->gen_num_chunks, 20)), dim3(256), 0, s, h->gen_num_chunks, h->gen_chunk, h->gen_pair, h->gen_entries, h->JpS,
                       h->JkS_intr, h->gen_L, h->gen_partial, tg);
  }
  if (!h->iterative && h->kk_num_pairs > 0) {      // the diagonal blocks: groups by camera, a workgroup per chunk (k_intr_kk), straight into S
    GenTarget tk = tg;
    tk.pair_chunk = h->kk_pair_chunk;
    if (h->kk_num_chunks > 0)
      hipLaunchKernelGGL(k_intr_kk, dim3((unsigned)h->kk_num_chunks), dim3(256), 0, s, h->kk_num_chunks, h->kk_chunk, h->kk_entries, h->kk_entries + h->kk_num_groups + 1,
                         h->JpS, h->JkS_intr, h->kk_partial, h->kk_pair, tk);
    if (h->kk_num_multi > 0) {
      if (h->kk_num_chunks > 16 * h->kk_num_pairs)
        hipLaunchKernelGGL(k_schur_gen_reduce<7>, dim3((unsigned)h->kk_num_multi), dim3(1024), 0, s, h->kk_multi, h->kk_pair, h->kk_partial, tk);
      else
        hipLaunchKernelGGL(k_schur_gen_reduce<1>, dim3((unsigned)h->kk_num_multi), dim3(256), 0, s, h->kk_multi, h->kk_pair, h->kk_partial, tk);
    }
  }
  if (h->gen_num_multi > 0) {
    // (a camera shared by every image: a few pairs of thousands of chunks - seven groups per pair; a camera per image: pairs of one chunk, finished by
    // the chunk kernels themselves, and a few of two or three)
    if (h->gen_num_chunks > 16 * h->gen_num_pairs)
      hipLaunchKernelGGL(k_schur_gen_reduce<7>, dim3((unsigned)h->gen_num_multi), dim3(1024), 0, s, h->gen_multi, h->gen_pair, h->gen_partial, tg);
    else
      hipLaunchKernelGGL(k_schur_gen_reduce<1>, dim3((unsigned)h->gen_num_multi), dim3(256), 0, s, h->gen_multi, h->gen_pair, h->gen_partial, tg);
  }
  PP_HIP_TRY(hipGetLastError());
  return PP_OK;
}

}  // namespace ppsfm

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
// Right-looking blocked algorithm with 64 x 64 blocks, one launch per phase (a dependent launch
// costs ~1.5 us on MI355X, cheaper than a grid-wide barrier):
//   k_potrf_block   diagonal block, one workgroup, LDS, 16-column inner panels
//   k_trsm_panel    rows below: one lane per row, the row lives in 128 VGPRs, L11 is broadcast
//                   from LDS (fully unrolled forward substitution)
//   k_syrk_tiles    trailing update C -= A_i A_j^T on v_mfma_f64_16x16x4_f64: 64x64 tile per
//                   workgroup, 4 wavefronts x (16 x 64) outputs, operands staged in LDS with a
//                   66-double row stride (conflict-free ds_read_b64 for the MFMA operand pattern)
//   k_trinv_blocks  (after the factorisation, all blocks in one launch) L_kk^-1 for the back substitution
//   k_backsub_step  x_k = L_kk^-T y_k, then y[0:k*64] -= L[k-block,:]^T x_k with coalesced row reads
// Roofline: the trailing update is fp64-MFMA bound (n^3/3 flop); the panel kernels are latency bound.
#include <vector>

#include "ba_impl.hpp"

namespace ppsfm {

constexpr int kNB = 64;
typedef double v4f64 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_potrf_block(double* __restrict__ S, int ld, int k, int32_t* __restrict__ flag) {
  __shared__ double A[kNB][kNB + 1];
  const int tid = threadIdx.x;
  const size_t base = (size_t)k * kNB * ld + (size_t)k * kNB;
  for (int idx = tid; idx < kNB * kNB; idx += 256) {
    const int r = idx >> 6, c = idx & 63;
    A[r][c] = S[base + (size_t)r * ld + c];
  }
  const int r = tid & 63, q = tid >> 6;
  for (int p = 0; p < 4; ++p) {
    const int c0 = 16 * p;
    for (int jj = 0; jj < 16; ++jj) {
      const int j = c0 + jj;
      __syncthreads();
      double d = A[j][j];
      if (!(d > 0.0)) { if (tid == 0) atomicOr(flag, 1); d = 1.0; }
      d = sqrt(d);
      const double inv = 1.0 / d;
      __syncthreads();
      if (q == 0) {
        if (r > j) A[r][j] *= inv;
        else if (r == j) A[j][j] = d;
      }
      __syncthreads();
      for (int c = j + 1 + q; c < c0 + 16; c += 4)
        if (r >= c) A[r][c] -= A[r][j] * A[c][j];
    }
    __syncthreads();
    // rank-16 update of the columns right of the inner panel
    for (int c = c0 + 16 + q; c < kNB; c += 4) {
      if (r >= c) {
        double s = 0.0;
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) s += A[r][c0 + kk] * A[c][c0 + kk];
        A[r][c] -= s;
      }
    }
  }
  __syncthreads();
  for (int idx = tid; idx < kNB * kNB; idx += 256) {
    const int rr = idx >> 6, c = idx & 63;
    if (c <= rr) S[base + (size_t)rr * ld + c] = A[rr][c];
  }
}

// x L^T = a for one 64-vector a held in registers; L (lower, factored) in LDS with stride 65.
__device__ __forceinline__ void SolveRowLt(double (&x)[kNB], const double (*L)[kNB + 1]) {
#pragma unroll
  for (int c = 0; c < kNB; ++c) {
    double s = x[c];
#pragma unroll
    for (int kk = 0; kk < c; ++kk) s -= x[kk] * L[c][kk];
    x[c] = s / L[c][c];
  }
}

// rows [ (k+1)*64, N ) of block column k:  X = A L11^-T.   One wavefront per 64 rows.
__global__ __launch_bounds__(64) void k_trsm_panel(double* __restrict__ S, int ld, int k) {
  __shared__ double L[kNB][kNB + 1];
  __shared__ double T[kNB][kNB + 1];
  const int lane = threadIdx.x;
  const size_t dbase = (size_t)k * kNB * ld + (size_t)k * kNB;
  const size_t row0 = (size_t)(k + 1 + blockIdx.x) * kNB;
  const size_t pbase = row0 * ld + (size_t)k * kNB;
  for (int rr = 0; rr < kNB; ++rr) {
    L[rr][lane] = S[dbase + (size_t)rr * ld + lane];
    T[rr][lane] = S[pbase + (size_t)rr * ld + lane];
  }
  __syncthreads();
  double x[kNB];
#pragma unroll
  for (int c = 0; c < kNB; ++c) x[c] = T[lane][c];
  SolveRowLt(x, L);
#pragma unroll
  for (int c = 0; c < kNB; ++c) T[lane][c] = x[c];
  __syncthreads();
  for (int rr = 0; rr < kNB; ++rr) S[pbase + (size_t)rr * ld + lane] = T[rr][lane];
}

// Linv[b] (64x64 row-major) = L_bb^-1 for every diagonal block, one wavefront per block.
__global__ __launch_bounds__(64) void k_trinv_blocks(const double* __restrict__ S, int ld, double* __restrict__ Linv) {
  __shared__ double L[kNB][kNB + 1];
  __shared__ double T[kNB][kNB + 1];
  const int lane = threadIdx.x, b = blockIdx.x;
  const size_t dbase = (size_t)b * kNB * ld + (size_t)b * kNB;
  for (int rr = 0; rr < kNB; ++rr) L[rr][lane] = S[dbase + (size_t)rr * ld + lane];
  __syncthreads();
  double x[kNB];
#pragma unroll
  for (int c = 0; c < kNB; ++c) x[c] = (c == lane) ? 1.0 : 0.0;
  SolveRowLt(x, L);                 // lane r now holds row r of L^-T = column r of L^-1
#pragma unroll
  for (int c = 0; c < kNB; ++c) T[c][lane] = x[c];   // transpose through LDS: T[c][r] = Linv[c][r]
  __syncthreads();
  double* out = Linv + (size_t)b * kNB * kNB;
  for (int rr = 0; rr < kNB; ++rr) out[rr * kNB + lane] = T[rr][lane];
}

// trailing update: tile (bi, bj), bi >= bj, both > k:  C -= A_i A_j^T with A_* = block column k
__global__ __launch_bounds__(256) void k_syrk_tiles(double* __restrict__ S, int ld, int k) {
  const int bj = k + 1 + blockIdx.x, bi = k + 1 + blockIdx.y;
  if (bj > bi) return;
  constexpr int kStride = kNB + 2;   // 66: lanes (l&15)*66 + (l>>4) hit 32 distinct bank pairs per half-wave
  __shared__ double As[kNB * kStride];
  __shared__ double Bs[kNB * kStride];
  const int tid = threadIdx.x;
  const size_t abase = (size_t)bi * kNB * ld + (size_t)k * kNB;
  const size_t bbase = (size_t)bj * kNB * ld + (size_t)k * kNB;
  for (int idx = tid; idx < kNB * kNB; idx += 256) {
    const int r = idx >> 6, c = idx & 63;
    As[r * kStride + c] = S[abase + (size_t)r * ld + c];
    Bs[r * kStride + c] = S[bbase + (size_t)r * ld + c];
  }
  __syncthreads();
  const int lane = tid & 63, w = tid >> 6;
  const int lr = lane & 15, lk = lane >> 4;
  v4f64 acc[4];
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) acc[ct] = (v4f64){0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int kk = 0; kk < 16; ++kk) {
    const double a = As[(16 * w + lr) * kStride + 4 * kk + lk];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      const double b = Bs[(16 * ct + lr) * kStride + 4 * kk + lk];
      acc[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[ct], 0, 0, 0);
    }
  }
  // D layout of v_mfma_f64_16x16x4_f64: lane l, register i -> row (l>>4) + 4 i, column l&15
  const size_t cbase = (size_t)bi * kNB * ld + (size_t)bj * kNB;
#pragma unroll
  for (int ct = 0; ct < 4; ++ct)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const size_t off = cbase + (size_t)(16 * w + lk + 4 * i) * ld + 16 * ct + lr;
      S[off] -= acc[ct][i];
    }
}

// one step of L^T x = y (y = row rhs_row of the factor): solves block k, then eliminates it from y[0 : 64k)
__global__ __launch_bounds__(256) void k_backsub_step(double* __restrict__ S, int ld, int k, int rhs_row, const double* __restrict__ Linv,
                                                      double* __restrict__ x_out, int n_out) {
  __shared__ double ys[kNB];
  __shared__ double xs[kNB];
  __shared__ double part[4][kNB];
  const int tid = threadIdx.x;
  double* y = S + (size_t)rhs_row * ld;
  if (tid < kNB) {
    const int col = k * kNB + tid;
    ys[tid] = (col < rhs_row) ? y[col] : 0.0;   // padding / the rhs row's own diagonal carry no unknown
  }
  __syncthreads();
  {  // x[i] = sum_j Linv[j][i] * y[j]  (L^-T y), 4 partial sums over j
    const int i = tid & 63, q = tid >> 6;
    const double* Lb = Linv + (size_t)k * kNB * kNB;
    double s = 0.0;
    for (int j = 16 * q; j < 16 * q + 16; ++j) s += Lb[j * kNB + i] * ys[j];
    part[q][i] = s;
  }
  __syncthreads();
  if (tid < kNB) {
    const double v = (part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid]);
    xs[tid] = v;
    const int col = k * kNB + tid;
    if (blockIdx.x == 0 && col < n_out) x_out[col] = v;
  }
  __syncthreads();
  const int c = blockIdx.x * 256 + tid;
  if (c < k * kNB) {
    double s = 0.0;
    const double* Lr = S + (size_t)k * kNB * ld + c;
#pragma unroll 8
    for (int r = 0; r < kNB; ++r) s += Lr[(size_t)r * ld] * xs[r];
    y[c] -= s;
  }
}

int CholeskySolveAugmented(double* S, int N, int rhs_row, double* Linv_ws, double* x_out, int32_t* d_flag, hipStream_t s) {
  const int T = N / kNB;
  for (int k = 0; k < T; ++k) {
    hipLaunchKernelGGL(k_potrf_block, dim3(1), dim3(256), 0, s, S, N, k, d_flag);
    const int nt = T - k - 1;
    if (nt > 0) {
      hipLaunchKernelGGL(k_trsm_panel, dim3(nt), dim3(64), 0, s, S, N, k);
      hipLaunchKernelGGL(k_syrk_tiles, dim3(nt, nt), dim3(256), 0, s, S, N, k);
    }
  }
  hipLaunchKernelGGL(k_trinv_blocks, dim3(T), dim3(64), 0, s, S, N, Linv_ws);
  for (int k = T - 1; k >= 0; --k) {
    const int grid = k > 0 ? CeilDiv((int64_t)k * kNB, 256) : 1;
    hipLaunchKernelGGL(k_backsub_step, dim3(grid), dim3(256), 0, s, S, N, k, rhs_row, Linv_ws, x_out, rhs_row);
  }
  PP_HIP_TRY(hipGetLastError());
  return PP_OK;
}

}  // namespace ppsfm

using namespace ppsfm;

extern "C" int pp_dense_cholesky_solve(int32_t n, const double* A, const double* b, double* x, int device, int32_t repeat,
                                       float* ms_per_solve) {
  PP_REQUIRE(n > 0 && A && b && x && repeat >= 1, "pp_dense_cholesky_solve: bad argument");
  PP_HIP_TRY(hipSetDevice(device));
  const int N = ((n + 1 + 63) / 64) * 64;
  std::vector<double> h((size_t)N * N, 0.0);
  for (int i = 0; i < n; ++i)
    for (int j = 0; j <= i; ++j) h[(size_t)i * N + j] = A[(size_t)i * n + j];
  for (int j = 0; j < n; ++j) h[(size_t)n * N + j] = b[j];
  h[(size_t)n * N + n] = 1e100;
  for (int i = n + 1; i < N; ++i) h[(size_t)i * N + i] = 1.0;
  double *dS = nullptr, *dS0 = nullptr, *dLinv = nullptr, *dx = nullptr;
  int32_t* dflag = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  int rc = PP_OK;
  auto cleanup = [&]() {
    if (dS) (void)hipFree(dS); if (dS0) (void)hipFree(dS0); if (dLinv) (void)hipFree(dLinv); if (dx) (void)hipFree(dx); if (dflag) (void)hipFree(dflag);
    if (e0) (void)hipEventDestroy(e0); if (e1) (void)hipEventDestroy(e1);
  };
#define TRYH(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { SetLastError("%s: %s", #expr, hipGetErrorString(e_)); cleanup(); return PP_ERR_HIP; } } while (0)
  if ((rc = DeviceAlloc(&dS, (size_t)N * N)) || (rc = DeviceAlloc(&dS0, (size_t)N * N)) || (rc = DeviceAlloc(&dLinv, (size_t)N * 64)) ||
      (rc = DeviceAlloc(&dx, (size_t)N)) || (rc = DeviceAlloc(&dflag, 4))) { cleanup(); return rc; }
  TRYH(hipEventCreate(&e0)); TRYH(hipEventCreate(&e1));
  TRYH(hipMemcpy(dS0, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice));
  TRYH(hipMemset(dflag, 0, sizeof(int32_t) * 4));
  float total = 0;
  for (int it = 0; it < repeat; ++it) {
    TRYH(hipMemcpy(dS, dS0, sizeof(double) * h.size(), hipMemcpyDeviceToDevice));
    TRYH(hipDeviceSynchronize());
    TRYH(hipEventRecord(e0, 0));
    rc = CholeskySolveAugmented(dS, N, n, dLinv, dx, dflag, 0);
    if (rc) { cleanup(); return rc; }
    TRYH(hipEventRecord(e1, 0));
    TRYH(hipEventSynchronize(e1));
    float ms = 0; TRYH(hipEventElapsedTime(&ms, e0, e1)); total += ms;
  }
  int32_t flag = 0;
  TRYH(hipMemcpy(&flag, dflag, sizeof(flag), hipMemcpyDeviceToHost));
  TRYH(hipMemcpy(x, dx, sizeof(double) * n, hipMemcpyDeviceToHost));
#undef TRYH
  cleanup();
  if (ms_per_solve) *ms_per_solve = total / repeat;
  if (flag) { SetLastError("pp_dense_cholesky_solve: matrix is not positive definite"); return PP_ERR_NUMERIC; }
  return PP_OK;
}

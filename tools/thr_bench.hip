// fp64 throughput per SIMD on MI355X: VALU v_fma_f64 vs v_mfma_f64_16x16x4, 1..4 wavefronts per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4f64 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ void k_thr(double* out, long long* ticks, int n) {
  double x[8];
  for (int j = 0; j < 8; ++j) x[j] = threadIdx.x * 1e-3 + j;
  v4f64 acc[4];
  for (int j = 0; j < 4; ++j) acc[j] = (v4f64){x[j], x[j], x[j], x[j]};
  const double y = 1.0000001;
  __syncthreads();
  const long long t0 = wall_clock64();
  for (int i = 0; i < n; ++i) {
    if (MODE == 0) {
#pragma unroll
      for (int j = 0; j < 8; ++j) x[j] = fma(x[j], y, 1e-9);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(y, 1e-3, acc[j], 0, 0, 0);
    }
  }
  __syncthreads();
  const long long t1 = wall_clock64();
  double s = 0;
  for (int j = 0; j < 8; ++j) s += x[j];
  for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][3];
  out[threadIdx.x] = s;
  if (threadIdx.x == 0) ticks[0] = t1 - t0;
}
__global__ void k_burn(double* out, int n) {
  double x = threadIdx.x * 1e-3, y = 1.0000001;
  for (int i = 0; i < n; ++i) x = fma(x, y, 1e-9);
  if (x == 123.0) out[0] = x;
}
int main() {
  double* out; long long* ticks; hipMalloc(&out, 1024 * 8); hipMalloc(&ticks, 16);
  hipLaunchKernelGGL(k_burn, dim3(4096), dim3(256), 0, 0, out, 400000); hipDeviceSynchronize();
  const int n = 20000;
  for (int waves : {4, 8, 16}) {
    long long t;
    hipLaunchKernelGGL(k_thr<0>, dim3(1), dim3(64 * waves), 0, 0, out, ticks, n); hipDeviceSynchronize();
    hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
    const double ns = t * 10.0;
    printf("VALU fma_f64: %2d waves/CU: %.2f ns per wave-instruction per SIMD, %.1f GFLOP/s per CU\n", waves, ns / (n * 8.0 * waves / 4), n * 8.0 * waves * 128 / ns);
    hipLaunchKernelGGL(k_thr<1>, dim3(1), dim3(64 * waves), 0, 0, out, ticks, n); hipDeviceSynchronize();
    hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
    const double ns2 = t * 10.0;
    printf("MFMA f64 16x16x4: %2d waves/CU: %.2f ns per MFMA per SIMD, %.1f GFLOP/s per CU\n", waves, ns2 / (n * 4.0 * waves / 4), n * 4.0 * waves * 2048 / ns2);
  }
  return 0;
}

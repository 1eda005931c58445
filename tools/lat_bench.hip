// Dependent-issue latencies of the instructions on the Cholesky panel's critical path (one wavefront, MI355X).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4f64 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ double ReadLane(double v, int src_lane) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, src_lane);
  hi = __builtin_amdgcn_readlane(hi, src_lane);
  return __hiloint2double(hi, lo);
}
template <int MODE>
__global__ __launch_bounds__(64) void k_lat(double* out, long long* ticks, int n, double seed) {
  double x = seed + threadIdx.x * 1e-3, y = 1.0000001;
  v4f64 acc = {x, x, x, x}, acc2 = acc, acc3 = acc, acc4 = acc;
  const long long t0 = wall_clock64();
  const long long c0 = clock64();
  for (int i = 0; i < n; ++i) {
    if (MODE == 0) { x = fma(x, y, 1e-9); }
    if (MODE == 1) { x = rsqrt(x) + 1.5; }
    if (MODE == 2) { acc = __builtin_amdgcn_mfma_f64_16x16x4f64(y, 1e-3, acc, 0, 0, 0); }
    if (MODE == 3) { const double s = ReadLane(x, i & 63); x = fma(x, 1e-9, s); }
    if (MODE == 4) { x = __builtin_amdgcn_rsq(x) + 1.5; }
    if (MODE == 5) { acc = __builtin_amdgcn_mfma_f64_16x16x4f64(y, 1e-3, acc, 0, 0, 0); const double s = ReadLane(acc[0], i & 63); y = s * 1e-9 + 1.0; }
    if (MODE == 6) { x = sqrt(x) + 1.5; }
    if (MODE == 7) { x = 1.0 / x + 1.5; }
    if (MODE == 8) {   // 4 independent accumulators
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(y, 1e-3, acc, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, 1e-3, acc2, 0, 0, 0);
      acc3 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, 1e-3, acc3, 0, 0, 0);
      acc4 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, 1e-3, acc4, 0, 0, 0);
    }
  }
  const long long c1 = clock64();
  const long long t1 = wall_clock64();
  out[threadIdx.x] = x + acc[0] + acc[1] + y + acc2[0] + acc3[1] + acc4[2];
  if (threadIdx.x == 0) { ticks[0] = t1 - t0; ticks[1] = c1 - c0; }
}
__global__ void k_burn(double* out, int n) {
  double x = threadIdx.x * 1e-3, y = 1.0000001;
  for (int i = 0; i < n; ++i) x = fma(x, y, 1e-9);
  if (x == 123.0) out[0] = x;
}
int main() {
  double* out; long long* ticks; hipMalloc(&out, 64 * 8); hipMalloc(&ticks, 16);
  const char* names[] = {"fma_f64 dependent", "rsqrt(double) ocml dependent", "mfma_f64_16x16x4 dependent", "readlane64 + fma", "v_rsq_f64 raw + add", "mfma -> readlane -> fma", "sqrt(double)", "1.0/x double", "4 independent mfma (per 4)"};
  const int n = 20000;
  { double* o; hipMalloc(&o, 8); hipLaunchKernelGGL(k_burn, dim3(4096), dim3(256), 0, 0, o, 400000); hipDeviceSynchronize(); }   // ~clock ramp
  for (int rep = 0; rep < 2; ++rep)
    for (int m = 0; m < 9; ++m) {
      switch (m) {
        case 0: hipLaunchKernelGGL(k_lat<0>, dim3(1), dim3(64), 0, 0, out, ticks, n, 1.0); break;
        case 1: hipLaunchKernelGGL(k_lat<1>, dim3(1), dim3(64), 0, 0, out, ticks, n, 1.0); break;
        case 2: hipLaunchKernelGGL(k_lat<2>, dim3(1), dim3(64), 0, 0, out, ticks, n, 1.0); break;
        case 3: hipLaunchKernelGGL(k_lat<3>, dim3(1), dim3(64), 0, 0, out, ticks, n, 1.0); break;
        case 4: hipLaunchKernelGGL(k_lat<4>, dim3(1), dim3(64), 0, 0, out, ticks, n, 1.0); break;
        case 5: hipLaunchKernelGGL(k_lat<5>, dim3(1), dim3(64), 0, 0, out, ticks, n, 1.0); break;
        case 6: hipLaunchKernelGGL(k_lat<6>, dim3(1), dim3(64), 0, 0, out, ticks, n, 1.0); break;
        case 7: hipLaunchKernelGGL(k_lat<7>, dim3(1), dim3(64), 0, 0, out, ticks, n, 1.0); break;
        case 8: hipLaunchKernelGGL(k_lat<8>, dim3(1), dim3(64), 0, 0, out, ticks, n, 1.0); break;
      }
      hipDeviceSynchronize();
      long long t[2]; hipMemcpy(t, ticks, 16, hipMemcpyDeviceToHost);
      if (rep == 1) printf("%-32s %7.1f ns/iter  %7.1f clk/iter (clock64)  => %.2f GHz\n", names[m], t[0] * 10.0 / n, (double)t[1] / n, (double)t[1] / (t[0] * 10.0));
    }
  return 0;
}

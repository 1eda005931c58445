// Issue rate of the fp64 VALU instructions the K4 scoring loop is made of (MI355X): cycles per wave-instruction
// per SIMD with 2 wavefronts per SIMD and 8 independent register chains per wavefront.
#include <hip/hip_runtime.h>
#include <cstdio>

#define OP3(name, text)                                                                                   \
  struct name {                                                                                           \
    static __device__ __forceinline__ void run(double& x, double y, double z) {                           \
      asm volatile(text : "+v"(x) : "v"(y), "v"(z));                                                      \
    }                                                                                                     \
    static const char* label() { return #name; }                                                          \
  };

OP3(fma_f64, "v_fma_f64 %0, %0, %1, %2")
OP3(mul_f64, "v_mul_f64 %0, %0, %1")
OP3(add_f64, "v_add_f64 %0, %0, %2")
OP3(rcp_f64, "v_rcp_f64 %0, %0")
OP3(rsq_f64, "v_rsq_f64 %0, %0")
OP3(sqrt_f64, "v_sqrt_f64 %0, %0")
struct div_scale_f64 {
  static __device__ __forceinline__ void run(double& x, double y, double z) { asm volatile("v_div_scale_f64 %0, vcc, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z) : "vcc"); }
  static const char* label() { return "div_scale_f64"; }
};
OP3(div_fmas_f64, "v_div_fmas_f64 %0, %0, %1, %2")
OP3(div_fixup_f64, "v_div_fixup_f64 %0, %0, %1, %2")
OP3(ldexp_f64, "v_ldexp_f64 %0, %0, 1")
struct cmp_f64 {
  static __device__ __forceinline__ void run(double& x, double y, double z) { asm volatile("v_cmp_le_f64 vcc, %0, %1" : : "v"(x), "v"(y) : "vcc"); }
  static const char* label() { return "cmp_f64"; }
};
struct cndmask_b32 {
  static __device__ __forceinline__ void run(double& x, double y, double z) {
    int lo = __double2loint(x);
    asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(lo) : "v"(__double2loint(y)));
    x = __hiloint2double(__double2hiint(x), lo);
  }
  static const char* label() { return "cndmask_b32"; }
};
struct add_u32 {
  static __device__ __forceinline__ void run(double& x, double y, double z) {
    int lo = __double2loint(x);
    asm volatile("v_add_u32 %0, %0, %1" : "+v"(lo) : "v"(__double2loint(y)));
    x = __hiloint2double(__double2hiint(x), lo);
  }
  static const char* label() { return "add_u32"; }
};

template <class Op>
__global__ void k_rate(double* out, long long* ticks, int n) {
  double x[8];
  for (int j = 0; j < 8; ++j) x[j] = 1.0 + threadIdx.x * 1e-3 + j;
  const double y = 1.0000001, z = 1e-9;
  __syncthreads();
  const long long t0 = wall_clock64();
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int j = 0; j < 8; ++j) Op::run(x[j], y, z);
  }
  __syncthreads();
  const long long t1 = wall_clock64();
  double s = 0;
  for (int j = 0; j < 8; ++j) s += x[j];
  out[threadIdx.x] = s;
  if (threadIdx.x == 0) ticks[0] = t1 - t0;
}

template <class Op>
void Measure(double* out, long long* ticks, double fma_ns) {
  const int n = 20000, waves = 8;
  long long t;
  hipLaunchKernelGGL(k_rate<Op>, dim3(1), dim3(64 * waves), 0, 0, out, ticks, n);
  hipDeviceSynchronize();
  hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
  const double ns = t * 10.0 / (n * 8.0 * waves / 4);
  printf("%-16s %.2f ns per wave-instruction per SIMD (%.2f x fma)\n", Op::label(), ns, fma_ns > 0 ? ns / fma_ns : 1.0);
}

__global__ void k_burn(double* out, int n) {
  double x = threadIdx.x * 1e-3, y = 1.0000001;
  for (int i = 0; i < n; ++i) x = fma(x, y, 1e-9);
  if (x == 123.0) out[0] = x;
}

int main() {
  double* out; long long* ticks; hipMalloc(&out, 1024 * 8); hipMalloc(&ticks, 16);
  hipLaunchKernelGGL(k_burn, dim3(4096), dim3(256), 0, 0, out, 400000); hipDeviceSynchronize();
  const int n = 20000, waves = 8;
  long long t;
  hipLaunchKernelGGL(k_rate<fma_f64>, dim3(1), dim3(64 * waves), 0, 0, out, ticks, n); hipDeviceSynchronize();
  hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
  const double fma_ns = t * 10.0 / (n * 8.0 * waves / 4);
  Measure<fma_f64>(out, ticks, fma_ns);
  Measure<mul_f64>(out, ticks, fma_ns);
  Measure<add_f64>(out, ticks, fma_ns);
  Measure<rcp_f64>(out, ticks, fma_ns);
  Measure<rsq_f64>(out, ticks, fma_ns);
  Measure<sqrt_f64>(out, ticks, fma_ns);
  Measure<div_scale_f64>(out, ticks, fma_ns);
  Measure<div_fmas_f64>(out, ticks, fma_ns);
  Measure<div_fixup_f64>(out, ticks, fma_ns);
  Measure<cmp_f64>(out, ticks, fma_ns);
  Measure<cndmask_b32>(out, ticks, fma_ns);
  Measure<ldexp_f64>(out, ticks, fma_ns);
  Measure<add_u32>(out, ticks, fma_ns);
  return 0;
}
